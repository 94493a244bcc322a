#!/bin/bash
# usage: tools/ablate.sh "<groups>" n1 n2 ...   builds lib/libmjhip_ab<n>.so with -DMJH_ABLATE=<n> (only the listed groups recompiled)
# The kernels carry no MJH_ABLATE guards: wrap the sections to be priced in `if (MJH_ABLATE != n) { ... }` for the experiment (profiles/r03/notes.md has the
# results of two such series), run tools/ablate_run.sh on the GPU box, and take the guards out again.  ABL_MACRO=<name> uses another macro (A/B of two code variants).
R=/root/repo; C=$R/mujoco-torch_amd/csrc
groups="$1"; shift
one() {
  n=$1
  rm -rf $C/build_ab$n; cp -r $C/build $C/build_ab$n
  (cd $C && MJH_BUILD_JOBS=2 MJH_BUILD_DIR=$C/build_ab$n MJH_BUILD_OUT=$R/mujoco-torch_amd/lib/libmjhip_ab$n.so MJH_BUILD_ONLY="$groups" ./build.sh -D${ABL_MACRO:-MJH_ABLATE}=$n > /dev/null 2>&1) && echo "built ab$n" || echo "FAILED ab$n"
  rm -rf $C/build_ab$n
}
export -f one; export R C groups ABL_MACRO
printf "%s\n" "$@" | xargs -P 4 -I{} bash -c 'one {}'
