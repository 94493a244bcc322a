#!/bin/bash
# usage: tools/kbuild.sh "<groups, e.g. 1d 1f | all>" [resource-usage grep pattern] [isa: group+dtype kernel-substring]
# Rebuilds the listed build groups of libmjhip.so (mjh_instances.h), prints the register allocation of the kernels matching the pattern and,
# with a third argument such as "1d Li1ELi64", the static ISA digest (tools/isa_stats.py) of those kernels.
R=/root/repo
C=$R/mujoco-torch_amd/csrc
if [ "$1" = all ]; then (cd $C && ./build.sh) || exit 1; else (cd $C && MJH_BUILD_ONLY="$1" ./build.sh) || exit 1; fi
[ -n "$2" ] && grep -E "$2" $R/mujoco-torch_amd/lib/resource_usage.txt
if [ -n "$3" ]; then
  set -- $3
  g=${1%[df]}; t=${1: -1}; real=double; [ "$t" = f ] && real=float
  mkdir -p /tmp/asm
  (cd $C && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -S --cuda-device-only -DMJH_INST_GROUP=$g -DMJH_INST_REAL=$real -o /tmp/asm/inst$1.s mjh_inst.hip 2>/dev/null)
  python3 $R/tools/isa_stats.py /tmp/asm/inst$1.s "$2"
fi
