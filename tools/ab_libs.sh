#!/bin/bash
# GPU box: tools/ab_libs.sh "<workloads>" lib1.so lib2.so ...  (paths relative to mujoco-torch_amd/lib/; "base" = libmjhip.so) -- bench digests, base first and last
wl="$1"; shift
for w in $wl; do
  for n in base "$@" base; do
    lib=mujoco-torch_amd/lib/$n; [ $n = base ] && lib=mujoco-torch_amd/lib/libmjhip.so
    MJH_LIB=$PWD/$lib python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$n"
  done
done
