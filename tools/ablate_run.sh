#!/bin/bash
# GPU box: tools/ablate_run.sh "<workloads>" n1 n2 ...  -- bench digest of the production library, then of every lib/libmjhip_ab<n>.so
wl="$1"; shift
for w in $wl; do
  for n in base "$@" base; do
    lib=mujoco-torch_amd/lib/libmjhip_ab$n.so; [ $n = base ] && lib=mujoco-torch_amd/lib/libmjhip.so
    MJH_LIB=$PWD/$lib python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "ab$n"
  done
done
