#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
for i in 1 2 3; do
run h_all humanoid MJH_X=0
run h_two humanoid MJH_FUSE_ALL=0
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
