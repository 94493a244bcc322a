#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run ant ant MJH_X=0
run ant_kcv ant MJH_KCV_MAX_ENVS=100000000
run ant ant MJH_X=0
run mesh mesh MJH_X=0
