#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run mesh mesh MJH_X=0
run mesh mesh MJH_X=0
timeout 1200 python -m pytest tests -m gpu -x -q -k "convex or mesh or boxes or outlier or collision" 2>&1 | tail -4
