#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run h_default humanoid MJH_X=0
run h_con2 humanoid MJH_CON2_PACK=1
run h_default humanoid MJH_X=0
run h_con2 humanoid MJH_CON2_PACK=1
