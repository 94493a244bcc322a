#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --dtype f64 --batch 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run ant64_reg ant MJH_X=0
run ant64_lds ant MJH_SOL2=0
run ant64_w16off ant MJH_SOL2_W16=0
run mesh64_reg mesh MJH_X=0
run mesh64_lds mesh MJH_SOL2=0
