#!/bin/bash
# round 4, experiment 2: kernel 13 default for the humanoid; B = 32768 with / without it; ant + mesh unchanged?
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run h_default humanoid MJH_X=0
run h_default humanoid MJH_X=0
run h32k_default humanoid32k MJH_X=0
run h32k_kcv_always humanoid32k MJH_KCV_MAX_ENVS=100000000
run h32k_crbpack0 humanoid32k MJH_CRB_PACK=0
run ant ant MJH_X=0
run mesh mesh MJH_X=0
