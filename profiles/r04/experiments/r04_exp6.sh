#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
run h_cs humanoid MJH_X=0
run h_nocs humanoid MJH_FUSE_CS=0
run h_cs humanoid MJH_X=0
run h32k_cs humanoid32k MJH_X=0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
