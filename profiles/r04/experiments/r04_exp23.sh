#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
for i in 1 2; do
run a_pad ant MJH_X=0
run a_nopad ant MJH_W16_PAD=0
run m_pad mesh MJH_X=0
run m_nopad mesh MJH_W16_PAD=0
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
