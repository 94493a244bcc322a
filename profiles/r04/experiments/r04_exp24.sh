#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
for i in 1 2; do
run m_mfma mesh MJH_X=0
run m_valu mesh MJH_SOL2_MFMA=0
run a_mfma ant MJH_X=0
run a_valu ant MJH_SOL2_MFMA=0
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
