#!/bin/bash
run() { tag=$1; wl=$2; shift 2; env "$@" python3 bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
python tools/fuzz_triage.py 2048 4:1:2029 15:1:1685 16:0:476 26:3:896 27:0:1525
run a_stage ant MJH_X=0
run a_direct ant MJH_SENSOR_STAGE=0
run a_stage16 ant MJH_SENSOR_WGS=16
run a_stage32 ant MJH_SENSOR_WGS=32
run a_stage64 ant MJH_SENSOR_WGS=64
timeout 1500 python -m pytest tests -m gpu -x -q -k "sensor or rig or ant" 2>&1 | tail -3
