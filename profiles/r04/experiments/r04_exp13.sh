#!/bin/bash
for a in 0 1 2 3; do
  MJH_SENSOR_ABLATE=$a MJH_LIB=$PWD/mujoco-torch_amd/lib/abl.so python3 bench.py --workload ant --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "ablate_$a"
done
