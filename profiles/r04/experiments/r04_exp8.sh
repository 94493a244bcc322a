#!/bin/bash
for n in $@; do
  MJH_LIB=$PWD/mujoco-torch_amd/lib/cs_$n.so python3 bench.py --workload humanoid --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "cs_$n"
done
