#!/bin/bash
for n in libmjhip fl_ilp fl_bias0 fl_mem fl_relax fl_o2 libmjhip; do
  MJH_LIB=$PWD/mujoco-torch_amd/lib/$n.so python3 bench.py --workload humanoid --steps 200 --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$n"
done
