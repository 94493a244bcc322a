#!/bin/bash
run() { tag=$1; wl=$2; lib=$3; MJH_LIB=$PWD/mujoco-torch_amd/lib/$lib python3 bench.py --workload $wl --dtype f64 --batch ${BATCH:-4096} --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag"; }
for i in 1 2; do
for wl in mesh ant; do
run base_$wl $wl libmjhip.so
run f64w1_$wl $wl libmjhip_f64w1.so
done
done
