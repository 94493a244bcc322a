#!/bin/bash
# GPU box: per-kernel times of one workload over batch sizes (HIP events of bench.py's per-kernel pass).  usage: tools/batch_sweep.sh <workload> "<B1 B2 ...>" [ENV=val ...]
wl=$1; bs=$2; shift 2
for b in $bs; do
  env "$@" python3 bench.py --workload $wl --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "B=$b"
done
