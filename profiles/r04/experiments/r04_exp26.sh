#!/bin/bash
for i in 1 2 3; do
for sp in 100 1000 3000; do
BENCH_SPIN_UP=$sp python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "spin$sp"
done
done
