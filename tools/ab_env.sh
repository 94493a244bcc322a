#!/bin/bash
# GPU box: A / B runs of one bench workload under different environment switches, interleaved and repeated (boxes drift by ~2 % within a call).
# usage: tools/ab_env.sh <workload> <steps> <repeats> "tag1:VAR=val VAR2=val" "tag2:" ...
wl=$1; steps=$2; reps=$3; shift 3
for r in $(seq 1 $reps); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    env $envs python3 bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run 2>/dev/null | python3 tools/benchline.py "$tag#$r"
  done
done
