#!/bin/bash
# GPU box: step time across batch sizes (small batches are bound by the host issue rate and the latency floor of the five phases)
for w in humanoid cartpole; do for b in 1 64 512 1024 2048 4096; do
  python bench.py --workload $w --batch $b --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$w B=%5d  %7.1f us/step  %10d env-steps/s' % (j['config']['envs_per_gpu'], j['ms_per_step']*1e3, j['value']))"
done; done
