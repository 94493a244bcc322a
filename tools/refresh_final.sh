#!/bin/bash
# GPU box: refreshes the judged artifacts of a round -> gpurun_out/final/ (copy into profiles/rNN/final/ afterwards):
# per-kernel rocprofv3 stats of the bench command, the bench line itself (with cpu_baseline for the headline) and the PMC traffic.
mkdir -p gpurun_out/final
for w in humanoid ant mesh; do
  bash tools/prof_kernels.sh $w 50 > gpurun_out/final/${w}_kernels.txt 2>&1
  cp gpurun_out/${w}_kernel_stats.csv gpurun_out/final/${w}_kernel_stats.csv
  bash tools/hbm_traffic.sh $w > gpurun_out/final/${w}_traffic.txt 2>&1
  cp gpurun_out/hbm_traffic_${w}.json gpurun_out/final/
done
python bench.py 2>/dev/null | tail -1 > gpurun_out/final/bench_humanoid.json
python bench.py --workload ant 2>/dev/null | tail -1 > gpurun_out/final/bench_ant.json
python bench.py --workload mesh 2>/dev/null | tail -1 > gpurun_out/final/bench_mesh.json
cat gpurun_out/final/*_kernels.txt
cut -c1-140 gpurun_out/final/bench_*.json
