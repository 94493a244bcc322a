#!/bin/bash
# usage (GPU box): tools/issue_mix.sh <workload> -> gpurun_out/issue_mix_<workload>.txt
# Instruction mix and SIMD occupancy of each kernel of a step from the SQ counters (rocprofv3 --pmc, one small group per pass, no trace domains):
# is a kernel bound by VALU issue (SQ_ACTIVE_INST_VALU close to SQ_BUSY_CU_CYCLES) or by waiting (few active cycles per busy cycle)?
w=$1
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT")
n=0
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/mix_${w}_$n
  rocprofv3 --pmc $g --output-format csv -d /tmp/mix_${w}_$n -- python3 /root/repo/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/mix_${w}_$n.log 2>&1
  n=$((n + 1))
done
mkdir -p /root/repo/gpurun_out
python3 - "$w" > /root/repo/gpurun_out/issue_mix_$w.txt <<'PY'
import csv, glob, sys, collections
w = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/mix_{w}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mjh_" in r["Kernel_Name"]:
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in per.items():
    print(k)
    for name in sorted(c):
        v = c[name][len(c[name]) // 2:]  # the later half of the dispatches: the timed trajectory, not the spin-up
        print(f"   {name:24s} {sum(v) / len(v):16.0f}   ({len(c[name])} dispatches)")
PY
cat /root/repo/gpurun_out/issue_mix_$w.txt
