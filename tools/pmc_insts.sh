#!/bin/bash
# usage (GPU box): tools/pmc_insts.sh <workload>  -> dynamic instruction mix and issue utilisation per kernel (rocprofv3 --pmc)
w=$1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/pmcx_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcx_$i -- python3 /root/repo/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/pmcx_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcx_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mjh_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[:58])
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(acc[k].items())))
PY
