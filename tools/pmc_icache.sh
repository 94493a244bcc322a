cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcy_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcy_$i -- python3 /root/repo/bench.py --workload ${1:-humanoid} --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/pmcy_$i.log 2>&1 || tail -3 /tmp/pmcy_$i.log
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcy_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mjh_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[:58])
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(acc[k].items())))
PY
grep -il "error\|invalid\|not found" /tmp/pmcy_*.log | head
