#!/bin/bash
# usage (GPU box): tools/pmc_latency.sh <workload> -> per-kernel average memory-instruction latencies and L1 behaviour (rocprofv3 --pmc, separate passes)
w=$1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1)); rm -rf /tmp/pmcl_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcl_$i -- python3 /root/repo/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/pmcl_$i.log 2>&1 || tail -2 /tmp/pmcl_$i.log
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcl_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mjh_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k[:60])
    print("   " + "  ".join(f"{n}={v:.3g}" for n, v in sorted(c.items())))
    g = lambda n: c.get(n, float("nan"))
    print(f"   avg VMEM latency {g('SQ_INST_LEVEL_VMEM') / max(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR'), 1):.0f} cyc, SMEM {g('SQ_INST_LEVEL_SMEM') / max(g('SQ_INSTS_SMEM'), 1):.0f}, LDS {g('SQ_INST_LEVEL_LDS') / max(g('SQ_INSTS_LDS'), 1):.0f};  L1 miss rate {g('TCP_TCC_READ_REQ_sum') / max(g('TCP_TOTAL_CACHE_ACCESSES_sum'), 1):.2f}")
PY
