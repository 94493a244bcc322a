#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <workload> [steps]  -> per-kernel stats (rocprofv3 --kernel-trace --stats) of a bench run
w=$1; steps=${2:-50}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python3 /root/repo/bench.py --workload $w --steps $steps --warmup 5 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/prof_$w.log 2>&1
f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1)
mkdir -p /root/repo/gpurun_out
cp $f /root/repo/gpurun_out/${w}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    if "mjh_" in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
PY
