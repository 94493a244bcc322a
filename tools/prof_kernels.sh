#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <workload> [steps]  -> per-kernel durations (rocprofv3 --kernel-trace --stats) of a bench run.
# Two averages per kernel: over EVERY dispatch of the process (what --stats prints: the 100 spin-up steps of the early, expensive trajectory included) and over the
# dispatches of the LAST 50 steps -- the window bench.py's per-kernel HIP events cover (roofline_of runs after the timed loops) -- so that the two tools are compared on
# one window (270 us against 204 us for the mesh solver was the whole process against the last 50 steps).
w=$1; steps=${2:-200}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python3 /root/repo/bench.py --workload $w --steps $steps --warmup 20 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/prof_$w.log 2>&1
f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/prof_$w -name "*kernel_trace.csv" | head -1)
mkdir -p /root/repo/gpurun_out
cp $f /root/repo/gpurun_out/${w}_kernel_stats.csv
python3 - "$f" "$t" <<'PY'
import csv, sys, collections
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1])) if "mjh_" in r["Name"]}
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "mjh_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
# launches per step of each kernel: the kernel with the fewest dispatches that runs every step is launched once per step
total_steps = min(len(v) for v in per.values() if len(v) > 100)
print(f"{'kernel':62s} {'calls':>6s} {'avg all (us)':>13s} {'avg last 50 steps (us)':>23s}  share")
for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    lps = max(1, round(len(d) / total_steps))
    last = d[-50 * lps:]
    pct = float(stats[name]["Percentage"]) if name in stats else 0.0
    print(f"{name[:62]:62s} {len(d):6d} {sum(d) / len(d) / 1e3:13.1f} {sum(last) / len(last) / 1e3:23.1f}  {pct:5.1f} %")
PY
