"""Summarise per-phase kernel durations from a rocprofv3 --kernel-trace CSV of bench.py."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mjh_phase" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in sorted(agg.items()):
    v = v[len(v) // 5:]
    m = sum(v) / len(v) / 1e3
    tot += m
    print(f"{k[0][:60]:60s} LDS {k[1]:>6} VGPR {k[2]:>4} SGPR {k[3]:>4} scratch {k[4]:>4}  n={len(v):4d}  avg {m:8.1f} us  min {min(v)/1e3:8.1f}")
print(f"sum of phase averages: {tot:.1f} us")
