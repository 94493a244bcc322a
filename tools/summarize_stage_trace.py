import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mjh_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["kin+com", "+crb+factor", "+collision", "+constraint", "+velocity", "+actuation", "+solve", "step"]
i = 0
while i + 32 <= len(rows):
    grid = rows[i]["Grid_Size_X"]
    prev = 0
    print(f"grid {grid} (LDS {rows[i]['LDS_Block_Size']} B, VGPR {rows[i]['VGPR_Count']}):")
    for k in range(8):
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[i + 4 * k + 1: i + 4 * k + 4]]
        d = sum(durs) / len(durs) / 1e3
        print(f"  {names[k]:14s} {d:9.1f} us  (+{d - prev:8.1f})" if k < 7 else f"  {names[k]:14s} {d:9.1f} us")
        if k < 7: prev = d
    i += 32
