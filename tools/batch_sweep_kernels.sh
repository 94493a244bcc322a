for b in 1024 2048 4096 8192; do
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 /root/repo/bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline > /tmp/pb.log 2>&1
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1)
python3 - "$f" $b <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "mjh_" in r["Name"]]
rows.sort(key=lambda r: r["Name"])
print("B", sys.argv[2], " ".join(f'{r["Name"].split("<")[1].split(">")[0].replace("double, ","P")}:{float(r["AverageNs"])/1e3:.1f}' for r in rows), "sum", round(sum(float(r["AverageNs"]) for r in rows)/1e3,1))
PY
done
