#!/bin/bash
# GPU box: builds tools/fetch_calib.hip, collects FETCH_SIZE and WRITE_SIZE in separate --pmc passes, prints counter / known bytes
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/fetch_calib /root/repo/tools/fetch_calib.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/calib_$c -- /tmp/fetch_calib > /tmp/calib_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
GiB = 1 << 30
known = {"read_k<double>": (GiB, 0), "read_k<float>": (GiB, 0), "read16_k": (GiB, 0), "write_k<double>": (0, GiB), "write_k<float>": (0, GiB),
         "rows_k<double, 1431>": ((GiB // 8 // 1431) * 1431 * 8,) * 2, "rows_k<float, 1504>": ((GiB // 4 // 1504) * 1504 * 4,) * 2}
for c, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    f = glob.glob(f"/tmp/calib_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(per.items()):
        key = next((n for n in known if n.replace(" ", "") in k.replace(" ", "")), None)
        if key is None or known[key][idx] == 0:
            continue
        kb = sum(v) / len(v)
        print(f"{c:10s} {key:24s} counter {kb * 1024 / 1e6:10.1f} MB   known {known[key][idx] / 1e6:10.1f} MB   counter / known = {kb * 1024 / known[key][idx]:.3f}")
PY
