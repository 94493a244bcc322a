#!/bin/bash
# usage (GPU box): tools/hbm_traffic.sh <workload> -> gpurun_out/hbm_traffic_<workload>.json
# HBM-side bytes per step from the PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (no trace domains), FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B).
w=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${w}_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${w}_$c -- python3 /root/repo/bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-other-workloads --no-long-run > /tmp/pmc_${w}_$c.log 2>&1
done
mkdir -p /root/repo/gpurun_out
python3 - "$w" <<'PY'
import csv, glob, json, sys, collections
w = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "mjh_" in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    res[c] = {k: (sum(v) / len(v), len(v)) for k, v in per.items()}
import hashlib, os, subprocess
def lib_fingerprint():
    h = hashlib.sha256()
    csrc = "/root/repo/mujoco-torch_amd/csrc"
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]
line = json.loads([x for x in open(f"/tmp/pmc_{w}_WRITE_SIZE.log").read().splitlines() if x.startswith('{"metric"')][-1])
# steps the process ran = forward passes / passes per step (RK4: four); every forward pass launches the kinematics kernel once
kin = sum(n for k, (m, n) in res["WRITE_SIZE"].items() if ("mjh_phase_kernel" in k and (", 0, " in k or ", 12, " in k or ", 13, " in k or ", 17, " in k)) or ("mjh_sol2_kernel" in k and (", 34>" in k or ", 36>" in k or ", 18>" in k)))  # 12 / 13 / 17: kinematics fused with the velocity (and crb) stages; sol2 <.., 34 | 36>: the whole pass in one kernel; <.., 18>: one RK4 stage in one kernel
steps_total = kin / (4 if "RK4" in line["config"]["workload"] else 1)
# per-step launches of each kernel = dispatches / steps (RK4 launches each phase four times per step)
fetch_kb = sum(m * n for m, n in res["FETCH_SIZE"].values()) / steps_total
write_kb = sum(m * n for m, n in res["WRITE_SIZE"].values()) / steps_total
out = {
    "config": line["config"]["workload"],
    "lib_fingerprint": lib_fingerprint(),  # sha256 of csrc/*.h, *.hip the counters were collected on (bench.py drops the figures when the kernels changed)
    "git_commit": (subprocess.run(["git", "-C", "/root/repo", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("MJH_GIT_COMMIT")),
    "kernels": {k: {"FETCH_SIZE_KB_raw_mean": res["FETCH_SIZE"].get(k, (0, 0))[0], "WRITE_SIZE_KB_mean": res["WRITE_SIZE"].get(k, (0, 0))[0], "dispatches_per_step": n / steps_total} for k, (m, n) in res["WRITE_SIZE"].items()},
    "FETCH_SIZE_KB_raw_per_step": fetch_kb, "WRITE_SIZE_KB_per_step": write_kb,
    "fetch_bytes_corrected": 2 * 1024 * fetch_kb, "write_bytes": 1024 * write_kb,
    "hbm_bytes_per_step": 2 * 1024 * fetch_kb + 1024 * write_kb,
    "algorithmic_bytes_per_step": line["roofline"]["step"]["algorithmic_bytes_per_step"],
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); per-kernel means x dispatches per step, summed over the kernels of one step",
}
json.dump(out, open(f"/root/repo/gpurun_out/hbm_traffic_{w}.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("FETCH_SIZE_KB_raw_per_step", "WRITE_SIZE_KB_per_step", "hbm_bytes_per_step", "algorithmic_bytes_per_step")}))
PY
