"""Digest of bench.py's JSON line on stdin: value, ms per step, per-kernel microseconds.  usage: python bench.py ... | python tools/benchline.py [label]"""
import json, sys
lines = [x for x in sys.stdin.read().strip().splitlines() if x.startswith("{")]
d = json.loads(lines[-1])
def dig(tag, r, v, ms):
    ks = [(k.get("id"), round(k["avg_us"], 1), k.get("launches_per_step", 1)) for k in r.get("per_kernel", [])]
    print(f"{tag:28s} {v / 1e6:7.3f} M  {ms:.4f} ms  " + " ".join(f"({i}) {u}" + (f"x{int(n)}" if n != 1 else "") for i, u, n in ks), flush=True)
lab = sys.argv[1] if len(sys.argv) > 1 else ""
dig(lab + " " + d["config"]["workload"][:18], d["roofline"], d["value"], d["ms_per_step"])
for k, w in (d.get("other_workloads") or {}).items():
    if isinstance(w, dict) and "value" in w:
        dig(lab + " " + k, w.get("roofline", {}), w["value"], w["ms_per_step"])
