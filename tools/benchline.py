"""Digest of bench.py's JSON line on stdin: value, ms per step, per-kernel microseconds.  usage: python bench.py ... | python tools/benchline.py [label]   or   python tools/benchline.py FILE [label]"""
import json, os, sys
args = sys.argv[1:]
if args and os.path.exists(args[0]):  # a file instead of a pipe
    text, args = open(args[0]).read(), args[1:]
elif args and (args[0].endswith(".json") or "/" in args[0]):  # a file that is not there (round 6: a missing path was taken for a label and the tool sat on an inherited stdin for 25 GPU-minutes)
    sys.exit(f"benchline: no such file: {args[0]}")
elif sys.stdin.isatty():              # a digest tool must never hang a GPU box waiting on a terminal
    sys.exit("benchline: no input (pipe a bench line in or pass a file)")
else:
    text = sys.stdin.read()
lines = [x for x in text.strip().splitlines() if x.startswith("{")]
if not lines:
    sys.exit("benchline: no JSON line in the input")
d = json.loads(lines[-1])
def dig(tag, r, v, ms):
    ks = [(k.get("id"), round(k["avg_us"], 1), k.get("launches_per_step", 1)) for k in r.get("per_kernel", [])]
    print(f"{tag:28s} {v / 1e6:7.3f} M  {ms:.4f} ms  " + " ".join(f"({i}) {u}" + (f"x{int(n)}" if n != 1 else "") for i, u, n in ks), flush=True)
lab = args[0] if args else ""
dig(lab + " " + d["config"]["workload"][:18], d["roofline"], d["value"], d["ms_per_step"])
for k, w in (d.get("other_workloads") or {}).items():
    if isinstance(w, dict) and "value" in w:
        dig(lab + " " + k, w.get("roofline", {}), w["value"], w["ms_per_step"])
