"""Prints a one-line digest of a bench.py JSON line: value, ms/step and per-kernel average times.

usage: python bench.py | python tools/bench_line.py [tag]      or      python tools/bench_line.py FILE [tag]
(refuses to wait on a terminal / an absent pipe: a digest tool must never hang a GPU box)"""
import json
import os
import sys

args = sys.argv[1:]
if args and os.path.exists(args[0]):
    text, args = open(args[0]).read(), args[1:]
elif sys.stdin.isatty():
    sys.exit("bench_line: no input (pipe a bench line in or pass a file)")
else:
    text = sys.stdin.read()
tag = args[0] if args else ""
line = [l for l in text.splitlines() if l.startswith("{")]
if not line:
    sys.exit("bench_line: no JSON line in the input")
j = json.loads(line[-1])
print(tag, j["config"]["workload"][:12], round(j["value"] / 1e6, 3), "M", round(j["ms_per_step"], 4), "ms",
      [(k["id"], round(k["avg_us"], 1), k["launches_per_step"]) for k in j["roofline"]["per_kernel"]])
