"""Prints a one-line digest of a bench.py JSON line read from stdin: value, ms/step and per-kernel average times."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
j = json.loads(sys.stdin.read())
print(tag, j["config"]["workload"][:12], round(j["value"] / 1e6, 3), "M", round(j["ms_per_step"], 4), "ms",
      [(k["id"], round(k["avg_us"], 1), k["launches_per_step"]) for k in j["roofline"]["per_kernel"]])
