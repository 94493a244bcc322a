"""Static instruction mix per source-line range of one kernel (listing compiled with -gline-tables-only).
usage: python tools/isa_lines.py FILE.s kernel-substring file.h:lo-hi[:label] ..."""
import re, sys, collections
t = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
ranges = []
for a in sys.argv[3:]:
    f, r = a.split(":")[0], a.split(":")[1]
    lab = a.split(":")[2] if a.count(":") > 1 else a
    lo, hi = map(int, r.split("-")); ranges.append((f, lo, hi, lab))
files = {}
for l in t:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
starts = [i for i, l in enumerate(t) if re.match(r"^_Z\w+:", l)]
for n, i0 in enumerate(starts):
    if want not in t[i0]: continue
    i1 = starts[n + 1] if n + 1 < len(starts) else len(t)
    cur = None
    cnt = {r[3]: collections.Counter() for r in ranges}
    for i in range(i0, i1):
        l = t[i]
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            if m.group(2) != "0": cur = (files.get(m.group(1)), int(m.group(2)))
            continue
        if not l.startswith("\t") or l.strip().startswith((";", ".")): continue
        if cur is None: continue
        for f, lo, hi, lab in ranges:
            if cur[0] == f and lo <= cur[1] <= hi:
                op = l.split()[0]
                c = cnt[lab]
                if op.startswith("ds_"): c["lds"] += 1
                elif op.startswith(("global_load", "flat_load", "buffer_load")): c["vmem_rd"] += 1
                elif op.startswith(("global_store", "flat_store", "buffer_store")): c["vmem_wr"] += 1
                elif op.startswith("scratch_"): c["scratch"] += 1
                elif op.startswith("s_waitcnt"): c["wait"] += 1
                elif op.startswith("s_load"): c["smem"] += 1
                elif "f64" in op: c["f64"] += 1
                elif op.startswith("v_"): c["valu"] += 1
                elif "branch" in op: c["br"] += 1
                else: c["salu"] += 1
                c["all"] += 1
    for lab, c in cnt.items():
        print(f"{lab:28s} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
