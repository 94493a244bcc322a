"""Which source lines a kernel's scratch spills and reloads sit on (listing compiled with -gline-tables-only).  usage: python tools/isa_spills.py FILE.s kernel-substring"""
import re, sys, collections
t = open(sys.argv[1]).read().split("\n")
files = {}
for l in t:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
st = [i for i, l in enumerate(t) if re.match(r"^_Z\w+:", l)] + [len(t)]
for n, i0 in enumerate(st[:-1]):
    if sys.argv[2] not in t[i0]: continue
    i1 = st[n + 1]
    last = None; sp = collections.Counter(); rl = collections.Counter()
    for i in range(i0, i1):
        l = t[i]
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m and m.group(2) != "0": last = (files.get(m.group(1)), int(m.group(2)))
        if "scratch_store" in l: sp[last] += 1
        if "scratch_load" in l: rl[last] += 1
    print(t[i0][:60])
    print("  spills :", sorted(sp.items(), key=lambda x: -x[1])[:12])
    print("  reloads:", sorted(rl.items(), key=lambda x: -x[1])[:16])
