"""Loops of one kernel in a hipcc -S listing: every backward branch (target label defined above it) with the instruction mix of the
lines it spans.  usage: python tools/isa_loops.py FILE.s kernel-substring [min instructions]"""
import collections, re, sys
text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 30
starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
for n, (i0, name) in enumerate(starts):
    if want not in name: continue
    i1 = starts[n + 1][0] if n + 1 < len(starts) else len(text)
    labels = {}
    for i in range(i0, i1):
        m = re.match(r"^(\.LBB\w+):", text[i])
        if m: labels[m.group(1)] = i
    loops = []
    for i in range(i0, i1):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\w+)", text[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(name, "lines", i0, i1, "loops", len(loops))
    for a, b in sorted(loops):
        ins = [l.strip() for l in text[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        if len(ins) < minins: continue
        c = collections.Counter()
        for x in ins:
            k = x.split()[0]
            if k.startswith("ds_"): c["lds"] += 1
            elif k.startswith(("global_load", "buffer_load", "flat_load")): c["vmem_rd"] += 1
            elif k.startswith(("global_store", "buffer_store", "flat_store")): c["vmem_wr"] += 1
            elif k.startswith("scratch_"): c["scratch"] += 1
            elif k.startswith("s_waitcnt"): c["wait"] += 1
            elif k.startswith("s_load"): c["smem"] += 1
            elif "f64" in k: c["f64"] += 1
            elif k.startswith("v_"): c["valu"] += 1
            elif "branch" in k: c["br"] += 1
            else: c["salu"] += 1
        nested = sum(1 for a2, b2 in loops if a < a2 and b2 < b)
        print(f"  L{a - i0:6d}..{b - i0:6d}  n={len(ins):5d} inner={nested:2d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
