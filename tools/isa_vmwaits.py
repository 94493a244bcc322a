"""Where a kernel waits for its own STORES: walks the listing with the in-order vector-memory queue (loads, stores and scratch accesses share
vmcnt on gfx9) and prints every `s_waitcnt vmcnt(n)` that has to retire stores, with what forced it (the loads / scratch reloads queued
behind them).  Straight-line approximation: branches are ignored.  usage: python tools/isa_vmwaits.py FILE.s kernel-substring"""
import re, sys
t = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(t) if re.match(r"^_Z\w+:", l)]
for n, i0 in enumerate(starts):
    if sys.argv[2] not in t[i0]: continue
    i1 = starts[n + 1] if n + 1 < len(starts) else len(t)
    q = []  # outstanding ops, oldest first: 'S' store, 'L' load, 'R' scratch load, 'W' scratch store
    tot = 0
    for i in range(i0, i1):
        l = t[i].strip(); op = l.split()[0] if l else ""
        if op.startswith(("global_store", "flat_store", "buffer_store")): q.append("S")
        elif op.startswith(("global_load", "flat_load", "buffer_load")): q.append("L")
        elif op.startswith("scratch_load"): q.append("R")
        elif op.startswith("scratch_store"): q.append("W")
        elif op == "s_waitcnt" and "vmcnt" in l:
            k = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
            retired = q[:max(0, len(q) - k)]; q = q[len(retired):]
            ns = retired.count("S")
            if ns:
                tot += 1
                print(f"line {i - i0:6d} vmcnt({k}) retires {ns:2d} stores, {retired.count('W')} scratch stores; forced by {retired.count('L')} loads, {retired.count('R')} scratch reloads   | next: {t[i + 1].strip()[:60]}")
        elif op == "s_barrier" or op == "s_endpgm": pass
    print("waits that retire stores:", tot)
