"""Static digest of a kernel's ISA (hipcc -S --cuda-device-only output): instruction counts by class, and how many LDS / global
reads are waited for on their own (an `s_waitcnt` that drains the counter with ONE request in flight: a dependent round trip).

usage: python tools/isa_stats.py FILE.s [kernel-name-substring]"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
for n, (i0, name) in enumerate(starts):
    if want not in name:
        continue
    i1 = starts[n + 1][0] if n + 1 < len(starts) else len(text)
    ins = [l.strip() for l in text[i0:i1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    c = collections.Counter(x.split()[0] for x in ins)
    cls = collections.Counter()
    for k, v in c.items():
        if k.startswith("ds_bpermute") or k.startswith("ds_swizzle"): cls["lds_permute"] += v
        elif k.startswith("ds_"): cls["lds"] += v
        elif k.startswith(("global_load", "buffer_load", "flat_load")): cls["vmem_rd"] += v
        elif k.startswith(("global_store", "buffer_store", "flat_store")): cls["vmem_wr"] += v
        elif k.startswith("scratch_"): cls["scratch"] += v
        elif k.startswith("s_waitcnt"): cls["waitcnt"] += v
        elif k.startswith("s_load"): cls["smem"] += v
        elif k.startswith("v_readlane") or k.startswith("v_readfirstlane"): cls["readlane"] += v
        elif "_dpp" in k: cls["dpp"] += v
        elif k.startswith("v_"): cls["valu"] += v
        elif k.startswith("s_cbranch") or k.startswith("s_branch"): cls["branch"] += v
        elif k.startswith("s_"): cls["salu"] += v
        else: cls["other"] += v
    # lone waits: an lgkmcnt(0) / vmcnt(0) wait with exactly one outstanding request of that kind since the previous wait
    lone_lds = lone_vm = 0
    out_lds = out_vm = 0
    for x in ins:
        op = x.split()[0]
        if op.startswith("ds_") or op.startswith("s_load"): out_lds += 1
        elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): out_vm += 1
        elif op.startswith("s_waitcnt"):
            if "lgkmcnt(0)" in x or x.endswith("s_waitcnt 0") or "lgkmcnt" not in x and "vmcnt" not in x:
                lone_lds += out_lds == 1
                out_lds = 0
            if "vmcnt(0)" in x:
                lone_vm += out_vm == 1
                out_vm = 0
        elif op.startswith(("s_cbranch", "s_branch", "s_barrier")):
            pass
    print(f"{name[:70]}: {len(ins)} instructions  " + "  ".join(f"{k}={v}" for k, v in sorted(cls.items())) + f"  lone LDS/SMEM waits={lone_lds}  lone VMEM waits={lone_vm}")
