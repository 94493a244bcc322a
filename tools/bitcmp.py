"""GPU box: two builds of the library on the same inputs, every leaf compared BIT FOR BIT -- the check for a refactor that claims to keep every operation and its order.

    python tools/bitcmp.py [LIB_A [LIB_B]]      (defaults: mujoco-torch_amd/lib/libmjhip_base.so, the shipped libmjhip.so)

Each library steps the same batches in a process of its own (native.LIB_PATH is read once per process): the three BASELINE workloads, their float64 / float32 twins, models with
sensors, mocap bodies, equality rows, tendons, muscles, convex pairs; Euler and RK4; three steps each (state carried).  Prints one line per case and exits non-zero on any difference.
Baseline library: build the committed tree into another directory, e.g.
    git worktree add /tmp/base_wt HEAD; MJH_BUILD_DIR=/tmp/base_build MJH_BUILD_OUT=$PWD/mujoco-torch_amd/lib/libmjhip_base.so bash /tmp/base_wt/mujoco-torch_amd/csrc/build.sh"""
import os
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("humanoid", {"solver": 1}, "float64", 96), ("humanoid", {}, "float64", 64), ("humanoid", {"solver": 1, "iterations": 3}, "float64", 64), ("humanoid", {"solver": 1}, "float32", 64),
         ("ant", {"integrator": 1, "solver": 2, "cone": 1}, "float32", 160), ("ant", {}, "float64", 64), ("ant", {"integrator": 1, "solver": 2, "cone": 1}, "float64", 64),
         ("mesh_contact", {}, "float32", 96), ("mesh_contact", {"integrator": 1}, "float64", 64), ("hopper", {}, "float64", 64), ("walker2d", {"integrator": 1}, "float64", 64),
         ("halfcheetah", {}, "float64", 64), ("sensor_rig2", {}, "float64", 64), ("mocap_target", {}, "float64", 64), ("mocap_child", {}, "float64", 64), ("mocap_chain", {"solver": 1, "iterations": 1, "ls_iterations": 4}, "float64", 64), ("equality_loops", {}, "float64", 64),
         ("tendon_fixed", {}, "float64", 64), ("muscle_arm", {}, "float64", 64), ("convex_primitives", {}, "float64", 64), ("centipede", {}, "float64", 33), ("cartpole", {}, "float64", 64),
         ("pendula", {"integrator": 1, "solver": 1}, "float32", 64), ("capsules_topk", {}, "float64", 64), ("gravcomp_arm", {"integrator": 1}, "float64", 64)]
CHILD = r'''
import sys, os
R = sys.argv[3]
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
from mujoco_torch_amd import native
native.LIB_PATH = sys.argv[1]
import mujoco_torch_amd as mt
from _util import load_model, REAL_LEAVES, INT_LEAVES
from _cases import fuzz_batch
CASES = eval(sys.argv[4])
out = {}
for xml, ov, dt, B in CASES:
    dtype = getattr(torch, dt)
    mx, d = fuzz_batch(xml, ov, dtype, B)
    mdev, dg = mx.to("cuda"), d.to("cuda")
    for s in range(3):
        dg = mt.step(mdev, dg)
        out[f"{xml}{sorted(ov.items())}{dt}#step{s}"] = {n: native.data_field_tensor(dg, n).cpu() for n in REAL_LEAVES + INT_LEAVES}
    fw = mt.forward(mdev, dg)
    out[f"{xml}{sorted(ov.items())}{dt}#forward"] = {n: native.data_field_tensor(fw, n).cpu() for n in REAL_LEAVES + INT_LEAVES}
torch.save(out, sys.argv[2])
print("ran")
'''
if __name__ == "__main__":
    import torch

    a = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "mujoco-torch_amd", "lib", "libmjhip_base.so")
    b = sys.argv[2] if len(sys.argv) > 2 else os.path.join(R, "mujoco-torch_amd", "lib", "libmjhip.so")
    with tempfile.TemporaryDirectory() as td:
        res = []
        for i, lib in enumerate((a, b)):
            f = os.path.join(td, f"{i}.pt")
            r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), f, R, repr(CASES)], capture_output=True, text=True, timeout=3000)
            if r.returncode != 0 or "ran" not in r.stdout:
                sys.exit(f"{lib}: {r.stdout[-1500:]}{r.stderr[-3000:]}")
            res.append(torch.load(f))
    bad = 0
    for case in res[0]:
        diff = []
        for n, t in res[0][case].items():
            o = res[1][case][n]
            same = torch.equal(t, o) or (t.is_floating_point() and torch.equal(torch.nan_to_num(t), torch.nan_to_num(o)) and torch.equal(torch.isnan(t), torch.isnan(o)))
            if not same:
                e = float((t.double() - o.double()).abs().max() / max(float(t.double().abs().max()), 1e-30)) if t.is_floating_point() else -1
                diff.append(f"{n} ({e:.1e})")
        print(f"{'SAME' if not diff else 'DIFF'}  {case}" + ("  " + " ".join(diff[:12]) if diff else ""), flush=True)
        bad += bool(diff)
    print(f"{len(res[0]) - bad} of {len(res[0])} cases bit-identical between {os.path.basename(a)} and {os.path.basename(b)}")
    sys.exit(1 if bad else 0)
