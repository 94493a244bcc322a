"""GPU box, one-off differential campaign: for every bundled / test model, large random batches (bigger perturbations than the
seeded unit tests: joint angles, un-normalised quaternions, velocities, controls, applied forces, warm starts) are stepped on the
GPU and checked leaf by leaf against the CPU oracle.  Prints one line per (model, dtype) and exits non-zero on a mismatch.

    python tools/fuzz_parity.py [B] [steps]
"""
import os
import sys
import time
import traceback
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
from _util import check_against_oracle, gpu_out_to_numpy, load_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
CASES = [  # (xml, overrides, dtype, solver tolerance)
    ("humanoid", {"solver": 1}, torch.float64, 1e-7), ("humanoid", {}, torch.float64, 1e-7), ("humanoid", {"iterations": 20, "ls_iterations": 20}, torch.float32, 5e-3),
    ("ant", {}, torch.float64, 1e-7), ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32, 5e-3), ("ant", {"solver": 1, "cone": 1}, torch.float64, 1e-5),
    ("halfcheetah", {}, torch.float64, 1e-6), ("hopper", {}, torch.float64, 1e-6), ("walker2d", {"integrator": 1}, torch.float64, 1e-6),
    ("swimmer", {"viscosity": 0.05}, torch.float64, 1e-7), ("cartpole", {}, torch.float64, 1e-7), ("satellite_small", {}, torch.float64, 1e-7),
    ("sensor_rig", {}, torch.float64, 1e-7), ("mesh_contact", {}, torch.float64, 1e-6), ("mesh_contact", {"integrator": 1}, torch.float64, 1e-6), ("convex_primitives", {}, torch.float64, 1e-5),
    ("equality_loops", {}, torch.float64, 1e-6), ("equality", {}, torch.float64, 1e-6), ("ball_limits", {}, torch.float64, 1e-6),
    ("tendon_fixed", {}, torch.float64, 1e-6), ("gravcomp_arm", {}, torch.float64, 1e-6), ("gravcomp_arm", {"integrator": 1}, torch.float64, 1e-6), ("ball_free_actuators", {}, torch.float64, 1e-6),
    ("mocap_target", {}, torch.float64, 1e-6), ("pendula", {}, torch.float64, 1e-6), ("pendula", {"integrator": 1, "solver": 1}, torch.float32, 5e-3),
    ("frictionloss_dof", {}, torch.float64, 1e-7), ("ant_frictionloss", {}, torch.float64, 1e-6),
    ("muscle_arm", {}, torch.float64, 1e-7), ("tendon_armature", {}, torch.float64, 1e-6), ("tendon_friction", {}, torch.float64, 1e-6), ("capsules_topk", {}, torch.float64, 1e-6),
]
TOL_PRE = {torch.float64: 1e-9, torch.float32: 1e-3}  # float32: near-degenerate contact normals amplify eps under these perturbations
import _util  # noqa: E402

ALL_SOLVER_LEAVES = list(_util.SOLVER_LEAVES)
bad = 0
for xml, ov, dt, tol_sol in CASES:
    # float32: qfrc_constraint = J^T efc_force cancels forces of ~1e5 (adjacent capsules of the ant interpenetrate at their shared
    # joint point) down to ~1e-2 -- pure rounding residue; the dynamics leaves carry the comparison there
    _util.SOLVER_LEAVES[:] = [n for n in ALL_SOLVER_LEAVES if dt == torch.float64 or n not in ("qfrc_constraint", "efc_force")]
    t0 = time.time()
    try:
        mx = load_model(xml, ov, dt)
        rng = np.random.RandomState(zlib.crc32(xml.encode()) % 1000)  # stable across processes (hash() is salted)
        d = mt.make_data(mx).expand(B).clone()
        q = d.qpos.clone()
        scale = torch.tensor(rng.uniform(0.0, 0.5, size=(B, 1)))  # per-environment perturbation size, some environments stay at qpos0
        q = q + scale * torch.tensor(rng.randn(B, mx.nq))
        kw = dict(qpos=q, qvel=torch.tensor(rng.randn(B, mx.nv)) * scale * 4, ctrl=torch.tensor(rng.uniform(-1.2, 1.2, size=(B, mx.nu))),
                  qfrc_applied=torch.tensor(0.5 * rng.randn(B, mx.nv)), xfrc_applied=torch.tensor(0.5 * rng.randn(B, mx.nbody, 6)),
                  qacc_warmstart=torch.tensor(rng.randn(B, mx.nv)) * scale)
        if mx.nmocap:
            kw["mocap_pos"] = d.mocap_pos + 0.1 * torch.tensor(rng.randn(B, mx.nmocap, 3))
            kw["mocap_quat"] = d.mocap_quat + 0.3 * torch.tensor(rng.randn(B, mx.nmocap, 4))
        if mx.neq:
            kw["eq_active"] = torch.tensor(rng.randint(0, 2, size=(B, mx.neq)), dtype=torch.int32) * d.eq_active.clamp(max=1) + d.eq_active * 0
            kw["eq_active"] = torch.where(torch.tensor(rng.rand(B, mx.neq) < 0.3), torch.zeros_like(d.eq_active), d.eq_active)
        d = d.replace(**kw)
        if dt != torch.float64:
            d = d.to(dt)
        mdev, dg = mx.to("cuda"), d.to("cuda")
        fracs = []
        for s in range(STEPS):
            og = mt.step(mdev, dg)
            frac, worst = check_against_oracle(mx, dg.cpu(), gpu_out_to_numpy(og), TOL_PRE[dt], tol_sol, what=f"{xml} step{s}", nthreads=16)
            fracs.append((round(frac, 3), float(f"{worst:.1e}")))
            dg = og
        print(f"ok   {xml:22s} {str(ov):55s} {str(dt)[6:]:8s} B={B} (alt-branch frac, worst solver err) per step: {fracs}  [{time.time() - t0:.0f}s]", flush=True)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print(f"FAIL {xml} {ov} {dt}: {str(ex)[:600]}", flush=True)
        traceback.print_exc(limit=2)
sys.exit(1 if bad else 0)
