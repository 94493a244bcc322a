"""GPU box: float32 Newton steps of every bundled model whose packed solver tier builds its Hessian on the matrix cores (9 - 16 dofs; nv < NMAX and NMAX = 16 included),
heavily perturbed batches against the float32 oracle -- the campaign's check at the campaign's float32 bound.   python tools/f32_newton_sweep.py [B] [steps]    (SWEEP_ELLIPTIC=1: elliptic cones)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import torch
import mujoco_torch_amd as mt
from _cases import FUZZ_BAND, FUZZ_TOL_PRE, fuzz_batch
from _util import check_against_oracle, gpu_out_to_numpy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for xml in ("ball_free_actuators", "gravcomp_arm", "halfcheetah", "walker2d", "mocap_target", "sensor_rig2", "tendon_fixed", "mesh_contact", "mesh_contact_arm"):
    try:
        mx, d = fuzz_batch(xml, dict({"solver": 2}, **({"cone": 1} if os.environ.get("SWEEP_ELLIPTIC") else {})), torch.float32, B)
        mdev, dg = mx.to("cuda"), d.to("cuda")
        res, tail = [], {}
        for s in range(STEPS):
            og = mt.step(mdev, dg)
            frac, worst = check_against_oracle(mx, dg.cpu(), gpu_out_to_numpy(og), FUZZ_TOL_PRE[torch.float32], 5e-3, what=f"{xml} step{s}", nthreads=16, band=FUZZ_BAND.get(xml), tail_rules=True, tail_out=tail)
            res.append(float(f"{worst:.1e}"))
            dg = og
        print(f"ok   {xml:22s} nv {int(mx.nv):2d}  worst solver err per step {res}  tail {tail}", flush=True)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print(f"FAIL {xml}: {str(ex)[:300]}", flush=True)
sys.exit(1 if bad else 0)
