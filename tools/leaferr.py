import sys
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); [sys.path.insert(0, os.path.join(R,p)) for p in ("tests","mujoco-torch_amd","oracle")]
import numpy as np, torch, mujoco_torch_amd as mt, pyoracle
from _util import *
for case in ["ant_rk4_newton_ell_f64", "ant_euler_newton_pyr_f64", "humanoid_cg_f64_perturbed", "ant_rk4_newton_ell_f32"]:
    g = Golden(case)
    mdev = g.model.to("cuda")
    d = g.input_data()
    for s in range(g.nsteps):
        out = gpu_out_to_numpy(mt.step(mdev, d.to("cuda")))
        want = lambda n: np.stack([g.expected(e, s, n) for e in range(g.nenv)])
        errs = sorted(((rel_err(out[n], want(n)), n) for n in REAL_LEAVES), reverse=True)
        print(case, s, [(f"{e:.1e}", n) for e, n in errs[:8]])
        d = pyoracle.apply(d, {n: want(n) for n in REAL_LEAVES + INT_LEAVES})
