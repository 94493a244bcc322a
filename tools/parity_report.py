"""GPU box: the "float64 max rel-err" half of BASELINE.json's metric as numbers -- per-leaf error of the HIP step against the CPU
oracle on the BASELINE configs (2: humanoid f64 Euler+CG, 3: ant f32 RK4+Newton elliptic, 5: mesh scene f32 Newton; config 4 is
config 2's model) in the bench's input recipe, three consecutive steps.  Writes gpurun_out/parity.json (copied to profiles/<round>/).

Errors are max-norm per leaf: |got - want|max / max(|want|max, floor) (tests/_util.rel_err), NOT element-wise -- except `state_leaves_elementwise_*`:
worst entry of qpos / qvel / qacc against max(|entry|, 1e-3 of its leaf's largest) on the accepted branch (tests/_util.elementwise_err).  Solver-dependent
leaves are reported twice: against the oracle's natural run and against the closest admissible branch per environment.
MJX / MuJoCo-C parity is unpinned in this container (no mujoco / jax wheel): the oracle is pinned by the reference's own Python
(tests/golden, oracle/gen_golden.py).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
from _cases import seeded_batch  # noqa: E402
from _util import INT_LEAVES, PRE_SOLVER, SOLVER_LEAVES, compare_with_oracle, gpu_out_to_numpy, rel_err, solver_floor  # noqa: E402

CONFIGS = {
    "config2_humanoid_f64_euler_cg": ("humanoid", {"solver": 1}, torch.float64, 1024),
    "config3_ant_f32_rk4_newton_elliptic": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32, 512),
    "config5_mesh_f32_newton": ("mesh_contact", {}, torch.float32, 512),
    "config2_twin_f64_of_config3": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float64, 256),
    "config2_twin_f64_of_config5": ("mesh_contact", {}, torch.float64, 256),
}
out = {"reference": "CPU oracle (oracle/mjoracle.c), pinned bit-for-bit-reproducibly by goldens recorded from the reference's own Python step; MJX parity unpinned (no jax / mujoco offline)",
       "metric": "max-norm relative error per leaf, |got - want|max / max(|want|max, floor); floor 1e-6 (1e-3 for solver leaves and float32; qfrc_constraint on the scale of the environment's largest |efc_force|)",
       "git_commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or os.environ.get("MJH_GIT_COMMIT"),
       "configs": {}, "summary": {}}
for name, (xml, ov, dt, B) in CONFIGS.items():
    mx, d = seeded_batch(xml, ov, dt, B)
    mdev, dg = mx.to("cuda"), d.to("cuda")
    steps = []
    for s in range(3):
        og = mt.step(mdev, dg)
        got = gpu_out_to_numpy(og)
        c = compare_with_oracle(mx, dg.cpu(), got, nthreads=32)
        best = {}
        for n in SOLVER_LEAVES:  # per leaf on each environment's accepted branch
            best[n] = max(rel_err(got[n][e], c["alts"][int(c["which"][e])][n][e], solver_floor(n, {"efc_force": c["alts"][int(c["which"][e])]["efc_force"][e]})) for e in range(B))
        steps.append({"pre_solver_leaves": {n: c["pre"][n] for n in PRE_SOLVER}, "pre_solver_leaves_elementwise": {n: c["pre_elem"][n] for n in PRE_SOLVER}, "integer_leaves_bit_exact": bool(c["ints_ok"]),
                      "solver_leaves_vs_natural_oracle_run": c["leaf_nat"], "solver_leaves_on_accepted_branch": best,
                      "state_leaves_elementwise_on_accepted_branch": float(c["elem_best"].max()),
                      "envs_on_non_natural_branch": float((c["err_nat"] > (1e-8 if dt == torch.float64 else 2e-3)).mean()),
                      "envs_with_noise_candidates": float((c["knife"] > 0).mean())})
        dg = og
    out["configs"][name] = {"xml": xml, "overrides": ov, "dtype": str(dt)[6:], "envs": B, "steps": steps}
    out["summary"][name] = {"dtype": str(dt)[6:], "max_pre_solver": max(max(s["pre_solver_leaves"].values()) for s in steps),
                            "max_pre_solver_elementwise": max(max(s["pre_solver_leaves_elementwise"].values()) for s in steps),
                            "max_solver_on_accepted_branch": max(max(s["solver_leaves_on_accepted_branch"].values()) for s in steps),
                            "max_state_qpos_qvel_on_accepted_branch": max(max(s["solver_leaves_on_accepted_branch"][k] for k in ("qpos", "qvel")) for s in steps),
                            "max_state_elementwise_on_accepted_branch": max(s["state_leaves_elementwise_on_accepted_branch"] for s in steps),
                            "integer_leaves_bit_exact": all(s["integer_leaves_bit_exact"] for s in steps),
                            "max_fraction_on_non_natural_branch": max(s["envs_on_non_natural_branch"] for s in steps)}
    print(name, json.dumps(out["summary"][name]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "parity.json"), "w") as f:
    json.dump(out, f, indent=1, default=float)
