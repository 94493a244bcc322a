"""Free-running rollouts against the reference's own rollouts (oracle/gen_trajectories.py): 40-50 steps, state and sensors
compared every 10 steps.  The reference's multi-step bar against MJX is atol 1e-5 over 100 steps
(test/mjx_correctness_test.py:216-330); here the comparison is against the reference itself, so the tolerances are the
accumulated rounding of two implementations of the same arithmetic: 1e-7 (oracle) / 1e-6 (HIP), relative to the leaf."""
import json
import os

import numpy as np
import pytest
import torch

import mujoco_torch_amd as mt
import pyoracle
from _util import GOLD, TRAJECTORY_CASES, load_model, rel_err


def _load(case):
    z = np.load(os.path.join(GOLD, f"traj_{case}.npz"))
    meta = json.loads(str(z["meta"]))
    mx = load_model(meta["xml"], meta["overrides"])
    d = mt.make_data(mx).replace(qpos=torch.tensor(z["qpos0"]), qvel=torch.tensor(z["qvel0"]), ctrl=torch.tensor(z["ctrl"]))
    return z, meta, mx, d


def _check(z, s, qpos, qvel, sensordata, tol, what):
    for name, got in (("qpos", qpos), ("qvel", qvel), ("sensordata", sensordata)):
        want = z[f"{name}/{s}"]
        assert rel_err(got, want, floor=1e-3) <= tol, f"{what} step {s}: {name} off by {rel_err(got, want, floor=1e-3):.2e}"


@pytest.mark.parametrize("case", TRAJECTORY_CASES)
def test_oracle_rollout_matches_reference(case, oracle_lib):
    z, meta, mx, d = _load(case)
    for s in range(1, meta["nsteps"] + 1):
        d = pyoracle.apply(d, pyoracle.run(mx, d, step=True))
        if s % meta["every"] == 0:
            _check(z, s, d.qpos.numpy(), d.qvel.numpy(), d.sensordata.numpy(), 1e-7, f"oracle {case}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", TRAJECTORY_CASES)
def test_hip_rollout_matches_reference(case):
    z, meta, mx, d = _load(case)
    mdev, dg = mx.to("cuda"), d.to("cuda")
    for s in range(1, meta["nsteps"] + 1):
        dg = mt.step(mdev, dg)
        if s % meta["every"] == 0:
            _check(z, s, dg.qpos.cpu().numpy(), dg.qvel.cpu().numpy(), dg.sensordata.cpu().numpy(), 1e-6, f"HIP {case}")
