"""Reference ROLLOUTS (free-running, not teacher-forced) for tests/test_trajectory.py.

TEST INFRASTRUCTURE, container-only (needs /root/reference, see oracle/ref_harness.py).  For each case the reference's own
`forward.step` is iterated NSTEPS times from a seeded state and (qpos, qvel, sensordata) are recorded every EVERY steps:
tests/golden/traj_<case>.npz.  Mirrors the reference's own multi-step checks (test/mjx_correctness_test.py:216-330:
100 steps, atol 1e-5 vs MJX)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "mujoco-torch_amd"))
import ref_harness  # noqa: E402
from mujoco_torch_amd import mjcf  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
# name -> (xml, overrides, nsteps, every).  Solver iteration counts are raised where the XML asks for a single iteration, so
# that the rollouts are not dominated by the line-search knife edge.
CASES = {
    "humanoid_newton50": ("humanoid", {"iterations": 50, "ls_iterations": 50}, 40, 10),
    "halfcheetah": ("halfcheetah", {}, 50, 10),
    "hopper": ("hopper", {}, 50, 10),
    "ant": ("ant", {}, 40, 10),
    "sensor_rig": ("sensor_rig", {}, 40, 10),
    "swimmer": ("swimmer", {}, 50, 10),
}


def initial_state(lite, seed=7):
    rng = np.random.RandomState(seed)
    q = lite.qpos0 + 0.02 * rng.randn(lite.nq)
    return q, 0.1 * rng.randn(lite.nv), np.clip(0.3 * rng.randn(lite.nu), -1, 1)


def main():
    ref = ref_harness.load()
    for case, (xml, overrides, nsteps, every) in CASES.items():
        lite = mjcf.from_xml_path(os.path.join(os.path.dirname(HERE), "mujoco-torch_amd", "mujoco_torch_amd", "test_data", xml + ".xml"))
        for k, v in overrides.items():
            setattr(lite.opt, k, v)
        m = ref_harness.put_model(ref, lite)
        q, v, u = initial_state(lite)
        d = ref.io.make_data(m).replace(qpos=torch.tensor(q), qvel=torch.tensor(v), ctrl=torch.tensor(u))
        store = {"qpos0": q, "qvel0": v, "ctrl": u}
        for s in range(1, nsteps + 1):
            d = ref.forward.step(m, d)
            if s % every == 0:
                store[f"qpos/{s}"] = d.qpos.numpy().copy()
                store[f"qvel/{s}"] = d.qvel.numpy().copy()
                store[f"sensordata/{s}"] = d.sensordata.numpy().copy()
        store["meta"] = np.array(json.dumps(dict(xml=xml, overrides=overrides, nsteps=nsteps, every=every)))
        np.savez_compressed(os.path.join(GOLD, f"traj_{case}.npz"), **store)
        print(case, "recorded", nsteps, "steps; final |qvel| max", float(d.qvel.abs().max()))


if __name__ == "__main__":
    main()
