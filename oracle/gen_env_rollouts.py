"""Reference physics under the ENV CALLER's control flow, for tests/test_zoo.py (GPU leg).

TEST INFRASTRUCTURE, container-only (needs /root/reference, see oracle/ref_harness.py).  The reference's environment
classes (zoo/base.py) cannot be instantiated here -- torchrl is not installed -- so this script drives the reference's own
`forward.step` the way `MujocoTorchEnv` does: patched XML (the build's `_patch_xml`, same regexes as zoo/base.py:235-264 /
zoo/ant.py:37-55), dx0 = step(make_data(m)) (zoo/base.py:128-135), a seeded reset state dx0 + noise (:266-273), then per
agent step ctrl := action (through `_prepare_ctrl`'s recipe) followed by FRAME_SKIP physics steps (:307-318).  Recorded
per agent step: qpos, qvel (+ cinert / cvel / qfrc_actuator for the rich humanoid) -> tests/golden/env_<name>.npz.
Observations, rewards and termination are recomputed from these states by formulas written out in the test.
"""
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
from mujoco_torch_amd.zoo import ENVS, base  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
NENV, NSTEP = 2, 3
CASES = ["ant", "halfcheetah", "hopper", "cartpole", "humanoid_rich", "satellite_large"]


def main():
    ref = ref_harness.load()
    for name in CASES:
        cls = ENVS[name]
        with open(os.path.join(base._MODEL_DIR, cls._xml_path())) as f:
            xml = cls._patch_xml(f.read())
        lite = mjcf.from_xml_string(xml, base_dir=base._MODEL_DIR)
        m = ref_harness.put_model(ref, lite)
        dx0 = ref.forward.step(m, ref.io.make_data(m))
        rng = np.random.RandomState(11)
        noise = cls.RESET_NOISE_SCALE
        nact = cls.N_GIMBALS if hasattr(cls, "N_GIMBALS") else lite.nu
        qpos0 = dx0.qpos.numpy()[None] + rng.uniform(-noise, noise, size=(NENV, lite.nq))
        qvel0 = dx0.qvel.numpy()[None] + rng.uniform(-noise, noise, size=(NENV, lite.nv))
        if hasattr(cls, "N_GIMBALS"):
            qvel0[:, [7 + 2 * i for i in range(cls.N_GIMBALS)]] = cls.ROTOR_SPEED
        actions = rng.uniform(-1, 1, size=(NSTEP, NENV, nact))
        store = {"qpos0": qpos0, "qvel0": qvel0, "actions": actions, "dx0_qpos": dx0.qpos.numpy(), "dx0_qvel": dx0.qvel.numpy(),
                 "dx0_time": dx0.time.numpy()}
        for e in range(NENV):
            d = dx0.replace(qpos=torch.tensor(qpos0[e]), qvel=torch.tensor(qvel0[e]))
            for t in range(NSTEP):
                ctrl = actions[t, e]
                if hasattr(cls, "N_GIMBALS"):
                    ctrl = np.concatenate([ctrl, np.full(cls.N_GIMBALS, cls.ROTOR_SPEED)])
                d = d.replace(ctrl=torch.tensor(ctrl))
                for _ in range(cls.FRAME_SKIP):
                    d = ref.forward.step(m, d)
                for leaf in ("qpos", "qvel", "cinert", "cvel", "qfrc_actuator", "time"):
                    store[f"{leaf}/{t}/{e}"] = getattr(d, leaf).to(torch.float64).numpy().copy()
        store["meta"] = np.array(json.dumps(dict(env=name, frame_skip=cls.FRAME_SKIP, nenv=NENV, nstep=NSTEP, dt=float(lite.opt.timestep))))
        np.savez_compressed(os.path.join(GOLD, f"env_{name}.npz"), **store)
        print(name, "recorded; final qpos[:3]", d.qpos.numpy()[:3])


if __name__ == "__main__":
    main()
