"""Times the reference's OWN eager Python step on this container's CPU.

TEST INFRASTRUCTURE, container-only (needs /root/reference through oracle/ref_harness.py).  B = 1 Python loop,
float64, bench inputs (qvel = 0.01 * RandomState(42).randn(nv)); prints env-steps/s for cartpole and humanoid."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "mujoco-torch_amd"))
import ref_harness  # noqa: E402
from mujoco_torch_amd import mjcf  # noqa: E402

ref = ref_harness.load()
for name, overrides, nsteps in (("cartpole", {}, 50), ("humanoid", {"solver": 1}, 10)):
    lite = mjcf.from_xml_path(os.path.join(os.path.dirname(HERE), "mujoco-torch_amd", "mujoco_torch_amd", "test_data", name + ".xml"))
    for k, v in overrides.items():
        setattr(lite.opt, k, v)
    m = ref_harness.put_model(ref, lite)
    d = ref.io.make_data(m).replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(lite.nv)))
    d = ref.forward.step(m, d)  # warm
    t0 = time.perf_counter()
    for _ in range(nsteps):
        d = ref.forward.step(m, d)
    dt = time.perf_counter() - t0
    print(f"{name}: {nsteps / dt:.2f} env-steps/s (reference forward.step, eager, B=1, float64, torch {torch.__version__}, {torch.get_num_threads()} threads)")
