"""GPU box: host time per `d = step(mx, d)` call against the device time of the step (humanoid, B = 4096, float64).
The loop is device-bound only while the host enqueues a step faster than the device runs it."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from _util import load_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mx = load_model("humanoid", {"solver": 1}, torch.float64)
d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
mdev, dg = mx.to("cuda"), d.to("cuda")
for _ in range(200): dg = mt.step(mdev, dg)
torch.cuda.synchronize()
for n in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(n): dg = mt.step(mdev, dg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step")
# host alone: tiny batch (device time negligible)
d1 = mt.make_data(mx).expand(2).clone().to("cuda")
for _ in range(50): d1 = mt.step(mdev, d1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): d1 = mt.step(mdev, d1)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"B=2: host {1e6 * (t1 - t0) / 1000:.1f} us/step, total {1e6 * (time.perf_counter() - t0) / 1000:.1f}")
