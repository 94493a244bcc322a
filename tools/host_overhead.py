"""GPU box: host time per `d = step(mx, d)` call against the device time of the step (humanoid, float64).
The host cost is taken over the first 12 calls after a synchronize (empty queue: a launch never waits for a slot), the device time from a long run."""
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
host = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12): dg = mt.step(mdev, dg)
    host.append((time.perf_counter() - t0) / 12)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000): dg = mt.step(mdev, dg)
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / 1000
print(f"B={B}: host {1e6 * np.median(host):.1f} us per call (median of 50 bursts of 12 calls on an empty queue; min {1e6 * min(host):.1f}, max {1e6 * max(host):.1f}); device-bound loop {1e6 * dev:.1f} us per step")
