"""GPU box: per-step device time of the first steps after a synchronize (the driver times 20 steps: is the start of the window slower?)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from _util import load_model
B = 4096
mx = load_model("humanoid", {"solver": 1}, torch.float64)
d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
mdev, dg = mx.to("cuda"), d.to("cuda")
for _ in range(300): dg = mt.step(mdev, dg)
N = 24
acc = np.zeros(N); wall = []
for rep in range(20):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(N):
        dg = mt.step(mdev, dg)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall.append((time.perf_counter() - t0) / N)
    acc += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(N)])
print("per-step device ms (mean of 20 windows):", " ".join(f"{x / 20:.3f}" for x in acc))
print(f"wall per step over the window: {1e3 * np.mean(wall):.4f} ms")
