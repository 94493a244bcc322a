"""GPU box: does the idle time between torch.cuda.synchronize() and the first launch of a timed window change the window?  20-step windows of the headline loop after a synchronize followed by
0 / 0.1 / 0.3 / 1 / 3 ms of host sleep (median of 30 windows each, interleaved)."""
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
gaps = [0.0, 1e-4, 3e-4, 1e-3, 3e-3]
res = {g: [] for g in gaps}
for rep in range(30):
    for g in gaps:
        for _ in range(5): dg = mt.step(mdev, dg)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + g
        while time.perf_counter() < t_end: pass
        t0 = time.perf_counter()
        for _ in range(20): dg = mt.step(mdev, dg)
        torch.cuda.synchronize()
        res[g].append((time.perf_counter() - t0) / 20)
for g in gaps:
    print(f"idle {1e3 * g:4.1f} ms before the first launch: {1e6 * np.median(res[g]):6.1f} us per step (min {1e6 * min(res[g]):6.1f})")
