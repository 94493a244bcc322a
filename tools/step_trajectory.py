"""GPU box: device time of every step of the headline trajectory from its initial state (the bench's recipe), after a spin-up on a scratch copy: how much slower are the early steps
(the driver times steps 5 .. 25) than the steady state, and in which kernel?   python tools/step_trajectory.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from _util import load_model
B, N = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 260
mx = load_model("humanoid", {"solver": 1}, torch.float64)
d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
mdev, d0 = mx.to("cuda"), d.to("cuda")
s = d0.clone()
for _ in range(100): s = mt.step(mdev, s)
del s
acc = np.zeros(N)
for rep in range(5):
    dg = d0.clone()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        dg = mt.step(mdev, dg)
        ev[i + 1].record()
    torch.cuda.synchronize()
    acc += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(N)])
acc /= 5
print("step: device us   (mean of 5 replays)")
for i in list(range(0, 30)) + list(range(30, N, 10)):
    print(f"{i:4d}: {1e3 * acc[i]:7.1f}")
print(f"steps 5..24: {1e3 * acc[5:25].mean():.1f} us   steps 25..224: {1e3 * acc[25:225].mean():.1f} us   last 30: {1e3 * acc[-30:].mean():.1f} us")
out = mt.step(mdev, dg)
nc = (out.contact.dist < 0).sum(1).float()
print("active contacts per environment at the end:", float(nc.mean()))
