"""GPU box: host time of `d = step(mx, d)` at B = 2, split into the C call (mjh_step: four launches) and the Python around it."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch
import mujoco_torch_amd as mt
from mujoco_torch_amd import native
from _util import load_model
mx = load_model("humanoid", {"solver": 1}, torch.float64)
mdev = mx.to("cuda")
d = mt.make_data(mx).expand(2).clone().to("cuda")
for _ in range(200): d = mt.step(mdev, d)
torch.cuda.synchronize()
nm = native.get_native_model(mdev, torch.device("cuda", 0), torch.float64)
real = nm.lib.mjh_step
acc = [0.0]
class Wrap:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k != "mjh_step": return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); acc[0] += time.perf_counter() - t0; return r
        return g
nm.lib = Wrap(nm.lib)
N = 3000
t0 = time.perf_counter()
for _ in range(N): d = mt.step(mdev, d)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"per call: total host {1e6 * (t1 - t0) / N:.1f} us, inside mjh_step {1e6 * acc[0] / N:.1f} us")
