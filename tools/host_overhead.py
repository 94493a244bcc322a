"""(GPU box) host-side cost of one mt.step call (ping-pong buffers) vs the device time, at small batches."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "mujoco-torch_amd"); sys.path.insert(0, "oracle")
import numpy as np, torch, mujoco_torch_amd as mt
from _util import load_model
mx = load_model("humanoid", {"solver": 1})
for B in (64, 256, 1024, 4096):
    d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(0).randn(B, mx.nv)))
    mdev = mx.to("cuda"); bufs = [d.to("cuda"), d.to("cuda").clone()]
    cur = 0
    for _ in range(20): mt.step(mdev, bufs[cur], out=bufs[1 - cur]); cur = 1 - cur
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): mt.step(mdev, bufs[cur], out=bufs[1 - cur]); cur = 1 - cur
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    print(f"B={B}: host issue {1e6 * t_issue / n:.0f} us/step, wall {1e6 * t_total / n:.0f} us/step, {B * n / t_total / 1e6:.2f} M env-steps/s")
