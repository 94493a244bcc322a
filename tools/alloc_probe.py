"""GPU box: does the drop-in loop at B = 32768 keep asking the driver for memory?  Prints the caching allocator's device-allocation counter around windows of steps."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from _util import load_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mx = load_model("humanoid", {"solver": 1}, torch.float64)
d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
mdev, dg = mx.to("cuda"), d.to("cuda")
def stat(): 
    s = torch.cuda.memory_stats()
    return s["num_device_alloc"], s["num_device_free"], s["reserved_bytes.all.current"] >> 20, s["num_alloc_retries"]
print("start", stat())
for w in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(25): dg = mt.step(mdev, dg)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"window {w}: host {1e3 * (t1 - t0) / 25:.3f} ms/step, total {1e3 * (t2 - t0) / 25:.3f} ms/step, allocator (device allocs, frees, reserved MiB, retries) {stat()}")
