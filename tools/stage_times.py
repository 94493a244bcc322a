"""Per-stage time of the fused kernel via mjh_forward stage prefixes (GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from _util import load_model

def run(xml, ov, dtype, B, n=20):
    mx = load_model(xml, ov, dtype)
    d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
    if dtype != torch.float64: d = d.to(dtype)
    mdev, dg = mx.to("cuda"), d.to("cuda")
    prev = 0.0
    names = {0x01: "kinematics+com_pos", 0x03: "+crb+factor", 0x07: "+collision", 0x0f: "+make_constraint", 0x1f: "+velocity(rne)", 0x3f: "+actuation+accel", 0x7f: "+solve"}
    for st in [0x01, 0x03, 0x07, 0x0f, 0x1f, 0x3f, 0x7f]:
        for _ in range(3): mt.forward(mdev, dg, stages=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): mt.forward(mdev, dg, stages=st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{xml} B={B} stages {st:#04x} {names[st]:22s} {ms*1e3:9.1f} us  (+{(ms-prev)*1e3:8.1f} us)")
        prev = ms
    for _ in range(3): o = mt.step(mdev, dg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): o = mt.step(mdev, dg)
    e1.record(); torch.cuda.synchronize()
    print(f"{xml} B={B} full step {e0.elapsed_time(e1)/n*1e3:9.1f} us")

if __name__ == "__main__":
    run("humanoid", {"solver": 1}, torch.float64, 4096)
    run("humanoid", {"solver": 1}, torch.float64, 768)
    run("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32, 16384, n=5)
