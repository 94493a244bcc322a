"""GPU box: per-launch kernel times of one step in launch order (RK4 models: which stage costs what).  usage: python tools/stage_times.py [ant|mesh|humanoid] [B]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from mujoco_torch_amd import native
from _util import load_model
which = sys.argv[1] if len(sys.argv) > 1 else "ant"
cfg = {"humanoid": ("humanoid", {"solver": 1}, torch.float64, 4096), "ant": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32, 16384),
       "mesh": ("mesh_contact", {}, torch.float32, 8192)}[which]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg[3]
mx = load_model(*cfg[:3])
d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
if cfg[2] != torch.float64: d = d.to(cfg[2])
mdev, dg = mx.to("cuda"), d.to("cuda")
for _ in range(150): dg = mt.step(mdev, dg)
lib = native.load_library()
lib.mjh_debug_phase_timing(1)
ms = (ctypes.c_float * 96)(); ids = (ctypes.c_int * 96)()
acc = None
N = 50
for _ in range(N):
    dg = mt.step(mdev, dg)
    n = lib.mjh_debug_phase_times(ms, ids, 96)
    seq = [(ids[i], ms[i]) for i in range(n)]
    if acc is None: acc = [[k, 0.0] for k, _ in seq]
    for i, (k, v) in enumerate(seq): acc[i][1] += v
lib.mjh_debug_phase_timing(0)
print(" ".join(f"({k}) {1e3 * v / N:.1f}" for k, v in acc))
