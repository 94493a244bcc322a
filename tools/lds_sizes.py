"""GPU box: per-environment LDS arena sizes of the BASELINE models' kernels (bytes) and how many wavefront-workgroups of each fit a CU's 160 KB."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch
import mujoco_torch_amd as mt
from mujoco_torch_amd import native
from _util import load_model
lib = native.load_library()
lib.mjh_model_lds_bytes.restype = ctypes.c_int
for name, (xml, ov, dt) in {"humanoid": ("humanoid", {"solver": 1}, torch.float64), "ant": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32), "mesh": ("mesh_contact", {}, torch.float32)}.items():
    mx = load_model(xml, ov, dt).to("cuda")
    nm = native.get_native_model(mx, torch.device("cuda:0"), dt)
    h = nm.handle if hasattr(nm, "handle") else nm._handle
    sizes = {ph: lib.mjh_model_lds_bytes(h, ph) for ph in (0, 1, 2, 3, 4, 5, 16, 17, 18)}
    print(name, sizes, flush=True)
