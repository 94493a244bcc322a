"""Dispatch mjh_forward with growing stage prefixes (B = one resident round and B = 4096); kernel durations
are read from a rocprofv3 --kernel-trace CSV of this run (tools/summarize_stage_trace.py)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
import mujoco_torch_amd as mt
from _util import load_model

xml, ov, dtype = (sys.argv[1], eval(sys.argv[2]), getattr(torch, sys.argv[3])) if len(sys.argv) > 3 else ("humanoid", {"solver": 1}, torch.float64)
mx = load_model(xml, ov, dtype)
for B in [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["768", "4096"])]:
    d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
    if dtype != torch.float64: d = d.to(dtype)
    mdev, dg = mx.to("cuda"), d.to("cuda")
    for st in [0x01, 0x03, 0x07, 0x0f, 0x1f, 0x3f, 0x7f]:
        for _ in range(4): mt.forward(mdev, dg, stages=st)
    for _ in range(4): mt.step(mdev, dg)
    torch.cuda.synchronize()
