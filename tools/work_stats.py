"""Solver work per environment along the bench trajectory (CPU oracle counters): solver iterations, line-search iterations, rows.

usage: python tools/work_stats.py [ant|mesh|humanoid] [B] [steps...]
Diagnostic for the solver kernels' load balance: a wavefront that carries several environments runs as long as its slowest one."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "mujoco-torch_amd"))
import bench  # noqa: E402
import mujoco_torch_amd as mt  # noqa: E402
import pyoracle  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "mesh"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
marks = [int(a) for a in sys.argv[3:]] or [1, 5, 20, 60, 120, 220]
W = bench.WORKLOADS[wl]
lite = mt.mjcf.from_xml_path(mt.test_data_path(W["xml"] + ".xml"))
for k, v in W["overrides"].items():
    setattr(lite.opt, k, v)
mx = mt.device_put(lite, dtype=None if W["dtype"] == torch.float64 else W["dtype"])
dtype = W["dtype"]
d = mt.make_data(mx).expand(B).clone()
d = d.replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
if dtype != torch.float64:
    d = d.to(dtype)
lib = pyoracle.lib()
import ctypes
lib.mjo_set_work_stats.argtypes = [ctypes.c_void_p]
lib.mjo_set_work_stats.restype = None
stats = np.zeros((B, 4), dtype=np.int32)
for s in range(1, max(marks) + 1):
    lib.mjo_set_work_stats(stats.ctypes.data)
    d = pyoracle.apply(d, pyoracle.run(mx, d, step=True, nthreads=8))
    lib.mjo_set_work_stats(None)
    if s in marks:
        sol, nit, ls, rows = stats.T.astype(np.float64)
        q = lambda a: [float(np.round(np.quantile(a, x), 1)) for x in (0.5, 0.9, 0.99, 1.0)]
        g = lambda a, k: float(np.mean(np.max(a[: B // k * k].reshape(-1, k), axis=1)))
        print(f"step {s}: solves/step {sol.mean():.1f}  iters/solve mean {np.mean(nit / sol):.2f} q50/90/99/max {q(nit / sol)}  "
              f"ls/iter {ls.sum() / max(nit.sum(), 1):.2f}  ls/step mean {ls.mean():.1f} q {q(ls)}  rows/solve {np.mean(rows / sol):.1f} q {q(rows / sol)}  "
              f"mean of max over groups of 2: {g(ls, 2):.1f}  of 4: {g(ls, 4):.1f}", flush=True)
