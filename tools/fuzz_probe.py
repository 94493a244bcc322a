"""GPU box: one saved campaign environment (tools/fuzz_triage.py) stepped under a grid of iteration caps, to find the solver iteration / line-search iteration at which
the GPU and the oracle part ways.   python tools/fuzz_probe.py triage_tmp/NAME.pt   (copy gpurun_out/triage/NAME.pt into triage_tmp/ first: git-ignored, but it travels to the box)   -> gpurun_out/triage/NAME.probe.pt = {(iterations, ls_iterations): qacc}"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
from _util import load_model  # noqa: E402

for path in sys.argv[1:]:
    t = torch.load(path, weights_only=False)
    dt = torch.float32 if "32" in t["dtype"] else torch.float64
    res = {}
    base_it = base_ls = None
    for it in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 100):
        for ls in (50,):
            ov = dict(t["ov"], iterations=it, ls_iterations=ls)
            mx = load_model(t["xml"], ov, dt)
            og = mt.step(mx.to("cuda"), t["d"].to("cuda"))
            res[(it, ls)] = {n: getattr(og, n).cpu().numpy() for n in ("qacc", "qvel", "qpos")}
    out = os.path.join(ROOT, "gpurun_out", "triage", os.path.basename(path)[:-3] + ".probe.pt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    torch.save(res, out)
    print("saved", out, flush=True)
