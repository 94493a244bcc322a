"""GPU box: replays campaign cases of tools/fuzz_parity.py up to a failing step and saves that environment's input Data and the GPU's outputs
(gpurun_out/triage/<name>.pt: {"d": one-environment Data on the CPU, "got": {leaf: array[1, ...]}}) for analysis against the oracle off the box.

    python tools/fuzz_triage.py B  case_index:step:env[,env...] ...      (case_index into tests/_cases.py FUZZ_CASES)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
from _cases import FUZZ_CASES, fuzz_batch  # noqa: E402
from _util import gpu_out_to_numpy  # noqa: E402

B = int(sys.argv[1])
os.makedirs(os.path.join(ROOT, "gpurun_out", "triage"), exist_ok=True)
for spec in sys.argv[2:]:
    ci, step, envs = spec.split(":", 2)
    xml, ov, dt, _ = FUZZ_CASES[int(ci)]
    mx, d = fuzz_batch(xml, ov, dt, B)
    mdev, dg = mx.to("cuda"), d.to("cuda")
    for s in range(int(step)):
        dg = mt.step(mdev, dg)
    og = mt.step(mdev, dg)
    got, dc = gpu_out_to_numpy(og), dg.cpu()
    if envs.startswith("auto:"):  # the environment with the largest error of that leaf against the natural oracle run
        import numpy as np
        import pyoracle

        name = envs[5:]
        from _util import HINT_LEAVES

        nat = pyoracle.run(mx, dc, step=True, nthreads=16, **({"contact_hint": {k: got[k] for k in HINT_LEAVES}} if mx.constraint_sizes_py[3] > 0 else {}))
        diff = np.abs(np.asarray(got[name], dtype=np.float64) - np.asarray(nat[name], dtype=np.float64)).reshape(B, -1).max(1)
        envs = str(int(diff.argmax()))
        print("auto:", name, "worst env", envs, "abs diff", float(diff.max()), flush=True)
    for e in (int(x) for x in envs.split(",")):
        path = os.path.join(ROOT, "gpurun_out", "triage", f"{xml}_{ci}_s{step}_e{e}.pt")
        torch.save({"d": dc[e : e + 1].clone(), "got": {n: got[n][e : e + 1] for n in got}, "xml": xml, "ov": ov, "dtype": str(dt)}, path)
        print("saved", path, flush=True)
