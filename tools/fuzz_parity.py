"""GPU box, one-off differential campaign: for every bundled / test model, large random batches (bigger perturbations than the
seeded unit tests: joint angles, un-normalised quaternions, velocities, controls, applied forces, warm starts) are stepped on the
GPU and checked leaf by leaf against the CPU oracle.  Prints one line per (model, dtype) and exits non-zero on a mismatch.

    python tools/fuzz_parity.py [B] [steps] [case indices into tests/_cases.py FUZZ_CASES, comma-separated]
"""
import os
import sys
import time
import traceback
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
from _cases import fuzz_batch  # noqa: E402
from _util import check_against_oracle, gpu_out_to_numpy  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from _cases import FUZZ_BAND, FUZZ_QUANTILE, FUZZ_CASES as CASES, FUZZ_TOL_PRE as TOL_PRE  # noqa: E402  (the bounds of the in-suite campaign, tests/test_gpu_parity.py::test_differential_campaign)
import _util  # noqa: E402

ONLY = {int(x) for x in sys.argv[3].split(",")} if len(sys.argv) > 3 else None
bad = 0
for ci, (xml, ov, dt, tol_sol) in enumerate(CASES):
    if ONLY is not None and ci not in ONLY:
        continue
    t0 = time.time()
    try:
        mx, d = fuzz_batch(xml, ov, dt, B)
        mdev, dg = mx.to("cuda"), d.to("cuda")
        fracs, tail = [], {}
        for s in range(STEPS):
            og = mt.step(mdev, dg)
            frac, worst = check_against_oracle(mx, dg.cpu(), gpu_out_to_numpy(og), TOL_PRE[dt], tol_sol, what=f"{xml} step{s}", nthreads=16, band=FUZZ_BAND.get(xml), tail_rules=True, tail_out=tail, quantile_tol=FUZZ_QUANTILE.get((xml, dt)))
            fracs.append((round(frac, 3), float(f"{worst:.1e}")))
            dg = og
        print(f"ok   {xml:22s} {str(ov):55s} {str(dt)[6:]:8s} B={B} (alt-branch frac, worst solver err) per step: {fracs}  tail {tail}  [{time.time() - t0:.0f}s]", flush=True)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print(f"FAIL {xml} {ov} {dt}: {str(ex)[:600]}", flush=True)
        traceback.print_exc(limit=2)
sys.exit(1 if bad else 0)
