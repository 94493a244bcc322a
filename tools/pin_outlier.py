"""Build container: a campaign environment saved by tools/fuzz_triage.py (gpurun_out/triage/NAME.pt) -> tests/golden/outliers/<xml>_r04_s<step>_e<env>.npz: the environment's full
input Data (`in/<leaf>`), the outputs the HIP step produced for it on the GPU box (`got/<leaf>`) and, in `meta`, the rule of tests/_util.check_against_oracle that accounts
for it -- determined here by running the check on the recorded outputs.    python tools/pin_outlier.py NAME [NAME ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from _cases import FUZZ_BAND, FUZZ_TOL_PRE  # noqa: E402
from _util import INT_LEAVES, REAL_LEAVES, check_against_oracle, leaf, load_model  # noqa: E402

F32_ACC = "--f32-accuracy" in sys.argv
for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    t = torch.load(os.path.join(ROOT, "gpurun_out", "triage", name + ".pt"), weights_only=False)
    dt = torch.float32 if "32" in t["dtype"] else torch.float64
    mx = load_model(t["xml"], t["ov"], dt)
    d = t["d"][0]
    arrs = {"in/" + n: leaf(d, n).numpy() for n in REAL_LEAVES + INT_LEAVES}
    for n in ("cacc", "cfrc_int", "subtree_linvel", "subtree_angmom"):
        x = getattr(d, n, None)
        if isinstance(x, torch.Tensor) and x.numel() and float(x.abs().max()) > 0:
            arrs["in/" + n] = x.numpy()
    for n, v in t["got"].items():
        arrs["got/" + n] = np.asarray(v)[0]
    if F32_ACC:
        from _util import f32_accuracy_of

        parts = name.split("_")
        short = f"{t['xml']}_r05_{parts[-2]}_{parts[-1]}"
        acc = f32_accuracy_of(mx, t["xml"], t["ov"], d, {n: np.asarray(v)[0] for n, v in t["got"].items()})
        meta = dict(xml=t["xml"], overrides=t["ov"], dtype=str(dt)[6:], rule="f32_accuracy", measured=acc,
                    source=f"tools/fuzz_parity.py 8192 4 (round 5, profiles/r05/fuzz_parity_8192.txt), case {name}")
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "outliers", short + ".npz"), meta=json.dumps(meta), **arrs)
        print(short, "rule: f32_accuracy", acc, flush=True)
        continue
    d2 = torch.cat([t["d"], t["d"]])
    got2 = {n: np.concatenate([t["got"][n]] * 2) for n in t["got"]}
    tail = {}
    tol = 5e-3 if dt == torch.float32 else 1e-8
    check_against_oracle(mx, d2, got2, FUZZ_TOL_PRE[dt], tol, what=name, band=FUZZ_BAND.get(t["xml"]), tail_rules=True, tail_out=tail)
    fired = [k for k, v in tail.items() if v]
    rule = fired[0] if fired else ("band" if FUZZ_BAND.get(t["xml"]) else "branch")
    parts = name.split("_")
    short = f"{t['xml']}_r04_{parts[-2]}_{parts[-1]}"
    meta = dict(xml=t["xml"], overrides=t["ov"], dtype=str(dt)[6:], rule=rule, source=f"tools/fuzz_parity.py (round 4), case {name}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "outliers", short + ".npz"), meta=json.dumps(meta), **arrs)
    print(short, "rule:", rule, tail, flush=True)
