"""GPU box: measures (does not judge) the agreement of the HIP step with the CPU oracle on every seeded and fuzz case, so that
the bounds asserted in tests/_cases.py are measured numbers.  Writes gpurun_out/parity_survey.json; environments whose solver
leaves match no admissible oracle branch at 1e-6 are dumped (inputs + GPU outputs) to gpurun_out/parity_outliers/ for study on CPU.

    python tools/parity_survey.py [fuzz batch] [steps]
"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
import _util  # noqa: E402
from _cases import FUZZ_CASES, SEEDED_CASES, case_id, fuzz_batch, seeded_batch  # noqa: E402
from _util import compare_with_oracle, gpu_out_to_numpy  # noqa: E402

FB = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(os.path.join(OUT, "parity_outliers"), exist_ok=True)
ALL_SOLVER_LEAVES = list(_util.SOLVER_LEAVES)
rows = []


def survey(kind, name, mx, d, dtype, steps):
    mdev, dg = mx.to("cuda"), d.to("cuda")
    per_step = []
    for s in range(steps):
        og = mt.step(mdev, dg)
        got = gpu_out_to_numpy(og)
        dc = dg.cpu()
        c = compare_with_oracle(mx, dc, got, nthreads=16)
        B = len(c["err_best"])
        worst_pre = max(c["pre"].items(), key=lambda kv: kv[1]) if c["pre"] else ("", 0.0)
        st = dict(step=s, pre_worst=c["pre_worst"], pre_worst_leaf=worst_pre[0], ints_ok=bool(c["ints_ok"]), sol_best_max=float(c["err_best"].max()),
                  sol_nat_max=float(c["err_nat"].max()), sol_best_p99=float(np.percentile(c["err_best"], 99)), n_alts=c["n_alts"],
                  alt_frac_1e8=float((c["err_nat"] > 1e-8).mean()), alt_frac_1e7=float((c["err_nat"] > 1e-7).mean()),
                  unmatched_1e8=int((c["err_best"] > 1e-8).sum()), unmatched_1e6=int((c["err_best"] > 1e-6).sum()),
                  tie_frac=float((c["tie_pairs"] > 0).mean()) if c["tie_pairs"] is not None else 0.0,
                  leaf_nat={k: float(v) for k, v in c["leaf_nat"].items()})
        per_step.append(st)
        if dtype == torch.float64:
            for e in np.nonzero(c["err_best"] > 1e-6)[0][:4]:
                e = int(e)
                np.savez_compressed(os.path.join(OUT, "parity_outliers", f"{kind}_{name}_s{s}_e{e}.npz"),
                                    **{f"in/{n}": _util.leaf(dc, n).numpy()[e] for n in _util.REAL_LEAVES + _util.INT_LEAVES},
                                    **{f"gpu/{n}": got[n][e] for n in got}, err_best=c["err_best"][e], err_nat=c["err_nat"][e])
        dg = og
    return per_step


for kind, cases in (("seeded", SEEDED_CASES), ("fuzz", [(x, o, d, FB, dict(tol_sol=t)) for x, o, d, t in FUZZ_CASES])):
    for c in cases:
        xml, ov, dt, B, bounds = c
        name = case_id(c)
        # float32: qfrc_constraint = J^T efc_force cancels forces of ~1e5 on the ant down to ~1e-2 -- rounding residue (fuzz only)
        _util.SOLVER_LEAVES[:] = [n for n in ALL_SOLVER_LEAVES if kind == "seeded" or dt == torch.float64 or n not in ("qfrc_constraint", "efc_force")]
        t0 = time.time()
        try:
            mx, d = (seeded_batch if kind == "seeded" else fuzz_batch)(xml, ov, dt, B)
            st = survey(kind, name, mx, d, dt, STEPS if kind == "seeded" else 2)
            rows.append(dict(kind=kind, case=name, dtype=str(dt)[6:], B=B, steps=st))
            print(f"{kind:6s} {name:70s} pre {max(s['pre_worst'] for s in st):.1e} best {max(s['sol_best_max'] for s in st):.1e} nat {max(s['sol_nat_max'] for s in st):.1e} "
                  f"alt@1e-8 {max(s['alt_frac_1e8'] for s in st):.3f} tie {max(s['tie_frac'] for s in st):.3f} unmatched@1e-8 {sum(s['unmatched_1e8'] for s in st)} [{time.time() - t0:.0f}s]", flush=True)
        except Exception as ex:  # noqa: BLE001
            rows.append(dict(kind=kind, case=name, error=str(ex)[:400]))
            print(f"ERROR {kind} {name}: {str(ex)[:300]}", flush=True)
            traceback.print_exc(limit=3)
        with open(os.path.join(OUT, "parity_survey.json"), "w") as f:
            json.dump(dict(note="HIP step vs CPU oracle; errors are max-norm per leaf (|diff|max / max(|want|max, floor)), see tests/_util.rel_err", rows=rows), f, indent=1)
