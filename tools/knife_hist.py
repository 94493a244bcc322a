"""Build container (CPU): histogram of |d0| / |d0(0)| over EVERY candidate the oracle's exact line search evaluates (solver.py:424-467 restated in
oracle/mjoracle_impl.h, linesearch) on BASELINE configs 2, 3, 5 at their FULL batches, over the first steps of the bench trajectory (bench.py's inputs).
VERDICT r05 weak 2: the band below which a candidate counts as "rounding noise" was 1e-8 (float64) / 1e-4 (float32) with nothing committed that showed
where the flagged candidates actually sit.  Output: profiles/r06/knife_hist_<config>.txt.

    python tools/knife_hist.py [humanoid|ant|mesh ...] [--steps N]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
import pyoracle  # noqa: E402
from mujoco_torch_amd import native  # noqa: E402

CONFIGS = {
    "humanoid": dict(xml="humanoid", ov={"solver": 1}, dtype=torch.float64, B=4096, steps=40, title="config 2: humanoid.xml B=4096 Euler+CG float64 (iterations=1, ls_iterations=4)"),
    "ant": dict(xml="ant", ov={"integrator": 1, "solver": 2, "cone": 1}, dtype=torch.float32, B=16384, steps=6, title="config 3: ant.xml B=16384 RK4+Newton elliptic float32"),
    "mesh": dict(xml="mesh_contact", ov={}, dtype=torch.float32, B=8192, steps=12, title="config 5: mesh_contact.xml B=8192 Euler+Newton float32"),
    # float64 twins of the float32 configs: the band of the float64 arithmetic on the Newton workloads
    "ant_f64": dict(xml="ant", ov={"integrator": 1, "solver": 2, "cone": 1}, dtype=torch.float64, B=4096, steps=4, title="float64 twin of config 3 (B=4096)"),
    "mesh_f64": dict(xml="mesh_contact", ov={}, dtype=torch.float64, B=4096, steps=8, title="float64 twin of config 5 (B=4096)"),
}
NB = pyoracle.KNIFE_BINS


def label(b):
    if b == 0:
        return "== 0 exactly     "
    if b == NB - 1:
        return ">= 1e+02         "
    k = b - 1
    return f"[1e{k - 30:+03d}, 1e{k - 29:+03d})   " if k > 0 else "< 1e-29          "


def run(key, steps):
    c = CONFIGS[key]
    lite = mt.mjcf.from_xml_path(mt.test_data_path(c["xml"] + ".xml"))
    for k, v in c["ov"].items():
        setattr(lite.opt, k, v)
    dtype = c["dtype"]
    mx = mt.device_put(lite, dtype=None if dtype == torch.float64 else dtype)
    B = c["B"]
    d = mt.make_data(mx).expand(B).clone()
    d = d.replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
    if dtype != torch.float64:
        d = d.to(dtype)
    band = pyoracle.knife_band(dtype)
    t0 = time.time()
    hist, fresh, knife = pyoracle.knife_histogram(mx, d, steps)
    flagged_envsteps = int((knife > 0).sum())
    lines = [f"# {c['title']}: {B} environments x {steps} steps of the bench trajectory (bench.py inputs: make_data state, qvel = 0.01 * RandomState(42).randn), oracle natural run, {time.time() - t0:.0f} s",
             f"# every candidate of the exact line search (lo-Newton, hi-Newton, midpoint per iteration; solver.py:431-467): ratio = |d0(candidate)| / |d0(alpha = 0)|",
             f"# 'fresh' = candidates that are not an end point of the current bracket: the only ones the band can flag",
             f"# band in force: ratio < {band:g}  ->  environment-steps with at least one flagged candidate: {flagged_envsteps} of {B * steps} ({flagged_envsteps / (B * steps):.4f})",
             f"# {'ratio':17s} {'all candidates':>16s} {'fresh':>14s}   flagged by the band"]
    eb = int(np.floor(np.log10(band))) + 30 + 1  # first bin at or above the band
    for i in range(NB):
        if hist[i] == 0 and fresh[i] == 0:
            continue
        lines.append(f"  {label(i)} {hist[i]:16d} {fresh[i]:14d}   {'yes' if (i == 0 or i < eb) else ''}")
    nz = [i for i in range(1, NB) if fresh[i]]
    below = [i for i in nz if i < eb]
    # the rounding cluster: the populated decades of fresh candidates below the band, and the gap to the first populated decade above them
    if below:
        # find the largest empty run of decades between populated bins (cluster edge)
        gaps = [(nz[j + 1] - nz[j] - 1, nz[j], nz[j + 1]) for j in range(len(nz) - 1)]
        g = max(gaps) if gaps else (0, nz[-1], nz[-1])
        lines.append(f"# widest empty run of decades among fresh candidates: {g[0]} decades, between {label(g[1]).strip()} and {label(g[2]).strip()}")
    lines.append(f"# totals: {int(hist.sum())} candidates, {int(fresh.sum())} fresh, {int(fresh[:eb].sum())} fresh candidates inside the band ({int(fresh[0])} of them exactly zero)")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps_override = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
    if steps_override:
        args = [a for a in args if a != str(steps_override)]
    for key in (args or ["humanoid", "ant", "mesh"]):
        txt = run(key, steps_override or CONFIGS[key]["steps"])
        out = os.path.join(ROOT, "profiles", "r06", f"knife_hist_{key}.txt")
        with open(out, "w") as f:
            f.write(txt)
        print(txt, flush=True)
