"""GPU box: what a float32 campaign case's solver-leaf error IS (pendula RK4 + CG measured 4.9e-3 against a 5e-3 bound).

For each step of the campaign batch, per environment: err(GPU float32, oracle float32) -- what the campaign bounds --, err(oracle float32, oracle float64 of the same inputs) -- the
float32 oracle's own distance from the float64 solution: rounding + CG stall --, and err(GPU float32, oracle float64).  If the last two are the same size as the first, the first is
float32 noise of the case, not a kernel difference.   usage: python tools/f32_yardstick.py [case index into FUZZ_CASES] [B] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mujoco-torch_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import mujoco_torch_amd as mt, pyoracle
from _cases import FUZZ_CASES, fuzz_batch
from _util import SOLVER_LEAVES, gpu_out_to_numpy, load_model, solver_err
ci = int(sys.argv[1]) if len(sys.argv) > 1 else [i for i, c in enumerate(FUZZ_CASES) if c[0] == "pendula" and c[2] == torch.float32][0]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
xml, ov, dt, tol = FUZZ_CASES[ci]
mx, d = fuzz_batch(xml, ov, dt, B)
mx64 = load_model(xml, ov, torch.float64)
mdev, dg = mx.to("cuda"), d.to("cuda")
print(f"case {ci}: {xml} {ov} {dt} B={B} bound {tol}")
for s in range(steps):
    og = mt.step(mdev, dg)
    got = gpu_out_to_numpy(og)
    dc = dg.cpu()
    w32 = pyoracle.run(mx, dc, step=True, nthreads=16)
    w64 = pyoracle.run(mx64, dc.to(torch.float64), step=True, nthreads=16)
    e_gpu_o32 = np.array([solver_err({n: got[n][e] for n in SOLVER_LEAVES}, {n: w32[n][e] for n in SOLVER_LEAVES}) for e in range(B)])
    e_o32_o64 = np.array([solver_err({n: w32[n][e] for n in SOLVER_LEAVES}, {n: w64[n][e] for n in SOLVER_LEAVES}) for e in range(B)])
    e_gpu_o64 = np.array([solver_err({n: got[n][e] for n in SOLVER_LEAVES}, {n: w64[n][e] for n in SOLVER_LEAVES}) for e in range(B)])
    q = lambda a: " ".join(f"{np.quantile(a, p):.1e}" for p in (0.5, 0.9, 0.99, 0.999, 1.0))
    print(f"step {s}: quantiles 50/90/99/99.9/100 %   GPU32-vs-oracle32 {q(e_gpu_o32)}   oracle32-vs-oracle64 {q(e_o32_o64)}   GPU32-vs-oracle64 {q(e_gpu_o64)}")
    worst = np.argsort(-e_gpu_o32)[:5]
    print("   worst environments (GPU32-vs-oracle32 | oracle32-vs-oracle64 | GPU32-vs-oracle64): " + "  ".join(f"e{e}: {e_gpu_o32[e]:.1e} | {e_o32_o64[e]:.1e} | {e_gpu_o64[e]:.1e}" for e in worst), flush=True)
    dg = og
