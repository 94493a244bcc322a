"""Per-section shader-clock profile of the phase kernels from a -DMJH_STAMPS diagnostic build.

    (container) python tools/stamps.py build      (GPU box)  python tools/stamps.py [humanoid|ant|mesh] [B] [warm steps]
Builds lib/libmjhip_stamps.so (never shipped, never loaded by the package), runs a few steps with a stamp
buffer and prints mean cycles between consecutive stamps of each phase (lane-0 s_memtime; diagnostic build
times are not comparable with the production build -- read shares, not totals)."""
import ctypes, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "mujoco-torch_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, p))
import numpy as np, torch
lib = os.path.join(R, "mujoco-torch_amd", "lib", "libmjhip_stamps.so")
src = [os.path.join(R, "mujoco-torch_amd", "csrc", f) for f in os.listdir(os.path.join(R, "mujoco-torch_amd", "csrc")) if f.endswith((".h", ".hip"))]
if not os.path.exists(lib) or (os.environ.get("MJH_STAMPS_NOBUILD") != "1" and os.path.getmtime(lib) < max(os.path.getmtime(f) for f in src)):  # build it in the container: it travels with the snapshot
    csrc = os.path.join(R, "mujoco-torch_amd", "csrc")
    subprocess.run([os.path.join(csrc, "build.sh"), "-DMJH_STAMPS"] + os.environ.get("MJH_STAMPS_FLAGS", "").split(), check=True,  # e.g. MJH_STAMPS_FLAGS=-DMJH_STAMPS_STAGE=2
                   env=dict(os.environ, MJH_BUILD_DIR=os.path.join(csrc, "build", "stamps"), MJH_BUILD_OUT=lib))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
from mujoco_torch_amd import native
native.LIB_PATH = lib
import mujoco_torch_amd as mt
from _util import load_model
which = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = {"humanoid": ("humanoid", {"solver": 1}, torch.float64), "ant": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, torch.float32),
       "mesh": ("mesh_contact", {}, torch.float32)}[which]
mx = load_model(*cfg)
d = mt.make_data(mx).expand(B).clone().replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B, mx.nv)))
if cfg[2] != torch.float64: d = d.to(cfg[2])
mdev, dg = mx.to("cuda"), d.to("cuda")
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3): dg = mt.step(mdev, dg)
stamps = torch.zeros((B, 128), dtype=torch.int64, device="cuda")
native.load_library().mjh_debug_set_stamps.argtypes = [ctypes.c_void_p]
native.load_library().mjh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
dg = mt.step(mdev, dg); torch.cuda.synchronize()
native.load_library().mjh_debug_set_stamps(None)
st = stamps.cpu().numpy().astype(np.float64)
names = {8: "joint-local quats (sincos) + sync", 9: "level sweep: constants + levels", 0: "KIN start", 1: "load qpos", 2: "chain walk + frames", 3: "quat wb, geoms, sites, cams", 4: "stores kin", 5: "subtree com", 6: "cinert + cdof", 7: "stores com",
         10: "CRB start", 11: "loads", 12: "crb subtree sums", 13: "inert_mul", 14: "qM", 15: "stores", 16: "chol_factor", 17: "store qLD",
         19: "CON start", 22: "load geoms (+ the rows' inputs)", 24: "pair cull (RK4 stages 1..3)", 20: "narrow phase", 21: "contact stores", 23: "loads + zero rows", 25: "limit + contact rows", 26: "kbi / aref rows", 27: "efc stores",
         30: "VEL start", 31: "loads", 32: "transmission", 33: "com_vel chain", 35: "passive", 36: "rne cacc chain + local frc", 37: "cfrc subtree sums", 38: "qfrc_bias", 39: "stores", 40: "actuator forces", 41: "qfrc_actuator, xfrc, smooth", 42: "chol_solve + stores",
         50: "SOL start", 51: "all loads issued + waited", 52: "inv_diag + chol_solve (qacc_smooth)", 53: "store qacc_smooth", 54: "warm/smooth contexts", 55: "main context (+gradient)", 57: "LS: mulM, mulJ, dots", 58: "LS: quad", 59: "LS: points + loop + update", 60: "(linesearch end)", 61: "update_constraint/gradient/search", 62: "solve stores"}
names.update({70: "sol2: J rows gather (+ rest of the loads)", 71: "sol2: qacc_smooth solve", 72: "sol2: contexts (mulM2, mulJ2, costs, qfrc)", 73: "sol2: H build", 74: "sol2: H Cholesky",
              75: "sol2: tri solve (M or H)", 76: "sol2 LS: mulM, mulJ", 77: "sol2 LS: dots, points, loop", 78: "sol2: cost + J^T force", 79: "sol2: stores"})
names.update({80: "sol2 loads: qfrc_smooth, factor rows (T.t), qM rows", 81: "sol2 loads: state, limit rows", 82: "sol2 loads: contact_dist + compaction", 83: "sol2 loads: D / aref of the dense rows"})
names.update({99: "convex pair: wave start -> convex_convex", 90: "convex_convex: frames into LDS", 91: "SAT axes", 92: "best axis, support faces", 93: "polygons, side planes", 94: "edge clipping, reference points", 95: "manifold points", 96: "contacts out"})
names.update({63: "newton: H build", 64: "newton: factor H", 65: "newton: solve", 66: "J^T force", 67: "update_constraint", 68: "context init (mulJ, mulM) / loop head"})
# every slot holds the cycles ACCUMULATED in the section that ends at that stamp (loops add up), summed over RK stages
for lo, hi in [(0, 9), (10, 18), (19, 29), (30, 49), (50, 89), (90, 99)]:
    tot = 0
    for k in range(lo, hi + 1):
        v = st[:, k].mean()
        if v == 0: continue
        tot += v
        print(f"  [{k:2d}] {names.get(k, ''):40s} {v:10.0f} cycles   (max {st[:, k].max():10.0f})")
    print(f"  phase total {tot:10.0f} cycles")
