"""bench.py -- env-steps/s of the native batched step on the BASELINE.json workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload humanoid|ant|cartpole]

One process per GPU (the driver launches N > 1 with torch.distributed.run); independent environments
are sharded across ranks with no data-path collective (weak scaling: B environments PER GPU).  A "step"
is one `mujoco_torch.step` over the whole resident batch; state is carried across steps (ping-pong
buffers), inputs follow the reference's bench recipe (benchmarks/_helpers.py:25-42: make_data state,
qvel = 0.01 * RandomState(42).randn(B, nv), ctrl = 0).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "mujoco-torch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mujoco_torch_amd as mt  # noqa: E402
from mujoco_torch_amd import native  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: humanoid.xml, batch 4096, Euler + CG, float64 (XML iterations=1, ls_iterations=4)
    "humanoid": dict(xml="humanoid", overrides={"solver": 1}, dtype=torch.float64, batch=4096,
                     name="humanoid.xml batch=4096/GPU Euler+CG float64 (iterations=1, ls_iterations=4)"),
    # configs[2]: ant.xml, RK4 + Newton, elliptic, float32
    "ant": dict(xml="ant", overrides={"integrator": 1, "solver": 2, "cone": 1}, dtype=torch.float32, batch=16384,
                name="ant.xml batch=16384/GPU RK4+Newton elliptic float32"),
    # configs[4]: plane + free box + free dodecahedron mesh, condim 6, Newton, float32
    "mesh": dict(xml="mesh_contact", overrides={}, dtype=torch.float32, batch=8192,
                 name="mesh_contact.xml (plane + box + dodecahedron mesh, condim 6) batch=8192/GPU Euler+Newton float32"),
    "cartpole": dict(xml="cartpole", overrides={}, dtype=torch.float64, batch=4096, name="cartpole.xml Euler float64"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_env_step(mx, dtype):
    """bytes of the Data leaves step() consumes + bytes of the leaves it produces."""
    from mujoco_torch_amd.forward import _written_names

    d = mt.make_data(mx)
    if dtype != torch.float64:
        d = d.to(dtype)
    written = _written_names(mx, step=True)
    nbytes = lambda n: native.data_field_tensor(d, n).numel() * native.data_field_tensor(d, n).element_size()
    out_b = sum(nbytes(n) for n in written)
    consumed = ["qpos", "qvel", "qacc", "act", "ctrl", "qfrc_applied", "xfrc_applied", "qacc_warmstart", "time", "qfrc_constraint"]
    in_b = sum(nbytes(n) for n in consumed)
    return in_b, out_b


# Data leaves each kernel of the step reads from / writes to global memory (mjh_kernels.h: the row_load / put calls of each phase).
# Kernel ids as mjh_debug_phase_times reports them; 5 = velocity phase with fluid forces, 6 = solver phase with frictionloss /
# equality rows (same I/O as 3 / 4).  "contact_*" stands for the per-contact leaves.
_CONTACT_OUT = ("contact_dist contact_pos contact_frame contact_includemargin contact_friction contact_solref contact_solreffriction "
                "contact_solimp contact_dim contact_geom1 contact_geom2 contact_geom contact_efc_address").split()
KERNEL_IO = {
    0: ("qpos", "qpos xpos xquat xmat xipos ximat xanchor xaxis geom_xpos geom_xmat site_xpos site_xmat cam_xpos cam_xmat light_xpos "
             "light_xdir subtree_com cdof cinert"),
    1: ("cinert cdof", "crb qM qLD"),
    2: ("geom_xpos geom_xmat subtree_com cdof qpos qvel", " ".join(_CONTACT_OUT) + " efc_J efc_D efc_aref efc_frictionloss"),
    3: ("qpos qvel act ctrl qfrc_applied xfrc_applied cdof cinert subtree_com xipos",
        "actuator_length actuator_moment actuator_velocity cvel cdof_dot qfrc_bias qfrc_passive actuator_force qfrc_actuator qfrc_smooth act_dot"),
    4: ("qLD qM efc_J efc_D efc_aref qfrc_smooth qacc_warmstart qpos qvel act act_dot time",
        "qacc_smooth qacc qacc_warmstart efc_force qfrc_constraint qpos qvel act time"),
    7: ("geom_xpos geom_xmat", "contact_dist contact_pos contact_frame"),
    8: ("site_xpos site_xmat geom_xpos geom_xmat cvel subtree_com qpos qvel", "sensordata"),
}
KERNEL_IO[5], KERNEL_IO[6] = KERNEL_IO[3], KERNEL_IO[4]
KERNEL_NAME = {0: "mjh_phase_kernel<{r}, 0, W> (kinematics)", 1: "mjh_phase_kernel<{r}, 1, 64> (crb / factor)", 2: "mjh_phase_kernel<{r}, 2, 64> (collision / constraint)",
               3: "mjh_phase_kernel<{r}, 3, W> (velocity)", 4: "mjh_phase_kernel<{r}, 4, 64> (solve / integrate)", 5: "mjh_phase_kernel<{r}, 5, W> (velocity + fluid)",
               6: "mjh_phase_kernel<{r}, 6, 64> (solve / integrate, general rows)", 7: "mjh_convex_kernel<{r}>", 8: "mjh_sensor_kernel<{r}>"}


def kernel_algorithmic_bytes(mx, dtype):
    """{kernel id: bytes of the Data leaves one environment's launch of that kernel reads + writes} (leaves absent from the model count 0)."""
    d = mt.make_data(mx)
    if dtype != torch.float64:
        d = d.to(dtype)

    def nbytes(name):
        t = native.data_field_tensor(d, name)
        return 0 if t is None else t.numel() * t.element_size()

    return {k: sum(nbytes(n) for n in r.split()) + sum(nbytes(n) for n in w.split()) for k, (r, w) in KERNEL_IO.items()}


def per_kernel_times(mdev, bufs, cur, steps, device):
    """Average duration of every kernel of a step, from HIP events recorded on the launch stream around each launch."""
    import ctypes

    lib = native.load_library()
    lib.mjh_debug_phase_timing(1)
    ms = (ctypes.c_float * 96)()
    ids = (ctypes.c_int * 96)()
    tot, cnt = {}, {}
    try:
        for _ in range(steps):
            mt.step(mdev, bufs[cur], out=bufs[1 - cur])
            cur = 1 - cur
            n = lib.mjh_debug_phase_times(ms, ids, 96)
            if n < 0:
                raise RuntimeError(lib.mjh_last_error().decode())
            for i in range(n):
                tot[ids[i]] = tot.get(ids[i], 0.0) + ms[i]
                cnt[ids[i]] = cnt.get(ids[i], 0) + 1
    finally:
        lib.mjh_debug_phase_timing(0)
    torch.cuda.synchronize(device)
    return {k: dict(avg_ms=tot[k] / cnt[k], launches_per_step=cnt[k] / steps, ms_per_step=tot[k] / steps) for k in tot}, cur


def build_inputs(mx, B, dtype, device, seed=42):
    d = mt.make_data(mx).expand(B).clone()
    d = d.replace(qvel=torch.tensor(0.01 * np.random.RandomState(seed).randn(B, mx.nv)))
    if dtype != torch.float64:
        d = d.to(dtype)
    return d.to(device)


def cpu_baseline(mx, dtype, B_sample, steps):
    """Times the CPU oracle (scalar C restatement of the reference step, OpenMP over envs) on host cores."""
    import pyoracle

    pyoracle.build()
    threads = pyoracle.lib().mjo_max_threads()
    d = mt.make_data(mx).expand(B_sample).clone()
    d = d.replace(qvel=torch.tensor(0.01 * np.random.RandomState(42).randn(B_sample, mx.nv)))
    if dtype != torch.float64:
        d = d.to(dtype)
    d = pyoracle.apply(d, pyoracle.run(mx, d, step=True, nthreads=threads))  # warm
    t0 = time.perf_counter()
    done = 0
    while True:  # a bounded sample: at least `steps` steps and ~12 s of CPU work, at most 40 s
        d = pyoracle.apply(d, pyoracle.run(mx, d, step=True, nthreads=threads))
        done += 1
        dt = time.perf_counter() - t0
        if (done >= steps and dt >= 12.0) or dt >= 40.0:
            break
    return dict(value=B_sample * done / dt, unit="env-steps/s", cores=threads, kind="port",
                sample=f"{B_sample} envs x {done} steps, oracle/mjoracle.c with OpenMP over environments ({dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="environments per GPU (default: the workload's)")
    ap.add_argument("--workload", default="humanoid", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for a one-GPU box: MJH_BENCH_BACKEND=gloo + MJH_BENCH_SHARE_GPU=1 run every rank on cuda:0 (RCCL refuses
    # two ranks on one device); the driver's multi-GPU runs use the defaults (RCCL, one GPU per rank)
    backend = os.environ.get("MJH_BENCH_BACKEND", "nccl")
    if os.environ.get("MJH_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    wl = WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    dtype = wl["dtype"]
    lite = mt.mjcf.from_xml_path(os.path.join(ROOT, "tests", "golden", "models", wl["xml"] + ".xml"))
    for k, v in wl["overrides"].items():
        setattr(lite.opt, k, v)
    mx = mt.device_put(lite, dtype=None if dtype == torch.float64 else dtype)
    mdev = mx.to(device)
    # different seeds per rank: independent environments, no collective on the data path
    bufs = [build_inputs(mx, B, dtype, device, seed=42 + rank), None]
    bufs[1] = bufs[0].clone()

    def run(n, cur):
        for _ in range(n):
            mt.step(mdev, bufs[cur], out=bufs[1 - cur])
            cur = 1 - cur
        return cur

    cur = run(args.warmup, 0)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ev0.record()
    cur = run(args.steps, cur)
    ev1.record()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # device time of one step (its phase kernels) on this stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final = bufs[cur]
    assert torch.isfinite(final.qpos).all(), "non-finite state after the timed steps"
    kernels = None
    if rank == 0:  # per-kernel durations of the same loop (events around every launch; outside the timed region)
        kernels, cur = per_kernel_times(mdev, bufs, cur, min(args.steps, 50), device)

    if rank == 0:
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"hbm_traffic_{args.workload}_b{B}_{'f64' if dtype == torch.float64 else 'f32'}.json")
        if os.path.exists(tfile):  # PMC counters cannot be collected from inside this process: committed rocprofv3 result
            with open(tfile) as f:
                traffic = json.load(f)["hbm_bytes_per_step"]
        in_b, out_b = algorithmic_bytes_per_env_step(mx, dtype)
        alg = in_b + out_b
        achieved = alg * B / (kernel_ms * 1e-3) / 1e9
        value = B * world * args.steps / elapsed
        rname = "double" if dtype == torch.float64 else "float"
        kbytes = kernel_algorithmic_bytes(mx, dtype)
        per_kernel = []
        for k, t in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"]):
            gbs = kbytes[k] * B / (t["avg_ms"] * 1e-3) / 1e9
            per_kernel.append({"kernel": KERNEL_NAME[k].format(r=rname), "avg_us": 1e3 * t["avg_ms"], "launches_per_step": t["launches_per_step"],
                               "algorithmic_bytes_per_env": kbytes[k], "achieved": gbs, "frac": gbs / HBM_PEAK_GBS})
        dom = per_kernel[0]
        ktraffic = None
        if os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            for name, v in tj.get("kernels", {}).items():  # PMC bytes per launch of the dominant kernel (FETCH_SIZE corrected x2)
                if dom["kernel"].split(" (")[0].replace(" ", "").replace("W>", "") in name.replace(" ", ""):
                    ktraffic = 2 * 1024 * v["FETCH_SIZE_KB_raw_mean"] + 1024 * v["WRITE_SIZE_KB_mean"]
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if dtype == torch.float64 else "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "envs_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"independent-envs x{world} (no collectives)",
                       "lds_bytes_per_env_by_phase": native.get_native_model(mdev, device, dtype).lds_bytes},
            # dominant kernel of the step: its Data-leaf bytes per launch / its average launch duration (HIP events around each launch,
            # on the launch stream); "step" = the same for the whole launch sequence of a step (SURVEY section 8(d) per-unit figure)
            "roofline": {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": ktraffic,
                         "kernel": dom["kernel"], "kernel_avg_us": dom["avg_us"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_env"] * B,
                         "per_kernel": per_kernel,
                         "step": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_env_step": alg,
                                  "algorithmic_bytes_per_step": alg * B, "kernel_ms": kernel_ms,
                                  "kernels": "every launch of one step (torch events on the launch stream over the timed region)"}},
        }
        if not args.no_cpu_baseline and world == 1:
            nB = min(B, 4096)
            line["cpu_baseline"] = cpu_baseline(mx, dtype, nB, 20 if args.workload != "ant" else 4)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # rank 0 runs the per-kernel pass after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
