"""ctypes wrapper of the CPU oracle (oracle/_build/libmjoracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Reuses the product's model packing (``mujoco_torch_amd.native.pack_model``) because the
oracle takes the very same ``mjhModelDesc`` / ``mjhData`` structs as the device library, with host
pointers -- the oracle never feeds the product.
"""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "mujoco-torch_amd"))

from mujoco_torch_amd import native  # noqa: E402

LIB = os.path.join(HERE, "_build", "libmjoracle.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or shutil.which("make"):
            build()  # make is a no-op when the library is newer than its sources and include/mjhip.h
        _lib = ctypes.CDLL(LIB)
        _lib.mjo_step.argtypes = [ctypes.POINTER(native.ModelDesc), ctypes.POINTER(native.DataPtrs), ctypes.POINTER(native.DataPtrs), ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        _lib.mjo_forward.argtypes = [ctypes.POINTER(native.ModelDesc), ctypes.POINTER(native.DataPtrs), ctypes.POINTER(native.DataPtrs), ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        _lib.mjo_max_threads.restype = ctypes.c_int
        _lib.mjo_set_contact_hint.argtypes = [ctypes.c_void_p] * 4
        _lib.mjo_set_contact_hint.restype = None
        _lib.mjo_set_stage_tie_flip.argtypes = [ctypes.c_uint, ctypes.c_void_p]
        _lib.mjo_set_stage_tie_flip.restype = None
        _lib.mjo_set_knife_band.argtypes = [ctypes.c_double, ctypes.c_double]
        _lib.mjo_set_knife_band.restype = None
        _lib.mjo_get_knife_band.argtypes = [ctypes.POINTER(ctypes.c_double)] * 2
        _lib.mjo_get_knife_band.restype = None
        _lib.mjo_set_knife_hist.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.mjo_set_knife_hist.restype = None
    return _lib


ALL_NAMES = native.LISTS["MJH_DATA_REALS"] + native.LISTS["MJH_DATA_I32"] + native.LISTS["MJH_DATA_I64"]


def data_to_numpy(d):
    """Data (torch, CPU) -> {abi leaf name: contiguous numpy array}."""
    out = {}
    for n in ALL_NAMES:
        t = native.data_field_tensor(d, n)
        out[n] = np.array(t.detach().cpu().contiguous().numpy(), order="C", copy=True)
    return out


EXTRA_IN = native.LISTS["MJH_DATA_EXTRA_IN"]  # input-only leaves the sensors read (cacc, cfrc_int, subtree_linvel, subtree_angmom); absent = zeros


def extra_inputs(d):
    """{name: numpy array} of the trailing input-only leaves the Data carries."""
    out = {}
    for n in EXTRA_IN:
        t = d._fields.get(n) if hasattr(d, "_fields") else getattr(d, n, None)
        if isinstance(t, torch.Tensor) and t.numel():
            out[n] = np.array(t.detach().cpu().contiguous().numpy(), order="C", copy=True)
    return out


def _ptrs(arrs):
    p = native.DataPtrs()
    for n, a in arrs.items():
        setattr(p, n, a.ctypes.data if a.size else None)
    return p


def run(m, d, step=True, stages=native.STAGE_ALL, fixed_iterations=False, nthreads=1, knife=None, knife_policy=-1, contact_hint=None, tie_pairs=None,
        stage_tie_flip=0, stage_ties=None):
    """Runs the oracle on a (possibly batched) CPU Data; returns {leaf: numpy array} of outputs.

    ``knife``: optional int32 array [B]; receives per env the number of line-search candidates whose
    derivative was rounding noise (the reference's result is implementation-defined on such steps).
    ``contact_hint``: optional {"contact_dist", "contact_pos", "contact_frame"} arrays (the outputs under test): where
    an index selection of the convex narrow phase is decided by rounding noise, the oracle keeps the admissible
    outcome closest to the hint (natural pick on equality); ``tie_pairs`` (int32 [B]) receives how many geom pairs
    per env ended on a non-natural outcome.
    ``stage_tie_flip``: bit mask of the narrow-phase tie events INSIDE RK4 stages 1..3 (which no hint can reach; per environment,
    in order of occurrence) that take their second candidate, 0 = none; ``stage_ties`` (int32 [B]) receives the number of such events.
    ``knife_policy``: -1 natural rounding; j >= 0 forces the first j such candidates to read as an exact
    zero (rejected by both bracket tests, solver.py:440-449) and the next one as non-zero (accepted)."""
    dtype = d.qpos.dtype
    desc, keep = native.pack_model(m, dtype)
    inp = data_to_numpy(d)
    batch = tuple(d.qpos.shape[:-1])
    B = int(np.prod(batch)) if batch else 1
    out = {n: np.array(a, copy=True) for n, a in inp.items()}
    extra = extra_inputs(d)  # (kept alive until the call returns: the struct only holds raw addresses)
    pin, pout = _ptrs({**inp, **extra}), _ptrs(out)
    flags = 1 if fixed_iterations else 0
    dt = 0 if dtype == torch.float64 else 1
    hint = None
    if contact_hint is not None and inp["contact_dist"].size:
        npdt = inp["contact_dist"].dtype
        hint = [np.array(np.asarray(contact_hint[k]).reshape(inp[k].shape), dtype=npdt, order="C", copy=True) for k in ("contact_dist", "contact_pos", "contact_frame")]
        lib().mjo_set_contact_hint(hint[0].ctypes.data, hint[1].ctypes.data, hint[2].ctypes.data, tie_pairs.ctypes.data if tie_pairs is not None else None)
    lib().mjo_set_stage_tie_flip(int(stage_tie_flip), stage_ties.ctypes.data if stage_ties is not None else None)
    try:
        rc = _call(step, desc, pin, pout, B, dt, stages, flags, nthreads, knife, knife_policy)
    finally:
        lib().mjo_set_contact_hint(None, None, None, None)
        lib().mjo_set_stage_tie_flip(0, None)
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    return out


def _call(step, desc, pin, pout, B, dt, stages, flags, nthreads, knife, knife_policy):
    if step:
        rc = lib().mjo_step(ctypes.byref(desc), ctypes.byref(pin), ctypes.byref(pout), B, dt, flags, nthreads, knife.ctypes.data if knife is not None else None, knife_policy)
    else:
        rc = lib().mjo_forward(ctypes.byref(desc), ctypes.byref(pin), ctypes.byref(pout), B, dt, stages, flags, nthreads, knife.ctypes.data if knife is not None else None, knife_policy)
    return rc


def time_steps(m, d, nthreads, min_steps, t_min, t_max):
    """bench.py's cpu_baseline leg: (env-steps/s, steps, seconds) of the oracle's C step alone on a state that is carried from step to step --
    model and Data are packed once, the two buffer sets ping-pong, and only the `mjo_step` calls sit between the clock reads (going through
    run() / apply() every step spends most of the time copying ~50 KB per environment through numpy and torch)."""
    import time

    dtype = d.qpos.dtype
    desc, keep = native.pack_model(m, dtype)
    a = data_to_numpy(d)
    b = {n: np.array(x, copy=True) for n, x in a.items()}
    batch = tuple(d.qpos.shape[:-1])
    B = int(np.prod(batch)) if batch else 1
    pa, pb = _ptrs(a), _ptrs(b)
    dt = 0 if dtype == torch.float64 else 1
    lib().mjo_set_contact_hint(None, None, None, None)
    lib().mjo_set_stage_tie_flip(0, None)
    if _call(True, desc, pa, pb, B, dt, native.STAGE_ALL, 0, nthreads, None, -1) != 0:  # warm (pages, thread pool)
        raise RuntimeError("oracle failed")
    pa, pb = pb, pa
    done, t0 = 0, time.perf_counter()
    while True:
        if _call(True, desc, pa, pb, B, dt, native.STAGE_ALL, 0, nthreads, None, -1) != 0:
            raise RuntimeError("oracle failed")
        pa, pb = pb, pa
        done += 1
        el = time.perf_counter() - t0
        if (done >= min_steps and el >= t_min) or el >= t_max:
            break
    return B * done / el, done, el


KNIFE_BINS = 34  # mjoracle.c: bin 0 = exactly zero, bin 1 + k = [1e(k-30), 1e(k-29)) (k = 0: everything below 1e-29), last bin = >= 1e2


def knife_band(dtype=None):
    """The line search's noise band in force, (float64, float32) or the one of `dtype`."""
    f64, f32 = ctypes.c_double(), ctypes.c_double()
    lib().mjo_get_knife_band(ctypes.byref(f64), ctypes.byref(f32))
    return (f64.value, f32.value) if dtype is None else (f64.value if dtype == torch.float64 else f32.value)


def knife_histogram(m, d, steps, nthreads=0, band=None):
    """`steps` steps of the oracle's natural run from the batched Data `d` (state carried, buffers ping-pong) with the line-search candidate histogram on:
    -> (hist[KNIFE_BINS] over every candidate, the same over candidates that are not a bracket end point, [steps, B] flagged-candidate counts).
    `band`: (float64, float32) bands for this run only."""
    dtype = d.qpos.dtype
    desc, keep = native.pack_model(m, dtype)
    a = data_to_numpy(d)
    b = {n: np.array(x, copy=True) for n, x in a.items()}
    B = int(np.prod(tuple(d.qpos.shape[:-1])))
    pa, pb = _ptrs(a), _ptrs(b)
    dt = 0 if dtype == torch.float64 else 1
    L = lib()
    L.mjo_set_contact_hint(None, None, None, None)
    L.mjo_set_stage_tie_flip(0, None)
    hist, fresh = np.zeros(KNIFE_BINS, np.int64), np.zeros(KNIFE_BINS, np.int64)
    knife = np.zeros((steps, B), np.int32)
    old = knife_band()
    if band is not None:
        L.mjo_set_knife_band(*band)
    L.mjo_set_knife_hist(hist.ctypes.data, fresh.ctypes.data)
    try:
        for s in range(steps):
            if _call(True, desc, pa, pb, B, dt, native.STAGE_ALL, 0, nthreads, knife[s], -1) != 0:
                raise RuntimeError("oracle failed")
            pa, pb = pb, pa
    finally:
        L.mjo_set_knife_hist(None, None)
        L.mjo_set_knife_band(*old)
    return hist, fresh, knife


def apply(d, out):
    """Returns a new Data with the oracle's output leaves (torch, CPU)."""
    top, con = {}, {}
    for n, a in out.items():
        t = torch.from_numpy(a)
        if t.is_floating_point():
            # recorded reference leaves can carry their own dtype (qfrc_actuator is float32 zeros when nu == 0,
            # forward.py:117-121): the ABI keeps every real leaf in the Data dtype
            t = t.to(d.qpos.dtype)
        path = native.DATA_PATH[n]
        if len(path) == 2:
            con[path[1]] = t
        else:
            top[n] = t
    res = d.replace(**top)
    return res.replace(contact=d.contact.replace(**con))
