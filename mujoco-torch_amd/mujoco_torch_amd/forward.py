"""``step`` / ``forward``: the reference's public hot-path API on top of the native library.

Signatures and ownership rules follow reference ``_src/forward.py``: ``step(m, d,
fixed_iterations=False) -> Data`` (:463-496) and ``forward`` (:373-401).  The caller's ``Data``
is never mutated; every leaf the step writes is fresh storage, untouched leaves alias the
input (forward.py:473-475, dataclasses.py:112-120).  Unlike the reference, a *batched* ``Data``
(leading dims on every leaf, exactly what ``make_data(mx).expand(B).clone()`` produces) is
stepped natively in one launch sequence -- no ``torch.vmap`` -- and an un-batched ``Data`` is the
B = 1 case.

Host side of one call (what keeps ``d = step(mx, d)`` device-bound at B = 4096):

* the written leaves of a call are TWO allocations (``torch.empty`` of the summed, 256-byte aligned sizes: the bulk, and a small one
  for the state leaves callers keep across steps -- ``_SMALL_LEAVES`` -- so that a logged ``d.qpos`` does not pin ~50 KB per
  environment) -- fresh storage per call, as the reference's ``update_`` of new tensors -- carved by a plan cached per (model, batch shape, dtype, stages); the
  returned ``Data`` materialises a leaf's tensor view only when somebody reads it (``container._carve``);
* every ``Data`` carries a table of the raw device pointers of its leaves (``_PtrTab``) that follows it through ``replace`` /
  ``update_`` / attribute assignment: only leaves that changed are looked at (and validated) again, and the result of a step
  gets its table from the plan's offsets, so a rollout loop never walks the ~80 leaves in Python.

There is no fallback: tensors must live on a HIP device and ``libmjhip.so`` must be built.
"""

from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import native
from .container import UnbatchedTensor
from .types import Data, Model

# leaves the native step writes (reference Appendix: "Data leaves written by one step")
_WRITTEN_ALWAYS = (
    "qpos xpos xquat xmat xipos ximat xanchor xaxis geom_xpos geom_xmat site_xpos site_xmat cam_xpos cam_xmat "
    "light_xpos light_xdir subtree_com cdof cinert crb actuator_length actuator_moment qM qLD actuator_velocity "
    "cvel cdof_dot qfrc_bias qfrc_passive actuator_force qfrc_actuator qfrc_smooth qacc_smooth qacc act_dot"
).split()
_WRITTEN_CONTACT = (
    "contact_dist contact_pos contact_frame contact_includemargin contact_friction contact_solref "
    "contact_solreffriction contact_solimp contact_dim contact_geom1 contact_geom2 contact_geom contact_efc_address"
).split()
_WRITTEN_EFC = "efc_J efc_frictionloss efc_D efc_aref efc_force qacc_warmstart qfrc_constraint".split()
_WRITTEN_STEP = "qvel act time".split()

_ALL_NAMES = native.LISTS["MJH_DATA_REALS"] + native.LISTS["MJH_DATA_I32"] + native.LISTS["MJH_DATA_I64"]
_NLEAF = len(_ALL_NAMES)
_IDX = {n: i for i, n in enumerate(_ALL_NAMES)}
_NREAL = len(native.LISTS["MJH_DATA_REALS"])
_INT_DTYPE = {n: torch.int32 for n in native.LISTS["MJH_DATA_I32"]} | {n: torch.int64 for n in native.LISTS["MJH_DATA_I64"]}
_KEY = {n: native.DATA_PATH[n][-1] for n in _ALL_NAMES}          # field key inside its container
_IN_CONTACT = [len(native.DATA_PATH[n]) == 2 for n in _ALL_NAMES]
_CONTACT_IDX = frozenset(i for i in range(_NLEAF) if _IN_CONTACT[i])
_ALIGN = 256
# Leaves a rollout typically keeps per step (logging, replay buffers).  They get a small allocation of their own: every leaf of a
# step's output is a view of the allocation it was carved from, so a kept `d.qpos` pins that whole allocation -- a few KB per
# environment-row here, not the ~50 KB per environment of the full output (198 MB per kept humanoid step at B = 4096).
_SMALL_LEAVES = frozenset("qpos qvel act time qacc qacc_warmstart sensordata".split())


def _written_names(m: Model, step: bool):
    ne, nf, nl, ncon, nefc = m.constraint_sizes_py
    names = list(_WRITTEN_ALWAYS)
    if ncon > 0:
        names += _WRITTEN_CONTACT
    if nefc > 0:
        names += _WRITTEN_EFC
    if len(m.tables.sensors["type"]) > 0:
        names += ["sensordata"]
    if m.has_gravcomp:
        names += ["qfrc_gravcomp"]
    if int(m.ntendon) > 0:
        names += ["ten_length", "ten_J", "ten_velocity"]
    if step:
        names += _WRITTEN_STEP
    return names


class _PtrTab:
    """Raw device pointers of one ``Data``'s leaves in ABI order (the ``mjhData`` struct the library takes), kept current
    incrementally: ``dirty`` holds the leaves whose tensor may have changed since the pointer was read."""

    __slots__ = ("struct", "arr", "xarr", "dirty", "keep", "con", "con_ver", "sig")

    def __init__(self):
        self.struct = native.DataPtrs()
        full = np.frombuffer(self.struct, dtype=np.uint64)  # shares the struct's memory
        self.arr = full[:_NLEAF]
        self.xarr = full[_NLEAF:]  # the trailing input-only leaves (MJH_DATA_EXTRA_IN): filled per call when a sensor reads them (_extra_inputs)
        self.dirty = set(range(_NLEAF))
        self.keep = {}          # contiguous copies of strided input leaves (re-made every call: the source may be written in place)
        self.con = None
        self.con_ver = -1
        self.sig = None

    def child(self, names):
        """The table of a container derived by ``replace(**names)``."""
        t = _PtrTab()
        t.arr[:] = self.arr
        t.dirty = set(self.dirty)
        t.keep = dict(self.keep)
        t.con, t.con_ver, t.sig = self.con, self.con_ver, self.sig
        t.mark(names)
        return t

    def mark(self, names):
        if names is None:       # an update of unknown extent
            self.dirty = set(range(_NLEAF))
            return
        for n in names:
            if n == "contact":
                self.dirty |= _CONTACT_IDX
                self.con = None
            else:
                i = _IDX.get(n)
                if i is not None:
                    self.dirty.add(i)


def _table(d: Data, sig, counts, B: int, dtype, device, dest: bool = False) -> _PtrTab:
    """``d``'s pointer table, refreshed and validated for this (model, dtype, device, batch).

    Every leaf handed to the kernels is checked ONCE per tensor (device, dtype, element count == B * per-environment count --
    the kernels index ``ptr + env * count`` unchecked) when its pointer is read.  A strided input leaf is copied to contiguous
    storage every call; a strided leaf of a destination container (``out=``, ``reset_where``) is an error: the kernels would
    fill a temporary and the caller's tensor would silently keep its old contents."""
    tab = d.__dict__.get("_ptab")
    if tab is None:
        tab = _PtrTab()
        object.__setattr__(d, "_ptab", tab)
    if tab.sig != sig:
        tab.dirty = set(range(_NLEAF))
        tab.keep = {}
        tab.sig = sig
    con = d._fields["contact"]
    cver = con.__dict__.get("_ver", 0)
    if tab.con is not con or tab.con_ver != cver:
        tab.dirty |= _CONTACT_IDX
        tab.con, tab.con_ver = con, cver
    if not tab.dirty:
        return tab
    arr, again = tab.arr, set()
    for i in tab.dirty:
        obj = con if _IN_CONTACT[i] else d
        key = _KEY[_ALL_NAMES[i]]
        lz = obj.__dict__.get("_lazy")
        if lz:
            spec = lz.get(key)
            if spec is not None:  # a leaf of a previous step's slab that nobody has touched: sized by the plan that made it
                arr[i] = spec[0].data_ptr() + spec[1]
                continue
        t = obj._fields.get(key)
        if t is None or not isinstance(t, torch.Tensor):
            arr[i] = 0
            continue
        n = t.numel()
        if n == 0:
            arr[i] = 0
            continue
        name = _ALL_NAMES[i]
        if t.device != device:
            raise RuntimeError(f"Data.{name} is on {t.device}, expected {device}")
        want = dtype if i < _NREAL else _INT_DTYPE[name]
        if t.dtype != want:
            raise RuntimeError(f"Data.{name} has dtype {t.dtype}, expected {want}" + (" (mixed-precision Data is not supported)" if i < _NREAL else ""))
        if n != B * int(counts[i]):
            raise ValueError(f"Data.{name} holds {n} elements (shape {tuple(t.shape)}); a batch of {B} environments of this model needs "
                             f"{B} x {int(counts[i])}.  Every leaf of a batched Data must carry the batch dimensions (make_data(mx).expand(B).clone()).")
        if not t.is_contiguous():
            if dest:
                raise ValueError(f"Data.{name} of a destination container is not contiguous (shape {tuple(t.shape)}, strides {t.stride()}): "
                                 "the kernels write batch-major contiguous leaves; pass cloned / contiguous storage.")
            t = t.contiguous()
            tab.keep[i] = t
            again.add(i)
        arr[i] = t.data_ptr()
    tab.dirty = again
    return tab


_EXTRA_NAMES = native.LISTS["MJH_DATA_EXTRA_IN"]
_EXTRA_WIDTH = {"cacc": 6, "cfrc_int": 6, "subtree_linvel": 3, "subtree_angmom": 3}


def _extra_inputs(tab: _PtrTab, d: Data, names, B: int, nbody: int, dtype, device):
    """Pointers of the trailing input-only leaves (read per call: four tensors at most).  An absent leaf stays NULL = zeros."""
    keep = []
    for n in names:
        slot = _EXTRA_NAMES.index(n)
        t = d._fields.get(n)
        if not isinstance(t, torch.Tensor) or t.numel() == 0:
            tab.xarr[slot] = 0
            continue
        if t.device != device or t.dtype != dtype:
            raise RuntimeError(f"Data.{n} is {t.dtype} on {t.device}, expected {dtype} on {device}")
        if t.numel() != B * nbody * _EXTRA_WIDTH[n]:
            raise ValueError(f"Data.{n} holds {t.numel()} elements (shape {tuple(t.shape)}); a batch of {B} environments of this model needs {B} x {nbody * _EXTRA_WIDTH[n]}")
        if not t.is_contiguous():
            t = t.contiguous()
            keep.append(t)
        tab.xarr[slot] = t.data_ptr()
    return keep


class _Plan:
    """Where the leaves one call writes live inside its output slab (cached per model / batch shape / dtype / stages)."""

    __slots__ = ("names", "widx", "wmask", "off", "total", "top", "con", "top_names", "con_names", "empties", "total_small", "widx_small", "widx_bulk")


_PLANS = {}


def _peek(obj, key):
    """(shape, dtype) of a leaf without materialising a lazily carved one."""
    lz = obj.__dict__.get("_lazy")
    if lz and key in lz:
        return tuple(lz[key][4]), lz[key][3]
    t = obj._fields[key]
    return tuple(t.shape), t.dtype


def _plan(m: Model, d: Data, names, batch, dtype, device, plan_key) -> _Plan:
    p = _PLANS.get(plan_key)
    if p is not None:
        return p
    p = _Plan()
    p.names = list(names)
    p.wmask = np.zeros(_NLEAF, dtype=bool)
    p.off = np.zeros(_NLEAF, dtype=np.uint64)
    p.top, p.con, p.empties = [], [], {}
    off, off_small = 0, 0
    small = np.zeros(_NLEAF, dtype=bool)
    con = d._fields["contact"]
    for n in names:
        i = _IDX[n]
        obj = con if _IN_CONTACT[i] else d
        shape, dt = _peek(obj, _KEY[n])
        want = dtype if i < _NREAL else _INT_DTYPE[n]
        nel = int(math.prod(shape))
        if nel == 0:
            p.empties[n] = torch.empty(shape, dtype=want, device=device)
            continue
        nbytes = nel * torch.empty((), dtype=want).element_size()
        is_small = n in _SMALL_LEAVES
        here = off_small if is_small else off
        # spec: (which allocation: 0 bulk / 1 small, byte offset, byte length, dtype, shape)
        (p.con if _IN_CONTACT[i] else p.top).append((_KEY[n], here, nbytes, want, shape, 1 if is_small else 0))
        p.wmask[i] = True
        p.off[i] = here
        small[i] = is_small
        step_bytes = (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        if is_small:
            off_small += step_bytes
        else:
            off += step_bytes
    p.total = off
    p.total_small = off_small
    p.widx = np.nonzero(p.wmask)[0]
    p.widx_small = np.nonzero(p.wmask & small)[0]
    p.widx_bulk = np.nonzero(p.wmask & ~small)[0]
    p.top_names = [s[0] for s in p.top] + [_KEY[n] for n in p.empties if not _IN_CONTACT[_IDX[n]]]
    p.con_names = [s[0] for s in p.con] + [_KEY[n] for n in p.empties if _IN_CONTACT[_IDX[n]]]
    if len(_PLANS) > 256:
        _PLANS.clear()
    _PLANS[plan_key] = p
    return p


_COUNT_CACHE = {}


def _counts(m: Model, device):
    """ncon / nefc as the unbatched device scalars the reference's Data carries (types.py:1172-1178); constant per model."""
    ne, nf, nl, ncon, nefc = m.constraint_sizes_py
    key = (ncon, nefc, device)
    hit = _COUNT_CACHE.get(key)
    if hit is None:
        hit = (UnbatchedTensor(torch.full((), ncon, dtype=torch.int32, device=device)),
               UnbatchedTensor(torch.full((), nefc, dtype=torch.int32, device=device)))
        _COUNT_CACHE[key] = hit
    return hit


def _require_device(device):
    """There is no CPU / PyTorch path: anything but a HIP device is an error (tests swap this check out to drive the host
    logic against the CPU oracle)."""
    if device.type != "cuda":
        raise RuntimeError(
            "mujoco_torch_amd.step/forward run only on a HIP device (tensors on "
            f"{device}); there is no CPU or PyTorch fallback. Move Model/Data with .to('cuda')."
        )


def _stream_and_guard(device):
    """(raw stream handle of the caller's current stream, device to restore or None)."""
    if device.type != "cuda":
        return 0, None
    stream = torch.cuda.current_stream(device).cuda_stream
    prev = torch.cuda.current_device()
    if prev != device.index:
        torch.cuda.set_device(device)
        return stream, prev
    return stream, None


def _plain(t) -> bool:
    """False for tensors without storage of their own: functorch batched tensors (``torch.vmap``), fake / functional tensors."""
    try:
        t.data_ptr()
        return True
    except RuntimeError:
        return False


def _run(m: Model, d: Data, fixed_iterations: bool, step: bool, out: Data | None = None, stages: int = native.STAGE_ALL) -> Data:
    """Traced by Dynamo (``torch.compile``, also ``fullgraph=True``) or called under ``torch.vmap`` -- the reference's batching idiom
    (README, benchmarks/_helpers.py:44-60, bench_compile.py:39-43): the call goes through ``torch.ops.mujoco_torch_amd.step_leaves``
    (compile_op.py), whose vmap rule turns the mapped call into ONE native batch.  Plain tensors take the direct path below."""
    if torch.compiler.is_compiling() or not _plain(d.qpos):
        if out is not None:
            raise ValueError("step(..., out=) cannot be used under torch.compile / torch.vmap")
        from . import compile_op

        return compile_op.run_through_op(m, d, fixed_iterations, step, stages)
    return _run_native(m, d, fixed_iterations, step, out, stages)


def _run_native(m: Model, d: Data, fixed_iterations: bool, step: bool, out: Data | None = None, stages: int = native.STAGE_ALL) -> Data:
    qpos = d.qpos
    _require_device(qpos.device)
    dtype = qpos.dtype
    if dtype not in (torch.float64, torch.float32):
        raise RuntimeError(f"unsupported Data dtype {dtype}")
    device = qpos.device
    batch = tuple(qpos.shape[:-1])
    B = int(math.prod(batch)) if batch else 1
    nm = native.get_native_model(m, device, dtype)
    T = m.tables
    sig = (T.uid, dtype, device, B)
    tab = _table(d, sig, nm.leaf_counts, B, dtype, device)
    plan_key = (T.uid, step, stages, batch, dtype, device)
    plan = _PLANS.get(plan_key)
    if plan is None:
        names = _written_names(m, step)
        if not step and not (stages & 0x40):
            names = [n for n in names if n != "sensordata"]  # sensors belong to complete forward passes
        plan = _plan(m, d, names, batch, dtype, device, plan_key)
    out_struct = native.DataPtrs()
    out_arr = np.frombuffer(out_struct, dtype=np.uint64)[:_NLEAF]
    if out is None:
        slab = torch.empty(plan.total, dtype=torch.uint8, device=device)
        slabs = (slab, torch.empty(plan.total_small, dtype=torch.uint8, device=device))  # the commonly kept state leaves live apart (_SMALL_LEAVES)
        out_arr[:] = plan.off
        out_arr[plan.widx_bulk] += np.uint64(slab.data_ptr())
        out_arr[plan.widx_small] += np.uint64(slabs[1].data_ptr())
    else:
        if out is d:
            raise ValueError("step(..., out=d) with out being the input itself is not supported: use a second buffer (ping-pong)")
        otab = _table(out, sig, nm.leaf_counts, B, dtype, device, dest=True)
        np.multiply(otab.arr, plan.wmask, out=out_arr, casting="unsafe")  # only the leaves this call writes are handed over
        missing = plan.wmask & (out_arr == 0)
        if missing.any():
            raise ValueError(f"out= lacks storage for written leaves: {[_ALL_NAMES[i] for i in np.nonzero(missing)[0]][:6]}")
        if (out_arr[plan.widx] == tab.arr[plan.widx]).any():
            raise ValueError("out= shares storage with the input on leaves the step writes: the phases read the caller's state "
                             "after the first outputs are written (and RK4 reads it in every stage); use distinct buffers")
    flags = native.FLAG_FIXED_ITERATIONS if fixed_iterations else 0
    extra = T.sensors["extra_leaves"]
    if extra:  # Data leaves no stage writes but a sensor of this model reads (cacc, cfrc_int, subtree_linvel / angmom: include/mjhip.h MJH_DATA_EXTRA_IN)
        keep = _extra_inputs(tab, d, extra, B, int(m.nbody), dtype, device)
    stream, prev = _stream_and_guard(device)
    try:
        work = nm.workspace(B, stream)  # RK4 stages; candidate contacts of max_contact_points over box / mesh pairs; None for most Euler models
        if step:
            rc = nm.lib.mjh_step(nm.handle, ctypes.byref(tab.struct), ctypes.byref(out_struct),
                                 ctypes.c_void_p(work.data_ptr() if work is not None else None), B, flags, ctypes.c_void_p(stream))
        else:
            rc = nm.lib.mjh_forward(nm.handle, ctypes.byref(tab.struct), ctypes.byref(out_struct),
                                    ctypes.c_void_p(work.data_ptr() if work is not None else None), B, stages, flags, ctypes.c_void_p(stream))
    finally:
        if prev is not None:
            torch.cuda.set_device(prev)
    if rc != 0:
        raise RuntimeError(f"native step failed ({rc}): {nm.lib.mjh_last_error().decode()}")
    if out is not None:
        return out
    # ---- the returned Data: the caller's container with the written leaves swapped for (lazy) views of the slab ----
    res = d.clone(recurse=False)
    f = res._fields
    if "_order" not in res.__dict__:
        object.__setattr__(res, "_order", tuple(f))  # leaves re-enter the dict when they are carved: keep the caller's field order
    for k in plan.top_names:
        f.pop(k, None)
    lz = res.__dict__.get("_lazy")
    if lz is None:
        lz = {}
        object.__setattr__(res, "_lazy", lz)
    for sp in plan.top:
        lz[sp[0]] = (slabs[sp[5]], sp[1], sp[2], sp[3], sp[4])
    con = f["contact"]
    if plan.con or any(_IN_CONTACT[_IDX[n]] for n in plan.empties):
        con = con.clone(recurse=False)
        cf = con._fields
        if "_order" not in con.__dict__:
            object.__setattr__(con, "_order", tuple(cf))
        for k in plan.con_names:
            cf.pop(k, None)
        clz = con.__dict__.get("_lazy")
        if clz is None:
            clz = {}
            object.__setattr__(con, "_lazy", clz)
        for sp in plan.con:
            clz[sp[0]] = (slabs[sp[5]], sp[1], sp[2], sp[3], sp[4])
        f["contact"] = con
    for n, t in plan.empties.items():
        (con._fields if _IN_CONTACT[_IDX[n]] else f)[_KEY[n]] = t
    f["ncon"], f["nefc"] = _counts(m, device)
    rt = _PtrTab.__new__(_PtrTab)
    rt.struct = native.DataPtrs()
    full = np.frombuffer(rt.struct, dtype=np.uint64)
    rt.arr, rt.xarr = full[:_NLEAF], full[_NLEAF:]
    np.copyto(rt.arr, np.where(plan.wmask, out_arr, tab.arr))
    rt.dirty = {i for i in tab.dirty if not plan.wmask[i]}
    rt.keep = {i: t for i, t in tab.keep.items() if not plan.wmask[i]}
    rt.con, rt.con_ver, rt.sig = con, con.__dict__.get("_ver", 0), sig
    object.__setattr__(res, "_ptab", rt)
    return res


def step(m: Model, d: Data, fixed_iterations: bool = False, *, out: Data | None = None) -> Data:
    """Advance simulation by one timestep (reference forward.py:463-496).

    ``out`` (extension): an existing ``Data`` of the same shape whose storage receives the written leaves instead of fresh
    storage (ping-pong buffers).  Leaves the step does not write (``ctrl``, ``qfrc_applied``, ``xfrc_applied``, ``mocap_*``,
    ``eq_active`` ...) are NOT carried from ``d`` into ``out``: ``out`` keeps its own.  ``out`` must not share storage with ``d``.
    """
    return _run(m, d, fixed_iterations, step=True, out=out)


def forward(m: Model, d: Data, fixed_iterations: bool = False, *, stages: int = native.STAGE_ALL) -> Data:
    """Forward dynamics (reference forward.py:373-401)."""
    return _run(m, d, fixed_iterations, step=False, stages=stages)


def reset_where(m: Model, d: Data, d0: Data, mask: torch.Tensor, qpos: torch.Tensor | None = None,
                qvel: torch.Tensor | None = None) -> Data:
    """In-place masked reset of a batched ``d``: ``d[mask] = d0`` with ``qpos`` / ``qvel`` rows taken from the given tensors.

    The env caller's ``self._dx[mask] = self._make_batch(n)`` (reference zoo/base.py:266-273, :289-293, :327-331) as one
    native launch over every leaf and without the host sync of a boolean-mask gather.  ``d0`` holds ONE environment;
    ``mask``: bool ``[B]``; ``qpos`` ``[B, nq]`` / ``qvel`` ``[B, nv]`` (Data dtype) are the per-environment reset states
    (``dx0 + noise``; rows of unmasked environments are ignored; None = ``d0``'s).  Returns ``d``.
    """
    dq = d.qpos
    _require_device(dq.device)
    device, dtype = dq.device, dq.dtype
    if dq.dim() != 2:
        raise ValueError("reset_where expects a Data with exactly one batch dimension")
    B = dq.shape[0]
    if d0.qpos.dtype != dtype or d0.qpos.device != device:
        raise RuntimeError(f"d0 is {d0.qpos.dtype} on {d0.qpos.device}, d is {dtype} on {device}")
    if d0.qpos.numel() != dq.shape[-1]:
        raise ValueError("d0 must hold exactly one environment")
    if mask.dtype != torch.bool or tuple(mask.shape) != (B,) or mask.device != device:
        raise ValueError(f"mask must be a bool tensor of shape ({B},) on {device}")
    mask = mask.contiguous()
    rows = []
    for name, z, n in (("qpos", qpos, dq.shape[-1]), ("qvel", qvel, d.qvel.shape[-1])):
        if z is not None:
            if z.dtype != dtype or tuple(z.shape) != (B, n) or z.device != device:
                raise ValueError(f"{name} must be {dtype} of shape ({B}, {n}) on {device}")
            z = z.contiguous()
        rows.append(z)
    nm = native.get_native_model(m, device, dtype)
    T = m.tables
    dtab = _table(d, (T.uid, dtype, device, B), nm.leaf_counts, B, dtype, device, dest=True)
    stab = _table(d0, (T.uid, dtype, device, 1), nm.leaf_counts, 1, dtype, device)
    stream, prev = _stream_and_guard(device)
    try:
        rc = nm.lib.mjh_reset_where(nm.handle, ctypes.byref(dtab.struct), ctypes.byref(stab.struct), ctypes.c_void_p(mask.data_ptr()),
                                    ctypes.c_void_p(rows[0].data_ptr() if rows[0] is not None else None),
                                    ctypes.c_void_p(rows[1].data_ptr() if rows[1] is not None else None), B, ctypes.c_void_p(stream))
    finally:
        if prev is not None:
            torch.cuda.set_device(prev)
    if rc != 0:
        raise RuntimeError(f"native reset failed ({rc}): {nm.lib.mjh_last_error().decode()}")
    return d
