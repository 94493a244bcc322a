"""``step`` / ``forward``: the reference's public hot-path API on top of the native library.

Signatures and ownership rules follow reference ``_src/forward.py``: ``step(m, d,
fixed_iterations=False) -> Data`` (:463-496) and ``forward`` (:373-401).  The caller's ``Data``
is never mutated; every leaf the step writes is a fresh tensor, untouched leaves alias the
input (forward.py:473-475, dataclasses.py:112-120).  Unlike the reference, a *batched* ``Data``
(leading dims on every leaf, exactly what ``make_data(mx).expand(B).clone()`` produces) is
stepped natively in one launch sequence -- no ``torch.vmap`` -- and an un-batched ``Data`` is the
B = 1 case.

There is no fallback: tensors must live on a HIP device and ``libmjhip.so`` must be built.
"""

from __future__ import annotations

import ctypes
import math

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
_REAL_NAMES = set(native.LISTS["MJH_DATA_REALS"])


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


def _fill_ptrs(d: Data, names, dtype, device, check=True):
    ptrs = native.DataPtrs()
    keep = []
    for n in names:
        t = native.data_field_tensor(d, n)
        if t is None:
            continue
        if check and t.numel():
            if t.device != device:
                raise RuntimeError(f"Data.{n} is on {t.device}, expected {device}")
            if n in _REAL_NAMES and t.dtype != dtype:
                raise RuntimeError(f"Data.{n} has dtype {t.dtype}, expected {dtype} (mixed-precision Data is not supported)")
            if not t.is_contiguous():
                t = t.contiguous()
        keep.append(t)
        setattr(ptrs, n, t.data_ptr() if t.numel() else None)
    return ptrs, keep


import os as _os

_PTR_CACHE = _os.environ.get("MJH_NO_PTR_CACHE") != "1"  # diagnostic switch (tools/host_overhead.py)


def _ptrs_cached(d: Data, tag, names, dtype, device, check=True):
    """``_fill_ptrs`` memoised on the container: the ~80 attribute walks and ``data_ptr()`` calls are most of the host cost
    of a step at small batches.  Valid while the leaf SET is unchanged (``replace`` makes a new container, ``update_`` /
    attribute assignment bump the version); in-place writes into the tensors keep their pointers."""
    con = d.contact
    key = (d.__dict__.get("_ver", 0), id(con), con.__dict__.get("_ver", 0), dtype, device)
    cache = d.__dict__.get("_ptr_cache") if _PTR_CACHE else None
    if cache is not None:
        hit = cache.get(tag)  # one slot per role: a ping-pong buffer is the input of one call and the output of the next
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
    ptrs, keep = _fill_ptrs(d, names, dtype, device, check)
    present = [n for n in names if native.data_field_tensor(d, n) is not None]
    if _PTR_CACHE and all(k is native.data_field_tensor(d, n) for k, n in zip(keep, present)):  # only when no contiguous copy had to be made
        if cache is None:
            cache = {}
            object.__setattr__(d, "_ptr_cache", cache)
        cache[tag] = (key, ptrs, keep)
    return ptrs, keep


def _under_vmap(m, d, fixed_iterations, step, stages):
    """``torch.vmap(lambda d: step(mx, d))(dx)`` -- the reference's batching idiom (README, benchmarks/_helpers.py:44-60).

    Inside vmap every mapped leaf is a functorch BatchedTensor.  The native step is batched already, so the leaves are
    unwrapped (mapped dimension moved to the front, unmapped leaves broadcast), stepped as ONE native batch and the result is
    wrapped back at the same vmap level: the idiom costs one launch sequence, not a per-sample loop."""
    F = torch._C._functorch
    level = F.maybe_get_level(d.qpos)
    raw = F.get_unwrapped(d.qpos)
    if F.is_batchedtensor(raw):
        raise NotImplementedError("nested torch.vmap over step is not supported: pass a Data with two leading batch dims instead")
    B = raw.shape[F.maybe_get_bdim(d.qpos)]

    def unwrap(t):
        if F.is_batchedtensor(t) and F.maybe_get_level(t) == level:
            return F.get_unwrapped(t).movedim(F.maybe_get_bdim(t), 0)
        return t.unsqueeze(0).expand(B, *t.shape)

    plain = d.map_tensors(unwrap)
    object.__setattr__(plain, "_bs", (B, *plain._bs))
    if isinstance(plain.contact, type(d.contact)):
        object.__setattr__(plain.contact, "_bs", (B, *plain.contact._bs))
    res = _run(m, plain, fixed_iterations, step, None, stages)
    wrapped = res.map_tensors(lambda t: F._add_batch_dim(t, 0, level))
    object.__setattr__(wrapped, "_bs", tuple(d._bs))
    object.__setattr__(wrapped.contact, "_bs", tuple(d.contact._bs))
    return wrapped


def _run(m: Model, d: Data, fixed_iterations: bool, step: bool, out: Data | None = None, stages: int = native.STAGE_ALL) -> Data:
    qpos = d.qpos
    if torch._C._functorch.is_batchedtensor(qpos):
        if out is not None:
            raise ValueError("step(..., out=) cannot be used under torch.vmap")
        return _under_vmap(m, d, fixed_iterations, step, stages)
    if qpos.device.type != "cuda":
        raise RuntimeError(
            "mujoco_torch_amd.step/forward run only on a HIP device (tensors on "
            f"{qpos.device}); there is no CPU or PyTorch fallback. Move Model/Data with .to('cuda')."
        )
    dtype = qpos.dtype
    if dtype not in (torch.float64, torch.float32):
        raise RuntimeError(f"unsupported Data dtype {dtype}")
    device = qpos.device
    batch = tuple(qpos.shape[:-1])
    B = int(math.prod(batch)) if batch else 1
    nm = native.get_native_model(m, device, dtype)
    names = _written_names(m, step)
    if not step and not (stages & 0x40):
        names = [n for n in names if n != "sensordata"]  # sensors belong to complete forward passes
    in_ptrs, keep_in = _ptrs_cached(d, ("in", id(m.tables)), _ALL_NAMES, dtype, device)
    if out is None:
        new = {}
        for n in names:
            src = native.data_field_tensor(d, n)
            new[n] = torch.empty_like(src, memory_format=torch.contiguous_format)
        contact_kw = {n[len("contact_"):] if n != "contact_dim" else "contact_dim": t for n, t in new.items() if n.startswith("contact_")}
        top_kw = {n: t for n, t in new.items() if not n.startswith("contact_")}
        res = d.replace(**top_kw)
        if contact_kw:
            res = res.replace(contact=d.contact.replace(**contact_kw))
    else:
        res = out
    out_ptrs, keep_out = _ptrs_cached(res, ("out", id(m.tables), step, stages), names, dtype, device, check=True) if out is not None else _fill_ptrs(res, names, dtype, device, check=False)
    stream = torch.cuda.current_stream(device).cuda_stream
    flags = native.FLAG_FIXED_ITERATIONS if fixed_iterations else 0
    with torch.cuda.device(device):
        if step:
            work = nm.workspace(B)
            rc = nm.lib.mjh_step(nm.handle, ctypes.byref(in_ptrs), ctypes.byref(out_ptrs),
                                 ctypes.c_void_p(work.data_ptr() if work is not None else None), B, flags, ctypes.c_void_p(stream))
        else:
            rc = nm.lib.mjh_forward(nm.handle, ctypes.byref(in_ptrs), ctypes.byref(out_ptrs), B, stages, flags, ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"native step failed ({rc}): {nm.lib.mjh_last_error().decode()}")
    ne, nf, nl, ncon, nefc = m.constraint_sizes_py
    res.update_(
        ncon=UnbatchedTensor(torch.full((), ncon, dtype=torch.int32, device=device)),
        nefc=UnbatchedTensor(torch.full((), nefc, dtype=torch.int32, device=device)),
    ) if out is None else None
    return res


def step(m: Model, d: Data, fixed_iterations: bool = False, *, out: Data | None = None) -> Data:
    """Advance simulation by one timestep (reference forward.py:463-496).

    ``out`` (extension): an existing ``Data`` of the same shape whose storage receives the
    result instead of freshly allocated tensors (ping-pong buffers for tight loops).
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
    if dq.device.type != "cuda":
        raise RuntimeError(f"mujoco_torch_amd.reset_where runs only on a HIP device (tensors on {dq.device}); there is no CPU fallback")
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
    d_ptrs, keep_d = _ptrs_cached(d, ("reset", id(m.tables)), _ALL_NAMES, dtype, device)
    s_ptrs, keep_s = _ptrs_cached(d0, ("reset0", id(m.tables)), _ALL_NAMES, dtype, device)
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        rc = nm.lib.mjh_reset_where(nm.handle, ctypes.byref(d_ptrs), ctypes.byref(s_ptrs), ctypes.c_void_p(mask.data_ptr()),
                                    ctypes.c_void_p(rows[0].data_ptr() if rows[0] is not None else None),
                                    ctypes.c_void_p(rows[1].data_ptr() if rows[1] is not None else None), B, ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"native reset failed ({rc}): {nm.lib.mjh_last_error().decode()}")
    return d
