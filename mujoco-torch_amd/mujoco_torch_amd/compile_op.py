"""``step`` / ``forward`` as a ``torch.library`` custom operator over the flat list of Data leaves.

Why: the reference's published batched mode is ``torch.compile(torch.vmap(lambda d: step(mx, d)), fullgraph=True)``
(``benchmarks/bench_compile.py:39-43``).  The native step is one opaque launch sequence: nothing in it can be traced, and under
``fullgraph=True`` nothing may graph-break either.  So when ``step`` is reached by Dynamo, or its Data holds functorch batched tensors
(plain ``torch.vmap``), it calls ``torch.ops.mujoco_torch_amd.step_leaves`` instead of the ctypes path:

* the operator takes the Data leaves in ABI order (``include/mjhip.h`` X-macro lists, then the input-only ``MJH_DATA_EXTRA_IN`` leaves; an absent leaf is an empty tensor), the Model as a
  process-unique number in a 0-dim int64 TENSOR (``Model._op_key_t``, resolved through a weak registry -- an operator cannot take a container; as a tensor it is an
  input of a traced graph, so ``mx.replace(body_mass=...)`` per episode steps the new values WITHOUT a recompile), the structure id ``tables.uid`` as a string (the
  graph's constant: shape propagation runs on any Model of that structure) and the call's flags; it returns the leaves the call writes, in ``forward._written_names`` order;
* its eager implementation rebuilds a ``Data`` around the tensors and runs the very same ``forward._run`` (the native batch);
* ``register_fake`` gives the output shapes (batch dims of ``qpos`` + the per-environment shape of each leaf), so Inductor / AOT see an
  ordinary opaque node;
* ``register_vmap`` maps a vmapped call onto ONE native batch: batched leaves move their mapped dimension to the front, unmapped ones are
  broadcast, outputs carry the mapped dimension first.  No private functorch call is involved.

The fast path of a direct ``step(mx, d)`` on plain tensors does not go through the operator (forward.py).
"""

from __future__ import annotations

import torch

from . import native
from .types import _MODELS_BY_KEY, structure_model

_NAMES = native.LISTS["MJH_DATA_REALS"] + native.LISTS["MJH_DATA_I32"] + native.LISTS["MJH_DATA_I64"]
_NREAL = len(native.LISTS["MJH_DATA_REALS"])
_INT_DTYPE = {n: torch.int32 for n in native.LISTS["MJH_DATA_I32"]} | {n: torch.int64 for n in native.LISTS["MJH_DATA_I64"]}
_QPOS = _NAMES.index("qpos")
# Input-only leaves that trail the ABI struct (cacc, cfrc_int, subtree_linvel, subtree_angmom: no stage writes them, sensors read them --
# include/mjhip.h MJH_DATA_EXTRA_IN).  They ride behind the ABI leaves in the operator's list so that a vmapped / traced call hands the
# CALLER's batched values to the native step (rebuilt from the unbatched template they failed the size check and were dropped).
_XNAMES = native.LISTS["MJH_DATA_EXTRA_IN"]
_TEMPLATES = {}  # tables uid -> an unbatched make_data(m): the fields outside the ABI and the per-environment shape of every leaf


def _model(key) -> "Model":
    """The Model whose VALUES this call steps: the operator receives its process-unique number as a 0-dim int64 tensor (an input of the traced graph, not a constant)."""
    m = _MODELS_BY_KEY.get(int(key))
    if m is None:
        raise RuntimeError("mujoco_torch_amd::step_leaves: the Model this call was made with no longer exists")
    return m


def _structure(uid: str):
    """Any live Model of that structure (tables.uid, as the string the graph carries: an int that changes between calls is made a symbolic
    shape by Dynamo's automatic dynamism, a string stays a guarded constant): what shape propagation needs does not depend on values."""
    m = structure_model(uid)
    if m is None:
        raise RuntimeError("mujoco_torch_amd::step_leaves: no Model of the structure this call was traced with exists any more")
    return m


def _template(m):
    from .io import make_data

    uid = m.tables.uid
    t = _TEMPLATES.get(uid)
    if t is None:
        if len(_TEMPLATES) > 64:
            _TEMPLATES.clear()
        from torch._subclasses.fake_tensor import unset_fake_temporarily

        with unset_fake_temporarily():  # first needed while an operator call is being shape-propagated: real tensors, outside the fake mode
            t = _TEMPLATES[uid] = make_data(m)
    return t


def out_names(m, do_step: bool, stages: int):
    """The leaves one call writes (what the operator returns, in this order)."""
    from .forward import _written_names  # (the package attribute `forward` is the function, not the module)

    names = _written_names(m, do_step)
    if not do_step and not (stages & 0x40):
        names = [n for n in names if n != "sensordata"]
    return names


@torch.library.custom_op("mujoco_torch_amd::step_leaves", mutates_args=())
def step_leaves(leaves: list[torch.Tensor], model_key: torch.Tensor, struct_uid: str, fixed_iterations: bool, do_step: bool, stages: int) -> list[torch.Tensor]:
    from .forward import _run_native

    m = _model(model_key)
    if m._struct_uid != struct_uid:
        raise RuntimeError("mujoco_torch_amd::step_leaves: the Model handed to the traced step has another structure than the one it was traced with "
                           "(different XML / cone / disabled constraints): trace a step of its own for it")
    tmpl = _template(m)
    top, con = {n: None for n in _XNAMES}, {}  # an absent input-only leaf is NULL (zeros) for the kernels -- never the template's unbatched copy
    for n, t in zip(_XNAMES, leaves[len(_NAMES):]):
        if t.numel() != 0:
            top[n] = t
    for n, t in zip(_NAMES, leaves):
        if t.numel() == 0 and t.dim() == 1 and native.data_field_tensor(tmpl, n) is not None and native.data_field_tensor(tmpl, n).numel() != 0:
            continue  # placeholder of a leaf the caller's Data did not carry: keep the template's
        path = native.DATA_PATH[n]
        (con if len(path) == 2 else top)[path[-1]] = t
    batch = tuple(leaves[_QPOS].shape[:-1])
    d = tmpl.replace(contact=tmpl.contact.replace(**con), **top)
    object.__setattr__(d, "_bs", batch)
    object.__setattr__(d.contact, "_bs", batch)
    # an operator's outputs may not share storage with each other: every written leaf gets an allocation of its own and the native call
    # fills them through its `out=` path (the direct call carves its leaves from two slabs instead)
    qpos = leaves[_QPOS]
    names = out_names(m, do_step, stages)
    outs, otop, ocon = [], {}, {}
    for n in names:
        ref = native.data_field_tensor(tmpl, n)
        t = torch.empty(batch + tuple(ref.shape), dtype=qpos.dtype if _NAMES.index(n) < _NREAL else _INT_DTYPE[n], device=qpos.device)
        outs.append(t)
        path = native.DATA_PATH[n]
        (ocon if len(path) == 2 else otop)[path[-1]] = t
    for n in _NAMES:  # the destination container carries the written leaves only (an expanded, stride-0 input leaf is not a valid destination)
        path = native.DATA_PATH[n]
        (ocon if len(path) == 2 else otop).setdefault(path[-1], None)
    dout = d.replace(contact=d.contact.replace(**ocon), **otop)
    _run_native(m, d, fixed_iterations, do_step, dout, stages)
    return outs


@step_leaves.register_fake
def _(leaves, model_key, struct_uid, fixed_iterations, do_step, stages):
    m = _structure(struct_uid)
    tmpl = _template(m)
    qpos = leaves[_QPOS]
    batch = tuple(qpos.shape[:-1])
    outs = []
    for n in out_names(m, do_step, stages):
        ref = native.data_field_tensor(tmpl, n)
        dt = qpos.dtype if _NAMES.index(n) < _NREAL else _INT_DTYPE[n]
        outs.append(qpos.new_empty(batch + tuple(ref.shape), dtype=dt))
    return outs


def _step_leaves_vmap(info, in_dims, leaves, model_key, struct_uid, fixed_iterations, do_step, stages):
    B = info.batch_size
    moved = []
    for t, bd in zip(leaves, in_dims[0]):
        moved.append(t.unsqueeze(0).expand(B, *t.shape) if bd is None else t.movedim(bd, 0))
    outs = step_leaves(moved, model_key, struct_uid, fixed_iterations, do_step, stages)
    return outs, [0] * len(outs)


torch.library.register_vmap(step_leaves, _step_leaves_vmap)

_ABSENT = torch.empty(0)


def run_through_op(m, d, fixed_iterations: bool, do_step: bool, stages: int):
    """``_run`` for calls that are being traced or vmapped: Dynamo-traceable Python around one operator call."""
    con = d.contact
    leaves = []
    for n in _NAMES:
        path = native.DATA_PATH[n]
        t = getattr(con if len(path) == 2 else d, path[-1], None)
        leaves.append(t if isinstance(t, torch.Tensor) else _ABSENT)
    for n in _XNAMES:
        t = getattr(d, n, None)
        leaves.append(t if isinstance(t, torch.Tensor) else _ABSENT)
    outs = torch.ops.mujoco_torch_amd.step_leaves(leaves, m._op_key_t, m._struct_uid, fixed_iterations, do_step, stages)
    top, cn = {}, {}
    names = out_names(m, do_step, stages)
    for i in range(len(names)):
        path = native.DATA_PATH[names[i]]
        if len(path) == 2:
            cn[path[-1]] = outs[i]
        else:
            top[path[-1]] = outs[i]
    if cn:
        top["contact"] = con.replace(**cn)
    return d.replace(**top)
