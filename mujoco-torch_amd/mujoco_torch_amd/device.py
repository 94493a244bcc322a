"""``device_put``: compiled model -> ``Model`` (+ the static tables the native kernels consume).

Mirrors reference ``_src/device.py``: ``device_put`` :1029-1112, ``_validate`` :919-964,
``_model_derived`` :632-884, ``_compute_constraint_sizes`` :226-264,
``_compute_constraint_data`` :267-378 and ``_compute_actuator_static_moment`` :588-629.
The source model is either this package's MJCF-subset compiler output (``mjcf.MjModelLite``)
or a real ``mujoco.MjModel`` (duck-typed on MuJoCo's field names).

What is NOT mirrored: the reference's vmap-grouping caches (``scan.precompute_scan_caches``) and
Dynamo stride workarounds -- the native kernels walk the kinematic tree per environment and
need neither.  Instead ``Model.tables`` carries the flat integer / real tables declared in
``include/mjhip.h`` (``mjhModelDesc``).
"""

from __future__ import annotations

import itertools

import collections
import weakref

import numpy as np
import torch

from . import collision_tables as ct
from . import convex
from ._enums import (
    BiasType,
    ConeType,
    DisableBit,
    DynType,
    GainType,
    GeomType,
    IntegratorType,
    JacobianType,
    JointType,
    SensorType,
    SolverType,
    TrnType,
    SUPPORTED_CONDIM,
)
from .container import UnbatchedTensor
from .types import MODEL_FIELDS, Model, Option, Statistic

_cache_id_counter = itertools.count(1)


def _get(value, name, default=None):
    try:
        return getattr(value, name)
    except AttributeError:
        if default is None:
            raise
        return default() if callable(default) else default


def _validate(m):
    """Feature gate (reference device.py:919-964): unsupported options raise NotImplementedError."""
    opt = m.opt
    if int(opt.integrator) not in (int(IntegratorType.EULER), int(IntegratorType.RK4)):
        raise NotImplementedError(f"integrator {IntegratorType(int(opt.integrator)).name} not implemented.")
    if int(opt.solver) not in (int(SolverType.CG), int(SolverType.NEWTON)):
        raise NotImplementedError(f"solver {SolverType(int(opt.solver)).name} not implemented.")
    if int(opt.cone) not in (int(ConeType.PYRAMIDAL), int(ConeType.ELLIPTIC)):
        raise NotImplementedError("unknown cone type")
    jac = int(opt.jacobian)
    if jac == int(JacobianType.SPARSE) or (jac == int(JacobianType.AUTO) and int(m.nv) >= 60):
        raise NotImplementedError(
            "sparse inertia (jacobian=sparse, or auto with nv >= 60) is not carried: the reference's own sparse factor_m loses "
            "eliminations (scatter with repeated indices, smooth.py:325-326) and its solve_m is 13-27 % off the dense solve on the "
            "bundled models (oracle/probe_reference_sparse.py, profiles/r02/reference_sparse_probe.txt), so there is no result to be "
            "identical to.  Set <option jacobian=\"dense\"/> (opt.jacobian = DENSE): the dense path serves models up to 256 dofs.")
    if any(int(d) not in SUPPORTED_CONDIM for d in np.asarray(m.geom_condim)) or any(
        int(d) not in SUPPORTED_CONDIM for d in np.asarray(m.pair_dim)
    ):
        raise NotImplementedError("Only condim=1, 3, 4 and 6 are supported.")
    for name, enum, ok in (
        ("actuator_trntype", TrnType, (TrnType.JOINT, TrnType.JOINTINPARENT, TrnType.TENDON)),
        ("actuator_dyntype", DynType, (DynType.NONE, DynType.INTEGRATOR, DynType.FILTER, DynType.FILTEREXACT, DynType.MUSCLE)),
        ("actuator_gaintype", GainType, (GainType.FIXED, GainType.AFFINE, GainType.MUSCLE)),
        ("actuator_biastype", BiasType, (BiasType.NONE, BiasType.AFFINE, BiasType.MUSCLE)),
    ):
        for v in np.asarray(getattr(m, name)).reshape(-1):
            if int(v) not in [int(x) for x in ok]:
                raise NotImplementedError(f"{enum(int(v)).name} {name} not implemented.")
    if int(_get(m, "ntendon", 0)):
        # Spatial tendons (site / geom / pulley wraps) are carried the way the reference carries them: smooth.tendon (:470-497) evaluates the
        # tendons whose FIRST wrap is a joint wrap (device.py:850) and leaves length 0 and a zero Jacobian row for every other one.  A tendon
        # that mixes joint and other wraps is not a valid MuJoCo tendon (and breaks the reference's segment arithmetic, device.py:853-870).
        wt = np.asarray(_get(m, "wrap_type", lambda: np.ones(0)))
        for t in range(int(m.ntendon)):
            a, n = int(m.tendon_adr[t]), int(m.tendon_num[t])
            if n and len({int(x) == 1 for x in wt[a : a + n]}) > 1:
                raise NotImplementedError("a tendon mixing joint wraps with site / geom / pulley wraps is not supported")
    et = np.asarray(_get(m, "eq_type", lambda: np.zeros(0, dtype=np.int32)))
    if np.any(et > 2):
        raise NotImplementedError("only connect / weld / joint equality constraints are supported")
    ot = np.asarray(_get(m, "eq_objtype", lambda: np.ones(len(et), dtype=np.int32)))
    if np.any((ot == 6) & np.asarray(_get(m, "eq_active0", lambda: np.zeros(len(et), dtype=bool))).astype(bool)):
        # the reference ignores eq_objtype and reads site ids as body ids (constraint.py:116-212): harmless while the
        # constraint is inactive (every row is multiplied by eq_active), meaningless once it is active
        raise NotImplementedError("site-based connect / weld constraints can only be carried inactive")


def _t(x, dtype):
    return torch.tensor(np.asarray(x, dtype=np.float64), dtype=dtype)


def _option(opt, dtype) -> Option:
    return Option(
        iterations=int(opt.iterations),
        ls_iterations=int(opt.ls_iterations),
        tolerance=float(opt.tolerance),
        ls_tolerance=float(opt.ls_tolerance),
        impratio=_t(opt.impratio, dtype),
        gravity=_t(opt.gravity, dtype),
        density=_t(opt.density, dtype),
        viscosity=_t(opt.viscosity, dtype),
        magnetic=_t(opt.magnetic, dtype),
        wind=_t(opt.wind, dtype),
        jacobian=JacobianType(int(opt.jacobian)),
        cone=ConeType(int(opt.cone)),
        disableflags=DisableBit(int(opt.disableflags)),
        enableflags=int(opt.enableflags),
        integrator=IntegratorType(int(opt.integrator)),
        solver=SolverType(int(opt.solver)),
        timestep=_t(opt.timestep, dtype),
        o_margin=_t(_get(opt, "o_margin", 0.0), dtype),
        o_solref=_t(_get(opt, "o_solref", lambda: np.array([0.02, 1.0])), dtype),
        o_solimp=_t(_get(opt, "o_solimp", lambda: np.array([0.9, 0.95, 0.001, 0.5, 2.0])), dtype),
        o_friction=_t(_get(opt, "o_friction", lambda: np.array([1, 1, 0.005, 1e-4, 1e-4])), dtype),
        disableactuator=int(_get(opt, "disableactuator", 0)),
        sdf_initpoints=int(_get(opt, "sdf_initpoints", 40)),
        has_fluid_params=bool(float(opt.density) > 0 or float(opt.viscosity) > 0 or np.any(np.asarray(opt.wind) != 0)),
        batch_size=[],
    )


_ST = SensorType
_SNS_SITE = (int(_ST.ACCELEROMETER), int(_ST.VELOCIMETER), int(_ST.GYRO), int(_ST.RANGEFINDER), int(_ST.FORCE), int(_ST.TORQUE), int(_ST.MAGNETOMETER))
_SNS_FRAME = (int(_ST.FRAMEPOS), int(_ST.FRAMEQUAT), int(_ST.FRAMEXAXIS), int(_ST.FRAMEYAXIS), int(_ST.FRAMEZAXIS), int(_ST.FRAMELINVEL), int(_ST.FRAMEANGVEL))
_SNS_PLAIN = (int(_ST.TENDONPOS), int(_ST.TENDONVEL), int(_ST.ACTUATORPOS), int(_ST.ACTUATORVEL), int(_ST.ACTUATORFRC), int(_ST.TENDONACTFRC),
              int(_ST.SUBTREECOM), int(_ST.SUBTREELINVEL), int(_ST.SUBTREEANGMOM), int(_ST.CLOCK))
# every type one of the reference's three stage functions has a branch for (sensor.py:92-218, 236-333, 373-425): the slots of all others are left
# untouched there (`else: continue`) and here -- touch, camprojection, the joint / tendon limit sensors, framelinacc / frameangacc, the energies, ...
_SNS_EVALUATED = frozenset(_SNS_SITE + _SNS_FRAME + _SNS_PLAIN + (int(_ST.JOINTPOS), int(_ST.JOINTVEL), int(_ST.JOINTACTFRC), int(_ST.BALLQUAT), int(_ST.BALLANGVEL)))
# ... and the ones that read a Data leaf no stage of the reference writes (ABI: MJH_DATA_EXTRA_IN)
SENSOR_EXTRA_LEAVES = {int(_ST.ACCELEROMETER): "cacc", int(_ST.FORCE): "cfrc_int", int(_ST.TORQUE): "cfrc_int", int(_ST.SUBTREELINVEL): "subtree_linvel", int(_ST.SUBTREEANGMOM): "subtree_angmom"}
_RAY_GEOM_ORDER = (GeomType.PLANE, GeomType.SPHERE, GeomType.CAPSULE, GeomType.ELLIPSOID, GeomType.CYLINDER, GeomType.BOX)  # ray.py:282-289


def _sensor_tables(m) -> dict:
    """Sensors the native stepper evaluates (reference sensor.py:56-440, device.py:381-585), flattened per sensor."""
    ns, nsd = int(getattr(m, "nsensor", 0) or 0), int(getattr(m, "nsensordata", 0) or 0)
    out = dict(type=[], adr=[], objid=[], bodyid=[], rootid=[], objtype=[], reftype=[], refid=[], refbodyid=[], refrootid=[], datatype=[], cutoff=[], rfadr=[0], rf_geom=[],
               slot=-np.ones(nsd, dtype=np.int32), extra_leaves=())
    if ns == 0 or (int(m.opt.disableflags) & DisableBit.SENSOR):
        return out
    A = lambda name: np.asarray(getattr(m, name))
    stype, sobj = A("sensor_type"), A("sensor_objid")
    sobjtype = A("sensor_objtype") if hasattr(m, "sensor_objtype") else np.zeros(ns, dtype=np.int32)
    sreftype = A("sensor_reftype") if hasattr(m, "sensor_reftype") else np.zeros(ns, dtype=np.int32)
    srefid = A("sensor_refid") if hasattr(m, "sensor_refid") else -np.ones(ns, dtype=np.int32)
    body_rootid = A("body_rootid")
    # body an object of a frame sensor rides on (device.py:525-532; an absent reference reads entry -1 of a one-element table there: the world)
    body_of = {0: lambda i: 0, 1: lambda i: i, 2: lambda i: i, 5: lambda i: int(A("geom_bodyid")[i]), 6: lambda i: int(A("site_bodyid")[i]), 7: lambda i: int(A("cam_bodyid")[i])}
    extra = set()
    for i in range(ns):
        t, oid = int(stype[i]), int(sobj[i])
        if t not in _SNS_EVALUATED:
            continue
        if t == int(_ST.TENDONACTFRC) and not hasattr(SensorType, "TENDONACTFRC"):
            continue
        obj, body, root, ot, rt, rid, rbody, rroot = oid, 0, 0, int(sobjtype[i]), 0, -1, 0, 0
        if t in _SNS_SITE:
            body = int(A("site_bodyid")[oid])
            root = int(body_rootid[body])
        elif t in (int(_ST.JOINTPOS), int(_ST.BALLQUAT)):
            obj = int(A("jnt_qposadr")[oid])
        elif t in (int(_ST.JOINTVEL), int(_ST.BALLANGVEL), int(_ST.JOINTACTFRC)):
            obj = int(A("jnt_dofadr")[oid])
        elif t in _SNS_FRAME:
            rt, rid = int(sreftype[i]), int(srefid[i])
            if ot not in body_of or rt not in body_of:
                raise NotImplementedError(f"sensor {i}: frame sensors take body / xbody / geom / site / camera objects (objtype {ot}, reftype {rt})")
            body = body_of[ot](oid)
            root = int(body_rootid[body])
            rbody = body_of[rt](rid) if rid >= 0 else 0
            rroot = int(body_rootid[rbody])
        k = len(out["type"])
        dim = int(A("sensor_dim")[i])
        out["slot"][int(A("sensor_adr")[i]) : int(A("sensor_adr")[i]) + dim] = k
        out["type"].append(t); out["adr"].append(int(A("sensor_adr")[i])); out["objid"].append(obj)
        out["bodyid"].append(body); out["rootid"].append(root)
        out["objtype"].append(ot); out["reftype"].append(rt); out["refid"].append(rid); out["refbodyid"].append(rbody); out["refrootid"].append(rroot)
        # (the reference applies the data type of the FIRST sensor of a type to the whole group, device.py:409: the compiler gives every sensor of a type the same one)
        out["datatype"].append(int(A("sensor_datatype")[i])); out["cutoff"].append(float(A("sensor_cutoff")[i]))
        if t in SENSOR_EXTRA_LEAVES:
            extra.add(SENSOR_EXTRA_LEAVES[t])
        if t == int(_ST.RANGEFINDER):  # ray.precompute_ray_data(flg_static=True, bodyexclude=site body), type-major order
            gtype, gbody = A("geom_type"), A("geom_bodyid")
            rgba = np.asarray(getattr(m, "geom_rgba", np.ones((int(m.ngeom), 4))))
            matid = np.asarray(getattr(m, "geom_matid", -np.ones(int(m.ngeom), dtype=np.int32)))
            for gt in _RAY_GEOM_ORDER:
                for g in range(int(m.ngeom)):
                    visible = (matid[g] != -1) or (rgba[g, 3] != 0)  # material alpha is not modelled by the MJCF subset
                    if int(gtype[g]) == int(gt) and int(gbody[g]) != body and visible:
                        out["rf_geom"].append(g)
        out["rfadr"].append(len(out["rf_geom"]))
    out["extra_leaves"] = tuple(sorted(extra))
    return out


class StaticTables:
    """Everything about a model that is constant across steps and environments.

    * ``pairs``: ordered geom pairs with their pair-function id and destination contact slots;
    * ``con_*``: per-contact static fields in FINAL (post-argsort) order;
    * ``desc_int`` / ``desc_real``: the arrays of ``mjhModelDesc`` (include/mjhip.h);
    * ``native``: cache of device blobs keyed by (device index, dtype), filled by ``native.py``.
    """

    def __init__(self):
        self.native = weakref.WeakValueDictionary()  # value digest -> device blob, alive while some Model (its _native_cache) or `native_recent` holds it
        self.native_recent = collections.OrderedDict()  # strong references to the few most recently used blobs (native._NATIVE_LRU)
        self.uid = next(_cache_id_counter)  # never reused (unlike id()): keys the host-side plan / pointer caches

    # process-local caches (device blobs, RK4 workspaces) never travel: a pickled / deep-copied tables object starts with empty ones and a
    # uid of its own (the uid keys caches of THIS process: a loaded value could collide with one handed out here)
    _LOCAL = ("native", "native_recent", "_workspaces", "uid")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._LOCAL}

    def __setstate__(self, state):
        self.__init__()
        self.__dict__.update(state)


def _build_tables(m, dtype) -> StaticTables:
    T = StaticTables()
    flags = int(m.opt.disableflags)
    T.convex = convex.geom_convex_tables(m)
    cands = ct.collision_candidates(m, convex_shape_key=lambda g: convex.shape_key(T.convex, g))
    ct.validate_candidates(cands)
    T.candidates = cands
    dims_sorted = ct.make_condim(m, cands)
    sizes = ct.constraint_sizes(m, dims_sorted)
    if flags & (DisableBit.CONSTRAINT | DisableBit.CONTACT):
        counts = (0, 0, 0, 0)
    else:
        counts = tuple(int(sum(1 for d in dims_sorted if d == c)) for c in (1, 3, 4, 6))
    T.constraint_sizes = sizes
    T.condim_counts = counts
    ne, nf, nl, ncon, nefc = sizes
    pairs = ct.ordered_pairs(cands) if ncon > 0 else []
    T.pairs = pairs
    T.total_contacts = sum(p[1] for p in ct.ordered_pairs(cands))
    # pre-sort per-contact arrays
    dims_unsorted, src_pair, src_k = [], [], []
    for pi, (fn, k, c, _) in enumerate(pairs):
        for kk in range(k):
            dims_unsorted.append(c.dim)
            src_pair.append(pi)
            src_k.append(kk)
    # max_contact_points (collision_driver.py:822-840): every candidate contact is computed, the `ncon` with the smallest dist are
    # kept per environment (torch.topk(-dist)) and re-ordered by the (unstable) argsort of their condims.  The per-contact tables then
    # stay in CANDIDATE (pre-sort) order and the kernels pick by a per-environment source index; `topk_slot[t]` is the final slot of
    # the t-th closest contact.
    cap = ct.max_contact_points(m)
    T.topk = bool(ncon > 0 and cap > -1 and len(dims_unsorted) > cap)
    if T.topk:
        if len(set(dims_unsorted)) != 1:
            raise NotImplementedError("max_contact_points with mixed condims is not supported: the reference sizes its row groups from the "
                                      "smallest condims (collision_driver.py:618-644) whatever the selected contacts' condims are, so a kept "
                                      "condim-3 contact is given a frictionless row and contact.efc_address disagrees with the rows "
                                      "(oracle/probe_reference_topk_mixed.py, profiles/r03/reference_topk_mixed_probe.txt)")
        order = ct.contact_order([dims_unsorted[0]] * ncon)  # final slot s holds the order[s]-th closest contact
        T.topk_slot = np.empty(ncon, dtype=np.int32)
        T.topk_slot[order] = np.arange(ncon)
        perm = np.arange(len(dims_unsorted))
    else:
        T.topk_slot = np.zeros(0, dtype=np.int32)
        perm = ct.contact_order(dims_unsorted)  # final slot s holds pre-sort contact perm[s]
    T.contact_perm = perm
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    pair_dst = -np.ones((len(pairs), 4), dtype=np.int32)
    for pre, (pi, kk) in enumerate(zip(src_pair, src_k)):
        pair_dst[pi, kk] = inv[pre]
    T.pair_dst = pair_dst
    con_pair = np.array([src_pair[p] for p in perm], dtype=np.int32)
    T.con_dim = np.array([dims_unsorted[p] for p in perm], dtype=np.int32)
    T.con_geom1 = np.array([pairs[pi][2].geom1 for pi in con_pair], dtype=np.int32)
    T.con_geom2 = np.array([pairs[pi][2].geom2 for pi in con_pair], dtype=np.int32)
    T.con_pair = con_pair
    elliptic = int(m.opt.cone) == ConeType.ELLIPTIC
    ns = ne + nf + nl
    rows = [1 if d == 1 else (d if elliptic else 2 * (d - 1)) for d in T.con_dim[:ncon]]  # (top-k: uniform condim, one entry per kept contact)
    T.con_rows = np.array(rows, dtype=np.int32)
    T.con_efc_address = (ns + np.concatenate([[0], np.cumsum(rows)[:-1]])).astype(np.int32) if ncon else np.zeros(0, dtype=np.int32)
    # the address collision() writes first (pyramidal-style, collision_driver.py:847-850) is
    # overwritten by make_constraint's cone-aware one (constraint.py:636-646): only the latter
    # reaches the returned Data.
    lim, lim_ball = [], []
    if not (flags & (DisableBit.CONSTRAINT | DisableBit.LIMIT)):
        jt = np.asarray(m.jnt_type)
        for j in range(int(m.njnt)):
            if bool(np.asarray(m.jnt_limited)[j]) and int(jt[j]) in (int(JointType.SLIDE), int(JointType.HINGE)):
                lim.append(j)
            elif bool(np.asarray(m.jnt_limited)[j]) and int(jt[j]) == int(JointType.BALL):
                lim_ball.append(j)
    T.lim_ball_jnt = np.array(lim_ball, dtype=np.int32)
    fric = []
    if not (flags & (DisableBit.CONSTRAINT | DisableBit.FRICTIONLOSS)):
        fric = [d for d in range(int(m.nv)) if float(np.asarray(m.dof_frictionloss)[d]) > 0]
    T.fric_dof = np.array(fric, dtype=np.int32)
    fric_t = []
    if not (flags & (DisableBit.CONSTRAINT | DisableBit.FRICTIONLOSS)) and int(_get(m, "ntendon", 0)):
        fric_t = [t for t in range(int(m.ntendon)) if float(np.asarray(m.tendon_frictionloss)[t]) > 0]
    T.fric_tendon = np.array(fric_t, dtype=np.int32)  # tendon-frictionloss rows follow the dof ones (constraint.py:215-251)
    assert len(fric) + len(fric_t) == nf, (len(fric), len(fric_t), nf)
    T.eq = _equality_tables(m, flags)
    assert T.eq["nrow"] == ne, (T.eq["nrow"], ne)
    T.sensors = _sensor_tables(m)
    T.lim_jnt = np.array(lim, dtype=np.int32)
    T.tendon = _tendon_tables(m, flags)
    T.nlt = len(T.tendon["lim"])  # tendon limit rows
    assert len(lim) + len(lim_ball) + T.nlt == nl, (len(lim), len(lim_ball), nl)
    return T


def _tendon_tables(m, flags) -> dict:
    """Fixed tendons as CSR rows over dofs (smooth.tendon :470-497: length = sum coef * qpos, ten_J[t, dof] = coef) and the list of
    limited tendons in row order (device.py:371-375)."""
    nt = int(_get(m, "ntendon", 0))
    out = dict(adr=[0], dof=[], qpos=[], coef=[], lim=[])
    for t in range(nt):
        a, n = int(m.tendon_adr[t]), int(m.tendon_num[t])
        if n and int(np.asarray(m.wrap_type)[a]) != 1:  # a spatial tendon: the reference leaves ten_length = 0 and ten_J = 0 for it (smooth.py:470-497)
            out["adr"].append(len(out["dof"]))
            continue
        for w in range(a, a + n):
            j = int(m.wrap_objid[w])
            out["dof"].append(int(m.jnt_dofadr[j])); out["qpos"].append(int(m.jnt_qposadr[j])); out["coef"].append(float(m.wrap_prm[w]))
        out["adr"].append(len(out["dof"]))
    if nt and not (flags & (DisableBit.CONSTRAINT | DisableBit.LIMIT)):
        out["lim"] = [t for t in range(nt) if bool(np.asarray(m.tendon_limited)[t])]
    return out


# state / input fields of MjData that device_put(MjData) carries over; everything derived is recomputed by the first step
_DATA_STATE_FIELDS = ("time", "qpos", "qvel", "act", "qacc_warmstart", "ctrl", "qfrc_applied", "xfrc_applied", "mocap_pos", "mocap_quat",
                      "qacc", "act_dot", "eq_active", "sensordata", "userdata")


def _device_put_data(value, dtype):
    """``device_put(MjData)`` (reference device.py:1011-1112) for any object with MjData's array attributes: a ``Data`` whose state
    and input leaves are copied from ``value``.  The model comes from ``value.model`` (MuJoCo >= 3.1 keeps it on MjData)."""
    from .io import make_data

    model = getattr(value, "model", None)
    if model is None:
        raise NotImplementedError("device_put(MjData) needs value.model (mujoco >= 3.1) to size the Data")
    mx = model if isinstance(model, Model) else device_put(model)
    d = make_data(mx)
    kw = {}
    for name in _DATA_STATE_FIELDS:
        if not hasattr(value, name):
            continue
        cur = getattr(d, name)
        a = np.asarray(getattr(value, name))
        if a.size != cur.numel():
            continue
        kw[name] = torch.as_tensor(a.reshape(tuple(cur.shape))).to(cur.dtype).clone()
    d = d.replace(**kw)
    return d.to(dtype) if dtype is not None else d


def device_get_into(result, value):
    """Copies a ``Data`` off the device into MjData-like object(s) (reference device.py:1119-1205): a single object receives the
    arrays with the batch dimension intact, a list must have one entry per environment.  Fields are matched by attribute name
    (contact leaves go to ``result.contact`` when it exists); fields the target lacks or holds with another shape are skipped."""
    if isinstance(result, (list, tuple)):
        bs = tuple(value.batch_size)
        if len(bs) < 1:
            raise ValueError("unrecognizable batch dimension in value")
        if len(result) != bs[0]:
            raise ValueError(f"result length ({len(result)}) doesn't match value batch size ({bs[0]})")
        host = value.to("cpu")
        for i, r in enumerate(result):
            device_get_into(r, host[i])
        return

    def put(target, name, v):
        if isinstance(v, UnbatchedTensor):
            v = v.data
        if not isinstance(v, torch.Tensor):
            return
        a = v.detach().cpu().numpy()
        cur = getattr(target, name, None)
        if cur is None:
            return
        if hasattr(cur, "shape") and tuple(np.shape(cur)) != a.shape:
            if np.size(cur) != a.size:
                return
            a = a.reshape(np.shape(cur))
        try:
            setattr(target, name, a)
        except (AttributeError, ValueError, TypeError):
            getattr(target, name)[...] = a

    for name, v in value.items():
        if name == "contact":
            con = getattr(result, "contact", None)
            if con is not None:
                for cn, cv in v.items():
                    put(con, "dim" if cn == "contact_dim" else cn, cv)
            continue
        put(result, name, v)


def _equality_tables(m, flags) -> dict:
    """Equality constraints in the reference's ROW order: all connects, all welds, all joint couplings (constraint.py:651-656,
    groups from device.py:296-319).  ``jadr`` = (dofadr1, dofadr2, qposadr1, qposadr2) of joint couplings; a missing second
    joint (id -1) indexes the LAST joint there, exactly as the reference's numpy lookup ``jnt_dofadr[-1]`` does."""
    et = np.asarray(getattr(m, "eq_type", np.zeros(0, dtype=np.int32)))
    out = dict(kind=[], id=[], obj1=[], obj2=[], row=[], jadr=[], nrow=0)
    if (flags & (DisableBit.CONSTRAINT | DisableBit.EQUALITY)) or len(et) == 0:
        return {k: (np.zeros((0, 4) if k == "jadr" else 0, dtype=np.int32) if k != "nrow" else 0) for k in out}
    row = 0
    for kind, width in ((0, 3), (1, 6), (2, 1)):
        for i in np.nonzero(et == kind)[0]:
            o1, o2 = int(m.eq_obj1id[i]), int(m.eq_obj2id[i])
            out["kind"].append(kind); out["id"].append(int(i)); out["obj1"].append(o1); out["obj2"].append(o2); out["row"].append(row)
            if kind == 2:
                da, qa = np.asarray(m.jnt_dofadr), np.asarray(m.jnt_qposadr)
                out["jadr"].append([int(da[o1]), int(da[o2]), int(qa[o1]), int(qa[o2])])
            else:
                out["jadr"].append([0, 0, 0, 0])
            row += width
    res = {k: np.array(v, dtype=np.int32) for k, v in out.items() if k != "nrow"}
    res["jadr"] = res["jadr"].reshape(-1, 4)
    res["nrow"] = row
    return res


def static_contact_fields(m_floats: dict, T: StaticTables, dtype):
    """Per-contact static Contact leaves in final order, evaluated in ``dtype`` (collision_driver.py:691-793)."""
    ncon = len(T.con_dim)
    if ncon == 0:
        z = lambda *s: torch.zeros((0,) + s, dtype=dtype)
        return dict(includemargin=z(), friction=z(5), solref=z(2), solreffriction=z(2), solimp=z(5))
    per_pair = ct.static_contact_params(m_floats, T.pairs, dtype)
    sel = [per_pair[pi] for pi in T.con_pair]
    return dict(
        friction=torch.stack([s[0] for s in sel]).contiguous(),
        solref=torch.stack([s[1] for s in sel]).contiguous(),
        solreffriction=torch.stack([s[2] for s in sel]).contiguous(),
        solimp=torch.stack([s[3] for s in sel]).contiguous(),
        includemargin=torch.stack([s[4] for s in sel]).contiguous(),
    )


def device_put(value, *, dtype: torch.dtype | None = None):
    """Places a compiled model onto the torch side (reference device.py:1029-1112).

    ``value``: ``mjcf.MjModelLite`` or ``mujoco.MjModel``.  ``dtype`` overrides the floating
    dtype of every float leaf (default float64, like the reference).
    """
    if not hasattr(value, "nq") and hasattr(value, "qpos") and hasattr(value, "qvel"):
        return _device_put_data(value, dtype)  # an MjData-like object (device.py:1081-1082)
    if not hasattr(value, "nq") or not hasattr(value, "opt"):
        raise NotImplementedError(f"{type(value)} is not supported for device_put.")
    _validate(value)
    fdtype = dtype or torch.float64
    kw = {}
    for name, kind in MODEL_FIELDS:
        try:
            v = getattr(value, name)
        except AttributeError:
            continue
        if kind == "int":
            kw[name] = int(v)
        elif kind == "np":
            kw[name] = np.array(v)
        elif kind == "t":
            a = np.asarray(v, dtype=np.float64)
            if name == "cam_mat0":
                a = a.reshape(-1, 9)
            kw[name] = torch.tensor(a, dtype=fdtype)
        elif kind == "ut":
            kw[name] = UnbatchedTensor(torch.tensor(np.ascontiguousarray(v)))
    st = value.stat
    stat = Statistic(
        meaninertia=float(st.meaninertia),
        meanmass=_t(_get(st, "meanmass", 0.0), fdtype),
        meansize=_t(_get(st, "meansize", 0.0), fdtype),
        extent=_t(_get(st, "extent", 1.0), fdtype),
        center=_t(_get(st, "center", lambda: np.zeros(3)), fdtype),
        batch_size=[],
    )
    nv = int(value.nv)
    ij = []
    for i in range(nv):
        j = i
        while j > -1:
            ij.append((i, j))
            j = int(value.dof_parentid[j])
    rows, cols = zip(*ij) if ij else ((), ())
    actuator_info = []
    for i in range(int(value.nu)):
        trnid = int(np.asarray(value.actuator_trnid)[i, 0])
        if int(value.actuator_trntype[i]) == int(TrnType.TENDON):  # no joint behind the transmission: type / addresses are placeholders
            actuator_info.append((int(TrnType.TENDON), trnid, int(JointType.SLIDE), 0, 0))
            continue
        actuator_info.append((int(value.actuator_trntype[i]), trnid, int(value.jnt_type[trnid]), int(value.jnt_dofadr[trnid]), int(value.jnt_qposadr[trnid])))
    T = _build_tables(value, fdtype)
    L = lambda a: torch.as_tensor(np.array(a), dtype=torch.long)
    m = Model(
        opt=_option(value.opt, fdtype),
        stat=stat,
        has_gravcomp=bool(np.any(np.asarray(_get(value, "body_gravcomp", lambda: np.zeros(int(value.nbody)))) != 0)),
        dof_tri_row=np.array(rows, dtype=np.int64),
        dof_tri_col=np.array(cols, dtype=np.int64),
        actuator_info=tuple(actuator_info),
        actuator_moment_is_batched_py=False,
        constraint_sizes_py=T.constraint_sizes,
        condim_counts_py=T.condim_counts,
        condim_tensor_py=torch.tensor(sorted(int(d) for d in T.con_dim), dtype=torch.long),
        collision_max_cp_py=int(ct.max_contact_points(value)),
        collision_total_contacts_py=int(T.total_contacts),
        cache_id=next(_cache_id_counter),
        body_rootid_t=L(value.body_rootid),
        dof_bodyid_t=L(value.dof_bodyid),
        dof_jntid_t=L(value.dof_jntid),
        geom_bodyid_t=L(value.geom_bodyid),
        site_bodyid_t=L(value.site_bodyid),
        cam_bodyid_t=L(value.cam_bodyid),
        light_bodyid_t=L(value.light_bodyid),
        geom_convex_face=tuple(None if c is None else torch.tensor(c["face"]) for c in T.convex),
        geom_convex_vert=tuple(None if c is None else torch.tensor(c["vert"], dtype=fdtype) for c in T.convex),
        geom_convex_edge=tuple(None if c is None else torch.tensor(c["edge"]) for c in T.convex),
        geom_convex_facenormal=tuple(None if c is None else torch.tensor(c["facenormal"], dtype=fdtype) for c in T.convex),
        batch_size=[],
        **kw,
    )
    T.source = value
    T.structure_key = (int(value.opt.cone), int(value.opt.disableflags) & (0b11111 | (1 << 13)), int(value.opt.jacobian))  # native._structure_key
    object.__setattr__(m, "_tables", T)
    from .types import _register_structure

    _register_structure(f"u{T.uid}", m)
    object.__setattr__(m, "_struct_uid", f"u{T.uid}")
    return m
