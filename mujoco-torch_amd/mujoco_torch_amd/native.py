"""ctypes binding of the C ABI declared in ``include/mjhip.h``.

PyTorch is used only for device memory and streams; the step itself is the hand-written HIP
library ``libmjhip.so`` (built in-tree by ``__graft_entry__.build()``).  There is NO CPU or
PyTorch fallback: if the library is missing, or the tensors are not on a HIP device, the call
raises.  The structure layouts are derived from the header's X-macro lists so the binding cannot
drift from the ABI.
"""

from __future__ import annotations

import ctypes
import os
import re

import numpy as np
import torch

from .container import UnbatchedTensor

from .io import model_float_leaves
from .device import static_contact_fields

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_PKG_DIR))
HEADER = os.path.join(_ROOT, "include", "mjhip.h")
LIB_PATH = os.environ.get("MJH_LIB") or os.path.join(os.path.dirname(_PKG_DIR), "lib", "libmjhip.so")  # MJH_LIB: A/B builds

MJH_F64, MJH_F32 = 0, 1
FLAG_FIXED_ITERATIONS = 1
STAGE_ALL = 0x7F


def _macro_names(text: str, macro: str):
    m = re.search(r"#define\s+" + macro + r"\(X\)(.*?)(?:\n\s*\n|\n#|\ntypedef)", text, re.S)
    if m is None:
        raise RuntimeError(f"{macro} not found in {HEADER}")
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    return re.findall(r"X\((\w+)\)", body)


def _load_lists():
    with open(HEADER) as f:
        text = f.read()
    return {k: _macro_names(text, k) for k in (
        "MJH_MODEL_INTS", "MJH_MODEL_REALS", "MJH_MODEL_INT_ARRAYS", "MJH_MODEL_REAL_ARRAYS",
        "MJH_DATA_REALS", "MJH_DATA_I32", "MJH_DATA_I64", "MJH_DATA_EXTRA_IN")}


LISTS = _load_lists()
with open(HEADER) as _f:
    ABI_VERSION = int(re.search(r"#define MJH_ABI_VERSION (\d+)", _f.read()).group(1))  # the header is the single source of the version


def _make_structs():
    f = [("abi_version", ctypes.c_int32)]
    f += [(n, ctypes.c_int32) for n in LISTS["MJH_MODEL_INTS"]]
    f += [(n, ctypes.c_double) for n in LISTS["MJH_MODEL_REALS"]]
    f += [(n, ctypes.c_void_p) for n in LISTS["MJH_MODEL_INT_ARRAYS"]]
    f += [(n, ctypes.c_void_p) for n in LISTS["MJH_MODEL_REAL_ARRAYS"]]
    f += [("len_" + n, ctypes.c_int64) for n in LISTS["MJH_MODEL_INT_ARRAYS"] + LISTS["MJH_MODEL_REAL_ARRAYS"]]

    class ModelDesc(ctypes.Structure):
        _fields_ = f

    d = [(n, ctypes.c_void_p) for n in LISTS["MJH_DATA_REALS"] + LISTS["MJH_DATA_I32"] + LISTS["MJH_DATA_I64"] + LISTS["MJH_DATA_EXTRA_IN"]]  # (the trailing input-only leaves: NULL unless a sensor reads them)

    class DataPtrs(ctypes.Structure):
        _fields_ = d

    return ModelDesc, DataPtrs


ModelDesc, DataPtrs = _make_structs()

# Data leaf name -> attribute path on the Data container
DATA_PATH = {n: (("contact", n[len("contact_"):]) if n.startswith("contact_") else (n,)) for n in
             LISTS["MJH_DATA_REALS"] + LISTS["MJH_DATA_I32"] + LISTS["MJH_DATA_I64"]}
DATA_PATH["contact_dim"] = ("contact", "contact_dim")


def _tendon_reals(src, T, dtype):
    """Tendon parameters from the source model, rounded through the compute dtype like every other real table."""
    nt = len(T.tendon["adr"]) - 1
    r = lambda a, shape: np.ascontiguousarray(torch.tensor(np.asarray(a, dtype=np.float64).reshape(shape)).to(dtype).to(torch.float64).numpy()).reshape(-1)
    g = lambda name, shape: r(getattr(src, name), shape) if nt else np.zeros(0)
    return dict(ten_coef=r(T.tendon["coef"], (-1,)), tendon_range=g("tendon_range", (nt, 2)), tendon_margin=g("tendon_margin", (nt,)),
                tendon_invweight0=g("tendon_invweight0", (nt,)), tendon_solref_lim=g("tendon_solref_lim", (nt, 2)),
                tendon_solimp_lim=g("tendon_solimp_lim", (nt, 5)), tendon_stiffness=g("tendon_stiffness", (nt,)),
                tendon_damping=g("tendon_damping", (nt,)), tendon_frictionloss=g("tendon_frictionloss", (nt,)), tendon_solref_fri=g("tendon_solref_fri", (nt, 2)), tendon_solimp_fri=g("tendon_solimp_fri", (nt, 5)), tendon_lengthspring=g("tendon_lengthspring", (nt, 2)), tendon_armature=g("tendon_armature", (nt,)))


def pack_model(m, dtype: torch.dtype):
    """Model -> (ModelDesc, keepalive list of numpy arrays). Floats are evaluated in ``dtype``."""
    T = m.tables
    src = T.source
    ne, nf, nl, ncon, nefc = m.constraint_sizes_py
    ints = dict(
        nq=m.nq, nv=m.nv, nu=m.nu, na=m.na, nbody=m.nbody, njnt=m.njnt, ngeom=m.ngeom, nsite=m.nsite,
        ncam=m.ncam, nlight=m.nlight, nmocap=m.nmocap, ne=ne, nf=len(T.fric_dof), nft=len(T.fric_tendon), nl=len(T.lim_jnt), nlb=len(T.lim_ball_jnt), nlt=int(T.nlt), ncon=ncon, ncand=len(T.con_dim), topk=int(T.topk), nefc=nefc, neq=int(m.neq), neqtab=len(T.eq['kind']), ntendon=int(m.ntendon), nwrapj=len(T.tendon['dof']),
        npair=len(T.pairs), nconvex=0, nsensor=len(T.sensors["type"]), nsensordata=int(getattr(m, "nsensordata", 0) or 0), integrator=int(m.opt.integrator), solver=int(m.opt.solver),
        cone=int(m.opt.cone), disableflags=int(m.opt.disableflags), iterations=int(m.opt.iterations),
        ls_iterations=int(m.opt.ls_iterations),
    )
    f64 = lambda t: np.ascontiguousarray(t.detach().to("cpu", dtype).to(torch.float64).numpy()).reshape(-1)
    grav = f64(m.opt.gravity)
    reals = dict(
        timestep=float(f64(m.opt.timestep)[0]), impratio=float(f64(m.opt.impratio)[0]),
        tolerance=float(m.opt.tolerance), ls_tolerance=float(m.opt.ls_tolerance),
        meaninertia=float(m.stat.meaninertia), gravity_x=float(grav[0]), gravity_y=float(grav[1]), gravity_z=float(grav[2]),
        density=float(f64(m.opt.density)[0]), viscosity=float(f64(m.opt.viscosity)[0]),
        wind_x=float(f64(m.opt.wind)[0]), wind_y=float(f64(m.opt.wind)[1]), wind_z=float(f64(m.opt.wind)[2]),
        magnetic_x=float(f64(m.opt.magnetic)[0]), magnetic_y=float(f64(m.opt.magnetic)[1]), magnetic_z=float(f64(m.opt.magnetic)[2]),
    )
    i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32)).reshape(-1)
    nu = m.nu
    info = m.actuator_info
    A = lambda name: np.asarray(getattr(m, name))
    int_arrays = dict(
        body_parentid=i32(A("body_parentid")), body_rootid=i32(A("body_rootid")), body_jntadr=i32(A("body_jntadr")),
        body_jntnum=i32(A("body_jntnum")), body_dofadr=i32(A("body_dofadr")), body_dofnum=i32(A("body_dofnum")),
        body_mocapid=i32(A("body_mocapid")), jnt_type=i32(m.jnt_type.data.cpu().numpy()), jnt_qposadr=i32(A("jnt_qposadr")),
        jnt_dofadr=i32(A("jnt_dofadr")), jnt_bodyid=i32(A("jnt_bodyid")), jnt_actfrclimited=i32(A("jnt_actfrclimited")), jnt_actgravcomp=i32(A("jnt_actgravcomp")),
        dof_bodyid=i32(A("dof_bodyid")), dof_jntid=i32(A("dof_jntid")), dof_parentid=i32(A("dof_parentid")),
        geom_type=i32(A("geom_type")), geom_bodyid=i32(A("geom_bodyid")), geom_convexid=-np.ones(m.ngeom, dtype=np.int32),
        site_bodyid=i32(A("site_bodyid")), cam_bodyid=i32(A("cam_bodyid")), cam_mode=i32(A("cam_mode")),
        cam_targetbodyid=i32(A("cam_targetbodyid")), light_bodyid=i32(A("light_bodyid")),
        act_trntype=i32([x[0] for x in info]), act_jnttype=i32([x[2] for x in info]), act_dofadr=i32([x[3] for x in info]),
        act_qposadr=i32([x[4] for x in info]), act_gaintype=i32(m.actuator_gaintype.data.cpu().numpy()),
        act_biastype=i32(m.actuator_biastype.data.cpu().numpy()), act_dyntype=i32(m.actuator_dyntype.data.cpu().numpy()),
        act_ctrllimited=i32(A("actuator_ctrllimited")), act_forcelimited=i32(A("actuator_forcelimited")),
        act_actlimited=i32(A("actuator_actlimited")), act_actadr=i32(A("actuator_actadr")), act_actnum=i32(A("actuator_actnum")),
        sns_type=i32(T.sensors["type"]), sns_adr=i32(T.sensors["adr"]), sns_objid=i32(T.sensors["objid"]), sns_bodyid=i32(T.sensors["bodyid"]),
        sns_rootid=i32(T.sensors["rootid"]), sns_objtype=i32(T.sensors["objtype"]), sns_reftype=i32(T.sensors["reftype"]), sns_refid=i32(T.sensors["refid"]),
        sns_refbodyid=i32(T.sensors["refbodyid"]), sns_refrootid=i32(T.sensors["refrootid"]), sns_datatype=i32(T.sensors["datatype"]), sns_rfadr=i32(T.sensors["rfadr"]),
        rf_geom=i32(T.sensors["rf_geom"]), slot_sensor=i32(T.sensors["slot"]),
        eq_kind=i32(T.eq['kind']), eq_id=i32(T.eq['id']), eq_obj1=i32(T.eq['obj1']), eq_obj2=i32(T.eq['obj2']), eq_row=i32(T.eq['row']), eq_jadr=i32(T.eq['jadr']),
        topk_slot=i32(T.topk_slot), fric_dof=i32(T.fric_dof), fric_tendon=i32(T.fric_tendon), ten_adr=i32(T.tendon['adr']), ten_dof=i32(T.tendon['dof']), ten_qposadr=i32(T.tendon['qpos']), lim_tendon=i32(T.tendon['lim']), act_trnid=i32([x[1] for x in info]),
        lim_ball_jnt=i32(T.lim_ball_jnt), lim_jnt=i32(T.lim_jnt), pair_fn=i32([p[0] for p in T.pairs]), pair_geom1=i32([p[2].geom1 for p in T.pairs]),
        pair_geom2=i32([p[2].geom2 for p in T.pairs]), pair_ncon=i32([p[1] for p in T.pairs]), pair_dst=i32(T.pair_dst),
        con_dim=i32(T.con_dim), con_geom1=i32(T.con_geom1), con_geom2=i32(T.con_geom2), con_efc_address=i32(T.con_efc_address),
        convex_nvert=i32([]), convex_nface=i32([]), convex_nfv=i32([]), convex_nedge=i32([]), convex_vertadr=i32([]),
        convex_faceadr=i32([]), convex_normadr=i32([]), convex_edgeadr=i32([]), convex_face=i32([]), convex_edge=i32([]),
    )
    # convex tables (box / mesh geoms): one record per convex geom, flat arrays with offsets
    cv = [(g, c) for g, c in enumerate(T.convex) if c is not None]
    convexid = -np.ones(m.ngeom, dtype=np.int32)
    for i, (g, _) in enumerate(cv):
        convexid[g] = i
    off = lambda ns: np.concatenate([[0], np.cumsum(ns)[:-1]]) if ns else []
    int_arrays.update(
        geom_convexid=convexid, convex_nvert=i32([len(c["vert"]) for _, c in cv]), convex_nface=i32([c["face"].shape[0] for _, c in cv]),
        convex_nfv=i32([c["face"].shape[1] for _, c in cv]), convex_nedge=i32([len(c["edge"]) for _, c in cv]),
        convex_vertadr=i32(off([len(c["vert"]) for _, c in cv])), convex_faceadr=i32(off([c["face"].size for _, c in cv])),
        convex_normadr=i32(off([c["face"].shape[0] for _, c in cv])), convex_edgeadr=i32(off([len(c["edge"]) for _, c in cv])),
        convex_face=i32(np.concatenate([c["face"].reshape(-1) for _, c in cv]) if cv else []),
        convex_edge=i32(np.concatenate([c["edge"].reshape(-1) for _, c in cv]) if cv else []),
    )
    ints["nconvex"] = len(cv)
    st = static_contact_fields(model_float_leaves(m, dtype), T, dtype)
    empty = np.zeros(0)
    cvf = lambda k: f64(torch.tensor(np.concatenate([c[k].reshape(-1) for _, c in cv]))) if cv else empty
    real_arrays = dict(
        qpos0=f64(m.qpos0), qpos_spring=f64(m.qpos_spring), body_pos=f64(m.body_pos), body_quat=f64(m.body_quat),
        body_ipos=f64(m.body_ipos), body_iquat=f64(m.body_iquat), body_mass=f64(m.body_mass),
        body_inertia=f64(m.body_inertia), body_invweight0=f64(m.body_invweight0[:, 0]), jnt_pos=f64(m.jnt_pos),
        jnt_axis=f64(m.jnt_axis), jnt_stiffness=f64(m.jnt_stiffness), jnt_range=f64(m.jnt_range),
        jnt_margin=f64(m.jnt_margin), jnt_solref=f64(m.jnt_solref), jnt_solimp=f64(m.jnt_solimp),
        jnt_actfrcrange=f64(m.jnt_actfrcrange), dof_armature=f64(m.dof_armature), dof_damping=f64(m.dof_damping),
        dof_invweight0=f64(m.dof_invweight0), sns_cutoff=f64(torch.tensor(np.asarray(T.sensors["cutoff"], dtype=np.float64))), dof_frictionloss=f64(m.dof_frictionloss), dof_solref=f64(m.dof_solref), dof_solimp=f64(m.dof_solimp),
        **_tendon_reals(src, T, dtype), body_gravcomp=f64(m.body_gravcomp), body_invweight0_rot=f64(m.body_invweight0[:, 1]), eq_data=f64(m.eq_data) if m.neq else empty, eq_solref=f64(m.eq_solref) if m.neq else empty, eq_solimp=f64(m.eq_solimp) if m.neq else empty,
        geom_pos=f64(m.geom_pos), geom_quat=f64(m.geom_quat),
        geom_size=f64(m.geom_size), site_pos=f64(m.site_pos), site_quat=f64(m.site_quat), cam_pos=f64(m.cam_pos),
        cam_quat=f64(m.cam_quat), cam_pos0=f64(m.cam_pos0), cam_mat0=f64(m.cam_mat0), light_pos=f64(m.light_pos),
        light_dir=f64(m.light_dir), act_gear=f64(m.actuator_gear) if nu else empty,
        act_gainprm=f64(m.actuator_gainprm[:, :9]) if nu else empty, act_biasprm=f64(m.actuator_biasprm[:, :9]) if nu else empty,
        act_lengthrange=f64(m.actuator_lengthrange) if nu else empty, act_acc0=f64(m.actuator_acc0) if nu else empty,
        act_dynprm=f64(m.actuator_dynprm[:, :3]) if nu else empty, act_ctrlrange=f64(m.actuator_ctrlrange) if nu else empty,
        act_forcerange=f64(m.actuator_forcerange) if nu else empty, act_actrange=f64(m.actuator_actrange) if nu else empty,
        con_includemargin=f64(st["includemargin"]), con_friction=f64(st["friction"]), con_solref=f64(st["solref"]),
        con_solreffriction=f64(st["solreffriction"]), con_solimp=f64(st["solimp"]), convex_vert=cvf("vert"), convex_facenormal=cvf("facenormal"),
    )
    desc = ModelDesc()
    desc.abi_version = ABI_VERSION
    keep = []
    for n in LISTS["MJH_MODEL_INTS"]:
        setattr(desc, n, int(ints[n]))
    for n in LISTS["MJH_MODEL_REALS"]:
        setattr(desc, n, float(reals[n]))
    for n in LISTS["MJH_MODEL_INT_ARRAYS"]:
        a = int_arrays[n]
        keep.append(a)
        setattr(desc, n, a.ctypes.data if a.size else None)
        setattr(desc, "len_" + n, a.size)
    for n in LISTS["MJH_MODEL_REAL_ARRAYS"]:
        a = np.ascontiguousarray(real_arrays[n], dtype=np.float64)
        keep.append(a)
        setattr(desc, n, a.ctypes.data if a.size else None)
        setattr(desc, "len_" + n, a.size)
    return desc, keep


def data_field_tensor(d, name):
    obj = d
    for p in DATA_PATH[name]:
        obj = getattr(obj, p)
    return obj


_lib = None


def load_library(path: str | None = None):
    """Loads libmjhip.so (once). Raises if it has not been built: no fallback path exists."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"native stepper library not found at {p}; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). mujoco_torch_amd has no CPU / PyTorch fallback."
        )
    lib = ctypes.CDLL(p)
    lib.mjh_model_create.argtypes = [ctypes.POINTER(ModelDesc), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.mjh_model_create.restype = ctypes.c_int
    lib.mjh_model_destroy.argtypes = [ctypes.c_void_p]
    lib.mjh_model_destroy.restype = None
    lib.mjh_forward.argtypes = [ctypes.c_void_p, ctypes.POINTER(DataPtrs), ctypes.POINTER(DataPtrs), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.mjh_forward.restype = ctypes.c_int
    lib.mjh_step.argtypes = [ctypes.c_void_p, ctypes.POINTER(DataPtrs), ctypes.POINTER(DataPtrs), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    lib.mjh_step.restype = ctypes.c_int
    lib.mjh_reset_where.argtypes = [ctypes.c_void_p, ctypes.POINTER(DataPtrs), ctypes.POINTER(DataPtrs), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.mjh_reset_where.restype = ctypes.c_int
    lib.mjh_debug_phase_timing.argtypes = [ctypes.c_int]
    lib.mjh_debug_phase_timing.restype = ctypes.c_int
    lib.mjh_debug_phase_times.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.mjh_debug_phase_times.restype = ctypes.c_int
    lib.mjh_model_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mjh_model_lds_bytes.restype = ctypes.c_int
    lib.mjh_model_work_bytes.argtypes = [ctypes.c_void_p]
    lib.mjh_model_work_bytes.restype = ctypes.c_int64
    lib.mjh_model_kernel_io.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    lib.mjh_model_kernel_io.restype = ctypes.c_int
    lib.mjh_model_leaf_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    lib.mjh_model_leaf_counts.restype = ctypes.c_int
    for fn in ("mjh_last_error", "mjh_data_fields", "mjh_model_fields"):
        getattr(lib, fn).restype = ctypes.c_char_p
    lib.mjh_abi_version.restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


def check_abi(lib):
    """Asserts the header this binding was generated from matches the loaded library."""
    data = lib.mjh_data_fields().decode().split(",")
    want = LISTS["MJH_DATA_REALS"] + LISTS["MJH_DATA_I32"] + LISTS["MJH_DATA_I64"]
    if data != want:
        raise RuntimeError("libmjhip.so Data field list does not match include/mjhip.h")
    lib.mjh_data_extra_fields.restype = ctypes.c_char_p
    if lib.mjh_data_extra_fields().decode().split(",") != LISTS["MJH_DATA_EXTRA_IN"] or int(lib.mjh_sizeof_data()) != ctypes.sizeof(DataPtrs):
        raise RuntimeError("libmjhip.so mjhData layout (trailing input-only leaves) does not match include/mjhip.h")
    model = lib.mjh_model_fields().decode().split(",")
    wantm = LISTS["MJH_MODEL_INTS"] + LISTS["MJH_MODEL_REALS"] + LISTS["MJH_MODEL_INT_ARRAYS"] + LISTS["MJH_MODEL_REAL_ARRAYS"]
    if model != wantm:
        raise RuntimeError("libmjhip.so Model field list does not match include/mjhip.h")


class NativeModel:
    """Device-resident constant blob for one (device, dtype) and one set of model VALUES."""

    def __init__(self, desc, keep, device: torch.device, dtype: torch.dtype):
        # `keep`: the host arrays `desc` points into; the library copies them into its device blob, nothing is retained
        self.lib = load_library()
        check_abi(self.lib)
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = self.lib.mjh_model_create(ctypes.byref(desc), MJH_F64 if dtype == torch.float64 else MJH_F32, ctypes.byref(handle))
        if rc != 0:
            raise RuntimeError(f"mjh_model_create failed ({rc}): {self.lib.mjh_last_error().decode()}")
        self.handle = handle
        self.device = device
        self.dtype = dtype
        self.work_bytes = int(self.lib.mjh_model_work_bytes(handle))
        self.lds_bytes = [int(self.lib.mjh_model_lds_bytes(handle, p)) for p in range(6)]  # five phases + the register solver's arena
        n = len(DATA_PATH)
        buf = (ctypes.c_int64 * n)()
        got = int(self.lib.mjh_model_leaf_counts(handle, buf, n))
        if got != n:
            raise RuntimeError("libmjhip.so leaf table does not match include/mjhip.h")
        self.leaf_counts = np.array(list(buf), dtype=np.int64)  # per-environment element count of every Data leaf, ABI order
        self._work = {"clock": 0, "bufs": {}}  # replaced by the pool shared through the tables object (get_native_model): the LRU clock lives WITH the pool

    def workspace(self, B: int, stream: int = 0):
        """Scratch for RK4 (stage Data + running sums), one per (batch size, stream): two streams stepping the same model
        must not share stage storage, and a buffer is only ever reused on the stream it was allocated on."""
        if self.work_bytes == 0:
            return None
        key = (B, stream)
        pool = self._work
        bufs = pool["bufs"]
        pool["clock"] += 1  # one clock per pool: every blob of a tables object ages the entries on the same scale (a per-blob clock made a new blob's fresh entries look oldest)
        hit = bufs.get(key)
        if hit is not None:
            hit[1] = pool["clock"]
            return hit[0]
        if len(bufs) >= 4:  # small LRU: ping-pong between a few batch sizes / streams without re-allocating
            del bufs[min(bufs, key=lambda k: bufs[k][1])]
        w = torch.empty(self.work_bytes * B, dtype=torch.uint8, device=self.device)
        bufs[key] = [w, pool["clock"]]
        return w

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mjh_model_destroy(self.handle)
        except Exception:
            pass


# option bits that decide the STATIC tables built by device_put (row counts, contact list, sensor list): changing them on a
# Model after device_put would need new tables and a new Data layout (the reference's own constraint_sizes_py goes stale too)
_STRUCT_DISABLE = 0b11111 | (1 << 13)  # CONSTRAINT, EQUALITY, FRICTIONLOSS, LIMIT, CONTACT, SENSOR


def _structure_key(opt):
    return (int(opt.cone), int(opt.disableflags) & _STRUCT_DISABLE, int(opt.jacobian))


def _stamp(m):
    """Cheap per-call check that nothing pack_model reads has changed since the blob was cached ON THIS CONTAINER: container
    versions (update_ / attribute assignment) and the in-place version counters of every tensor leaf pack_model reads -- Model,
    opt and stat (``stat.meaninertia`` is packed), plain or wrapped in an ``UnbatchedTensor``.  The opt / stat containers are
    compared by identity through references held in the cache entry (an ``id()`` can be reused after garbage collection)."""
    opt, stat = m.opt, m.stat
    v = 0
    for c in (m, opt, stat):
        # the tensor leaves of a container only change with its version (attribute assignment / update_): the list is kept per container and version, so a
        # step pays one `_version` read per tensor instead of walking ~450 fields through isinstance (45 of the host's 110 us per call)
        d = c.__dict__
        ver = d.get("_ver", 0)
        ts = d.get("_stamp_ts")
        if ts is not None and ts[0] == ver:
            for w, t0 in ts[2]:
                if w.data is not t0:  # `unbatched.data = new_tensor` bumps no version counter: the wrapper is re-read every call
                    ts = (ver, None, None, ts[3] + 1)
                    break
        if ts is None or ts[0] != ver or ts[1] is None:
            lst, wrapped = [], []
            for t in c._fields.values():
                if isinstance(t, UnbatchedTensor):
                    if isinstance(t.data, torch.Tensor):
                        wrapped.append((t, t.data))
                    t = t.data
                if isinstance(t, torch.Tensor):
                    lst.append(t)
            ts = (ver, lst, wrapped, ts[3] if ts is not None else 0)
            object.__setattr__(c, "_stamp_ts", ts)
        v += ts[3] << 32
        for t in ts[1]:
            v += t._version
    return (m.__dict__.get("_ver", 0), opt, opt.__dict__.get("_ver", 0), stat, stat.__dict__.get("_ver", 0), v)


def _same_stamp(a, b):
    return a[0] == b[0] and a[1] is b[1] and a[2] == b[2] and a[3] is b[3] and a[4] == b[4] and a[5] == b[5]


_NATIVE_LRU = 4  # blobs the shared tables keep alive on their own (most recently used); every Model also holds the one it stepped


def _desc_digest(desc, keep):
    import hashlib

    h = hashlib.blake2b(digest_size=16)
    h.update(bytes(memoryview(desc))[: ModelDesc.__dict__[LISTS["MJH_MODEL_INT_ARRAYS"][0]].offset])  # the scalar head: ints and reals
    for a in keep:
        h.update(a.dtype.str.encode())
        h.update(np.int64(a.size).tobytes())
        h.update(a.tobytes())
    return h.digest()


def get_native_model(m, device: torch.device, dtype: torch.dtype) -> NativeModel:
    """The device blob for this Model's CURRENT values.  Blobs are shared through ``m.tables.native`` keyed by a digest of
    everything ``pack_model`` hands to the library, so ``mx.replace(body_mass=...)`` / ``mx.tree_replace({'opt.timestep': ..})``
    (reference test/smooth_test.py:204) get a blob of their own while ``mx.to(...)`` copies share one."""
    key = (device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else -1), dtype)
    stamp = _stamp(m)
    local = m.__dict__.get("_native_cache")
    if local is not None:
        hit = local.get(key)
        if hit is not None and _same_stamp(hit[0], stamp):
            return hit[1]
    T = m.tables
    built = getattr(T, "structure_key", None)
    if built is not None and _structure_key(m.opt) != built:
        raise NotImplementedError(
            "opt.cone / opt.jacobian / the constraint-, contact- or sensor-disabling bits of opt.disableflags changed after "
            "device_put: they size the static row and contact tables (and the Data layout).  Set them on the source model and "
            "call device_put again.")
    desc, keep = pack_model(m, dtype)
    shared_key = (key, _desc_digest(desc, keep))
    nm = T.native.get(shared_key)
    if nm is None:
        nm = NativeModel(desc, keep, torch.device(device.type, key[0]) if key[0] >= 0 else device, dtype)
        # RK4 workspaces are sized by the model's STRUCTURE (leaf counts), not its values: one pool per tables object and (device, dtype), so
        # per-episode domain randomisation (mx.replace(body_mass=...)) re-uses the same scratch instead of growing a pool per blob
        nm._work = T.__dict__.setdefault("_workspaces", {}).setdefault((key, nm.work_bytes), {"clock": 0, "bufs": {}})
        T.native[shared_key] = nm
    # T.native holds blobs weakly: one lives as long as a Model that stepped it (its _native_cache) does, so `mx.replace(body_mass=...)` per
    # episode frees the previous episode's blob with its Model instead of leaking device memory.  The few most recently used ones are also
    # held strongly, so alternating between a handful of value sets through fresh `replace` results does not rebuild every time.
    recent = T.native_recent
    recent.pop(shared_key, None)
    recent[shared_key] = nm
    while len(recent) > _NATIVE_LRU:
        recent.popitem(last=False)
    if local is None:
        local = {}
        object.__setattr__(m, "_native_cache", local)
    local[key] = (stamp, nm)
    return nm
