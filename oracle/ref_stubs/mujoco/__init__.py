"""Container-only stand-in for the ``mujoco`` Python package (constants, enums, struct shells).

TEST INFRASTRUCTURE for ``oracle/gen_golden.py``: lets the reference's own ``device_put`` /
``make_data`` / ``step`` run on a model produced by this repo's MJCF-subset compiler
(``mujoco_torch_amd.mjcf``).  No MuJoCo arithmetic is provided except ``mj_kinematics``
(needed by the reference's static-field probe, device.py:967-1007), which calls the numpy
kinematics in ``mjcf``.  Enum numbering follows mujoco 3.5.0 (reference uv.lock:287-288).
"""
import enum
import sys
import types as _pytypes

import numpy as np

__version__ = "3.5.0"

mjMINVAL = 1e-15
mjMAXVAL = 1e10
mjMINIMP = 1e-4
mjMAXIMP = 0.9999
mjMINMU = 1e-5
mjNREF = 2
mjNIMP = 5
mjNEQDATA = 11
mjNGROUP = 6
mjPI = np.pi


def _enum(name, members, flag=False):
    base = enum.IntFlag if flag else enum.IntEnum
    return base(name, members)


mjtDisableBit = _enum(
    "mjtDisableBit",
    dict(
        mjDSBL_CONSTRAINT=1 << 0, mjDSBL_EQUALITY=1 << 1, mjDSBL_FRICTIONLOSS=1 << 2, mjDSBL_LIMIT=1 << 3,
        mjDSBL_CONTACT=1 << 4, mjDSBL_SPRING=1 << 5, mjDSBL_DAMPER=1 << 6, mjDSBL_GRAVITY=1 << 7,
        mjDSBL_CLAMPCTRL=1 << 8, mjDSBL_WARMSTART=1 << 9, mjDSBL_FILTERPARENT=1 << 10, mjDSBL_ACTUATION=1 << 11,
        mjDSBL_REFSAFE=1 << 12, mjDSBL_SENSOR=1 << 13, mjDSBL_MIDPHASE=1 << 14, mjDSBL_EULERDAMP=1 << 15,
        mjDSBL_AUTORESET=1 << 16, mjDSBL_NATIVECCD=1 << 17, mjDSBL_ISLAND=1 << 18, mjNDISABLE=19,
    ),
)
mjtEnableBit = _enum(
    "mjtEnableBit",
    dict(mjENBL_OVERRIDE=1 << 0, mjENBL_ENERGY=1 << 1, mjENBL_FWDINV=1 << 2, mjENBL_INVDISCRETE=1 << 3,
         mjENBL_MULTICCD=1 << 4, mjENBL_SLEEP=1 << 5, mjNENABLE=6),
)
mjtJoint = _enum("mjtJoint", dict(mjJNT_FREE=0, mjJNT_BALL=1, mjJNT_SLIDE=2, mjJNT_HINGE=3))
mjtGeom = _enum(
    "mjtGeom",
    dict(mjGEOM_PLANE=0, mjGEOM_HFIELD=1, mjGEOM_SPHERE=2, mjGEOM_CAPSULE=3, mjGEOM_ELLIPSOID=4,
         mjGEOM_CYLINDER=5, mjGEOM_BOX=6, mjGEOM_MESH=7, mjGEOM_SDF=8),
)
mjtIntegrator = _enum("mjtIntegrator", dict(mjINT_EULER=0, mjINT_RK4=1, mjINT_IMPLICIT=2, mjINT_IMPLICITFAST=3))
mjtCone = _enum("mjtCone", dict(mjCONE_PYRAMIDAL=0, mjCONE_ELLIPTIC=1))
mjtJacobian = _enum("mjtJacobian", dict(mjJAC_DENSE=0, mjJAC_SPARSE=1, mjJAC_AUTO=2))
mjtSolver = _enum("mjtSolver", dict(mjSOL_PGS=0, mjSOL_CG=1, mjSOL_NEWTON=2))
mjtEq = _enum("mjtEq", dict(mjEQ_CONNECT=0, mjEQ_WELD=1, mjEQ_JOINT=2, mjEQ_TENDON=3, mjEQ_FLEX=4, mjEQ_DISTANCE=5))
mjtWrap = _enum("mjtWrap", dict(mjWRAP_NONE=0, mjWRAP_JOINT=1, mjWRAP_PULLEY=2, mjWRAP_SITE=3, mjWRAP_SPHERE=4, mjWRAP_CYLINDER=5))
mjtTrn = _enum("mjtTrn", dict(mjTRN_JOINT=0, mjTRN_JOINTINPARENT=1, mjTRN_SLIDERCRANK=2, mjTRN_TENDON=3, mjTRN_SITE=4, mjTRN_BODY=5))
mjtDyn = _enum("mjtDyn", dict(mjDYN_NONE=0, mjDYN_INTEGRATOR=1, mjDYN_FILTER=2, mjDYN_FILTEREXACT=3, mjDYN_MUSCLE=4, mjDYN_USER=5))
mjtGain = _enum("mjtGain", dict(mjGAIN_FIXED=0, mjGAIN_AFFINE=1, mjGAIN_MUSCLE=2, mjGAIN_USER=3))
mjtBias = _enum("mjtBias", dict(mjBIAS_NONE=0, mjBIAS_AFFINE=1, mjBIAS_MUSCLE=2, mjBIAS_USER=3))
mjtConstraint = _enum(
    "mjtConstraint",
    dict(mjCNSTR_EQUALITY=0, mjCNSTR_FRICTION_DOF=1, mjCNSTR_FRICTION_TENDON=2, mjCNSTR_LIMIT_JOINT=3,
         mjCNSTR_LIMIT_TENDON=4, mjCNSTR_CONTACT_FRICTIONLESS=5, mjCNSTR_CONTACT_PYRAMIDAL=6, mjCNSTR_CONTACT_ELLIPTIC=7),
)
mjtCamLight = _enum(
    "mjtCamLight",
    dict(mjCAMLIGHT_FIXED=0, mjCAMLIGHT_TRACK=1, mjCAMLIGHT_TRACKCOM=2, mjCAMLIGHT_TARGETBODY=3, mjCAMLIGHT_TARGETBODYCOM=4),
)
_SENS = ("TOUCH ACCELEROMETER VELOCIMETER GYRO FORCE TORQUE MAGNETOMETER RANGEFINDER CAMPROJECTION JOINTPOS JOINTVEL "
         "TENDONPOS TENDONVEL ACTUATORPOS ACTUATORVEL ACTUATORFRC JOINTACTFRC TENDONACTFRC BALLQUAT BALLANGVEL "
         "JOINTLIMITPOS JOINTLIMITVEL JOINTLIMITFRC TENDONLIMITPOS TENDONLIMITVEL TENDONLIMITFRC FRAMEPOS FRAMEQUAT "
         "FRAMEXAXIS FRAMEYAXIS FRAMEZAXIS FRAMELINVEL FRAMEANGVEL FRAMELINACC FRAMEANGACC SUBTREECOM SUBTREELINVEL "
         "SUBTREEANGMOM INSIDESITE GEOMDIST GEOMNORMAL GEOMFROMTO CONTACT E_POTENTIAL E_KINETIC CLOCK TACTILE PLUGIN USER").split()
mjtSensor = _enum("mjtSensor", {"mjSENS_" + n: i for i, n in enumerate(_SENS)})
mjtObj = _enum("mjtObj", dict(mjOBJ_UNKNOWN=0, mjOBJ_BODY=1, mjOBJ_XBODY=2, mjOBJ_JOINT=3, mjOBJ_DOF=4, mjOBJ_GEOM=5, mjOBJ_SITE=6, mjOBJ_CAMERA=7))
mjtStage = _enum("mjtStage", dict(mjSTAGE_NONE=0, mjSTAGE_POS=1, mjSTAGE_VEL=2, mjSTAGE_ACC=3))
mjtDataType = _enum("mjtDataType", dict(mjDATATYPE_REAL=0, mjDATATYPE_POSITIVE=1, mjDATATYPE_AXIS=2, mjDATATYPE_QUATERNION=3))


def _empty(shape=(0,), dtype=np.float64):
    return lambda l: np.zeros(shape if not callable(shape) else shape(l), dtype=dtype)


# fields the MJCF-subset compiler does not produce; shapes only matter for being well-formed
_DEFAULTS = {
    "nexclude": lambda l: len(l.exclude_signature),
    "nmesh": lambda l: 0, "ntex": lambda l: 0, "ntexdata": lambda l: 0, "nwrap": lambda l: 0,
    "names": lambda l: b"\x00" * 16,
}
for _n in ("body_sameframe", "geom_sameframe", "site_sameframe"):
    pass
_INT0 = ("mesh_vertadr mesh_vertnum mesh_faceadr mesh_normaladr mesh_normalnum mesh_graphadr mesh_graph mesh_texcoordadr "
         "mesh_texcoordnum hfield_nrow hfield_ncol hfield_adr mat_texid tex_type tex_height tex_width tex_nchannel tex_adr "
         "tex_data tendon_adr tendon_num wrap_type wrap_objid sensor_datatype sensor_needstage sensor_objtype sensor_reftype "
         "sensor_refid name_bodyadr name_jntadr name_geomadr name_siteadr name_camadr name_meshadr name_pairadr name_eqadr "
         "name_tendonadr name_actuatoradr name_sensoradr name_numericadr eq_objtype").split()
_FLT0 = ("mesh_vert mesh_normal mesh_pos mesh_quat mesh_texcoord hfield_data mat_rgba mat_texrepeat mat_texuniform wrap_prm "
         "tendon_solref_lim tendon_solimp_lim tendon_solref_fri tendon_solimp_fri tendon_range tendon_actfrcrange tendon_margin "
         "tendon_stiffness tendon_damping tendon_armature tendon_lengthspring tendon_length0 tendon_invweight0 key_act key_mpos key_mquat").split()
for _n in _INT0:
    _DEFAULTS[_n] = _empty((0,), np.int32)
for _n in _FLT0:
    _DEFAULTS[_n] = _empty((0,), np.float64)
_DEFAULTS["mesh_face"] = _empty((0, 3), np.int32)
_DEFAULTS["mesh_vert"] = _empty((0, 3), np.float64)
_DEFAULTS["hfield_size"] = _empty((0, 4), np.float64)
# one opaque dummy material: ray.precompute_ray_data indexes mat_rgba[geom_matid] even for geom_matid == -1
_DEFAULTS["mat_rgba"] = lambda l: np.ones((1, 4), dtype=np.float32)
_DEFAULTS["tendon_actfrclimited"] = _empty((0,), np.uint8)
_DEFAULTS["sensor_intprm"] = lambda l: np.zeros((l.nsensor, 3), dtype=np.int32)


class MjOption:
    def __init__(self, ns):
        self.__dict__["_ns"] = ns

    def __getattr__(self, name):
        v = getattr(self.__dict__["_ns"], name)
        return int(v) if isinstance(v, enum.IntEnum) else v

    def __setattr__(self, name, value):
        setattr(self.__dict__["_ns"], name, value)


for _n in ("cone", "integrator", "solver"):
    setattr(MjOption, _n, property(lambda self, n=_n: int(getattr(self.__dict__["_ns"], n)), lambda self, v, n=_n: setattr(self.__dict__["_ns"], n, v)))


class MjStatistic:
    def __init__(self, ns):
        self._ns = ns

    def __getattr__(self, name):
        return getattr(self.__dict__["_ns"], name)


class MjModel:
    """Wraps an ``mjcf.MjModelLite``; unknown MuJoCo fields resolve to well-formed empties."""

    def __init__(self, lite):
        self.__dict__["_l"] = lite
        self.__dict__["_opt"] = MjOption(lite.opt)
        self.__dict__["_stat"] = MjStatistic(lite.stat)

    def __getattr__(self, name):
        l = self.__dict__["_l"]
        if name == "stat":
            return self.__dict__["_stat"]
        if hasattr(l, name):
            return getattr(l, name)
        if name in _DEFAULTS:
            return _DEFAULTS[name](l)
        raise AttributeError(name)


def _mk_prop(n):
    return property(lambda self: getattr(self.__dict__["_l"], n))


for _n in ("actuator_biastype", "actuator_dyntype", "eq_type", "actuator_gaintype", "actuator_trntype"):
    setattr(MjModel, _n, _mk_prop(_n))
MjModel.opt = property(lambda self: self.__dict__["_opt"])


class MjData:
    def __init__(self, model):
        self.model = model
        l = model.__dict__["_l"]
        self.qpos = np.array(l.qpos0, dtype=np.float64)
        self.xaxis = np.zeros((l.njnt, 3))
        self.xanchor = np.zeros((l.njnt, 3))


def mj_kinematics(model, data):
    from mujoco_torch_amd import mjcf

    l = model.__dict__["_l"]
    out = mjcf._kinematics0(l, data.qpos)
    data.xanchor, data.xaxis = out[5], out[6]


def mj_normalizeQuat(model, qpos):
    l = model.__dict__["_l"]
    for j in range(l.njnt):
        a = int(l.jnt_qposadr[j])
        if l.jnt_type[j] == 0:
            qpos[a + 3 : a + 7] /= np.linalg.norm(qpos[a + 3 : a + 7])
        elif l.jnt_type[j] == 1:
            qpos[a : a + 4] /= np.linalg.norm(qpos[a : a + 4])


class _MjContactList:
    pass


_structs = _pytypes.ModuleType("mujoco._structs")
_structs._MjContactList = _MjContactList
sys.modules["mujoco._structs"] = _structs
_functions = _pytypes.ModuleType("mujoco._functions")
sys.modules["mujoco._functions"] = _functions
