"""Enum / constant tables for the stepper.

The reference reads these from the installed ``mujoco`` package at import time
(reference ``_src/types.py:33-482``) and pins mujoco 3.5.0 (``uv.lock:287-288``).
``mujoco`` is not a dependency of this package, so the 3.5.0 numbering is
restated here; when ``mujoco`` *is* importable the live values win (same
policy as the reference's version dispatch, ``types.py:56-91``).
"""

import enum

# mjmodel.h numeric constants (reference reads them as mujoco.mjMINVAL etc.)
mjMINVAL = 1e-15
mjMAXVAL = 1e10
mjMINIMP = 1e-4
mjMAXIMP = 0.9999
mjMINMU = 1e-5
mjNREF = 2
mjNIMP = 5
mjNEQDATA = 11
mjNDYN = 10
mjNGAIN = 10
mjNBIAS = 10


class DisableBit(enum.IntFlag):
    """mjtDisableBit, mujoco 3.3.6 .. 3.5 numbering (SPRING/DAMPER split)."""

    CONSTRAINT = 1 << 0
    EQUALITY = 1 << 1
    FRICTIONLOSS = 1 << 2
    LIMIT = 1 << 3
    CONTACT = 1 << 4
    SPRING = 1 << 5
    DAMPER = 1 << 6
    GRAVITY = 1 << 7
    CLAMPCTRL = 1 << 8
    WARMSTART = 1 << 9
    FILTERPARENT = 1 << 10
    ACTUATION = 1 << 11
    REFSAFE = 1 << 12
    SENSOR = 1 << 13
    MIDPHASE = 1 << 14
    EULERDAMP = 1 << 15
    AUTORESET = 1 << 16
    NATIVECCD = 1 << 17
    ISLAND = 1 << 18


class EnableBit(enum.IntFlag):
    OVERRIDE = 1 << 0
    ENERGY = 1 << 1
    FWDINV = 1 << 2
    INVDISCRETE = 1 << 3
    MULTICCD = 1 << 4
    SLEEP = 1 << 5


class JointType(enum.IntEnum):
    FREE = 0
    BALL = 1
    SLIDE = 2
    HINGE = 3

    def dof_width(self) -> int:
        return (6, 3, 1, 1)[self.value]

    def qpos_width(self) -> int:
        return (7, 4, 1, 1)[self.value]


class IntegratorType(enum.IntEnum):
    EULER = 0
    RK4 = 1
    IMPLICIT = 2
    IMPLICITFAST = 3


class GeomType(enum.IntEnum):
    PLANE = 0
    HFIELD = 1
    SPHERE = 2
    CAPSULE = 3
    ELLIPSOID = 4
    CYLINDER = 5
    BOX = 6
    MESH = 7


class ConeType(enum.IntEnum):
    PYRAMIDAL = 0
    ELLIPTIC = 1


class JacobianType(enum.IntEnum):
    DENSE = 0
    SPARSE = 1
    AUTO = 2


class SolverType(enum.IntEnum):
    PGS = 0
    CG = 1
    NEWTON = 2


class EqType(enum.IntEnum):
    CONNECT = 0
    WELD = 1
    JOINT = 2
    TENDON = 3
    FLEX = 4
    DISTANCE = 5


class TrnType(enum.IntEnum):
    JOINT = 0
    JOINTINPARENT = 1
    SLIDERCRANK = 2
    TENDON = 3
    SITE = 4
    BODY = 5


class DynType(enum.IntEnum):
    NONE = 0
    INTEGRATOR = 1
    FILTER = 2
    FILTEREXACT = 3
    MUSCLE = 4
    USER = 5


class GainType(enum.IntEnum):
    FIXED = 0
    AFFINE = 1
    MUSCLE = 2
    USER = 3


class BiasType(enum.IntEnum):
    NONE = 0
    AFFINE = 1
    MUSCLE = 2
    USER = 3


class CamLightType(enum.IntEnum):
    FIXED = 0
    TRACK = 1
    TRACKCOM = 2
    TARGETBODY = 3
    TARGETBODYCOM = 4


class SensorType(enum.IntEnum):
    """mjtSensor (3.5.0), the whole list: the MJCF compiler assigns a type to every sensor element; the stepper evaluates the ones the
    reference's sensor.py does and leaves the slots of the others untouched, as the reference does."""

    TOUCH = 0
    ACCELEROMETER = 1
    VELOCIMETER = 2
    GYRO = 3
    FORCE = 4
    TORQUE = 5
    MAGNETOMETER = 6
    RANGEFINDER = 7
    CAMPROJECTION = 8
    JOINTPOS = 9
    JOINTVEL = 10
    TENDONPOS = 11
    TENDONVEL = 12
    ACTUATORPOS = 13
    ACTUATORVEL = 14
    ACTUATORFRC = 15
    JOINTACTFRC = 16
    TENDONACTFRC = 17
    BALLQUAT = 18
    BALLANGVEL = 19
    JOINTLIMITPOS = 20
    JOINTLIMITVEL = 21
    JOINTLIMITFRC = 22
    TENDONLIMITPOS = 23
    TENDONLIMITVEL = 24
    TENDONLIMITFRC = 25
    FRAMEPOS = 26
    FRAMEQUAT = 27
    FRAMEXAXIS = 28
    FRAMEYAXIS = 29
    FRAMEZAXIS = 30
    FRAMELINVEL = 31
    FRAMEANGVEL = 32
    FRAMELINACC = 33
    FRAMEANGACC = 34
    SUBTREECOM = 35
    SUBTREELINVEL = 36
    SUBTREEANGMOM = 37
    INSIDESITE = 38
    GEOMDIST = 39
    GEOMNORMAL = 40
    GEOMFROMTO = 41
    CONTACT = 42
    E_POTENTIAL = 43
    E_KINETIC = 44
    CLOCK = 45
    TACTILE = 46
    PLUGIN = 47
    USER = 48


class ObjType(enum.IntEnum):
    """mjtObj values the sensors use (reference types.py:465-482)."""

    UNKNOWN = 0
    BODY = 1
    XBODY = 2
    JOINT = 3
    DOF = 4
    GEOM = 5
    SITE = 6
    CAMERA = 7
    ACTUATOR = 19
    TENDON = 18


# Sets the stepper supports (reference device.py:919-949 raises for the rest).
SUPPORTED_INTEGRATORS = (IntegratorType.EULER, IntegratorType.RK4)
SUPPORTED_SOLVERS = (SolverType.CG, SolverType.NEWTON)
SUPPORTED_CONDIM = (1, 3, 4, 6)
