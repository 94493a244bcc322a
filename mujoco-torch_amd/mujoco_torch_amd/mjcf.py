"""MJCF-subset compiler: XML -> arrays with MuJoCo's ``MjModel`` field names.

The reference delegates XML compilation and ``mj_setConst`` to the MuJoCo C
library (``mujoco.MjModel.from_xml_*`` then reference ``_src/device.py:1045-1080``
copies every field by name).  That library is not a dependency here, so the
subset of the compiler the bundled models need is restated in numpy: default
classes, ``fromto``/``euler``/``xyaxes``/``zaxis`` frames, density -> mass and
inertia, principal-axis inertial frames, ``autolimits``, degree/radian
conversion, ``<pair>``/``<exclude>``, motor/position/velocity actuators,
keyframes, binary-STL convex meshes, and the qpos0 constants ``mj_setConst``
derives (``dof_invweight0``, ``body_invweight0``, ``dof_M0``,
``stat.meaninertia``, ``actuator_acc0``, cam/light ``*0`` frames).

Parity of these constants with the real MuJoCo compiler is UNPINNED in this
repository (no ``mujoco`` wheel offline); ``device_put`` accepts a real
``mujoco.MjModel`` unchanged when one is available.  This is host-side,
setup-time code: nothing here runs inside ``step``.
"""

from __future__ import annotations

import math
import os
import re
import struct
import xml.etree.ElementTree as ET
from types import SimpleNamespace

import numpy as np

from ._enums import (
    BiasType,
    CamLightType,
    ConeType,
    DisableBit,
    DynType,
    EnableBit,
    GainType,
    GeomType,
    IntegratorType,
    JacobianType,
    JointType,
    ObjType,
    SensorType,
    SolverType,
    TrnType,
    mjMINVAL,
    mjNBIAS,
    mjNDYN,
    mjNGAIN,
)

# --------------------------------------------------------------------------
# small quaternion / frame helpers (host, float64)
# --------------------------------------------------------------------------


def _quat_mul(u, v):
    return np.array(
        [
            u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3],
            u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2],
            u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1],
            u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0],
        ]
    )


def _quat_to_mat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def _mat_to_quat(m):
    """Rotation matrix -> unit quaternion (w >= 0 branch selection as mju_mat2Quat)."""
    if m[0, 0] + m[1, 1] + m[2, 2] > 0:
        q0 = 0.5 * math.sqrt(1 + m[0, 0] + m[1, 1] + m[2, 2])
        q = np.array([q0, 0.25 * (m[2, 1] - m[1, 2]) / q0, 0.25 * (m[0, 2] - m[2, 0]) / q0, 0.25 * (m[1, 0] - m[0, 1]) / q0])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        q1 = 0.5 * math.sqrt(1 + m[0, 0] - m[1, 1] - m[2, 2])
        q = np.array([0.25 * (m[2, 1] - m[1, 2]) / q1, q1, 0.25 * (m[0, 1] + m[1, 0]) / q1, 0.25 * (m[0, 2] + m[2, 0]) / q1])
    elif m[1, 1] > m[2, 2]:
        q2 = 0.5 * math.sqrt(1 - m[0, 0] + m[1, 1] - m[2, 2])
        q = np.array([0.25 * (m[0, 2] - m[2, 0]) / q2, 0.25 * (m[0, 1] + m[1, 0]) / q2, q2, 0.25 * (m[1, 2] + m[2, 1]) / q2])
    else:
        q3 = 0.5 * math.sqrt(1 - m[0, 0] - m[1, 1] + m[2, 2])
        q = np.array([0.25 * (m[1, 0] - m[0, 1]) / q3, 0.25 * (m[0, 2] + m[2, 0]) / q3, 0.25 * (m[1, 2] + m[2, 1]) / q3, q3])
    return q / np.linalg.norm(q)


def _rotate(v, q):
    return _quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


def _z_to_quat(vec):
    """Quaternion rotating +z onto ``vec`` (mjuu_z2quat)."""
    vec = np.asarray(vec, dtype=np.float64)
    n = np.linalg.norm(vec)
    if n < 1e-10:
        return np.array([1.0, 0, 0, 0])
    vec = vec / n
    z = np.array([0.0, 0, 1])
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0, 0])
    else:
        axis = axis / s
    ang = math.atan2(s, vec[2])
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def _axisangle_quat(axis, ang):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-10:
        return np.array([1.0, 0, 0, 0])
    axis = axis / n
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


# --------------------------------------------------------------------------
# attribute parsing helpers
# --------------------------------------------------------------------------


def _floats(s):
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def _bool(s):
    return s.strip().lower() == "true"


_ACT_TAGS = ("motor", "position", "velocity", "general", "intvelocity", "damper", "cylinder", "muscle")


class _Attrs(dict):
    """Attribute dict that records which keys the compiler looked at.  Every XML element is read through one of these (its own, or a
    copy merged with default classes that forwards to it): after the build, an attribute nobody looked at is an attribute this
    compiler does not honour -- reported instead of silently dropped ("refuse or honour, never ignore")."""

    __slots__ = ("used", "sinks")

    def __init__(self, *args, sinks=(), **kw):
        super().__init__(*args, **kw)
        self.used = set()
        self.sinks = list(sinks)

    def _mark(self, k):
        self.used.add(k)
        for s in self.sinks:
            s._mark(k)

    def __getitem__(self, k):
        self._mark(k)
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        self._mark(k)
        return dict.get(self, k, d)

    def __contains__(self, k):
        self._mark(k)
        return dict.__contains__(self, k)

    def pop(self, k, *d):
        self._mark(k)
        return dict.pop(self, k, *d)

    def mark_all(self):
        for k in list(dict.keys(self)):
            self._mark(k)


# elements that never reach the physics (skipped with everything below them) ...
_COSMETIC_ELEMENTS = {"visual", "size", "texture", "material", "skin", "text", "tuple"}
# ... and attributes that do not either, on any element
_COSMETIC_ATTRS = {"rgba", "material", "group", "user", "class", "childclass", "name", "fovy", "ipd", "resolution", "focal", "focalpixel", "principal", "principalpixel", "sensorsize",
                   "orthographic", "directional", "castshadow", "active", "attenuation", "cutoff_light", "exponent", "ambient", "diffuse", "specular", "bulbradius", "intensity", "range_light",
                   "texcoord", "smoothnormal", "extent", "center", "meansize", "meanmass", "texture"}
# children each element may have (anything else is refused); None = any child is handled by that section's own check
_CHILDREN = {
    "mujoco": {"compiler", "option", "size", "visual", "statistic", "default", "custom", "asset", "worldbody", "contact", "equality", "tendon", "actuator", "sensor", "keyframe"},
    "option": {"flag"}, "custom": {"numeric", "text", "tuple"}, "asset": {"mesh", "texture", "material", "skin"},
    "worldbody": {"body", "geom", "site", "camera", "light", "frame"}, "body": {"body", "geom", "site", "camera", "light", "joint", "freejoint", "inertial", "frame"},
    "frame": {"body", "geom", "site", "camera", "light", "frame"},
    "contact": {"pair", "exclude"}, "equality": {"connect", "weld", "joint"}, "tendon": {"fixed", "spatial"}, "fixed": {"joint"}, "spatial": {"site", "geom", "pulley"},
    "actuator": set(_ACT_TAGS), "keyframe": {"key"}, "default": None, "sensor": None,
}


class _Defaults:
    """One default class: per-tag attribute dicts, inherited from the parent."""

    def __init__(self, parent=None):
        self.attrs = {} if parent is None else {k: dict(v) for k, v in parent.attrs.items()}

    def update(self, tag, d):
        key = "actuator" if tag in _ACT_TAGS else tag
        self.attrs.setdefault(key, {}).update(d)
        if tag in _ACT_TAGS:
            self.attrs[key]["__tag__"] = tag

    def get(self, tag):
        key = "actuator" if tag in _ACT_TAGS else tag
        return self.attrs.get(key, {})


# --------------------------------------------------------------------------
# geometry: mass / inertia of primitives (mjCGeom::GetVolume / SetInertia)
# --------------------------------------------------------------------------


def _geom_volume(gtype, size):
    if gtype == GeomType.SPHERE:
        return 4.0 / 3.0 * math.pi * size[0] ** 3
    if gtype == GeomType.CAPSULE:
        h = 2 * size[1]
        return math.pi * (size[0] ** 2 * h + 4.0 / 3.0 * size[0] ** 3)
    if gtype == GeomType.CYLINDER:
        return math.pi * size[0] ** 2 * 2 * size[1]
    if gtype == GeomType.ELLIPSOID:
        return 4.0 / 3.0 * math.pi * size[0] * size[1] * size[2]
    if gtype == GeomType.BOX:
        return 8.0 * size[0] * size[1] * size[2]
    return 0.0


def _geom_inertia(gtype, size, mass):
    if gtype == GeomType.SPHERE:
        i = 2.0 * mass * size[0] ** 2 / 5.0
        return np.array([i, i, i])
    if gtype == GeomType.CAPSULE:
        r, h = size[0], 2 * size[1]
        sphere_mass = mass * 4 * r / (4 * r + 3 * h)
        cyl_mass = mass - sphere_mass
        i0 = cyl_mass * (3 * r * r + h * h) / 12.0
        i2 = cyl_mass * r * r / 2.0
        sph = 2.0 * sphere_mass * r * r / 5.0
        i0 += sph + sphere_mass * h * (3 * r + 2 * h) / 8.0
        i2 += sph
        return np.array([i0, i0, i2])
    if gtype == GeomType.CYLINDER:
        r, h = size[0], 2 * size[1]
        i0 = mass * (3 * r * r + h * h) / 12.0
        return np.array([i0, i0, mass * r * r / 2.0])
    if gtype == GeomType.ELLIPSOID:
        a, b, c = size
        return mass / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GeomType.BOX:
        a, b, c = size
        return mass / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    return np.zeros(3)


def _principal_axes(full):
    """Symmetric 3x3 -> (quat, diag) with eigenvalues in decreasing order, right-handed frame."""
    w, v = np.linalg.eigh(full)
    order = np.argsort(-w, kind="stable")
    w, v = w[order], v[:, order]
    if np.linalg.det(v) < 0:
        v[:, 2] = -v[:, 2]
    return _mat_to_quat(v), w


def _read_stl(path):
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0]
    tris = np.zeros((ntri, 3, 3), dtype=np.float64)
    off = 84
    for i in range(ntri):
        vals = struct.unpack_from("<12f", data, off)
        tris[i] = np.array(vals[3:12]).reshape(3, 3)
        off += 50
    return tris


def _mesh_mass_props(verts, faces):
    """Volume, COM and unit-density inertia (about COM) of a closed triangle mesh."""
    vol = 0.0
    com = np.zeros(3)
    # covariance integral via signed tetrahedra (origin apex)
    canon = np.array([[2, 1, 1], [1, 2, 1], [1, 1, 2]], dtype=np.float64) / 120.0
    cov = np.zeros((3, 3))
    for f in faces:
        a, b, c = verts[f[0]], verts[f[1]], verts[f[2]]
        A = np.stack([a, b, c], axis=1)
        det = np.linalg.det(A)
        vol += det / 6.0
        com += det / 24.0 * (a + b + c)
        cov += det * (A @ canon @ A.T)
    com = com / vol
    cov = cov - vol * np.outer(com, com)
    inertia = np.trace(cov) * np.eye(3) - cov
    return vol, com, inertia


# --------------------------------------------------------------------------
# the compiler
# --------------------------------------------------------------------------


def _hull_graph(d):
    """mjModel.mesh_graph record of one mesh: numvert, numface, vert_edgeadr[numvert],
    vert_globalid[numvert], edge_localid[numvert + 3 numface] (neighbour lists, -1 terminated),
    face_globalid[3 numface]."""
    vid = np.sort(np.asarray(d["hull_vertices"]))
    local = {int(v): i for i, v in enumerate(vid)}
    simp = np.array(d["hull_simplices"])
    for i, eq in enumerate(np.asarray(d["hull_equations"])):  # outward-facing triangles
        a, b, c = d["vert"][simp[i]]
        if np.dot(np.cross(b - a, c - a), eq[:3]) < 0:
            simp[i] = simp[i][[0, 2, 1]]
    nbr = [[] for _ in vid]
    for t in simp:
        for k in range(3):
            a, b = local[int(t[k])], local[int(t[(k + 1) % 3])]
            if b not in nbr[a]:
                nbr[a].append(b)
            if a not in nbr[b]:
                nbr[b].append(a)
    edgeadr, edges = [], []
    for lst in nbr:
        edgeadr.append(len(edges))
        edges.extend(lst + [-1])
    assert len(edges) == len(vid) + 3 * len(simp)
    faces = np.array([[local[int(v)] for v in t] for t in simp])
    return np.concatenate([[len(vid), len(simp)], edgeadr, vid, edges, vid[faces].reshape(-1)]).astype(np.int32)


class MjModelLite(SimpleNamespace):
    """Attribute bag with MuJoCo ``MjModel`` field names (numpy arrays / ints)."""


_DISABLE_FLAGS = {
    "constraint": DisableBit.CONSTRAINT,
    "equality": DisableBit.EQUALITY,
    "frictionloss": DisableBit.FRICTIONLOSS,
    "limit": DisableBit.LIMIT,
    "contact": DisableBit.CONTACT,
    "spring": DisableBit.SPRING,
    "damper": DisableBit.DAMPER,
    "passive": DisableBit.SPRING | DisableBit.DAMPER,
    "gravity": DisableBit.GRAVITY,
    "clampctrl": DisableBit.CLAMPCTRL,
    "warmstart": DisableBit.WARMSTART,
    "filterparent": DisableBit.FILTERPARENT,
    "actuation": DisableBit.ACTUATION,
    "refsafe": DisableBit.REFSAFE,
    "sensor": DisableBit.SENSOR,
    "midphase": DisableBit.MIDPHASE,
    "eulerdamp": DisableBit.EULERDAMP,
    "autoreset": DisableBit.AUTORESET,
    "nativeccd": DisableBit.NATIVECCD,
    "island": DisableBit.ISLAND,
}
_ENABLE_FLAGS = {
    "override": EnableBit.OVERRIDE,
    "energy": EnableBit.ENERGY,
    "fwdinv": EnableBit.FWDINV,
    "invdiscrete": EnableBit.INVDISCRETE,
    "multiccd": EnableBit.MULTICCD,
    "sleep": EnableBit.SLEEP,
}

_GEOM_TYPES = {
    "plane": GeomType.PLANE,
    "hfield": GeomType.HFIELD,
    "sphere": GeomType.SPHERE,
    "capsule": GeomType.CAPSULE,
    "ellipsoid": GeomType.ELLIPSOID,
    "cylinder": GeomType.CYLINDER,
    "box": GeomType.BOX,
    "mesh": GeomType.MESH,
}


def _geom_type(name, what="geom"):
    """mjtGeom of a type attribute; what MuJoCo has and this build does not (sdf) is refused by name (it surfaced as a bare KeyError)."""
    if name not in _GEOM_TYPES:
        if name == "sdf":
            raise NotImplementedError(f'{what} type="sdf" (signed-distance-field geoms need plugins) is not supported')
        raise ValueError(f"unknown {what} type {name!r}")
    return _GEOM_TYPES[name]


_JOINT_TYPES = {"free": JointType.FREE, "ball": JointType.BALL, "slide": JointType.SLIDE, "hinge": JointType.HINGE}
_CAM_MODES = {
    "fixed": CamLightType.FIXED,
    "track": CamLightType.TRACK,
    "trackcom": CamLightType.TRACKCOM,
    "targetbody": CamLightType.TARGETBODY,
    "targetbodycom": CamLightType.TARGETBODYCOM,
}
# sensor element -> (type, dim, needstage [mjtStage: 1 pos, 2 vel, 3 acc], datatype [mjtDataType: 0 real, 1 positive, 2 axis, 3 quaternion], attachment), as the MuJoCo
# compiler sets them per sensor type (user_objects.cc, mjCSensor::Compile).  attachment: the attribute that names the object ("frame": objtype / objname + reftype / refname)
_SENSOR_DIMS = {
    "touch": (SensorType.TOUCH, 1, 3, 1, "site"),
    "accelerometer": (SensorType.ACCELEROMETER, 3, 3, 0, "site"),
    "velocimeter": (SensorType.VELOCIMETER, 3, 2, 0, "site"),
    "gyro": (SensorType.GYRO, 3, 2, 0, "site"),
    "force": (SensorType.FORCE, 3, 3, 0, "site"),
    "torque": (SensorType.TORQUE, 3, 3, 0, "site"),
    "magnetometer": (SensorType.MAGNETOMETER, 3, 1, 0, "site"),
    "rangefinder": (SensorType.RANGEFINDER, 1, 1, 0, "site"),
    "camprojection": (SensorType.CAMPROJECTION, 2, 1, 0, "site+camera"),  # (the reference's sensor_pos has no branch for it: the slot keeps the caller's value, sensor.py:196)
    "jointpos": (SensorType.JOINTPOS, 1, 1, 0, "joint"),
    "jointvel": (SensorType.JOINTVEL, 1, 2, 0, "joint"),
    "tendonpos": (SensorType.TENDONPOS, 1, 1, 0, "tendon"),
    "tendonvel": (SensorType.TENDONVEL, 1, 2, 0, "tendon"),
    "actuatorpos": (SensorType.ACTUATORPOS, 1, 1, 0, "actuator"),
    "actuatorvel": (SensorType.ACTUATORVEL, 1, 2, 0, "actuator"),
    "actuatorfrc": (SensorType.ACTUATORFRC, 1, 3, 0, "actuator"),
    "jointactuatorfrc": (SensorType.JOINTACTFRC, 1, 3, 0, "joint"),
    "tendonactuatorfrc": (SensorType.TENDONACTFRC, 1, 3, 0, "tendon"),
    "ballquat": (SensorType.BALLQUAT, 4, 1, 3, "joint"),
    "ballangvel": (SensorType.BALLANGVEL, 3, 2, 0, "joint"),
    "jointlimitpos": (SensorType.JOINTLIMITPOS, 1, 1, 0, "joint"),
    "jointlimitvel": (SensorType.JOINTLIMITVEL, 1, 2, 0, "joint"),
    "jointlimitfrc": (SensorType.JOINTLIMITFRC, 1, 3, 1, "joint"),
    "tendonlimitpos": (SensorType.TENDONLIMITPOS, 1, 1, 0, "tendon"),
    "tendonlimitvel": (SensorType.TENDONLIMITVEL, 1, 2, 0, "tendon"),
    "tendonlimitfrc": (SensorType.TENDONLIMITFRC, 1, 3, 1, "tendon"),
    "framepos": (SensorType.FRAMEPOS, 3, 1, 0, "frame"),
    "framequat": (SensorType.FRAMEQUAT, 4, 1, 3, "frame"),
    "framexaxis": (SensorType.FRAMEXAXIS, 3, 1, 2, "frame"),
    "frameyaxis": (SensorType.FRAMEYAXIS, 3, 1, 2, "frame"),
    "framezaxis": (SensorType.FRAMEZAXIS, 3, 1, 2, "frame"),
    "framelinvel": (SensorType.FRAMELINVEL, 3, 2, 0, "frame"),
    "frameangvel": (SensorType.FRAMEANGVEL, 3, 2, 0, "frame"),
    "framelinacc": (SensorType.FRAMELINACC, 3, 3, 0, "frame"),
    "frameangacc": (SensorType.FRAMEANGACC, 3, 3, 0, "frame"),
    "subtreecom": (SensorType.SUBTREECOM, 3, 1, 0, "body"),
    "subtreelinvel": (SensorType.SUBTREELINVEL, 3, 2, 0, "body"),
    "subtreeangmom": (SensorType.SUBTREEANGMOM, 3, 2, 0, "body"),
    "e_potential": (SensorType.E_POTENTIAL, 1, 1, 0, None),
    "e_kinetic": (SensorType.E_KINETIC, 1, 2, 0, None),
    "clock": (SensorType.CLOCK, 1, 1, 0, None),
    "user": (SensorType.USER, None, None, 0, "user"),  # dim / needstage / datatype / object from the element; evaluated by a user callback in MuJoCo, by nobody in the reference (slot untouched)
}
_STAGE_NAMES = {"pos": 1, "vel": 2, "acc": 3}
_DATATYPE_NAMES = {"real": 0, "positive": 1, "axis": 2, "quaternion": 3}
_USER_OBJTYPES = {"body": ObjType.BODY, "xbody": ObjType.XBODY, "joint": ObjType.JOINT, "geom": ObjType.GEOM, "site": ObjType.SITE, "camera": ObjType.CAMERA,
                  "tendon": ObjType.TENDON, "actuator": ObjType.ACTUATOR}
_OBJTYPE_NAMES = {"body": ObjType.BODY, "xbody": ObjType.XBODY, "geom": ObjType.GEOM, "site": ObjType.SITE, "camera": ObjType.CAMERA}

_DEF_SOLREF = np.array([0.02, 1.0])
_DEF_SOLIMP = np.array([0.9, 0.95, 0.001, 0.5, 2.0])


def _pad(a, n, fill):
    a = np.asarray(a, dtype=np.float64)
    out = np.array(fill, dtype=np.float64).copy()
    out[: len(a)] = a[:n]
    return out


class _Compiler:
    def __init__(self, root, base_dir):
        self.root = root
        self.base_dir = base_dir
        self.angle_deg = True
        self.autolimits = True
        self.eulerseq = "xyz"
        self.meshdir = ""
        self.inertiafromgeom = "auto"
        self.boundmass = 0.0
        self.boundinertia = 0.0
        self.settotalmass = -1.0
        self.defaults = {"main": _Defaults()}
        self.meshes = {}
        self.bodies = []  # dicts
        self.joints = []
        self.geoms = []
        self.sites = []
        self.cams = []
        self.lights = []
        self.opt = {}
        self.stat_meaninertia = None
        self._trackers = {}   # id(element) -> (element, _Attrs of its own attributes)
        self._consumed = {}   # tag -> attribute names looked at on ANY element of that tag (validates the default classes)

    def _t(self, node):
        """The tracked attribute dict of an element (one per element)."""
        hit = self._trackers.get(id(node))
        if hit is None:
            hit = (node, _Attrs(node.attrib))
            self._trackers[id(node)] = hit
        return hit[1]

    def _merged(self, node, base=None):
        """A working copy of an element's attributes on top of `base` (default-class values): lookups are recorded on the element."""
        a = _Attrs(base or {}, sinks=[self._t(node)])
        a.update(node.attrib)
        return a

    # ---- orientation -------------------------------------------------
    def _ang(self, x):
        return x * math.pi / 180.0 if self.angle_deg else x

    def _orientation(self, a):
        if "quat" in a:
            q = _floats(a["quat"])
            return q / np.linalg.norm(q)
        if "axisangle" in a:
            v = _floats(a["axisangle"])
            return _axisangle_quat(v[:3], self._ang(v[3]))
        if "euler" in a:
            e = _floats(a["euler"])
            q = np.array([1.0, 0, 0, 0])
            for i, ch in enumerate(self.eulerseq):
                ax = {"x": [1.0, 0, 0], "y": [0, 1.0, 0], "z": [0, 0, 1.0]}[ch.lower()]
                r = _axisangle_quat(ax, self._ang(e[i]))
                # lower-case: intrinsic (rotating frame) -> post-multiply
                q = _quat_mul(q, r) if ch.islower() else _quat_mul(r, q)
            return q / np.linalg.norm(q)
        if "xyaxes" in a:
            v = _floats(a["xyaxes"])
            x = v[:3] / np.linalg.norm(v[:3])
            y = v[3:] - x * np.dot(x, v[3:])
            y = y / np.linalg.norm(y)
            z = np.cross(x, y)
            return _mat_to_quat(np.stack([x, y, z], axis=1))
        if "zaxis" in a:
            return _z_to_quat(_floats(a["zaxis"]))
        return np.array([1.0, 0, 0, 0])

    # ---- defaults ------------------------------------------------------
    def _parse_defaults(self, node, parent_name):
        name = node.get("class", "main" if parent_name is None else None)
        if parent_name is None:
            d = self.defaults["main"]
            name = "main"
        else:
            d = _Defaults(self.defaults[parent_name])
            self.defaults[name] = d
        for child in node:
            if child.tag == "default":
                continue
            d.update(child.tag, dict(child.attrib))
        for child in node:
            if child.tag == "default":
                self._parse_defaults(child, name)

    def _resolve(self, node, childclass):
        cls = node.get("class", childclass or "main")
        if cls not in self.defaults:
            raise ValueError(f"unknown default class {cls!r}")
        self._t(node).get("class")
        base = dict(self.defaults[cls].get(node.tag))
        base.pop("__tag__", None)
        own = dict(node.attrib)
        # an element's own orientation / fromto spec replaces (not merges with) a defaulted one
        orient = ("quat", "axisangle", "euler", "xyaxes", "zaxis")
        if any(k in own for k in orient) or "fromto" in own:
            for k in orient:
                base.pop(k, None)
        return self._merged(node, base)

    # ---- top-level sections -------------------------------------------
    def parse(self):
        root = self.root
        for comp in root.findall("compiler"):
            c = self._t(comp)
            if "angle" in c:
                self.angle_deg = c["angle"] == "degree"
            if "autolimits" in c:
                self.autolimits = _bool(c["autolimits"])
            if "eulerseq" in c:
                self.eulerseq = c["eulerseq"]
            if "meshdir" in c:
                self.meshdir = c["meshdir"]
            if "inertiafromgeom" in c:
                self.inertiafromgeom = c["inertiafromgeom"]
            if "boundmass" in c:
                self.boundmass = float(c["boundmass"])
            if "boundinertia" in c:
                self.boundinertia = float(c["boundinertia"])
            if "settotalmass" in c:
                self.settotalmass = float(c["settotalmass"])
            if c.get("coordinate", "local") != "local":
                raise NotImplementedError("only coordinate='local' is supported")
        for dnode in root.findall("default"):
            self._parse_defaults(dnode, None)
        for st in root.findall("statistic"):
            if "meaninertia" in self._t(st):
                self.stat_meaninertia = float(self._t(st).get("meaninertia"))
        for asset in root.findall("asset"):
            for mesh in asset.findall("mesh"):
                a = self._merged(mesh, dict(self.defaults["main"].get("mesh")))
                name = a.get("name") or os.path.splitext(os.path.basename(a["file"]))[0]
                self.meshes[name] = a
        self._parse_option()
        wb = root.find("worldbody")
        world = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), inertial=None, mocap=False, gravcomp=0.0, id=0, joints=[], geoms=[])
        self.bodies.append(world)
        if wb is not None:
            self._parse_body_children(wb, 0, None)

    def _parse_option(self):
        o = dict(
            timestep=0.002,
            gravity=np.array([0.0, 0, -9.81]),
            wind=np.zeros(3),
            magnetic=np.array([0.0, -0.5, 0.0]),
            density=0.0,
            viscosity=0.0,
            impratio=1.0,
            tolerance=1e-8,
            ls_tolerance=0.01,
            iterations=100,
            ls_iterations=50,
            integrator=IntegratorType.EULER,
            solver=SolverType.NEWTON,
            cone=ConeType.PYRAMIDAL,
            jacobian=JacobianType.AUTO,
            disableflags=0,
            enableflags=0,
            o_margin=0.0,
            o_solref=_DEF_SOLREF.copy(),
            o_solimp=_DEF_SOLIMP.copy(),
            o_friction=np.array([1.0, 1.0, 0.005, 0.0001, 0.0001]),
            disableactuator=0,
            sdf_initpoints=40,
        )
        for on in self.root.findall("option"):
            a = self._t(on)
            for k in ("timestep", "density", "viscosity", "impratio", "tolerance", "ls_tolerance", "o_margin"):
                if k in a:
                    o[k] = float(a[k])
            for k in ("iterations", "ls_iterations", "sdf_initpoints"):
                if k in a:
                    o[k] = int(a[k])
            for k in ("gravity", "wind", "magnetic", "o_solref", "o_solimp", "o_friction"):
                if k in a:
                    o[k] = _floats(a[k])
            if "integrator" in a:
                o["integrator"] = {"euler": IntegratorType.EULER, "rk4": IntegratorType.RK4, "implicit": IntegratorType.IMPLICIT, "implicitfast": IntegratorType.IMPLICITFAST}[a["integrator"].lower()]
            if "solver" in a:
                o["solver"] = {"pgs": SolverType.PGS, "cg": SolverType.CG, "newton": SolverType.NEWTON}[a["solver"].lower()]
            if "cone" in a:
                o["cone"] = {"pyramidal": ConeType.PYRAMIDAL, "elliptic": ConeType.ELLIPTIC}[a["cone"].lower()]
            if "jacobian" in a:
                o["jacobian"] = {"dense": JacobianType.DENSE, "sparse": JacobianType.SPARSE, "auto": JacobianType.AUTO}[a["jacobian"].lower()]
            for fl in on.findall("flag"):
                for k, v in fl.attrib.items():
                    if k in _DISABLE_FLAGS or k in _ENABLE_FLAGS:
                        self._t(fl).get(k)
                    if k in _DISABLE_FLAGS:
                        if v == "disable":
                            o["disableflags"] |= int(_DISABLE_FLAGS[k])
                        else:
                            o["disableflags"] &= ~int(_DISABLE_FLAGS[k])
                    elif k in _ENABLE_FLAGS:
                        if v == "enable":
                            o["enableflags"] |= int(_ENABLE_FLAGS[k])
                        else:
                            o["enableflags"] &= ~int(_ENABLE_FLAGS[k])
        self.opt = o

    # ---- kinematic tree -------------------------------------------------
    def _in_frame(self, a, frame):
        """Pose of an element declared inside <frame> elements, re-expressed in the enclosing body: p' = fp + R(fq) p, q' = fq * q (both endpoints of a fromto)."""
        if frame is None:
            return a
        fp, fq = frame
        if "fromto" in a:
            ft = _floats(a["fromto"])
            p0, p1 = fp + _rotate(ft[:3], fq), fp + _rotate(ft[3:], fq)
            a["fromto"] = " ".join(repr(float(x)) for x in np.concatenate([p0, p1]))
            return a
        q = self._orientation(a)
        pos = _floats(a["pos"]) if "pos" in a else np.zeros(3)
        for k in ("quat", "axisangle", "euler", "xyaxes", "zaxis"):
            a.pop(k, None)
        a["pos"] = " ".join(repr(float(x)) for x in fp + _rotate(pos, fq))
        a["quat"] = " ".join(repr(float(x)) for x in _quat_mul(fq, q))
        return a

    def _parse_body_children(self, node, body_id, childclass, frame=None, pending=None):
        """One pass over the children IN DOCUMENT ORDER, as MuJoCo expands them: a <frame> is visited in place (its geoms / sites / cameras / lights take their
        ids between those of the body's direct children around it) and the child bodies -- direct ones and those nested in frames -- are created in the order
        they appear (frames used to be visited in a second loop, so a model interleaving frames with direct children got other ids than the
        MuJoCo-compiled model the reference's device_put consumes)."""
        body = self.bodies[body_id]
        top = pending is None
        if top:
            pending = []  # (element, attributes, childclass, enclosing frame) of the child bodies, in document order
        for child in node:
            tag = child.tag
            if tag == "inertial":
                if frame is not None:
                    raise ValueError("<inertial> belongs to a body, not to a <frame>")
                body["inertial"] = self._merged(child)
            elif tag in ("joint", "freejoint"):
                if frame is not None:
                    raise ValueError("<joint> belongs to a body, not to a <frame>")
                if tag == "freejoint":
                    a = self._merged(child)
                    a["type"] = "free"
                else:
                    a = self._resolve(child, childclass)
                a["__body__"] = body_id
                body["joints"].append(len(self.joints))
                self.joints.append(a)
            elif tag == "geom":
                a = self._in_frame(self._resolve(child, childclass), frame)
                a["__body__"] = body_id
                body["geoms"].append(len(self.geoms))
                self.geoms.append(a)
            elif tag == "site":
                a = self._in_frame(self._resolve(child, childclass), frame)
                a["__body__"] = body_id
                self.sites.append(a)
            elif tag == "camera":
                a = self._in_frame(self._resolve(child, childclass), frame)
                a["__body__"] = body_id
                self.cams.append(a)
            elif tag == "light":
                a = self._resolve(child, childclass)
                if frame is not None:  # (a light has a position and a direction, no orientation)
                    fp, fq = frame
                    a["pos"] = " ".join(repr(float(x)) for x in fp + _rotate(_floats(a["pos"]) if "pos" in a else np.zeros(3), fq))
                    a["dir"] = " ".join(repr(float(x)) for x in _rotate(_floats(a["dir"]) if "dir" in a else np.array([0.0, 0.0, -1.0]), fq))
                a["__body__"] = body_id
                self.lights.append(a)
            elif tag == "frame":  # <frame>: a pure coordinate transformation of what it contains (no body, no dofs); frames nest
                a = self._t(child)
                fpos = _floats(a["pos"]) if "pos" in a else np.zeros(3)
                fquat = self._orientation(a)
                if frame is not None:
                    fpos, fquat = frame[0] + _rotate(fpos, frame[1]), _quat_mul(frame[1], fquat)
                self._parse_body_children(child, body_id, a.get("childclass", childclass), (fpos, fquat), pending)
            elif tag == "body":
                a = self._t(child)
                pending.append((child, a, a.get("childclass", childclass), frame))
        if not top:
            return
        for child, a, cc, fr in pending:
            bpos = _floats(a["pos"]) if "pos" in a else np.zeros(3)
            bquat = self._orientation(a)
            if fr is not None:
                bpos, bquat = fr[0] + _rotate(bpos, fr[1]), _quat_mul(fr[1], bquat)
            nb = dict(
                name=a.get("name", ""),
                parent=body_id,
                pos=bpos,
                quat=bquat,
                inertial=None,
                mocap=_bool(a.get("mocap", "false")),
                gravcomp=float(a.get("gravcomp", 0.0)),
                id=len(self.bodies),
                joints=[],
                geoms=[],
            )
            self.bodies.append(nb)
            self._parse_body_children(child, nb["id"], cc)

    # ---- build ---------------------------------------------------------
    def build(self) -> MjModelLite:
        self.parse()
        m = MjModelLite()
        m.opt = SimpleNamespace(**self.opt)
        # MuJoCo numbers bodies depth-first in XML order, which is the parse order
        # of _parse_body_children only if children are visited right after their
        # parent's elements: re-number with an explicit pre-order walk.
        order = []
        kids = {}
        for b in self.bodies:
            if b["id"] != 0:
                kids.setdefault(b["parent"], []).append(b["id"])

        def walk(i):
            order.append(i)
            for k in kids.get(i, []):
                walk(k)

        walk(0)
        remap = {old: new for new, old in enumerate(order)}
        bodies = [self.bodies[i] for i in order]
        nbody = len(bodies)
        m.nbody = nbody
        m.names_body = [b["name"] for b in bodies]
        m.body_parentid = np.array([remap[b["parent"]] for b in bodies], dtype=np.int32)
        m.body_pos = np.stack([b["pos"] for b in bodies])
        m.body_quat = np.stack([b["quat"] for b in bodies])
        m.body_gravcomp = np.array([b["gravcomp"] for b in bodies])
        mocap = [b["mocap"] for b in bodies]
        m.nmocap = int(sum(mocap))
        mid = -np.ones(nbody, dtype=np.int32)
        k = 0
        for i, mc in enumerate(mocap):
            if mc:
                mid[i] = k
                k += 1
        m.body_mocapid = mid

        # ---- joints / dofs (in body order) ----
        jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid = [], [], [], []
        jnt_pos, jnt_axis, jnt_range, jnt_limited, jnt_margin = [], [], [], [], []
        jnt_stiffness, jnt_solref, jnt_solimp, jnt_actfrcrange, jnt_actfrclimited, jnt_actgravcomp = [], [], [], [], [], []
        jnt_names = []
        jnt_springdamper = []
        dof_bodyid, dof_jntid, dof_parentid = [], [], []
        dof_armature, dof_damping, dof_frictionloss, dof_solref, dof_solimp = [], [], [], [], []
        qpos0, qpos_spring = [], []
        body_jntnum = np.zeros(nbody, dtype=np.int32)
        body_jntadr = -np.ones(nbody, dtype=np.int32)
        body_dofnum = np.zeros(nbody, dtype=np.int32)
        body_dofadr = -np.ones(nbody, dtype=np.int32)
        last_dof_of_body = -np.ones(nbody, dtype=np.int32)
        nq = nv = 0
        for bi, b in enumerate(bodies):
            # last dof of the nearest ancestor that has dofs
            p = int(m.body_parentid[bi])
            parent_dof = last_dof_of_body[p] if bi > 0 else -1
            cur_parent = parent_dof
            for jl in b["joints"]:
                a = self.joints[jl]
                jt = _JOINT_TYPES[a.get("type", "hinge")]
                jid = len(jnt_type)
                if body_jntadr[bi] < 0:
                    body_jntadr[bi] = jid
                body_jntnum[bi] += 1
                jnt_names.append(a.get("name", ""))
                jnt_type.append(int(jt))
                jnt_qposadr.append(nq)
                jnt_dofadr.append(nv)
                jnt_bodyid.append(bi)
                if jt == JointType.FREE:
                    a.get("pos")  # (a free joint sits at its body's origin whatever the attribute says: MuJoCo ignores it too)
                    jnt_pos.append(np.zeros(3))
                else:
                    jnt_pos.append(_floats(a["pos"]) if "pos" in a else np.zeros(3))
                ax = _floats(a["axis"]) if "axis" in a else np.array([0.0, 0, 1])
                n = np.linalg.norm(ax)
                jnt_axis.append(ax / n if n > 0 else np.array([0.0, 0, 1]))
                has_range = "range" in a
                rng = _floats(a["range"]) if has_range else np.zeros(2)
                if jt in (JointType.HINGE, JointType.BALL):
                    rng = np.array([self._ang(rng[0]), self._ang(rng[1])])
                lim = a.get("limited", "auto")
                if lim == "auto":
                    limited = has_range and self.autolimits
                else:
                    limited = _bool(lim)
                jnt_range.append(rng)
                jnt_limited.append(limited)
                jnt_margin.append(float(a.get("margin", 0.0)))
                jnt_stiffness.append(float(a.get("stiffness", 0.0)))
                jnt_springdamper.append(_floats(a["springdamper"]) if "springdamper" in a else None)
                jnt_solref.append(_pad(_floats(a["solreflimit"]), 2, _DEF_SOLREF) if "solreflimit" in a else _DEF_SOLREF.copy())
                jnt_solimp.append(_pad(_floats(a["solimplimit"]), 5, _DEF_SOLIMP) if "solimplimit" in a else _DEF_SOLIMP.copy())
                has_afr = "actuatorfrcrange" in a
                jnt_actfrcrange.append(_floats(a["actuatorfrcrange"]) if has_afr else np.zeros(2))
                afl = a.get("actuatorfrclimited", "auto")
                jnt_actfrclimited.append((has_afr and self.autolimits) if afl == "auto" else _bool(afl))
                jnt_actgravcomp.append(_bool(a.get("actuatorgravcomp", "false")))
                ref = float(a.get("ref", 0.0))
                sref = float(a.get("springref", 0.0))
                if jt == JointType.HINGE:
                    ref, sref = self._ang(ref), self._ang(sref)
                if jt == JointType.FREE:
                    q0 = np.concatenate([b["pos"], b["quat"]])
                    qpos0.extend(q0)
                    qpos_spring.extend(q0)
                elif jt == JointType.BALL:
                    qpos0.extend([1.0, 0, 0, 0])
                    qpos_spring.extend([1.0, 0, 0, 0])
                else:
                    qpos0.append(ref)
                    qpos_spring.append(sref)
                w = jt.dof_width()
                if body_dofadr[bi] < 0:
                    body_dofadr[bi] = nv
                body_dofnum[bi] += w
                dsolref = _pad(_floats(a["solreffriction"]), 2, _DEF_SOLREF) if "solreffriction" in a else _DEF_SOLREF.copy()
                dsolimp = _pad(_floats(a["solimpfriction"]), 5, _DEF_SOLIMP) if "solimpfriction" in a else _DEF_SOLIMP.copy()
                for _ in range(w):
                    dof_bodyid.append(bi)
                    dof_jntid.append(jid)
                    dof_parentid.append(cur_parent)
                    cur_parent = nv
                    dof_armature.append(float(a.get("armature", 0.0)))
                    dof_damping.append(float(a.get("damping", 0.0)))
                    dof_frictionloss.append(float(a.get("frictionloss", 0.0)))
                    dof_solref.append(dsolref)
                    dof_solimp.append(dsolimp)
                    nv += 1
                nq += jt.qpos_width()
            last_dof_of_body[bi] = cur_parent
        m.nq, m.nv, m.njnt = nq, nv, len(jnt_type)
        m.names_jnt = jnt_names
        self._jnt_springdamper = jnt_springdamper
        m.jnt_type = np.array(jnt_type, dtype=np.int32)
        m.jnt_qposadr = np.array(jnt_qposadr, dtype=np.int32)
        m.jnt_dofadr = np.array(jnt_dofadr, dtype=np.int32)
        m.jnt_bodyid = np.array(jnt_bodyid, dtype=np.int32)
        m.jnt_group = np.zeros(m.njnt, dtype=np.int32)
        m.jnt_pos = np.array(jnt_pos, dtype=np.float64).reshape(-1, 3)
        m.jnt_axis = np.array(jnt_axis, dtype=np.float64).reshape(-1, 3)
        m.jnt_range = np.array(jnt_range, dtype=np.float64).reshape(-1, 2)
        m.jnt_limited = np.array(jnt_limited, dtype=bool)
        m.jnt_margin = np.array(jnt_margin, dtype=np.float64)
        m.jnt_stiffness = np.array(jnt_stiffness, dtype=np.float64)
        m.jnt_solref = np.array(jnt_solref, dtype=np.float64).reshape(-1, 2)
        m.jnt_solimp = np.array(jnt_solimp, dtype=np.float64).reshape(-1, 5)
        m.jnt_actfrcrange = np.array(jnt_actfrcrange, dtype=np.float64).reshape(-1, 2)
        m.jnt_actfrclimited = np.array(jnt_actfrclimited, dtype=bool)
        m.jnt_actgravcomp = np.array(jnt_actgravcomp, dtype=np.uint8)
        m.dof_bodyid = np.array(dof_bodyid, dtype=np.int32)
        m.dof_jntid = np.array(dof_jntid, dtype=np.int32)
        m.dof_parentid = np.array(dof_parentid, dtype=np.int32)
        m.dof_armature = np.array(dof_armature, dtype=np.float64)
        m.dof_damping = np.array(dof_damping, dtype=np.float64)
        m.dof_frictionloss = np.array(dof_frictionloss, dtype=np.float64)
        m.dof_solref = np.array(dof_solref, dtype=np.float64).reshape(-1, 2)
        m.dof_solimp = np.array(dof_solimp, dtype=np.float64).reshape(-1, 5)
        m.qpos0 = np.array(qpos0, dtype=np.float64)
        m.qpos_spring = np.array(qpos_spring, dtype=np.float64)
        m.body_jntnum, m.body_jntadr = body_jntnum, body_jntadr
        m.body_dofnum, m.body_dofadr = body_dofnum, body_dofadr
        # sparse-M addressing (dof_Madr, nM)
        madr, nM = [], 0
        for i in range(nv):
            madr.append(nM)
            j = i
            while j >= 0:
                nM += 1
                j = int(m.dof_parentid[j])
        m.dof_Madr = np.array(madr + [nM], dtype=np.int32)[:nv] if nv else np.zeros(0, dtype=np.int32)
        m.dof_Madr_ext = np.array(madr + [nM], dtype=np.int32)
        m.nM = nM
        # weld / root / tree ids
        weld = np.zeros(nbody, dtype=np.int32)
        rootid = np.zeros(nbody, dtype=np.int32)
        for bi in range(1, nbody):
            p = int(m.body_parentid[bi])
            weld[bi] = bi if body_jntnum[bi] > 0 else weld[p]
            rootid[bi] = bi if p == 0 else rootid[p]
        m.body_weldid, m.body_rootid = weld, rootid
        treeid = -np.ones(nbody, dtype=np.int32)
        ntree = 0
        for bi in range(1, nbody):
            p = int(m.body_parentid[bi])
            if body_dofnum[bi] > 0 and treeid[p] < 0 and weld[p] == 0:
                treeid[bi] = ntree
                ntree += 1
            else:
                treeid[bi] = treeid[p]
        m.body_treeid = treeid
        m.dof_treeid = treeid[m.dof_bodyid] if nv else np.zeros(0, dtype=np.int32)
        m.body_sameframe = np.zeros(nbody, dtype=np.uint8)
        m.body_simple = np.zeros(nbody, dtype=np.uint8)
        m.dof_simplenum = np.zeros(nv, dtype=np.int32)

        self._build_meshes(m)
        self._build_geoms(m, bodies, remap)
        self._build_inertia(m, bodies)
        self._build_sites_cams_lights(m, remap)
        self._build_contact(m)
        self._build_tendons(m)
        self._build_actuators(m)
        self._build_sensors(m)
        self._build_equality(m)
        self._build_empty_sections(m)
        self._build_keyframes(m)
        _set_const(m, self.stat_meaninertia)
        _equality_set0(m)
        self._auto_spring_damper(m)
        self._validate_consumed()
        return m

    def _validate_consumed(self):
        """Refuse or honour, never ignore: every element of the document must be one this compiler handles, and every attribute must have been looked at while
        it did (cosmetic ones -- colours, materials, camera intrinsics, light properties, rendering groups -- excepted).  MuJoCo's own compiler rejects unknown
        elements / attributes through its schema; what it accepts and this subset does not model (composite, flex, plugins, hfields, ...) raises here."""
        root = self.root

        def walk(node, path):
            allowed = _CHILDREN.get(node.tag, set())
            for child in node:
                if child.tag in _COSMETIC_ELEMENTS:
                    continue
                if allowed is not None and child.tag not in allowed:
                    raise NotImplementedError(f"MJCF element <{child.tag}> inside <{node.tag}> is not supported by this compiler ({path})")
                if node.tag == "default" and child.tag != "default":
                    continue  # (default-class attributes: checked against what the elements of that tag consumed, below)
                if child.tag in ("default",):
                    walk(child, path + "/default")
                    continue
                trk = self._trackers.get(id(child))
                used = trk[1].used if trk is not None else set()
                self._consumed.setdefault(child.tag, set()).update(used)
                extra = [k for k in child.attrib if k not in used and k not in _COSMETIC_ATTRS]
                if child.tag in ("light",):
                    extra = [k for k in extra if k in ("mode", "target")]  # (lights never reach the physics: only their tracking modes are read)
                if child.tag in ("statistic",):
                    extra = []
                if extra:
                    raise NotImplementedError(f"MJCF attribute(s) {extra} of <{child.tag} name={child.get('name', '')!r}> are not honoured by this compiler ({path}/{child.tag})")
                walk(child, path + "/" + child.tag)

        walk(root, "mujoco")
        # default classes: an attribute given for a tag must be one the compiler reads on elements of that tag (when the model has any)
        def walk_defaults(node):
            for child in node:
                if child.tag == "default":
                    walk_defaults(child)
                    continue
                key = "actuator" if child.tag in _ACT_TAGS else child.tag
                seen = set()
                for t, u in self._consumed.items():
                    if ("actuator" if t in _ACT_TAGS else t) == key:
                        seen |= u
                if not seen:
                    continue
                extra = [k for k in child.attrib if k not in seen and k not in _COSMETIC_ATTRS]
                if extra and child.tag != "light":
                    raise NotImplementedError(f"MJCF default attribute(s) {extra} of <{child.tag}> are not honoured by this compiler")
        for dn in root.findall("default"):
            walk_defaults(dn)

    def _auto_spring_damper(self, m):
        """joint springdamper="timeconst dampratio": stiffness and damping of the 1-dof mass-spring-damper with the joint's
        reference-pose inertia 1 / dof_invweight0 (MuJoCo compiler, mjCModel::AutoSpringDamper)."""
        for j, sd in enumerate(self._jnt_springdamper):
            if sd is None or sd[0] <= 0 or sd[1] <= 0:
                continue
            adr, w = int(m.jnt_dofadr[j]), JointType(int(m.jnt_type[j])).dof_width()
            inertia = w / max(mjMINVAL, float(m.dof_invweight0[adr : adr + w].sum()))
            m.jnt_stiffness[j] = inertia / max(mjMINVAL, sd[0] * sd[0] * sd[1] * sd[1])
            m.dof_damping[adr : adr + w] = 2 * inertia / max(mjMINVAL, sd[0])

    # ---- meshes -----------------------------------------------------------
    def _build_meshes(self, m):
        """Convex meshes: STL -> scaled verts -> hull -> centred on COM in the principal frame."""
        m.mesh_names = list(self.meshes.keys())
        m.nmesh = len(self.meshes)
        m._mesh_data = {}
        for name, a in self.meshes.items():
            from scipy.spatial import ConvexHull  # host-side, setup only

            path = os.path.join(self.base_dir, self.meshdir, a["file"])
            tris = _read_stl(path)
            verts = tris.reshape(-1, 3).astype(np.float32).astype(np.float64)
            verts, inv = np.unique(verts, axis=0, return_inverse=True)
            faces = inv.reshape(-1, 3)
            scale = _floats(a["scale"]) if "scale" in a else np.ones(3)
            verts = verts * scale
            vol, com, inertia = _mesh_mass_props(verts, faces)
            if vol < 0:
                faces = faces[:, ::-1]
                vol, com, inertia = _mesh_mass_props(verts, faces)
            quat, diag = _principal_axes(inertia)
            R = _quat_to_mat(quat)
            local = (verts - com) @ R  # coordinates in the principal frame
            hull = ConvexHull(local)
            local = local.astype(np.float32).astype(np.float64)  # MuJoCo keeps mesh_vert in float32
            hull = ConvexHull(local)
            m._mesh_data[name] = dict(vert=local, face=faces, hull_vertices=hull.vertices, hull_simplices=hull.simplices, hull_equations=hull.equations, pos=com, quat=quat, volume=vol, inertia_unit=diag)
        # MuJoCo-layout flat arrays (mjModel.mesh_*)
        D = [m._mesh_data[n] for n in m.mesh_names]
        cat = lambda xs, shape, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(shape, dtype=dt)
        adr = lambda ns: np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(np.int32) if ns else np.zeros(0, dtype=np.int32)
        m.mesh_vertnum = np.array([len(d["vert"]) for d in D], dtype=np.int32)
        m.mesh_facenum = np.array([len(d["face"]) for d in D], dtype=np.int32)
        m.mesh_vertadr = adr([len(d["vert"]) for d in D])
        m.mesh_faceadr = adr([len(d["face"]) for d in D])
        m.mesh_vert = cat([d["vert"] for d in D], (0, 3), np.float32)
        m.mesh_face = cat([d["face"] for d in D], (0, 3), np.int32)
        m.mesh_pos = cat([d["pos"].reshape(1, 3) for d in D], (0, 3), np.float64)
        m.mesh_quat = cat([d["quat"].reshape(1, 4) for d in D], (0, 4), np.float64)
        graphs = [_hull_graph(d) for d in D]
        m.mesh_graphadr = adr([len(g) for g in graphs])
        m.mesh_graph = cat(graphs, (0,), np.int32)

    # ---- geoms -----------------------------------------------------------
    def _build_geoms(self, m, bodies, remap):
        ngeom = len(self.geoms)
        # geoms are numbered by body order, then XML order within the body
        order = []
        for b in bodies:
            order.extend(b["geoms"])
        gremap = {old: new for new, old in enumerate(order)}
        self._geom_remap = gremap
        G = [self.geoms[i] for i in order]
        m.ngeom = ngeom
        m.names_geom = [g.get("name", "") for g in G]
        gt, gsize, gpos, gquat, gbody = [], [], [], [], []
        gmass, ginertia = [], []
        m.geom_dataid = -np.ones(ngeom, dtype=np.int32)
        f = lambda g, k, d: float(g.get(k, d))
        for gi, g in enumerate(G):
            t = _geom_type(g.get("type", "sphere"))
            size = np.zeros(3)
            if "size" in g:
                s = _floats(g["size"])
                size[: len(s)] = s[:3]
            pos = _floats(g["pos"]) if "pos" in g else np.zeros(3)
            quat = self._orientation(g)
            if "fromto" in g:
                ft = _floats(g["fromto"])
                vec = ft[:3] - ft[3:]
                size[1] = np.linalg.norm(vec) / 2.0
                pos = (ft[:3] + ft[3:]) / 2.0
                quat = _z_to_quat(vec)
            mesh_off_pos, mesh_off_quat = None, None
            if t == GeomType.MESH:
                md = m._mesh_data[g["mesh"]]
                m.geom_dataid[gi] = m.mesh_names.index(g["mesh"])
                # geom frame is composed with the mesh's COM / principal frame
                mesh_off_pos, mesh_off_quat = md["pos"], md["quat"]
                pos = pos + _rotate(mesh_off_pos, quat)
                quat = _quat_mul(quat, mesh_off_quat)
                ext = md["vert"][md["hull_vertices"]]
                size = (ext.max(0) - ext.min(0)) / 2.0
            gt.append(int(t))
            gsize.append(size)
            gpos.append(pos)
            gquat.append(quat / np.linalg.norm(quat))
            gbody.append(remap[g["__body__"]])
            # mass
            if t == GeomType.MESH:
                vol = md["volume"]
                unit_inertia = md["inertia_unit"]
            else:
                vol = _geom_volume(t, size)
            if "mass" in g:
                mass = float(g["mass"])
            else:
                mass = f(g, "density", 1000.0) * vol
            if t == GeomType.MESH:
                inertia = unit_inertia * (mass / vol if vol > 0 else 0.0)
            else:
                inertia = _geom_inertia(t, size, mass)
            gmass.append(mass)
            ginertia.append(inertia)
        m.geom_type = np.array(gt, dtype=np.int32)
        m.geom_size = np.array(gsize, dtype=np.float64).reshape(-1, 3)
        m.geom_pos = np.array(gpos, dtype=np.float64).reshape(-1, 3)
        m.geom_quat = np.array(gquat, dtype=np.float64).reshape(-1, 4)
        m.geom_bodyid = np.array(gbody, dtype=np.int32)
        m._geom_mass = np.array(gmass, dtype=np.float64)
        m._geom_inertia = np.array(ginertia, dtype=np.float64).reshape(-1, 3)
        m.geom_contype = np.array([int(g.get("contype", 1)) for g in G], dtype=np.int32)
        m.geom_conaffinity = np.array([int(g.get("conaffinity", 1)) for g in G], dtype=np.int32)
        m.geom_condim = np.array([int(g.get("condim", 3)) for g in G], dtype=np.int32)
        m.geom_priority = np.array([int(g.get("priority", 0)) for g in G], dtype=np.int32)
        m.geom_group = np.array([int(g.get("group", 0)) for g in G], dtype=np.int32)
        m.geom_solmix = np.array([f(g, "solmix", 1.0) for g in G])
        m.geom_margin = np.array([f(g, "margin", 0.0) for g in G])
        m.geom_gap = np.array([f(g, "gap", 0.0) for g in G])
        m.geom_solref = np.array([_pad(_floats(g["solref"]), 2, _DEF_SOLREF) if "solref" in g else _DEF_SOLREF for g in G]).reshape(-1, 2)
        m.geom_solimp = np.array([_pad(_floats(g["solimp"]), 5, _DEF_SOLIMP) if "solimp" in g else _DEF_SOLIMP for g in G]).reshape(-1, 5)
        deff = np.array([1.0, 0.005, 0.0001])
        m.geom_friction = np.array([_pad(_floats(g["friction"]), 3, deff) if "friction" in g else deff for g in G]).reshape(-1, 3)
        m.geom_rgba = np.array([_pad(_floats(g["rgba"]), 4, [0.5, 0.5, 0.5, 1]) if "rgba" in g else [0.5, 0.5, 0.5, 1.0] for g in G], dtype=np.float32).reshape(-1, 4)
        m.geom_matid = -np.ones(ngeom, dtype=np.int32)
        m.geom_sameframe = np.zeros(ngeom, dtype=np.uint8)
        m.geom_fluid = np.zeros((ngeom, 12))
        for g in G:  # the ellipsoid fluid model (per-geom interaction coefficients) is not modelled: only the inertia-box model of option density / viscosity is
            if g.get("fluidshape", "none") != "none":
                raise NotImplementedError("geom fluidshape='ellipsoid' (the ellipsoid fluid-interaction model) is not supported")
        # bounding radius / aabb (setup-only consumers)
        rb = []
        for gi in range(ngeom):
            t, s = m.geom_type[gi], m.geom_size[gi]
            if t == GeomType.SPHERE:
                rb.append(s[0])
            elif t == GeomType.CAPSULE:
                rb.append(s[0] + s[1])
            elif t == GeomType.CYLINDER:
                rb.append(math.hypot(s[0], s[1]))
            elif t in (GeomType.BOX, GeomType.ELLIPSOID, GeomType.MESH):
                rb.append(float(np.linalg.norm(s)) if t != GeomType.ELLIPSOID else float(np.max(s)))
            else:
                rb.append(0.0)
        m.geom_rbound = np.array(rb, dtype=np.float64)
        m.geom_aabb = np.concatenate([np.zeros((ngeom, 3)), m.geom_size], axis=1) if ngeom else np.zeros((0, 6))
        geomadr = -np.ones(m.nbody, dtype=np.int32)
        geomnum = np.zeros(m.nbody, dtype=np.int32)
        for gi, b in enumerate(m.geom_bodyid):
            if geomadr[b] < 0:
                geomadr[b] = gi
            geomnum[b] += 1
        m.body_geomadr, m.body_geomnum = geomadr, geomnum

    # ---- body inertial frames ---------------------------------------------
    def _build_inertia(self, m, bodies):
        nbody = m.nbody
        ipos = np.zeros((nbody, 3))
        iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
        mass = np.zeros(nbody)
        inertia = np.zeros((nbody, 3))
        for bi, b in enumerate(bodies):
            if bi == 0:
                continue
            inr = b["inertial"]
            use_geoms = self.inertiafromgeom == "true" or (self.inertiafromgeom == "auto" and inr is None)
            if not use_geoms:
                if inr is None:
                    continue
                ipos[bi] = _floats(inr["pos"]) if "pos" in inr else np.zeros(3)
                iquat[bi] = self._orientation(inr)
                mass[bi] = float(inr["mass"])
                if "diaginertia" in inr:
                    inertia[bi] = _floats(inr["diaginertia"])
                elif "fullinertia" in inr:
                    fi = _floats(inr["fullinertia"])
                    full = np.array([[fi[0], fi[3], fi[4]], [fi[3], fi[1], fi[5]], [fi[4], fi[5], fi[2]]])
                    q, d = _principal_axes(full)
                    iquat[bi] = _quat_mul(iquat[bi], q)
                    inertia[bi] = d
                continue
            gids = [g for g in range(m.ngeom) if m.geom_bodyid[g] == bi and m._geom_mass[g] > 0]
            if not gids:
                continue
            if len(gids) == 1:
                g = gids[0]
                ipos[bi] = m.geom_pos[g]
                iquat[bi] = m.geom_quat[g]
                mass[bi] = m._geom_mass[g]
                inertia[bi] = m._geom_inertia[g]
                continue
            tot = sum(m._geom_mass[g] for g in gids)
            com = sum(m._geom_mass[g] * m.geom_pos[g] for g in gids) / tot
            full = np.zeros((3, 3))
            for g in gids:
                R = _quat_to_mat(m.geom_quat[g])
                full += R @ np.diag(m._geom_inertia[g]) @ R.T
                d = m.geom_pos[g] - com
                full += m._geom_mass[g] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            q, dg = _principal_axes(full)
            ipos[bi], iquat[bi], mass[bi], inertia[bi] = com, q, tot, dg
        if self.boundmass > 0:
            mass[1:] = np.maximum(mass[1:], self.boundmass)
        if self.boundinertia > 0:
            inertia[1:] = np.maximum(inertia[1:], self.boundinertia)
        if self.settotalmass > 0 and mass.sum() > 0:
            s = self.settotalmass / mass.sum()
            mass *= s
            inertia *= s
        m.body_ipos, m.body_iquat, m.body_mass, m.body_inertia = ipos, iquat, mass, inertia
        sub = mass.copy()
        for bi in range(nbody - 1, 0, -1):
            sub[m.body_parentid[bi]] += sub[bi]
        m.body_subtreemass = sub

    # ---- sites / cameras / lights -------------------------------------------
    def _build_sites_cams_lights(self, m, remap):
        def by_body(items):
            idx = sorted(range(len(items)), key=lambda i: (remap[items[i]["__body__"]], i))
            return [items[i] for i in idx]

        S = by_body(self.sites)
        m.nsite = len(S)
        m.names_site = [s.get("name", "") for s in S]
        m.site_bodyid = np.array([remap[s["__body__"]] for s in S], dtype=np.int32)
        m.site_type = np.array([int(_geom_type(s.get("type", "sphere"), "site")) for s in S], dtype=np.int32)
        spos, squat, ssize = [], [], []
        for s in S:
            pos = _floats(s["pos"]) if "pos" in s else np.zeros(3)
            quat = self._orientation(s)
            size = np.array([0.005, 0.005, 0.005])
            if "size" in s:
                v = _floats(s["size"])
                size[: len(v)] = v[:3]
            if "fromto" in s:
                ft = _floats(s["fromto"])
                vec = ft[:3] - ft[3:]
                size[1] = np.linalg.norm(vec) / 2
                pos = (ft[:3] + ft[3:]) / 2
                quat = _z_to_quat(vec)
            spos.append(pos)
            squat.append(quat)
            ssize.append(size)
        m.site_pos = np.array(spos, dtype=np.float64).reshape(-1, 3)
        m.site_quat = np.array(squat, dtype=np.float64).reshape(-1, 4)
        m.site_size = np.array(ssize, dtype=np.float64).reshape(-1, 3)
        m.site_sameframe = np.zeros(m.nsite, dtype=np.uint8)

        C = by_body(self.cams)
        m.ncam = len(C)
        m.names_cam = [c.get("name", "") for c in C]
        m.cam_bodyid = np.array([remap[c["__body__"]] for c in C], dtype=np.int32)
        m.cam_mode = np.array([int(_CAM_MODES[c.get("mode", "fixed")]) for c in C], dtype=np.int32)
        m.cam_targetbodyid = np.array([m.names_body.index(c["target"]) if "target" in c else -1 for c in C], dtype=np.int32)
        m.cam_pos = np.array([_floats(c["pos"]) if "pos" in c else np.zeros(3) for c in C], dtype=np.float64).reshape(-1, 3)
        m.cam_quat = np.array([self._orientation(c) for c in C], dtype=np.float64).reshape(-1, 4)
        m.cam_fovy = np.array([float(c.get("fovy", 45.0)) for c in C], dtype=np.float64)
        m.cam_resolution = np.ones((m.ncam, 2), dtype=np.int32)
        m.cam_sensorsize = np.zeros((m.ncam, 2), dtype=np.float32)
        m.cam_intrinsic = np.tile(np.array([0.01, 0.01, 0, 0], dtype=np.float32), (m.ncam, 1))

        L = by_body(self.lights)
        m.nlight = len(L)
        m.light_bodyid = np.array([remap[c["__body__"]] for c in L], dtype=np.int32)
        m.light_mode = np.array([int(_CAM_MODES[c.get("mode", "fixed")]) for c in L], dtype=np.int32)
        m.light_targetbodyid = np.array([m.names_body.index(c["target"]) if "target" in c else -1 for c in L], dtype=np.int32)
        m.light_pos = np.array([_floats(c["pos"]) if "pos" in c else np.zeros(3) for c in L], dtype=np.float64).reshape(-1, 3)
        ld = []
        for c in L:
            d = _floats(c["dir"]) if "dir" in c else np.array([0.0, 0, -1])
            ld.append(d / np.linalg.norm(d))
        m.light_dir = np.array(ld, dtype=np.float64).reshape(-1, 3)
        m.light_type = np.array([1 if _bool(c.get("directional", "false")) else 0 for c in L], dtype=np.int32)
        m.light_castshadow = np.array([_bool(c.get("castshadow", "true")) for c in L], dtype=bool)
        m.light_active = np.array([_bool(c.get("active", "true")) for c in L], dtype=bool)
        m.light_cutoff = np.array([float(c.get("cutoff", 45.0)) for c in L], dtype=np.float32)
        m.light_exponent = np.array([float(c.get("exponent", 10.0)) for c in L], dtype=np.float32)
        m.light_attenuation = np.array([_pad(_floats(c["attenuation"]), 3, [1, 0, 0]) if "attenuation" in c else [1.0, 0, 0] for c in L], dtype=np.float32).reshape(-1, 3)
        m.light_diffuse = np.array([_pad(_floats(c["diffuse"]), 3, [0.7] * 3) if "diffuse" in c else [0.7] * 3 for c in L], dtype=np.float32).reshape(-1, 3)
        m.light_ambient = np.array([_pad(_floats(c["ambient"]), 3, [0.0] * 3) if "ambient" in c else [0.0] * 3 for c in L], dtype=np.float32).reshape(-1, 3)
        m.light_specular = np.array([_pad(_floats(c["specular"]), 3, [0.3] * 3) if "specular" in c else [0.3] * 3 for c in L], dtype=np.float32).reshape(-1, 3)

    # ---- contact pairs / excludes --------------------------------------------
    def _build_contact(self, m):
        pairs, excl = [], []
        for cn in self.root.findall("contact"):
            for p in cn.findall("pair"):
                a = self._merged(p, dict(self.defaults[self._t(p).get("class", "main")].get("pair")))
                pairs.append(a)
            for e in cn.findall("exclude"):
                excl.append(self._t(e))
        rows = []
        for a in pairs:
            g1, g2 = m.names_geom.index(a["geom1"]), m.names_geom.index(a["geom2"])
            b1, b2 = int(m.geom_bodyid[g1]), int(m.geom_bodyid[g2])
            if b1 > b2:  # signature is (body1 << 16) + body2 with body1 <= body2
                g1, g2, b1, b2 = g2, g1, b2, b1
            dim = int(a["condim"]) if "condim" in a else int(max(m.geom_condim[g1], m.geom_condim[g2]))
            if "friction" in a:
                fr = _pad(_floats(a["friction"]), 5, [1, 1, 0.005, 0.0001, 0.0001])
                fv = _floats(a["friction"])
                if len(fv) < 2:
                    fr[1] = fr[0]
                if len(fv) < 4:
                    pass
                if len(fv) < 5 and len(fv) >= 4:
                    fr[4] = fr[3]
            else:
                f3 = np.maximum(m.geom_friction[g1], m.geom_friction[g2])
                fr = np.array([f3[0], f3[0], f3[1], f3[2], f3[2]])
            s1, s2 = m.geom_solmix[g1], m.geom_solmix[g2]
            if s1 >= mjMINVAL and s2 >= mjMINVAL:
                mix = s1 / (s1 + s2)
            elif s1 < mjMINVAL and s2 < mjMINVAL:
                mix = 0.5
            elif s1 < mjMINVAL:
                mix = 0.0
            else:
                mix = 1.0
            if "solref" in a:
                solref = _pad(_floats(a["solref"]), 2, _DEF_SOLREF)
            elif m.geom_solref[g1][0] > 0 and m.geom_solref[g2][0] > 0:
                solref = mix * m.geom_solref[g1] + (1 - mix) * m.geom_solref[g2]
            else:
                solref = np.minimum(m.geom_solref[g1], m.geom_solref[g2])
            solimp = _pad(_floats(a["solimp"]), 5, _DEF_SOLIMP) if "solimp" in a else mix * m.geom_solimp[g1] + (1 - mix) * m.geom_solimp[g2]
            solreffriction = _pad(_floats(a["solreffriction"]), 2, [0, 0]) if "solreffriction" in a else np.zeros(2)
            margin = float(a["margin"]) if "margin" in a else max(m.geom_margin[g1], m.geom_margin[g2])
            gap = float(a["gap"]) if "gap" in a else max(m.geom_gap[g1], m.geom_gap[g2])
            rows.append(dict(g1=g1, g2=g2, sig=(b1 << 16) + b2, dim=dim, friction=fr, solref=solref, solimp=solimp, solreffriction=solreffriction, margin=margin, gap=gap))
        rows.sort(key=lambda r: r["sig"])  # stable: pairs are ordered by body signature
        m.npair = len(rows)
        m.pair_dim = np.array([r["dim"] for r in rows], dtype=np.int32)
        m.pair_geom1 = np.array([r["g1"] for r in rows], dtype=np.int32)
        m.pair_geom2 = np.array([r["g2"] for r in rows], dtype=np.int32)
        m.pair_signature = np.array([r["sig"] for r in rows], dtype=np.int32)
        m.pair_friction = np.array([r["friction"] for r in rows], dtype=np.float64).reshape(-1, 5)
        m.pair_solref = np.array([r["solref"] for r in rows], dtype=np.float64).reshape(-1, 2)
        m.pair_solreffriction = np.array([r["solreffriction"] for r in rows], dtype=np.float64).reshape(-1, 2)
        m.pair_solimp = np.array([r["solimp"] for r in rows], dtype=np.float64).reshape(-1, 5)
        m.pair_margin = np.array([r["margin"] for r in rows], dtype=np.float64)
        m.pair_gap = np.array([r["gap"] for r in rows], dtype=np.float64)
        sigs = []
        for e in excl:
            b1, b2 = m.names_body.index(e["body1"]), m.names_body.index(e["body2"])
            if b1 > b2:
                b1, b2 = b2, b1
            sigs.append((b1 << 16) + b2)
        m.nexclude = len(sigs)
        m.exclude_signature = np.array(sorted(sigs), dtype=np.int32)

    # ---- actuators ---------------------------------------------------------
    def _build_actuators(self, m):
        acts = []
        for an in self.root.findall("actuator"):
            for node in an:
                if node.tag not in _ACT_TAGS:
                    raise NotImplementedError(f"actuator <{node.tag}> not supported")
                base = dict(self.defaults[self._t(node).get("class", "main")].get(node.tag))
                base.pop("__tag__", None)
                a = self._merged(node, base)
                a["__tag__"] = node.tag
                acts.append(a)
        nu = len(acts)
        m.nu = nu
        m.names_actuator = [a.get("name", "") for a in acts]
        m.actuator_trntype = np.zeros(nu, dtype=np.int32)
        m.actuator_dyntype = np.zeros(nu, dtype=np.int32)
        m.actuator_gaintype = np.zeros(nu, dtype=np.int32)
        m.actuator_biastype = np.zeros(nu, dtype=np.int32)
        m.actuator_trnid = -np.ones((nu, 2), dtype=np.int32)
        m.actuator_dynprm = np.zeros((nu, mjNDYN))
        m.actuator_gainprm = np.zeros((nu, mjNGAIN))
        m.actuator_biasprm = np.zeros((nu, mjNBIAS))
        m.actuator_gear = np.zeros((nu, 6))
        m.actuator_ctrlrange = np.zeros((nu, 2))
        m.actuator_forcerange = np.zeros((nu, 2))
        m.actuator_actrange = np.zeros((nu, 2))
        m.actuator_ctrllimited = np.zeros(nu, dtype=bool)
        m.actuator_forcelimited = np.zeros(nu, dtype=bool)
        m.actuator_actlimited = np.zeros(nu, dtype=bool)
        m.actuator_actadr = -np.ones(nu, dtype=np.int32)
        m.actuator_actnum = np.zeros(nu, dtype=np.int32)
        m.actuator_group = np.zeros(nu, dtype=np.int32)
        m.actuator_actearly = np.zeros(nu, dtype=np.uint8)
        m.actuator_cranklength = np.zeros(nu)
        m.actuator_lengthrange = np.zeros((nu, 2))
        m.actuator_acc0 = np.zeros(nu)
        na = 0
        for i, a in enumerate(acts):
            tag = a["__tag__"]
            if "joint" in a:
                m.actuator_trntype[i] = TrnType.JOINT
                m.actuator_trnid[i, 0] = m.names_jnt.index(a["joint"])
            elif "jointinparent" in a:
                m.actuator_trntype[i] = TrnType.JOINTINPARENT
                m.actuator_trnid[i, 0] = m.names_jnt.index(a["jointinparent"])
            elif "tendon" in a:
                m.actuator_trntype[i] = TrnType.TENDON
                m.actuator_trnid[i, 0] = m.names_tendon.index(a["tendon"])
            else:
                raise NotImplementedError("only joint / jointinparent / tendon transmissions are supported")
            g = _floats(a["gear"]) if "gear" in a else np.array([1.0])
            m.actuator_gear[i, : len(g)] = g
            if "gear" not in a:
                m.actuator_gear[i, 0] = 1.0
            m.actuator_gainprm[i, 0] = 1.0
            if tag == "motor":
                pass
            elif tag == "position":
                kp = float(a.get("kp", 1.0))
                kv = float(a.get("kv", 0.0))
                m.actuator_gainprm[i, 0] = kp
                m.actuator_biastype[i] = BiasType.AFFINE
                m.actuator_biasprm[i, :3] = [0.0, -kp, -kv]
            elif tag == "velocity":
                kv = float(a.get("kv", 1.0))
                m.actuator_gainprm[i, 0] = kv
                m.actuator_biastype[i] = BiasType.AFFINE
                m.actuator_biasprm[i, :3] = [0.0, 0.0, -kv]
            elif tag == "intvelocity":  # integrated-velocity servo: an integrator activation tracked by a position servo (MuJoCo XML reference)
                kp = float(a.get("kp", 1.0))
                kv = float(a.get("kv", 0.0))
                m.actuator_dyntype[i] = DynType.INTEGRATOR
                m.actuator_gainprm[i, 0] = kp
                m.actuator_biastype[i] = BiasType.AFFINE
                m.actuator_biasprm[i, :3] = [0.0, -kp, -kv]
            elif tag == "cylinder":  # pneumatic / hydraulic cylinder (MuJoCo XML reference): dyntype filter, gaintype fixed (area), biastype affine
                area = float(a.get("area", 1.0))
                if "diameter" in a:
                    area = math.pi / 4.0 * float(a["diameter"]) ** 2
                m.actuator_dyntype[i] = DynType.FILTER
                m.actuator_dynprm[i, 0] = float(a.get("timeconst", 1.0))
                m.actuator_gainprm[i, 0] = area
                m.actuator_biastype[i] = BiasType.AFFINE
                bias = _floats(a.get("bias", "0 0 0"))
                m.actuator_biasprm[i, : len(bias[:3])] = bias[:3]
            elif tag == "damper":  # force = -kv * velocity * ctrl (affine gain on the velocity), ctrl >= 0
                kv = float(a.get("kv", 1.0))
                m.actuator_gaintype[i] = GainType.AFFINE
                m.actuator_gainprm[i, :3] = [0.0, 0.0, -kv]
            elif tag == "muscle":
                # <muscle> shortcut (MuJoCo XML reference): dyntype = gaintype = biastype = muscle; dynprm = timeconst (2), tausmooth;
                # gainprm = biasprm = range (2), force, scale, lmin, lmax, vmax, fpmax, fvmax.  The length range must be given: the
                # compiler's simulation-based mj_setLengthRange is not restated here.
                if "lengthrange" not in a:
                    raise NotImplementedError("<muscle> needs an explicit lengthrange (automatic length-range computation is not supported)")
                m.actuator_dyntype[i] = DynType.MUSCLE
                m.actuator_gaintype[i] = GainType.MUSCLE
                m.actuator_biastype[i] = BiasType.MUSCLE
                tc = _floats(a.get("timeconst", "0.01 0.04"))
                m.actuator_dynprm[i, :] = 0
                m.actuator_dynprm[i, :3] = [tc[0], tc[1], float(a.get("tausmooth", 0.0))]
                rg = _floats(a.get("range", "0.75 1.05"))
                prm = [rg[0], rg[1], float(a.get("force", -1.0)), float(a.get("scale", 200.0)), float(a.get("lmin", 0.5)), float(a.get("lmax", 1.6)),
                       float(a.get("vmax", 1.5)), float(a.get("fpmax", 1.3)), float(a.get("fvmax", 1.2))]
                m.actuator_gainprm[i, :] = 0
                m.actuator_biasprm[i, :] = 0
                m.actuator_gainprm[i, :9] = prm
                m.actuator_biasprm[i, :9] = prm
            elif tag == "general":
                dt = a.get("dyntype", "none")
                m.actuator_dyntype[i] = {"none": DynType.NONE, "integrator": DynType.INTEGRATOR, "filter": DynType.FILTER, "filterexact": DynType.FILTEREXACT, "muscle": DynType.MUSCLE}[dt]
                gt_ = a.get("gaintype", "fixed")
                m.actuator_gaintype[i] = {"fixed": GainType.FIXED, "affine": GainType.AFFINE, "muscle": GainType.MUSCLE}[gt_]
                bt = a.get("biastype", "none")
                m.actuator_biastype[i] = {"none": BiasType.NONE, "affine": BiasType.AFFINE, "muscle": BiasType.MUSCLE}[bt]
                for key, arr in (("dynprm", m.actuator_dynprm), ("gainprm", m.actuator_gainprm), ("biasprm", m.actuator_biasprm)):
                    if key in a:
                        v = _floats(a[key])
                        arr[i, :] = 0
                        arr[i, : len(v)] = v
            else:
                raise NotImplementedError(f"actuator <{tag}> not supported")
            for key, arr, limarr, attr in (
                ("ctrlrange", m.actuator_ctrlrange, m.actuator_ctrllimited, "ctrllimited"),
                ("forcerange", m.actuator_forcerange, m.actuator_forcelimited, "forcelimited"),
                ("actrange", m.actuator_actrange, m.actuator_actlimited, "actlimited"),
            ):
                has = key in a
                if has:
                    arr[i] = _floats(a[key])
                lim = a.get(attr, "auto")
                limarr[i] = (has and self.autolimits) if lim == "auto" else _bool(lim)
            if "lengthrange" in a:
                m.actuator_lengthrange[i] = _floats(a["lengthrange"])
            elif m.actuator_gaintype[i] == GainType.MUSCLE or m.actuator_biastype[i] == BiasType.MUSCLE:
                # (the <muscle> shortcut raised above; this is <general gaintype="muscle"> / biastype="muscle"): MuJoCo would compute the range by
                # simulation (mj_setLengthRange); with (0, 0) the muscle's L0 is 0, clamped to mjMINVAL, and the forces are silently wrong
                raise NotImplementedError(f"actuator {a.get('name', i)!r}: muscle gain / bias needs an explicit lengthrange (automatic length-range computation is not supported)")
            if m.actuator_dyntype[i] != DynType.NONE:
                m.actuator_actadr[i] = na
                m.actuator_actnum[i] = 1
                na += 1
        m.na = na

    def _build_sensors(self, m):
        sens = []
        for sn in self.root.findall("sensor"):
            for node in sn:
                if node.tag not in _SENSOR_DIMS:
                    raise NotImplementedError(f"sensor <{node.tag}> not supported")
                sens.append(node)
        m.nsensor = len(sens)
        def user_attr(sn, key, table, default):
            v = self._t(sn).get(key)
            if v is None:
                return default
            if v not in table:
                raise ValueError(f"sensor <user>: {key} must be one of {sorted(table)}, got {v!r}")
            return table[v]

        def spec(sn, k):  # dim, needstage, datatype of one element (<user> carries its own: dim is required, needstage defaults to acc, datatype to real)
            if sn.tag != "user":
                return _SENSOR_DIMS[sn.tag][k]
            if k == 1:
                dim = self._t(sn).get("dim")
                if dim is None or int(dim) < 0:
                    raise ValueError("sensor <user>: a non-negative dim is required")
                return int(dim)
            return user_attr(sn, "needstage", _STAGE_NAMES, 3) if k == 2 else user_attr(sn, "datatype", _DATATYPE_NAMES, 0)

        dims = [spec(s, 1) for s in sens]
        m.sensor_type = np.array([int(_SENSOR_DIMS[s.tag][0]) for s in sens], dtype=np.int32)
        m.sensor_dim = np.array(dims, dtype=np.int32)
        m.sensor_adr = np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(np.int32) if sens else np.zeros(0, dtype=np.int32)
        m.nsensordata = int(sum(dims))
        lists = {ObjType.BODY: m.names_body, ObjType.XBODY: m.names_body, ObjType.GEOM: m.names_geom, ObjType.SITE: m.names_site, ObjType.CAMERA: m.names_cam}

        def lookup(sn, kind, names, attr):
            name = self._t(sn).get(attr)
            if name is None or name not in names:
                raise ValueError(f"sensor <{sn.tag}>: {kind} {name!r} not found")
            return names.index(name)

        objid, objtype, reftype, refid = [], [], [], []
        for sn in sens:
            attach = _SENSOR_DIMS[sn.tag][4]
            rt, ri = int(ObjType.UNKNOWN), -1
            if attach == "site":
                ot, oi = int(ObjType.SITE), lookup(sn, "site", m.names_site, "site")
            elif attach == "site+camera":  # camprojection: the site is the object, the camera the reference (user_objects.cc)
                ot, oi = int(ObjType.SITE), lookup(sn, "site", m.names_site, "site")
                rt, ri = int(ObjType.CAMERA), lookup(sn, "camera", m.names_cam, "camera")
            elif attach == "user":  # optional object
                kind, name = self._t(sn).get("objtype"), self._t(sn).get("objname")
                if (kind is None) != (name is None):
                    raise ValueError("sensor <user>: objtype and objname go together")
                ot, oi = int(ObjType.UNKNOWN), -1
                if kind is not None:
                    if kind not in _USER_OBJTYPES:
                        raise ValueError(f"sensor <user>: objtype must be one of {sorted(_USER_OBJTYPES)}, got {kind!r}")
                    names = {ObjType.JOINT: m.names_jnt, ObjType.TENDON: getattr(m, "names_tendon", []), ObjType.ACTUATOR: getattr(m, "names_actuator", [])}
                    ot = int(_USER_OBJTYPES[kind])
                    oi = lookup(sn, kind, names.get(_USER_OBJTYPES[kind]) if _USER_OBJTYPES[kind] in names else lists[_USER_OBJTYPES[kind]], "objname")
            elif attach == "joint":
                ot, oi = int(ObjType.JOINT), lookup(sn, "joint", m.names_jnt, "joint")
                jt = int(m.jnt_type[oi])
                if sn.tag in ("ballquat", "ballangvel") and jt != int(JointType.BALL):
                    raise ValueError(f"sensor <{sn.tag}>: joint {self._t(sn).get('joint')!r} must be a ball joint")
                if sn.tag in ("jointpos", "jointvel", "jointlimitpos", "jointlimitvel", "jointlimitfrc") and jt not in (int(JointType.SLIDE), int(JointType.HINGE)):
                    raise ValueError(f"sensor <{sn.tag}>: joint {self._t(sn).get('joint')!r} must be a slide or hinge joint")
            elif attach == "tendon":
                ot, oi = int(ObjType.TENDON), lookup(sn, "tendon", getattr(m, "names_tendon", []), "tendon")
            elif attach == "actuator":
                ot, oi = int(ObjType.ACTUATOR), lookup(sn, "actuator", getattr(m, "names_actuator", []), "actuator")
            elif attach == "body":
                ot, oi = int(ObjType.BODY), lookup(sn, "body", m.names_body, "body")
            elif attach == "frame":
                kind = self._t(sn).get("objtype")
                if kind not in _OBJTYPE_NAMES:
                    raise ValueError(f"sensor <{sn.tag}>: objtype must be one of {sorted(_OBJTYPE_NAMES)}, got {kind!r}")
                ot = int(_OBJTYPE_NAMES[kind])
                oi = lookup(sn, kind, lists[_OBJTYPE_NAMES[kind]], "objname")
                if self._t(sn).get("reftype") is not None or self._t(sn).get("refname") is not None:
                    rkind = self._t(sn).get("reftype")
                    if rkind not in _OBJTYPE_NAMES:
                        raise ValueError(f"sensor <{sn.tag}>: reftype must be one of {sorted(_OBJTYPE_NAMES)}, got {rkind!r}")
                    rt = int(_OBJTYPE_NAMES[rkind])
                    ri = lookup(sn, rkind, lists[_OBJTYPE_NAMES[rkind]], "refname")
            else:
                ot, oi = int(ObjType.UNKNOWN), -1
            objtype.append(ot); objid.append(oi); reftype.append(rt); refid.append(ri)
        m.sensor_objid = np.array(objid, dtype=np.int32)
        m.sensor_objtype = np.array(objtype, dtype=np.int32)
        m.sensor_needstage = np.array([spec(s, 2) for s in sens], dtype=np.int32)
        m.sensor_datatype = np.array([spec(s, 3) for s in sens], dtype=np.int32)
        m.sensor_reftype = np.array(reftype, dtype=np.int32)
        m.sensor_refid = np.array(refid, dtype=np.int32)
        m.sensor_cutoff = np.array([float(self._t(s).get("cutoff", 0.0)) for s in sens], dtype=np.float64)

    def _build_equality(self, m):
        """<equality>: connect / weld / joint (body form; site form is compiled to site ids like MuJoCo does).

        eq_data follows MuJoCo's layout: connect [anchor in body1 (3), anchor in body2 (3)]; weld [anchor in body2 (3), anchor in
        body1 (3), relpose quat (4), torquescale]; joint [polycoef (5)].  The qpos0-dependent parts (second anchor, relpose) are
        filled by ``_equality_set0`` after the reference-pose kinematics, as mj_setConst does."""
        nodes = [e for en in self.root.findall("equality") for e in en]
        m.neq = len(nodes)
        kinds = {"connect": 0, "weld": 1, "joint": 2}
        m.eq_type = np.zeros(m.neq, dtype=np.int32)
        m.eq_obj1id = np.zeros(m.neq, dtype=np.int32)
        m.eq_obj2id = np.zeros(m.neq, dtype=np.int32)
        m.eq_objtype = np.zeros(m.neq, dtype=np.int32)
        m.eq_active0 = np.zeros(m.neq, dtype=bool)
        m.eq_solref = np.tile(_DEF_SOLREF, (m.neq, 1)).reshape(m.neq, 2)
        m.eq_solimp = np.tile(_DEF_SOLIMP, (m.neq, 1)).reshape(m.neq, 5)
        m.eq_data = np.zeros((m.neq, 11))
        m.names_eq = []
        for i, node in enumerate(nodes):
            if node.tag not in kinds:
                raise NotImplementedError(f"equality type <{node.tag}> is outside this build's MJCF subset (connect, weld, joint)")
            base = dict(self.defaults[self._t(node).get("class", "main")].get("equality"))
            base.pop("__tag__", None)
            a = self._merged(node, base)
            m.names_eq.append(a.get("name", ""))
            m.eq_type[i] = kinds[node.tag]
            m.eq_active0[i] = _bool(a.get("active", "true"))
            if "solref" in a:
                m.eq_solref[i] = _pad(_floats(a["solref"]), 2, _DEF_SOLREF)
            if "solimp" in a:
                m.eq_solimp[i] = _pad(_floats(a["solimp"]), 5, _DEF_SOLIMP)
            if node.tag == "joint":
                m.eq_objtype[i] = 3  # mjOBJ_JOINT
                m.eq_obj1id[i] = m.names_jnt.index(a["joint1"])
                m.eq_obj2id[i] = m.names_jnt.index(a["joint2"]) if "joint2" in a else -1
                m.eq_data[i, :5] = _pad(_floats(a["polycoef"]), 5, np.array([0.0, 1, 0, 0, 0])) if "polycoef" in a else [0.0, 1, 0, 0, 0]
                continue
            if "site1" in a:  # site form: MuJoCo stores site ids and no anchors
                m.eq_objtype[i] = 6  # mjOBJ_SITE
                m.eq_obj1id[i] = m.names_site.index(a["site1"])
                m.eq_obj2id[i] = m.names_site.index(a["site2"])
                if node.tag == "weld":
                    m.eq_data[i, 10] = float(a.get("torquescale", 1.0))
                continue
            m.eq_objtype[i] = 1  # mjOBJ_BODY
            m.eq_obj1id[i] = m.names_body.index(a["body1"])
            m.eq_obj2id[i] = m.names_body.index(a["body2"]) if "body2" in a else 0
            if node.tag == "connect":
                m.eq_data[i, 0:3] = _floats(a["anchor"])
            else:
                m.eq_data[i, 0:3] = _floats(a["anchor"]) if "anchor" in a else np.zeros(3)
                m.eq_data[i, 10] = float(a.get("torquescale", 1.0))
                if "relpose" in a and np.any(_floats(a["relpose"])[3:] != 0):
                    # an explicit pose of body2 in the frame of body1 (position, quaternion): stored as given -- the quaternion normalised -- and left alone by
                    # the reference-pose pass below (MuJoCo's set0 skips welds whose quaternion is set); all-zero quaternion = "use qpos0", the default
                    rp = _floats(a["relpose"])
                    m.eq_data[i, 3:6] = rp[:3]
                    m.eq_data[i, 6:10] = rp[3:7] / np.linalg.norm(rp[3:7])

    def _build_tendons(self, m):
        """<tendon><fixed>: linear combinations of scalar joint positions (MuJoCo wrap objects of type JOINT)."""
        nodes = [t for tn in self.root.findall("tendon") for t in tn]
        m.ntendon = len(nodes)
        nt = m.ntendon
        m.names_tendon = []
        m.tendon_adr = np.zeros(nt, dtype=np.int32)
        m.tendon_num = np.zeros(nt, dtype=np.int32)
        m.tendon_limited = np.zeros(nt, dtype=bool)
        m.tendon_range = np.zeros((nt, 2))
        m.tendon_margin = np.zeros(nt)
        m.tendon_stiffness = np.zeros(nt)
        m.tendon_damping = np.zeros(nt)
        m.tendon_armature = np.zeros(nt)
        m.tendon_frictionloss = np.zeros(nt)
        m.tendon_lengthspring = -np.ones((nt, 2))
        m.tendon_length0 = np.zeros(nt)
        m.tendon_invweight0 = np.zeros(nt)
        m.tendon_solref_lim = np.tile(_DEF_SOLREF, (nt, 1)).reshape(nt, 2)
        m.tendon_solimp_lim = np.tile(_DEF_SOLIMP, (nt, 1)).reshape(nt, 5)
        m.tendon_solref_fri = np.tile(_DEF_SOLREF, (nt, 1)).reshape(nt, 2)
        m.tendon_solimp_fri = np.tile(_DEF_SOLIMP, (nt, 1)).reshape(nt, 5)
        wrap_type, wrap_objid, wrap_prm = [], [], []
        for i, node in enumerate(nodes):
            if node.tag not in ("fixed", "spatial"):
                raise NotImplementedError(f"tendon <{node.tag}> is not an MJCF tendon element (fixed, spatial)")
            base = dict(self.defaults[self._t(node).get("class", "main")].get("tendon"))
            base.pop("__tag__", None)
            a = self._merged(node, base)
            m.names_tendon.append(a.get("name", ""))
            m.tendon_adr[i] = len(wrap_type)
            if node.tag == "spatial":
                # <spatial>: the path as MuJoCo's wrap objects -- site (mjWRAP_SITE 3), sphere / cylinder geom with an optional side site (4 / 5), pulley (2).
                # device_put carries site-only paths in the form the reference evaluates them (zero length and Jacobian, smooth.py:470-497) and refuses the rest.
                for wn in node:
                    w = self._t(wn)
                    if wn.tag == "site":
                        wrap_type.append(3); wrap_objid.append(m.names_site.index(w["site"])); wrap_prm.append(0.0)
                    elif wn.tag == "geom":
                        g = m.names_geom.index(w["geom"])
                        gt = int(m.geom_type[g])
                        if gt not in (int(GeomType.SPHERE), int(GeomType.CYLINDER)):
                            raise ValueError("a tendon wraps sphere or cylinder geoms")
                        wrap_type.append(4 if gt == int(GeomType.SPHERE) else 5); wrap_objid.append(g)
                        wrap_prm.append(float(m.names_site.index(w["sidesite"])) if "sidesite" in w else -1.0)
                    elif wn.tag == "pulley":
                        wrap_type.append(2); wrap_objid.append(-1); wrap_prm.append(float(w["divisor"]))
                    else:
                        raise NotImplementedError(f"<spatial> tendon child <{wn.tag}>")
                for k in ("width",):
                    a.get(k)  # (rendering only)
            for jn in (node.findall("joint") if node.tag == "fixed" else []):
                jt = self._t(jn)
                j = m.names_jnt.index(jt.get("joint"))
                if int(m.jnt_type[j]) not in (int(JointType.SLIDE), int(JointType.HINGE)):
                    raise ValueError("fixed tendons act on slide / hinge joints")
                wrap_type.append(1)  # mjWRAP_JOINT
                wrap_objid.append(j)
                wrap_prm.append(float(jt.get("coef")))
            m.tendon_num[i] = len(wrap_type) - int(m.tendon_adr[i])
            has_range = "range" in a
            if has_range:
                m.tendon_range[i] = _floats(a["range"])
            lim = a.get("limited", "auto")
            m.tendon_limited[i] = (has_range and self.autolimits) if lim == "auto" else _bool(lim)
            m.tendon_margin[i] = float(a.get("margin", 0.0))
            m.tendon_stiffness[i] = float(a.get("stiffness", 0.0))
            m.tendon_damping[i] = float(a.get("damping", 0.0))
            m.tendon_armature[i] = float(a.get("armature", 0.0))
            m.tendon_frictionloss[i] = float(a.get("frictionloss", 0.0))
            if "springlength" in a:
                v = _floats(a["springlength"])
                m.tendon_lengthspring[i] = [v[0], v[1] if len(v) > 1 else v[0]]
            if "solreflimit" in a:
                m.tendon_solref_lim[i] = _pad(_floats(a["solreflimit"]), 2, _DEF_SOLREF)
            if "solimplimit" in a:
                m.tendon_solimp_lim[i] = _pad(_floats(a["solimplimit"]), 5, _DEF_SOLIMP)
            if "solreffriction" in a:
                m.tendon_solref_fri[i] = _pad(_floats(a["solreffriction"]), 2, _DEF_SOLREF)
            if "solimpfriction" in a:
                m.tendon_solimp_fri[i] = _pad(_floats(a["solimpfriction"]), 5, _DEF_SOLIMP)
        m.nwrap = len(wrap_type)
        m.wrap_type = np.array(wrap_type, dtype=np.int32)
        m.wrap_objid = np.array(wrap_objid, dtype=np.int32)
        m.wrap_prm = np.array(wrap_prm, dtype=np.float64)

    def _build_empty_sections(self, m):
        """<custom><numeric name=.. data=.. [size=..]/>: MuJoCo's user numerics (the reference reads `max_contact_points` from them,
        collision_driver.py:571-578).  `names` holds only the numerics' names, null-terminated, as MuJoCo's name buffer would."""
        nums = [n for cn in self.root.findall("custom") for n in cn.findall("numeric")]
        m.nnumeric = len(nums)
        m.nuserdata = 0
        adr, data, nadr, names = [], [], [], b""
        for n in nums:
            vals = _floats(self._t(n).get("data", "0"))
            size = int(self._t(n).get("size", len(vals)))
            vals = (list(vals) + [0.0] * size)[:size]
            adr.append(len(data))
            data.extend(vals)
            nadr.append(len(names))
            names += self._t(n).get("name", "").encode("utf-8") + b"\x00"
        m.numeric_adr = np.array(adr, dtype=np.int32)
        m.numeric_size = np.array([int(n.get("size", len(_floats(n.get("data", "0"))))) for n in nums], dtype=np.int32)
        m.numeric_data = np.array(data, dtype=np.float64)
        m.name_numericadr = np.array(nadr, dtype=np.int32)
        m.names = names

    def _build_keyframes(self, m):
        keys = [k for kn in self.root.findall("keyframe") for k in kn.findall("key")]
        m.nkey = len(keys)
        m.key_qpos = np.tile(m.qpos0, (m.nkey, 1)) if m.nkey else np.zeros((0, m.nq))
        m.key_qvel = np.zeros((m.nkey, m.nv))
        m.key_ctrl = np.zeros((m.nkey, m.nu))
        m.key_time = np.zeros(m.nkey)
        m.names_key = [self._t(k).get("name", "") for k in keys]
        for i, k in enumerate(keys):
            if "qpos" in self._t(k):
                m.key_qpos[i] = _floats(self._t(k).get("qpos"))
            if "qvel" in self._t(k):
                m.key_qvel[i] = _floats(self._t(k).get("qvel"))
            if "ctrl" in self._t(k):
                m.key_ctrl[i] = _floats(self._t(k).get("ctrl"))
            if "time" in self._t(k):
                m.key_time[i] = float(self._t(k).get("time"))


# --------------------------------------------------------------------------
# mj_setConst restatement (qpos0-derived constants)
# --------------------------------------------------------------------------


def _kinematics0(m, qpos):
    """World frames at ``qpos`` (numpy, host). Returns xpos, xquat, xmat, xipos, ximat, xanchor, xaxis."""
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nb):
        p = int(m.body_parentid[b])
        pos = xpos[p] + _rotate(m.body_pos[b], xquat[p])
        quat = _quat_mul(xquat[p], m.body_quat[b])
        for j in range(int(m.body_jntadr[b]), int(m.body_jntadr[b]) + int(m.body_jntnum[b])) if m.body_jntnum[b] else []:
            jt = int(m.jnt_type[j])
            qa = int(m.jnt_qposadr[j])
            if jt == JointType.FREE:
                xanchor[j] = qpos[qa : qa + 3]
                xaxis[j] = [0, 0, 1]
                pos = qpos[qa : qa + 3].copy()
                quat = qpos[qa + 3 : qa + 7] / np.linalg.norm(qpos[qa + 3 : qa + 7])
                continue
            anchor = _rotate(m.jnt_pos[j], quat) + pos
            axis = _rotate(m.jnt_axis[j], quat)
            xanchor[j], xaxis[j] = anchor, axis
            if jt == JointType.BALL:
                ql = qpos[qa : qa + 4] / np.linalg.norm(qpos[qa : qa + 4])
                quat = _quat_mul(quat, ql)
                pos = anchor - _rotate(m.jnt_pos[j], quat)
            elif jt == JointType.HINGE:
                quat = _quat_mul(quat, _axisangle_quat(m.jnt_axis[j], qpos[qa] - m.qpos0[qa]))
                pos = anchor - _rotate(m.jnt_pos[j], quat)
            else:
                pos = pos + axis * (qpos[qa] - m.qpos0[qa])
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    xmat = np.stack([_quat_to_mat(q) for q in xquat])
    xipos = np.stack([xpos[b] + xmat[b] @ m.body_ipos[b] for b in range(nb)])
    ximat = np.stack([_quat_to_mat(_quat_mul(xquat[b], m.body_iquat[b])) for b in range(nb)])
    return xpos, xquat, xmat, xipos, ximat, xanchor, xaxis


def _body_jac(m, body, point, xpos, xmat, xanchor, xaxis):
    """(3,nv) translational and rotational Jacobians of ``point`` fixed to ``body`` (qvel convention)."""
    jp = np.zeros((3, m.nv))
    jr = np.zeros((3, m.nv))
    b = body
    while b > 0:
        for j in range(int(m.body_jntadr[b]), int(m.body_jntadr[b]) + int(m.body_jntnum[b])) if m.body_jntnum[b] else []:
            jt = int(m.jnt_type[j])
            d = int(m.jnt_dofadr[j])
            if jt == JointType.FREE:
                jp[:, d : d + 3] = np.eye(3)
                for k in range(3):
                    ax = xmat[b][:, k]
                    jr[:, d + 3 + k] = ax
                    jp[:, d + 3 + k] = np.cross(ax, point - xpos[b])
            elif jt == JointType.BALL:
                for k in range(3):
                    ax = xmat[b][:, k]
                    jr[:, d + k] = ax
                    jp[:, d + k] = np.cross(ax, point - xanchor[j])
            elif jt == JointType.HINGE:
                jr[:, d] = xaxis[j]
                jp[:, d] = np.cross(xaxis[j], point - xanchor[j])
            else:
                jp[:, d] = xaxis[j]
        b = int(m.body_parentid[b])
    return jp, jr


def mass_matrix0(m, qpos=None):
    """Dense joint-space inertia at ``qpos`` (default qpos0), via body Jacobians."""
    qpos = m.qpos0 if qpos is None else qpos
    xpos, xquat, xmat, xipos, ximat, xanchor, xaxis = _kinematics0(m, qpos)
    M = np.diag(m.dof_armature.astype(np.float64)) if m.nv else np.zeros((0, 0))
    for b in range(1, m.nbody):
        if m.body_mass[b] == 0 and not m.body_inertia[b].any():
            continue
        jp, jr = _body_jac(m, b, xipos[b], xpos, xmat, xanchor, xaxis)
        I = ximat[b] @ np.diag(m.body_inertia[b]) @ ximat[b].T
        M = M + m.body_mass[b] * jp.T @ jp + jr.T @ I @ jr
    return M, (xpos, xquat, xmat, xipos, ximat, xanchor, xaxis)


def _set_const(m, stat_meaninertia=None):
    """qpos0 constants: invweights, dof_M0, meaninertia, acc0, cam/light reference frames."""
    nv = m.nv
    M, kin = mass_matrix0(m)
    xpos, xquat, xmat, xipos, ximat, xanchor, xaxis = kin
    m.dof_M0 = np.diag(M).copy() if nv else np.zeros(0)
    Minv = np.linalg.inv(M) if nv else np.zeros((0, 0))
    # dof_invweight0: diagonal of M^-1, averaged over the dofs of free (3+3) / ball (3) joints
    diw = np.diag(Minv).copy() if nv else np.zeros(0)
    for j in range(m.njnt):
        d = int(m.jnt_dofadr[j])
        jt = int(m.jnt_type[j])
        if jt == JointType.FREE:
            diw[d : d + 3] = diw[d : d + 3].mean()
            diw[d + 3 : d + 6] = diw[d + 3 : d + 6].mean()
        elif jt == JointType.BALL:
            diw[d : d + 3] = diw[d : d + 3].mean()
    m.dof_invweight0 = diw
    biw = np.zeros((m.nbody, 2))
    for b in range(1, m.nbody):
        if m.body_weldid[b] == 0:
            continue
        jp, jr = _body_jac(m, b, xipos[b], xpos, xmat, xanchor, xaxis)
        biw[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        biw[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    m.body_invweight0 = biw
    meaninertia = float(np.mean(np.diag(M))) if nv else 1.0
    subtree_com0 = np.zeros(3)
    tot = m.body_mass.sum()
    center = (m.body_mass[:, None] * xipos).sum(0) / tot if tot > 0 else np.zeros(3)
    m.stat = SimpleNamespace(
        meaninertia=stat_meaninertia if stat_meaninertia is not None else meaninertia,
        meanmass=float(m.body_mass[1:].mean()) if m.nbody > 1 else 0.0,
        meansize=float(np.mean(m.geom_rbound)) if m.ngeom else 0.0,
        extent=float(max(np.linalg.norm(xipos - center, axis=1).max(), 1e-5)) if m.nbody > 1 else 1.0,
        center=center,
    )
    # tendons at qpos0: length, spring rest lengths (springlength -1 = the reference length), invweight0 = J M^-1 J^T
    ten_J = np.zeros((m.ntendon, nv))
    for t in range(m.ntendon):
        a, n = int(m.tendon_adr[t]), int(m.tendon_num[t])
        if n and int(m.wrap_type[a]) != 1:
            # a spatial tendon (mj_tendon at qpos0): over site-only paths the length is the sum of the segment lengths and the Jacobian the sum of
            # (J_site[k+1] - J_site[k]) . direction; paths with wrapping geoms or pulleys are left at zero here -- device_put refuses them anyway
            if all(int(m.wrap_type[w_]) == 3 for w_ in range(a, a + n)):
                site_x = lambda sid: xpos[int(m.site_bodyid[sid])] + xmat[int(m.site_bodyid[sid])] @ m.site_pos[sid]
                for w_ in range(a, a + n - 1):
                    s0, s1 = int(m.wrap_objid[w_]), int(m.wrap_objid[w_ + 1])
                    p0, p1 = site_x(s0), site_x(s1)
                    seg = p1 - p0
                    ln = float(np.linalg.norm(seg))
                    m.tendon_length0[t] += ln
                    if ln > 1e-15 and nv:
                        j0, _ = _body_jac(m, int(m.site_bodyid[s0]), p0, xpos, xmat, xanchor, xaxis)
                        j1, _ = _body_jac(m, int(m.site_bodyid[s1]), p1, xpos, xmat, xanchor, xaxis)
                        ten_J[t] += (seg / ln) @ (j1 - j0)
        else:
          for w_ in range(a, a + n):
            j = int(m.wrap_objid[w_])
            ten_J[t, int(m.jnt_dofadr[j])] += m.wrap_prm[w_]
            m.tendon_length0[t] += m.wrap_prm[w_] * m.qpos0[int(m.jnt_qposadr[j])]
        if m.tendon_lengthspring[t, 0] == -1 and m.tendon_lengthspring[t, 1] == -1:
            m.tendon_lengthspring[t] = m.tendon_length0[t]
        m.tendon_invweight0[t] = float(ten_J[t] @ Minv @ ten_J[t]) if nv else 0.0
    # actuator_acc0 = || M^-1 moment ||
    for i in range(m.nu):
        mom = np.zeros(nv)
        j = int(m.actuator_trnid[i, 0])
        if int(m.actuator_trntype[i]) == int(TrnType.TENDON):
            m.actuator_acc0[i] = float(np.linalg.norm(Minv @ (ten_J[j] * m.actuator_gear[i, 0])))
            continue
        d = int(m.jnt_dofadr[j])
        jt = int(m.jnt_type[j])
        w = JointType(jt).dof_width()
        if jt in (JointType.SLIDE, JointType.HINGE):
            mom[d] = m.actuator_gear[i, 0]
        else:
            mom[d : d + w] = m.actuator_gear[i, :w]
        m.actuator_acc0[i] = float(np.linalg.norm(Minv @ mom))
    # subtree COM per body at qpos0 (for cam/light *com0 offsets)
    mass = m.body_mass.copy()
    mpos = mass[:, None] * xipos
    for b in range(m.nbody - 1, 0, -1):
        p = int(m.body_parentid[b])
        mass[p] += mass[b]
        mpos[p] += mpos[b]
    sub = np.where(mass[:, None] > mjMINVAL, mpos / np.maximum(mass[:, None], mjMINVAL), xipos)
    m._subtree_com0 = sub
    cx = np.stack([xpos[b] + xmat[b] @ m.cam_pos[i] for i, b in enumerate(m.cam_bodyid)]) if m.ncam else np.zeros((0, 3))
    m.cam_pos0 = np.stack([cx[i] - xpos[b] for i, b in enumerate(m.cam_bodyid)]) if m.ncam else np.zeros((0, 3))
    m.cam_poscom0 = np.stack([cx[i] - sub[b] for i, b in enumerate(m.cam_bodyid)]) if m.ncam else np.zeros((0, 3))
    m.cam_mat0 = np.stack([(_quat_to_mat(_quat_mul(xquat[b], m.cam_quat[i]))).reshape(9) for i, b in enumerate(m.cam_bodyid)]) if m.ncam else np.zeros((0, 9))
    lx = np.stack([xpos[b] + xmat[b] @ m.light_pos[i] for i, b in enumerate(m.light_bodyid)]) if m.nlight else np.zeros((0, 3))
    m.light_pos0 = np.stack([lx[i] - xpos[b] for i, b in enumerate(m.light_bodyid)]) if m.nlight else np.zeros((0, 3))
    m.light_poscom0 = np.stack([lx[i] - sub[b] for i, b in enumerate(m.light_bodyid)]) if m.nlight else np.zeros((0, 3))
    m.light_dir0 = np.stack([xmat[b] @ m.light_dir[i] for i, b in enumerate(m.light_bodyid)]) if m.nlight else np.zeros((0, 3))


def _equality_set0(m):
    """Reference-pose parts of eq_data (mj_setConst): the second anchor of body-form connect / weld and the weld relpose."""
    if not m.neq:
        return
    xpos, xquat, xmat, *_ = _kinematics0(m, m.qpos0)
    for i in range(m.neq):
        if int(m.eq_objtype[i]) != 1:
            continue
        b1, b2 = int(m.eq_obj1id[i]), int(m.eq_obj2id[i])
        if int(m.eq_type[i]) == 0:  # connect: data[0:3] is in body1, data[3:6] the same point in body2
            glob = xpos[b1] + xmat[b1] @ m.eq_data[i, 0:3]
            m.eq_data[i, 3:6] = xmat[b2].T @ (glob - xpos[b2])
        elif np.any(m.eq_data[i, 6:10] != 0):  # weld with an explicit relpose: nothing to derive
            continue
        else:  # weld: data[0:3] is in body2, data[3:6] the same point in body1; relpose = body2 in the frame of body1
            glob = xpos[b2] + xmat[b2] @ m.eq_data[i, 0:3]
            m.eq_data[i, 3:6] = xmat[b1].T @ (glob - xpos[b1])
            q1 = xquat[b1] * np.array([1.0, -1, -1, -1])
            m.eq_data[i, 6:10] = _quat_mul(q1, xquat[b2])


# --------------------------------------------------------------------------
# public entry points
# --------------------------------------------------------------------------


_XML_COMMENT = re.compile(r"<!--.*?-->", re.S)


def _expand_includes(node, cur_dir, main_dir, seen):
    """<include file=...>: the children of the included document's <mujoco> root take the place of the element (MuJoCo XML reference).  The path is looked up
    relative to the INCLUDING file, then relative to the main model file; a missing file, a repeated file or a non-MJCF root is an error, never a silent skip."""
    for i, child in enumerate(list(node)):
        if child.tag != "include":
            _expand_includes(child, cur_dir, main_dir, seen)
            continue
        name = child.get("file")
        if not name:
            raise ValueError("<include> needs a file attribute")
        cands = [os.path.join(cur_dir, name), os.path.join(main_dir, name)]
        path = next((c for c in cands if os.path.isfile(c)), None)
        if path is None:
            raise FileNotFoundError(f"<include file={name!r}>: not found (looked in {cur_dir} and {main_dir})")
        real = os.path.realpath(path)
        if real in seen:
            raise ValueError(f"<include file={name!r}>: the file is included more than once")
        seen.add(real)
        with open(path) as f:
            sub = ET.fromstring(_XML_COMMENT.sub("", f.read()))
        if sub.tag != "mujoco":
            raise ValueError(f"<include file={name!r}>: the root element must be <mujoco>, got <{sub.tag}>")
        _expand_includes(sub, os.path.dirname(os.path.abspath(path)), main_dir, seen)
        pos = list(node).index(child)
        node.remove(child)
        for k, c in enumerate(list(sub)):
            node.insert(pos + k, c)


def from_xml_string(xml: str, base_dir: str = ".") -> MjModelLite:
    # MuJoCo's parser tolerates "--" inside comments (ASCII tables in bundled models); expat does not: drop comments first
    root = ET.fromstring(_XML_COMMENT.sub("", xml))
    if root.tag != "mujoco":
        raise ValueError(f"not an MJCF document: root element <{root.tag}>")
    _expand_includes(root, base_dir, base_dir, set())
    return _Compiler(root, base_dir).build()


def from_xml_path(path: str) -> MjModelLite:
    with open(path) as f:
        xml = f.read()
    return from_xml_string(xml, os.path.dirname(os.path.abspath(path)))
