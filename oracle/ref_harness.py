"""Import the reference's own Python hot path in the build container.

TEST INFRASTRUCTURE, container-only: `/root/reference` does not exist on the GPU box and the
reference's third-party dependencies (mujoco, tensordict, pyvers, trimesh, etils) are not
installed, so they are replaced by the minimal stand-ins under ``oracle/ref_stubs`` and the
reference package is registered without running its ``__init__`` (which applies functorch
monkey-patches, reference ``mujoco_torch/__init__.py:22-35``).  Everything numerical that runs is
the reference's own code: ``device.device_put``, ``io.make_data``, ``forward.step`` and the
per-stage functions.  ``oracle/gen_golden.py`` uses this to write ``tests/golden/*.npz``.
"""

import importlib
import os
import sys
import types

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "mujoco_torch", "_src"))


_loaded = None


def load():
    """Returns a namespace with the reference modules (types, device, io, forward, ...)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (container-only harness)")
    sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
    pkg_dir = os.path.join(REPO, "mujoco-torch_amd")
    if pkg_dir not in sys.path:
        sys.path.insert(0, pkg_dir)
    # register the package shells so `mujoco_torch/__init__.py` (patches) never runs
    pkg = types.ModuleType("mujoco_torch")
    pkg.__path__ = [os.path.join(REF_ROOT, "mujoco_torch")]
    sys.modules["mujoco_torch"] = pkg
    src = types.ModuleType("mujoco_torch._src")
    src.__path__ = [os.path.join(REF_ROOT, "mujoco_torch", "_src")]
    sys.modules["mujoco_torch._src"] = src
    ns = types.SimpleNamespace()
    for name in (
        "math", "diff_config", "dataclasses", "collision_types", "types", "scan", "support", "smooth",
        "collision_primitive", "collision_convex", "mesh", "collision_hfield", "collision_driver",
        "constraint", "solver", "passive", "derivative", "ray", "sensor", "forward", "device", "io",
    ):
        mod = importlib.import_module(f"mujoco_torch._src.{name}")
        setattr(src, name, mod)
        setattr(ns, name, mod)
    import mujoco  # the stub

    ns.mujoco = mujoco
    _loaded = ns
    return ns


def put_model(ref, lite, dtype=None, keep_sensors=True):
    """reference device_put on an MJCF-subset-compiled model."""
    import copy

    lite = copy.copy(lite)
    lite.opt = copy.copy(lite.opt)
    import numpy as np

    if not keep_sensors:  # goldens recorded before sensors were on the native path keep their recorded (sensor-less) layout
        lite.nsensor = 0
        lite.nsensordata = 0
        for k in ("sensor_type", "sensor_dim", "sensor_adr", "sensor_objid", "sensor_objtype", "sensor_needstage", "sensor_datatype", "sensor_reftype", "sensor_refid"):
            setattr(lite, k, np.zeros(0, dtype=np.int32))
        lite.sensor_cutoff = np.zeros(0)
    mj = ref.mujoco.MjModel(lite)
    return ref.device.device_put(mj, dtype=dtype)
