"""TEST-ONLY stand-in for the device library: lets the CPU suite drive the product's HOST logic (``forward.py``: output slab
and lazy leaves, pointer tables, size validation, model-edit detection in ``native.get_native_model``) end to end without a GPU.

The stand-in answers ``mjh_step`` / ``mjh_forward`` / ``mjh_reset_where`` by handing the very same pointer structs to the CPU
oracle (host pointers instead of device pointers).  Nothing here is importable from the package; the product path still raises
on CPU tensors unless a test installs this through the ``hostsim`` fixture (monkeypatch, undone at teardown).
"""
import ctypes

import numpy as np
import torch

import pyoracle
import sys

from mujoco_torch_amd import native

forward = sys.modules["mujoco_torch_amd.forward"]  # the package attribute `forward` is the function


def leaf_counts(desc):
    """Per-environment element counts in ABI order, restated from the header comments (the device library reports its own
    through mjh_model_leaf_counts; tests/test_gpu_parity.py checks the two agree)."""
    m = desc
    nq, nv, nu, na, nb, nj, ng = m.nq, m.nv, m.nu, m.na, m.nbody, m.njnt, m.ngeom
    ncon, nefc = m.ncon, m.nefc
    c = dict(time=1, qpos=nq, qvel=nv, act=na, qacc_warmstart=nv, ctrl=nu, qfrc_applied=nv, xfrc_applied=nb * 6, mocap_pos=m.nmocap * 3,
             mocap_quat=m.nmocap * 4, qacc=nv, act_dot=na, xpos=nb * 3, xquat=nb * 4, xmat=nb * 9, xipos=nb * 3, ximat=nb * 9, xanchor=nj * 3,
             xaxis=nj * 3, geom_xpos=ng * 3, geom_xmat=ng * 9, site_xpos=m.nsite * 3, site_xmat=m.nsite * 9, cam_xpos=m.ncam * 3,
             cam_xmat=m.ncam * 9, light_xpos=m.nlight * 3, light_xdir=m.nlight * 3, subtree_com=nb * 3, cdof=nv * 6, cinert=nb * 10,
             crb=nb * 10, ten_length=m.ntendon, ten_J=m.ntendon * nv, ten_velocity=m.ntendon, actuator_length=nu, actuator_moment=nu * nv,
             qM=nv * nv, qLD=nv * nv, contact_dist=ncon, contact_pos=ncon * 3, contact_frame=ncon * 9, contact_includemargin=ncon,
             contact_friction=ncon * 5, contact_solref=ncon * 2, contact_solreffriction=ncon * 2, contact_solimp=ncon * 5,
             sensordata=m.nsensordata, efc_J=nefc * nv, efc_frictionloss=nefc, efc_D=nefc, efc_aref=nefc, efc_force=nefc,
             actuator_velocity=nu, cvel=nb * 6, cdof_dot=nv * 6, qfrc_bias=nv, qfrc_passive=nv, qfrc_gravcomp=nv, actuator_force=nu,
             qfrc_actuator=nv, qfrc_smooth=nv, qacc_smooth=nv, qfrc_constraint=nv, contact_dim=ncon, eq_active=m.neq, contact_geom1=ncon,
             contact_geom2=ncon, contact_geom=2 * ncon, contact_efc_address=ncon)
    return np.array([c[n] for n in forward._ALL_NAMES], dtype=np.int64)


class _Lib:
    def __init__(self, owner):
        self.o = owner
        self.calls = 0

    def mjh_step(self, handle, pin, pout, work, B, flags, stream):
        self.calls += 1
        return pyoracle.lib().mjo_step(ctypes.byref(self.o.desc), pin, pout, B, self.o.dt, flags, 1, None, -1)

    def mjh_forward(self, handle, pin, pout, work, B, stages, flags, stream):
        self.calls += 1
        return pyoracle.lib().mjo_forward(ctypes.byref(self.o.desc), pin, pout, B, self.o.dt, stages, flags, 1, None, -1)

    def mjh_last_error(self):
        return b"hostsim"


REAL_NATIVE_MODEL = native.NativeModel  # the product's class (its host-only methods, e.g. the workspace pool, are tested on the stand-in's instances)


class HostSimModel:
    """What ``native.NativeModel`` is to the device: built from the packed descriptor, one per distinct set of model values."""

    built = 0

    def __init__(self, desc, keep, device, dtype):
        HostSimModel.built += 1
        self.desc, self.keep = desc, keep
        self.dt = 0 if dtype == torch.float64 else 1
        self.device, self.dtype = device, dtype
        self.handle = None
        self.leaf_counts = leaf_counts(desc)
        self.work_bytes = 0
        self.lib = _Lib(self)

    def workspace(self, B, stream=0):
        return None


def install(monkeypatch):
    monkeypatch.setattr(native, "NativeModel", HostSimModel)
    monkeypatch.setattr(forward, "_require_device", lambda device: None)
    return HostSimModel
