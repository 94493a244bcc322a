"""``make_data``: allocate a ``Data`` with the reference's static shapes and init values.

Mirrors reference ``_src/io.py:29-221``: every leaf is zero except ``qpos = qpos0``, the
model-constant contact leaves (baked from the static contact table, io.py:134-147) and the
constant actuator moment rows (device.py:588-629 / 887-911).  Like the reference, leaves are
float64 regardless of the model dtype; use ``Data.to(torch.float32)`` for a float32 run.
"""

from __future__ import annotations

import numpy as np
import torch

from .container import UnbatchedTensor
from .device import static_contact_fields
from .types import Contact, Data, Model

DEFAULT_DTYPE = torch.float64


def model_float_leaves(m: Model, dtype):
    names = ("geom_friction geom_solref geom_solimp geom_solmix geom_margin geom_gap pair_friction pair_solref "
             "pair_solreffriction pair_solimp pair_margin pair_gap").split()
    return {n: getattr(m, n).to(dtype=dtype, device="cpu") for n in names}


def is_sparse(m: Model) -> bool:
    from ._enums import JacobianType

    if int(m.opt.jacobian) == JacobianType.AUTO:
        return m.nv >= 60
    return int(m.opt.jacobian) == JacobianType.SPARSE


def actuator_static_moment(m: Model):
    mom = torch.zeros((m.nu, m.nv), dtype=DEFAULT_DTYPE)
    for i, (trntype, trnid, jnt_type, dofadr, qposadr) in enumerate(m.actuator_info):
        width = {0: 6, 1: 3}.get(int(jnt_type), 1)  # free / ball joints take the gear vector (smooth.py:565-583)
        mom[i, dofadr : dofadr + width] = m.actuator_gear[i, :width].to(DEFAULT_DTYPE)
    return mom


def make_data(m: Model) -> Data:
    if not isinstance(m, Model):
        from .device import device_put

        m = device_put(m)
    T = m.tables
    ne, nf, nl, ncon, nefc = m.constraint_sizes_py
    z = lambda *s, dt=DEFAULT_DTYPE: torch.zeros(s, dtype=dt)
    mdtype = m.qpos0.dtype
    st = static_contact_fields(model_float_leaves(m, mdtype), T, mdtype)
    g1 = torch.as_tensor(T.con_geom1, dtype=torch.int64)
    g2 = torch.as_tensor(T.con_geom2, dtype=torch.int64)
    con_dim = torch.as_tensor(T.con_dim, dtype=torch.int32)
    if getattr(T, "topk", False):  # max_contact_points: which contacts are kept is decided per step, nothing to bake (collision_driver.py:709-713)
        st = {k: torch.zeros((ncon,) + tuple(v.shape[1:]), dtype=v.dtype) for k, v in st.items()}
        g1, g2, con_dim = torch.zeros(ncon, dtype=torch.int64), torch.zeros(ncon, dtype=torch.int64), con_dim[:ncon]
    contact = Contact(
        dist=z(ncon), pos=z(ncon, 3), frame=z(ncon, 3, 3),
        includemargin=st["includemargin"].to(DEFAULT_DTYPE),
        friction=st["friction"].to(DEFAULT_DTYPE),
        solref=st["solref"].to(DEFAULT_DTYPE),
        solreffriction=st["solreffriction"].to(DEFAULT_DTYPE),
        solimp=st["solimp"].to(DEFAULT_DTYPE),
        contact_dim=con_dim.clone(),
        geom1=g1, geom2=g2, geom=torch.stack([g1, g2], dim=-1) if ncon else torch.zeros((0, 2), dtype=torch.int64),
        efc_address=torch.as_tensor(T.con_efc_address, dtype=torch.int64).clone(),
        batch_size=[ncon],
    )
    nM = (m.nv, m.nv) if not is_sparse(m) else (m.nM,)
    d = Data(
        solver_niter=torch.tensor(0, dtype=torch.int32),
        ne=torch.tensor(ne, dtype=torch.int32),
        nf=torch.tensor(nf, dtype=torch.int32),
        nl=torch.tensor(nl, dtype=torch.int32),
        nefc=UnbatchedTensor(torch.tensor(nefc, dtype=torch.int32)),
        ncon=UnbatchedTensor(torch.tensor(ncon, dtype=torch.int32)),
        time=z(), qpos=m.qpos0.to(dtype=DEFAULT_DTYPE, device="cpu").clone(), qvel=z(m.nv), act=z(m.na),
        qacc_warmstart=z(m.nv), ctrl=z(m.nu), qfrc_applied=z(m.nv), xfrc_applied=z(m.nbody, 6),
        eq_active=torch.as_tensor(np.asarray(m.eq_active0), dtype=torch.int32).reshape(m.neq).clone(), mocap_pos=z(m.nmocap, 3), mocap_quat=z(m.nmocap, 4),
        qacc=z(m.nv), act_dot=z(m.na), userdata=z(getattr(m, "nuserdata", 0) or 0), sensordata=z(m.nsensordata),
        xpos=z(m.nbody, 3), xquat=z(m.nbody, 4), xmat=z(m.nbody, 3, 3), xipos=z(m.nbody, 3),
        ximat=z(m.nbody, 3, 3), xanchor=z(m.njnt, 3), xaxis=z(m.njnt, 3), ten_length=z(m.ntendon),
        geom_xpos=z(m.ngeom, 3), geom_xmat=z(m.ngeom, 3, 3), site_xpos=z(m.nsite, 3), site_xmat=z(m.nsite, 3, 3),
        cam_xpos=z(m.ncam, 3), cam_xmat=z(m.ncam, 3, 3), light_xpos=z(m.nlight, 3), light_xdir=z(m.nlight, 3),
        subtree_com=z(m.nbody, 3), cdof=z(m.nv, 6), cinert=z(m.nbody, 10), crb=z(m.nbody, 10),
        actuator_length=z(m.nu), actuator_moment=actuator_static_moment(m), qM=z(*nM), qLD=z(*nM), qLDiagInv=z(m.nv),
        ten_wrapadr=torch.zeros(m.ntendon, dtype=torch.int32), ten_wrapnum=torch.zeros(m.ntendon, dtype=torch.int32),
        ten_J=z(m.ntendon, m.nv), ten_velocity=z(m.ntendon), wrap_obj=torch.zeros((m.nwrap, 2), dtype=torch.int32),
        wrap_xpos=z(m.nwrap, 6), contact=contact, efc_type=torch.zeros(nefc, dtype=torch.int32),
        efc_J=z(nefc, m.nv), efc_pos=z(nefc), efc_margin=z(nefc), efc_frictionloss=z(nefc), efc_D=z(nefc),
        efc_aref=z(nefc), efc_force=z(nefc), actuator_velocity=z(m.nu), cvel=z(m.nbody, 6), cdof_dot=z(m.nv, 6),
        qfrc_bias=z(m.nv), qfrc_gravcomp=z(m.nv), qfrc_fluid=z(m.nv), qfrc_passive=z(m.nv), actuator_force=z(m.nu),
        qfrc_actuator=z(m.nv), qfrc_smooth=z(m.nv), qacc_smooth=z(m.nv), qfrc_constraint=z(m.nv),
        qfrc_inverse=z(m.nv), cacc=z(m.nbody, 6), cfrc_int=z(m.nbody, 6), cfrc_ext=z(m.nbody, 6),
        subtree_linvel=z(m.nbody, 3), subtree_angmom=z(m.nbody, 3),
        batch_size=[],
    )
    if m.nmocap > 0:  # mocap bodies start at their model pose (reference io.py:208-219, as the C library does)
        ids = np.asarray(m.body_mocapid)
        mask = ids >= 0
        mp, mq = z(m.nmocap, 3), z(m.nmocap, 4)
        mp[ids[mask]] = m.body_pos.to(DEFAULT_DTYPE).cpu()[mask]
        mq[ids[mask]] = m.body_quat.to(DEFAULT_DTYPE).cpu()[mask]
        d = d.replace(mocap_pos=mp, mocap_quat=mq)
    return d
