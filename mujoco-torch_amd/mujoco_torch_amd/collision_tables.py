"""Static collision / constraint tables (host side, setup time, integer-exact).

Restates the model-only part of the reference's collision driver:
``collision_candidates`` (_src/collision_driver.py:581-615), ``_add_candidate`` (:135-174),
``_body_pair_filter`` (:299-315), ``precompute_collision_indices`` (:440-497, parameter-source
grouping order), ``make_condim`` (:618-644), the static contact parameters
``_pair_params/_priority_params/_dynamic_params`` (:177-257) and the contact ordering
(``torch.argsort(contact_dim)`` WITHOUT ``stable=True``, :764 and :842).

The reference has no runtime broad phase: every statically filtered geom pair is narrow-phased
every step, so the contact list -- which pair produces which contact slot, with which condim
and solver parameters -- is a pure function of the model.  The kernels therefore receive it as
a table.  The unstable argsort is reproduced by calling the very same torch op on the CPU at
``device_put`` time (hard part 2), never by sorting on the device.
"""

from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch

from ._enums import ConeType, DisableBit, GeomType, mjMINMU, mjMINVAL

# (type1, type2) -> (fn id in include/mjhip.h, contacts per pair)   collision_driver.py:106-125
FN_PLANE_SPHERE, FN_PLANE_CAPSULE, FN_SPHERE_SPHERE, FN_SPHERE_CAPSULE, FN_CAPSULE_CAPSULE = 0, 1, 2, 3, 4
FN_PLANE_CONVEX, FN_SPHERE_CONVEX, FN_CAPSULE_CONVEX, FN_CONVEX_CONVEX = 5, 6, 7, 8
_G = GeomType
COLLISION_FN = {
    (_G.PLANE, _G.SPHERE): (FN_PLANE_SPHERE, 1),
    (_G.PLANE, _G.CAPSULE): (FN_PLANE_CAPSULE, 2),
    (_G.PLANE, _G.BOX): (FN_PLANE_CONVEX, 4),
    (_G.PLANE, _G.MESH): (FN_PLANE_CONVEX, 4),
    (_G.SPHERE, _G.SPHERE): (FN_SPHERE_SPHERE, 1),
    (_G.SPHERE, _G.CAPSULE): (FN_SPHERE_CAPSULE, 1),
    (_G.SPHERE, _G.BOX): (FN_SPHERE_CONVEX, 1),
    (_G.SPHERE, _G.MESH): (FN_SPHERE_CONVEX, 1),
    (_G.CAPSULE, _G.CAPSULE): (FN_CAPSULE_CAPSULE, 1),
    (_G.CAPSULE, _G.BOX): (FN_CAPSULE_CONVEX, 2),
    (_G.CAPSULE, _G.MESH): (FN_CAPSULE_CONVEX, 2),
    (_G.BOX, _G.BOX): (FN_CONVEX_CONVEX, 4),
    (_G.BOX, _G.MESH): (FN_CONVEX_CONVEX, 4),
    (_G.MESH, _G.MESH): (FN_CONVEX_CONVEX, 4),
}
# hfield pairs exist in the reference table but are outside this build's scope
_HFIELD_KEYS = {(_G.HFIELD, t) for t in (_G.SPHERE, _G.CAPSULE, _G.BOX, _G.MESH)}


@dataclass(frozen=True)
class Candidate:
    geom1: int
    geom2: int
    ipair: int
    geomp: int
    dim: int


def _body_pair_filter(m, b1: int, b2: int) -> bool:
    weld1, weld2 = int(m.body_weldid[b1]), int(m.body_weldid[b2])
    parent_weld1 = int(m.body_weldid[m.body_parentid[weld1]])
    parent_weld2 = int(m.body_weldid[m.body_parentid[weld2]])
    if weld1 == weld2:
        return True
    filterparent = not (int(m.opt.disableflags) & DisableBit.FILTERPARENT)
    if filterparent and weld1 != 0 and weld2 != 0 and (weld1 == parent_weld2 or weld2 == parent_weld1):
        return True
    return False


def collision_candidates(m, convex_shape_key=None) -> "OrderedDict[tuple, list[Candidate]]":
    """Ordered candidate set; key = (type1, type2, shapekey1, shapekey2) in insertion order."""
    result: OrderedDict = OrderedDict()

    def mesh_key(g):
        return convex_shape_key(g) if convex_shape_key is not None else (-1,)

    def add(g1, g2, ipair=-1):
        g1, g2, ipair = int(g1), int(g2), int(ipair)
        t1, t2 = int(m.geom_type[g1]), int(m.geom_type[g2])
        if t1 > t2:
            t1, t2, g1, g2 = t2, t1, g2, g1
        key = (t1, t2, mesh_key(g1), mesh_key(g2))
        if any((c.geom1, c.geom2) == (g1, g2) for c in result.get(key, [])):
            return
        if ipair > -1:
            cand = Candidate(g1, g2, ipair, -1, int(m.pair_dim[ipair]))
        elif m.geom_priority[g1] != m.geom_priority[g2]:
            gp = g1 if m.geom_priority[g1] > m.geom_priority[g2] else g2
            cand = Candidate(g1, g2, -1, int(gp), int(m.geom_condim[gp]))
        else:
            cand = Candidate(g1, g2, -1, -1, int(max(m.geom_condim[g1], m.geom_condim[g2])))
        result.setdefault(key, []).append(cand)

    for ipair in range(int(m.npair)):
        add(m.pair_geom1[ipair], m.pair_geom2[ipair], ipair)
    exclude = set(int(s) for s in m.exclude_signature)
    for b1 in range(int(m.nbody)):
        for b2 in range(b1, int(m.nbody)):
            if ((b1 << 16) + b2) in exclude or _body_pair_filter(m, b1, b2):
                continue
            for g1 in range(int(m.body_geomadr[b1]), int(m.body_geomadr[b1]) + int(m.body_geomnum[b1])) if m.body_geomnum[b1] else []:
                for g2 in range(int(m.body_geomadr[b2]), int(m.body_geomadr[b2]) + int(m.body_geomnum[b2])) if m.body_geomnum[b2] else []:
                    mask = int(m.geom_contype[g1]) & int(m.geom_conaffinity[g2])
                    mask |= int(m.geom_contype[g2]) & int(m.geom_conaffinity[g1])
                    if mask != 0:
                        add(g1, g2)
    return result


def _fn_for(key):
    t = (GeomType(key[0]), GeomType(key[1]))
    return COLLISION_FN.get(t)


def validate_candidates(cands):
    for key in cands:
        t1, t2 = GeomType(key[0]), GeomType(key[1])
        if t1 == GeomType.PLANE and t2 in (GeomType.PLANE, GeomType.HFIELD):
            continue  # MuJoCo does not collide planes with planes / hfields
        if (t1, t2) in _HFIELD_KEYS:
            raise NotImplementedError(f"({t1.name}, {t2.name}) collisions are outside this build's scope (hfield).")
        if (t1, t2) not in COLLISION_FN:
            raise NotImplementedError(f"({t1.name}, {t2.name}) collisions not implemented.")


def ordered_pairs(cands):
    """Pairs in pre-sort contact order: group key order, then pair / priority / dynamic source
    groups in first-appearance order, then candidate order (collision_driver.py:446-459)."""
    out = []
    for gi, (key, lst) in enumerate(cands.items()):
        fn = _fn_for(key)
        if fn is None:
            continue
        typ = OrderedDict()
        for c in lst:
            typ.setdefault((c.ipair > -1, c.geomp > -1), []).append(c)
        for t, cs in typ.items():
            for c in cs:
                out.append((fn[0], fn[1], c, (gi, t)))
    return out


def max_contact_points(m) -> int:
    """The custom numeric `max_contact_points`, -1 when absent (reference collision_driver.py:571-578)."""
    names = getattr(m, "names", b"")
    if isinstance(names, str):
        names = names.encode("utf-8")
    for i in range(int(getattr(m, "nnumeric", 0) or 0)):
        name = bytes(names[int(m.name_numericadr[i]):]).decode("utf-8").split("\x00", 1)[0]
        if name == "max_contact_points":
            return int(np.asarray(m.numeric_data)[int(m.numeric_adr[i])])
    return -1


def make_condim(m, cands) -> list:
    """Per-contact condim, ascending, capped at max_contact_points (reference collision_driver.py:618-644)."""
    if int(m.opt.disableflags) & DisableBit.CONTACT:
        return []
    dims = []
    for key, lst in cands.items():
        fn = _fn_for(key)
        if fn is None:
            continue
        for c in lst:
            dims.extend([c.dim] * fn[1])
    dims = sorted(dims)
    cap = max_contact_points(m)
    if cap > -1 and len(dims) > cap:
        dims = dims[:cap]
    return dims


def constraint_sizes(m, dims) -> tuple:
    """(ne, nf, nl, ncon, nefc), reference device.py:226-264."""
    flags = int(m.opt.disableflags)
    if flags & DisableBit.CONSTRAINT:
        return (0, 0, 0, 0, 0)
    et = np.asarray(getattr(m, "eq_type", np.zeros(0, dtype=np.int32)))
    ne = 0 if flags & DisableBit.EQUALITY else int(3 * (et == 0).sum() + 6 * (et == 1).sum() + (et == 2).sum())
    nf = 0 if flags & DisableBit.FRICTIONLOSS else int((np.asarray(m.dof_frictionloss) > 0).sum()) + int((np.asarray(getattr(m, "tendon_frictionloss", np.zeros(0))) > 0).sum())
    nl = 0 if flags & DisableBit.LIMIT else int(np.asarray(m.jnt_limited).sum()) + int(np.asarray(getattr(m, "tendon_limited", np.zeros(0))).sum())
    if flags & DisableBit.CONTACT:
        ncon, nc = 0, 0
    else:
        ncon = len(dims)
        elliptic = int(m.opt.cone) == ConeType.ELLIPTIC
        nc = sum(1 if d == 1 else (d if elliptic else 2 * (d - 1)) for d in dims)
    return (ne, nf, nl, ncon, ne + nf + nl + nc)


def contact_order(dims_unsorted) -> np.ndarray:
    """The reference's contact permutation: torch.argsort on an int32 CPU tensor, not stable."""
    if not dims_unsorted:
        return np.zeros(0, dtype=np.int64)
    return torch.argsort(torch.tensor(dims_unsorted, dtype=torch.int32)).numpy()


# ---- static contact parameters, evaluated in the model dtype with the reference's op order ----

_FRICTION_IDX = [0, 0, 1, 2, 2]


def static_contact_params(mt, pairs, dtype):
    """Per-pair (friction[5], solref[2], solreffriction[2], solimp[5], includemargin).

    ``mt``: dict of float torch tensors of the model in ``dtype`` (geom_* and pair_* leaves).
    Mirrors the batched evaluation per parameter-source group, including the reference's
    ``solref1[0] > 0`` test on the FIRST ROW of the group (collision_driver.py:243).
    """
    out = [None] * len(pairs)
    # regroup consecutive pairs of the same source type, exactly as the reference batches them
    i = 0
    while i < len(pairs):
        c0 = pairs[i][2]
        typ = (c0.ipair > -1, c0.geomp > -1)
        j = i
        while j < len(pairs) and pairs[j][3] == pairs[i][3]:
            j += 1
        cs = [p[2] for p in pairs[i:j]]
        if typ[0]:
            ip = torch.tensor([c.ipair for c in cs])
            friction = torch.clamp_min(mt["pair_friction"][ip], mjMINMU)
            solref = mt["pair_solref"][ip]
            solreffriction = mt["pair_solreffriction"][ip]
            solimp = mt["pair_solimp"][ip]
            margin, gap = mt["pair_margin"][ip], mt["pair_gap"][ip]
        elif typ[1]:
            gp = torch.tensor([c.geomp for c in cs])
            gpairs = torch.tensor([(c.geom1, c.geom2) for c in cs])
            friction = mt["geom_friction"][gp][:, _FRICTION_IDX]
            solref = mt["geom_solref"][gp]
            solreffriction = torch.zeros((len(cs), 2), dtype=dtype)
            solimp = mt["geom_solimp"][gp]
            margin = torch.amax(mt["geom_margin"][gpairs.T], dim=0)
            gap = torch.amax(mt["geom_gap"][gpairs.T], dim=0)
        else:
            g1 = torch.tensor([c.geom1 for c in cs])
            g2 = torch.tensor([c.geom2 for c in cs])
            friction = torch.maximum(mt["geom_friction"][g1], mt["geom_friction"][g2])[:, _FRICTION_IDX]
            minval = torch.tensor(mjMINVAL, dtype=dtype)
            s1, s2 = mt["geom_solmix"][g1], mt["geom_solmix"][g2]
            mix = s1 / (s1 + s2)
            mix = torch.where((s1 < minval) & (s2 < minval), 0.5, mix)
            mix = torch.where((s1 < minval) & (s2 >= minval), 0.0, mix)
            mix_u = mix.unsqueeze(-1)
            r1, r2 = mt["geom_solref"][g1], mt["geom_solref"][g2]
            solref = torch.minimum(r1, r2)
            s_mix = mix_u * r1 + (1 - mix_u) * r2
            solref = torch.where((r1[0] > 0) & (r2[0] > 0), s_mix, solref)  # row 0, literally
            solreffriction = torch.zeros((len(cs), 2), dtype=dtype)
            solimp = mix_u * mt["geom_solimp"][g1] + (1 - mix_u) * mt["geom_solimp"][g2]
            margin = torch.maximum(mt["geom_margin"][g1], mt["geom_margin"][g2])
            gap = torch.maximum(mt["geom_gap"][g1], mt["geom_gap"][g2])
        im = margin - gap
        for k in range(len(cs)):
            out[i + k] = (friction[k], solref[k], solreffriction[k], solimp[k], im[k])
        i = j
    return out
