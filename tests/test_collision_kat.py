"""Known-answer tests of the narrow phase, taken from the reference's own collision tests
(reference test/collision_driver_test.py): the scenes are its MJCF fixtures and the expected numbers are the ones it
asserts (they come from the MuJoCo C library there).  Checked on the CPU oracle here (stages = kinematics + collision),
and on the HIP path in tests/test_gpu_parity.py::test_collision_kat_gpu.  Also pins the hull topology
(reference test/mesh_test.py:46-67)."""
import numpy as np
import pytest
import torch

import mujoco_torch_amd as mt
import pyoracle
from mujoco_torch_amd import convex

STAGES_COLLISION = 0x07  # kinematics, com_pos/crb prefix, collision (include/mjhip.h MJH_STAGE_*)

BOX_PLANE = """<mujoco><worldbody><geom size="40 40 40" type="plane"/>
  <body pos="0 0 0.7" euler="45 0 0"><joint axis="1 0 0" type="free"/><geom size="0.5 0.5 0.5" type="box"/></body>
</worldbody></mujoco>"""
BOX_BOX = """<mujoco><worldbody>
  <body pos="0.0 1.0 0.2"><joint axis="1 0 0" type="free"/><geom size="0.2 0.2 0.2" type="box"/></body>
  <body pos="0.1 1.0 0.495" euler="0.1 -0.1 0"><joint axis="1 0 0" type="free"/><geom size="0.1 0.1 0.1" type="box"/></body>
</worldbody></mujoco>"""
BOX_BOX_EDGE = """<mujoco><worldbody>
  <body pos="-1.0 -1.0 0.2"><joint axis="1 0 0" type="free"/><geom size="0.2 0.2 0.2" type="box"/></body>
  <body pos="-1.0 -1.2 0.55" euler="0 45 30"><joint axis="1 0 0" type="free"/><geom size="0.1 0.1 0.1" type="box"/></body>
</worldbody></mujoco>"""
CAP_BOX = """<mujoco><worldbody>
  <body pos="0 0 0.54"><joint axis="1 0 0" type="free"/><geom fromto="-0.4 0 0 0.4 0 0" size="0.05" type="capsule"/></body>
  <body><joint axis="1 0 0" type="free"/><geom size="0.5 0.5 0.5" type="box"/></body>
</worldbody></mujoco>"""
CAP_EDGE_BOX = """<mujoco><worldbody>
  <body pos="0.5 0 0.55" euler="0 30 0"><joint axis="1 0 0" type="free"/><geom fromto="-0.6 0 0 0.6 0 0" size="0.05" type="capsule"/></body>
  <body><joint axis="1 0 0" type="free"/><geom size="0.5 0.5 0.5" type="box"/></body>
</worldbody></mujoco>"""
PARALLEL_CAP = """<mujoco><worldbody>
  <body><joint type="free"/><geom fromto="-0.5 0.1 0.25 0.5 0.1 0.25" size="0.1" type="capsule"/></body>
  <body><joint type="free"/><geom fromto="-0.5 0.1 0.1 0.5 0.1 0.1" size="0.1" type="capsule"/></body>
</worldbody></mujoco>"""
SPHERE_BOX = """<mujoco><worldbody>
  <body pos="0 0 0.58"><joint type="free"/><geom size="0.1" type="sphere"/></body>
  <body><joint type="free"/><geom size="0.5 0.5 0.5" type="box"/></body>
</worldbody></mujoco>"""


def collide(xml, runner=None):
    lite = mt.mjcf.from_xml_string(xml)
    mx = mt.device_put(lite)
    d = mt.make_data(mx)
    out = (runner or (lambda m, dd: pyoracle.run(m, dd, step=False, stages=STAGES_COLLISION)))(mx, d)
    ncon = mx.constraint_sizes_py[3]
    return out["contact_dist"].reshape(ncon), out["contact_pos"].reshape(ncon, 3), out["contact_frame"].reshape(ncon, 3, 3)


def check_box_plane(dist, pos, frame):  # ConvexTest.test_box_plane: two edge corners penetrate, the other two slots are inactive
    assert (dist[:2] < 0).all() and (dist[2:] > 0).all()
    np.testing.assert_allclose(dist[:2], 0.7 - 0.5 * np.sqrt(2.0), atol=1e-6)  # corner height of a box rolled by 45 degrees
    np.testing.assert_allclose(frame[:, 0], [[0, 0, 1]] * 4, atol=1e-12)
    np.testing.assert_allclose(np.sort(pos[:2, 0]), [-0.5, 0.5], atol=1e-6)


def check_box_box(dist, pos, frame):  # ConvexTest.test_box_box: face contact, four points at z ~ 0.39, normal +z
    assert dist.shape == (4,) and (dist < 0).all()
    np.testing.assert_array_almost_equal(pos[:, 2], [0.39] * 4, 2)
    np.testing.assert_array_almost_equal(frame[:, 0, :], [[0.0, 0.0, 1.0]] * 4)


def check_box_box_edge(dist, pos, frame):  # ConvexTest.test_box_box_edge: exactly one contact point
    assert dist[0] < 0 and (dist[1:] > 0).all()


def check_cap_box(dist, pos, frame):  # CapsuleCollisionTest.test_capsule_convex: face contact, both capsule ends, depth 0.01
    np.testing.assert_allclose(dist, [-0.01, -0.01], atol=1e-5)
    np.testing.assert_allclose(np.abs(frame[:, 0, 2]), [1, 1], atol=1e-6)
    np.testing.assert_allclose(np.sort(pos[:, 0]), [-0.4, 0.4], atol=1e-4)


def check_cap_edge_box(dist, pos, frame):  # test_capsule_convex_edge: one penetrating point, the second slot is inactive
    assert dist.shape == (2,) and dist[0] < 0 and dist[1] > 0


def check_parallel_cap(dist, pos, frame):  # test_parallel_capsules: dist -0.05, midpoint contact, normal -z
    np.testing.assert_allclose(dist, [-0.05], atol=1e-12)
    np.testing.assert_allclose(pos[0], [0.0, 0.1, (0.15 + 0.2) / 2.0], atol=1e-5)
    np.testing.assert_allclose(frame[0, 0, :], [0, 0.0, -1.0], atol=1e-5)


def check_sphere_box(dist, pos, frame):  # analytic: sphere of radius 0.1 centred 0.58 above a box top at 0.5
    np.testing.assert_allclose(dist, [-0.02], atol=1e-9)
    np.testing.assert_allclose(pos[0], [0, 0, 0.49], atol=1e-9)
    np.testing.assert_allclose(np.abs(frame[0, 0]), [0, 0, 1], atol=1e-9)


KATS = {
    "box_plane": (BOX_PLANE, check_box_plane), "box_box": (BOX_BOX, check_box_box), "box_box_edge": (BOX_BOX_EDGE, check_box_box_edge),
    "capsule_box_face": (CAP_BOX, check_cap_box), "capsule_box_edge": (CAP_EDGE_BOX, check_cap_edge_box),
    "parallel_capsules": (PARALLEL_CAP, check_parallel_cap), "sphere_box": (SPHERE_BOX, check_sphere_box),
}


@pytest.mark.parametrize("name", sorted(KATS))
def test_collision_kat_oracle(name, oracle_lib):
    xml, check = KATS[name]
    check(*collide(xml))


def test_hull_topology():
    """box: 8 vertices, 6 quads, 12 edges; dodecahedron: 20 vertices, 12 pentagons, 30 edges; unit outward normals."""
    import os

    t = convex.tables_from_points(np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=float) * [0.1, 0.2, 0.3])
    assert t["vert"].shape == (8, 3) and t["face"].shape == (6, 4) and t["edge"].shape == (12, 2)
    lite = mt.mjcf.from_xml_path(mt.test_data_path("mesh_contact.xml"))
    dod = convex.geom_convex_tables(lite)[2]
    assert dod["vert"].shape == (20, 3) and dod["face"].shape == (12, 5) and dod["edge"].shape == (30, 2)
    for tab in (t, dod):
        n = tab["facenormal"]
        np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
        centroid = tab["vert"].mean(0)
        for f, nf in zip(tab["face"], n):  # outward: the face centre is on the positive side of its normal
            assert np.dot(tab["vert"][f].mean(0) - centroid, nf) > 0
        assert len({tuple(e) for e in tab["edge"]}) == len(tab["edge"])


# ---- dynamics known answers (reference test/smooth_test.py:181-208, test/forward_test.py:128-140) --------------------
FREE_BODY = """<mujoco><option timestep="0.01"/><worldbody><body><joint type="free"/><geom size="0.1"/></body></worldbody></mujoco>"""


def test_free_fall_one_step_oracle(oracle_lib):
    """One 0.01 s semi-implicit Euler step of a free body: z = -9.81e-4; with gravity disabled nothing moves."""
    from mujoco_torch_amd._enums import DisableBit

    lite = mt.mjcf.from_xml_string(FREE_BODY)
    mx = mt.device_put(lite)
    out = pyoracle.run(mx, mt.make_data(mx), step=True)
    np.testing.assert_array_almost_equal(out["qpos"], [0.0, 0.0, -9.81e-4, 1.0, 0.0, 0.0, 0.0], decimal=7)
    lite.opt.disableflags = int(lite.opt.disableflags) | int(DisableBit.GRAVITY)
    mx = mt.device_put(lite)
    out = pyoracle.run(mx, mt.make_data(mx), step=True)
    np.testing.assert_equal(out["qpos"], [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])


def test_euler_without_eulerdamp_oracle(oracle_lib):
    """With EULERDAMP disabled the Euler update is exactly qvel + dt * qacc even when dofs are damped: a free body with
    joint damping falls like an undamped one for its first step from rest (qacc = g, qvel = g dt)."""
    from mujoco_torch_amd._enums import DisableBit

    xml = FREE_BODY.replace('<joint type="free"/>', '<joint type="free" damping="3"/>')
    lite = mt.mjcf.from_xml_string(xml)
    lite.opt.disableflags = int(lite.opt.disableflags) | int(DisableBit.EULERDAMP)
    mx = mt.device_put(lite)
    d = mt.make_data(mx).replace(qvel=torch.tensor([0.0, 0.0, 1.0, 0.0, 0.0, 0.0], dtype=torch.float64))
    out = pyoracle.run(mx, d, step=True)
    # passive damping force -3 * 1 on a 4.19 kg sphere: qacc = -9.81 - 3 / m; explicit update, no implicit damping
    mass = float(mx.body_mass[1])
    np.testing.assert_allclose(out["qvel"][2], 1.0 + 0.01 * (-9.81 - 3.0 / mass), rtol=1e-12)
