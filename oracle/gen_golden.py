"""Generate tests/golden/*.npz by running the REFERENCE's own Python step in the build container.

TEST INFRASTRUCTURE, container-only (needs /root/reference; see oracle/ref_harness.py).  For each
case: compile the bundled XML with this repo's MJCF-subset compiler, apply the option overrides
of BASELINE.json's configs, hand the model to the reference's ``device_put`` / ``make_data``, set
seeded inputs, and record every Data leaf the step writes after each of N reference
``forward.step`` calls (eager, one environment at a time -- the reference's per-env semantics).

Run:  python oracle/gen_golden.py            (writes tests/golden/<case>.npz, a few hundred KB each)
"""

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "mujoco-torch_amd"))

import ref_harness  # noqa: E402

from mujoco_torch_amd import mjcf, native  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")

# name -> (xml, option overrides, dtype, nenv, nsteps, input recipe)
CASES = {
    # BASELINE config 1: cartpole, Euler (constraints disabled in the XML), float64
    "cartpole_f64": ("cartpole", {}, "float64", 3, 4, "cartpole"),
    # BASELINE config 2/4: humanoid, Euler + CG, XML iterations=1 / ls_iterations=4, float64
    "humanoid_cg_f64": ("humanoid", {"solver": 1}, "float64", 3, 3, "bench"),
    "humanoid_cg_f64_perturbed": ("humanoid", {"solver": 1}, "float64", 3, 3, "perturbed"),
    # bundled XML as is (Newton, the configuration the reference's README numbers were taken on)
    "humanoid_newton_f64": ("humanoid", {}, "float64", 2, 3, "perturbed"),
    # full-length solver loops (iterations 100 / ls 50) on the humanoid with CG
    "humanoid_cg_iter100_f64": ("humanoid", {"solver": 1, "iterations": 100, "ls_iterations": 50}, "float64", 2, 2, "perturbed"),
    # bundled ant as is (Euler, Newton, pyramidal) and BASELINE config 3 (RK4 + Newton + elliptic)
    "ant_euler_newton_pyr_f64": ("ant", {}, "float64", 2, 3, "bench_ctrl"),
    "ant_rk4_newton_ell_f64": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, "float64", 2, 3, "bench_ctrl"),
    "ant_rk4_newton_ell_f32": ("ant", {"integrator": 1, "solver": 2, "cone": 1}, "float32", 2, 3, "bench_ctrl"),
    "ant_euler_cg_ell_f64": ("ant", {"solver": 1, "cone": 1}, "float64", 2, 2, "bench_ctrl"),
    # BASELINE config 5: plane + free box + free dodecahedron mesh, condim 6, Newton (pyramidal), float32 (+ float64 twin)
    "mesh_contact_newton_f32": ("mesh_contact", {}, "float32", 4, 3, "convex"),
    "mesh_contact_newton_f64": ("mesh_contact", {}, "float64", 4, 3, "convex"),
    "mesh_contact_cg_ell_f64": ("mesh_contact", {"solver": 1, "cone": 1}, "float64", 2, 2, "convex"),
    # every convex pair function: mesh stack (plane/convex-convex) and box / sphere / capsule mixes
    "convex_meshes_f64": ("convex_meshes", {}, "float64", 4, 3, "convex"),
    "convex_primitives_f64": ("convex_primitives", {}, "float64", 4, 3, "convex"),
    "convex_primitives_f32": ("convex_primitives", {}, "float32", 3, 2, "convex"),
    # the other bundled models of the reference (mujoco_torch/test_data/*.xml) that the path supports, as they are
    "halfcheetah_f64": ("halfcheetah", {}, "float64", 2, 3, "generic"),
    "hopper_f64": ("hopper", {}, "float64", 2, 3, "generic"),
    "walker2d_f64": ("walker2d", {}, "float64", 2, 3, "generic"),
    "walker2d_rk4_f32": ("walker2d", {"integrator": 1}, "float32", 2, 2, "generic"),
    "satellite_large_f64": ("satellite_large", {}, "float64", 2, 3, "generic"),
    "satellite_small_f64": ("satellite_small", {}, "float64", 2, 3, "generic"),
    "convex_bundled_f64": ("convex", {}, "float64", 2, 2, "generic"),
    # dof frictionloss rows (constraint.py:215-251, solver.py:326-342, :404-416): the bundled single-hinge model and the ant
    # with frictionloss on every joint, Newton and CG
    # sensors on the step path (sensor.py:56-440, ray.py): velocimeter / gyro / accelerometer / rangefinders against every primitive
    # geom type / joint sensors, with cutoffs
    "sensor_rig_f64": ("sensor_rig", {}, "float64", 4, 3, "sensor_rig"),
    "sensor_rig_rk4_f64": ("sensor_rig", {"integrator": 1}, "float64", 2, 2, "sensor_rig"),
    # ... one of every other type sensor.py evaluates: magnetometer, tendon / actuator / joint-actuator sensors, ball joint sensors, frame sensors in every
    # object x reference kind, subtree sensors, clock, force / torque / accelerometer / subtree momentum reading the Data leaves no stage writes (set by the recipe), and types the
    # reference leaves untouched (touch, jointlimitpos, framelinacc, e_potential)
    "sensor_rig2_f64": ("sensor_rig2", {}, "float64", 4, 3, "sensor_rig2"),
    "sensor_rig2_rk4_f32": ("sensor_rig2", {"integrator": 1}, "float32", 3, 2, "sensor_rig2"),
    # fluid forces (passive.py:31-78, :158-173): the bundled swimmer (density only) and a variant with viscosity and wind
    "swimmer_f64": ("swimmer", {}, "float64", 2, 3, "generic"),
    "swimmer_viscous_wind_f64": ("swimmer", {"viscosity": 0.05, "wind": [0.3, -0.2, 0.1]}, "float64", 2, 2, "generic"),
    "swimmer_rk4_f32": ("swimmer", {"integrator": 1, "viscosity": 0.02}, "float32", 2, 2, "generic"),
    # equality constraints (constraint.py:116-212, 254-296): the bundled model (site-form welds / connect carried inactive, one
    # active joint coupling) and a model with every body-form kind active: closed loop, weld, couplings, an inactive connect
    "equality_f64": ("equality", {}, "float64", 2, 3, "generic"),
    "equality_loops_f64": ("equality_loops", {}, "float64", 3, 3, "generic"),
    "equality_loops_cg_f64": ("equality_loops", {"solver": 1}, "float64", 2, 2, "generic"),
    "equality_loops_rk4_f32": ("equality_loops", {"integrator": 1}, "float32", 2, 2, "generic"),
    # mocap bodies (smooth.py:105-113): a free capsule resting on a mocap-driven sphere and pad over a plane
    "mocap_f64": ("mocap_target", {}, "float64", 3, 3, "mocap"),
    "mocap_rk4_f32": ("mocap_target", {"integrator": 1}, "float32", 2, 2, "mocap"),
    # a mocap body that carries a jointed subtree (round 6): the children hang off the static body_pos / body_quat chain, the override comes after the scan
    "mocap_child_f64": ("mocap_child", {}, "float64", 3, 3, "mocap"),
    # ... and a 20-dof chain under a mocap body: the humanoid-class kernels (whole pass in one launch, level-sweep kinematics) on such a tree
    "mocap_chain_f64": ("mocap_chain", {}, "float64", 3, 3, "mocap"),
    "mocap_chain_cg1_f64": ("mocap_chain", {"solver": 1, "iterations": 1, "ls_iterations": 4}, "float64", 2, 3, "mocap"),
    # gravity compensation (passive.py:148-156, forward.py:206-207): passive on two links, through the actuator channel on the third
    "gravcomp_f64": ("gravcomp_arm", {}, "float64", 3, 3, "generic"),
    "gravcomp_rk4_f32": ("gravcomp_arm", {"integrator": 1}, "float32", 2, 2, "generic"),
    # motors on ball / free joints, child-frame and parent-frame transmissions (smooth.py:565-583)
    "ball_free_actuators_f64": ("ball_free_actuators", {}, "float64", 3, 3, "generic"),
    "ball_free_actuators_rk4_f32": ("ball_free_actuators", {"integrator": 1}, "float32", 2, 2, "generic"),
    # ball-joint limits (constraint.py:299-335)
    "ball_limits_f64": ("ball_limits", {}, "float64", 3, 3, "ball_limits"),
    "ball_limits_cg_rk4_f32": ("ball_limits", {"integrator": 1, "solver": 1}, "float32", 2, 2, "ball_limits"),
    # fixed tendons: lengths, limit rows, springs / dampers, tendon transmissions (smooth.py:470-497, constraint.py:375-405, passive.py:119-144)
    "tendon_fixed_f64": ("tendon_fixed", {}, "float64", 3, 3, "tendon"),
    "tendon_fixed_cg_rk4_f32": ("tendon_fixed", {"integrator": 1, "solver": 1}, "float32", 2, 2, "tendon"),
    # spatial tendons in the reference's own (degenerate) form: a tendon whose first wrap is not a joint wrap keeps length 0 and a zero
    # Jacobian row (smooth.py:470-497, device.py:850).  The fixed-tendon model with t1 (limited, spring, damper) and t4 (spring) retyped to
    # site wraps: their limit row, passive force and constants stay in the model, their kinematics read zero.
    "tendon_spatial_degenerate_f64": ("tendon_fixed", {"model.wrap_type": [3, 3, 1, 1, 1, 1, 1, 3]}, "float64", 3, 3, "tendon"),
    # ... the same two tendons declared as <spatial> in the XML: the compiler produces the site wraps and the qpos0 constants
    "tendon_spatial_f64": ("tendon_spatial", {}, "float64", 3, 3, "tendon"),
    # tendon armature (smooth.py:500-522): qM gains J^T diag(armature) J, off the kinematic tree's sparsity pattern
    "tendon_armature_f64": ("tendon_armature", {}, "float64", 3, 3, "tendon"),
    "tendon_armature_cg_rk4_f32": ("tendon_armature", {"integrator": 1, "solver": 1}, "float32", 2, 2, "tendon"),
    # tendon frictionloss rows (constraint.py:230-234) next to dof frictionloss, tendon limits and a contact
    "tendon_friction_f64": ("tendon_friction", {}, "float64", 3, 3, "tendon"),
    "tendon_friction_cg_f64": ("tendon_friction", {"solver": 1}, "float64", 2, 2, "tendon"),
    "tendon_friction_rk4_f32": ("tendon_friction", {"integrator": 1}, "float32", 2, 2, "tendon"),
    # max_contact_points (collision_driver.py:822-840): 13 candidate contacts, the 5 closest kept per environment
    # (every environment jittered: at the XML pose several candidates are at exactly equal distance and torch.topk orders equal
    # values by the internals of its partial sort -- [10, 7, 8, 9, 6] for thirteen equal values -- which is not a property of the
    # algorithm; this build orders ties by candidate index)
    # muscle actuators (support.py:197-296, forward.py:102-219; reference test/forward_test.py::test_muscle): <muscle> shortcut and <general> form
    "muscle_arm_f64": ("muscle_arm", {}, "float64", 4, 4, "muscle"),
    "muscle_arm_rk4_f32": ("muscle_arm", {"integrator": 1}, "float32", 3, 3, "muscle"),
    "capsules_topk_f64": ("capsules_topk", {}, "float64", 4, 3, "topk"),
    "capsules_topk_ell_rk4_f32": ("capsules_topk", {"integrator": 1, "cone": 1}, "float32", 3, 2, "topk"),
    # ... with box / mesh candidates: 22 candidate contacts of boxes, a capsule and a sphere, the 6 closest kept
    "boxes_topk_f64": ("boxes_topk", {}, "float64", 4, 3, "topk"),
    "boxes_topk_ell_rk4_f64": ("boxes_topk", {"integrator": 1, "cone": 1}, "float64", 3, 2, "topk"),
    # the last bundled model: every joint type stacked, ball limits, gravity compensation, mocap bodies, fixed tendons, motors on
    # ball / free joints, camera modes (contacts disabled in the XML)
    "pendula_f64": ("pendula", {}, "float64", 3, 3, "pendula"),
    "pendula_rk4_f32": ("pendula", {"integrator": 1}, "float32", 2, 2, "pendula"),
    "frictionloss_dof_f64": ("frictionloss_dof", {}, "float64", 3, 3, "friction_hinge"),
    "ant_frictionloss_newton_f64": ("ant_frictionloss", {}, "float64", 2, 3, "bench_ctrl"),
    "ant_frictionloss_cg_f64": ("ant_frictionloss", {"solver": 1}, "float64", 2, 2, "bench_ctrl"),
    "ant_frictionloss_rk4_ell_f32": ("ant_frictionloss", {"integrator": 1, "solver": 2, "cone": 1}, "float32", 2, 2, "bench_ctrl"),
    # step(..., fixed_iterations=True): the static-graph loops torch.compile users of the reference run (solver.py:64 fixed_loop,
    # :484-487 line search, :536-537 main loop) -- every iteration executes, converged or not
    "humanoid_cg_iter100_fixed_f64": ("humanoid", {"solver": 1, "iterations": 100, "ls_iterations": 50}, "float64", 2, 2, "perturbed", True),
    "humanoid_cg_fixed_f64": ("humanoid", {"solver": 1}, "float64", 3, 3, "bench", True),
    "ant_newton_fixed_f64": ("ant", {}, "float64", 2, 2, "bench_ctrl", True),
    "ant_rk4_newton_ell_fixed_f32": ("ant", {"integrator": 1, "solver": 2, "cone": 1, "iterations": 8, "ls_iterations": 6}, "float32", 2, 2, "bench_ctrl", True),
}

INPUT_LEAVES = ["time", "qpos", "qvel", "act", "qacc_warmstart", "ctrl", "qfrc_applied", "xfrc_applied", "qacc", "subtree_com", "mocap_pos", "mocap_quat"]


EXTRA_INPUTS = ["sensordata", "cacc", "cfrc_int", "subtree_linvel", "subtree_angmom"]  # inputs only some recipes set (sensor slots the step leaves alone; MJH_DATA_EXTRA_IN)


def make_inputs(recipe, lite, env):
    """Seeded per-env input overrides (numpy, float64). ``env`` is the seed index."""
    rng = np.random.RandomState(1000 + env)
    nq, nv, nu, nb = lite.nq, lite.nv, lite.nu, lite.nbody
    out = {}
    if recipe == "cartpole":
        out["qpos"] = np.array([0.1 * env, 0.7 - 0.4 * env])
        out["qvel"] = np.array([0.3, -0.5 + env])
        out["ctrl"] = np.array([0.5 * env - 0.3])
    elif recipe == "bench":  # benchmarks/_helpers.py:25-42 : qpos0, qvel = 0.01 * randn, ctrl = 0
        out["qvel"] = 0.01 * np.random.RandomState(42 + env).randn(nv)
    elif recipe == "bench_ctrl":
        out["qvel"] = 0.3 * rng.randn(nv)
        out["ctrl"] = np.clip(0.8 * rng.randn(nu), -1.5, 1.5)
        out["qpos"] = lite.qpos0 + 0.2 * rng.randn(nq)
        out["qfrc_applied"] = 0.5 * rng.randn(nv)
        out["xfrc_applied"] = 0.5 * rng.randn(nb, 6)
    elif recipe == "muscle":  # activations, controls beyond [0, 1] (clamped by the dynamics), joint angles and speeds across the force-length-velocity curves
        out["qpos"] = lite.qpos0 + np.array([0.6, 0.9]) * rng.randn(nq)
        out["qvel"] = 3.0 * rng.randn(nv)
        out["ctrl"] = rng.uniform(-0.3, 1.3, nu)
        out["act"] = rng.uniform(-0.1, 1.1, lite.na)
    elif recipe == "convex":  # free bodies resting on each other: jitter the poses (env 0 keeps the XML pose) and add velocity
        q = lite.qpos0.copy()
        for j in range(lite.njnt):
            a = int(lite.jnt_qposadr[j])
            if int(lite.jnt_type[j]) == 0 and env > 0:
                q[a : a + 3] += 0.01 * rng.randn(3)
                q[a + 3 : a + 7] += 0.03 * rng.randn(4)  # un-normalised on purpose
        out["qpos"] = q
        out["qvel"] = 0.2 * rng.randn(nv)
    elif recipe == "topk":  # free bodies, every environment jittered (no two candidate contacts at exactly the same distance)
        q = lite.qpos0.copy()
        for j in range(lite.njnt):
            a = int(lite.jnt_qposadr[j])
            q[a : a + 3] += 0.02 * rng.randn(3)
            q[a + 3 : a + 7] += 0.05 * rng.randn(4)
        out["qpos"] = q
        out["qvel"] = 0.3 * rng.randn(nv)
    elif recipe == "sensor_rig":
        q = lite.qpos0.copy()
        q[:3] += 0.1 * rng.randn(3) * (env > 0)
        q[3:7] += 0.2 * rng.randn(4) * (env > 0)
        q[7:] += 0.3 * rng.randn(nq - 7)
        out["qpos"] = q
        out["qvel"] = 0.5 * rng.randn(nv)
    elif recipe == "sensor_rig2":
        q = lite.qpos0.copy()
        q[:3] += 0.1 * rng.randn(3) * (env > 0)
        q[3:7] += 0.2 * rng.randn(4) * (env > 0)
        q[7:11] += 0.5 * rng.randn(4)          # the ball joint, un-normalised on purpose
        q[11:] += np.array([0.6, 0.08]) * rng.randn(2)
        out["qpos"] = q
        out["qvel"] = 0.6 * rng.randn(nv)
        out["ctrl"] = np.clip(0.8 * rng.randn(nu), -1.5, 1.5)
        out["time"] = np.array(0.003 * env)
        out["sensordata"] = rng.randn(lite.nsensordata)  # the slots of the untouched types keep these
        # leaves no stage of the reference writes, read by accelerometer / force / torque / subtreelinvel / subtreeangmom: env 0 keeps make_data's zeros
        if env > 0:
            out["cacc"] = 0.5 * rng.randn(nb, 6)
            out["cfrc_int"] = 2.0 * rng.randn(nb, 6)
            out["subtree_linvel"] = rng.randn(nb, 3)
            out["subtree_angmom"] = rng.randn(nb, 3)
    elif recipe == "friction_hinge":  # stick (small torque), slip both ways (large torque): reference test/solver_test.py:77-108
        out["qvel"] = np.array([0.0, 0.5, -0.5][env % 3]) * np.ones(nv)
        out["qfrc_applied"] = np.array([0.5, 100.0, -100.0][env % 3]) * np.ones(nv)
    elif recipe == "mocap":  # the caller moves the mocap bodies (un-normalised quaternion on purpose); free bodies get velocities
        out["qvel"] = 0.3 * rng.randn(nv)
        out["mocap_pos"] = lite.body_pos[lite.body_mocapid >= 0] + 0.05 * rng.randn(lite.nmocap, 3) * (env > 0)
        out["mocap_quat"] = lite.body_quat[lite.body_mocapid >= 0] + 0.3 * rng.randn(lite.nmocap, 4) * (env > 0)
    elif recipe == "ball_limits":  # swing the ball joints past their cones: big rotations about random axes
        q = lite.qpos0.copy()
        for j in range(lite.njnt):
            a = int(lite.jnt_qposadr[j])
            if int(lite.jnt_type[j]) == 1:
                ax = rng.randn(3); ax /= np.linalg.norm(ax)
                ang = [0.3, 0.6, 0.9][env % 3] * (1 + 0.3 * j)
                q[a : a + 4] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax]) * (1 + 0.05 * rng.randn())  # un-normalised
            else:
                q[a] += 0.3 * rng.randn()
        out["qpos"] = q
        out["qvel"] = 0.5 * rng.randn(nv)
    elif recipe == "tendon":  # joint angles large enough to put the tendons beyond their ranges on both sides
        out["qpos"] = lite.qpos0 + np.concatenate([[0.6, 0.5, 0.4, 0.3, 0.5][: nq - 7] * np.array([1, -1, 1, -1, 1])[: nq - 7] * (0.5 + env), 0.02 * rng.randn(7) * (env > 0)])
        out["qvel"] = 0.5 * rng.randn(nv)
        out["ctrl"] = np.clip(0.5 * rng.randn(nu), -1, 1)
    elif recipe == "pendula":  # generic state plus moved mocap bodies
        out["qpos"] = lite.qpos0 + 0.3 * rng.randn(nq) * (env > 0)
        out["qvel"] = 0.5 * rng.randn(nv)
        out["ctrl"] = np.clip(0.5 * rng.randn(nu), -1, 1)
        out["mocap_pos"] = lite.body_pos[lite.body_mocapid >= 0] + 0.1 * rng.randn(lite.nmocap, 3)
        out["mocap_quat"] = lite.body_quat[lite.body_mocapid >= 0] + 0.2 * rng.randn(lite.nmocap, 4)
    elif recipe == "generic":  # any model: jittered qpos (env 0 keeps qpos0), velocities, clipped controls
        out["qpos"] = lite.qpos0 + 0.05 * rng.randn(nq) * (env > 0)
        out["qvel"] = 0.3 * rng.randn(nv)
        out["ctrl"] = np.clip(0.5 * rng.randn(nu), -1, 1)
    elif recipe == "perturbed":
        q = lite.qpos0.copy()
        q[7:] += 0.4 * rng.randn(nq - 7)
        q[2] -= 0.06 + 0.05 * env  # push the feet into the floor
        q[3:7] += 0.15 * rng.randn(4)  # left un-normalised on purpose: kinematics normalises it
        out["qpos"] = q
        out["qvel"] = 0.5 * rng.randn(nv)
        out["ctrl"] = np.clip(0.7 * rng.randn(nu), -1.5, 1.5)
        out["qfrc_applied"] = 0.2 * rng.randn(nv)
        out["xfrc_applied"] = 1.0 * rng.randn(nb, 6)
        out["qacc_warmstart"] = 0.1 * rng.randn(nv)
    return out


def leaf(d, name):
    obj = d
    for p in native.DATA_PATH[name]:
        obj = getattr(obj, p)
    return obj


def main(only=None):
    ref = ref_harness.load()
    os.makedirs(GOLD, exist_ok=True)
    names = native.LISTS["MJH_DATA_REALS"] + native.LISTS["MJH_DATA_I32"] + native.LISTS["MJH_DATA_I64"]
    for case, spec in CASES.items():
        xml, overrides, dtype_s, nenv, nsteps, recipe = spec[:6]
        fixed = bool(spec[6]) if len(spec) > 6 else False
        if only and case not in only:
            continue
        dtype = getattr(torch, dtype_s)
        lite = mjcf.from_xml_path(os.path.join(REPO, "mujoco-torch_amd", "mujoco_torch_amd", "test_data", xml + ".xml"))
        for k, v in overrides.items():
            if k.startswith("model."):  # an edit of the compiled model itself (tests/_util.load_model applies the same)
                setattr(lite, k[6:], np.array(v, dtype=np.asarray(getattr(lite, k[6:])).dtype) if isinstance(v, list) else v)
                continue
            setattr(lite.opt, k, np.array(v, dtype=np.float64) if isinstance(v, list) else v)
        has_convex = any(int(t) in (6, 7) for t in lite.geom_type) and not (int(lite.opt.disableflags) & (1 << 4))  # convex tables matter only with contacts on
        # float32 + rangefinder raises inside the reference (float64 ray tables, ray.py:317): record those cases sensor-less
        keep_sensors = not (dtype != torch.float64 and any(int(t) == 7 for t in getattr(lite, "sensor_type", [])))
        if has_convex:
            # the reference's device_put(dtype=float32) leaves the convex tables in float64 and its step then fails on
            # mixed dtypes (constraint.py:475); Model.to(float32) is the route that works
            mref = ref_harness.put_model(ref, lite, keep_sensors=keep_sensors)
            if dtype != torch.float64:
                mref = mref.to(dtype)
        else:
            mref = ref_harness.put_model(ref, lite, dtype=dtype if dtype != torch.float64 else None, keep_sensors=keep_sensors)
        store = {}
        for g in range(lite.ngeom):  # the convex tables the reference derived (mesh.py:405-447, on oracle/ref_stubs/trimesh)
            if mref.geom_convex_face[g] is not None:
                for k in ("face", "vert", "edge", "facenormal"):
                    store[f"convex/{g}/{k}"] = getattr(mref, "geom_convex_" + k)[g].numpy().copy()
        for env in range(nenv):
            inp = make_inputs(recipe, lite, env)
            d = ref.io.make_data(mref)
            d = d.replace(**{k: torch.tensor(np.asarray(v, dtype=np.float64)) for k, v in inp.items()})
            if dtype != torch.float64:
                d = d.to(dtype)
            for n in INPUT_LEAVES:
                store[f"in/{env}/{n}"] = leaf(d, n).numpy().copy()
            for n in EXTRA_INPUTS:  # only where the recipe sets them: the other cases' key sets stay as they were
                if n in inp:
                    store[f"in/{env}/{n}"] = getattr(d, n).numpy().copy()
            for s in range(nsteps):
                d = ref.forward.step(mref, d, fixed_iterations=fixed)
                for n in names:
                    t = leaf(d, n)
                    store[f"out/{env}/{s}/{n}"] = t.numpy().copy()
        meta = dict(xml=xml, overrides=overrides, dtype=dtype_s, nenv=nenv, nsteps=nsteps, recipe=recipe, keep_sensors=keep_sensors, fixed_iterations=fixed,
                    constraint_sizes=list(mref.constraint_sizes_py), torch=torch.__version__)
        store["meta"] = np.array(json.dumps(meta))
        path = os.path.join(GOLD, case + ".npz")
        np.savez_compressed(path, **store)
        print(f"{case}: {os.path.getsize(path) / 1024:.0f} KB, sizes {mref.constraint_sizes_py}")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
