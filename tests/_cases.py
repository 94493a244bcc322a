"""Seeded parity cases shared by tests/test_gpu_parity.py and tools/parity_survey.py.

Each case: (xml, option overrides, dtype, batch) plus the bounds the HIP step must meet against the oracle on that batch:
``tol_sol`` -- solver-dependent leaves on the accepted branch (relative to the leaf's max magnitude, see _util.rel_err);
``max_alt`` -- largest fraction of environments allowed on a non-natural line-search branch (0: the natural oracle run must match).
The bounds are measured ones (profiles/r02/parity_survey.json) with headroom, not wishes; north_star's bar for float64 is 1e-8.
"""
import zlib

import numpy as np
import torch

import mujoco_torch_amd as mt
from _util import load_model

F64, F32 = torch.float64, torch.float32
TOL_PRE = {F64: 1e-9, F32: 2e-4}
TOL_SOL = {F64: 1e-8, F32: 2e-3}

# (xml, overrides, dtype, B, {bounds})
SEEDED_CASES = [
    # BASELINE config 2: iterations=1 / ls_iterations=4 stop the line search on its knife edge: measured 28.5 % of the
    # environments on a non-natural branch (worst error on the accepted branch 1e-13), Newton 55 % (profiles/r02/parity_survey_first.log)
    ("humanoid", {"solver": 1}, F64, 256, dict(max_alt=0.40)),
    ("humanoid", {}, F64, 64, dict(max_alt=0.65)),
    ("humanoid", {"solver": 1, "iterations": 100, "ls_iterations": 50}, F64, 64, dict(max_alt=0.0)),   # converged: branches re-converge
    ("ant", {"integrator": 1, "solver": 2, "cone": 1}, F32, 128, dict(tol_sol=5e-6)),  # BASELINE config 3 (measured 3.9e-7)
    ("ant", {"integrator": 1, "solver": 2, "cone": 1}, F64, 64, dict(max_alt=0.0)),
    ("ant", {}, F64, 64, dict(max_alt=0.0)),
    ("cartpole", {}, F64, 64, dict(max_alt=0.0)),
    ("mesh_contact", {}, F32, 256, dict(tol_sol=5e-4)),                         # BASELINE config 5 (box + mesh, condim 6, Newton, float32; measured 5.8e-5)
    ("mesh_contact", {}, F64, 64, dict(max_alt=0.0)),
    # CG on a piecewise-quadratic cost does not reach 1e-8: it ends where a line search stops improving the cost (improvement <
    # tolerance, solver.py:501-508) with the scaled gradient still ~1e-6, and two correct implementations end at points that far
    # apart -- measured worst error 5.8e-7 here, the same (7.8e-7) with tolerance = 1e-14 and 300 iterations, cond(M^-1 H) = 3.2e2
    # (profiles/r02/parity_survey_first.log).  What bounds the agreement is the solver's accuracy, not rounding: Newton on the
    # same scene (two lines up) agrees to 2e-14.
    ("mesh_contact", {"solver": 1, "cone": 1}, F64, 32, dict(tol_sol=1e-5)),
    ("convex_meshes", {}, F64, 32, dict(max_alt=0.0)),
    ("convex_primitives", {}, F64, 32, dict(max_alt=0.0)),
    ("convex_primitives", {}, F32, 32, dict(tol_sol=2e-4)),                      # measured 1.5e-5 (profiles/r04/notes.md: every float32 bound below is 10 x the measured worst of three steps)
    ("sensor_rig", {}, F64, 64, dict(max_alt=0.0)),                             # sensors: IMU, rangefinders, joint sensors
    ("sensor_rig", {"integrator": 1}, F32, 64, dict(tol_sol=6e-5)),             # ... RK4, float32 (measured 6.1e-6)
    ("sensor_rig2", {}, F64, 64, dict(max_alt=0.0)),                            # every other sensor type sensor.py evaluates (frame, tendon, actuator, ball, subtree, clock, force / torque ...)
    ("sensor_rig2", {"integrator": 1}, F32, 33, dict(tol_sol=2e-4)),            # ... RK4, float32 (measured 1.6e-5)
    ("swimmer", {"viscosity": 0.05, "wind": [0.3, -0.2, 0.1]}, F64, 64, dict(max_alt=0.0)),  # fluid forces: density + viscosity + wind
    ("ant_frictionloss", {}, F64, 64, dict(max_alt=0.0)),                       # dof frictionloss rows, Newton
    ("ant_frictionloss", {"solver": 1}, F64, 64, dict(tol_sol=1e-5)),           # ... CG: stalls across frictionloss zone switches (measured 8.1e-7; Newton one line up: 1e-13)
    ("halfcheetah", {}, F64, 64, dict(max_alt=0.0)),
    ("hopper", {}, F64, 64, dict(max_alt=0.0)),
    ("equality_loops", {}, F64, 64, dict(max_alt=0.0)),                         # equality rows: closed loop, weld, joint couplings
    # (equality_loops with RK4 + CG -- four stalling solves on stiff always-active rows per step -- is covered by the PROPERTY it has, not by
    # a loose bound: tests/test_gpu_parity.py::test_stalling_cg_stays_inside_the_oracles_own_band)
    ("equality", {}, F64, 32, dict(max_alt=0.0)),                               # bundled: site-form constraints carried inactive
    ("ant", {"disableflags": 1 << 4}, F64, 32, dict(max_alt=0.0)),              # disable flags (test/constraint_test.py:148-200): contacts off
    ("humanoid", {"disableflags": 1}, F64, 32, dict(max_alt=0.0)),              # ... every constraint off (nefc = 0)
    ("ant", {"disableflags": (1 << 12) | (1 << 9) | (1 << 8)}, F64, 32, dict(max_alt=0.0)),  # ... refsafe, warm start and ctrl clamping off
    ("hopper", {"disableflags": (1 << 7) | (1 << 11)}, F64, 32, dict(max_alt=0.0)),         # ... gravity and actuation off
    ("halfcheetah", {"disableflags": (1 << 5) | (1 << 3)}, F64, 32, dict(max_alt=0.0)),      # ... springs (hence every passive force) and limits off
    ("pendula", {}, F64, 64, dict(max_alt=0.0)),                                # bundled: every joint type, ball limits, gravcomp, mocap, tendons
    ("tendon_fixed", {"solver": 1}, F64, 64, dict(max_alt=0.0)),
    ("tendon_armature", {}, F64, 64, dict(max_alt=0.0)),                          # tendon armature: qM += J^T diag(armature) J off the tree's sparsity pattern
    ("capsules_topk", {}, F64, 64, dict(max_alt=0.0)),                           # max_contact_points: 13 candidates, the 5 closest kept per environment
    ("capsules_topk", {"integrator": 1, "cone": 1}, F32, 64, dict(tol_sol=5e-5)),  # measured 4.9e-6
    ("boxes_topk", {}, F64, 48, dict(max_alt=0.0)),                               # ... over box candidates: the convex narrow phase hands 22 candidates to the selection through the workspace
    ("boxes_topk", {"integrator": 1, "cone": 1}, F32, 33, dict(tol_sol=1.5e-4)),  # measured 1.4e-5
    ("capsules_topk", {"solver": 1, "iterations": 100, "ls_iterations": 50}, F64, 33, dict(tol_sol=1e-5)),  # ... through the register solver (CG stall accuracy)
    ("centipede", {}, F64, 24, dict(max_alt=0.0)),                                # 72 dofs / 74 bodies (jacobian=dense): multi-word dof masks, more dofs than lanes in the LDS solver
    ("centipede", {"integrator": 1}, F32, 17, dict(tol_sol=6e-5)),                # measured 5.5e-6
    ("muscle_arm", {}, F64, 64, dict(max_alt=0.0)),                               # muscle actuators: activation dynamics, force-length-velocity gain, passive bias
    ("muscle_arm", {"integrator": 1}, F32, 33, dict(tol_sol=2e-5)),              # measured 2.0e-6
    ("tendon_friction", {}, F64, 64, dict(max_alt=0.0)),                         # tendon + dof frictionloss rows, Newton
    ("tendon_friction", {"solver": 1, "integrator": 1}, F64, 32, dict(tol_sol=1e-5)),  # ... CG (stall accuracy, see above), RK4
    # CG models served by the register solver (mjh_sol2_kernel: two environments per wavefront): Euler with the eulerdamp re-solve,
    # RK4, float32, odd batch sizes (the last wavefront runs half empty), warm start off, one- and many-iteration loops
    ("halfcheetah", {"solver": 1}, F64, 33, dict(tol_sol=1e-6)),
    ("hopper", {"solver": 1, "integrator": 1}, F64, 17, dict(tol_sol=1e-6)),
    ("walker2d", {"solver": 1, "disableflags": 1 << 9}, F64, 32, dict(tol_sol=1e-6)),
    ("humanoid", {"solver": 1, "integrator": 1, "iterations": 100, "ls_iterations": 50}, F64, 31, dict(max_alt=0.0, tol_sol=2e-8)),  # CG's own stall accuracy: measured 1.8e-9
    ("humanoid", {"solver": 1, "iterations": 100, "ls_iterations": 50}, F32, 64, dict(tol_sol=1e-2)),  # float32 CG stalls at ~2e-3 (measured 2.2e-3)
    # (steps with SEVERAL knife-edged solves -- four RK4 stages x one iteration, or a few capped iterations -- have 2^k admissible
    # outcomes, k = 15 - 70 noise candidates per environment: neither the oracle's single-switch policies nor any spread estimate
    # covers them; oracle and reference themselves agree there only under such policies (checked with the reference's Python on
    # iterations = 2, 3).  They are covered by their converged twins above and by the one-solve BASELINE configuration.)
    ("humanoid", {"solver": 1, "disableflags": 0}, F64, 1, dict(max_alt=1.0)),     # eulerdamp on (the XML disables it), a single environment
]


def seeded_tol_sol(xml, overrides, dtype):
    """The solver bound tests/_cases.py states for a seeded configuration (its `tol_sol`, else the dtype's default)."""
    for x, ov, dt, _, bounds in SEEDED_CASES:
        if x == xml and ov == overrides and dt == dtype:
            return bounds.get("tol_sol", TOL_SOL[dtype])
    return TOL_SOL[dtype]


def case_id(c):
    xml, ov, dt, B, _ = c
    return f"{xml}-{'-'.join(f'{k}{v}' for k, v in ov.items()) or 'default'}-{str(dt)[6:]}-B{B}"


def seeded_batch(xml, overrides, dtype, B):
    """The bench's input recipe (make_data state, small random velocities, controls), with poses jittered where the scene would
    otherwise be a single symmetric configuration."""
    mx = load_model(xml, overrides, dtype)
    rng = np.random.RandomState(42)
    d = mt.make_data(mx).expand(B).clone()
    d = d.replace(qvel=torch.tensor(0.01 * rng.randn(B, mx.nv)), ctrl=torch.tensor(0.3 * rng.randn(B, mx.nu)))
    if any(p[0] >= 5 for p in mx.tables.pairs):  # convex pairs: free bodies resting on each other, jitter the poses too
        q = d.qpos.clone()
        for j in range(mx.njnt):
            a = int(mx.jnt_qposadr[j])
            if int(mx.jnt_type.data[j]) != 0:  # hinge / slide joints of mixed models: a small angle
                q[:, a] += torch.tensor(0.05 * rng.randn(B))
                continue
            q[:, a : a + 3] += torch.tensor(0.01 * rng.randn(B, 3))
            q[:, a + 3 : a + 7] += torch.tensor(0.03 * rng.randn(B, 4))
        d = d.replace(qpos=q, qvel=torch.tensor(0.2 * rng.randn(B, mx.nv)))
    if xml == "muscle_arm":  # activations and controls across (and beyond) [0, 1], joint angles and speeds across the force-length-velocity curves
        d = d.replace(qpos=d.qpos + torch.tensor(np.array([0.6, 0.9]) * rng.randn(B, mx.nq)), qvel=torch.tensor(3.0 * rng.randn(B, mx.nv)),
                      ctrl=torch.tensor(rng.uniform(-0.3, 1.3, (B, mx.nu))), act=torch.tensor(rng.uniform(-0.1, 1.1, (B, mx.na))))
    if xml == "centipede":  # bend the legs so that tips reach the floor and limited joints pass their ranges
        d = d.replace(qpos=d.qpos + torch.tensor(0.5 * rng.randn(B, mx.nq)), qvel=torch.tensor(0.5 * rng.randn(B, mx.nv)))
    if xml == "sensor_rig":  # move and spin the rover so every sensor reads something different per environment
        q = d.qpos.clone()
        q[:, :3] += torch.tensor(0.1 * rng.randn(B, 3))
        q[:, 3:7] += torch.tensor(0.2 * rng.randn(B, 4))
        q[:, 7:] += torch.tensor(0.3 * rng.randn(B, mx.nq - 7))
        d = d.replace(qpos=q, qvel=torch.tensor(0.5 * rng.randn(B, mx.nv)))
    if xml == "sensor_rig2":  # every joint moved (the ball joint's quaternion left un-normalised), the Data leaves no stage writes but sensors read set per environment
        q = d.qpos.clone()
        q[:, :3] += torch.tensor(0.1 * rng.randn(B, 3))
        q[:, 3:11] += torch.tensor(0.3 * rng.randn(B, 8))
        q[:, 11:] += torch.tensor(np.array([0.6, 0.08]) * rng.randn(B, 2))
        nb = int(mx.nbody)
        d = d.replace(qpos=q, qvel=torch.tensor(0.6 * rng.randn(B, mx.nv)), ctrl=torch.tensor(np.clip(0.8 * rng.randn(B, mx.nu), -1.5, 1.5)),
                      time=torch.tensor(0.003 * np.arange(B, dtype=np.float64)), sensordata=torch.tensor(rng.randn(B, int(mx.nsensordata))),
                      cacc=torch.tensor(0.5 * rng.randn(B, nb, 6)), cfrc_int=torch.tensor(2.0 * rng.randn(B, nb, 6)),
                      subtree_linvel=torch.tensor(rng.randn(B, nb, 3)), subtree_angmom=torch.tensor(rng.randn(B, nb, 3)))
    if dtype != torch.float64:
        d = d.to(dtype)
    return mx, d


# the differential campaign of round 1 (tools/fuzz_parity.py): big perturbations, every input leaf randomised
FUZZ_CASES = [  # (xml, overrides, dtype, solver tolerance)
    ("humanoid", {"solver": 1}, F64, 1e-8), ("humanoid", {}, F64, 1e-8), ("humanoid", {"iterations": 20, "ls_iterations": 20}, F32, 5e-3),
    ("ant", {}, F64, 1e-8), ("ant", {"integrator": 1, "solver": 2, "cone": 1}, F32, 5e-3), ("ant", {"solver": 1, "cone": 1}, F64, 1e-5),
    ("halfcheetah", {}, F64, 1e-8), ("hopper", {}, F64, 1e-8), ("walker2d", {"integrator": 1}, F64, 1e-8),
    ("swimmer", {"viscosity": 0.05}, F64, 1e-8), ("cartpole", {}, F64, 1e-8), ("satellite_small", {}, F64, 1e-8),
    ("sensor_rig", {}, F64, 1e-8), ("sensor_rig2", {}, F64, 1e-8), ("mesh_contact", {}, F64, 1e-8), ("mesh_contact", {"integrator": 1}, F64, 1e-8), ("convex_primitives", {}, F64, 1e-8),
    ("equality_loops", {}, F64, 1e-8), ("equality", {}, F64, 1e-8), ("ball_limits", {}, F64, 1e-8),
    ("tendon_fixed", {}, F64, 1e-8), ("gravcomp_arm", {}, F64, 1e-8), ("gravcomp_arm", {"integrator": 1}, F64, 1e-8), ("ball_free_actuators", {}, F64, 1e-8),
    ("mocap_target", {}, F64, 1e-8), ("pendula", {}, F64, 1e-8),
    # float32 RK4 + CG on the stiff pendula: 2e-2 is this case's FLOAT32 ACCURACY, not slack -- the float32 ORACLE is 4.5e-3 .. 5.5e-3 (worst of 4096 environments, four steps; 99.9 %: 2.7e-3 .. 4.5e-3)
    # from the float64 solution of the same inputs, the GPU 3.0e-3 .. 6.1e-3, GPU against float32 oracle 5.2e-3 .. 7.2e-3 on the natural branch (4.9e-3 accepted; medians 3.5e-7 / 1e-5 / 1e-5:
    # tools/f32_yardstick.py, profiles/r05/yardstick_pendula.txt).  CG stalls at ~1e-3 of float32 here (the bound used to be 5e-3, 2 % above the measurement).  What
    # a kernel difference would move is the DISTRIBUTION: tests/test_gpu_parity.py::test_float32_stall_case_is_float32_accuracy holds the GPU to the float64 yardstick quantile by quantile.
    ("pendula", {"integrator": 1, "solver": 1}, F32, 2e-2),
    ("frictionloss_dof", {}, F64, 1e-8), ("ant_frictionloss", {}, F64, 1e-8),
    ("muscle_arm", {}, F64, 1e-8), ("tendon_armature", {}, F64, 1e-8), ("tendon_friction", {}, F64, 1e-8), ("capsules_topk", {}, F64, 1e-8),
    ("centipede", {}, F64, 1e-8), ("tendon_spatial", {}, F64, 1e-8),
    ("mesh_contact", {}, F32, 5e-3),  # BASELINE config 5's model and dtype: the packed Newton tier with the Hessian on the matrix cores (round 4)
    ("mesh_contact_arm", {}, F32, 5e-3), ("mesh_contact_arm", {}, F64, 1e-8),  # 15 dofs, condim 6 + 3 contacts, hinge limits: the 16-wide instantiations of that tier (four tile rows)
]
# float32 cases of the campaign: near-degenerate contact normals amplify eps under these perturbations (pre-solver 1e-3); qfrc_constraint /
# efc_force of the ant cancel forces of ~1e5, the dynamics leaves carry the comparison there
FUZZ_TOL_PRE = {F64: 1e-9, F32: 1e-3}
# (quantile, bound) beside a case's worst-environment bound: the float32 RK4 + CG pendula case states its float32 accuracy (2e-2) for the WORST environment; 99 % of
# the environments stay inside the old 5e-3 (measured 1.1e-3 .. 1.6e-3 at the 99 % quantile over 4096 x 4 environment-steps, 2.8e-3 .. 4.5e-3 at 99.9 %: profiles/r05/yardstick_pendula.txt)
FUZZ_QUANTILE = {("pendula", F32): (0.99, 5e-3)}
# convex_primitives under the campaign's perturbations runs Newton into its 10-iteration cap on ~0.6 % of the environments: the reference's own
# admissible outcomes are 2e-6 .. 0.76 apart there (tests/test_oracle_golden.py::test_iteration_capped_newton_states_are_implementation_defined,
# test_pinned_campaign_outliers): those environments are held to the oracle's own spread instead (band), the rest to 1e-8
FUZZ_BAND = {"convex_primitives": 4.0}


def fuzz_batch(xml, overrides, dtype, B):
    mx = load_model(xml, overrides, dtype)
    rng = np.random.RandomState(zlib.crc32(xml.encode()) % 1000)  # stable across processes (hash() is salted)
    d = mt.make_data(mx).expand(B).clone()
    q = d.qpos.clone()
    scale = torch.tensor(rng.uniform(0.0, 0.5, size=(B, 1)))  # per-environment perturbation size, some environments stay at qpos0
    q = q + scale * torch.tensor(rng.randn(B, mx.nq))
    kw = dict(qpos=q, qvel=torch.tensor(rng.randn(B, mx.nv)) * scale * 4, ctrl=torch.tensor(rng.uniform(-1.2, 1.2, size=(B, mx.nu))),
              qfrc_applied=torch.tensor(0.5 * rng.randn(B, mx.nv)), xfrc_applied=torch.tensor(0.5 * rng.randn(B, mx.nbody, 6)),
              qacc_warmstart=torch.tensor(rng.randn(B, mx.nv)) * scale)
    if mx.nmocap:
        kw["mocap_pos"] = d.mocap_pos + 0.1 * torch.tensor(rng.randn(B, mx.nmocap, 3))
        kw["mocap_quat"] = d.mocap_quat + 0.3 * torch.tensor(rng.randn(B, mx.nmocap, 4))
    if mx.neq:
        kw["eq_active"] = torch.tensor(rng.randint(0, 2, size=(B, mx.neq)), dtype=torch.int32) * d.eq_active.clamp(max=1) + d.eq_active * 0
        kw["eq_active"] = torch.where(torch.tensor(rng.rand(B, mx.neq) < 0.3), torch.zeros_like(d.eq_active), d.eq_active)
    if mx.tables.sensors["extra_leaves"]:  # sensors reading Data leaves no stage writes: every such leaf randomised, like the other inputs
        for n, w in (("cacc", 6), ("cfrc_int", 6), ("subtree_linvel", 3), ("subtree_angmom", 3)):
            kw[n] = torch.tensor(rng.randn(B, int(mx.nbody), w))
        kw["sensordata"] = torch.tensor(rng.randn(B, int(mx.nsensordata)))
    d = d.replace(**kw)
    if dtype != torch.float64:
        d = d.to(dtype)
    return mx, d
