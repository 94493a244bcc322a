/*
 * mjhip.h -- C ABI of the MI355X-native batched stepper behind mujoco_torch.step().
 *
 * The reference (vmoens/mujoco-torch) has NO native boundary: its step is pure Python
 * (mujoco_torch/_src/forward.py:463-496) vectorised with torch.vmap.  This header is the
 * boundary a maintainer binds instead; each entry point names the reference call it replaces:
 *
 *   mjh_model_create   <- Model.to(device) + _build_device_precomp   (_src/types.py:949-1013)
 *   mjh_forward        <- forward.forward(m, d)                       (_src/forward.py:373-401)
 *   mjh_step           <- forward.step(m, d, fixed_iterations)        (_src/forward.py:463-496)
 *
 * Conventions
 *  - every Data leaf is batch-major contiguous: shape [B, ...] exactly as
 *    make_data(mx).expand(B).clone() lays it out (reference benchmarks/_helpers.py:37-41);
 *  - all pointers in mjhData are DEVICE pointers for the mjh_* entry points and HOST pointers
 *    for the oracle twin (oracle/mjoracle.c: mjo_forward / mjo_step, same structs);
 *  - "real" fields are double (dtype 0) or float (dtype 1) for the whole call;
 *  - functions return 0 on success or a negative errno-style code; they never throw, never
 *    allocate, never synchronise: work is enqueued on the given hipStream_t (passed as void*);
 *  - the caller owns every buffer.  `in` and `out` may alias field-by-field only for fields
 *    the step does not write (qacc/qacc_warmstart outputs must be distinct storage, mirroring
 *    reference solver.py:541-548).
 *
 * The X-macro lists below are the single source of truth for field order; the Python host
 * side (mujoco_torch_amd/native.py) parses them to build its ctypes structures.
 */
#ifndef MJHIP_H_
#define MJHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJH_ABI_VERSION 14

/* ---- dtype / flags ------------------------------------------------------------------- */
#define MJH_F64 0
#define MJH_F32 1

#define MJH_FLAG_FIXED_ITERATIONS 1 /* step(..., fixed_iterations=True), forward.py:463 */

/* stage bits for mjh_forward(stages): run the pipeline up to and including the highest bit */
#define MJH_STAGE_KINEMATICS 0x001 /* smooth.kinematics + com_pos        smooth.py:34-288   */
#define MJH_STAGE_CRB 0x002        /* smooth.crb + factor_m              smooth.py:291-332  */
#define MJH_STAGE_COLLISION 0x004  /* collision_driver.collision         :800-875           */
#define MJH_STAGE_CONSTRAINT 0x008 /* constraint.make_constraint         :600-768           */
#define MJH_STAGE_VELOCITY 0x010   /* transmission, _velocity, passive, rne                  */
#define MJH_STAGE_ACTUATION 0x020  /* forward._actuation + _acceleration :102-228           */
#define MJH_STAGE_SOLVE 0x040      /* solver.solve                       solver.py:244-553  */
#define MJH_STAGE_ALL 0x07f

/* pair-function ids of the static collision table (collision_driver.py:106-125) */
#define MJH_FN_PLANE_SPHERE 0
#define MJH_FN_PLANE_CAPSULE 1
#define MJH_FN_SPHERE_SPHERE 2
#define MJH_FN_SPHERE_CAPSULE 3
#define MJH_FN_CAPSULE_CAPSULE 4
#define MJH_FN_PLANE_CONVEX 5
#define MJH_FN_SPHERE_CONVEX 6
#define MJH_FN_CAPSULE_CONVEX 7
#define MJH_FN_CONVEX_CONVEX 8
#define MJH_MAX_PAIR_CONTACTS 4

/* ---- model description (host arrays; copied into a device blob by mjh_model_create) ---- */

/* int32 scalars */
#define MJH_MODEL_INTS(X)                                                                        \
  X(nq) X(nv) X(nu) X(na) X(nbody) X(njnt) X(ngeom) X(nsite) X(ncam) X(nlight) X(nmocap)         \
  X(ne) X(nf) /* dof-frictionloss rows */ X(nft) /* tendon-frictionloss rows (dense; they follow the dof ones) */ X(nl) /* slide / hinge limit rows (single-column) */ X(nlb) /* ball-joint limit rows */ X(nlt) /* tendon limit rows */ X(ncon) X(ncand) /* candidate contacts the narrow phase computes: == ncon unless max_contact_points keeps the ncon closest (collision_driver.py:822-840) */ X(topk) /* 1: that selection is on; the con_* tables are then in candidate order, ncand long */ X(nefc) X(npair) X(nconvex) \
  X(ntendon) /* fixed tendons = length of the ten_length / ten_velocity leaves */ X(nwrapj) /* joint terms of all tendons (entries of ten_dof / ten_qposadr / ten_coef) */ \
  X(neq) /* equality constraints of the model = length of the eq_active leaf */ X(neqtab) /* entries of the eq_* tables (0 when equality rows are disabled) */ \
  X(nsensor) /* sensors the stepper computes (sns_* tables) */ X(nsensordata) /* length of the sensordata leaf */ \
  X(integrator) X(solver) X(cone) X(disableflags) X(iterations) X(ls_iterations)

/* double scalars */
#define MJH_MODEL_REALS(X)                                                                       \
  X(timestep) X(impratio) X(tolerance) X(ls_tolerance) X(meaninertia)                            \
  X(gravity_x) X(gravity_y) X(gravity_z)                                                         \
  X(density) X(viscosity) X(wind_x) X(wind_y) X(wind_z) /* fluid model of passive.py:31-78; all zero = no fluid forces */ \
  X(magnetic_x) X(magnetic_y) X(magnetic_z) /* opt.magnetic: magnetometer sensors (sensor.py:92-94) */

/* const int32_t* arrays (length in comment) */
#define MJH_MODEL_INT_ARRAYS(X)                                                                  \
  X(body_parentid)  /* nbody */                                                                  \
  X(body_rootid)    /* nbody */                                                                  \
  X(body_jntadr)    /* nbody */                                                                  \
  X(body_jntnum)    /* nbody */                                                                  \
  X(body_dofadr)    /* nbody */                                                                  \
  X(body_dofnum)    /* nbody */                                                                  \
  X(body_mocapid)   /* nbody */                                                                  \
  X(jnt_type)       /* njnt */                                                                   \
  X(jnt_qposadr)    /* njnt */                                                                   \
  X(jnt_dofadr)     /* njnt */                                                                   \
  X(jnt_bodyid)     /* njnt */                                                                   \
  X(jnt_actfrclimited) /* njnt */                                                                \
  X(jnt_actgravcomp) /* njnt: gravity compensation of the joint's dofs is applied as an actuator force (forward.py:206-207) */ \
  X(dof_bodyid)     /* nv */                                                                     \
  X(dof_jntid)      /* nv */                                                                     \
  X(dof_parentid)   /* nv */                                                                     \
  X(geom_type)      /* ngeom */                                                                  \
  X(geom_bodyid)    /* ngeom */                                                                  \
  X(geom_convexid)  /* ngeom: index into convex_* tables or -1 */                                \
  X(site_bodyid)    /* nsite */                                                                  \
  X(cam_bodyid)     /* ncam */                                                                   \
  X(cam_mode)       /* ncam */                                                                   \
  X(cam_targetbodyid) /* ncam */                                                                 \
  X(light_bodyid)   /* nlight */                                                                 \
  X(act_trntype)    /* nu */                                                                     \
  X(act_jnttype)    /* nu */                                                                     \
  X(act_dofadr)     /* nu */                                                                     \
  X(act_qposadr)    /* nu */                                                                     \
  X(act_gaintype)   /* nu */                                                                     \
  X(act_biastype)   /* nu */                                                                     \
  X(act_dyntype)    /* nu */                                                                     \
  X(act_ctrllimited)  /* nu */                                                                   \
  X(act_forcelimited) /* nu */                                                                   \
  X(act_actlimited)   /* nu */                                                                   \
  X(act_actadr)     /* nu */                                                                     \
  X(act_actnum)     /* nu */                                                                     \
  X(sns_type)       /* nsensor: mjtSensor of the sensors sensor.py evaluates (every type of its three stage functions; the others keep their slots: slot_sensor) */ \
  X(sns_adr)        /* nsensor: first slot in sensordata */                                      \
  X(sns_objid)      /* nsensor: id of the object -- site / body / geom / camera / tendon / actuator id; the qpos address (jointpos, ballquat) or dof address (jointvel, ballangvel, jointactuatorfrc) of a joint */ \
  X(sns_bodyid)     /* nsensor: body the object rides on (site sensors, frame sensors) */       \
  X(sns_rootid)     /* nsensor: root body of that body */                                        \
  X(sns_objtype)    /* nsensor: mjtObj of the object: 0 unknown, 1 body (inertial frame), 2 xbody, 5 geom, 6 site, 7 camera (frame sensors, sensor.py:62-74) */ \
  X(sns_reftype)    /* nsensor: mjtObj of the reference frame of a frame sensor (0 = none) */   \
  X(sns_refid)      /* nsensor: its id, -1 = none (the value is reported in the world frame) */ \
  X(sns_refbodyid)  /* nsensor: body of the reference object */                                  \
  X(sns_refrootid)  /* nsensor: root body of that body */                                        \
  X(sns_datatype)   /* nsensor: mjtDataType (0 real, 1 positive) for the cutoff rule */          \
  X(sns_rfadr)      /* nsensor+1: rangefinders, range into rf_geom; geoms in the reference's evaluation order */ \
  X(rf_geom)        /* geom ids a rangefinder ray is tested against (ray.py:292-325: site's own body excluded, invisible geoms dropped) */ \
  X(slot_sensor)    /* nsensordata: sns_* index that produces the slot, -1 = the slot keeps the caller's value */ \
  X(eq_kind)        /* neqtab: 0 connect (3 rows), 1 weld (6 rows), 2 joint coupling (1 row); reference ROW order = all connects, all welds, all joint couplings (constraint.py:651-656) */ \
  X(eq_id)          /* neqtab: index of the constraint in the model (eq_active / eq_data / eq_solref / eq_solimp) */ \
  X(eq_obj1)        /* neqtab: body ids (connect, weld) or joint id (joint coupling) */          \
  X(eq_obj2)        /* neqtab: second body / joint (-1 = none for a joint coupling) */           \
  X(eq_row)         /* neqtab: first efc row */                                                  \
  X(eq_jadr)        /* neqtab*4: joint couplings: dofadr1, dofadr2, qposadr1, qposadr2 (device.py:310-314; a missing second joint reads the LAST joint, as the reference's jnt_dofadr[-1] does) */ \
  X(fric_dof)       /* nf: dof of each dof-frictionloss row, reference row order (constraint.py:215-251) */ \
  X(topk_slot)      /* ncon (top-k only): final contact slot of the t-th closest candidate (the unstable argsort of the kept contacts' equal condims, collision_driver.py:828) */ \
  X(fric_tendon)    /* nft: tendon id of each tendon-frictionloss row (constraint.py:230-234) */ \
  X(ten_adr)        /* ntendon+1: CSR of the joint terms of each fixed tendon (smooth.py:470-497) */ \
  X(ten_dof)        /* nwrapj: dof of the term */                                                \
  X(ten_qposadr)    /* nwrapj: qpos address of the term */                                       \
  X(lim_tendon)     /* nlt: tendon id of each tendon limit row (constraint.py:375-405); these rows follow the slide / hinge ones */ \
  X(act_trnid)      /* nu: joint id, or tendon id for tendon transmissions (act_trntype 3) */    \
  X(lim_ball_jnt)   /* nlb: joint id of each ball-joint limit row (constraint.py:299-335); these rows precede the slide / hinge ones */ \
  X(lim_jnt)        /* nl: joint id of each slide/hinge limit row, reference row order */        \
  X(pair_fn)        /* npair: MJH_FN_* */                                                        \
  X(pair_geom1)     /* npair */                                                                  \
  X(pair_geom2)     /* npair */                                                                  \
  X(pair_ncon)      /* npair */                                                                  \
  X(pair_dst)       /* npair*MJH_MAX_PAIR_CONTACTS: contact slot of each contact of the pair */  \
  X(con_dim)        /* ncon */                                                                   \
  X(con_geom1)      /* ncon */                                                                   \
  X(con_geom2)      /* ncon */                                                                   \
  X(con_efc_address) /* ncon: cone-aware first row (constraint.py:636-646) */                    \
  X(convex_nvert)   /* nconvex */                                                                \
  X(convex_nface)   /* nconvex */                                                                \
  X(convex_nfv)     /* nconvex: vertices per (padded) face */                                    \
  X(convex_nedge)   /* nconvex */                                                                \
  X(convex_vertadr) /* nconvex: offset (in vec3) into convex_vert */                             \
  X(convex_faceadr) /* nconvex: offset (in ints) into convex_face */                             \
  X(convex_normadr) /* nconvex: offset (in vec3) into convex_facenormal */                       \
  X(convex_edgeadr) /* nconvex: offset (in int pairs) into convex_edge */                        \
  X(convex_face)    /* sum(nface*nfv): vertex ids */                                             \
  X(convex_edge)    /* sum(nedge)*2 */

/* const double* arrays */
#define MJH_MODEL_REAL_ARRAYS(X)                                                                 \
  X(qpos0)          /* nq */                                                                     \
  X(qpos_spring)    /* nq */                                                                     \
  X(body_pos)       /* nbody*3 */                                                                \
  X(body_quat)      /* nbody*4 */                                                                \
  X(body_ipos)      /* nbody*3 */                                                                \
  X(body_iquat)     /* nbody*4 */                                                                \
  X(body_mass)      /* nbody */                                                                  \
  X(body_inertia)   /* nbody*3 */                                                                \
  X(body_invweight0) /* nbody (translational component) */                                       \
  X(jnt_pos)        /* njnt*3 */                                                                 \
  X(jnt_axis)       /* njnt*3 */                                                                 \
  X(jnt_stiffness)  /* njnt */                                                                   \
  X(jnt_range)      /* njnt*2 */                                                                 \
  X(jnt_margin)     /* njnt */                                                                   \
  X(jnt_solref)     /* njnt*2 */                                                                 \
  X(jnt_solimp)     /* njnt*5 */                                                                 \
  X(jnt_actfrcrange) /* njnt*2 */                                                                \
  X(dof_armature)   /* nv */                                                                     \
  X(dof_damping)    /* nv */                                                                     \
  X(dof_invweight0) /* nv */                                                                     \
  X(sns_cutoff)     /* nsensor */                                                                \
  X(dof_frictionloss) /* nv */                                                                   \
  X(dof_solref)     /* nv*2 */                                                                   \
  X(dof_solimp)     /* nv*5 */                                                                   \
  X(ten_coef)       /* nwrapj: coefficient of the term */                                        \
  X(tendon_range)   /* ntendon*2 */                                                              \
  X(tendon_margin)  /* ntendon */                                                                \
  X(tendon_invweight0) /* ntendon */                                                             \
  X(tendon_solref_lim) /* ntendon*2 */                                                           \
  X(tendon_solimp_lim) /* ntendon*5 */                                                           \
  X(tendon_stiffness) /* ntendon */                                                              \
  X(tendon_damping) /* ntendon */                                                                \
  X(tendon_frictionloss) /* ntendon */ X(tendon_solref_fri) /* ntendon*2 */ X(tendon_solimp_fri) /* ntendon*5 */ \
  X(tendon_lengthspring) /* ntendon*2: the spring is slack between the two lengths (passive.py:121-127) */ \
  X(tendon_armature) /* ntendon: qM += J^T diag(armature) J (smooth.py:500-522) */ \
  X(body_gravcomp)  /* nbody: fraction of the body's weight compensated (passive.py:148-156); all zero = none */ \
  X(body_invweight0_rot) /* nbody (rotational component; weld rows 3..5, constraint.py:193-194) */         \
  X(eq_data)        /* neq*11 (MuJoCo layout: connect anchors; weld anchors, relpose, torquescale; joint polycoef) */ \
  X(eq_solref)      /* neq*2 */                                                                  \
  X(eq_solimp)      /* neq*5 */                                                                  \
  X(geom_pos)       /* ngeom*3 */                                                                \
  X(geom_quat)      /* ngeom*4 */                                                                \
  X(geom_size)      /* ngeom*3 */                                                                \
  X(site_pos)       /* nsite*3 */                                                                \
  X(site_quat)      /* nsite*4 */                                                                \
  X(cam_pos)        /* ncam*3 */                                                                 \
  X(cam_quat)       /* ncam*4 */                                                                 \
  X(cam_pos0)       /* ncam*3 */                                                                 \
  X(cam_mat0)       /* ncam*9 */                                                                 \
  X(light_pos)      /* nlight*3 */                                                               \
  X(light_dir)      /* nlight*3 */                                                               \
  X(act_gear)       /* nu*6 */                                                                   \
  X(act_gainprm)    /* nu*9 (muscle gains read all nine, support.py:246-272; fixed / affine the first three) */ \
  X(act_biasprm)    /* nu*9 */                                                                   \
  X(act_lengthrange) /* nu*2 (muscles) */                                                        \
  X(act_acc0)       /* nu (muscles: force = scale / acc0 when gainprm[2] < 0) */                  \
  X(act_dynprm)     /* nu*3 */                                                                   \
  X(act_ctrlrange)  /* nu*2 */                                                                   \
  X(act_forcerange) /* nu*2 */                                                                   \
  X(act_actrange)   /* nu*2 */                                                                   \
  X(con_includemargin)  /* ncon */                                                               \
  X(con_friction)       /* ncon*5 */                                                             \
  X(con_solref)         /* ncon*2 */                                                             \
  X(con_solreffriction) /* ncon*2 */                                                             \
  X(con_solimp)         /* ncon*5 */                                                             \
  X(convex_vert)        /* sum(nvert)*3 */                                                       \
  X(convex_facenormal)  /* sum(nface)*3 */

typedef struct mjhModelDesc {
  int32_t abi_version;
#define X(n) int32_t n;
  MJH_MODEL_INTS(X)
#undef X
#define X(n) double n;
  MJH_MODEL_REALS(X)
#undef X
#define X(n) const int32_t* n;
  MJH_MODEL_INT_ARRAYS(X)
#undef X
#define X(n) const double* n;
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
#define X(n) int64_t len_##n;
  MJH_MODEL_INT_ARRAYS(X)
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
} mjhModelDesc;

/* ---- Data leaves (reference _src/types.py:1091-1261, Contact :1036-1088) --------------- */

/* real leaves; per-env element count in the comment */
#define MJH_DATA_REALS(X)                                                                        \
  X(time)             /* 1 */                                                                    \
  X(qpos)             /* nq */                                                                   \
  X(qvel)             /* nv */                                                                   \
  X(act)              /* na */                                                                   \
  X(qacc_warmstart)   /* nv */                                                                   \
  X(ctrl)             /* nu */                                                                   \
  X(qfrc_applied)     /* nv */                                                                   \
  X(xfrc_applied)     /* nbody*6 */                                                              \
  X(mocap_pos)        /* nmocap*3 */                                                             \
  X(mocap_quat)       /* nmocap*4 */                                                             \
  X(qacc)             /* nv */                                                                   \
  X(act_dot)          /* na */                                                                   \
  X(xpos)             /* nbody*3 */                                                              \
  X(xquat)            /* nbody*4 */                                                              \
  X(xmat)             /* nbody*9 */                                                              \
  X(xipos)            /* nbody*3 */                                                              \
  X(ximat)            /* nbody*9 */                                                              \
  X(xanchor)          /* njnt*3 */                                                               \
  X(xaxis)            /* njnt*3 */                                                               \
  X(geom_xpos)        /* ngeom*3 */                                                              \
  X(geom_xmat)        /* ngeom*9 */                                                              \
  X(site_xpos)        /* nsite*3 */                                                              \
  X(site_xmat)        /* nsite*9 */                                                              \
  X(cam_xpos)         /* ncam*3 */                                                               \
  X(cam_xmat)         /* ncam*9 */                                                               \
  X(light_xpos)       /* nlight*3 */                                                             \
  X(light_xdir)       /* nlight*3 */                                                             \
  X(subtree_com)      /* nbody*3 */                                                              \
  X(cdof)             /* nv*6 */                                                                 \
  X(cinert)           /* nbody*10 */                                                             \
  X(crb)              /* nbody*10 */                                                             \
  X(ten_length)       /* ntendon */                                                              \
  X(ten_J)            /* ntendon*nv (dense; constant for fixed tendons) */                       \
  X(ten_velocity)     /* ntendon */                                                              \
  X(actuator_length)  /* nu */                                                                   \
  X(actuator_moment)  /* nu*nv */                                                                \
  X(qM)               /* nv*nv (dense) */                                                        \
  X(qLD)              /* nv*nv (dense Cholesky L) */                                             \
  X(contact_dist)           /* ncon */                                                           \
  X(contact_pos)            /* ncon*3 */                                                         \
  X(contact_frame)          /* ncon*9 */                                                         \
  X(contact_includemargin)  /* ncon */                                                           \
  X(contact_friction)       /* ncon*5 */                                                         \
  X(contact_solref)         /* ncon*2 */                                                         \
  X(contact_solreffriction) /* ncon*2 */                                                         \
  X(contact_solimp)         /* ncon*5 */                                                         \
  X(sensordata)       /* nsensordata (sensor.py:56-440; written by forward passes that include the solver stage) */ \
  X(efc_J)            /* nefc*nv */                                                              \
  X(efc_frictionloss) /* nefc */                                                                 \
  X(efc_D)            /* nefc */                                                                 \
  X(efc_aref)         /* nefc */                                                                 \
  X(efc_force)        /* nefc */                                                                 \
  X(actuator_velocity) /* nu */                                                                  \
  X(cvel)             /* nbody*6 */                                                              \
  X(cdof_dot)         /* nv*6 */                                                                 \
  X(qfrc_bias)        /* nv */                                                                   \
  X(qfrc_passive)     /* nv */                                                                   \
  X(qfrc_gravcomp)    /* nv (written only by models with gravity compensation; otherwise the caller's value stays, passive.py:190-194) */ \
  X(actuator_force)   /* nu */                                                                   \
  X(qfrc_actuator)    /* nv */                                                                   \
  X(qfrc_smooth)      /* nv */                                                                   \
  X(qacc_smooth)      /* nv */                                                                   \
  X(qfrc_constraint)  /* nv */

/* Input-only real leaves that NO stage of the reference writes -- they hold what make_data put there (zeros) or what the caller did -- but its sensor functions
 * read: cacc (accelerometer, sensor.py:383-392), cfrc_int (force / torque, :399-416), subtree_linvel / subtree_angmom (:261-266).  They trail the struct, outside
 * the leaf lists above (no kernel writes them, the goldens' key sets do not change); NULL = zeros.  [B, nbody*6], [B, nbody*6], [B, nbody*3], [B, nbody*3]. */
#define MJH_DATA_EXTRA_IN(X) X(cacc) X(cfrc_int) X(subtree_linvel) X(subtree_angmom)

#define MJH_DATA_I32(X) X(contact_dim) /* ncon */ X(eq_active) /* neq: input, enable / disable each equality constraint (types.py:1103) */

#define MJH_DATA_I64(X)                                                                          \
  X(contact_geom1)       /* ncon */                                                              \
  X(contact_geom2)       /* ncon */                                                              \
  X(contact_geom)        /* ncon*2 */                                                            \
  X(contact_efc_address) /* ncon */

typedef struct mjhData {
#define X(n) void* n;
  MJH_DATA_REALS(X)
#undef X
#define X(n) int32_t* n;
  MJH_DATA_I32(X)
#undef X
#define X(n) int64_t* n;
  MJH_DATA_I64(X)
#undef X
#define X(n) const void* n;
  MJH_DATA_EXTRA_IN(X)
#undef X
} mjhData;

/* ---- entry points ------------------------------------------------------------------------ */

typedef struct mjhModel mjhModel; /* opaque: device-resident constant blob + launch geometry */

/* Builds the device blob for one dtype.  replaces Model.to(device) (types.py:991-1013). */
int mjh_model_create(const mjhModelDesc* desc, int dtype, mjhModel** out);
void mjh_model_destroy(mjhModel* m);

/* forward dynamics for B environments (forward.py:373-401), optionally only a prefix of stages.
 * `work`: as for mjh_step; a forward pass needs it only for models whose max_contact_points selection runs over box / mesh candidates
 * (the convex narrow phase hands its candidate contacts to the constraint phase there), NULL otherwise. */
int mjh_forward(const mjhModel* m, const mjhData* in, mjhData* out, void* work, int64_t B, int stages, int flags,
                void* hip_stream);

/* one simulation step for B environments (forward.py:463-496): _check_state, forward, Euler/RK4.
 * `work`: caller-owned device scratch of mjh_model_work_bytes(m) * B bytes (contents undefined, may be
 * NULL when that is 0).  RK4 keeps its stage Data (stages 1..3 of forward.py:356-367) and the running
 * sums there; max_contact_points over box / mesh pairs keeps the candidate contacts of the convex narrow phase there
 * (collision_driver.py:822-840: every candidate is computed, the closest are kept); other Euler models need none. */
int mjh_step(const mjhModel* m, const mjhData* in, mjhData* out, void* work, int64_t B, int flags, void* hip_stream);
int64_t mjh_model_work_bytes(const mjhModel* m);

/* per-environment ELEMENT count of every mjhData leaf in ABI order (reals, then int32, then int64 leaves): a leaf handed to
 * mjh_forward / mjh_step / mjh_reset_where must hold exactly B * count elements.  The binding validates tensor sizes against
 * this before it passes raw pointers (the kernels index `ptr + env * count` unchecked).  Writes min(n, max) entries, returns n. */
int mjh_model_leaf_counts(const mjhModel* m, int64_t* counts, int max);

/* masked in-place reset of environments -- the env caller's `self._dx[mask] = self._make_batch(n)` (zoo/base.py:266-273,
 * :289-293 partial reset, :327-331 fused auto-reset) as ONE launch without a host sync.  For every environment e with
 * mask[e] != 0, every leaf that is non-NULL in `d` is overwritten with that leaf of `d0` (ONE environment, same dtype; the
 * leaf must be non-NULL there too), except qpos / qvel, which take row e of qpos_rows [B, nq] / qvel_rows [B, nv] when those
 * are non-NULL (the caller's dx0 + reset noise).  Environments with mask[e] == 0 are untouched.  `mask`: B bytes (torch.bool). */
int mjh_reset_where(const mjhModel* m, mjhData* d, const mjhData* d0, const unsigned char* mask, const void* qpos_rows,
                    const void* qvel_rows, int64_t B, void* hip_stream);

/* bytes of dynamic LDS one environment occupies in pipeline phase `phase` (0..4: kinematics, crb/factor,
 * collision/constraint, velocity/acceleration, solve/integrate); the number of phases is 5. */
int mjh_model_lds_bytes(const mjhModel* m, int phase);

/* measurement aid used by bench.py for the per-kernel roofline: while enabled, every kernel launch of mjh_step / mjh_forward is
 * bracketed by HIP events on the launch stream; mjh_debug_phase_times() waits for the most recent call and returns, per launch,
 * the elapsed milliseconds and the kernel id (0..4 pipeline phases, 5 velocity phase with fluid / gravcomp / tendons, 6 solver phase
 * with frictionloss / equality / dense limit rows, 7 constraint phase with those rows or max_contact_points, 8 constraint phase of small
 * models (contact rows straight to the leaf), 9 solver phase as the register solver (mjh_sol2_kernel: two environments per wavefront),
 * 10 convex narrow phase, 11 sensors, 12 kinematics + velocity phases as one kernel -- then 0 and 3 do not appear --, 13 kinematics + crb / factor + velocity as one kernel: small float32
 * models while the batch is one round of its waves, then 12 and 1 do not appear; 14 constraint stage + register solver + integrator as one kernel, 15 the solver's
 * environment sort (opt-in), 16 the whole pass as one kernel (humanoid-class models), 17 kernel 13 on two wavefronts per workgroup, 18 one RK4 stage of a small Newton model
 * as one kernel, 19 that kernel running the constraint phase + first solver tier only).  Returns the number of launches (<= max) or a negative code. */
int mjh_debug_phase_timing(int enable);
int mjh_debug_phase_times(float* ms, int* kernel_ids, int max);

/* diagnostic builds only (-DMJH_STAMPS, tools/stamps.py): device buffer the kernels write their s_memtime section stamps into; NULL turns
 * them off.  The shipped library is built without MJH_STAMPS: the pointer is stored and nothing reads it. */
void mjh_debug_set_stamps(void* dev_ptr);

/* global-memory bytes ONE launch of kernel `kernel` (ids as above) reads and writes per environment in a step: the library's own
 * account of its loads / stores through the Data leaves (csrc/mjh_io.h) -- the per-kernel "algorithmic bytes" of the roofline.
 * read_write_bytes[0] = read, [1] = written.  RK4 models: the mean over the four stage launches of a step (stages 1..3 write a private
 * workspace holding only the leaves a later phase reads).  Returns 0, or -2 when this model's step does not launch that kernel. */
int mjh_model_kernel_io(const mjhModel* m, int kernel, int64_t* read_write_bytes);

/* last error message of the calling thread ("" if none) */
const char* mjh_last_error(void);

/* comma-separated field lists in ABI order (lets the binding assert it is in sync) */
const char* mjh_data_fields(void);
const char* mjh_data_extra_fields(void); /* ... of the input-only leaves that trail mjhData (MJH_DATA_EXTRA_IN) */
int mjh_sizeof_data(void);               /* sizeof(mjhData) as the library was built */
const char* mjh_model_fields(void);
int mjh_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MJHIP_H_ */
