// mjh_device.h -- device-side model view, wave-64 helpers and small-vector math (gfx950 / CDNA4).
//
// Execution model of the whole library: ONE ENVIRONMENT PER 64-LANE WAVEFRONT (one 64-thread
// workgroup).  Per-environment state lives in LDS; lanes parallelise over bodies / joints / dofs /
// constraint rows / contact pairs; Data leaves are batch-major so every global access of a wave is
// a contiguous run of one environment's row.  The formulas restate the reference
// (mujoco_torch/_src/math.py, cited per function) in the same operation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mjhip.h"

#define MJH_WAVE 64
#define MJH_MAX_DEPTH 128
#define MJH_CHAIN_PRE 3  /* joints of a body whose descriptors the tree sweeps keep in registers (the rest are read from the table inside the sweep) */

// ---- constants (reference reads them from the mujoco package) ---------------------------------
#define mjMINVAL 1e-15
#define mjMAXVAL 1e10
#define mjMINIMP 1e-4
#define mjMAXIMP 0.9999
// constants the reference keeps in _CachedConst are float32 literals up-cast (math.py:34-45)
#define MINVAL_CACHED ((float)1e-15)

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { DSBL_CONSTRAINT = 1 << 0, DSBL_SPRING = 1 << 5, DSBL_DAMPER = 1 << 6, DSBL_GRAVITY = 1 << 7,
       DSBL_CLAMPCTRL = 1 << 8, DSBL_WARMSTART = 1 << 9, DSBL_ACTUATION = 1 << 11, DSBL_REFSAFE = 1 << 12,
       DSBL_EULERDAMP = 1 << 15 };
enum { INT_EULER = 0, INT_RK4 = 1, SOL_CG = 1, SOL_NEWTON = 2, CONE_ELLIPTIC = 1 };
enum { CAM_FIXED = 0, CAM_TRACK = 1, CAM_TRACKCOM = 2, CAM_TARGETBODY = 3, CAM_TARGETBODYCOM = 4 };
enum { GAIN_FIXED = 0, GAIN_AFFINE = 1, GAIN_MUSCLE = 2, BIAS_NONE = 0, BIAS_AFFINE = 1, BIAS_MUSCLE = 2 };
enum { DYN_NONE = 0, DYN_INTEGRATOR = 1, DYN_FILTER = 2, DYN_FILTEREXACT = 3, DYN_MUSCLE = 4 };
#define INLINE_CHOL_MAX 16  // math.py:84

// ---- LDS arenas ------------------------------------------------------------------------------------------
// The step is a pipeline of five kernels ("phases"); each phase carves its own per-environment LDS arena
// holding only its working set, so small phases run at high occupancy.  X(name, count, phases): count in
// REALs, phases = bitmask of the phases that keep the array in LDS.
#define PH_KIN 1   /* kinematics + com_pos                      */
#define PH_CRB 2   /* crb + make_m + factor_m                   */
#define PH_CON 4   /* collision + make_constraint               */
#define PH_VEL 8   /* transmission, _velocity, _actuation, _acceleration */
#define PH_SOL 16  /* solve + Euler / RK4 bookkeeping           */
#define MJH_JC_ROWS(m) ((m.sol2_row_cap > 0 && m.sol2_row_cap < m.nefc - m.nf - m.nl) ? m.sol2_row_cap : m.nefc - m.nf - m.nl)
#define PH_SOL2 32   /* register solver (mjh_sol2_kernel): the part of its arena that lives while the solve runs ...            */
#define PH_SOL2T 64  /* ... and the arrays of the integrator tail, carved over it once the constraint rows are dead           */
#define PH_SOL2P 128 /* ... after the state the tail integrates, which is parked for the whole phase                           */
#define PH_KINVEL 256 /* arena of the fused kinematics + velocity kernel (12): the arrays both phases use, then the kinematics-only and the velocity-only arrays OVER each other */
#define PH_KCV 512  /* arena of the fused kinematics + crb + velocity kernel (13): what kinematics and velocity share, then the kinematics-only, the crb-only and the velocity-only arrays OVER each other */
#define PH_KCV2 2048 /* arena of the two-wave form of kernel 13 (mjh_phase_kernel<.., 17, W>): kinematics, then the velocity stage on the first wave BESIDE the crb / factor stage on a second one -- the crb-only arrays get a region of their own */
#define PH_CS 1024  /* arena of the fused constraint + register-solver kernel (mjh_cs_kernel): the contact rows of efc_J stay where the constraint stage built them */
#define MJH_NPHASE 5
#define MJH_NARENA 6  /* arenas carved per model: the five phases + the register solver */
#define MJH_LDS_ARRAYS(X, m)                                                                                   \
  X(qpos, m.nq, PH_KIN | PH_VEL | PH_SOL | PH_SOL2P) X(qpos_con, m.con_general ? m.nq : 0, PH_CON) /* general constraint phase only */ X(qvel, m.nv, PH_CON | PH_VEL | PH_SOL | PH_SOL2P)                     \
  X(act, m.na, PH_VEL | PH_SOL | PH_SOL2P)                                                                                \
  X(xpos, 3 * m.nbody, PH_KIN) X(xquat, 4 * m.nbody, PH_KIN) X(xmat, m.lds_diet ? 0 : 9 * m.nbody, PH_KIN)                      \
  X(xipos, 3 * m.nbody, PH_KIN | PH_VEL) X(ximat, m.lds_diet ? 0 : 9 * m.nbody, PH_KIN)                                         \
  X(xanchor, 3 * m.njnt, PH_KIN) X(xaxis, 3 * m.njnt, PH_KIN)                                                  \
  X(jquat, 4 * m.njnt, PH_KIN) /* per-joint local rotation (or slide offset), computed before the chain walk */ \
  X(geom_xpos, 3 * m.ngeom, 0) X(geom_xmat, 9 * m.ngeom, 0) /* PH_CON, aliased over efc_J (dead before the rows are built): see lds_carve */ \
  X(subtree_com, 3 * m.nbody, PH_KIN | PH_CON | PH_VEL) X(cinert, 10 * m.nbody, PH_KIN | PH_CRB | PH_VEL)      \
  X(crb, 10 * m.nbody, PH_CRB) X(cdof, 6 * m.nv, PH_KIN | PH_CRB | PH_CON | PH_VEL)                            \
  X(cdof_dot, 6 * m.nv, PH_VEL) X(cvel, 6 * m.nbody, PH_VEL) X(cacc, 6 * m.nbody, PH_VEL)                      \
  X(cfrc, m.lds_diet ? 0 : 6 * m.nbody, PH_VEL) X(crb_cdof, 6 * m.nv, PH_CRB) X(sub_mass, m.nbody, PH_KIN)                      \
  X(sub_pos, 3 * m.nbody, PH_KIN)                                                                              \
  X(qMp, m.nv * (m.nv + 1) / 2, PH_CRB) /* lower triangle, packed rows */                                      \
  X(qLD, m.nv * m.nv, 0) /* PH_CRB only, aliased over the dead arrays when the factor is written: see lds_carve */ \
  X(qLDp, m.nv * (m.nv + 1) / 2, PH_SOL) /* lower triangle, packed rows */                                      \
  X(qMs, m.sol_qm_lds ? m.nv * m.nv : 0, PH_SOL) /* only when the solver iterates enough to amortise the copy */ \
  X(qLD_inv, m.nv, PH_SOL)                                                                            \
  X(HL_inv, (m.solver == SOL_NEWTON || !(m.disableflags & DSBL_EULERDAMP)) ? m.nv : 0, PH_SOL)                 \
  X(H, (m.solver == SOL_NEWTON || !(m.disableflags & DSBL_EULERDAMP)) ? m.nv * (m.nv + 1) / 2 : 0, PH_SOL) /* packed lower rows */ \
  X(HL, (m.solver == SOL_NEWTON || !(m.disableflags & DSBL_EULERDAMP)) ? m.nv * m.nv : 0, PH_SOL)              \
  /* integrator tail of the register solver (implicit joint damping, forward.py:313-328): n <= 16 factorises M + dt D from packed rows in LDS; 16 < n <= 28 factorises in   \
     registers straight from the qM leaf and only needs a two-column broadcast buffer -- the n x n image HL (729 reals for the humanoid) made this arena 11.8 KB: six   \
     workgroups per CU where B = 4096 needs eight (profiles/r04/notes.md) */                                                                                          \
  X(H2, (!(m.disableflags & DSBL_EULERDAMP) && m.nv <= 16) ? m.nv * (m.nv + 1) / 2 : 0, PH_SOL2T) X(chol_col, (!(m.disableflags & DSBL_EULERDAMP) && m.nv > 16) ? 64 : 0, PH_SOL2T) \
  X(con_dist, m.ncand, PH_CON) X(con_pos, 3 * m.ncand, PH_CON) X(con_frame, 9 * m.ncand, PH_CON) /* candidate contacts (== the contacts unless max_contact_points selects) */ \
  X(i_con_src, m.topk ? m.ncon : 0, PH_CON) /* top-k: candidate kept in each contact slot (ints) */                  \
  X(efc_J, m.con_general ? m.nefc * m.nv : (m.con_direct ? 0 : (m.nefc - m.nl) * m.nv), PH_CON) /* plain instantiation: contact rows only, none when they go straight to the leaf (con_direct) */ X(efc_jl, m.con_general ? 0 : m.nl, PH_CON) /* plain: contact rows only + the limit rows' single entries */ X(i_con_act, m.con_direct ? m.ncon : 0, PH_CON) X(i_crow_act, m.con_direct ? (m.crow_by_con ? m.ncon : m.nefc - m.nl) : 0, PH_CON) /* small models: compact list of the active contacts; activity per contact row -- per CONTACT where dense row q is contact q / con_rows: 480 B less for the ant, whose 5200 B arena fitted 15 two-environment workgroups per CU, i.e. THREE rounds of waves at B = 16384 (30 + 30 + 4 environments per CU) instead of two (ints) */ X(efc_D, m.nefc, PH_SOL)                                            \
  X(efc_Jc, MJH_JC_ROWS(m) * m.nv, PH_SOL | PH_SOL2) /* dense rows of the contacts (sol2_row_cap: the register solver's first tier keeps 32 of them) */                                   \
  X(efc_Jl, m.nf + m.nl, PH_SOL) /* the single non-zero of each frictionloss / joint-limit row (column crow_dof[r]) */ \
  X(efc_fl, m.nf + m.nft, PH_SOL) /* frictionloss of the dof- and tendon-friction rows */                \
  X(i_row_src, m.nefc - m.ne - m.nf - m.nft - m.nl - m.nlb - m.nlt, PH_SOL) X(i_row_dst, m.nefc - m.ne - m.nf - m.nft - m.nl - m.nlb - m.nlt, PH_SOL) /* active-contact row tables (ints) */ \
  X(i_lim_dof, m.nf + m.nl, PH_SOL) X(i_dof_limrow, (m.nf + m.nl) ? 2 * m.nv : 0, PH_SOL) /* int copies of the model tables: lane-indexed reads stay on chip */ \
  X(efc_aref, m.nefc, PH_SOL)                                                                                  \
  X(efc_pos, m.ne + m.nf + m.nft + m.nlb + m.nl + m.nlt, PH_CON) X(efc_pos_norm, m.con_general ? m.ne + m.nf + m.nft + m.nlb + m.nl + m.nlt : 0, PH_CON) /* plain: equal to efc_pos for slide / hinge limits */ X(efc_invweight, m.ne + m.nf + m.nft + m.nlb + m.nl + m.nlt, PH_CON) /* contact rows recompute theirs */ \
  X(act_length, m.nu, PH_VEL) X(act_velocity, m.nu, PH_VEL) X(act_force, m.nu, PH_VEL) X(act_rot, m.act_has_rot ? 3 * m.nu : 0, PH_VEL) X(ten_len, m.ntendon, PH_VEL) X(ten_frc, m.ntendon, PH_VEL)                         \
  X(act_dot, m.na, PH_VEL | PH_SOL | PH_SOL2P)                                                                            \
  X(qfrc_bias, m.nv, PH_VEL) X(qfrc_passive, m.nv, PH_VEL) X(qfrc_actuator, m.nv, PH_VEL) X(qfrc_gravcomp, m.has_gravcomp ? m.nv : 0, PH_VEL)                      \
  X(qfrc_smooth, m.nv, PH_VEL | PH_SOL | PH_SOL2T) X(qacc_smooth, m.nv, PH_SOL) /* (the velocity stage never touched it: _acceleration's solve sits at the head of the solver phase) */                                  \
  X(qacc_warm, m.nv, PH_SOL) X(qacc, m.nv, PH_SOL | PH_SOL2T) X(qfrc_constraint, m.nv, PH_SOL | PH_SOL2T)                            \
  X(s_qacc, m.nv, PH_SOL) X(s_qfrc, m.nv, PH_SOL) X(s_Ma, m.nv, PH_SOL) X(s_grad, m.nv, PH_SOL | PH_SOL2T)                \
  X(s_Mgrad, m.nv, PH_SOL | PH_SOL2T) X(s_search, m.nv, PH_SOL) X(s_mv, m.nv, PH_SOL | PH_SOL2T) X(s_pgrad, m.nv, PH_SOL | PH_SOL2T)            \
  X(s_pMgrad, m.nv, PH_SOL) X(tmp_nv, m.nv, PH_SOL | PH_SOL2T) X(tmp_nv2, m.nv, PH_SOL | PH_SOL2T)                                   \
  X(s_Jaref, m.nefc, PH_SOL) X(s_force, m.nefc, PH_SOL) X(s_jv, m.nefc, PH_SOL) X(s_quad, 3 * m.nefc, PH_SOL)  \
  X(tmp_nq, m.nq, PH_SOL | PH_SOL2T)                                                                                        \
  X(r_vs, m.nv, PH_SOL2) X(r_vs2, m.nv, PH_SOL2) /* register solver: vectors staged for broadcast reads */     \
  X(r_src, MJH_JC_ROWS(m), PH_SOL2) /* compact dense row -> Data row (ints; as many as this tier keeps rows) */ X(r_dst, (m.nefc - m.nl + 1) / 2, PH_SOL2) /* Data row -> compact row, 16-bit entries, 0xffff = inactive */ X(r_pg, 2 * m.nv, PH_SOL2) /* previous gradient pair of the Polak-Ribiere step */ X(r_fs, MJH_JC_ROWS(m) + m.nl, PH_SOL2) /* ... and the row forces: dense rows first, then J * force of the single-column rows */

// (16-bit entries, round 6: offsets are counted in REALs and an arena is at most 160 KB = 40 K floats; as ints two of these tables filled 744 B of the 4 KiB kernel-argument
//  block and left no room for the third one the stage kernel of RK4 models needs)
struct LdsOff {
#define X(n, c, p) unsigned short n;
  MJH_LDS_ARRAYS(X, _)
#undef X
};


// ---- device view of the model -------------------------------------------------------------------
// Pointers into one device blob (built by mjh_model_create).  REAL arrays are stored in the compute
// dtype.  All indices are wave-uniform in the kernels, so these loads become scalar (SMEM) loads.
template <typename REAL>
struct DevModel {
#define X(n) int n;
  MJH_MODEL_INTS(X)
#undef X
  REAL timestep, impratio, gravity[3];
  REAL density, viscosity, wind[3];  // fluid model (passive.py:31-78)
  REAL magnetic[3];                  // opt.magnetic (magnetometer sensors)
  int has_fluid;
  int has_gravcomp;  // some body_gravcomp != 0 (device.py:667)
  double meaninertia, tolerance, ls_tolerance;  // python floats in the reference (solver.py:256-265)
#define X(n) const int* n;
  MJH_MODEL_INT_ARRAYS(X)
#undef X
#define X(n) const REAL* n;
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
  // derived topology tables (host-computed, mjh_abi.hip)
  const int* body_depth;       // nbody: number of non-world ancestors-or-self
  const int* body_chain;       // nbody*max_depth: chain[b][k] = k-th body on the path world -> b (k=0: child of world)
  const int* body_subtree_end; // nbody: one past the last body of b's subtree (bodies are in DFS order)
  const unsigned long long* body_dofmask;  // nbody * mask_words: dofs whose body is an ancestor-or-self of b (bit d & 63 of word d >> 6)
  int mask_words;                          // 64-bit words per dof mask: 1 up to 64 dofs
  int kv_defer;                            // fused kinematics + velocity kernel: the kinematics stage's leaves are stored by the velocity stage, in front of its LDS-only sweeps (the arena keeps them intact until then)
  int big;                                 // nv > 64: the kernels with multi-word mask reads serve the model (5, 7; kinematics + velocity not fused)
  const int* chain_dof;        // nbody*max_depth: dofadr | dofnum << 16 of the k-th body on the path world -> b
  const int* chain_jnt;        // nbody*max_depth*max_jnt: (type + 1) | dofadr << 8 of that body's joints in order, 0 = none
  int max_jnt;
  // kinematics by pointer jumping (null: every lane walks world -> its body), one allocation (the kernel argument block was at its 4 KiB limit when this was added; 3,936 B since the arena offset tables went to 16-bit entries in round 6):
  //   int  anc[R][nbody]   the ancestor 2^r levels above body b, 0 where that is the world or beyond; R = ceil(log2(max_depth))
  //   REAL start[nbody][7] body_pos, body_quat (from byte offset 8 * ((R * nbody + 1) / 2)); for children of the world composed with the world body's own frame
  const int* kin_tab;
  const int* efc_row_con;                  // nefc: contact index of a contact row, -1 for limit rows
  const int* rf_sensor;                    // nrfq: rangefinder (sns_* index) of entry q of rf_geom
  const int* rf_site;                      // nrfq: ... its site
  const int* rf_gtype;                     // nrfq: type of geom rf_geom[q]
  const REAL* rf_gsize;                    // 3 * nrfq: ... and its size
  int sns_full;                            // some sensor is of a kind the lean sensor kernel does not carry
  int nrfq;                                // entries of rf_geom = (rangefinder, geom) ray tests per environment
  const int* efc_row_eq;                   // ne: eq_* table entry of an equality row
  int max_depth;
  int kin_lvl;                             // kinematics as a level sweep with every lane's constants read up front (one body per lane in every instantiation the model runs at; bit-identical to the walk).  (Sits in the padding behind max_depth.)
  const int* qm_pair;                      // nqmpair: i << 8 | j (j <= i) of the inertia-matrix entries that can be non-zero
  int nqmpair;
  const int* qm_slot;                      // nv*nv: packed lower-triangle slot holding entry (i, j), -1 where qM is structurally zero
  const REAL* act_moment;                  // nu*nv: the (constant) moment matrix of joint transmissions (reference device.py:588-629)
  const int* dof_act_adr;                  // nv+1: CSR of the actuators driving each dof, in actuator order
  const int* dof_act_id;
  const REAL* dof_act_coef;                // per entry of dof_act_id: the constant moment coefficient (a gear component)
  const int* dof_frc_lim;                  // nv: jnt_actfrclimited of the dof's joint (one read instead of a chain dof -> joint -> flag -> range)
  const REAL* dof_frc_range;               // 2*nv: jnt_actfrcrange of the dof's joint
  const int* dof_act_rot;                  // per entry: -1, or the component of the actuator's rotated gear axis that is the coefficient
  const int* act_ent_adr;                  // nu+1: CSR of the non-zeros of each actuator's moment row
  const int* act_ent_dof;
  const REAL* act_ent_coef;
  const int* act_ent_rot;
  const REAL* ten_J0;                      // ntendon*nv: the constant Jacobian of the fixed tendons (ten_J[t, dof] = coef, last term wins: smooth.py:492-494)
  const REAL* ten_JTAJ;                    // nv*(nv+1)/2 packed lower rows: J^T diag(tendon_armature) J, a model constant for fixed tendons (smooth.py:500-522); valid when has_ten_armature
  int has_ten_armature;
  int con_direct;                          // plain constraint phase of a small model: contact rows written straight to the efc_J leaf, two environments per wavefront
  int con_general;                         // equality / frictionloss / ball- or tendon-limit rows present: constraint phase kernel 7
  int act_simple;                          // every actuator drives a slide / hinge joint
  int act_has_rot;                         // some JOINTINPARENT transmission on a ball / free joint (moment depends on qpos)
  float inv_nv;                            // 1 / nv for the index splits below
  const int* dof_limrow;                   // 2*nv: the (up to two: frictionloss, then joint limit) single-column rows of dof d, -1 = none
  const int* lim_dof;                      // nf+nl: dof of single-column row r (frictionloss rows, then joint-limit rows)
  int sol_qm_lds;                          // solver keeps qM in LDS (many iterations) instead of re-reading it from L2
  int con_rows;                            // constraint rows of every contact when they all have the same number (one condim), else 0
  int nt_all;                              // the model's full passes run as the whole-pass kernel with the on-chip seam: geom frames, subtree_com, cdof, contact_dist are output only (non-temporal stores)
  int all_handoff;                         // whole-pass kernel: the constraint stage's inputs cross the seam on chip (ngeom, nbody <= 32; MJH_ALL_HANDOFF=0: off)
  int lds_diet;                            // small models (four environments per wavefront): xmat / ximat are stored from registers and recomputed from xquat where read again, the subtree force sums fold into qfrc_bias -- the arena of kernel 13 drops from 3064 to 2272 B for the ant, i.e. 16 four-environment workgroups per CU fit (one round of waves at B = 16384 instead of two)
  int crow_by_con;                         // ... and the contact rows are in contact order, no gaps: dense row q belongs to contact q / con_rows (the small-model constraint phase then keeps its activity flags per CONTACT)
  int sol2_row_cap;                        // only while the arena of the register solver's first tier is carved (mjhip.hip): dense rows it keeps; 0 otherwise
  int pair_cull_on;                        // non-zero: collision() in RK4 stages 1..3 narrow-phases only the pairs whose bounding spheres are within reach (MJH_PAIR_CULL=0: off)
  int sol2_incr;                           // non-zero: the register solver updates the Newton Hessian by the rows whose activity flipped instead of rebuilding it every iteration (MJH_SOL2_INCR=0: off)
  int sol2_hs;                             // non-zero: the register solver builds the Newton Hessian with block matrix instructions (float32, nv <= 16)
  // static per-row / per-contact tables of the plain constraint phase (no max_contact_points: slot c IS candidate c), so that a lane reaches
  // everything a row needs with ONE table read indexed by its own row number instead of a chain row -> contact -> geom -> body
  const REAL* crow_par;                    // 9 * ncrow, parameter-major [k * ncrow + q]: solref (2, friction rows of elliptic cones resolved), solimp (5), invweight, includemargin of contact row q
  const int* crow_info;                    // ncrow: contact | (row within the contact) << 16 | (elliptic cone and condim > 1) << 24
  int ncrow;                               // contact rows = nefc - (rows ahead of the contacts)
  const int* con_body;                     // 4 * ncon: body1, body2, root body of body1, of body2
  const unsigned long long* con_dmask;     // 2 * ncon: body_dofmask of body1, body2
  const REAL* pair_cull;                   // 2 * npair: [reach^2, r1 + r2] of the bounding-sphere test RK4 stages 1..3 put before the narrow phase (reach^2 < 0: never culled)
  const int* cvx_pairs;                    // ncvxpair: indices (into pair_*) of the pairs with a convex pair function
  int ncvxpair;
};

// ---- wave helpers --------------------------------------------------------------------------------------
// lane of this thread inside its wavefront.  Workgroups are one wavefront (launch bound 64: the mask folds away) except the two-wave form of kernel 13 (mjh_phase_kernel<.., 17, W>)
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (MJH_WAVE - 1)); }

// LDS visibility point for a single-wave workgroup (s_waitcnt + barrier; the barrier is free for 1 wave).
// A workgroup is one wavefront: what a "barrier" has to order is the LDS traffic of its lanes.  __syncthreads() carries a workgroup-scope
// fence over GLOBAL memory too, i.e. s_waitcnt vmcnt(0): every barrier behind a burst of leaf stores stalled the wave until L2 had taken all of
// them (the stores of one phase are 60 - 80 MB per launch, issued by all waves at the same moment) -- 10 of the fused kinematics + velocity
// kernel's 75 us on the humanoid.  The fences below are restricted to the LDS address space; the stores drain behind the arithmetic that follows.
// (round 5: no s_barrier instruction between the fences.  For a one-wave workgroup the hardware treats it as a no-op anyway -- a wave's LDS operations complete in issue order --
// and the two-wave form of kernel 13 needs the stage functions' internal sync points to stay INSIDE a wave: a real barrier there would pair up with the other wave's.)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// ... and the full barrier, for the one place where a wave reads back through L2 what it has just stored (the small-model constraint phase)
__device__ __forceinline__ void wave_sync_global() { __syncthreads(); }

template <typename T>
__device__ __forceinline__ T wave_bcast(T v, int src_lane) { return __shfl(v, src_lane, MJH_WAVE); }

// ---- wave all-reduce (sum) on DPP: quad_perm xor-1 / xor-2, row_half_mirror, row_mirror give every lane the
// sum of its 16-lane row without touching LDS; the four row sums are read back as scalars (v_readlane) and
// added uniformly, so every lane ends with the same bits.  EXEC must be all ones (callers are wave-uniform).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));  // (folds into v_add_f32_dpp)
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  int lo = (int)(u & 0xffffffffull), hi = (int)(u >> 32);
  // every lane has a source under the controls used here: no `old` value to set up (update_dpp(0, ...) cost a v_mov of zero per half and step)
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float read_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int read_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double read_lane(double v, int lane) {
  unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  int lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffull), lane);
  int hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#ifdef MJH_NO_DPP
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, MJH_WAVE);
  return v;
#else
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return ((read_lane(v, 0) + read_lane(v, 16)) + read_lane(v, 32)) + read_lane(v, 48);
#endif
}
__device__ __forceinline__ int wave_any(int p) { return __any(p); }

// w -> (w / n, w % n) for a wave-uniform runtime n without the ~30-instruction integer division: float estimate + fix-up
__device__ __forceinline__ void split_index(int w, int n, float inv_n, int& q, int& r) {
  q = (int)((float)w * inv_n);
  r = w - q * n;
  if (r < 0) { q--; r += n; }
  if (r >= n) { q++; r -= n; }
}

// ---- sub-wave helpers: W lanes (64 or 32) serve one environment, 64 / W environments share a wavefront ----------------
template <int W> __device__ __forceinline__ int sub_lane() { return (W == MJH_WAVE) ? lane_id() : (int)(threadIdx.x & (W - 1)); }
// any() over the lanes of this environment
template <int W>
__device__ __forceinline__ bool sub_any(bool p) {
  if (W == MJH_WAVE) return __any(p) != 0;
  const unsigned long long m = __ballot(p);
  const int sh = lane_id() & ~(W - 1);
  return ((m >> sh) & ((W >= 64) ? ~0ull : ((1ull << W) - 1))) != 0;
}
// number of lanes below this one in the environment's lane group whose predicate holds, and the group's total: one ballot and two
// population counts, where a shuffle-based prefix sum is log2(W) dependent LDS-pipe round trips
template <int W>
__device__ __forceinline__ int sub_prefix_count(bool p, int& total) {
  const unsigned long long m = __ballot(p);
  const int sh = lane_id() & ~(W - 1), l = (int)(threadIdx.x & (W - 1));
  const unsigned long long g = (m >> sh) & ((W >= 64) ? ~0ull : ((1ull << W) - 1));
  total = __popcll(g);
  return __popcll(g & ((1ull << l) - 1ull));
}
// value of lane k OF THIS ENVIRONMENT's lane group (k wave-uniform): v_readlane for a whole wave; for two 32-lane environments
// one v_readlane per half and a select -- scalar broadcasts on the VALU / SALU, where a bpermute would be an LDS-pipe round trip
// in the middle of the triangular solves' dependent chains
template <int W, typename T>
__device__ __forceinline__ T sub_read(T v, int k) {
  if (W == MJH_WAVE) return read_lane(v, k);
  if (W == 32) {
    const T a = read_lane(v, k), b = read_lane(v, 32 + k);
    return lane_id() < 32 ? a : b;
  }
  return __shfl(v, (lane_id() & ~(W - 1)) + k, MJH_WAVE);
}
// value of lane K (0..15) of each 16-lane DPP row, in every lane of that row: three DPP moves, no LDS-pipe round trip and no SGPR hop.
// quad_perm [a, a, a, a] leaves lane 4 q + a in all lanes of quad q; row_half_mirror (lane i <- lane 7 - i of its half) copies the quad that
// holds lane K into its neighbour -- bank_mask enables the destination write per quad, the other quads keep `old` -- and row_mirror (lane i <-
// lane 15 - i) copies those two into the other half.  The triangular kernels for n <= 16 keep one dof per lane inside ONE row (the first row of the
// environment's lane group), so this is their broadcast: a v_readlane pair + select per 32-lane half costs five instructions and two SGPR
// hazards, a bpermute a dependent LDS round trip inside the substitutions' chains.
template <int K>
__device__ __forceinline__ int row_bcast_i32(int v) {
  constexpr int a = K & 3, kq = (K >> 2) & 3;
  int x = __builtin_amdgcn_update_dpp(0, v, a * 0x55, 0xf, 0xf, false);
  x = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 1 << (kq ^ 1), false);
  x = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, (1 << (3 - kq)) | (1 << (3 - (kq ^ 1))), false);
  return x;
}
template <int K> __device__ __forceinline__ int row_bcast_k(int v) { return row_bcast_i32<K>(v); }
template <int K> __device__ __forceinline__ float row_bcast_k(float v) { return __builtin_bit_cast(float, row_bcast_i32<K>(__builtin_bit_cast(int, v))); }
template <int K> __device__ __forceinline__ double row_bcast_k(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = row_bcast_i32<K>((int)(u & 0xffffffffull)), hi = row_bcast_i32<K>((int)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// k is a constant after unrolling at every call site: the switch folds to one case
template <typename T>
__device__ __forceinline__ T row_bcast(T v, int k) {
  switch (k & 15) {
    case 0: return row_bcast_k<0>(v); case 1: return row_bcast_k<1>(v); case 2: return row_bcast_k<2>(v); case 3: return row_bcast_k<3>(v);
    case 4: return row_bcast_k<4>(v); case 5: return row_bcast_k<5>(v); case 6: return row_bcast_k<6>(v); case 7: return row_bcast_k<7>(v);
    case 8: return row_bcast_k<8>(v); case 9: return row_bcast_k<9>(v); case 10: return row_bcast_k<10>(v); case 11: return row_bcast_k<11>(v);
    case 12: return row_bcast_k<12>(v); case 13: return row_bcast_k<13>(v); case 14: return row_bcast_k<14>(v); default: return row_bcast_k<15>(v);
  }
}
// element k of an n-vector kept one element per lane (n <= NMAX) in the lanes of this environment's group
template <int W, int NMAX, typename T>
__device__ __forceinline__ T dof_read(T v, int k) {
  if constexpr (W < MJH_WAVE && NMAX <= 16) return row_bcast(v, k);
  else return sub_read<W>(v, k);
}
// x[even row] + x[odd row] of each pair of 16-lane rows, in all 32 lanes of the pair: v_permlane16_swap (gfx950) exchanges the odd rows of one register with the even rows
// of the other -- fed two copies of x it leaves [r0, r0, r2, r2] and [r1, r1, r3, r3]; one add.  (Was: four v_readlane per dword, scalar adds, a select.)
__device__ __forceinline__ float row_pair_sum(float x) {
  const int u = __builtin_bit_cast(int, x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ double row_pair_sum(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const int lo = (int)(u & 0xffffffffull), hi = (int)(u >> 32);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[0] << 32) | (unsigned)rl[0]);
  const double b = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[1] << 32) | (unsigned)rl[1]);
  return a + b;
}
// sum over the lanes of this environment's group, every lane of the group ends with the same bits.  W = 32: the DPP steps leave
// each 16-lane row's sum in all its lanes; rows 0 + 1 serve the first environment, rows 2 + 3 the second.
template <int W, typename T>
__device__ __forceinline__ T sub_sum(T v) {
  if (W == MJH_WAVE) return wave_sum(v);
  if (W == 16) { v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v); return v; }  // one 16-lane DPP row per environment
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return row_pair_sum(v);  // row 0 + row 1 in the first environment's lanes, row 2 + row 3 in the second's
}

// ---- small-vector math, reference math.py ------------------------------------------------------------------
template <typename R> __device__ __forceinline__ R r_sqrt(R x);
template <> __device__ __forceinline__ double r_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float r_sqrt<float>(float x) { return sqrtf(x); }
template <typename R> __device__ __forceinline__ R r_sin(R x);
template <> __device__ __forceinline__ double r_sin<double>(double x) { return sin(x); }
template <> __device__ __forceinline__ float r_sin<float>(float x) { return sinf(x); }
template <typename R> __device__ __forceinline__ R r_cos(R x);
template <> __device__ __forceinline__ double r_cos<double>(double x) { return cos(x); }
template <> __device__ __forceinline__ float r_cos<float>(float x) { return cosf(x); }
template <typename R> __device__ __forceinline__ R r_atan2(R y, R x);
template <> __device__ __forceinline__ double r_atan2<double>(double y, double x) { return atan2(y, x); }
template <> __device__ __forceinline__ float r_atan2<float>(float y, float x) { return atan2f(y, x); }
template <typename R> __device__ __forceinline__ R r_pow(R x, R y);
template <> __device__ __forceinline__ double r_pow<double>(double x, double y) { return pow(x, y); }
template <> __device__ __forceinline__ float r_pow<float>(float x, float y) { return powf(x, y); }
template <typename R> __device__ __forceinline__ R r_exp(R x);
template <> __device__ __forceinline__ double r_exp<double>(double x) { return exp(x); }
template <> __device__ __forceinline__ float r_exp<float>(float x) { return expf(x); }
template <typename R> __device__ __forceinline__ R r_abs(R x) { return x < 0 ? -x : x; }
template <> __device__ __forceinline__ double r_abs<double>(double x) { return fabs(x); }
template <> __device__ __forceinline__ float r_abs<float>(float x) { return fabsf(x); }
template <typename R> __device__ __forceinline__ bool r_finite(R x) { return isfinite(x); }
__device__ __forceinline__ float r_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double r_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float r_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double r_min(double a, double b) { return __builtin_fmin(a, b); }

template <typename R>
__device__ __forceinline__ void cross3(const R* a, const R* b, R* o) {  // math.py:63-76
  R o0 = a[1] * b[2] - a[2] * b[1];
  R o1 = a[2] * b[0] - a[0] * b[2];
  R o2 = a[0] * b[1] - a[1] * b[0];
  o[0] = o0; o[1] = o1; o[2] = o2;
}
template <typename R>
__device__ __forceinline__ R dot3(const R* a, const R* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

template <typename R, int N>
__device__ __forceinline__ R norm_n(const R* x) {  // math.norm :196-213
  bool all_zero = true;
#pragma unroll
  for (int i = 0; i < N; i++) all_zero = all_zero && (x[i] == 0);
  if (all_zero) return 0;
  R s = 0;
#pragma unroll
  for (int i = 0; i < N; i++) s += x[i] * x[i];
  return r_sqrt<R>(s);
}
template <typename R, int N>
__device__ __forceinline__ R normalize_n(R* x) {  // normalize_with_norm :216-230
  R nn = norm_n<R, N>(x);
  R den = nn + (R)1e-6 * (R)(nn == 0);
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = x[i] / den;
  return nn;
}
template <typename R>
__device__ __forceinline__ void rotate(const R* v, const R* q, R* o) {  // math.rotate :246-261
  R s = q[0];
  const R* u = q + 1;
  R uv = dot3(u, v), uu = dot3(u, u);
  R c[3], r[3];
  cross3(u, v, c);
#pragma unroll
  for (int i = 0; i < 3; i++) r[i] = 2 * (uv * u[i]) + (s * s - uu) * v[i];
#pragma unroll
  for (int i = 0; i < 3; i++) o[i] = r[i] + 2 * s * c[i];
}
template <typename R>
__device__ __forceinline__ void quat_mul(const R* u, const R* v, R* o) {  // :283-300
  R o0 = u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3];
  R o1 = u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2];
  R o2 = u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1];
  R o3 = u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0];
  o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
template <typename R>
__device__ __forceinline__ void quat_to_mat(const R* q, R* m) {  // :323-351
  R p00 = q[0] * q[0], p01 = q[0] * q[1], p02 = q[0] * q[2], p03 = q[0] * q[3];
  R p11 = q[1] * q[1], p12 = q[1] * q[2], p13 = q[1] * q[3];
  R p22 = q[2] * q[2], p23 = q[2] * q[3], p33 = q[3] * q[3];
  m[0] = p00 + p11 - p22 - p33;
  m[1] = 2 * (p12 - p03);
  m[2] = 2 * (p13 + p02);
  m[3] = 2 * (p12 + p03);
  m[4] = p00 - p11 + p22 - p33;
  m[5] = 2 * (p23 - p01);
  m[6] = 2 * (p13 - p02);
  m[7] = 2 * (p23 + p01);
  m[8] = p00 - p11 - p22 + p33;
}
template <typename R>
__device__ __forceinline__ void axis_angle_to_quat(const R* axis, R angle, R* q) {  // :363-374
  R s = r_sin<R>(angle * (R)0.5), c = r_cos<R>(angle * (R)0.5);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
template <typename R>
__device__ __forceinline__ void quat_to_axis_angle(const R* q, R* axis, R& angle) {  // :354-360
  axis[0] = q[1]; axis[1] = q[2]; axis[2] = q[3];
  const R sin_a_2 = normalize_n<R, 3>(axis);
  R a = 2 * r_atan2<R>(sin_a_2, q[0]);
  const R pi = (R)3.14159265358979323846;
  if (a > pi) a = a - 2 * pi;
  angle = a;
}
template <typename R>
__device__ __forceinline__ void quat_sub(const R* u, const R* v, R* o) {  // :276-280, :354-360
  R vi[4] = {v[0], -v[1], -v[2], -v[3]}, q[4];
  quat_mul(vi, u, q);
  R axis[3] = {q[1], q[2], q[3]};
  R sin_a_2 = normalize_n<R, 3>(axis);
  R a = 2 * r_atan2<R>(sin_a_2, q[0]);
  const R pi = (R)3.14159265358979323846;
  if (a > pi) a = a - 2 * pi;
#pragma unroll
  for (int i = 0; i < 3; i++) o[i] = axis[i] * a;
}
template <typename R>
__device__ __forceinline__ void quat_integrate(const R* q, const R* w, R dt, R* o) {  // :377-383
  R v[3] = {w[0], w[1], w[2]};
  R nrm = normalize_n<R, 3>(v);
  R angle = dt * nrm, qr[4], r[4];
  axis_angle_to_quat(v, angle, qr);
  quat_mul(q, qr, r);
  normalize_n<R, 4>(r);
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = r[i];
}
template <typename R>
__device__ __forceinline__ void inert_mul(const R* in, const R* v, R* o) {  // :415-429
  const R* pos = in + 6;
  R mass = in[9];
  R c1[3], c2[3];
  cross3(pos, v + 3, c1);
  cross3(pos, v, c2);
  o[0] = (in[0] * v[0] + in[3] * v[1] + in[4] * v[2]) + c1[0];
  o[1] = (in[3] * v[0] + in[1] * v[1] + in[5] * v[2]) + c1[1];
  o[2] = (in[4] * v[0] + in[5] * v[1] + in[2] * v[2]) + c1[2];
#pragma unroll
  for (int i = 0; i < 3; i++) o[3 + i] = mass * v[3 + i] - c2[i];
}
template <typename R>
__device__ __forceinline__ void motion_cross(const R* u, const R* v, R* o) {  // :455-467
  R a[3], b[3], c[3];
  cross3(u, v, a);
  cross3(u + 3, v, b);
  cross3(u, v + 3, c);
#pragma unroll
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
template <typename R>
__device__ __forceinline__ void motion_cross_force(const R* v, const R* f, R* o) {  // :470-482
  R a[3], b[3], c[3];
  cross3(v, f, a);
  cross3(v + 3, f + 3, b);
  cross3(v, f + 3, c);
#pragma unroll
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
template <typename R>
__device__ __forceinline__ void make_frame(const R* a_in, R* frame) {  // orthogonals / make_frame :485-500
  R a[3] = {a_in[0], a_in[1], a_in[2]};
  normalize_n<R, 3>(a);
  R b[3] = {0, 0, 0};
  if ((R)-0.5 < a[1] && a[1] < (R)0.5) b[1] = 1; else b[2] = 1;
  R ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
#pragma unroll
  for (int i = 0; i < 3; i++) b[i] = b[i] - a[i] * ab;
  normalize_n<R, 3>(b);
  R any = (R)((a[0] != 0) || (a[1] != 0) || (a[2] != 0));
#pragma unroll
  for (int i = 0; i < 3; i++) b[i] = b[i] * any;
  R c[3];
  cross3(a, b, c);
#pragma unroll
  for (int i = 0; i < 3; i++) { frame[i] = a[i]; frame[3 + i] = b[i]; frame[6 + i] = c[i]; }
}
template <typename R>
__device__ __forceinline__ void local_to_global(const R* wpos, const R* wquat, const R* lpos, const R* lquat, R* pos, R* mat) {
  // support.local_to_global :99-108
  R r[3], q[4];
  rotate(lpos, wquat, r);
#pragma unroll
  for (int i = 0; i < 3; i++) pos[i] = wpos[i] + r[i];
  quat_mul(wquat, lquat, q);
  quat_to_mat(q, mat);
}
