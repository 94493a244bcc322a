// mjh_io.h -- the global-memory extents each kernel of a step touches, per environment and launch (bytes).
//
// This is the "algorithmic bytes" account behind bench.py's per-kernel roofline (mjh_model_kernel_io): it lives next to the
// kernels and restates, load by load and store by store, what their code moves through the Data leaves -- the PACKED triangle
// of the factor (not the whole qLD leaf), the contact rows of efc_J plus ONE entry per single-column row (not the dense leaf),
// qM once per solve, and so on.  Model constants (tens of KB, L2-resident) are not counted.  `do_step` = 1: a plain forward / Euler step
// launch (stage 0 of an RK4 step, which writes the returned Data in full); `do_step` = 2: a launch of RK4 stages 1..3, which write a private
// workspace Data holding only the leaves a later phase of the same stage reads (mjhip.hip: kStageLeaves) -- and, in the small-model constraint
// phase, only the rows of the active contacts (data dependent: counted as zero, the figure is a lower bound there).
// tools/hbm_traffic.sh measures the PMC counterpart; profiles/r02/notes.md compares the two.
#pragma once
#include "mjh_device.h"

template <typename REAL>
inline int mjh_kernel_io(const DevModel<REAL>& m, int kernel, int do_step, int64_t* read_bytes, int64_t* write_bytes) {
  const int64_t R = (int64_t)sizeof(REAL);
  const int64_t nq = m.nq, nv = m.nv, nu = m.nu, na = m.na, nb = m.nbody, nj = m.njnt, ng = m.ngeom, ncon = m.ncon, nefc = m.nefc;
  const int64_t nsingle = m.nf + m.nl;             // single-column rows (dof frictionloss, slide / hinge limits)
  const int64_t ndense = nefc - nsingle;           // dense rows (equality, ball / tendon limits, contacts)
  const int64_t tri = nv * (nv + 1) / 2;
  const bool general_con = m.con_general != 0, general_sol = (m.nf > 0 || m.nft > 0 || m.ne > 0 || m.nlb > 0 || m.nlt > 0);
  const bool opt_vel = (m.has_fluid || m.has_gravcomp || m.ntendon > 0 || m.big);
  int64_t rd = 0, wr = 0;
  const bool scratch = do_step == 2;  // RK4 stage 1..3
  switch (kernel) {
    case 0:  // kinematics + com_pos: load_qpos; kinematics() stores; com_pos() stores
      rd = nq + 7 * (int64_t)m.nmocap;
      wr = nq + (3 + 4 + 9 + 3 + 9) * nb + 6 * nj + 12 * ng + 12 * (int64_t)m.nsite + 12 * (int64_t)m.ncam + 6 * (int64_t)m.nlight
           + 3 * nb + 10 * nb + 6 * nv;
      if (scratch) wr = nq + 3 * nb /* xipos */ + 12 * ng + 3 * nb + 10 * nb + 6 * nv;
      break;
    case 1:  // crb_factor(): multi_load cinert, cdof; stores qM (full symmetric), crb, qLD (full, zeros above the diagonal)
      rd = 10 * nb + 6 * nv;
      wr = nv * nv + 10 * nb + nv * nv;
      if (scratch) wr = 2 * nv * nv;
      break;
    case 2: case 7: case 8: {  // collision() + make_constraint() (8: small models, contact rows straight to the leaf)
      if ((kernel == 7) != general_con || (kernel == 8) != (m.con_direct != 0)) return -1;
      if (ncon > 0) { rd += 12 * ng; if (m.ncvxpair > 0) rd += 13 * ncon; }
      if (nefc > 0) {
        rd += nv + 3 * nb + 6 * nv + (general_con ? nq : (int64_t)m.nl);   // qvel, subtree_com, cdof, qpos (plain: one entry per limit row)
        if (general_con) rd += 32 * (int64_t)m.neqtab;                       // body frames of the equality constraints (xpos, xmat, xquat of two bodies)
        wr += nefc * nv + 3 * nefc;                                          // efc_J, efc_D, efc_aref, efc_frictionloss
      }
      wr += (13 + 15) * ncon;                                                // dist / pos / frame + the five model-constant contact leaves
      if (scratch) {
        wr = ncon + (m.ncvxpair > 0 ? 12 * ncon : 0);                        // contact_dist (+ pos / frame where the convex kernel hands them over)
        if (nefc > 0) wr += (kernel == 8) ? (int64_t)m.nl * (nv + 2) : nefc * (nv + 2);  // efc_J, efc_D, efc_aref (small models: limit rows + the active contacts' rows, the latter not counted)
      }
      *read_bytes = rd * R + (general_con ? 4 * (int64_t)m.neq : 0);
      *write_bytes = wr * R + (scratch ? 0 : 44 * ncon);                     // contact_dim i32; geom1, geom2, geom[2], efc_address i64
      return 0;
    }
    case 3: case 5:  // velocity<FLUID>() + actuation<FLUID>()
      if ((kernel == 5) != opt_vel) return -1;
      rd = nq + nv + na + 6 * nv + 10 * nb + 3 * nb + 3 * nb + nu + 6 * nb + nv;   // qpos qvel act cdof cinert subtree_com xipos ctrl xfrc_applied qfrc_applied
      wr = nu * nv + 2 * nu + 6 * nb + 6 * nv + 2 * nv + nu + na + 2 * nv;        // actuator_moment / length / velocity, cvel, cdof_dot, passive, bias, force, act_dot, actuator, smooth
      if (kernel == 5) {
        if (m.has_fluid) rd += 9 * nb;                                             // ximat
        if (m.has_gravcomp) wr += nv;
        wr += (int64_t)m.ntendon * (2 + nv);                                       // ten_length, ten_velocity, ten_J
      }
      if (scratch) wr = nv + na;                                                   // qfrc_smooth, act_dot
      break;
    case 4: case 6:  // load_factor_and_accelerate(), load_solver_inputs(), solve(), integrator
      if ((kernel == 6) != general_sol) return -1;
      rd = nv /* qfrc_smooth */ + tri /* packed factor */;
      wr = nv /* qacc_smooth */ + nv /* qacc */;
      if (do_step || nefc > 0) rd += nq + nv + na + na;                            // qpos, qvel, act, act_dot
      if (nefc > 0) {
        rd += nv /* warm start */ + 2 * nefc /* efc_D, efc_aref */ + nsingle /* one entry per single-column row */ + ndense * nv /* dense rows */
              + nv * nv /* qM: once per solve (re-read per iteration from L2 unless staged in LDS) */;
        wr += nv /* qacc_warmstart */ + nv /* qfrc_constraint */ + nefc /* efc_force */;
      }
      if (do_step) { rd += 1; wr += nq + nv + na + 1; if (!(m.disableflags & DSBL_EULERDAMP) && m.integrator == INT_EULER) rd += tri; }
      if (scratch) { wr -= (nefc > 0 ? nefc : 0); rd += 2 * nv + na; wr += 2 * nv + na; }  // no efc_force; the running sums of the tableau
      if (m.integrator == INT_RK4 && do_step == 1) { rd += nv + na; wr += 3 * nv + 2 * na; }  // stage 0 starts the sums and keeps qvel0 / act0
      break;
    case 10:  // mjh_convex_kernel: the two geom frames of every convex pair in, its (up to four) contacts out
      if (m.ncvxpair == 0) return -1;
      rd = 24 * (int64_t)m.ncvxpair;
      wr = 13 * 4 * (int64_t)m.ncvxpair;
      break;
    case 11:  // mjh_sensor_kernel: site frames of the sensors, geom frames for the rays, cvel / subtree_com of the IMU bodies, joint state
      if (m.nsensor == 0) return -1;
      rd = 12 * (int64_t)m.nsensor + 12 * ng + 9 * (int64_t)m.nsensor + (int64_t)m.nsensordata;
      wr = (int64_t)m.nsensordata;
      break;
    default:
      return -1;
  }
  *read_bytes = rd * R;
  *write_bytes = wr * R;
  return 0;
}
