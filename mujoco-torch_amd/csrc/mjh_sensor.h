// mjh_sensor.h -- sensors on the step path (reference mujoco_torch/_src/sensor.py:56-440, ray.py:28-373).
//
// One wavefront per environment, one lane per sensordata slot.  Every type the reference's three stage functions evaluate: the site sensors (accelerometer,
// velocimeter, gyro, force, torque, magnetometer, rangefinder), joint / ball-joint / tendon / actuator readings, the frame sensors (position, quaternion, axes,
// linear and angular velocity of a body / xbody / geom / site / camera frame, optionally relative to a reference frame), subtree sensors and the clock.  They read
// the leaves the KIN and VEL phases streamed out (L2-resident); a rangefinder's (ray, geom) tests are spread over the lanes first.  Force / torque / accelerometer /
// subtree momentum read Data leaves NO stage of the reference writes (smooth.rne_postconstraint / subtree_vel do not exist there): the caller's values, MJH_DATA_EXTRA_IN.
// Slots of sensor types the reference leaves untouched (touch, joint / tendon limit sensors, framelinacc / frameangacc, energies, ...) keep the caller's value.  Launched after the VEL
// phase of a full forward pass (RK4: stage 0 only -- the returned Data carries the sensors of its own forward pass).
//
// Ray intersections run in the Data dtype.  The reference keeps its ray tables' geom sizes in float64 (ray.py:317) and with float32 Data its own call raises on the
// mixed dtypes, so float32 + rangefinder has no reference result: it is this build's choice -- everything in float32, like every other float32 leaf (rounds 1 - 3
// intersected in double and rounded: the float64 square roots and divisions made the ant's kernel VALU-bound, profiles/r04/notes.md).
#pragma once
#include "mjh_kernels.h"

#define M (kargs<REAL>().M)
#define in (kargs<REAL>().in)
#define out (kargs<REAL>().cur)
#define KA (kargs<REAL>())

// RT: the type the intersections run in = the Data dtype (see the header comment)
template <typename RT> __device__ __forceinline__ RT ray_safe_div(RT num, RT den) { return num / (den + (den == 0 ? (RT)(float)mjMINVAL : (RT)0)); }
template <typename RT>
__device__ __forceinline__ void ray_quad(RT a, RT b, RT c, RT& x0, RT& x1) {  // ray.py:28-40
  const RT det = b * b - a * c, det2 = r_sqrt<RT>(det);
  const RT r0 = ray_safe_div<RT>(-b - det2, a), r1 = ray_safe_div<RT>(-b + det2, a);
  const RT inf = (RT)__builtin_inf();
  x0 = ((det < (RT)mjMINVAL) || (r0 < 0)) ? inf : r0;
  x1 = ((det < (RT)mjMINVAL) || (r1 < 0)) ? inf : r1;
}
template <typename RT> __device__ __forceinline__ RT ray_dot3(const RT* a, const RT* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename RT>
__device__ __forceinline__ RT ray_geom(int type, const RT* size, const RT* pnt, const RT* vec) {
  const RT inf = (RT)__builtin_inf();
  if (type == 0) {  // plane :43-57
    const RT x = -ray_safe_div<RT>(pnt[2], vec[2]);
    bool valid = (vec[2] <= -(RT)mjMINVAL) && (x >= 0);
    for (int i = 0; i < 2; i++) { const RT p = pnt[i] + x * vec[i]; valid = valid && ((size[i] <= 0) || (r_abs(p) <= size[i])); }
    return valid ? x : inf;
  }
  if (type == 2) {  // sphere :60-69
    RT x0, x1;
    ray_quad<RT>(ray_dot3(vec, vec), ray_dot3(vec, pnt), ray_dot3(pnt, pnt) - size[0] * size[0], x0, x1);
    return isinf(x0) ? x1 : x0;
  }
  if (type == 3 || type == 5) {  // capsule :72-106, cylinder :235-268: the round side first
    const RT a = vec[0] * vec[0] + vec[1] * vec[1], b = vec[0] * pnt[0] + vec[1] * pnt[1], c = (pnt[0] * pnt[0] + pnt[1] * pnt[1]) - size[0] * size[0];
    RT x0, x1;
    ray_quad<RT>(a, b, c, x0, x1);
    RT x = isinf(x0) ? x1 : x0;
    x = (r_abs(pnt[2] + x * vec[2]) <= size[1]) ? x : inf;
    for (int cap = 0; cap < 2; cap++) {
      if (type == 3) {  // spherical caps
        const RT dif[3] = {pnt[0], pnt[1], cap == 0 ? pnt[2] - size[1] : pnt[2] + size[1]};
        ray_quad<RT>(ray_dot3(vec, vec), ray_dot3(vec, dif), ray_dot3(dif, dif) - size[0] * size[0], x0, x1);
        if (cap == 0) {
          if ((pnt[2] + x0 * vec[2] >= size[1]) && (x0 < x)) x = x0;
          if ((pnt[2] + x1 * vec[2] >= size[1]) && (x1 < x)) x = x1;
        } else {
          if ((pnt[2] + x0 * vec[2] <= -size[1]) && (x0 < x)) x = x0;
          if ((pnt[2] + x1 * vec[2] <= -size[1]) && (x1 < x)) x = x1;
        }
      } else {  // flat caps
        const RT t = ray_safe_div<RT>((cap == 0 ? size[1] : -size[1]) - pnt[2], vec[2]);
        const RT p0 = pnt[0] + t * vec[0], p1 = pnt[1] + t * vec[1];
        if ((t >= 0) && (p0 * p0 + p1 * p1 <= size[0] * size[0]) && (t < x)) x = t;
      }
    }
    return x;
  }
  if (type == 4) {  // ellipsoid :109-129
    RT s[3], sv[3], sp[3];
    for (int i = 0; i < 3; i++) { s[i] = ray_safe_div<RT>((RT)1, size[i] * size[i]); sv[i] = s[i] * vec[i]; sp[i] = s[i] * pnt[i]; }
    RT x0, x1;
    ray_quad<RT>(ray_dot3(sv, vec), ray_dot3(sv, pnt), ray_dot3(sp, pnt) - 1, x0, x1);
    return isinf(x0) ? x1 : x0;
  }
  if (type == 6) {  // box :132-161
    RT best = inf;
    for (int f = 0; f < 6; f++) {
      const int ax = f % 3, i0 = ax == 0 ? 1 : 0, i1 = ax == 2 ? 1 : 2;
      const RT x = f < 3 ? ray_safe_div<RT>(size[ax] - pnt[ax], vec[ax]) : -ray_safe_div<RT>(size[ax] + pnt[ax], vec[ax]);
      const RT p0 = pnt[i0] + x * vec[i0], p1 = pnt[i1] + x * vec[i1];
      const bool valid = (r_abs(p0) <= size[i0]) && (r_abs(p1) <= size[i1]) && (x >= 0);
      if (valid && x < best) best = x;
    }
    return best;
  }
  return inf;
}

// one ray test: rangefinder M.rf_sensor[q] against geom M.rf_geom[q].  The ray starts at the site and runs along its z axis
// (sensor.py:94-108); it is moved into the geom frame and intersected (ray.py:28-290) in the Data dtype.
template <typename REAL>
__device__ __forceinline__ double rf_task(int64_t e, int q) {
  const int g = M.rf_geom[q], obj = M.rf_site[q];  // (per-task tables: the site and the geom's type / size are read by task index, not through sensor -> site / geom -> size chains)
  const REAL* rot = out.site_xmat + (e * M.nsite + obj) * 9;
  const REAL* posp = out.site_xpos + (e * M.nsite + obj) * 3;
  const REAL pos[3] = {posp[0], posp[1], posp[2]};
  const REAL vec[3] = {rot[2], rot[5], rot[8]};
  const REAL *gm = out.geom_xmat + (e * M.ngeom + g) * 9, *gp = out.geom_xpos + (e * M.ngeom + g) * 3;
  const REAL d3[3] = {pos[0] - gp[0], pos[1] - gp[1], pos[2] - gp[2]};
  REAL dp[3], dv[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    dp[i] = gm[i] * d3[0] + gm[3 + i] * d3[1] + gm[6 + i] * d3[2];
    dv[i] = gm[i] * vec[0] + gm[3 + i] * vec[1] + gm[6 + i] * vec[2];
  }
  const REAL size[3] = {M.rf_gsize[3 * q], M.rf_gsize[3 * q + 1], M.rf_gsize[3 * q + 2]};
  return (double)ray_geom<REAL>(M.rf_gtype[q], size, dp, dv);
}

// frame of an object of a frame sensor (sensor.py:62-74): position and orientation by mjtObj (0 unknown: origin / identity, 1 body: inertial frame, 2 xbody, 5 geom, 6 site, 7 camera)
template <typename REAL>
__device__ __forceinline__ void sns_frame(int64_t e, int objtype, int id, REAL* pos, REAL* mat) {
  const REAL *p = nullptr, *m = nullptr;
  switch (objtype) {
    case 1: p = out.xipos + (e * M.nbody + id) * 3; m = out.ximat + (e * M.nbody + id) * 9; break;
    case 2: p = out.xpos + (e * M.nbody + id) * 3; m = out.xmat + (e * M.nbody + id) * 9; break;
    case 5: p = out.geom_xpos + (e * M.ngeom + id) * 3; m = out.geom_xmat + (e * M.ngeom + id) * 9; break;
    case 6: p = out.site_xpos + (e * M.nsite + id) * 3; m = out.site_xmat + (e * M.nsite + id) * 9; break;
    case 7: p = out.cam_xpos + (e * M.ncam + id) * 3; m = out.cam_xmat + (e * M.ncam + id) * 9; break;
    default: break;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) pos[i] = p ? p[i] : (REAL)0;
#pragma unroll
  for (int i = 0; i < 9; i++) mat[i] = m ? m[i] : (REAL)((i & 3) == 0);
}
// orientation of such an object as a quaternion (sensor.py:164-181)
template <typename REAL>
__device__ __forceinline__ void sns_quat(int64_t e, int objtype, int id, int body, REAL* q) {
  const REAL* xq = out.xquat + (e * M.nbody + (objtype == 1 || objtype == 2 ? id : body)) * 4;
  const REAL w[4] = {xq[0], xq[1], xq[2], xq[3]};
  const REAL* lq = objtype == 1 ? M.body_iquat + 4 * id : (objtype == 5 ? M.geom_quat + 4 * id : (objtype == 6 ? M.site_quat + 4 * id : (objtype == 7 ? M.cam_quat + 4 * id : nullptr)));
  if (objtype == 2) { q[0] = w[0]; q[1] = w[1]; q[2] = w[2]; q[3] = w[3]; }
  else if (lq) { const REAL l4[4] = {lq[0], lq[1], lq[2], lq[3]}; quat_mul(w, l4, q); }
  else { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; }
}

// FULL = false: the instantiation for models whose sensors are the ant's kinds only (accelerometer, velocimeter, gyro, rangefinder, jointpos, jointvel) -- it carries
// none of the other types' code (measured on the ant, B = 16384: 39.5 us against 44.5 us for the full kernel, which spills)
template <typename REAL, bool FULL>
__device__ __forceinline__ REAL sensor_value(int64_t e, int s, int comp, const double* rf_x) {
  const int type = M.sns_type[s], obj = M.sns_objid[s], body = M.sns_bodyid[s], root = M.sns_rootid[s];
  const bool from_in = !KA.state_from_cur;
  if (type == 9) return out.qpos[e * M.nq + obj];  // jointpos: the normalised qpos of this pass
  if (type == 10 || (FULL && type == 19)) {                   // jointvel, ballangvel: the (checked) velocity this pass ran on
    const REAL v = (from_in ? in.qvel : KA.cur.qvel)[e * M.nv + obj + (type == 19 ? comp : 0)];
    return (from_in && KA.do_step && (!r_finite(v) || r_abs(v) > (REAL)mjMAXVAL)) ? (REAL)0 : v;
  }
  if constexpr (FULL) {
  if (type == 11) return out.ten_length[e * M.ntendon + obj];        // tendonpos :111-112
  if (type == 12) return out.ten_velocity[e * M.ntendon + obj];      // tendonvel :254-255
  if (type == 13) return out.actuator_length[e * M.nu + obj];        // actuatorpos :113-114
  if (type == 14) return out.actuator_velocity[e * M.nu + obj];      // actuatorvel :256-257
  if (type == 15) return out.actuator_force[e * M.nu + obj];         // actuatorfrc :417-418
  if (type == 16) return out.qfrc_actuator[e * M.nv + obj];          // jointactuatorfrc :419-420
  if (type == 17) {  // tendonactuatorfrc :421-423: force_mask @ actuator_force
    REAL acc = 0;
    for (int i = 0; i < M.nu; i++) acc += (REAL)(M.act_trntype[i] == 3 && M.act_trnid[i] == obj) * out.actuator_force[e * M.nu + i];
    return acc;
  }
  if (type == 18) {  // ballquat :115-118
    const REAL* qp = out.qpos + e * M.nq + obj;
    REAL q[4] = {qp[0], qp[1], qp[2], qp[3]};
    normalize_n<REAL, 4>(q);
    return q[comp];
  }
  if (type == 35) return out.subtree_com[(e * M.nbody + obj) * 3 + comp];                                     // subtreecom :211-213
  if (type == 36) return in.subtree_linvel ? in.subtree_linvel[(e * M.nbody + obj) * 3 + comp] : (REAL)0;     // :261-263: a leaf no stage writes
  if (type == 37) return in.subtree_angmom ? in.subtree_angmom[(e * M.nbody + obj) * 3 + comp] : (REAL)0;     // :264-266
  if (type == 45) return in.time ? in.time[e] : (REAL)0;                                                      // clock :214-215
#define ROT_T(R_, v, o) for (int i_ = 0; i_ < 3; i_++) (o)[i_] = (R_)[i_] * (v)[0] + (R_)[3 + i_] * (v)[1] + (R_)[6 + i_] * (v)[2];
  if (type >= 26 && type <= 32) {  // frame sensors: the object seen from the reference object or, without one, from the world
    const int ot = M.sns_objtype[s], rt = M.sns_reftype[s], rid = M.sns_refid[s], rbody = M.sns_refbodyid[s], rroot = M.sns_refrootid[s];
    REAL xpos[3], xmat[9], rpos[3], rmat[9];
    sns_frame<REAL>(e, ot, obj, xpos, xmat);
    sns_frame<REAL>(e, rid >= 0 ? rt : 0, rid >= 0 ? rid : 0, rpos, rmat);
    if (type == 26) {  // framepos :119-138
      if (rid < 0) return xpos[comp];
      const REAL d3[3] = {xpos[0] - rpos[0], xpos[1] - rpos[1], xpos[2] - rpos[2]};
      REAL o[3];
      ROT_T(rmat, d3, o)
      return o[comp];
    }
    if (type >= 28 && type <= 30) {  // frame{x,y,z}axis :139-160
      const int k = type - 28;
      const REAL axis[3] = {xmat[k], xmat[3 + k], xmat[6 + k]};
      REAL o[3];
      if (rid < 0) return axis[comp];
      ROT_T(rmat, axis, o)
      return o[comp];
    }
    if (type == 27) {  // framequat :161-199
      REAL q[4], r[4], o[4];
      sns_quat<REAL>(e, ot, obj, body, q);
      if (rid < 0) return q[comp];
      sns_quat<REAL>(e, rt, rid, rbody, r);
      const REAL ri[4] = {r[0] * (REAL)1, r[1] * (REAL)-1, r[2] * (REAL)-1, r[3] * (REAL)-1};  // quat_inv :264-273
      quat_mul(ri, q, o);
      return o[comp];
    }
    // framelinvel / frameangvel :267-328
    const REAL *cvp = out.cvel + (e * M.nbody + body) * 6, *cvrp = out.cvel + (e * M.nbody + rbody) * 6;
    const REAL cv[6] = {cvp[0], cvp[1], cvp[2], cvp[3], cvp[4], cvp[5]}, cvr[6] = {cvrp[0], cvrp[1], cvrp[2], cvrp[3], cvrp[4], cvrp[5]};
    if (type == 32) {
      if (rid < 0) return cv[comp];
      const REAL rel[3] = {cv[0] - cvr[0], cv[1] - cvr[1], cv[2] - cvr[2]};
      REAL o[3];
      ROT_T(rmat, rel, o)
      return o[comp];
    }
    const REAL *sc = out.subtree_com + (e * M.nbody + root) * 3, *scr = out.subtree_com + (e * M.nbody + rroot) * 3;
    const REAL off[3] = {xpos[0] - sc[0], xpos[1] - sc[1], xpos[2] - sc[2]};
    REAL c[3], xl[3];
    cross3(off, cv, c);
#pragma unroll
    for (int i = 0; i < 3; i++) xl[i] = cv[3 + i] - c[i];
    if (rid < 0) return xl[comp];
    const REAL offr[3] = {rpos[0] - scr[0], rpos[1] - scr[1], rpos[2] - scr[2]}, rvec[3] = {xpos[0] - rpos[0], xpos[1] - rpos[1], xpos[2] - rpos[2]};
    REAL cr[3], xlr[3], cw[3], rel[3], o[3];
    cross3(offr, cvr, cr);
#pragma unroll
    for (int i = 0; i < 3; i++) xlr[i] = cvr[3 + i] - cr[i];
    cross3(rvec, cvr, cw);
#pragma unroll
    for (int i = 0; i < 3; i++) rel[i] = (xl[i] - xlr[i]) + cw[i];
    ROT_T(rmat, rel, o)
    return o[comp];
  }
  }  // FULL
#ifndef ROT_T
#define ROT_T(R_, v, o) for (int i_ = 0; i_ < 3; i_++) (o)[i_] = (R_)[i_] * (v)[0] + (R_)[3 + i_] * (v)[1] + (R_)[6 + i_] * (v)[2];
#endif
  const REAL* rot = out.site_xmat + (e * M.nsite + obj) * 9;
  const REAL* posp = out.site_xpos + (e * M.nsite + obj) * 3;
  const REAL pos[3] = {posp[0], posp[1], posp[2]};
  REAL R[9];
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = rot[i];
  if (FULL && type == 6) {  // magnetometer :92-94
    REAL o[3];
    ROT_T(R, M.magnetic, o)
    return o[comp];
  }
  if (type == 7) {  // rangefinder: nearest of the per-(sensor, geom) ray tests the wave staged in LDS (ray.py:327-372: min over geoms)
    double best = __builtin_inf();
    for (int q = M.sns_rfadr[s]; q < M.sns_rfadr[s + 1]; q++) { const double x = rf_x[q]; if (x < best) best = x; }
    return isinf(best) ? (REAL)-1 : (REAL)best;
  }
  const REAL* sc = out.subtree_com + (e * M.nbody + root) * 3;
  const REAL dif[3] = {pos[0] - sc[0], pos[1] - sc[1], pos[2] - sc[2]};
  if (FULL && (type == 4 || type == 5)) {  // force :399-406, torque :407-416: from Data.cfrc_int, a leaf no stage of the reference writes (the caller's: zeros from make_data)
    REAL cf[6], o[3];
#pragma unroll
    for (int i = 0; i < 6; i++) cf[i] = in.cfrc_int ? in.cfrc_int[(e * M.nbody + body) * 6 + i] : (REAL)0;
    if (type == 4) { ROT_T(R, cf + 3, o) return o[comp]; }
    REAL c[3], v[3];
    cross3(dif, cf + 3, c);
#pragma unroll
    for (int i = 0; i < 3; i++) v[i] = cf[i] - c[i];
    ROT_T(R, v, o)
    return o[comp];
  }
  const REAL* cv = out.cvel + (e * M.nbody + body) * 6;
  const REAL cvel[6] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5]};
  if (type == 3) { REAL o[3]; ROT_T(R, cvel, o) return o[comp]; }  // gyro :246-251
  REAL c[3], v[3], lin[3];
  cross3(dif, cvel, c);
#pragma unroll
  for (int i = 0; i < 3; i++) v[i] = cvel[3 + i] - c[i];
  ROT_T(R, v, lin)
  if (type == 2) return lin[comp];  // velocimeter :235-245
  // accelerometer :379-399 with Data.cacc, which no stage of the reference ever writes (the caller's leaf: zeros from make_data)
  REAL ang[3], ca[3], av[3], acc[3], corr[3], cacc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) cacc[i] = in.cacc ? in.cacc[(e * M.nbody + body) * 6 + i] : (REAL)0;
  ROT_T(R, cvel, ang)
  cross3(dif, cacc, ca);
#pragma unroll
  for (int i = 0; i < 3; i++) av[i] = cacc[3 + i] - ca[i];
  ROT_T(R, av, acc)
  cross3(ang, lin, corr);
#undef ROT_T
  return (acc[comp] + corr[comp]) + 0;  // + gravity term, zero for mujoco >= 3.3.7 (sensor.py:36-38)
}

template <typename REAL, int FULL>
#ifndef MJH_SENSOR32_WAVES
#define MJH_SENSOR32_WAVES 8  /* float32 lean instantiation: 70 VGPRs are seven waves per SIMD = 28 one-environment workgroups per CU -- three rounds of waves for the ant's 64 environments per CU (28 + 28 + 8); 64 VGPRs are two */
#endif
__global__ __launch_bounds__(MJH_WAVE, (sizeof(REAL) == 4 && FULL == 0) ? MJH_SENSOR32_WAVES : 1) void mjh_sensor_kernel(KArgs<REAL> args) {
  // rangefinders dominate: their (sensor, geom) ray tests are spread over the lanes first (ant: 8 sensors x 13 geoms = two
  // trips instead of 13 dependent tests on 8 lanes), then every sensordata slot is produced by one lane
  extern __shared__ double rf_x[];
  const int nsd = M.nsensordata, nrf = M.nrfq;
  // KArgs::sns_epw environments per wavefront (round 5): the ant's 20 slots left 44 lanes idle in the slot phase and 16384 one-environment workgroups cost 8 us of
  // launch alone.  Ray tasks and slots of the wave's environments are laid end to end over the lanes: task t = (environment t / nrf, ray test t % nrf), slot t likewise.
  {  // no grid-stride loop (the host launches per 2^20 workgroups)
    const int epw = KA.sns_epw;
    const int64_t e0 = KA.env_begin + (int64_t)blockIdx.x * epw;
    const int here = (int)((KA.env_begin + KA.env_count - e0) < epw ? (KA.env_begin + KA.env_count - e0) : epw);  // environments of this workgroup
    const float inv_nrf = 1.0f / (float)(nrf > 0 ? nrf : 1), inv_nsd = 1.0f / (float)(nsd > 0 ? nsd : 1);
#ifdef MJH_SENSOR_ABLATE
    if (!(KA.flags & 0x100))
#endif
    for (int t = lane_id(); t < here * nrf; t += MJH_WAVE) {
      int k, q;
      split_index(t, nrf, inv_nrf, k, q);
      rf_x[k * (nrf + 1) + q] = rf_task<REAL>(e0 + k, q);
    }
    wave_sync();
#ifdef MJH_SENSOR_ABLATE
    if (!(KA.flags & 0x200))
#endif
    for (int t = lane_id(); t < here * nsd; t += MJH_WAVE) {
      int ke, k;
      split_index(t, nsd, inv_nsd, ke, k);
      const int64_t e = e0 + ke;
      const int s = M.slot_sensor[k];
      REAL v;
      if (s < 0) {
        v = in.sensordata ? in.sensordata[e * nsd + k] : (REAL)0;  // slot keeps the caller's value
      } else {
        v = sensor_value<REAL, FULL != 0>(e, s, k - M.sns_adr[s], rf_x + ke * (nrf + 1));
        const REAL cutoff = M.sns_cutoff[s];
        const int dt = M.sns_datatype[s];
        if (cutoff > 0) {  // _apply_cutoff :41-53
          if (dt == 0) v = v < -cutoff ? -cutoff : (v > cutoff ? cutoff : v);
          else if (dt == 1) v = v < cutoff ? v : cutoff;
        }
      }
      out.sensordata[e * nsd + k] = v;
    }
  }
}

#undef M
#undef in
#undef out
#undef KA
