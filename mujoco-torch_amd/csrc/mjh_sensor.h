// mjh_sensor.h -- sensors on the step path (reference mujoco_torch/_src/sensor.py:56-440, ray.py:28-373).
//
// One wavefront per environment, one lane per sensordata slot: velocimeter / gyro / accelerometer / joint position and
// velocity read the frames and velocities the KIN and VEL phases streamed out (L2-resident), a rangefinder lane walks
// its list of candidate geoms (the site's own body excluded, reference ray.precompute_ray_data) and keeps the nearest
// hit.  Slots of sensor types the reference leaves untouched (touch) keep the caller's value.  Launched after the VEL
// phase of a full forward pass (RK4: stage 0 only -- the returned Data carries the sensors of its own forward pass).
//
// Ray intersections run in double whatever the Data dtype: the reference keeps its ray tables' geom sizes in float64
// (ray.py:317) and torch promotes; with float32 Data its own call raises, so float32 + rangefinder is this build's
// choice (transform in the Data dtype, intersect in double, round the distance).
#pragma once
#include "mjh_kernels.h"

#define M (kargs<REAL>().M)
#define in (kargs<REAL>().in)
#define out (kargs<REAL>().cur)
#define KA (kargs<REAL>())

__device__ __forceinline__ double ray_safe_div(double num, double den) { return num / (den + (den == 0 ? (double)(float)mjMINVAL : 0.0)); }
__device__ __forceinline__ void ray_quad(double a, double b, double c, double& x0, double& x1) {  // ray.py:28-40
  const double det = b * b - a * c, det2 = sqrt(det);
  const double r0 = ray_safe_div(-b - det2, a), r1 = ray_safe_div(-b + det2, a);
  const double inf = __builtin_inf();
  x0 = ((det < mjMINVAL) || (r0 < 0)) ? inf : r0;
  x1 = ((det < mjMINVAL) || (r1 < 0)) ? inf : r1;
}
__device__ __forceinline__ double ray_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double ray_geom(int type, const double* size, const double* pnt, const double* vec) {
  const double inf = __builtin_inf();
  if (type == 0) {  // plane :43-57
    const double x = -ray_safe_div(pnt[2], vec[2]);
    bool valid = (vec[2] <= -mjMINVAL) && (x >= 0);
    for (int i = 0; i < 2; i++) { const double p = pnt[i] + x * vec[i]; valid = valid && ((size[i] <= 0) || (fabs(p) <= size[i])); }
    return valid ? x : inf;
  }
  if (type == 2) {  // sphere :60-69
    double x0, x1;
    ray_quad(ray_dot3(vec, vec), ray_dot3(vec, pnt), ray_dot3(pnt, pnt) - size[0] * size[0], x0, x1);
    return isinf(x0) ? x1 : x0;
  }
  if (type == 3 || type == 5) {  // capsule :72-106, cylinder :235-268: the round side first
    const double a = vec[0] * vec[0] + vec[1] * vec[1], b = vec[0] * pnt[0] + vec[1] * pnt[1], c = (pnt[0] * pnt[0] + pnt[1] * pnt[1]) - size[0] * size[0];
    double x0, x1;
    ray_quad(a, b, c, x0, x1);
    double x = isinf(x0) ? x1 : x0;
    x = (fabs(pnt[2] + x * vec[2]) <= size[1]) ? x : inf;
    for (int cap = 0; cap < 2; cap++) {
      if (type == 3) {  // spherical caps
        const double dif[3] = {pnt[0], pnt[1], cap == 0 ? pnt[2] - size[1] : pnt[2] + size[1]};
        ray_quad(ray_dot3(vec, vec), ray_dot3(vec, dif), ray_dot3(dif, dif) - size[0] * size[0], x0, x1);
        if (cap == 0) {
          if ((pnt[2] + x0 * vec[2] >= size[1]) && (x0 < x)) x = x0;
          if ((pnt[2] + x1 * vec[2] >= size[1]) && (x1 < x)) x = x1;
        } else {
          if ((pnt[2] + x0 * vec[2] <= -size[1]) && (x0 < x)) x = x0;
          if ((pnt[2] + x1 * vec[2] <= -size[1]) && (x1 < x)) x = x1;
        }
      } else {  // flat caps
        const double t = ray_safe_div((cap == 0 ? size[1] : -size[1]) - pnt[2], vec[2]);
        const double p0 = pnt[0] + t * vec[0], p1 = pnt[1] + t * vec[1];
        if ((t >= 0) && (p0 * p0 + p1 * p1 <= size[0] * size[0]) && (t < x)) x = t;
      }
    }
    return x;
  }
  if (type == 4) {  // ellipsoid :109-129
    double s[3], sv[3], sp[3];
    for (int i = 0; i < 3; i++) { s[i] = ray_safe_div(1, size[i] * size[i]); sv[i] = s[i] * vec[i]; sp[i] = s[i] * pnt[i]; }
    double x0, x1;
    ray_quad(ray_dot3(sv, vec), ray_dot3(sv, pnt), ray_dot3(sp, pnt) - 1, x0, x1);
    return isinf(x0) ? x1 : x0;
  }
  if (type == 6) {  // box :132-161
    double best = inf;
    for (int f = 0; f < 6; f++) {
      const int ax = f % 3, i0 = ax == 0 ? 1 : 0, i1 = ax == 2 ? 1 : 2;
      const double x = f < 3 ? ray_safe_div(size[ax] - pnt[ax], vec[ax]) : -ray_safe_div(size[ax] + pnt[ax], vec[ax]);
      const double p0 = pnt[i0] + x * vec[i0], p1 = pnt[i1] + x * vec[i1];
      const bool valid = (fabs(p0) <= size[i0]) && (fabs(p1) <= size[i1]) && (x >= 0);
      if (valid && x < best) best = x;
    }
    return best;
  }
  return inf;
}

// one ray test: rangefinder M.rf_sensor[q] against geom M.rf_geom[q].  The ray starts at the site and runs along its z axis
// (sensor.py:94-108); it is moved into the geom frame in the Data dtype and intersected in double (ray.py:28-290).
template <typename REAL>
__device__ __forceinline__ double rf_task(int64_t e, int q) {
  const int s = M.rf_sensor[q], g = M.rf_geom[q], obj = M.sns_objid[s];
  const REAL* rot = out.site_xmat + (e * M.nsite + obj) * 9;
  const REAL* posp = out.site_xpos + (e * M.nsite + obj) * 3;
  const REAL pos[3] = {posp[0], posp[1], posp[2]};
  const REAL vec[3] = {rot[2], rot[5], rot[8]};
  const REAL *gm = out.geom_xmat + (e * M.ngeom + g) * 9, *gp = out.geom_xpos + (e * M.ngeom + g) * 3;
  const REAL d3[3] = {pos[0] - gp[0], pos[1] - gp[1], pos[2] - gp[2]};
  double dp[3], dv[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    dp[i] = (double)(gm[i] * d3[0] + gm[3 + i] * d3[1] + gm[6 + i] * d3[2]);
    dv[i] = (double)(gm[i] * vec[0] + gm[3 + i] * vec[1] + gm[6 + i] * vec[2]);
  }
  const double size[3] = {(double)M.geom_size[3 * g], (double)M.geom_size[3 * g + 1], (double)M.geom_size[3 * g + 2]};
  return ray_geom(M.geom_type[g], size, dp, dv);
}

template <typename REAL>
__device__ __forceinline__ REAL sensor_value(int64_t e, int s, int comp, const double* rf_x) {
  const int type = M.sns_type[s], obj = M.sns_objid[s], body = M.sns_bodyid[s], root = M.sns_rootid[s];
  if (type == 9) return out.qpos[e * M.nq + obj];  // jointpos: the normalised qpos of this pass
  if (type == 10) {                                 // jointvel: the (checked) velocity this pass ran on
    const bool from_in = !KA.state_from_cur;
    const REAL v = (from_in ? in.qvel : KA.cur.qvel)[e * M.nv + obj];
    return (from_in && KA.do_step && (!r_finite(v) || r_abs(v) > (REAL)mjMAXVAL)) ? (REAL)0 : v;
  }
  const REAL* rot = out.site_xmat + (e * M.nsite + obj) * 9;
  const REAL* posp = out.site_xpos + (e * M.nsite + obj) * 3;
  const REAL pos[3] = {posp[0], posp[1], posp[2]};
  REAL R[9];
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = rot[i];
  if (type == 7) {  // rangefinder: nearest of the per-(sensor, geom) ray tests the wave staged in LDS (ray.py:327-372: min over geoms)
    double best = __builtin_inf();
    for (int q = M.sns_rfadr[s]; q < M.sns_rfadr[s + 1]; q++) { const double x = rf_x[q]; if (x < best) best = x; }
    return isinf(best) ? (REAL)-1 : (REAL)best;
  }
  const REAL* cv = out.cvel + (e * M.nbody + body) * 6;
  const REAL* sc = out.subtree_com + (e * M.nbody + root) * 3;
  const REAL cvel[6] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5]};
  const REAL dif[3] = {pos[0] - sc[0], pos[1] - sc[1], pos[2] - sc[2]};
#define ROT_T(v, o) for (int i_ = 0; i_ < 3; i_++) (o)[i_] = R[i_] * (v)[0] + R[3 + i_] * (v)[1] + R[6 + i_] * (v)[2];
  if (type == 3) { REAL o[3]; ROT_T(cvel, o) return o[comp]; }  // gyro :246-251
  REAL c[3], v[3], lin[3];
  cross3(dif, cvel, c);
#pragma unroll
  for (int i = 0; i < 3; i++) v[i] = cvel[3 + i] - c[i];
  ROT_T(v, lin)
  if (type == 2) return lin[comp];  // velocimeter :235-245
  // accelerometer :379-399 with Data.cacc, which no stage of the reference ever writes (zeros from make_data)
  REAL ang[3], ca[3], av[3], acc[3], corr[3];
  const REAL zero[3] = {0, 0, 0};
  ROT_T(cvel, ang)
  cross3(dif, zero, ca);
#pragma unroll
  for (int i = 0; i < 3; i++) av[i] = (REAL)0 - ca[i];
  ROT_T(av, acc)
  cross3(ang, lin, corr);
#undef ROT_T
  return (acc[comp] + corr[comp]) + 0;  // + gravity term, zero for mujoco >= 3.3.7 (sensor.py:36-38)
}

template <typename REAL>
__global__ __launch_bounds__(MJH_WAVE) void mjh_sensor_kernel(KArgs<REAL> args) {
  // rangefinders dominate: their (sensor, geom) ray tests are spread over the lanes first (ant: 8 sensors x 13 geoms = two
  // trips instead of 13 dependent tests on 8 lanes), then every sensordata slot is produced by one lane
  extern __shared__ double rf_x[];
  const int nsd = M.nsensordata, nrf = M.nrfq;
  for (int64_t e = blockIdx.x; e < KA.B; e += gridDim.x) {
    for (int q = lane_id(); q < nrf; q += MJH_WAVE) rf_x[q] = rf_task<REAL>(e, q);
    wave_sync();
    for (int k = lane_id(); k < nsd; k += MJH_WAVE) {
      const int s = M.slot_sensor[k];
      REAL v;
      if (s < 0) {
        v = in.sensordata ? in.sensordata[e * nsd + k] : (REAL)0;  // slot keeps the caller's value
      } else {
        v = sensor_value<REAL>(e, s, k - M.sns_adr[s], rf_x);
        const REAL cutoff = M.sns_cutoff[s];
        const int dt = M.sns_datatype[s];
        if (cutoff > 0) {  // _apply_cutoff :41-53
          if (dt == 0) v = v < -cutoff ? -cutoff : (v > cutoff ? cutoff : v);
          else if (dt == 1) v = v < cutoff ? v : cutoff;
        }
      }
      out.sensordata[e * nsd + k] = v;
    }
    wave_sync();  // the next environment of this workgroup reuses rf_x
  }
}

#undef M
#undef in
#undef out
#undef KA
