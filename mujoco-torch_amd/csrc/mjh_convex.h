// mjh_convex.h -- convex narrow phase (box / mesh geoms): ONE WAVEFRONT PER (environment, convex geom pair).
//
// Restates reference mujoco_torch/_src/collision_convex.py (hard-selection branch): plane_convex :604-623,
// sphere_convex :626-699, capsule_convex :702-802, convex_convex :805-856 with _sat_hull_hull :464-601,
// _create_contact_manifold :395-449, _clip :330-392, _clip_edge_to_planes :265-327, _manifold_points :183-235.
//
// The reference vmaps each pair function over the statically filtered pairs of a (type, type, shape) group and runs
// every array op over all vertices / faces / edge pairs / clipped points at once.  Here a pair is one workgroup of one
// wave: the candidates of every selection (vertices, separating axes, faces, clipped points) are spread over the 64
// lanes, each torch.argmax / argmin becomes a per-lane running best plus a wave butterfly whose tie-break is the lower
// index (torch's documented first-occurrence rule), and the few sequential stages (a -> b -> c -> d of the manifold)
// stay sequential.  The pair grid (B x pairs) is what fills the chip: a config-5 batch of 8192 environments launches
// 24.6k independent waves.  Results go straight to the contact slots of the pair in the batch-major contact leaves;
// the constraint phase (PH_CON) picks them up from there.
//
// Per-lane arithmetic follows the operation order of oracle/mjoracle_impl.h (the pinned restatement) expression by
// expression, so selections that are not rounding-noise ties resolve identically.
#pragma once
#include "mjh_kernels.h"

#define M (kargs<REAL>().M)
#define out (kargs<REAL>().cur)
#define KA (kargs<REAL>())

template <typename REAL>
struct CvxView {
  int nvert, nface, nfv, nedge;
  const REAL *vert, *norm;
  const int *face, *edge;
};

// argmax (SGN = +1) / argmin (SGN = -1) over the wave; ties -> lowest index.  Every lane returns the winner.
// "Best" is the minimum of a total order on (value, index) pairs, so any reduction tree gives the same winner.  Four DPP steps (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror) leave each 16-lane row's best in all of its lanes; the four rows are then combined from v_readlane broadcasts.  No LDS-pipe
// round trip: the __shfl_xor butterfly this replaces was six dependent ds_bpermute pairs per call (~1.4 k cycles each, ten calls on the box-mesh pair's path).
template <int CTRL> __device__ __forceinline__ int dpp_move_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int SGN, typename REAL>
__device__ __forceinline__ void argbest_combine(REAL& v, int& idx, REAL ov, int oi) {
  const bool better = SGN > 0 ? (ov > v) : (ov < v);
  if (better || (ov == v && oi < idx)) { v = ov; idx = oi; }
}
template <int SGN, typename REAL>
__device__ __forceinline__ void wave_argbest(REAL& v, int& idx) {
#ifdef MJH_NO_DPP
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const REAL ov = __shfl_xor(v, o, MJH_WAVE);
    const int oi = __shfl_xor(idx, o, MJH_WAVE);
    argbest_combine<SGN>(v, idx, ov, oi);
  }
#else
  { const REAL ov = dpp_move<0xB1>(v); const int oi = dpp_move_int<0xB1>(idx); argbest_combine<SGN>(v, idx, ov, oi); }
  { const REAL ov = dpp_move<0x4E>(v); const int oi = dpp_move_int<0x4E>(idx); argbest_combine<SGN>(v, idx, ov, oi); }
  { const REAL ov = dpp_move<0x141>(v); const int oi = dpp_move_int<0x141>(idx); argbest_combine<SGN>(v, idx, ov, oi); }
  { const REAL ov = dpp_move<0x140>(v); const int oi = dpp_move_int<0x140>(idx); argbest_combine<SGN>(v, idx, ov, oi); }
  REAL bv = read_lane(v, 0);
  int bi = read_lane(idx, 0);
#pragma unroll
  for (int r = 1; r < 4; r++) argbest_combine<SGN>(bv, bi, read_lane(v, 16 * r), read_lane(idx, 16 * r));
  v = bv; idx = bi;
#endif
}
template <int SGN, typename REAL>
__device__ __forceinline__ void lane_best(REAL& bv, int& bi, REAL v, int i) {
  const bool better = SGN > 0 ? (v > bv) : (v < bv);
  if (bi < 0 || better || (v == bv && i < bi)) { bv = v; bi = i; }
}
#define CVX_NONE 0x7fffffff

#ifdef MJH_STAMPS
#define CSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
    if (kargs<REAL>().stamps && l == 0) kargs<REAL>().stamps[e * 128 + (slot)] += t_ - cstamp_prev; cstamp_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CSTAMP(slot) do {} while (0)
#endif
template <typename REAL>
struct CvxPair {
  unsigned long long cstamp_prev = 0;
  REAL* L;       // this wave's LDS scratch
  int64_t e;
  int p, l;
  static constexpr REAL kEps32 = (REAL)(float)1e-6;  // `1e-6 * (x == 0.0)` is a float32 product in torch

  __device__ __forceinline__ CvxPair(REAL* lds, int64_t env, int pair) : L(lds), e(env), p(pair), l(lane_id()) {}

  __device__ __forceinline__ CvxView<REAL> view(int geom) const {
    const int c = M.geom_convexid[geom];
    CvxView<REAL> r;
    r.nvert = M.convex_nvert[c]; r.nface = M.convex_nface[c]; r.nfv = M.convex_nfv[c]; r.nedge = M.convex_nedge[c];
    r.vert = M.convex_vert + 3 * M.convex_vertadr[c];
    r.norm = M.convex_facenormal + 3 * M.convex_normadr[c];
    r.face = M.convex_face + M.convex_faceadr[c];
    r.edge = M.convex_edge + 2 * M.convex_edgeadr[c];
    return r;
  }
  // vertex id k of face f with faces padded to K >= nfv by repeating the last id (F.pad replicate, :810-816)
  __device__ __forceinline__ static int fv(const CvxView<REAL>& c, int f, int k) { return c.face[f * c.nfv + (k < c.nfv ? k : c.nfv - 1)]; }
  __device__ __forceinline__ static void mat_t_vec(const REAL* R, const REAL* v, REAL* o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
  }
  __device__ __forceinline__ static void mat_vec(const REAL* R, const REAL* v, REAL* o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
  }
  __device__ __forceinline__ static void seg_point_plane(const REAL* a, const REAL* b, const REAL* p0, const REAL* n, REAL* o) {  // :39-63
    const REAL ba[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const REAL d = dot3(p0, n);
    const REAL denom = dot3(n, ba);
    REAL t = (d - dot3(n, a)) / (denom + (denom == 0 ? kEps32 : (REAL)0));
    t = t < 0 ? (REAL)0 : (t > 1 ? (REAL)1 : t);
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a[i] + t * ba[i];
  }
  __device__ __forceinline__ static void project_pt(const REAL* pt, const REAL* plane_pt, const REAL* n, REAL* o) {  // :238-241
    const REAL d[3] = {pt[0] - plane_pt[0], pt[1] - plane_pt[1], pt[2] - plane_pt[2]};
    const REAL dist = dot3(d, n);
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = pt[i] - dist * n[i];
  }
  __device__ __forceinline__ static void closest_segment_point(const REAL* a, const REAL* b, const REAL* pt, REAL* o) { Env<REAL>::closest_segment_point(a, b, pt, o); }

  // _manifold_points :183-235.  poly: n x 3 and msk: n (1 = candidate) in LDS; result uniform over the wave.
  __device__ __forceinline__ void manifold_points(const REAL* poly, const REAL* msk, int n, const REAL* norm, int* idx) const {
    REAL bv = 0; int bi = -1;
    for (int i = l; i < n; i += MJH_WAVE) lane_best<+1>(bv, bi, msk[i] != 0 ? (REAL)0 : (REAL)-1e6, i);
    if (bi < 0) { bv = (REAL)-1e30; bi = CVX_NONE; }
    wave_argbest<+1>(bv, bi);
    const int ai = bi;
    const REAL a[3] = {poly[3 * ai], poly[3 * ai + 1], poly[3 * ai + 2]};
    bv = 0; bi = -1;
    for (int i = l; i < n; i += MJH_WAVE) {
      const REAL e0 = a[0] - poly[3 * i], e1 = a[1] - poly[3 * i + 1], e2 = a[2] - poly[3 * i + 2];
      lane_best<+1>(bv, bi, (e0 * e0 + e1 * e1 + e2 * e2) + (msk[i] != 0 ? (REAL)0 : (REAL)-1e6), i);
    }
    if (bi < 0) { bv = (REAL)-1e30; bi = CVX_NONE; }
    wave_argbest<+1>(bv, bi);
    const int bidx = bi;
    const REAL b[3] = {poly[3 * bidx], poly[3 * bidx + 1], poly[3 * bidx + 2]};
    const REAL amb[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    REAL ab[3];
    cross3(norm, amb, ab);
    bv = 0; bi = -1;
    for (int i = l; i < n; i += MJH_WAVE) {
      const REAL ap[3] = {a[0] - poly[3 * i], a[1] - poly[3 * i + 1], a[2] - poly[3 * i + 2]};
      lane_best<+1>(bv, bi, r_abs(dot3(ap, ab)) + (msk[i] != 0 ? (REAL)0 : (REAL)-1e6), i);
    }
    if (bi < 0) { bv = (REAL)-1e30; bi = CVX_NONE; }
    wave_argbest<+1>(bv, bi);
    const int ci = bi;
    const REAL c[3] = {poly[3 * ci], poly[3 * ci + 1], poly[3 * ci + 2]};
    const REAL amc[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]}, bmc[3] = {b[0] - c[0], b[1] - c[1], b[2] - c[2]};
    REAL ac[3], bc[3];
    cross3(norm, amc, ac);
    cross3(norm, bmc, bc);
    bv = 0; bi = -1;
    for (int i = l; i < n; i += MJH_WAVE) {
      const REAL ap[3] = {a[0] - poly[3 * i], a[1] - poly[3 * i + 1], a[2] - poly[3 * i + 2]};
      const REAL bp[3] = {b[0] - poly[3 * i], b[1] - poly[3 * i + 1], b[2] - poly[3 * i + 2]};
      const REAL dm = msk[i] != 0 ? (REAL)0 : (REAL)-1e6;
      lane_best<+1>(bv, bi, r_abs(dot3(bp, bc)) + dm, i);       // torch.cat([dist_bp, dist_ap]).argmax() % n
      lane_best<+1>(bv, bi, r_abs(dot3(ap, ac)) + dm, n + i);
    }
    if (bi < 0) { bv = (REAL)-1e30; bi = CVX_NONE; }
    wave_argbest<+1>(bv, bi);
    idx[0] = ai; idx[1] = bidx; idx[2] = ci; idx[3] = bi % n;
  }

  // _clip_edge_to_planes :265-327 against K planes (point j: pts[(j-1) mod K], normal: nrm[j]); returns the mask
  __device__ __forceinline__ static bool clip_edge_to_planes(const REAL* p0, const REAL* p1, const REAL* pts, const REAL* nrm, int K, REAL* o0, REAL* o1) {
    const REAL p10[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, p01[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    bool any_both = false;
    REAL b0 = 0, b1 = 0, n0[3] = {p0[0], p0[1], p0[2]}, n1[3] = {p1[0], p1[1], p1[2]};
    for (int j = 0; j < K; j++) {
      const int jm = j == 0 ? K - 1 : j - 1;
      const REAL pp[3] = {pts[3 * jm], pts[3 * jm + 1], pts[3 * jm + 2]}, pn[3] = {nrm[3 * j], nrm[3 * j + 1], nrm[3 * j + 2]};
      const REAL e0[3] = {p0[0] - pp[0], p0[1] - pp[1], p0[2] - pp[2]}, e1[3] = {p1[0] - pp[0], p1[1] - pp[1], p1[2] - pp[2]};
      const bool f0 = dot3(e0, pn) > (REAL)1e-6, f1 = dot3(e1, pn) > (REAL)1e-6;
      any_both = any_both || (f0 && f1);
      REAL cand[3];
      seg_point_plane(p0, p1, pp, pn, cand);
      REAL s0[3], s1[3], q0[3], q1[3];
#pragma unroll
      for (int i = 0; i < 3; i++) { s0[i] = f0 ? cand[i] : p0[i]; s1[i] = f1 ? cand[i] : p1[i]; q0[i] = s0[i] - p0[i]; q1[i] = s1[i] - p1[i]; }
      const REAL d0 = dot3(q0, p10), d1 = dot3(q1, p01);
      if (j == 0 || d0 > b0) { b0 = d0; n0[0] = s0[0]; n0[1] = s0[1]; n0[2] = s0[2]; }
      if (j == 0 || d1 > b1) { b1 = d1; n1[0] = s1[0]; n1[1] = s1[1]; n1[2] = s1[2]; }
    }
    bool mask = !any_both;
#pragma unroll
    for (int i = 0; i < 3; i++) { o0[i] = mask ? n0[i] : p0[i]; o1[i] = mask ? n1[i] : p1[i]; }
    const REAL dd[3] = {o0[0] - o1[0], o0[1] - o1[1], o0[2] - o1[2]};
    if (dot3(p01, dd) < 0) mask = false;
    return mask;
  }

  // writes contact q of the pair (world frame) to its slot
  __device__ __forceinline__ void emit(int q, REAL dist, const REAL* pos, const REAL* normal_w) const {
    const int c = M.pair_dst[p * MJH_MAX_PAIR_CONTACTS + q];
    const bool cand = M.topk != 0;  // max_contact_points: `c` is a CANDIDATE index, the arrays are the workspace's (the constraint phase selects)
    const int64_t nc = cand ? M.ncand : M.ncon, B = KA.B;
    REAL* const gd = cand ? KA.cand : out.contact_dist;
    REAL* const gp = cand ? KA.cand + B * nc : out.contact_pos;
    REAL* const gf = cand ? KA.cand + 4 * B * nc : out.contact_frame;
    REAL frame[9];
    make_frame(normal_w, frame);
    gd[e * nc + c] = dist;
#pragma unroll
    for (int i = 0; i < 3; i++) gp[(e * nc + c) * 3 + i] = pos[i];
#pragma unroll
    for (int i = 0; i < 9; i++) gf[(e * nc + c) * 9 + i] = frame[i];
  }

  // ---- plane_convex :604-623 ------------------------------------------------------------------------------------------
  __device__ __forceinline__ void plane_convex(const REAL* ppos, const REAL* pmat, const REAL* cpos, const REAL* cmat, const CvxView<REAL>& cv) {
    const REAL rel[3] = {ppos[0] - cpos[0], ppos[1] - cpos[1], ppos[2] - cpos[2]}, nw[3] = {pmat[2], pmat[5], pmat[8]};
    REAL plane_pos[3], n[3];
    mat_t_vec(cmat, rel, plane_pos);
    mat_t_vec(cmat, nw, n);
    const int V = cv.nvert;
    REAL *vert = L, *support = L + 3 * V, *msk = support + V;
    for (int v = l; v < V; v += MJH_WAVE) {
      const REAL x[3] = {cv.vert[3 * v], cv.vert[3 * v + 1], cv.vert[3 * v + 2]};
      const REAL d[3] = {plane_pos[0] - x[0], plane_pos[1] - x[1], plane_pos[2] - x[2]};
      const REAL s = dot3(d, n);
      vert[3 * v] = x[0]; vert[3 * v + 1] = x[1]; vert[3 * v + 2] = x[2];
      support[v] = s;
      msk[v] = s > 0 ? (REAL)1 : (REAL)0;
    }
    wave_sync();
    int idx[4];
    manifold_points(vert, msk, V, n, idx);
    if (l < 4) {
      const int q = l;
      REAL r[3], pos[3];
      mat_vec(cmat, vert + 3 * idx[q], r);
#pragma unroll
      for (int i = 0; i < 3; i++) pos[i] = cpos[i] + r[i];
      int cnt = 0;
      for (int j = 0; j <= q; j++) cnt += idx[j] == idx[q];
      emit(q, cnt == 1 ? -support[idx[q]] : (REAL)1, pos, nw);
    }
  }

  // support of every face of `cv` for a sphere-swept point set; returns the face with the largest negative support
  // (:634-657, :720-744).  NP = 1 (sphere centre) or 2 (capsule end points).  has_support: all(support < 0).
  template <int NP>
  __device__ __forceinline__ int best_face(const CvxView<REAL>& cv, const REAL (*pts)[3], REAL r, bool& has_support) const {
    REAL bv = 0; int bi = -1;
    bool all_neg = true;
    for (int f = l; f < cv.nface; f += MJH_WAVE) {
      const REAL nf[3] = {cv.norm[3 * f], cv.norm[3 * f + 1], cv.norm[3 * f + 2]};
      const REAL* v0 = cv.vert + 3 * fv(cv, f, 0);
      REAL s = 0;
#pragma unroll
      for (int q = 0; q < NP; q++) {
        const REAL d[3] = {(pts[q][0] - nf[0] * r) - v0[0], (pts[q][1] - nf[1] * r) - v0[1], (pts[q][2] - nf[2] * r) - v0[2]};
        const REAL sq = dot3(d, nf);
        s = (q == 0 || sq < s) ? sq : s;
      }
      all_neg = all_neg && (s < 0);
      lane_best<+1>(bv, bi, s >= 0 ? (REAL)-1e12 : s, f);
    }
    if (bi < 0) { bv = (REAL)-1e30; bi = CVX_NONE; }
    wave_argbest<+1>(bv, bi);
    has_support = __all(all_neg);
    return bi;
  }

  // ---- sphere_convex :626-699 (scalar tail, every lane computes it; lane 0 writes) ------------------------------------------
  __device__ __forceinline__ void sphere_convex(const REAL* spos, REAL r, const REAL* cpos, const REAL* cmat, const CvxView<REAL>& cv) {
    const REAL rel[3] = {spos[0] - cpos[0], spos[1] - cpos[1], spos[2] - cpos[2]};
    REAL sp[1][3];
    mat_t_vec(cmat, rel, sp[0]);
    bool hs;
    const int bf = best_face<1>(cv, sp, r, hs);
    const int K = cv.nfv;
    const REAL normal[3] = {cv.norm[3 * bf], cv.norm[3 * bf + 1], cv.norm[3 * bf + 2]};
    REAL pt[3];
    project_pt(sp[0], cv.vert + 3 * fv(cv, bf, 0), normal, pt);
    bool inside = true;
    REAL best = 0; int ei = 0;
    for (int k = 0; k < K; k++) {
      const REAL *f1 = cv.vert + 3 * fv(cv, bf, k), *f0 = cv.vert + 3 * fv(cv, bf, k == 0 ? K - 1 : k - 1);
      const REAL ed[3] = {f1[0] - f0[0], f1[1] - f0[1], f1[2] - f0[2]}, q[3] = {pt[0] - f0[0], pt[1] - f0[1], pt[2] - f0[2]};
      REAL en[3];
      cross3(ed, normal, en);
      const REAL d = dot3(q, en);
      if (!(d <= 0)) inside = false;
      const bool degenerate = en[0] == 0 && en[1] == 0 && en[2] == 0;
      const REAL dm = (degenerate || d < 0) ? (REAL)1e12 : d;
      if (k == 0 || dm < best) { best = dm; ei = k; }
    }
    if (!inside) {
      REAL ept[3];
      closest_segment_point(cv.vert + 3 * fv(cv, bf, ei == 0 ? K - 1 : ei - 1), cv.vert + 3 * fv(cv, bf, ei), pt, ept);
      pt[0] = ept[0]; pt[1] = ept[1]; pt[2] = ept[2];
    }
    REAL n[3] = {pt[0] - sp[0][0], pt[1] - sp[0][1], pt[2] - sp[0][2]};
    const REAL d = normalize_n<REAL, 3>(n);
    REAL lp[3], nw[3], pw[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { const REAL spt = sp[0][i] + n[i] * r; lp[i] = (pt[i] + spt) * (REAL)0.5; }
    mat_vec(cmat, n, nw);
    mat_vec(cmat, lp, pw);
#pragma unroll
    for (int i = 0; i < 3; i++) pw[i] = pw[i] + cpos[i];
    if (l == 0) emit(0, d - r, pw, nw);
  }

  // ---- capsule_convex :702-802 -----------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void capsule_convex(const REAL* kpos, const REAL* kmat, REAL r, REAL halflen, const REAL* cpos, const REAL* cmat, const CvxView<REAL>& cv) {
    const REAL rel[3] = {kpos[0] - cpos[0], kpos[1] - cpos[1], kpos[2] - cpos[2]}, axw[3] = {kmat[2], kmat[5], kmat[8]};
    REAL cp[3], axis[3], pts[2][3];
    mat_t_vec(cmat, rel, cp);
    mat_t_vec(cmat, axw, axis);
#pragma unroll
    for (int i = 0; i < 3; i++) { const REAL sg = axis[i] * halflen; pts[0][i] = cp[i] - sg; pts[1][i] = cp[i] + sg; }
    bool has_support;
    const int bf = best_face<2>(cv, pts, r, has_support);
    const int K = cv.nfv;
    const REAL normal[3] = {cv.norm[3 * bf], cv.norm[3 * bf + 1], cv.norm[3 * bf + 2]};
    // face polygon and its side-plane normals in LDS; the closest points of every face edge to the capsule segment per lane
    REAL *face = L, *en = L + 3 * K;
    REAL ebv = 0; int ebi = -1;
    REAL ecl[3] = {0, 0, 0}, ccl[3] = {0, 0, 0};
    for (int k = l; k < K; k += MJH_WAVE) {
      const REAL *f1 = cv.vert + 3 * fv(cv, bf, k), *f0 = cv.vert + 3 * fv(cv, bf, k == 0 ? K - 1 : k - 1);
      const REAL ed[3] = {f1[0] - f0[0], f1[1] - f0[1], f1[2] - f0[2]};
      REAL n3[3];
      cross3(ed, normal, n3);
#pragma unroll
      for (int i = 0; i < 3; i++) { face[3 * k + i] = f1[i]; en[3 * k + i] = n3[i]; }
      REAL ea[3], ca[3];
      Env<REAL>::closest_segment_to_segment(f0, f1, pts[0], pts[1], ea, ca);
      const REAL x0 = ea[0] - ca[0], x1 = ea[1] - ca[1], x2 = ea[2] - ca[2];
      const REAL dist2 = x0 * x0 + x1 * x1 + x2 * x2;
      if (ebi < 0 || dist2 < ebv) { ebv = dist2; ebi = k; ecl[0] = ea[0]; ecl[1] = ea[1]; ecl[2] = ea[2]; ccl[0] = ca[0]; ccl[1] = ca[1]; ccl[2] = ca[2]; }
    }
    if (ebi < 0) { ebv = (REAL)1e30; ebi = CVX_NONE; }
    wave_argbest<-1>(ebv, ebi);
    const int owner = ebi % MJH_WAVE;  // lane k % 64 handled edge k
#pragma unroll
    for (int i = 0; i < 3; i++) { ecl[i] = read_lane(ecl[i], owner); ccl[i] = read_lane(ccl[i], owner); }  // (owner is wave-uniform: a v_readlane, not an LDS-pipe shuffle)
    wave_sync();
    REAL cl[2][3];
    const bool mask = clip_edge_to_planes(pts[0], pts[1], face, en, K, cl[0], cl[1]);
    REAL lpos[2][3], lnorm[2][3], pen[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      REAL fp[3], d3[3];
#pragma unroll
      for (int i = 0; i < 3; i++) cl[q][i] = cl[q][i] - normal[i] * r;
      project_pt(cl[q], face, normal, fp);
#pragma unroll
      for (int i = 0; i < 3; i++) { lpos[q][i] = (cl[q][i] + fp[i]) * (REAL)0.5; lnorm[q][i] = normal[i]; d3[i] = fp[i] - cl[q][i]; }
      pen[q] = (mask && has_support) ? dot3(d3, normal) : (REAL)-1;
    }
    REAL eax[3] = {ccl[0] - ecl[0], ccl[1] - ecl[1], ccl[2] - ecl[2]};
    const REAL edist = normalize_n<REAL, 3>(eax);
    const REAL epen = r - edist;
    if (epen > 0) {
#pragma unroll
      for (int i = 0; i < 3; i++) { lpos[0][i] = (ecl[i] + (ccl[i] - eax[i] * r)) * (REAL)0.5; lnorm[0][i] = eax[i]; }
      pen[0] = epen;
    }
    if (l < 2) {
      const int q = l;
      const REAL nl[3] = {-lnorm[q][0], -lnorm[q][1], -lnorm[q][2]};
      REAL pw[3], nw[3];
      mat_vec(cmat, lpos[q], pw);
      mat_vec(cmat, nl, nw);
#pragma unroll
      for (int i = 0; i < 3; i++) pw[i] = cpos[i] + pw[i];
      emit(q, -pen[q], pw, nw);
    }
  }

  // separating axis number a (:495-503): normals of hull 1, normals of hull 2, normalised edge x edge (index j * E1 + i)
  __device__ __forceinline__ static void sat_axis(int a, const CvxView<REAL>& c1, const CvxView<REAL>& c2, const REAL* v1, const REAL* n1, const REAL* v2,
                                                  const REAL* n2, REAL* axis) {
    const int F1 = c1.nface, F2 = c2.nface, E1 = c1.nedge;
    if (a < F1) { axis[0] = n1[3 * a]; axis[1] = n1[3 * a + 1]; axis[2] = n1[3 * a + 2]; return; }
    if (a < F1 + F2) { const int f = a - F1; axis[0] = n2[3 * f]; axis[1] = n2[3 * f + 1]; axis[2] = n2[3 * f + 2]; return; }
    const int ed = a - F1 - F2, i1 = ed % E1, j2 = ed / E1;
    const REAL *a0 = v1 + 3 * c1.edge[2 * i1], *a1 = v1 + 3 * c1.edge[2 * i1 + 1], *b0 = v2 + 3 * c2.edge[2 * j2], *b1 = v2 + 3 * c2.edge[2 * j2 + 1];
    const REAL da[3] = {a0[0] - a1[0], a0[1] - a1[1], a0[2] - a1[2]}, db[3] = {b0[0] - b1[0], b0[1] - b1[1], b0[2] - b1[2]};
    cross3(da, db, axis);
    normalize_n<REAL, 3>(axis);
  }

  // ---- convex_convex :805-856 with _sat_hull_hull :464-601 ------------------------------------------------------------------------
  __device__ __forceinline__ void convex_convex(const REAL* pos1_in, const REAL* mat1_in, CvxView<REAL> c1, const REAL* pos2_in, const REAL* mat2_in, CvxView<REAL> c2) {
    CSTAMP(99);
    const int K = c1.nfv > c2.nfv ? c1.nfv : c2.nfv;
    const bool swapped = c1.nvert > c2.nvert;
    // the two frames are swapped by VALUE (selects on registers): swapping the pointers made the compiler keep the caller's four arrays in
    // scratch memory (112 bytes per lane) for every later read
    REAL pos1[3], pos2[3], mat1[9], mat2[9];
#pragma unroll
    for (int i = 0; i < 3; i++) { pos1[i] = swapped ? pos2_in[i] : pos1_in[i]; pos2[i] = swapped ? pos1_in[i] : pos2_in[i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) { mat1[i] = swapped ? mat2_in[i] : mat1_in[i]; mat2[i] = swapped ? mat1_in[i] : mat2_in[i]; }
    if (swapped) {
      const CvxView<REAL> c = c1; c1 = c2; c2 = c;
    }
    const int V1 = c1.nvert, V2 = c2.nvert, F1 = c1.nface, F2 = c2.nface, E1 = c1.nedge, E2 = c2.nedge;
    const REAL rel[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
    REAL tlp[3], tlm[9];
    mat_t_vec(mat2, rel, tlp);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) tlm[3 * i + j] = mat2[i] * mat1[j] + mat2[3 + i] * mat1[3 + j] + mat2[6 + i] * mat1[6 + j];
    // LDS: hull 1 in hull 2's frame, hull 2 as is, then the polygons and the clipped point set
    REAL *v1 = L, *n1 = v1 + 3 * V1, *v2 = n1 + 3 * F1, *n2 = v2 + 3 * V2;
    REAL *clip = n2 + 3 * F2, *subj = clip + 3 * K, *cpn = subj + 3 * K, *spn = cpn + 3 * K, *c1s = spn + 3 * K;
    const int P = 4 * K;
    REAL *inc = c1s + 3 * K, *ref = inc + 3 * P, *msk = ref + 3 * P;
    for (int v = l; v < V1; v += MJH_WAVE) {
      const REAL x[3] = {c1.vert[3 * v], c1.vert[3 * v + 1], c1.vert[3 * v + 2]};
      REAL t[3];
      mat_vec(tlm, x, t);
#pragma unroll
      for (int i = 0; i < 3; i++) v1[3 * v + i] = tlp[i] + t[i];
    }
    for (int f = l; f < F1; f += MJH_WAVE) {
      const REAL x[3] = {c1.norm[3 * f], c1.norm[3 * f + 1], c1.norm[3 * f + 2]};
      REAL t[3];
      mat_vec(tlm, x, t);
#pragma unroll
      for (int i = 0; i < 3; i++) n1[3 * f + i] = t[i];
    }
    for (int i = l; i < 3 * V2; i += MJH_WAVE) v2[i] = c2.vert[i];
    for (int i = l; i < 3 * F2; i += MJH_WAVE) n2[i] = c2.norm[i];
    wave_sync();
    CSTAMP(90);
    // separating axis test: lanes over axes
    const int NA = F1 + F2 + E1 * E2;
    REAL bd = 0; int ba = -1, bsign = 1;
    REAL baxis[3] = {0, 0, 0};
    // two axes per lane and pass (a, a + 64): every vertex read from LDS serves both projections and the multiply-adds pair up into packed
    // float32 operations; each projection is the same dot3 expression, and a lane still meets its axes in increasing index order
    for (int a0 = l; a0 < NA; a0 += 2 * MJH_WAVE) {
      const int a1 = a0 + MJH_WAVE;
      const bool has1 = a1 < NA;
      REAL ax0[3], ax1[3];
      sat_axis(a0, c1, c2, v1, n1, v2, n2, ax0);
      sat_axis(has1 ? a1 : a0, c1, c2, v1, n1, v2, n2, ax1);
      // running extremes as hardware max / min from +-infinity (one instruction each; the same values as compare-and-keep for finite projections)
      const REAL inf = (REAL)INFINITY;
      REAL amax0 = -inf, amin0 = inf, bmax0 = -inf, bmin0 = inf, amax1 = -inf, amin1 = inf, bmax1 = -inf, bmin1 = inf;
      // Four vertices per trip: their twelve coordinates are requested together (one LDS round trip instead of four -- the loop was one exposed ds_read latency per
      // vertex) and their extremes are folded as a tree before they meet the running ones (a max / min whose operand came round the loop is canonicalised first:
      // one extra instruction per accumulator and trip, now per four vertices).  max / min are exact and order-independent: the same values as the one-by-one scan.
      auto project = [&](const REAL* vv, int nvx, REAL& mx0, REAL& mn0, REAL& mx1, REAL& mn1) {
        int v = 0;
        for (; v + 4 <= nvx; v += 4) {
          REAL c[12], s0[4], s1[4];
#pragma unroll
          for (int t = 0; t < 12; t++) c[t] = vv[3 * v + t];
          if constexpr (sizeof(REAL) == 4) {
            // float32: the two axes of the lane ride in the two halves of packed registers -- three v_pk_mul_f32 and two v_pk_add_f32 per vertex for BOTH projections
            // (the compiler's own pairing left the additions scalar: 16 v_add_f32 per trip).  Each half is the same (a x + b y) + c z expression, rounded the same way.
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 ax = {(float)ax0[0], (float)ax1[0]}, ay = {(float)ax0[1], (float)ax1[1]}, az = {(float)ax0[2], (float)ax1[2]};
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const f2 x = {(float)c[3 * t], (float)c[3 * t]}, y = {(float)c[3 * t + 1], (float)c[3 * t + 1]}, z = {(float)c[3 * t + 2], (float)c[3 * t + 2]};
              const f2 sv = (ax * x + ay * y) + az * z;
              s0[t] = (REAL)sv[0]; s1[t] = (REAL)sv[1];
            }
          } else {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            s0[t] = (ax0[0] * c[3 * t] + ax0[1] * c[3 * t + 1]) + ax0[2] * c[3 * t + 2];
            s1[t] = (ax1[0] * c[3 * t] + ax1[1] * c[3 * t + 1]) + ax1[2] * c[3 * t + 2];
          }
          }
          mx0 = r_max(mx0, r_max(r_max(s0[0], s0[1]), r_max(s0[2], s0[3]))); mn0 = r_min(mn0, r_min(r_min(s0[0], s0[1]), r_min(s0[2], s0[3])));
          mx1 = r_max(mx1, r_max(r_max(s1[0], s1[1]), r_max(s1[2], s1[3]))); mn1 = r_min(mn1, r_min(r_min(s1[0], s1[1]), r_min(s1[2], s1[3])));
        }
        for (; v < nvx; v++) {
          const REAL x = vv[3 * v], y = vv[3 * v + 1], z = vv[3 * v + 2];
          const REAL s0 = (ax0[0] * x + ax0[1] * y) + ax0[2] * z, s1 = (ax1[0] * x + ax1[1] * y) + ax1[2] * z;
          mx0 = r_max(mx0, s0); mn0 = r_min(mn0, s0); mx1 = r_max(mx1, s1); mn1 = r_min(mn1, s1);
        }
      };
      project(v1, V1, amax0, amin0, amax1, amin1);
      project(v2, V2, bmax0, bmin0, bmax1, bmin1);
      {
        const REAL d1 = amax0 - bmin0, d2 = bmax0 - amin0;
        REAL d = d1 < d2 ? d1 : d2;
        if (ax0[0] == 0 && ax0[1] == 0 && ax0[2] == 0) d = (REAL)1e6;
        if (ba < 0 || d < bd) { bd = d; ba = a0; bsign = d1 > d2 ? -1 : 1; baxis[0] = ax0[0]; baxis[1] = ax0[1]; baxis[2] = ax0[2]; }
      }
      if (has1) {
        const REAL d1 = amax1 - bmin1, d2 = bmax1 - amin1;
        REAL d = d1 < d2 ? d1 : d2;
        if (ax1[0] == 0 && ax1[1] == 0 && ax1[2] == 0) d = (REAL)1e6;
        if (d < bd) { bd = d; ba = a1; bsign = d1 > d2 ? -1 : 1; baxis[0] = ax1[0]; baxis[1] = ax1[1]; baxis[2] = ax1[2]; }
      }
    }
    CSTAMP(91);
    if (ba < 0) { bd = (REAL)1e30; ba = CVX_NONE; }
    wave_argbest<-1>(bd, ba);
    const int owner = ba % MJH_WAVE;
    const int best_sign = read_lane(bsign, owner);
    REAL best_axis[3];
#pragma unroll
    for (int i = 0; i < 3; i++) best_axis[i] = read_lane(baxis[i], owner);
    const bool is_edge = ba >= F1 + F2;
    // faces most aligned / most opposed to the axis: lanes over faces
    REAL v_amax = 0, v_amin = 0, v_bmax = 0, v_bmin = 0;
    int a_max = -1, a_min = -1, b_max = -1, b_min = -1;
    for (int f = l; f < F1; f += MJH_WAVE) { const REAL s = dot3(best_axis, n1 + 3 * f); lane_best<+1>(v_amax, a_max, s, f); lane_best<-1>(v_amin, a_min, s, f); }
    for (int f = l; f < F2; f += MJH_WAVE) { const REAL s = dot3(best_axis, n2 + 3 * f); lane_best<+1>(v_bmax, b_max, s, f); lane_best<-1>(v_bmin, b_min, s, f); }
    if (a_max < 0) { v_amax = (REAL)-1e30; a_max = CVX_NONE; v_amin = (REAL)1e30; a_min = CVX_NONE; }
    if (b_max < 0) { v_bmax = (REAL)-1e30; b_max = CVX_NONE; v_bmin = (REAL)1e30; b_min = CVX_NONE; }
    wave_argbest<+1>(v_amax, a_max); wave_argbest<-1>(v_amin, a_min); wave_argbest<+1>(v_bmax, b_max); wave_argbest<-1>(v_bmin, b_min);
    REAL clip_n[3], subj_n[3], sep[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      clip_n[i] = best_sign > 0 ? n1[3 * a_max + i] : n2[3 * b_max + i];
      subj_n[i] = best_sign > 0 ? n2[3 * b_min + i] : n1[3 * a_min + i];
      sep[i] = (REAL)(-best_sign) * best_axis[i];
    }
    CSTAMP(92);
    // reference (clipping) and incident (subject) polygons
    for (int k = l; k < K; k += MJH_WAVE) {
      const REAL* rp = best_sign > 0 ? v1 + 3 * fv(c1, a_max, k) : v2 + 3 * fv(c2, b_max, k);
      const REAL* ip = best_sign > 0 ? v2 + 3 * fv(c2, b_min, k) : v1 + 3 * fv(c1, a_min, k);
#pragma unroll
      for (int i = 0; i < 3; i++) { clip[3 * k + i] = rp[i]; subj[3 * k + i] = ip[i]; }
    }
    wave_sync();
    {  // side-plane normals (:351-367) and the clipping polygon projected onto the subject plane (:249-257, :377-379)
      const REAL d = dot3(subj, subj_n);
      const REAL denom = dot3(clip_n, subj_n);
      const REAL den = denom + (denom == 0 ? kEps32 : (REAL)0);
      for (int k = l; k < K; k += MJH_WAVE) {
        const int km = k == 0 ? K - 1 : k - 1;
        const REAL ec[3] = {clip[3 * k] - clip[3 * km], clip[3 * k + 1] - clip[3 * km + 1], clip[3 * k + 2] - clip[3 * km + 2]};
        const REAL es[3] = {subj[3 * k] - subj[3 * km], subj[3 * k + 1] - subj[3 * km + 1], subj[3 * k + 2] - subj[3 * km + 2]};
        REAL t3[3];
        cross3(ec, clip_n, t3);
        cpn[3 * k] = t3[0]; cpn[3 * k + 1] = t3[1]; cpn[3 * k + 2] = t3[2];
        cross3(es, subj_n, t3);
        spn[3 * k] = t3[0]; spn[3 * k + 1] = t3[1]; spn[3 * k + 2] = t3[2];
        const REAL t1 = (d - dot3(clip + 3 * k, subj_n)) / den;
#pragma unroll
        for (int i = 0; i < 3; i++) c1s[3 * k + i] = clip[3 * k + i] + t1 * clip_n[i];
      }
    }
    wave_sync();
    CSTAMP(93);
    // clip: K subject edges against the clipping side planes, K projected clipping edges against the subject side planes
    for (int t = l; t < 2 * K; t += MJH_WAVE) {
      const bool first = t < K;
      const int k = first ? t : t - K, km = k == 0 ? K - 1 : k - 1;
      const REAL* poly = first ? subj : c1s;
      const REAL p0[3] = {poly[3 * km], poly[3 * km + 1], poly[3 * km + 2]}, p1[3] = {poly[3 * k], poly[3 * k + 1], poly[3 * k + 2]};
      REAL o0[3], o1[3];
      const bool mk = clip_edge_to_planes(p0, p1, first ? clip : subj, first ? cpn : spn, K, o0, o1);
#pragma unroll
      for (int i = 0; i < 3; i++) { inc[6 * t + i] = o0[i]; inc[6 * t + 3 + i] = o1[i]; }
      msk[2 * t] = msk[2 * t + 1] = mk ? (REAL)1 : (REAL)0;
    }
    wave_sync();
    {  // reference points = incident points projected onto the clipping plane; keep those behind it (:420-424)
      REAL nn[3] = {clip_n[0], clip_n[1], clip_n[2]};
      normalize_n<REAL, 3>(nn);
      const REAL neg[3] = {-clip_n[0], -clip_n[1], -clip_n[2]};
      for (int q = l; q < P; q += MJH_WAVE) {
        REAL r3[3];
        project_pt(inc + 3 * q, clip, nn, r3);
        ref[3 * q] = r3[0]; ref[3 * q + 1] = r3[1]; ref[3 * q + 2] = r3[2];
        const REAL d3[3] = {inc[3 * q] - clip[0], inc[3 * q + 1] - clip[1], inc[3 * q + 2] - clip[2]};
        msk[q] = (msk[q] != 0 && dot3(d3, neg) > (REAL)1e-6) ? (REAL)1 : (REAL)0;
      }
    }
    wave_sync();
    CSTAMP(94);
    int best[4];
    manifold_points(ref, msk, P, clip_n, best);
    CSTAMP(95);
    REAL ldist[4], lpos[4][3];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int b = best[q];
      const REAL pd[3] = {inc[3 * b] - ref[3 * b], inc[3 * b + 1] - ref[3 * b + 1], inc[3 * b + 2] - ref[3 * b + 2]};
      const REAL neg[3] = {-clip_n[0], -clip_n[1], -clip_n[2]};
      const REAL pen = dot3(pd, neg);
      ldist[q] = msk[b] != 0 ? -pen : (REAL)1;
      lpos[q][0] = ref[3 * b]; lpos[q][1] = ref[3 * b + 1]; lpos[q][2] = ref[3 * b + 2];
    }
    if (is_edge) {  // :581-599
      int idx = 0;
#pragma unroll
      for (int q = 1; q < 4; q++) if (ldist[q] < ldist[idx]) idx = q;
      REAL dd = ldist[0], pp[3] = {lpos[0][0], lpos[0][1], lpos[0][2]};
#pragma unroll
      for (int q = 1; q < 4; q++) if (q == idx) { dd = ldist[q]; pp[0] = lpos[q][0]; pp[1] = lpos[q][1]; pp[2] = lpos[q][2]; }
#pragma unroll
      for (int q = 0; q < 4; q++) { ldist[q] = q == 0 ? dd : (REAL)1; lpos[q][0] = pp[0]; lpos[q][1] = pp[1]; lpos[q][2] = pp[2]; }
    }
    const REAL lnormal[3] = {-sep[0], -sep[1], -sep[2]};
    REAL nw[3];
    mat_vec(mat2, lnormal, nw);
    if (swapped) { nw[0] = -nw[0]; nw[1] = -nw[1]; nw[2] = -nw[2]; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (l == q) {
        REAL pw[3];
        mat_vec(mat2, lpos[q], pw);
        pw[0] = pos2[0] + pw[0]; pw[1] = pos2[1] + pw[1]; pw[2] = pos2[2] + pw[2];
        emit(q, ldist[q], pw, nw);
      }
    }
    CSTAMP(96);
  }

  __device__ __forceinline__ void run() {
#ifdef MJH_STAMPS
    cstamp_prev = __builtin_amdgcn_s_memtime();
#endif
    const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p], fn = M.pair_fn[p];
    const int64_t ng = M.ngeom;
    const REAL *gp = out.geom_xpos + e * ng * 3, *gm = out.geom_xmat + e * ng * 9;
    REAL p1[3], m1[9], p2[3], m2[9];
#pragma unroll
    for (int i = 0; i < 3; i++) { p1[i] = gp[3 * g1 + i]; p2[i] = gp[3 * g2 + i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) { m1[i] = gm[9 * g1 + i]; m2[i] = gm[9 * g2 + i]; }
    const REAL* s1 = M.geom_size + 3 * g1;
    if (fn == MJH_FN_PLANE_CONVEX) plane_convex(p1, m1, p2, m2, view(g2));
    else if (fn == MJH_FN_SPHERE_CONVEX) sphere_convex(p1, s1[0], p2, m2, view(g2));
    else if (fn == MJH_FN_CAPSULE_CONVEX) capsule_convex(p1, m1, s1[0], s1[1], p2, m2, view(g2));
    else if (fn == MJH_FN_CONVEX_CONVEX) convex_convex(p1, m1, view(g1), p2, m2, view(g2));
  }
};

// grid: B * ncvxpair workgroups of one wave (grid-stride beyond 2^20)
template <typename REAL>
#ifndef MJH_CVX32_WAVES
#define MJH_CVX32_WAVES 4  /* float32 convex narrow phase: waves per SIMD the register allocation must fit -- mesh scene, B = 8192: 3 waves (161 VGPRs) 114.7 us, 4 waves (128 VGPRs + 52 B scratch) 103.5 us, 5 waves: see profiles/r03/notes.md */
#endif
__global__ __launch_bounds__(MJH_WAVE, sizeof(REAL) == 4 ? MJH_CVX32_WAVES : 1) void mjh_convex_kernel(KArgs<REAL> args) {
  extern __shared__ unsigned char cvx_smem[];
  REAL* lds = reinterpret_cast<REAL*>(cvx_smem);
  const int npc = M.ncvxpair;
  {  // one workgroup per (environment, pair): no grid-stride loop (the host launches per 2^22 items, KArgs::env_begin = first item) -- see mjh_phase_kernel
    const int64_t item = KA.env_begin + blockIdx.x;
    const int64_t env = item / npc;
    const int pair = M.cvx_pairs[(int)(item - env * npc)];
    CvxPair<REAL>(lds, env, pair).run();
    wave_sync();
  }
}

#undef M
#undef out
#undef KA
