/*
 * mjoracle_impl.h -- scalar CPU restatement of the reference step, one environment at a time.
 *
 * TEST INFRASTRUCTURE (see oracle/README.md): this is the checker for the HIP path and the
 * "port" CPU baseline of bench.py.  It is never linked into or called from the product package.
 * Included twice by mjoracle.c with REAL = double / float and a name suffix SFX.
 *
 * Every function cites the reference code it follows (paths under
 * /root/reference/mujoco_torch/_src/).  Operation order follows the Python expressions; where the
 * reference calls a BLAS/LAPACK routine (matmul, linalg.cholesky) the textbook loop is used and
 * agreement is to rounding (tests state the tolerance).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#ifndef MJO_COMMON_
#define MJO_COMMON_
#define mjMINVAL 1e-15
#define mjMAXVAL 1e10
#define mjMINIMP 1e-4
#define mjMAXIMP 0.9999
#define JNT_FREE 0
#define JNT_BALL 1
#define JNT_SLIDE 2
#define JNT_HINGE 3
#define DSBL_CONSTRAINT (1 << 0)
#define DSBL_SPRING (1 << 5)
#define DSBL_DAMPER (1 << 6)
#define DSBL_GRAVITY (1 << 7)
#define DSBL_CLAMPCTRL (1 << 8)
#define DSBL_WARMSTART (1 << 9)
#define DSBL_ACTUATION (1 << 11)
#define DSBL_REFSAFE (1 << 12)
#define DSBL_EULERDAMP (1 << 15)
#define INT_EULER 0
#define INT_RK4 1

/* MJO_TRACE=1: the solver prints its iterations and line searches to stderr (triage of campaign environments, tools/fuzz_triage.py); read once */
#ifndef MJO_TRACE_HELPER
#define MJO_TRACE_HELPER
static int mjo_trace_flag = -1;
static inline int mjo_trace_on(void) {
  if (mjo_trace_flag < 0) mjo_trace_flag = getenv("MJO_TRACE") != NULL;
  return mjo_trace_flag;
}
#endif
#define SOL_CG 1
#define SOL_NEWTON 2
#define CONE_ELLIPTIC 1
#define CAM_FIXED 0
#define CAM_TRACK 1
#define CAM_TRACKCOM 2
#define CAM_TARGETBODY 3
#define CAM_TARGETBODYCOM 4
#define GAIN_FIXED 0
#define GAIN_AFFINE 1
#define BIAS_NONE 0
#define BIAS_AFFINE 1
#define GAIN_MUSCLE 2
#define BIAS_MUSCLE 2
#define DYN_NONE 0
#define DYN_INTEGRATOR 1
#define DYN_FILTER 2
#define DYN_FILTEREXACT 3
#define DYN_MUSCLE 4
#define INLINE_CHOL_MAX 16 /* math.py:84 */
#define MJO_MAX_TIE 16
#define MJO_MAX_TIE_RUNS 512
/* constants the reference keeps in _CachedConst are float32 literals up-cast to the data dtype */
#define MINVAL_CACHED ((float)1e-15)
#endif

#ifdef REAL_IS_FLOAT
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_COS cosf
#define R_ATAN2 atan2f
#define R_POW powf
#define R_FABS fabsf
#define R_EXP expf
#else
#define R_SQRT sqrt
#define R_SIN sin
#define R_COS cos
#define R_ATAN2 atan2
#define R_POW pow
#define R_FABS fabs
#define R_EXP exp
#endif

/* ---- model constants converted to REAL -------------------------------------------------- */
typedef struct FN(MjoModel) {
  const mjhModelDesc* d;
  REAL timestep, impratio, meaninertia, gravity[3];
  REAL density, viscosity, wind[3];
  int has_fluid;
  int has_gravcomp;
#define X(n) REAL* n;
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
} FN(MjoModel);

/* ---- per-environment workspace (all Data leaves + solver scratch) ------------------------ */
typedef struct FN(MjoWork) {
#define X(n) REAL* n;
  MJH_DATA_REALS(X)
#undef X
  /* scratch */
  REAL *cacc, *cfrc, *sub_mass, *sub_pos, *crb_cdof, *jacdiff, *tmp_nv, *tmp_nv2, *tmp_nefc;
  REAL *efc_pos, *efc_pos_norm, *efc_invweight, *efc_solref, *efc_solimp;
  REAL *qM2, *qLD2, *H, *HL;
  /* solver context (solver.py:95-126) */
  REAL *s_qacc, *s_qfrc, *s_Jaref, *s_force, *s_Ma, *s_grad, *s_Mgrad, *s_search, *s_mv, *s_jv, *s_quad;
  REAL *s_prev_grad, *s_prev_Mgrad;
  unsigned char* s_active;
  /* rk4 */
  REAL *rk_qpos0, *rk_qvel0, *rk_act0, *rk_qvel, *rk_qacc, *rk_actdot, *rk_kqvel;
  REAL* in_subtree_com;
  const REAL *x_cacc, *x_cfrc_int, *x_subtree_linvel, *x_subtree_angmom; /* this env's rows of the input-only leaves the sensors read (MJH_DATA_EXTRA_IN), NULL = zeros */
  /* diagnostic: number of line-search candidates that were distinct points with a derivative that is
     pure rounding noise (|deriv_0| < 1e-8 of the initial slope).  Whether such a candidate is
     accepted depends on the sign / exact-zeroness of that noise (solver.py:440-463), so the
     reference's own result is implementation-defined on these steps; tests exempt them from the
     tight solver-output tolerance (DESIGN.md "line-search knife edge"). */
  int nf, ne_nf; /* copies of the model's row counts for the line search's point function */
  int knife;
  int knife_policy; /* <0: natural; j>=0: first j noise candidates read as exact zero, the (j+1)-th as non-zero */
  /* index selections (argmax / argmin) of the convex narrow phase whose two best candidates differ by rounding noise
     only: the reference's pick is implementation-defined there.  Each such selection is an "event"; tie_digit[e]
     chooses the e-th event's candidate (0 = the natural pick).  collision() enumerates the digit combinations of a
     pair and keeps the outcome closest to the caller's contact hint (DESIGN.md "narrow-phase ties"). */
  int tie_on, tie_n, tie_digit[MJO_MAX_TIE], tie_count[MJO_MAX_TIE];
  const REAL *hint_dist, *hint_pos, *hint_frame; /* this env's expected contact leaves or NULL */
  int tie_pairs; /* pairs whose kept outcome is not the natural one */
  /* ties inside RK4 stages 1..3 cannot be hinted (those contacts are never returned): there tie_on == 2 counts the events of the
     environment's step in stage_tie_n, and the events whose bit is set in stage_tie_flip (first 32 events) take their second
     candidate instead of the natural one -- callers enumerate single and double flips (tests/_util.oracle_alternatives) */
  int stage_mode, stage_tie_n; unsigned stage_tie_flip;
  int stat_solves, stat_niter, stat_ls, stat_rows; /* work counters of the step (diagnostics: solver calls, solver iterations, line-search iterations, active contact rows) */
  const REAL* prim_hint_n; /* hinted normal of the primitive pair being evaluated (coincident-centre case of sphere_sphere_), or NULL */
  int prim_adopted;
  const int32_t* eq_active; /* this env's Data.eq_active (input leaf, types.py:1103) */
  /* max_contact_points (collision_driver.py:822-840): candidate contacts of the step and the candidate kept in each contact slot */
  REAL *cand_dist, *cand_pos, *cand_frame;
  int* con_src;
} FN(MjoWork);

/* ---- small vector math (math.py) ------------------------------------------------------------ */
static inline void FN(cross3)(const REAL* a, const REAL* b, REAL* o) { /* math.py:63-76 */
  REAL o0 = a[1] * b[2] - a[2] * b[1];
  REAL o1 = a[2] * b[0] - a[0] * b[2];
  REAL o2 = a[0] * b[1] - a[1] * b[0];
  o[0] = o0; o[1] = o1; o[2] = o2;
}
static inline REAL FN(dot3)(const REAL* a, const REAL* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static inline REAL FN(norm_n)(const REAL* x, int n) { /* math.norm :196-213 */
  int all_zero = 1;
  for (int i = 0; i < n; i++) if (x[i] != 0) all_zero = 0;
  if (all_zero) return 0;
  REAL s = 0;
  for (int i = 0; i < n; i++) s += x[i] * x[i];
  return R_SQRT(s);
}
static inline REAL FN(normalize_n)(REAL* x, int n) { /* normalize_with_norm :216-230 */
  REAL nn = FN(norm_n)(x, n);
  REAL den = nn + (REAL)1e-6 * (REAL)(nn == 0);
  for (int i = 0; i < n; i++) x[i] = x[i] / den;
  return nn;
}
static inline void FN(rotate)(const REAL* v, const REAL* q, REAL* o) { /* math.rotate :246-261 */
  REAL s = q[0];
  const REAL* u = q + 1;
  REAL uv = FN(dot3)(u, v), uu = FN(dot3)(u, u);
  REAL c[3];
  FN(cross3)(u, v, c);
  REAL r[3];
  for (int i = 0; i < 3; i++) r[i] = 2 * (uv * u[i]) + (s * s - uu) * v[i];
  for (int i = 0; i < 3; i++) o[i] = r[i] + 2 * s * c[i];
}
static inline void FN(quat_mul)(const REAL* u, const REAL* v, REAL* o) { /* :283-300 */
  REAL o0 = u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3];
  REAL o1 = u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2];
  REAL o2 = u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1];
  REAL o3 = u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0];
  o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}
static inline void FN(quat_to_mat)(const REAL* q, REAL* m) { /* :323-351 */
  REAL p[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p[i][j] = q[i] * q[j];
  m[0] = p[0][0] + p[1][1] - p[2][2] - p[3][3];
  m[1] = 2 * (p[1][2] - p[0][3]);
  m[2] = 2 * (p[1][3] + p[0][2]);
  m[3] = 2 * (p[1][2] + p[0][3]);
  m[4] = p[0][0] - p[1][1] + p[2][2] - p[3][3];
  m[5] = 2 * (p[2][3] - p[0][1]);
  m[6] = 2 * (p[1][3] - p[0][2]);
  m[7] = 2 * (p[2][3] + p[0][1]);
  m[8] = p[0][0] - p[1][1] - p[2][2] + p[3][3];
}
static inline void FN(axis_angle_to_quat)(const REAL* axis, REAL angle, REAL* q) { /* :363-374 */
  REAL s = R_SIN(angle * (REAL)0.5), c = R_COS(angle * (REAL)0.5);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static inline void FN(quat_to_axis_angle)(const REAL* q, REAL* axis, REAL* angle) { /* :354-360 */
  axis[0] = q[1]; axis[1] = q[2]; axis[2] = q[3];
  REAL sin_a_2 = FN(normalize_n)(axis, 3);
  REAL a = 2 * R_ATAN2(sin_a_2, q[0]);
  if (a > (REAL)M_PI) a = a - 2 * (REAL)M_PI;
  *angle = a;
}
static inline void FN(quat_sub)(const REAL* u, const REAL* v, REAL* o) { /* :276-280 */
  REAL vi[4] = {v[0], -v[1], -v[2], -v[3]}, q[4], axis[3], angle;
  FN(quat_mul)(vi, u, q);
  FN(quat_to_axis_angle)(q, axis, &angle);
  for (int i = 0; i < 3; i++) o[i] = axis[i] * angle;
}
static inline void FN(quat_integrate)(const REAL* q, const REAL* w, REAL dt, REAL* o) { /* :377-383 */
  REAL v[3] = {w[0], w[1], w[2]};
  REAL nrm = FN(normalize_n)(v, 3);
  REAL angle = dt * nrm, qr[4], r[4];
  FN(axis_angle_to_quat)(v, angle, qr);
  FN(quat_mul)(q, qr, r);
  FN(normalize_n)(r, 4);
  for (int i = 0; i < 4; i++) o[i] = r[i];
}
static inline void FN(inert_mul)(const REAL* in, const REAL* v, REAL* o) { /* :415-429 */
  static const int tri[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};
  const REAL* pos = in + 6;
  REAL mass = in[9];
  REAL c1[3], c2[3];
  FN(cross3)(pos, v + 3, c1);
  FN(cross3)(pos, v, c2);
  for (int i = 0; i < 3; i++) {
    REAL s = in[tri[i][0]] * v[0] + in[tri[i][1]] * v[1] + in[tri[i][2]] * v[2];
    o[i] = s + c1[i];
  }
  for (int i = 0; i < 3; i++) o[3 + i] = mass * v[3 + i] - c2[i];
}
static inline void FN(motion_cross)(const REAL* u, const REAL* v, REAL* o) { /* :455-467 */
  REAL a[3], b[3], c[3];
  FN(cross3)(u, v, a);
  FN(cross3)(u + 3, v, b);
  FN(cross3)(u, v + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static inline void FN(motion_cross_force)(const REAL* v, const REAL* f, REAL* o) { /* :470-482 */
  REAL a[3], b[3], c[3];
  FN(cross3)(v, f, a);
  FN(cross3)(v + 3, f + 3, b);
  FN(cross3)(v, f + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
static inline void FN(make_frame)(const REAL* a_in, REAL* frame) { /* orthogonals/make_frame :485-500 */
  REAL a[3] = {a_in[0], a_in[1], a_in[2]};
  FN(normalize_n)(a, 3);
  REAL b[3] = {0, 0, 0};
  if ((REAL)-0.5 < a[1] && a[1] < (REAL)0.5) b[1] = 1; else b[2] = 1;
  REAL ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  for (int i = 0; i < 3; i++) b[i] = b[i] - a[i] * ab;
  FN(normalize_n)(b, 3);
  int any = (a[0] != 0) || (a[1] != 0) || (a[2] != 0);
  for (int i = 0; i < 3; i++) b[i] = b[i] * (REAL)any;
  REAL c[3];
  FN(cross3)(a, b, c);
  for (int i = 0; i < 3; i++) { frame[i] = a[i]; frame[3 + i] = b[i]; frame[6 + i] = c[i]; }
}

/* ---- dense Cholesky (math.small_cholesky :87-129 / small_cholesky_solve :132-168) ---------- */
static void FN(cholesky)(const REAL* A, REAL* L, int n) {
  for (int i = 0; i < n * n; i++) L[i] = 0;
  if (n > INLINE_CHOL_MAX) {
    /* torch.linalg.cholesky(A + 1e-10*I): LAPACK potrf; textbook column loop here */
    for (int j = 0; j < n; j++) {
      REAL s = A[j * n + j] + (REAL)1e-10;
      for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
      L[j * n + j] = R_SQRT(s);
      for (int i = j + 1; i < n; i++) {
        REAL t = A[i * n + j];
        for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
        L[i * n + j] = t / L[j * n + j];
      }
    }
    return;
  }
  for (int j = 0; j < n; j++) {
    REAL s = A[j * n + j];
    for (int k = 0; k < j; k++) s = s - L[j * n + k] * L[j * n + k];
    L[j * n + j] = R_SQRT(s > (REAL)1e-12 ? s : (REAL)1e-12);
    for (int i = j + 1; i < n; i++) {
      REAL t = A[i * n + j];
      for (int k = 0; k < j; k++) t = t - L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
}
static void FN(cholesky_solve)(const REAL* L, const REAL* x, REAL* out, int n, REAL* y) {
  for (int i = 0; i < n; i++) {
    REAL s = x[i];
    for (int k = 0; k < i; k++) s = s - L[i * n + k] * y[k];
    y[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    REAL s = y[i];
    for (int k = i + 1; k < n; k++) s = s - L[k * n + i] * out[k];
    out[i] = s / L[i * n + i];
  }
}

/* ---- kinematics (smooth.py:34-207, support.local_to_global :99-108) ------------------------- */
static void FN(local_to_global)(const REAL* wpos, const REAL* wquat, const REAL* lpos, const REAL* lquat, REAL* pos, REAL* mat) {
  REAL r[3], q[4];
  FN(rotate)(lpos, wquat, r);
  for (int i = 0; i < 3; i++) pos[i] = wpos[i] + r[i];
  FN(quat_mul)(wquat, lquat, q);
  FN(quat_to_mat)(q, mat);
}

static void FN(kinematics)(const FN(MjoModel) * M, FN(MjoWork) * w, int with_cams) {
  const mjhModelDesc* m = M->d;
  for (int b = 0; b < m->nbody; b++) {
    REAL pos[3] = {M->body_pos[3 * b], M->body_pos[3 * b + 1], M->body_pos[3 * b + 2]};
    REAL quat[4] = {M->body_quat[4 * b], M->body_quat[4 * b + 1], M->body_quat[4 * b + 2], M->body_quat[4 * b + 3]};
    if (b > 0) {
      int p = m->body_parentid[b];
      REAL r[3];
      FN(rotate)(pos, w->xquat + 4 * p, r);
      for (int i = 0; i < 3; i++) pos[i] = w->xpos[3 * p + i] + r[i];
      FN(quat_mul)(w->xquat + 4 * p, quat, quat);
    }
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj;
      int t = m->jnt_type[j], qa = m->jnt_qposadr[j];
      REAL* anchor = w->xanchor + 3 * j;
      REAL* axis = w->xaxis + 3 * j;
      const REAL* jpos = M->jnt_pos + 3 * j;
      const REAL* jaxis = M->jnt_axis + 3 * j;
      if (t == JNT_FREE) {
        for (int i = 0; i < 3; i++) anchor[i] = w->qpos[qa + i];
        axis[0] = 0; axis[1] = 0; axis[2] = 1;
        for (int i = 0; i < 3; i++) pos[i] = w->qpos[qa + i];
        for (int i = 0; i < 4; i++) quat[i] = w->qpos[qa + 3 + i];
        FN(normalize_n)(quat, 4);
        for (int i = 0; i < 4; i++) w->qpos[qa + 3 + i] = quat[i];
      } else {
        REAL r[3];
        FN(rotate)(jpos, quat, r);
        for (int i = 0; i < 3; i++) anchor[i] = r[i] + pos[i];
        FN(rotate)(jaxis, quat, axis);
        if (t == JNT_BALL) {
          REAL ql[4];
          for (int i = 0; i < 4; i++) ql[i] = w->qpos[qa + i];
          FN(normalize_n)(ql, 4);
          for (int i = 0; i < 4; i++) w->qpos[qa + i] = ql[i];
          FN(quat_mul)(quat, ql, quat);
          FN(rotate)(jpos, quat, r);
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - r[i];
        } else if (t == JNT_HINGE) {
          REAL angle = w->qpos[qa] - M->qpos0[qa], ql[4];
          FN(axis_angle_to_quat)(jaxis, angle, ql);
          FN(quat_mul)(quat, ql, quat);
          FN(rotate)(jpos, quat, r);
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - r[i];
        } else { /* slide */
          REAL dq = w->qpos[qa] - M->qpos0[qa];
          for (int i = 0; i < 3; i++) pos[i] = pos[i] + axis[i] * dq;
        }
      }
    }
    for (int i = 0; i < 3; i++) w->xpos[3 * b + i] = pos[i];
    for (int i = 0; i < 4; i++) w->xquat[4 * b + i] = quat[i];
    FN(quat_to_mat)(quat, w->xmat + 9 * b);
  }
  for (int b = 0; b < m->nbody && m->nmocap > 0; b++) { /* mocap bodies take the caller's pose after the tree pass (smooth.py:105-113) */
    int k = m->body_mocapid[b];
    if (k < 0) continue;
    REAL q[4];
    for (int i = 0; i < 3; i++) w->xpos[3 * b + i] = w->mocap_pos[3 * k + i];
    for (int i = 0; i < 4; i++) q[i] = w->mocap_quat[4 * k + i];
    FN(normalize_n)(q, 4);
    for (int i = 0; i < 4; i++) w->xquat[4 * b + i] = q[i];
    FN(quat_to_mat)(q, w->xmat + 9 * b);
  }
  for (int b = 0; b < m->nbody; b++)
    FN(local_to_global)(w->xpos + 3 * b, w->xquat + 4 * b, M->body_ipos + 3 * b, M->body_iquat + 4 * b, w->xipos + 3 * b, w->ximat + 9 * b);
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    FN(local_to_global)(w->xpos + 3 * b, w->xquat + 4 * b, M->geom_pos + 3 * g, M->geom_quat + 4 * g, w->geom_xpos + 3 * g, w->geom_xmat + 9 * g);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    FN(local_to_global)(w->xpos + 3 * b, w->xquat + 4 * b, M->site_pos + 3 * s, M->site_quat + 4 * s, w->site_xpos + 3 * s, w->site_xmat + 9 * s);
  }
  if (with_cams) {
    for (int c = 0; c < m->ncam; c++) { /* smooth.py:139-198 */
      int b = m->cam_bodyid[c], mode = m->cam_mode[c], tgt = m->cam_targetbodyid[c];
      REAL* cp = w->cam_xpos + 3 * c;
      REAL* cm = w->cam_xmat + 9 * c;
      FN(local_to_global)(w->xpos + 3 * b, w->xquat + 4 * b, M->cam_pos + 3 * c, M->cam_quat + 4 * c, cp, cm);
      if (mode == CAM_TRACK) {
        for (int i = 0; i < 3; i++) cp[i] = w->xpos[3 * b + i] + M->cam_pos0[3 * c + i];
        for (int i = 0; i < 9; i++) cm[i] = M->cam_mat0[9 * c + i];
      } else if (mode == CAM_TRACKCOM) {
        /* uses the subtree_com the caller passed in (previous step's), smooth.py:162-166 */
        REAL r[3];
        FN(rotate)(M->cam_pos + 3 * c, w->xquat + 4 * b, r);
        for (int i = 0; i < 3; i++) cp[i] = w->in_subtree_com[3 * b + i] + r[i];
      } else if ((mode == CAM_TARGETBODY || mode == CAM_TARGETBODYCOM) && tgt >= 0) {
        const REAL* tp = (mode == CAM_TARGETBODY) ? w->xpos + 3 * tgt : w->in_subtree_com + 3 * tgt;
        REAL f[3] = {tp[0] - cp[0], tp[1] - cp[1], tp[2] - cp[2]};
        FN(normalize_n)(f, 3);
        REAL up_hint[3] = {0, 0, 1}, right[3], up[3];
        FN(cross3)(f, up_hint, right);
        FN(normalize_n)(right, 3);
        FN(cross3)(right, f, up);
        for (int i = 0; i < 3; i++) { cm[3 * i + 0] = right[i]; cm[3 * i + 1] = up[i]; cm[3 * i + 2] = -f[i]; }
      }
    }
    for (int l = 0; l < m->nlight; l++) { /* :200-204 */
      int b = m->light_bodyid[l];
      REAL r[3];
      FN(rotate)(M->light_pos + 3 * l, w->xquat + 4 * b, r);
      for (int i = 0; i < 3; i++) w->light_xpos[3 * l + i] = w->xpos[3 * b + i] + r[i];
      FN(rotate)(M->light_dir + 3 * l, w->xquat + 4 * b, w->light_xdir + 3 * l);
    }
  }
}

/* ---- com_pos (smooth.py:210-288) ------------------------------------------------------------ */
static void FN(com_pos)(const FN(MjoModel) * M, FN(MjoWork) * w) {
  const mjhModelDesc* m = M->d;
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) {
    for (int i = 0; i < 3; i++) w->sub_pos[3 * b + i] = w->xipos[3 * b + i] * M->body_mass[b];
    w->sub_mass[b] = M->body_mass[b];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int i = 0; i < 3; i++) w->sub_pos[3 * p + i] += w->sub_pos[3 * b + i];
    w->sub_mass[p] += w->sub_mass[b];
  }
  for (int b = 0; b < nb; b++) {
    REAL ms = w->sub_mass[b];
    REAL den = ms > (REAL)MINVAL_CACHED ? ms : (REAL)MINVAL_CACHED;
    for (int i = 0; i < 3; i++) w->subtree_com[3 * b + i] = (ms < (REAL)mjMINVAL) ? w->xipos[3 * b + i] : w->sub_pos[3 * b + i] / den;
  }
  for (int b = 0; b < nb; b++) { /* inert_com :236-243 */
    const REAL* rc = w->subtree_com + 3 * m->body_rootid[b];
    REAL off[3] = {w->xipos[3 * b] - rc[0], w->xipos[3 * b + 1] - rc[1], w->xipos[3 * b + 2] - rc[2]};
    REAL mass = M->body_mass[b];
    const REAL* xi = w->ximat + 9 * b;
    const REAL* in = M->body_inertia + 3 * b;
    REAL h[3][3] = {{0, -off[2], off[1]}, {off[2], 0, -off[0]}, {-off[1], off[0], 0}};
    REAL A[3][3], I[3][3];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) A[i][k] = xi[3 * i + k] * in[k];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        REAL s = 0;
        for (int k = 0; k < 3; k++) s += A[i][k] * xi[3 * j + k];
        REAL hh = 0;
        for (int k = 0; k < 3; k++) hh += h[i][k] * h[j][k];
        I[i][j] = s + hh * mass;
      }
    REAL* ci = w->cinert + 10 * b;
    ci[0] = I[0][0]; ci[1] = I[1][1]; ci[2] = I[2][2]; ci[3] = I[0][1]; ci[4] = I[0][2]; ci[5] = I[1][2];
    ci[6] = off[0] * mass; ci[7] = off[1] * mass; ci[8] = off[2] * mass; ci[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) { /* cdof_fn :250-273 */
    int b = m->jnt_bodyid[j], t = m->jnt_type[j], d = m->jnt_dofadr[j];
    const REAL* rc = w->subtree_com + 3 * m->body_rootid[b];
    REAL off[3] = {rc[0] - w->xanchor[3 * j], rc[1] - w->xanchor[3 * j + 1], rc[2] - w->xanchor[3 * j + 2]};
    if (t == JNT_FREE || t == JNT_BALL) {
      if (t == JNT_FREE) {
        for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) w->cdof[6 * (d + r) + k] = (k == 3 + r) ? 1 : 0;
        d += 3;
      }
      for (int r = 0; r < 3; r++) {
        REAL a[3] = {w->xmat[9 * b + 0 + r], w->xmat[9 * b + 3 + r], w->xmat[9 * b + 6 + r]}; /* xmat.T row r */
        REAL c[3];
        FN(cross3)(a, off, c);
        for (int k = 0; k < 3; k++) { w->cdof[6 * (d + r) + k] = a[k]; w->cdof[6 * (d + r) + 3 + k] = c[k]; }
      }
    } else if (t == JNT_HINGE) {
      REAL c[3];
      FN(cross3)(w->xaxis + 3 * j, off, c);
      for (int k = 0; k < 3; k++) { w->cdof[6 * d + k] = w->xaxis[3 * j + k]; w->cdof[6 * d + 3 + k] = c[k]; }
    } else {
      for (int k = 0; k < 3; k++) { w->cdof[6 * d + k] = 0; w->cdof[6 * d + 3 + k] = w->xaxis[3 * j + k]; }
    }
  }
}

/* ---- crb + make_m + factor_m (smooth.py:291-332, support.make_m :50-80) --------------------- */
static void FN(crb_factor)(const FN(MjoModel) * M, FN(MjoWork) * w) {
  const mjhModelDesc* m = M->d;
  int nb = m->nbody, nv = m->nv;
  for (int i = 0; i < 10 * nb; i++) w->crb[i] = w->cinert[i];
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int i = 0; i < 10; i++) w->crb[10 * p + i] += w->crb[10 * b + i];
  }
  for (int i = 0; i < 10; i++) w->crb[i] = 0;
  for (int d = 0; d < nv; d++) FN(inert_mul)(w->crb + 10 * m->dof_bodyid[d], w->cdof + 6 * d, w->crb_cdof + 6 * d);
  for (int i = 0; i < nv * nv; i++) w->qM[i] = 0;
  for (int i = 0; i < nv; i++) {
    int j = i;
    while (j >= 0) { /* (i, j) with j an ancestor-or-self dof of i */
      REAL s = 0;
      for (int k = 0; k < 6; k++) s += w->crb_cdof[6 * i + k] * w->cdof[6 * j + k];
      if (i == j) s = s + M->dof_armature[i];
      w->qM[i * nv + j] = s;
      if (i != j) w->qM[j * nv + i] = s;
      j = m->dof_parentid[j];
    }
  }
  if (m->ntendon > 0) { /* smooth.tendon_armature :500-522: qM += J^T diag(armature) J, ahead of factor_m */
    int any = 0;
    for (int t = 0; t < m->ntendon; t++) any |= M->tendon_armature[t] != 0;
    if (any)
      for (int i = 0; i < nv; i++)
        for (int j = 0; j < nv; j++) {
          REAL s = 0;
          for (int t = 0; t < m->ntendon; t++) s += w->ten_J[t * nv + i] * (w->ten_J[t * nv + j] * M->tendon_armature[t]);
          w->qM[i * nv + j] = w->qM[i * nv + j] + s;
        }
  }
  FN(cholesky)(w->qM, w->qLD, nv);
}

/* ---- collision primitives (collision_primitive.py, math.py:506-569) -------------------------- */
static void FN(plane_sphere_)(const REAL* n, const REAL* ppos, const REAL* spos, REAL r, REAL* dist, REAL* pos) {
  REAL d[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  *dist = FN(dot3)(d, n) - r;
  for (int i = 0; i < 3; i++) pos[i] = spos[i] - n[i] * (r + (REAL)0.5 * (*dist));
}
/* `hint_n`: normal of the outputs under test, or NULL.  When the two centres coincide to rounding noise (adjacent capsules of a
 * leg share their joint point: the closest points of the two segments are the same point) the reference's normal is that noise
 * normalised -- any unit vector is an admissible outcome and two correct implementations disagree.  The oracle then adopts the
 * hinted normal (if it is a unit vector) and counts the pair as a non-natural tie outcome when it differs from its own. */
static void FN(sphere_sphere_)(const REAL* p1, REAL r1, const REAL* p2, REAL r2, REAL* dist, REAL* pos, REAL* n, const REAL* hint_n, int* adopted) {
  for (int i = 0; i < 3; i++) n[i] = p2[i] - p1[i];
  REAL d = FN(normalize_n)(n, 3);
  if (d == 0) { n[0] = 1; n[1] = 0; n[2] = 0; }
  if (hint_n && d <= (sizeof(REAL) == 4 ? (REAL)2e-5 : (REAL)1e-11) * (r1 + r2)) {
    REAL hn = R_SQRT((hint_n[0] * hint_n[0] + hint_n[1] * hint_n[1]) + hint_n[2] * hint_n[2]);
    REAL dev = 0;
    for (int i = 0; i < 3; i++) { REAL e = R_FABS(hint_n[i] - n[i]); if (e > dev) dev = e; }
    /* only a materially different direction is adopted: an agreeing hint leaves the natural result bit for bit */
    if (R_FABS(hn - 1) < (sizeof(REAL) == 4 ? (REAL)1e-4 : (REAL)1e-9) && dev > (sizeof(REAL) == 4 ? (REAL)1e-3 : (REAL)1e-6)) {
      if (adopted) (*adopted)++;
      n[0] = hint_n[0]; n[1] = hint_n[1]; n[2] = hint_n[2];
      d = 0;
    }
  }
  d = d - (r1 + r2);
  for (int i = 0; i < 3; i++) pos[i] = p1[i] + n[i] * (r1 + d * (REAL)0.5);
  *dist = d;
}
static void FN(closest_segment_point)(const REAL* a, const REAL* b, const REAL* pt, REAL* o) { /* math.py:506-510 */
  REAL ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  REAL pa[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
  REAL t = FN(dot3)(pa, ab) / (FN(dot3)(ab, ab) + (REAL)1e-6);
  t = t < 0 ? 0 : (t > 1 ? 1 : t);
  for (int i = 0; i < 3; i++) o[i] = a[i] + t * ab[i];
}
static void FN(closest_segment_to_segment)(const REAL* a0, const REAL* a1, const REAL* b0, const REAL* b1, REAL* best_a, REAL* best_b) { /* :523-569 */
  REAL dir_a[3], dir_b[3];
  for (int i = 0; i < 3; i++) { dir_a[i] = a1[i] - a0[i]; dir_b[i] = b1[i] - b0[i]; }
  REAL len_a = FN(normalize_n)(dir_a, 3), len_b = FN(normalize_n)(dir_b, 3);
  REAL hla = len_a * (REAL)0.5, hlb = len_b * (REAL)0.5;
  REAL a_mid[3], b_mid[3], trans[3];
  for (int i = 0; i < 3; i++) { a_mid[i] = a0[i] + dir_a[i] * hla; b_mid[i] = b0[i] + dir_b[i] * hlb; trans[i] = a_mid[i] - b_mid[i]; }
  REAL dadb = FN(dot3)(dir_a, dir_b), dat = FN(dot3)(dir_a, trans), dbt = FN(dot3)(dir_b, trans);
  REAL denom = 1 - dadb * dadb;
  REAL ota = (-dat + dadb * dbt) / (denom + (REAL)1e-6);
  REAL otb = dbt + ota * dadb;
  REAL ta = ota < -hla ? -hla : (ota > hla ? hla : ota);
  REAL tb = otb < -hlb ? -hlb : (otb > hlb ? hlb : otb);
  for (int i = 0; i < 3; i++) { best_a[i] = a_mid[i] + dir_a[i] * ta; best_b[i] = b_mid[i] + dir_b[i] * tb; }
  REAL new_a[3], new_b[3];
  FN(closest_segment_point)(a0, a1, best_b, new_a);
  FN(closest_segment_point)(b0, b1, best_a, new_b);
  REAL d1 = 0, d2 = 0;
  for (int i = 0; i < 3; i++) { REAL e = best_b[i] - new_a[i]; d1 += e * e; }
  for (int i = 0; i < 3; i++) { REAL e = best_a[i] - new_b[i]; d2 += e * e; }
  if (d1 < d2) { for (int i = 0; i < 3; i++) best_a[i] = new_a[i]; }
  else { for (int i = 0; i < 3; i++) best_b[i] = new_b[i]; }
}

/* ---- convex narrow phase (collision_convex.py) -------------------------------------------------- */
#ifndef MJO_CVX_COMMON_
#define MJO_CVX_COMMON_
#define MJO_MAXK 20              /* mesh.py:32 _MAX_HULL_FACE_VERTICES */
#define MJO_MAXP (4 * MJO_MAXK) /* _clip output: 2 points per subject edge + 2 per clipping edge */
#endif
/* `1e-6 * (x == 0.0)`: bool tensor times Python float is float32 in torch; up-cast on use */
#define EPS_F32 ((REAL)(float)1e-6)

typedef struct FN(Cvx) {
  int nvert, nface, nfv, nedge;
  const REAL *vert, *norm;
  const int *face, *edge;
} FN(Cvx);
static FN(Cvx) FN(cvx_of)(const FN(MjoModel) * M, int geom) {
  const mjhModelDesc* m = M->d;
  int c = m->geom_convexid[geom];
  FN(Cvx) r;
  r.nvert = m->convex_nvert[c]; r.nface = m->convex_nface[c]; r.nfv = m->convex_nfv[c]; r.nedge = m->convex_nedge[c];
  r.vert = M->convex_vert + 3 * m->convex_vertadr[c];
  r.norm = M->convex_facenormal + 3 * m->convex_normadr[c];
  r.face = m->convex_face + m->convex_faceadr[c];
  r.edge = m->convex_edge + 2 * m->convex_edgeadr[c];
  return r;
}
/* vertex id k of face f when faces are padded to K >= nfv by repeating the last id (F.pad replicate, :810-816) */
static inline int FN(cvx_fv)(const FN(Cvx) * c, int f, int k) { return c->face[f * c->nfv + (k < c->nfv ? k : c->nfv - 1)]; }
static inline void FN(mat_t_vec)(const REAL* R, const REAL* v, REAL* o) { /* (mat.T * v).sum(-1) */
  for (int i = 0; i < 3; i++) o[i] = R[0 + i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}
static inline void FN(mat_vec)(const REAL* R, const REAL* v, REAL* o) { /* (mat * v).sum(-1) == (v[:,None] * mat.T).sum(-2) */
  for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
static int FN(argmax_)(const REAL* x, int n) { int b = 0; for (int i = 1; i < n; i++) if (x[i] > x[b]) b = i; return b; }
static int FN(argmin_)(const REAL* x, int n) { int b = 0; for (int i = 1; i < n; i++) if (x[i] < x[b]) b = i; return b; }
#undef TIE_EPS
#ifdef REAL_IS_FLOAT
#define TIE_EPS ((REAL)2e-5)
#else
#define TIE_EPS ((REAL)1e-11)
#endif
/* argmax (sgn = +1) / argmin (sgn = -1) with tie events.  `scale`: magnitude of the terms the scores were summed from;
   `key` (optional, n x 3): candidates with bit-identical key rows are the same outcome and count once. */
static int FN(pick)(FN(MjoWork) * w, const REAL* x, int n, int sgn, REAL scale, const REAL* key, int keymod) {
  int b = sgn > 0 ? FN(argmax_)(x, n) : FN(argmin_)(x, n);
  if (!w || !w->tie_on) return b;
  if (R_FABS(x[b]) >= (REAL)1e5) return b; /* best is a masked-out sentinel (-1e6, +-1e12): nothing is in contact */
  REAL tol = TIE_EPS * scale;
  int cand[64], nc = 0;
  cand[nc++] = b;
  for (int i = 0; i < n && nc < 64; i++) {
    if (i == b || !(R_FABS(x[i] - x[b]) <= tol)) continue;
    int dup = 0;
    if (key) for (int c = 0; c < nc && !dup; c++) {
      const REAL *ka = key + 3 * (cand[c] % keymod), *kb = key + 3 * (i % keymod);
      dup = ka[0] == kb[0] && ka[1] == kb[1] && ka[2] == kb[2];
    }
    if (!dup) cand[nc++] = i;
  }
  if (nc == 1) return b;
  if (w->tie_on == 2) { int ev = w->stage_tie_n++; return cand[(ev < 32 && ((w->stage_tie_flip >> ev) & 1u)) ? 1 : 0]; }
  int e = w->tie_n++;
  if (e >= MJO_MAX_TIE) return b;
  w->tie_count[e] = nc;
  int d = w->tie_digit[e];
  return cand[d < nc ? d : 0];
}

static void FN(closest_segment_point_plane)(const REAL* a, const REAL* b, const REAL* p0, const REAL* n, REAL* o) { /* :39-63 */
  REAL ba[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  REAL d = FN(dot3)(p0, n);
  REAL denom = FN(dot3)(n, ba);
  REAL t = (d - FN(dot3)(n, a)) / (denom + (denom == 0 ? EPS_F32 : (REAL)0));
  t = t < 0 ? 0 : (t > 1 ? 1 : t);
  for (int i = 0; i < 3; i++) o[i] = a[i] + t * ba[i];
}
static void FN(project_pt_onto_plane)(const REAL* pt, const REAL* plane_pt, const REAL* n, REAL* o) { /* :238-241 */
  REAL d[3] = {pt[0] - plane_pt[0], pt[1] - plane_pt[1], pt[2] - plane_pt[2]};
  REAL dist = FN(dot3)(d, n);
  for (int i = 0; i < 3; i++) o[i] = pt[i] - dist * n[i];
}
/* _manifold_points :183-235 (hard selection branch) */
static void FN(manifold_points)(FN(MjoWork) * w, const REAL (*poly)[3], const unsigned char* mask, int n, const REAL* norm, int* out) {
  REAL dm[MJO_MAXP > 64 ? MJO_MAXP : 64], sc[2 * (MJO_MAXP > 64 ? MJO_MAXP : 64)];
  REAL* dmask = dm; REAL* score = sc;
  REAL *hd = NULL, *hs = NULL;
  if (n > (MJO_MAXP > 64 ? MJO_MAXP : 64)) { hd = (REAL*)malloc(sizeof(REAL) * n); hs = (REAL*)malloc(sizeof(REAL) * 2 * n); dmask = hd; score = hs; }
  for (int i = 0; i < n; i++) dmask[i] = mask[i] ? (REAL)0 : (REAL)-1e6;
  int ai = FN(argmax_)(dmask, n);
  const REAL* a = poly[ai];
  REAL scale = 0;
  for (int i = 0; i < n; i++) {
    REAL e0 = a[0] - poly[i][0], e1 = a[1] - poly[i][1], e2 = a[2] - poly[i][2];
    score[i] = (e0 * e0 + e1 * e1 + e2 * e2) + dmask[i];
    if (mask[i] && score[i] > scale) scale = score[i];
  }
  int bi = FN(pick)(w, score, n, +1, scale, &poly[0][0], n);
  const REAL* b = poly[bi];
  REAL amb[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, ab[3];
  FN(cross3)(norm, amb, ab);
  scale = 0;
  for (int i = 0; i < n; i++) {
    REAL ap[3] = {a[0] - poly[i][0], a[1] - poly[i][1], a[2] - poly[i][2]};
    score[i] = R_FABS(FN(dot3)(ap, ab)) + dmask[i];
    REAL mag = R_FABS(ap[0] * ab[0]) + R_FABS(ap[1] * ab[1]) + R_FABS(ap[2] * ab[2]);
    if (mask[i] && mag > scale) scale = mag;
  }
  int ci = FN(pick)(w, score, n, +1, scale, &poly[0][0], n);
  const REAL* c = poly[ci];
  REAL amc[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]}, bmc[3] = {b[0] - c[0], b[1] - c[1], b[2] - c[2]}, ac[3], bc[3];
  FN(cross3)(norm, amc, ac);
  FN(cross3)(norm, bmc, bc);
  scale = 0;
  for (int i = 0; i < n; i++) {
    REAL ap[3] = {a[0] - poly[i][0], a[1] - poly[i][1], a[2] - poly[i][2]};
    REAL bp[3] = {b[0] - poly[i][0], b[1] - poly[i][1], b[2] - poly[i][2]};
    score[i] = R_FABS(FN(dot3)(bp, bc)) + dmask[i];
    score[n + i] = R_FABS(FN(dot3)(ap, ac)) + dmask[i];
    REAL m1 = R_FABS(bp[0] * bc[0]) + R_FABS(bp[1] * bc[1]) + R_FABS(bp[2] * bc[2]);
    REAL m2 = R_FABS(ap[0] * ac[0]) + R_FABS(ap[1] * ac[1]) + R_FABS(ap[2] * ac[2]);
    if (mask[i] && m1 > scale) scale = m1;
    if (mask[i] && m2 > scale) scale = m2;
  }
  int di = FN(pick)(w, score, 2 * n, +1, scale, &poly[0][0], n) % n;
  out[0] = ai; out[1] = bi; out[2] = ci; out[3] = di;
  free(hd); free(hs);
}
/* _clip_edge_to_planes :265-327; returns the mask */
static int FN(clip_edge_to_planes)(const REAL* p0, const REAL* p1, const REAL (*plane_pts)[3], const REAL (*plane_n)[3], int K, REAL* o0, REAL* o1) {
  unsigned char f0[MJO_MAXK], f1[MJO_MAXK];
  REAL cand[MJO_MAXK][3], d0[MJO_MAXK], d1[MJO_MAXK];
  int any_both = 0;
  REAL p10[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, p01[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
  for (int k = 0; k < K; k++) {
    REAL e0[3] = {p0[0] - plane_pts[k][0], p0[1] - plane_pts[k][1], p0[2] - plane_pts[k][2]};
    REAL e1[3] = {p1[0] - plane_pts[k][0], p1[1] - plane_pts[k][1], p1[2] - plane_pts[k][2]};
    f0[k] = FN(dot3)(e0, plane_n[k]) > (REAL)1e-6;
    f1[k] = FN(dot3)(e1, plane_n[k]) > (REAL)1e-6;
    any_both |= (f0[k] && f1[k]);
    FN(closest_segment_point_plane)(p0, p1, plane_pts[k], plane_n[k], cand[k]);
    REAL q0[3], q1[3];
    for (int i = 0; i < 3; i++) { q0[i] = (f0[k] ? cand[k][i] : p0[i]) - p0[i]; q1[i] = (f1[k] ? cand[k][i] : p1[i]) - p1[i]; }
    d0[k] = FN(dot3)(q0, p10);
    d1[k] = FN(dot3)(q1, p01);
  }
  int i0 = FN(argmax_)(d0, K), i1 = FN(argmax_)(d1, K);
  REAL n0[3], n1[3];
  for (int i = 0; i < 3; i++) { n0[i] = f0[i0] ? cand[i0][i] : p0[i]; n1[i] = f1[i1] ? cand[i1][i] : p1[i]; }
  int mask = !any_both;
  for (int i = 0; i < 3; i++) { o0[i] = mask ? n0[i] : p0[i]; o1[i] = mask ? n1[i] : p1[i]; }
  REAL dd[3] = {o0[0] - o1[0], o0[1] - o1[1], o0[2] - o1[2]};
  if (FN(dot3)(p01, dd) < 0) mask = 0;
  return mask;
}
/* _create_contact_manifold :395-449 with _clip :330-392 inlined.  Faces are K-gons (padded). */
static void FN(create_contact_manifold)(FN(MjoWork) * w, const REAL (*clip_poly)[3], const REAL (*subj_poly)[3], int K, const REAL* clip_n, const REAL* subj_n,
                                        const REAL* sep_axis, REAL* dist, REAL (*pos)[3], REAL* normal) {
  REAL cp0[MJO_MAXK][3], cpn[MJO_MAXK][3], sp0[MJO_MAXK][3], spn[MJO_MAXK][3];
  REAL inc[MJO_MAXP][3], ref[MJO_MAXP][3];
  unsigned char mask[MJO_MAXP];
  for (int k = 0; k < K; k++) {
    int km = (k + K - 1) % K;
    REAL e[3];
    for (int i = 0; i < 3; i++) { cp0[k][i] = clip_poly[km][i]; e[i] = clip_poly[k][i] - clip_poly[km][i]; }
    FN(cross3)(e, clip_n, cpn[k]);
    for (int i = 0; i < 3; i++) { sp0[k][i] = subj_poly[km][i]; e[i] = subj_poly[k][i] - subj_poly[km][i]; }
    FN(cross3)(e, subj_n, spn[k]);
  }
  /* subject edges against the clipping polygon's side planes */
  for (int k = 0; k < K; k++) {
    int mk = FN(clip_edge_to_planes)(sp0[k], subj_poly[k], cp0, cpn, K, inc[2 * k], inc[2 * k + 1]);
    mask[2 * k] = mask[2 * k + 1] = (unsigned char)mk;
  }
  /* clipping polygon projected onto the subject plane along the clipping normal (:249-257) */
  REAL d = FN(dot3)(subj_poly[0], subj_n);
  REAL denom = FN(dot3)(clip_n, subj_n);
  REAL den = denom + (denom == 0 ? EPS_F32 : (REAL)0);
  REAL c0s[MJO_MAXK][3], c1s[MJO_MAXK][3];
  for (int k = 0; k < K; k++) {
    REAL t0 = (d - FN(dot3)(cp0[k], subj_n)) / den, t1 = (d - FN(dot3)(clip_poly[k], subj_n)) / den;
    for (int i = 0; i < 3; i++) { c0s[k][i] = cp0[k][i] + t0 * clip_n[i]; c1s[k][i] = clip_poly[k][i] + t1 * clip_n[i]; }
  }
  for (int k = 0; k < K; k++) {
    int mk = FN(clip_edge_to_planes)(c0s[k], c1s[k], sp0, spn, K, inc[2 * K + 2 * k], inc[2 * K + 2 * k + 1]);
    mask[2 * K + 2 * k] = mask[2 * K + 2 * k + 1] = (unsigned char)mk;
  }
  int P = 4 * K;
  REAL nn[3] = {clip_n[0], clip_n[1], clip_n[2]}, neg[3] = {-clip_n[0], -clip_n[1], -clip_n[2]};
  FN(normalize_n)(nn, 3);
  for (int p = 0; p < P; p++) {
    FN(project_pt_onto_plane)(inc[p], clip_poly[0], nn, ref[p]);
    REAL e[3] = {inc[p][0] - clip_poly[0][0], inc[p][1] - clip_poly[0][1], inc[p][2] - clip_poly[0][2]};
    mask[p] = mask[p] && (FN(dot3)(e, neg) > (REAL)1e-6);
  }
  int best[4];
  FN(manifold_points)(w, (const REAL(*)[3])ref, mask, P, clip_n, best);
  for (int q = 0; q < 4; q++) {
    int b = best[q];
    REAL pd[3] = {inc[b][0] - ref[b][0], inc[b][1] - ref[b][1], inc[b][2] - ref[b][2]};
    REAL pen = FN(dot3)(pd, neg);
    dist[q] = mask[b] ? -pen : (REAL)1;
    for (int i = 0; i < 3; i++) pos[q][i] = ref[b][i];
  }
  for (int i = 0; i < 3; i++) normal[i] = -sep_axis[i];
}

/* plane_convex :604-623 */
static void FN(plane_convex_)(FN(MjoWork) * w, const REAL* ppos, const REAL* pmat, const REAL* cpos, const REAL* cmat, const FN(Cvx) * cv, REAL* dist, REAL (*pos)[3], REAL (*frame)[9]) {
  REAL rel[3] = {ppos[0] - cpos[0], ppos[1] - cpos[1], ppos[2] - cpos[2]}, plane_pos[3], nw[3] = {pmat[2], pmat[5], pmat[8]}, n[3];
  FN(mat_t_vec)(cmat, rel, plane_pos);
  FN(mat_t_vec)(cmat, nw, n);
  int V = cv->nvert;
  REAL* support = (REAL*)malloc(sizeof(REAL) * V);
  unsigned char* mask = (unsigned char*)malloc(V);
  for (int v = 0; v < V; v++) {
    REAL e[3] = {plane_pos[0] - cv->vert[3 * v], plane_pos[1] - cv->vert[3 * v + 1], plane_pos[2] - cv->vert[3 * v + 2]};
    support[v] = FN(dot3)(e, n);
    mask[v] = support[v] > 0;
  }
  int idx[4];
  FN(manifold_points)(w, (const REAL(*)[3])cv->vert, mask, V, n, idx);
  for (int q = 0; q < 4; q++) {
    REAL r[3];
    FN(mat_vec)(cmat, cv->vert + 3 * idx[q], r);
    for (int i = 0; i < 3; i++) pos[q][i] = cpos[i] + r[i];
    FN(make_frame)(nw, frame[q]);
    int cnt = 0;
    for (int j = 0; j <= q; j++) cnt += idx[j] == idx[q];
    dist[q] = cnt == 1 ? -support[idx[q]] : (REAL)1;
  }
  free(support); free(mask);
}

/* face of the convex with the largest negative support (:644-657, :732-744) */
static int FN(best_face_)(FN(MjoWork) * w, REAL* support, int F, REAL scale) {
  for (int f = 0; f < F; f++) {
    if (support[f] < 0 && -support[f] > scale) scale = -support[f];
    if (support[f] >= 0) support[f] = (REAL)-1e12;
  }
  return FN(pick)(w, support, F, +1, scale, NULL, 1);
}

/* sphere_convex :626-699 */
static void FN(sphere_convex_)(FN(MjoWork) * w, const REAL* spos, REAL r, const REAL* cpos, const REAL* cmat, const FN(Cvx) * cv, REAL* dist, REAL* pos, REAL* frame) {
  REAL rel[3] = {spos[0] - cpos[0], spos[1] - cpos[1], spos[2] - cpos[2]}, sp[3];
  FN(mat_t_vec)(cmat, rel, sp);
  int F = cv->nface, K = cv->nfv;
  REAL* support = (REAL*)malloc(sizeof(REAL) * F);
  for (int f = 0; f < F; f++) {
    const REAL *nf = cv->norm + 3 * f, *v0 = cv->vert + 3 * FN(cvx_fv)(cv, f, 0);
    REAL e[3];
    for (int i = 0; i < 3; i++) e[i] = (sp[i] - nf[i] * r) - v0[i];
    support[f] = FN(dot3)(e, nf);
  }
  int bf = FN(best_face_)(w, support, F, r);
  free(support);
  const REAL* normal = cv->norm + 3 * bf;
  REAL face[MJO_MAXK][3] = {{0}};
  for (int k = 0; k < K; k++) for (int i = 0; i < 3; i++) face[k][i] = cv->vert[3 * FN(cvx_fv)(cv, bf, k) + i];
  REAL pt[3];
  FN(project_pt_onto_plane)(sp, face[0], normal, pt);
  REAL ed[MJO_MAXK], edm[MJO_MAXK];
  int inside = 1;
  for (int k = 0; k < K; k++) {
    int km = (k + K - 1) % K;
    REAL e[3] = {face[k][0] - face[km][0], face[k][1] - face[km][1], face[k][2] - face[km][2]}, en[3];
    FN(cross3)(e, normal, en);
    REAL q[3] = {pt[0] - face[km][0], pt[1] - face[km][1], pt[2] - face[km][2]};
    ed[k] = FN(dot3)(q, en);
    if (!(ed[k] <= 0)) inside = 0;
    int degenerate = en[0] == 0 && en[1] == 0 && en[2] == 0;
    edm[k] = (degenerate || ed[k] < 0) ? (REAL)1e12 : ed[k];
  }
  REAL escale = 0;
  for (int k = 0; k < K; k++) if (edm[k] < (REAL)1e11 && edm[k] > escale) escale = edm[k];
  int ei = FN(pick)(w, edm, K, -1, escale + r * r, NULL, 1);
  if (!inside) {
    REAL ept[3];
    FN(closest_segment_point)(face[(ei + K - 1) % K], face[ei], pt, ept);
    for (int i = 0; i < 3; i++) pt[i] = ept[i];
  }
  REAL n[3] = {pt[0] - sp[0], pt[1] - sp[1], pt[2] - sp[2]};
  REAL d = FN(normalize_n)(n, 3);
  REAL lp[3], nw[3], pw[3];
  for (int i = 0; i < 3; i++) { REAL spt = sp[i] + n[i] * r; lp[i] = (pt[i] + spt) * (REAL)0.5; }
  *dist = d - r;
  FN(mat_vec)(cmat, n, nw);
  FN(mat_vec)(cmat, lp, pw);
  for (int i = 0; i < 3; i++) pos[i] = pw[i] + cpos[i];
  FN(make_frame)(nw, frame);
}

/* capsule_convex :702-802 */
static void FN(capsule_convex_)(FN(MjoWork) * w, const REAL* kpos, const REAL* kmat, REAL r, REAL halflen, const REAL* cpos, const REAL* cmat, const FN(Cvx) * cv,
                                REAL* dist, REAL (*pos)[3], REAL (*frame)[9]) {
  REAL rel[3] = {kpos[0] - cpos[0], kpos[1] - cpos[1], kpos[2] - cpos[2]}, cp[3], axw[3] = {kmat[2], kmat[5], kmat[8]}, axis[3];
  FN(mat_t_vec)(cmat, rel, cp);
  FN(mat_t_vec)(cmat, axw, axis);
  REAL pts[2][3];
  for (int i = 0; i < 3; i++) { REAL sg = axis[i] * halflen; pts[0][i] = cp[i] - sg; pts[1][i] = cp[i] + sg; }
  int F = cv->nface, K = cv->nfv;
  REAL* support = (REAL*)malloc(sizeof(REAL) * F);
  int has_support = 1;
  for (int f = 0; f < F; f++) {
    const REAL *nf = cv->norm + 3 * f, *v0 = cv->vert + 3 * FN(cvx_fv)(cv, f, 0);
    REAL s2[2];
    for (int q = 0; q < 2; q++) {
      REAL e[3];
      for (int i = 0; i < 3; i++) e[i] = (pts[q][i] - nf[i] * r) - v0[i];
      s2[q] = FN(dot3)(e, nf);
    }
    support[f] = s2[1] < s2[0] ? s2[1] : s2[0];
    if (!(support[f] < 0)) has_support = 0;
  }
  int bf = FN(best_face_)(w, support, F, r + halflen);
  free(support);
  const REAL* normal = cv->norm + 3 * bf;
  REAL face[MJO_MAXK][3], ep0[MJO_MAXK][3], en[MJO_MAXK][3];
  for (int k = 0; k < K; k++) for (int i = 0; i < 3; i++) face[k][i] = cv->vert[3 * FN(cvx_fv)(cv, bf, k) + i];
  for (int k = 0; k < K; k++) {
    int km = (k + K - 1) % K;
    REAL e[3];
    for (int i = 0; i < 3; i++) { ep0[k][i] = face[km][i]; e[i] = face[k][i] - face[km][i]; }
    FN(cross3)(e, normal, en[k]);
  }
  REAL cl[2][3];
  int mask = FN(clip_edge_to_planes)(pts[0], pts[1], (const REAL(*)[3])ep0, (const REAL(*)[3])en, K, cl[0], cl[1]);
  REAL lpos[2][3], lnorm[2][3], pen[2];
  for (int q = 0; q < 2; q++) {
    REAL fp[3];
    for (int i = 0; i < 3; i++) cl[q][i] = cl[q][i] - normal[i] * r;
    FN(project_pt_onto_plane)(cl[q], face[0], normal, fp);
    REAL e[3];
    for (int i = 0; i < 3; i++) { lpos[q][i] = (cl[q][i] + fp[i]) * (REAL)0.5; lnorm[q][i] = normal[i]; e[i] = fp[i] - cl[q][i]; }
    pen[q] = (mask && has_support) ? FN(dot3)(e, normal) : (REAL)-1;
  }
  /* potential edge contact */
  REAL ed[MJO_MAXK], ecl[MJO_MAXK][3], ccl[MJO_MAXK][3];
  for (int k = 0; k < K; k++) {
    FN(closest_segment_to_segment)(ep0[k], face[k], pts[0], pts[1], ecl[k], ccl[k]);
    REAL e0 = ecl[k][0] - ccl[k][0], e1 = ecl[k][1] - ccl[k][1], e2 = ecl[k][2] - ccl[k][2];
    ed[k] = e0 * e0 + e1 * e1 + e2 * e2;
  }
  REAL escale = 0;
  for (int k = 0; k < K; k++) if (ed[k] > escale) escale = ed[k];
  int ei = FN(pick)(w, ed, K, -1, escale, NULL, 1);
  REAL eax[3] = {ccl[ei][0] - ecl[ei][0], ccl[ei][1] - ecl[ei][1], ccl[ei][2] - ecl[ei][2]};
  REAL edist = FN(normalize_n)(eax, 3);
  REAL epen = r - edist;
  if (epen > 0) {
    for (int i = 0; i < 3; i++) { lpos[0][i] = (ecl[ei][i] + (ccl[ei][i] - eax[i] * r)) * (REAL)0.5; lnorm[0][i] = eax[i]; }
    pen[0] = epen;
  }
  for (int q = 0; q < 2; q++) {
    REAL nl[3] = {-lnorm[q][0], -lnorm[q][1], -lnorm[q][2]}, pw[3], nw[3];
    FN(mat_vec)(cmat, lpos[q], pw);
    FN(mat_vec)(cmat, nl, nw);
    for (int i = 0; i < 3; i++) pos[q][i] = cpos[i] + pw[i];
    dist[q] = -pen[q];
    FN(make_frame)(nw, frame[q]);
  }
}

/* separating axis number a of _sat_hull_hull (:495-503): normals of hull 1, of hull 2, then normalised edge x edge */
static void FN(sat_axis)(int a, const FN(Cvx) * cv1, const FN(Cvx) * cv2, const REAL (*v1)[3], const REAL (*n1)[3], REAL* axis) {
  int F1 = cv1->nface, F2 = cv2->nface, E1 = cv1->nedge;
  const REAL(*v2)[3] = (const REAL(*)[3])cv2->vert;
  if (a < F1) { for (int i = 0; i < 3; i++) axis[i] = n1[a][i]; return; }
  if (a < F1 + F2) { for (int i = 0; i < 3; i++) axis[i] = cv2->norm[3 * (a - F1) + i]; return; }
  int e = a - F1 - F2, i1 = e % E1, j2 = e / E1;
  const REAL *a0 = v1[cv1->edge[2 * i1]], *a1 = v1[cv1->edge[2 * i1 + 1]], *b0 = v2[cv2->edge[2 * j2]], *b1 = v2[cv2->edge[2 * j2 + 1]];
  REAL da[3] = {a0[0] - a1[0], a0[1] - a1[1], a0[2] - a1[2]}, db[3] = {b0[0] - b1[0], b0[1] - b1[1], b0[2] - b1[2]};
  FN(cross3)(da, db, axis);
  FN(normalize_n)(axis, 3);
}
/* convex_convex :805-856 with _sat_hull_hull :464-601 */
static void FN(convex_convex_)(FN(MjoWork) * w, const REAL* pos1, const REAL* mat1, const FN(Cvx) * cv1, const REAL* pos2, const REAL* mat2, const FN(Cvx) * cv2,
                               REAL* dist, REAL (*pos)[3], REAL (*frame)[9]) {
  int K = cv1->nfv > cv2->nfv ? cv1->nfv : cv2->nfv;
  int swapped = cv1->nvert > cv2->nvert;
  if (swapped) { const REAL* t; const FN(Cvx) * c; t = pos1; pos1 = pos2; pos2 = t; t = mat1; mat1 = mat2; mat2 = t; c = cv1; cv1 = cv2; cv2 = c; }
  int V1 = cv1->nvert, V2 = cv2->nvert, F1 = cv1->nface, F2 = cv2->nface, E1 = cv1->nedge, E2 = cv2->nedge;
  REAL rel[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]}, tlp[3], tlm[9];
  FN(mat_t_vec)(mat2, rel, tlp);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tlm[3 * i + j] = mat2[0 + i] * mat1[0 + j] + mat2[3 + i] * mat1[3 + j] + mat2[6 + i] * mat1[6 + j];
  REAL(*v1)[3] = (REAL(*)[3])malloc(sizeof(REAL) * 3 * V1);
  REAL(*n1)[3] = (REAL(*)[3])malloc(sizeof(REAL) * 3 * F1);
  for (int v = 0; v < V1; v++) { REAL t[3]; FN(mat_vec)(tlm, cv1->vert + 3 * v, t); for (int i = 0; i < 3; i++) v1[v][i] = tlp[i] + t[i]; }
  for (int f = 0; f < F1; f++) FN(mat_vec)(tlm, cv1->norm + 3 * f, n1[f]);
  const REAL(*v2)[3] = (const REAL(*)[3])cv2->vert;
  const REAL(*n2)[3] = (const REAL(*)[3])cv2->norm;
  /* separating axes: face normals of 1, of 2, then edge x edge (index j * E1 + i) */
  int NA = F1 + F2 + E1 * E2;
  REAL* sup = (REAL*)malloc(sizeof(REAL) * NA);
  signed char* sgn = (signed char*)malloc(NA);
  REAL sscale = 0;
  for (int a = 0; a < NA; a++) {
    REAL axis[3];
    FN(sat_axis)(a, cv1, cv2, (const REAL(*)[3])v1, (const REAL(*)[3])n1, axis);
    REAL amax = 0, amin = 0, bmax = 0, bmin = 0;
    for (int v = 0; v < V1; v++) { REAL sv = FN(dot3)(axis, v1[v]); if (v == 0 || sv > amax) amax = sv; if (v == 0 || sv < amin) amin = sv; }
    for (int v = 0; v < V2; v++) { REAL sv = FN(dot3)(axis, v2[v]); if (v == 0 || sv > bmax) bmax = sv; if (v == 0 || sv < bmin) bmin = sv; }
    REAL d1 = amax - bmin, d2 = bmax - amin;
    sgn[a] = d1 > d2 ? -1 : 1;
    REAL d = d1 < d2 ? d1 : d2;   /* torch.minimum */
    if (axis[0] == 0 && axis[1] == 0 && axis[2] == 0) d = (REAL)1e6;
    else { REAL mg = R_FABS(amax) + R_FABS(amin) + R_FABS(bmax) + R_FABS(bmin); if (mg > sscale) sscale = mg; }
    sup[a] = d;
  }
  int best = FN(pick)(w, sup, NA, -1, sscale, NULL, 1);
  int best_sign = sgn[best];
  REAL best_axis[3];
  FN(sat_axis)(best, cv1, cv2, (const REAL(*)[3])v1, (const REAL(*)[3])n1, best_axis);
  free(sup); free(sgn);
  int is_edge = best >= F1 + F2;
  REAL* fa = (REAL*)malloc(sizeof(REAL) * (F1 + F2));
  REAL* fb = fa + F1;
  for (int f = 0; f < F1; f++) fa[f] = FN(dot3)(best_axis, n1[f]);
  for (int f = 0; f < F2; f++) fb[f] = FN(dot3)(best_axis, n2[f]);
  int a_max = FN(pick)(w, fa, F1, +1, (REAL)1, NULL, 1), b_max = FN(pick)(w, fb, F2, +1, (REAL)1, NULL, 1);
  int a_min = FN(pick)(w, fa, F1, -1, (REAL)1, NULL, 1), b_min = FN(pick)(w, fb, F2, -1, (REAL)1, NULL, 1);
  free(fa);
  REAL ref_face[MJO_MAXK][3], inc_face[MJO_MAXK][3], ref_n[3], inc_n[3], sep[3];
  for (int k = 0; k < K; k++) for (int i = 0; i < 3; i++) {
    ref_face[k][i] = best_sign > 0 ? v1[FN(cvx_fv)(cv1, a_max, k)][i] : v2[FN(cvx_fv)(cv2, b_max, k)][i];
    inc_face[k][i] = best_sign > 0 ? v2[FN(cvx_fv)(cv2, b_min, k)][i] : v1[FN(cvx_fv)(cv1, a_min, k)][i];
  }
  for (int i = 0; i < 3; i++) {
    ref_n[i] = best_sign > 0 ? n1[a_max][i] : n2[b_max][i];
    inc_n[i] = best_sign > 0 ? n2[b_min][i] : n1[a_min][i];
    sep[i] = (REAL)(-best_sign) * best_axis[i];
  }
  REAL ldist[4], lpos[4][3], lnormal[3];
  FN(create_contact_manifold)(w, (const REAL(*)[3])ref_face, (const REAL(*)[3])inc_face, K, ref_n, inc_n, sep, ldist, lpos, lnormal);
  if (is_edge) { /* :581-599 */
    int idx = FN(pick)(w, ldist, 4, -1, (REAL)1, &lpos[0][0], 4);
    REAL dd = ldist[idx], pp[3] = {lpos[idx][0], lpos[idx][1], lpos[idx][2]};
    for (int q = 0; q < 4; q++) { ldist[q] = q == 0 ? dd : (REAL)1; for (int i = 0; i < 3; i++) lpos[q][i] = pp[i]; }
  }
  REAL nw[3];
  FN(mat_vec)(mat2, lnormal, nw);
  if (swapped) for (int i = 0; i < 3; i++) nw[i] = -nw[i];
  for (int q = 0; q < 4; q++) {
    REAL pw[3];
    FN(mat_vec)(mat2, lpos[q], pw);
    for (int i = 0; i < 3; i++) pos[q][i] = pos2[i] + pw[i];
    dist[q] = ldist[q];
    FN(make_frame)(nw, frame[q]);
  }
  free(v1); free(n1);
}

/* narrow phase of pair p (collision_driver.py:106-125 dispatch table) */
static void FN(pair_contacts)(const FN(MjoModel) * M, FN(MjoWork) * w, int p, REAL* dist, REAL (*pos)[3], REAL (*frame)[9]) {
  const mjhModelDesc* m = M->d;
  int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], fn = m->pair_fn[p], k = m->pair_ncon[p];
  const REAL *p1 = w->geom_xpos + 3 * g1, *m1 = w->geom_xmat + 9 * g1, *s1 = M->geom_size + 3 * g1;
  const REAL *p2 = w->geom_xpos + 3 * g2, *m2 = w->geom_xmat + 9 * g2, *s2 = M->geom_size + 3 * g2;
  if (fn == MJH_FN_PLANE_SPHERE) {
    REAL n[3] = {m1[2], m1[5], m1[8]};
    FN(plane_sphere_)(n, p1, p2, s2[0], &dist[0], pos[0]);
    FN(make_frame)(n, frame[0]);
  } else if (fn == MJH_FN_PLANE_CAPSULE) { /* collision_primitive.py:48-74 */
    REAL n[3] = {m1[2], m1[5], m1[8]}, axis[3] = {m2[2], m2[5], m2[8]};
    REAL na = FN(dot3)(n, axis), b[3];
    for (int i = 0; i < 3; i++) b[i] = axis[i] - n[i] * na;
    REAL bn = FN(normalize_n)(b, 3);
    if (bn < (REAL)0.5) {
      b[0] = 0; b[1] = 0; b[2] = 0;
      if ((REAL)-0.5 < n[1] && n[1] < (REAL)0.5) b[1] = 1; else b[2] = 1;
    }
    REAL c[3];
    FN(cross3)(n, b, c);
    REAL seg[3] = {axis[0] * s2[1], axis[1] * s2[1], axis[2] * s2[1]};
    for (int q = 0; q < 2; q++) {
      REAL sp[3];
      for (int i = 0; i < 3; i++) sp[i] = p2[i] + (q == 0 ? seg[i] : -seg[i]);
      FN(plane_sphere_)(n, p1, sp, s2[0], &dist[q], pos[q]);
      for (int i = 0; i < 3; i++) { frame[q][i] = n[i]; frame[q][3 + i] = b[i]; frame[q][6 + i] = c[i]; }
    }
  } else if (fn == MJH_FN_SPHERE_SPHERE) {
    REAL n[3];
    FN(sphere_sphere_)(p1, s1[0], p2, s2[0], &dist[0], pos[0], n, w->prim_hint_n, &w->prim_adopted);
    FN(make_frame)(n, frame[0]);
  } else if (fn == MJH_FN_SPHERE_CAPSULE) { /* :195-201 */
    REAL axis[3] = {m2[2], m2[5], m2[8]}, a[3], b[3], pt[3], n[3];
    for (int i = 0; i < 3; i++) { REAL sg = axis[i] * s2[1]; a[i] = p2[i] - sg; b[i] = p2[i] + sg; }
    FN(closest_segment_point)(a, b, p1, pt);
    FN(sphere_sphere_)(p1, s1[0], pt, s2[0], &dist[0], pos[0], n, w->prim_hint_n, &w->prim_adopted);
    FN(make_frame)(n, frame[0]);
  } else if (fn == MJH_FN_CAPSULE_CAPSULE) { /* :204-221 */
    REAL ax1[3] = {m1[2], m1[5], m1[8]}, ax2[3] = {m2[2], m2[5], m2[8]};
    REAL a0[3], a1[3], b0[3], b1[3], pt1[3], pt2[3], n[3];
    for (int i = 0; i < 3; i++) {
      REAL sg1 = ax1[i] * s1[1], sg2 = ax2[i] * s2[1];
      a0[i] = p1[i] - sg1; a1[i] = p1[i] + sg1; b0[i] = p2[i] - sg2; b1[i] = p2[i] + sg2;
    }
    FN(closest_segment_to_segment)(a0, a1, b0, b1, pt1, pt2);
    FN(sphere_sphere_)(pt1, s1[0], pt2, s2[0], &dist[0], pos[0], n, w->prim_hint_n, &w->prim_adopted);
    FN(make_frame)(n, frame[0]);
  } else if (fn == MJH_FN_PLANE_CONVEX) {
    FN(Cvx) c2 = FN(cvx_of)(M, g2);
    FN(plane_convex_)(w, p1, m1, p2, m2, &c2, dist, pos, frame);
  } else if (fn == MJH_FN_SPHERE_CONVEX) {
    FN(Cvx) c2 = FN(cvx_of)(M, g2);
    FN(sphere_convex_)(w, p1, s1[0], p2, m2, &c2, &dist[0], pos[0], frame[0]);
  } else if (fn == MJH_FN_CAPSULE_CONVEX) {
    FN(Cvx) c2 = FN(cvx_of)(M, g2);
    FN(capsule_convex_)(w, p1, m1, s1[0], s1[1], p2, m2, &c2, dist, pos, frame);
  } else if (fn == MJH_FN_CONVEX_CONVEX) {
    FN(Cvx) c1 = FN(cvx_of)(M, g1), c2 = FN(cvx_of)(M, g2);
    FN(convex_convex_)(w, p1, m1, &c1, p2, m2, &c2, dist, pos, frame);
  } else {
    for (int q = 0; q < k; q++) { dist[q] = 1; for (int i = 0; i < 3; i++) pos[q][i] = 0; for (int i = 0; i < 9; i++) frame[q][i] = 0; }
  }
}

static void FN(collision)(const FN(MjoModel) * M, FN(MjoWork) * w) { /* collision_driver.py:800-875 */
  const mjhModelDesc* m = M->d;
  for (int p = 0; p < m->npair; p++) {
    REAL dist[MJH_MAX_PAIR_CONTACTS], pos[MJH_MAX_PAIR_CONTACTS][3], frame[MJH_MAX_PAIR_CONTACTS][9];
    int k = m->pair_ncon[p];
    const int* dst = m->pair_dst + p * MJH_MAX_PAIR_CONTACTS;
    if (w->hint_dist && !m->topk && m->pair_fn[p] >= MJH_FN_PLANE_CONVEX) {  /* (hints are in contact-slot order: with max_contact_points the slots are per environment) */
      /* enumerate the pair's tie events; keep the outcome closest to the hint (the natural one on equality) */
      REAL bd[MJH_MAX_PAIR_CONTACTS], bp[MJH_MAX_PAIR_CONTACTS][3], bf[MJH_MAX_PAIR_CONTACTS][9], best_err = 0;
      int runs = 0, natural = 1;
      w->tie_on = 1;
      memset(w->tie_digit, 0, sizeof(w->tie_digit));
      do {
        w->tie_n = 0;
        FN(pair_contacts)(M, w, p, dist, pos, frame);
        REAL err = 0;
        for (int q = 0; q < k; q++) {
          int c = dst[q];
          REAL e = R_FABS(dist[q] - w->hint_dist[c]);
          if (e > err) err = e;
          for (int i = 0; i < 3; i++) { e = R_FABS(pos[q][i] - w->hint_pos[3 * c + i]); if (e > err) err = e; }
          for (int i = 0; i < 9; i++) { e = R_FABS(frame[q][i] - w->hint_frame[9 * c + i]); if (e > err) err = e; }
        }
        if (runs == 0 || err < best_err) { best_err = err; natural = runs == 0; memcpy(bd, dist, sizeof(bd)); memcpy(bp, pos, sizeof(bp)); memcpy(bf, frame, sizeof(bf)); }
        runs++;
        int e = (w->tie_n < MJO_MAX_TIE ? w->tie_n : MJO_MAX_TIE) - 1;
        while (e >= 0 && w->tie_digit[e] + 1 >= w->tie_count[e]) { w->tie_digit[e] = 0; e--; }
        if (e < 0) break;
        w->tie_digit[e]++;
      } while (runs < MJO_MAX_TIE_RUNS);
      w->tie_on = 0;
      if (!natural) w->tie_pairs++;
      memcpy(dist, bd, sizeof(bd)); memcpy(pos, bp, sizeof(bp)); memcpy(frame, bf, sizeof(bf));
    } else {
      w->prim_hint_n = (w->hint_dist && w->hint_frame && !m->topk) ? w->hint_frame + 9 * dst[0] : NULL; /* sphere / capsule pairs: one contact, normal = first frame row */
      w->prim_adopted = 0;
      if (w->stage_mode && m->pair_fn[p] >= MJH_FN_PLANE_CONVEX) w->tie_on = 2;
      FN(pair_contacts)(M, w, p, dist, pos, frame);
      w->tie_on = 0;
      if (w->prim_adopted) w->tie_pairs++;
      w->prim_hint_n = NULL;
    }
    for (int q = 0; q < k; q++) {
      int c = m->pair_dst[p * MJH_MAX_PAIR_CONTACTS + q];
      REAL *cd = m->topk ? w->cand_dist : w->contact_dist, *cp = m->topk ? w->cand_pos : w->contact_pos, *cf = m->topk ? w->cand_frame : w->contact_frame;
      cd[c] = dist[q];
      for (int i = 0; i < 3; i++) cp[3 * c + i] = pos[q][i];
      for (int i = 0; i < 9; i++) cf[9 * c + i] = frame[q][i];
    }
  }
  if (m->topk) {
    /* torch.topk(-dist, k = ncon): the ncon closest candidates, closest first (equal distances by candidate index: torch leaves
       their order to its partial sort), then contact[argsort(contact_dim)] over equal condims = the static permutation topk_slot */
    for (int q = 0; q < m->ncand; q++) {
      int rank = 0;
      for (int q2 = 0; q2 < m->ncand; q2++) rank += (w->cand_dist[q2] < w->cand_dist[q]) || (w->cand_dist[q2] == w->cand_dist[q] && q2 < q);
      if (rank < m->ncon) w->con_src[m->topk_slot[rank]] = q;
    }
    for (int c = 0; c < m->ncon; c++) {
      int q = w->con_src[c];
      w->contact_dist[c] = w->cand_dist[q];
      for (int i = 0; i < 3; i++) w->contact_pos[3 * c + i] = w->cand_pos[3 * q + i];
      for (int i = 0; i < 9; i++) w->contact_frame[9 * c + i] = w->cand_frame[9 * q + i];
    }
  } else {
    for (int c = 0; c < m->ncon; c++) w->con_src[c] = c;
  }
  for (int c = 0; c < m->ncon; c++) {
    int q = w->con_src[c];
    w->contact_includemargin[c] = M->con_includemargin[q];
    for (int i = 0; i < 5; i++) w->contact_friction[5 * c + i] = M->con_friction[5 * q + i];
    for (int i = 0; i < 2; i++) w->contact_solref[2 * c + i] = M->con_solref[2 * q + i];
    for (int i = 0; i < 2; i++) w->contact_solreffriction[2 * c + i] = M->con_solreffriction[2 * q + i];
    for (int i = 0; i < 5; i++) w->contact_solimp[5 * c + i] = M->con_solimp[5 * q + i];
  }
}

/* ---- constraint rows (constraint.py) ----------------------------------------------------------- */
static void FN(kbi)(const FN(MjoModel) * M, const REAL* solref, const REAL* solimp, REAL pos, REAL* k, REAL* b, REAL* imp) { /* :69-113 */
  REAL timeconst = solref[0], dampratio = solref[1];
  if (!(M->d->disableflags & DSBL_REFSAFE)) {
    REAL t2 = 2 * M->timestep;
    timeconst = (timeconst > t2 ? timeconst : t2) * (REAL)(timeconst > 0);
  }
  REAL dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
#define CLAMP_(x, lo, hi) ((x) < (lo) ? (lo) : ((x) > (hi) ? (hi) : (x)))
  dmin = CLAMP_(dmin, (REAL)mjMINIMP, (REAL)mjMAXIMP);
  dmax = CLAMP_(dmax, (REAL)mjMINIMP, (REAL)mjMAXIMP);
  width = width > (REAL)MINVAL_CACHED ? width : (REAL)MINVAL_CACHED;
  mid = CLAMP_(mid, (REAL)mjMINIMP, (REAL)mjMAXIMP);
  power = power > 1 ? power : 1;
  REAL kk = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  REAL bb = 2 / (dmax * timeconst);
  if (dampratio <= 0) kk = -dampratio / (dmax * dmax);
  if (timeconst <= 0) bb = -timeconst / dmax;
  REAL imp_x = R_FABS(pos) / width;
  REAL imp_a = (1 / R_POW(mid, power - 1)) * R_POW(imp_x, power);
  REAL imp_b = 1 - (1 / R_POW(1 - mid, power - 1)) * R_POW(1 - imp_x, power);
  REAL imp_y = imp_x < mid ? imp_a : imp_b;
  REAL im = dmin + imp_y * (dmax - dmin);
  im = CLAMP_(im, dmin, dmax);
  if (imp_x > 1) im = dmax;
  *k = kk; *b = bb; *imp = im;
}

/* support.jac :138-153 : jacp/jacr rows of `point` on `body`, masked to ancestor dofs */
static void FN(jac_dof)(const FN(MjoModel) * M, const FN(MjoWork) * w, const REAL* point, int body, int dof, REAL* jp, REAL* jr) {
  const mjhModelDesc* m = M->d;
  /* mask: dof's body is an ancestor-or-self of `body` */
  int db = m->dof_bodyid[dof], b = body, on = 0;
  while (b > 0) { if (b == db) { on = 1; break; } b = m->body_parentid[b]; }
  const REAL* rc = w->subtree_com + 3 * m->body_rootid[body];
  REAL off[3] = {point[0] - rc[0], point[1] - rc[1], point[2] - rc[2]};
  const REAL* cd = w->cdof + 6 * dof;
  REAL c[3];
  FN(cross3)(cd, off, c);
  for (int i = 0; i < 3; i++) { jp[i] = (cd[3 + i] + c[i]) * (REAL)on; jr[i] = cd[i] * (REAL)on; }
}

static void FN(make_constraint)(const FN(MjoModel) * M, FN(MjoWork) * w) { /* :600-768 */
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nefc = m->nefc;
  if (nefc == 0) return;
  for (int i = 0; i < nefc * nv; i++) w->efc_J[i] = 0;
  int row = 0;
  for (int r = 0; r < nefc; r++) w->efc_frictionloss[r] = 0;
  for (int q = 0; q < m->neqtab; q++) { /* equality rows: connects, welds, joint couplings (constraint.py:116-212, 254-296) */
    int kind = m->eq_kind[q], id = m->eq_id[q], id1 = m->eq_obj1[q], id2 = m->eq_obj2[q];
    const REAL* data = M->eq_data + 11 * id;
    REAL active = (REAL)w->eq_active[id];
    int width = kind == 0 ? 3 : (kind == 1 ? 6 : 1);
    if (row != m->eq_row[q]) abort();
    for (int r = row; r < row + width; r++) {
      for (int i = 0; i < 2; i++) w->efc_solref[2 * r + i] = M->eq_solref[2 * id + i];
      for (int i = 0; i < 5; i++) w->efc_solimp[5 * r + i] = M->eq_solimp[5 * id + i];
    }
    if (kind == 2) { /* _instantiate_equality_joint :254-296 */
      const int* ja = m->eq_jadr + 4 * q; /* dofadr1, dofadr2, qposadr1, qposadr2 */
      REAL has2 = (REAL)(id2 > -1);
      REAL pos1 = w->qpos[ja[2]], pos2 = w->qpos[ja[3]] * has2;
      REAL ref1 = M->qpos0[ja[2]], ref2 = M->qpos0[ja[3]] * has2;
      REAL dif = pos2 - ref2;
      REAL pw[5];
      for (int i = 0; i < 5; i++) pw[i] = R_POW(dif, (REAL)i);
      REAL deriv = 0, poly = 0;
      for (int i = 0; i < 4; i++) deriv += data[1 + i] * pw[i] * (REAL)(i + 1);
      for (int i = 0; i < 5; i++) poly += data[i] * pw[i];
      w->efc_J[row * nv + ja[0]] = 1;
      w->efc_J[row * nv + ja[1]] = -deriv; /* second scatter: overwrites the 1 when both joints share the dof (and, without a second joint, lands on the last joint's dof) */
      for (int d = 0; d < nv; d++) w->efc_J[row * nv + d] = w->efc_J[row * nv + d] * active;
      REAL pos = (pos1 - ref1 - poly) * active;
      w->efc_pos[row] = pos;
      w->efc_pos_norm[row] = pos;
      w->efc_invweight[row] = M->dof_invweight0[ja[0]] + M->dof_invweight0[ja[1]] * has2;
      row += 1;
      continue;
    }
    const REAL *xm1 = w->xmat + 9 * id1, *xm2 = w->xmat + 9 * id2;
    /* connect: anchor1 = data[0:3] on body1, anchor2 = data[3:6] on body2; weld: the point on body1 is data[3:6], on body2 data[0:3] */
    const REAL* a1 = kind == 0 ? data : data + 3;
    const REAL* a2 = kind == 0 ? data + 3 : data;
    REAL pos1[3], pos2[3], cpos[3];
    for (int i = 0; i < 3; i++) {
      pos1[i] = ((xm1[3 * i] * a1[0] + xm1[3 * i + 1] * a1[1]) + xm1[3 * i + 2] * a1[2]) + w->xpos[3 * id1 + i];
      pos2[i] = ((xm2[3 * i] * a2[0] + xm2[3 * i + 1] * a2[1]) + xm2[3 * i + 2] * a2[2]) + w->xpos[3 * id2 + i];
      cpos[i] = pos1[i] - pos2[i];
    }
    if (kind == 0) { /* _instantiate_equality_connect :116-157 */
      for (int d = 0; d < nv; d++) {
        REAL jp1[3], jr1[3], jp2[3], jr2[3];
        FN(jac_dof)(M, w, pos1, id1, d, jp1, jr1);
        FN(jac_dof)(M, w, pos2, id2, d, jp2, jr2);
        for (int i = 0; i < 3; i++) w->efc_J[(row + i) * nv + d] = (jp1[i] - jp2[i]) * active;
      }
      REAL nrm = FN(norm_n)(cpos, 3);
      for (int i = 0; i < 3; i++) {
        w->efc_pos[row + i] = cpos[i] * active;
        w->efc_pos_norm[row + i] = nrm * active;
        w->efc_invweight[row + i] = M->body_invweight0[id1] + M->body_invweight0[id2];
      }
      row += 3;
      continue;
    }
    /* _instantiate_equality_weld :160-212 */
    REAL torquescale = data[10];
    REAL quat[4], quat1[4], qd[4];
    FN(quat_mul)(w->xquat + 4 * id1, data + 6, quat);
    quat1[0] = w->xquat[4 * id2]; for (int i = 1; i < 4; i++) quat1[i] = w->xquat[4 * id2 + i] * (REAL)-1;
    FN(quat_mul)(quat1, quat, qd);
    REAL pos6[6] = {cpos[0], cpos[1], cpos[2], qd[1] * torquescale, qd[2] * torquescale, qd[3] * torquescale};
    for (int d = 0; d < nv; d++) {
      REAL jp1[3], jr1[3], jp2[3], jr2[3];
      FN(jac_dof)(M, w, pos1, id1, d, jp1, jr1);
      FN(jac_dof)(M, w, pos2, id2, d, jp2, jr2);
      REAL ax[3] = {(jr1[0] - jr2[0]) * torquescale, (jr1[1] - jr2[1]) * torquescale, (jr1[2] - jr2[2]) * torquescale};
      /* quat_mul(quat_mul_axis(quat1, ax), quat)[1:] (math.py:303-321) */
      REAL t[4] = {-quat1[1] * ax[0] - quat1[2] * ax[1] - quat1[3] * ax[2], quat1[0] * ax[0] + quat1[2] * ax[2] - quat1[3] * ax[1],
                   quat1[0] * ax[1] + quat1[3] * ax[0] - quat1[1] * ax[2], quat1[0] * ax[2] + quat1[1] * ax[1] - quat1[2] * ax[0]};
      REAL o[4];
      FN(quat_mul)(t, quat, o);
      for (int i = 0; i < 3; i++) {
        w->efc_J[(row + i) * nv + d] = (jp1[i] - jp2[i]) * active;
        w->efc_J[(row + 3 + i) * nv + d] = ((REAL)0.5 * o[1 + i]) * active;
      }
    }
    REAL nrm = FN(norm_n)(pos6, 6);
    for (int i = 0; i < 6; i++) {
      w->efc_pos[row + i] = pos6[i] * active;
      w->efc_pos_norm[row + i] = nrm * active;
      w->efc_invweight[row + i] = i < 3 ? M->body_invweight0[id1] + M->body_invweight0[id2] : M->body_invweight0_rot[id1] + M->body_invweight0_rot[id2];
    }
    row += 6;
  }
  for (int f = 0; f < m->nf; f++, row++) { /* _instantiate_friction :215-251 (dof rows) */
    int da = m->fric_dof[f];
    w->efc_J[row * nv + da] = 1;
    w->efc_pos[row] = 0;
    w->efc_pos_norm[row] = 0;
    w->efc_invweight[row] = M->dof_invweight0[da];
    w->efc_frictionloss[row] = M->dof_frictionloss[da];
    for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = M->dof_solref[2 * da + i];
    for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = M->dof_solimp[5 * da + i];
  }
  for (int f = 0; f < m->nft; f++, row++) { /* _instantiate_friction :215-251 (tendon rows: J = ten_J[t]) */
    int t = m->fric_tendon[f];
    for (int q = m->ten_adr[t]; q < m->ten_adr[t + 1]; q++) w->efc_J[row * nv + m->ten_dof[q]] = M->ten_coef[q]; /* last term wins, smooth.py:492-494 */
    w->efc_pos[row] = 0;
    w->efc_pos_norm[row] = 0;
    w->efc_invweight[row] = M->tendon_invweight0[t];
    w->efc_frictionloss[row] = M->tendon_frictionloss[t];
    for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = M->tendon_solref_fri[2 * t + i];
    for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = M->tendon_solimp_fri[5 * t + i];
  }
  for (int l = 0; l < m->nlb; l++, row++) { /* _instantiate_limit_ball :299-335 */
    int j = m->lim_ball_jnt[l], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    REAL q[4] = {w->qpos[qa], w->qpos[qa + 1], w->qpos[qa + 2], w->qpos[qa + 3]}, axis[3], angle;
    FN(quat_to_axis_angle)(q, axis, &angle);
    REAL rmax = M->jnt_range[2 * j] > M->jnt_range[2 * j + 1] ? M->jnt_range[2 * j] : M->jnt_range[2 * j + 1];
    REAL pos = rmax - angle - M->jnt_margin[j];
    REAL active = (REAL)(pos < 0);
    for (int k = 0; k < 3; k++) w->efc_J[row * nv + da + k] = (-axis[k]) * active;
    w->efc_pos[row] = pos * active;
    w->efc_pos_norm[row] = pos * active;
    w->efc_invweight[row] = M->dof_invweight0[da];
    for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = M->jnt_solref[2 * j + i];
    for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = M->jnt_solimp[5 * j + i];
  }
  for (int l = 0; l < m->nl; l++, row++) { /* _instantiate_limit_slide_hinge :338-372 */
    int j = m->lim_jnt[l], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    REAL q = w->qpos[qa];
    REAL dist_min = q - M->jnt_range[2 * j], dist_max = M->jnt_range[2 * j + 1] - q;
    REAL val = (REAL)(dist_min < dist_max) * 2 - 1;
    REAL pos = (dist_min < dist_max ? dist_min : dist_max) - M->jnt_margin[j];
    REAL active = (REAL)(pos < 0);
    w->efc_J[row * nv + da] = val * active;
    w->efc_pos[row] = pos * active;
    w->efc_pos_norm[row] = pos * active;
    w->efc_invweight[row] = M->dof_invweight0[da];
    for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = M->jnt_solref[2 * j + i];
    for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = M->jnt_solimp[5 * j + i];
  }
  for (int l = 0; l < m->nlt; l++, row++) { /* _instantiate_limit_tendon :375-405 */
    int t = m->lim_tendon[l];
    REAL length = w->ten_length[t];
    REAL dist_min = length - M->tendon_range[2 * t], dist_max = M->tendon_range[2 * t + 1] - length;
    REAL pos = (dist_min < dist_max ? dist_min : dist_max) - M->tendon_margin[t];
    REAL active = (REAL)(pos < 0);
    REAL sign = ((REAL)(dist_min < dist_max) * 2 - 1) * active;
    for (int d = 0; d < nv; d++) w->efc_J[row * nv + d] = w->ten_J[t * nv + d] * sign;
    w->efc_pos[row] = pos * active;
    w->efc_pos_norm[row] = pos * active;
    w->efc_invweight[row] = M->tendon_invweight0[t];
    for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = M->tendon_solref_lim[2 * t + i];
    for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = M->tendon_solimp_lim[5 * t + i];
  }
  int elliptic = m->cone == CONE_ELLIPTIC;
  for (int c = 0; c < m->ncon; c++) {
    int cq = w->con_src[c];
    int dim = m->con_dim[cq];
    int b1 = m->geom_bodyid[m->con_geom1[cq]], b2 = m->geom_bodyid[m->con_geom2[cq]];
    const REAL* fr = w->contact_frame + 9 * c;
    const REAL* cpos = w->contact_pos + 3 * c;
    const REAL* fric = w->contact_friction + 5 * c;
    REAL dist = w->contact_dist[c] - w->contact_includemargin[c];
    REAL t = M->body_invweight0[b1] + M->body_invweight0[b2];
    REAL active = (REAL)(dist < 0);
    /* diff rows: frame @ (jacp2-jacp1).T  (+ frame @ (jacr2-jacr1).T) -> jacdiff[6][nv] */
    for (int d = 0; d < nv; d++) {
      REAL jp1[3], jr1[3], jp2[3], jr2[3];
      FN(jac_dof)(M, w, cpos, b2, d, jp2, jr2);
      FN(jac_dof)(M, w, cpos, b1, d, jp1, jr1);
      REAL dp[3] = {jp2[0] - jp1[0], jp2[1] - jp1[1], jp2[2] - jp1[2]};
      REAL dr[3] = {jr2[0] - jr1[0], jr2[1] - jr1[1], jr2[2] - jr1[2]};
      for (int r = 0; r < 3; r++) {
        w->jacdiff[r * nv + d] = fr[3 * r] * dp[0] + fr[3 * r + 1] * dp[1] + fr[3 * r + 2] * dp[2];
        w->jacdiff[(3 + r) * nv + d] = fr[3 * r] * dr[0] + fr[3 * r + 1] * dr[1] + fr[3 * r + 2] * dr[2];
      }
    }
    if (dim == 1) { /* _instantiate_contact_frictionless :408-451 */
      for (int d = 0; d < nv; d++) w->efc_J[row * nv + d] = w->jacdiff[d] * active;
      w->efc_pos[row] = dist * active;
      w->efc_pos_norm[row] = dist * active;
      w->efc_invweight[row] = t;
      for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = w->contact_solref[2 * c + i];
      for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = w->contact_solimp[5 * c + i];
      row++;
    } else if (!elliptic) { /* _instantiate_contact_pyramidal :454-516 */
      int nedge = 2 * (dim - 1);
      REAL mu = fric[0];
      REAL iw = (t + mu * mu * t) * 2 * mu * mu / M->impratio;
      for (int e = 0; e < nedge; e++, row++) {
        REAL f = fric[e / 2] * ((e & 1) ? (REAL)-1 : (REAL)1);
        for (int d = 0; d < nv; d++) w->efc_J[row * nv + d] = (w->jacdiff[d] + w->jacdiff[(1 + e / 2) * nv + d] * f) * active;
        w->efc_pos[row] = dist * active;
        w->efc_pos_norm[row] = dist * active;
        w->efc_invweight[row] = iw;
        for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = w->contact_solref[2 * c + i];
        for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = w->contact_solimp[5 * c + i];
      }
    } else { /* _instantiate_contact_elliptic :519-583 */
      const REAL* sr = w->contact_solref + 2 * c;
      const REAL* srf0 = w->contact_solreffriction + 2 * c;
      int any = (srf0[0] != 0) || (srf0[1] != 0);
      REAL srf[2] = {srf0[0] + sr[0] * (REAL)(!any), srf0[1] + sr[1] * (REAL)(!any)};
      REAL iwf = t / M->impratio;
      for (int r = 0; r < dim; r++, row++) {
        for (int d = 0; d < nv; d++) w->efc_J[row * nv + d] = w->jacdiff[r * nv + d] * active;
        w->efc_pos[row] = (r == 0 ? dist : 0) * active;
        w->efc_pos_norm[row] = dist;
        if (r == 0) w->efc_invweight[row] = t;
        else if (r == 1) w->efc_invweight[row] = iwf;
        else w->efc_invweight[row] = iwf * ((fric[0] * fric[0]) / (fric[r - 1] * fric[r - 1]));
        for (int i = 0; i < 2; i++) w->efc_solref[2 * row + i] = (r == 0) ? sr[i] : srf[i];
        for (int i = 0; i < 5; i++) w->efc_solimp[5 * row + i] = w->contact_solimp[5 * c + i];
      }
    }
  }
  for (int r = 0; r < nefc; r++) { /* :683-693 */
    REAL k, b, imp;
    FN(kbi)(M, w->efc_solref + 2 * r, w->efc_solimp + 5 * r, w->efc_pos_norm[r], &k, &b, &imp);
    REAL rr = w->efc_invweight[r] * (1 - imp) / imp;
    rr = rr > (REAL)MINVAL_CACHED ? rr : (REAL)MINVAL_CACHED;
    REAL jv = 0;
    for (int d = 0; d < nv; d++) jv += w->efc_J[r * nv + d] * w->qvel[d];
    w->efc_aref[r] = -b * jv - k * imp * w->efc_pos[r];
    w->efc_D[r] = 1 / rr;
  }
}

/* smooth.tendon :470-497 (fixed tendons): length = sum coef * qpos, ten_J[t, dof] = coef */
static void FN(tendon)(const FN(MjoModel) * M, FN(MjoWork) * w) {
  const mjhModelDesc* m = M->d;
  int nv = m->nv;
  for (int i = 0; i < m->ntendon * nv; i++) w->ten_J[i] = 0;
  for (int t = 0; t < m->ntendon; t++) {
    REAL len = 0;
    for (int q = m->ten_adr[t]; q < m->ten_adr[t + 1]; q++) {
      len += M->ten_coef[q] * w->qpos[m->ten_qposadr[q]];
      w->ten_J[t * nv + m->ten_dof[q]] = M->ten_coef[q];
    }
    w->ten_length[t] = len;
  }
}

/* ---- velocity stage: transmission, com_vel, passive, rne ------------------------------------- */
static void FN(velocity)(const FN(MjoModel) * M, FN(MjoWork) * w) {
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nb = m->nbody, nu = m->nu;
  /* smooth.transmission :535-591 (joint transmissions on slide/hinge) */
  for (int i = 0; i < nu * nv; i++) w->actuator_moment[i] = 0;
  for (int i = 0; i < nu; i++) {
    const REAL* gear = M->act_gear + 6 * i;
    int jt = m->act_jnttype[i], da = m->act_dofadr[i], qa = m->act_qposadr[i];
    int inparent = m->act_trntype[i] == 1; /* TrnType.JOINTINPARENT */
    if (m->act_trntype[i] == 3) { /* TrnType.TENDON :558-561 */
      int t = m->act_trnid[i];
      w->actuator_length[i] = w->ten_length[t] * gear[0];
      for (int d = 0; d < nv; d++) w->actuator_moment[i * nv + d] = w->ten_J[t * nv + d] * gear[0];
    } else if (jt == JNT_FREE) { /* :565-574 */
      REAL vals[6] = {gear[0], gear[1], gear[2], gear[3], gear[4], gear[5]};
      if (inparent) {
        REAL qn[4] = {w->qpos[qa + 3], w->qpos[qa + 4] * (REAL)-1, w->qpos[qa + 5] * (REAL)-1, w->qpos[qa + 6] * (REAL)-1};
        FN(rotate)(gear + 3, qn, vals + 3);
      }
      w->actuator_length[i] = 0;
      for (int k = 0; k < 6; k++) w->actuator_moment[i * nv + da + k] = vals[k];
    } else if (jt == JNT_BALL) { /* :575-583 */
      REAL q[4] = {w->qpos[qa], w->qpos[qa + 1], w->qpos[qa + 2], w->qpos[qa + 3]};
      REAL axis[3], angle, ga[3] = {gear[0], gear[1], gear[2]};
      FN(quat_to_axis_angle)(q, axis, &angle);
      if (inparent) {
        REAL qn[4] = {q[0], q[1] * (REAL)-1, q[2] * (REAL)-1, q[3] * (REAL)-1};
        FN(rotate)(gear, qn, ga);
      }
      w->actuator_length[i] = ((axis[0] * angle) * ga[0] + (axis[1] * angle) * ga[1]) + (axis[2] * angle) * ga[2];
      for (int k = 0; k < 3; k++) w->actuator_moment[i * nv + da + k] = ga[k];
    } else { /* slide / hinge :584-586 */
      w->actuator_length[i] = w->qpos[qa] * gear[0];
      w->actuator_moment[i * nv + da] = gear[0];
    }
  }
  /* forward._velocity :87-99 */
  for (int t = 0; t < m->ntendon; t++) {
    REAL s = 0;
    for (int d = 0; d < nv; d++) s += w->ten_J[t * nv + d] * w->qvel[d];
    w->ten_velocity[t] = s;
  }
  for (int i = 0; i < nu; i++) {
    REAL s = 0;
    for (int d = 0; d < nv; d++) s += w->actuator_moment[i * nv + d] * w->qvel[d];
    w->actuator_velocity[i] = s;
  }
  /* smooth.com_vel :385-424 */
  for (int b = 0; b < nb; b++) {
    REAL cvel[6] = {0, 0, 0, 0, 0, 0};
    if (b > 0) for (int k = 0; k < 6; k++) cvel[k] = w->cvel[6 * m->body_parentid[b] + k];
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, t = m->jnt_type[j], d = m->jnt_dofadr[j];
      if (t == JNT_FREE) {
        REAL s[6];
        for (int k = 0; k < 6; k++) s[k] = (w->cdof[6 * d + k] * w->qvel[d] + w->cdof[6 * (d + 1) + k] * w->qvel[d + 1]) + w->cdof[6 * (d + 2) + k] * w->qvel[d + 2];
        for (int k = 0; k < 6; k++) cvel[k] = cvel[k] + s[k];
        for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) w->cdof_dot[6 * (d + r) + k] = 0;
        for (int r = 3; r < 6; r++) FN(motion_cross)(cvel, w->cdof + 6 * (d + r), w->cdof_dot + 6 * (d + r));
        for (int k = 0; k < 6; k++) s[k] = (w->cdof[6 * (d + 3) + k] * w->qvel[d + 3] + w->cdof[6 * (d + 4) + k] * w->qvel[d + 4]) + w->cdof[6 * (d + 5) + k] * w->qvel[d + 5];
        for (int k = 0; k < 6; k++) cvel[k] = cvel[k] + s[k];
      } else {
        int width = (t == JNT_BALL) ? 3 : 1;
        for (int r = 0; r < width; r++) FN(motion_cross)(cvel, w->cdof + 6 * (d + r), w->cdof_dot + 6 * (d + r));
        REAL s[6];
        for (int k = 0; k < 6; k++) {
          s[k] = w->cdof[6 * d + k] * w->qvel[d];
          for (int r = 1; r < width; r++) s[k] = s[k] + w->cdof[6 * (d + r) + k] * w->qvel[d + r];
        }
        for (int k = 0; k < 6; k++) cvel[k] = cvel[k] + s[k];
      }
    }
    for (int k = 0; k < 6; k++) w->cvel[6 * b + k] = cvel[k];
  }
  /* passive.passive :176-200, _spring_damper :80-145 */
  if (m->disableflags & (DSBL_SPRING | DSBL_DAMPER)) {
    for (int d = 0; d < nv; d++) w->qfrc_passive[d] = 0;
    for (int d = 0; d < nv; d++) w->qfrc_gravcomp[d] = 0; /* passive.py:178-183 */
  } else {
    for (int j = 0; j < m->njnt; j++) {
      int t = m->jnt_type[j], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
      REAL k = M->jnt_stiffness[j];
      if (t == JNT_FREE) {
        for (int i = 0; i < 3; i++) w->qfrc_passive[da + i] = -k * (w->qpos[qa + i] - M->qpos_spring[qa + i]);
        REAL r[3];
        FN(quat_sub)(w->qpos + qa + 3, M->qpos_spring + qa + 3, r);
        for (int i = 0; i < 3; i++) w->qfrc_passive[da + 3 + i] = -k * r[i];
      } else if (t == JNT_BALL) {
        REAL r[3];
        FN(quat_sub)(w->qpos + qa, M->qpos_spring + qa, r);
        for (int i = 0; i < 3; i++) w->qfrc_passive[da + i] = -k * r[i];
      } else {
        w->qfrc_passive[da] = -k * (w->qpos[qa] - M->qpos_spring[qa]);
      }
    }
    for (int d = 0; d < nv; d++) w->qfrc_passive[d] = (0 + w->qfrc_passive[d]) - M->dof_damping[d] * w->qvel[d];
    if (m->ntendon > 0) { /* tendon-level springs and dampers :119-144 (both flags are clear on this branch) */
      for (int t = 0; t < m->ntendon; t++) {
        REAL below = M->tendon_lengthspring[2 * t] - w->ten_length[t], above = M->tendon_lengthspring[2 * t + 1] - w->ten_length[t];
        REAL fs = below > 0 ? M->tendon_stiffness[t] * below : (REAL)0;
        fs = above < 0 ? M->tendon_stiffness[t] * above : fs;
        w->tmp_nefc[t] = fs + (-M->tendon_damping[t] * w->ten_velocity[t]);
      }
      for (int d = 0; d < nv; d++) {
        REAL s = 0;
        for (int t = 0; t < m->ntendon; t++) s += w->ten_J[t * nv + d] * w->tmp_nefc[t];
        w->qfrc_passive[d] = w->qfrc_passive[d] + s;
      }
    }
    if (M->has_gravcomp && !(m->disableflags & DSBL_GRAVITY)) { /* passive._gravcomp :148-156; with gravity off the input leaf is carried (:190-194) */
      for (int d = 0; d < nv; d++) {
        REAL acc = 0;
        for (int b = 0; b < nb; b++) {
          REAL mg = M->body_mass[b] * M->body_gravcomp[b];
          REAL f[3] = {-M->gravity[0] * mg, -M->gravity[1] * mg, -M->gravity[2] * mg};
          REAL jp[3], jr[3];
          FN(jac_dof)(M, w, w->xipos + 3 * b, b, d, jp, jr);
          acc += (jp[0] * f[0] + jp[1] * f[1]) + jp[2] * f[2];
        }
        w->qfrc_gravcomp[d] = acc;
        w->qfrc_passive[d] = w->qfrc_passive[d] + acc * (REAL)(1 - m->jnt_actgravcomp[m->dof_jntid[d]]);
      }
    }
    if (M->has_fluid) { /* passive._fluid :158-173 with _inertia_box_fluid_model :31-78 */
      const REAL pi = (REAL)3.14159265358979323846;
      for (int d = 0; d < nv; d++) w->tmp_nv[d] = 0;
      for (int b = 0; b < nb; b++) {
        const REAL* inr = M->body_inertia + 3 * b;
        REAL mass = M->body_mass[b], box[3];
        for (int i = 0; i < 3; i++) {
          REAL s3 = (inr[0] * (i == 0 ? (REAL)-1 : (REAL)1) + inr[1] * (i == 1 ? (REAL)-1 : (REAL)1)) + inr[2] * (i == 2 ? (REAL)-1 : (REAL)1);
          s3 = s3 > (REAL)1e-12 ? s3 : (REAL)1e-12;
          REAL mm = mass > (REAL)(float)1e-12 ? mass : (REAL)(float)1e-12;
          box[i] = R_SQRT(((REAL)6.0 * s3) / mm) * (REAL)(mass > 0);
        }
        const REAL *xi = w->ximat + 9 * b, *cv = w->cvel + 6 * b;
        const REAL* rc = w->subtree_com + 3 * m->body_rootid[b];
        REAL off[3] = {w->xipos[3 * b] - rc[0], w->xipos[3 * b + 1] - rc[1], w->xipos[3 * b + 2] - rc[2]};
        REAL c[3], v3[3], lvel[6], lwind[3];
        FN(cross3)(off, cv, c); /* math.transform_motion :437-452 */
        for (int i = 0; i < 3; i++) v3[i] = cv[3 + i] - c[i];
        for (int i = 0; i < 3; i++) {
          lvel[3 + i] = xi[i] * v3[0] + xi[3 + i] * v3[1] + xi[6 + i] * v3[2];
          lvel[i] = xi[i] * cv[0] + xi[3 + i] * cv[1] + xi[6 + i] * cv[2];
          lwind[i] = xi[i] * M->wind[0] + xi[3 + i] * M->wind[1] + xi[6 + i] * M->wind[2];
        }
        for (int i = 0; i < 3; i++) lvel[3 + i] = lvel[3 + i] + (-lwind[i]);
        REAL diam = ((box[0] + box[1]) + box[2]) / 3;
        REAL d3 = diam * diam * diam;
        REAL fa[3], fv[3];
        for (int i = 0; i < 3; i++) {
          fa[i] = lvel[i] * -pi * d3 * M->viscosity;
          fv[i] = lvel[3 + i] * (REAL)-3.0 * pi * diam * M->viscosity;
        }
        REAL sv[3] = {box[1] * box[2], box[0] * box[2], box[0] * box[1]};
        REAL b4[3] = {R_POW(box[0], (REAL)4), R_POW(box[1], (REAL)4), R_POW(box[2], (REAL)4)};
        REAL sa[3] = {box[0] * (b4[1] + b4[2]), box[1] * (b4[0] + b4[2]), box[2] * (b4[0] + b4[1])};
        for (int i = 0; i < 3; i++) {
          fv[i] = fv[i] - (REAL)0.5 * M->density * sv[i] * R_FABS(lvel[3 + i]) * lvel[3 + i];
          fa[i] = fa[i] - ((REAL)1.0 * M->density * sa[i] * R_FABS(lvel[i]) * lvel[i] / (REAL)64.0);
        }
        REAL force[3], torque[3];
        for (int i = 0; i < 3; i++) {
          force[i] = xi[3 * i] * fv[0] + xi[3 * i + 1] * fv[1] + xi[3 * i + 2] * fv[2];
          torque[i] = xi[3 * i] * fa[0] + xi[3 * i + 1] * fa[1] + xi[3 * i + 2] * fa[2];
        }
        for (int d = 0; d < nv; d++) { /* support.apply_ft :169-181, summed over bodies in order */
          REAL jp[3], jr[3];
          FN(jac_dof)(M, w, w->xipos + 3 * b, b, d, jp, jr);
          w->tmp_nv[d] += FN(dot3)(jp, force) + FN(dot3)(jr, torque);
        }
      }
      for (int d = 0; d < nv; d++) w->qfrc_passive[d] = w->qfrc_passive[d] + w->tmp_nv[d];
    }
  }
  /* smooth.rne :427-467 */
  for (int b = 0; b < nb; b++) {
    REAL cacc[6];
    if (b == 0) {
      int nograv = m->disableflags & DSBL_GRAVITY;
      for (int k = 0; k < 3; k++) { cacc[k] = 0; cacc[3 + k] = nograv ? 0 : -M->gravity[k]; }
    } else {
      for (int k = 0; k < 6; k++) cacc[k] = w->cacc[6 * m->body_parentid[b] + k];
    }
    int d0 = m->body_dofadr[b], nd = m->body_dofnum[b];
    if (nd > 0) {
      for (int k = 0; k < 6; k++) {
        REAL s = w->cdof_dot[6 * d0 + k] * w->qvel[d0];
        for (int r = 1; r < nd; r++) s = s + w->cdof_dot[6 * (d0 + r) + k] * w->qvel[d0 + r];
        cacc[k] = cacc[k] + s;
      }
    }
    for (int k = 0; k < 6; k++) w->cacc[6 * b + k] = cacc[k];
    REAL f1[6], f2[6], f3[6];
    FN(inert_mul)(w->cinert + 10 * b, cacc, f1);
    FN(inert_mul)(w->cinert + 10 * b, w->cvel + 6 * b, f2);
    FN(motion_cross_force)(w->cvel + 6 * b, f2, f3);
    for (int k = 0; k < 6; k++) w->cfrc[6 * b + k] = f1[k] + f3[k];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int k = 0; k < 6; k++) w->cfrc[6 * p + k] += w->cfrc[6 * b + k];
  }
  for (int d = 0; d < nv; d++) {
    REAL s = 0;
    for (int k = 0; k < 6; k++) s += w->cdof[6 * d + k] * w->cfrc[6 * m->dof_bodyid[d] + k];
    w->qfrc_bias[d] = s;
  }
}

/* ---- actuation + acceleration (forward.py:102-228) ---------------------------------------------- */
/* ---- muscle actuators (support.py:197-296) ---- */
static REAL FN(clamp_min)(REAL x, REAL lo) { return x > lo ? x : lo; }
static REAL FN(sq)(REAL x) { return x * x; }
static REAL FN(muscle_sigmoid)(REAL x) { /* :197-202 */
  REAL sol = x * x * x * (3 * x * (2 * x - 5) + 10);
  sol = x <= 0 ? (REAL)0 : sol;
  return x >= 1 ? (REAL)1 : sol;
}
static REAL FN(muscle_dynamics)(REAL ctrl, REAL act, const REAL* prm) { /* :205-232 */
  REAL ctrlclamp = ctrl < 0 ? (REAL)0 : (ctrl > 1 ? (REAL)1 : ctrl), actclamp = act < 0 ? (REAL)0 : (act > 1 ? (REAL)1 : act);
  REAL tau_act = prm[0] * ((REAL)0.5 + (REAL)1.5 * actclamp), tau_deact = prm[1] / ((REAL)0.5 + (REAL)1.5 * actclamp), width = prm[2];
  REAL dctrl = ctrlclamp - act;
  REAL tau_hard = dctrl > 0 ? tau_act : tau_deact;
  REAL q = dctrl / (width + (width == 0 ? (REAL)(float)mjMINVAL : (REAL)0)); /* math.safe_div */
  REAL tau_smooth = tau_deact + (tau_act - tau_deact) * FN(muscle_sigmoid)(q + (REAL)0.5);
  REAL tau = width < (REAL)mjMINVAL ? tau_hard : tau_smooth;
  return dctrl / FN(clamp_min)(tau, (REAL)mjMINVAL);
}
static REAL FN(muscle_gain_length)(REAL len, REAL lmin, REAL lmax) { /* :235-249 */
  REAL a = (REAL)0.5 * (lmin + 1), b = (REAL)0.5 * (1 + lmax);
  REAL out0 = (REAL)0.5 * FN(sq)((len - lmin) / FN(clamp_min)(a - lmin, (REAL)mjMINVAL));
  REAL out1 = 1 - (REAL)0.5 * FN(sq)((1 - len) / FN(clamp_min)(1 - a, (REAL)mjMINVAL));
  REAL out2 = 1 - (REAL)0.5 * FN(sq)((len - 1) / FN(clamp_min)(b - 1, (REAL)mjMINVAL));
  REAL out3 = (REAL)0.5 * FN(sq)((lmax - len) / FN(clamp_min)(lmax - b, (REAL)mjMINVAL));
  REAL o = len <= b ? out2 : out3;
  o = len <= 1 ? out1 : o;
  o = len <= a ? out0 : o;
  return (lmin <= len && len <= lmax) ? o : (REAL)0;
}
static REAL FN(muscle_gain)(REAL len, REAL vel, const REAL* lr, REAL acc0, const REAL* prm) { /* :252-278 */
  REAL force = prm[2], scale = prm[3], lmin = prm[4], lmax = prm[5], vmax = prm[6], fvmax = prm[8];
  force = force < 0 ? scale / FN(clamp_min)(acc0, (REAL)mjMINVAL) : force;
  REAL L0 = (lr[1] - lr[0]) / FN(clamp_min)(prm[1] - prm[0], (REAL)mjMINVAL);
  REAL L = prm[0] + (len - lr[0]) / FN(clamp_min)(L0, (REAL)mjMINVAL);
  REAL V = vel / FN(clamp_min)(L0 * vmax, (REAL)mjMINVAL);
  REAL FL = FN(muscle_gain_length)(L, lmin, lmax);
  REAL y = fvmax - 1;
  REAL FV = V <= y ? fvmax - FN(sq)(y - V) / FN(clamp_min)(y, (REAL)mjMINVAL) : fvmax;
  FV = V <= 0 ? FN(sq)(V + 1) : FV;
  FV = V <= -1 ? (REAL)0 : FV;
  return -force * FL * FV;
}
static REAL FN(muscle_bias)(REAL len, const REAL* lr, REAL acc0, const REAL* prm) { /* :281-296 */
  REAL force = prm[2], scale = prm[3], lmax = prm[5], fpmax = prm[7];
  force = force < 0 ? scale / FN(clamp_min)(acc0, (REAL)mjMINVAL) : force;
  REAL L0 = (lr[1] - lr[0]) / FN(clamp_min)(prm[1] - prm[0], (REAL)mjMINVAL);
  REAL L = prm[0] + (len - lr[0]) / FN(clamp_min)(L0, (REAL)mjMINVAL);
  REAL b = (REAL)0.5 * (1 + lmax);
  REAL out1 = -force * fpmax * (REAL)0.5 * FN(sq)((L - 1) / FN(clamp_min)(b - 1, (REAL)mjMINVAL));
  REAL out2 = -force * fpmax * ((REAL)0.5 + (L - b) / FN(clamp_min)(b - 1, (REAL)mjMINVAL));
  REAL o = L <= b ? out1 : out2;
  return L <= 1 ? (REAL)0 : o;
}

static void FN(actuation)(const FN(MjoModel) * M, FN(MjoWork) * w) {
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nu = m->nu;
  if (nu == 0 || (m->disableflags & DSBL_ACTUATION)) {
    for (int i = 0; i < m->na; i++) w->act_dot[i] = 0;
    for (int d = 0; d < nv; d++) w->qfrc_actuator[d] = 0;
  } else {
    for (int d = 0; d < nv; d++) w->qfrc_actuator[d] = 0;
    for (int i = 0; i < nu; i++) {
      REAL ctrl = w->ctrl[i];
      if (!(m->disableflags & DSBL_CLAMPCTRL) && m->act_ctrllimited[i]) {
        REAL lo = M->act_ctrlrange[2 * i], hi = M->act_ctrlrange[2 * i + 1];
        ctrl = ctrl > lo ? ctrl : lo;
        ctrl = ctrl < hi ? ctrl : hi;
      }
      REAL ctrl_act = ctrl;
      int dyn = m->act_dyntype[i];
      if (dyn != DYN_NONE) {
        int a = m->act_actadr[i];
        REAL act = w->act[a];
        if (dyn == DYN_INTEGRATOR) w->act_dot[a] = ctrl;
        else if (dyn == DYN_MUSCLE) w->act_dot[a] = FN(muscle_dynamics)(ctrl, act, M->act_dynprm + 3 * i);
        else {
          REAL tau = M->act_dynprm[3 * i];
          tau = tau > (REAL)mjMINVAL ? tau : (REAL)mjMINVAL;
          w->act_dot[a] = (ctrl - act) / tau;
        }
        ctrl_act = w->act[a + m->act_actnum[i] - 1];
      }
      REAL len = w->actuator_length[i], vel = w->actuator_velocity[i];
      const REAL* gp = M->act_gainprm + 9 * i;
      const REAL* bp = M->act_biasprm + 9 * i;
      REAL gain = (m->act_gaintype[i] == GAIN_FIXED) ? gp[0] : gp[0] + gp[1] * len + gp[2] * vel;
      REAL bias = (m->act_biastype[i] == BIAS_AFFINE) ? bp[0] + bp[1] * len + bp[2] * vel : 0;
      if (m->act_gaintype[i] == GAIN_MUSCLE) gain = FN(muscle_gain)(len, vel, M->act_lengthrange + 2 * i, M->act_acc0[i], gp);
      if (m->act_biastype[i] == BIAS_MUSCLE) bias = FN(muscle_bias)(len, M->act_lengthrange + 2 * i, M->act_acc0[i], bp);
      REAL force = gain * ctrl_act + bias;
      if (m->act_forcelimited[i]) {
        REAL lo = M->act_forcerange[2 * i], hi = M->act_forcerange[2 * i + 1];
        force = force < lo ? lo : (force > hi ? hi : force);
      }
      w->actuator_force[i] = force;
    }
    for (int d = 0; d < nv; d++) {
      REAL s = 0;
      for (int i = 0; i < nu; i++) s += w->actuator_moment[i * nv + d] * w->actuator_force[i];
      int j = m->dof_jntid[d];
      if (M->has_gravcomp) s = s + w->qfrc_gravcomp[d] * (REAL)m->jnt_actgravcomp[j]; /* forward.py:206-207 */
      if (m->jnt_actfrclimited[j]) {
        REAL lo = M->jnt_actfrcrange[2 * j], hi = M->jnt_actfrcrange[2 * j + 1];
        s = s < lo ? lo : (s > hi ? hi : s);
      }
      w->qfrc_actuator[d] = s;
    }
  }
  /* _acceleration :222-228, support.xfrc_accumulate :184-194 */
  for (int d = 0; d < nv; d++) w->tmp_nv[d] = 0;
  for (int b = 0; b < m->nbody; b++) {
    const REAL* f = w->xfrc_applied + 6 * b;
    for (int d = 0; d < nv; d++) {
      REAL jp[3], jr[3];
      FN(jac_dof)(M, w, w->xipos + 3 * b, b, d, jp, jr);
      w->tmp_nv[d] += FN(dot3)(jp, f) + FN(dot3)(jr, f + 3);
    }
  }
  for (int d = 0; d < nv; d++) {
    REAL applied = w->qfrc_applied[d] + w->tmp_nv[d];
    w->qfrc_smooth[d] = ((w->qfrc_passive[d] - w->qfrc_bias[d]) + w->qfrc_actuator[d]) + applied;
  }
  FN(cholesky_solve)(w->qLD, w->qfrc_smooth, w->qacc_smooth, nv, w->tmp_nv2);
}

/* ---- solver (solver.py:244-553) ---------------------------------------------------------------- */
typedef struct FN(LSPoint) { REAL alpha, cost, d0, d1; } FN(LSPoint);

typedef struct FN(SolveCtx) {
  REAL gauss, cost, prev_cost;
  int niter;
} FN(SolveCtx);

static void FN(update_constraint)(const FN(MjoModel) * M, FN(MjoWork) * w, FN(SolveCtx) * c) { /* :320-357 */
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nefc = m->nefc;
  REAL csum = 0, fneg = 0, fpos = 0;
  int ne_nf = m->ne + m->nf + m->nft;
  for (int r = 0; r < nefc; r++) {
    REAL ja = w->s_Jaref[r];
    int active = (ja < 0) || (r < ne_nf);
    REAL floss_force = 0;
    if (m->nf + m->nft > 0) { /* quadratic inside |Jaref| < R f, linear outside (solver.py:326-342) */
      REAL fl = w->efc_frictionloss[r];
      REAL rr = 1 / (w->efc_D[r] + (REAL)(w->efc_D[r] == 0) * (REAL)(float)mjMINVAL);
      int lin_neg = (ja <= -rr * fl) && (fl > 0), lin_pos = (ja >= rr * fl) && (fl > 0);
      active = active && !lin_neg && !lin_pos;
      floss_force = lin_neg ? fl : (lin_pos ? -fl : (REAL)0);
      fneg += (REAL)lin_neg * ((REAL)-0.5 * rr * fl * fl - fl * ja);
      fpos += (REAL)lin_pos * ((REAL)-0.5 * rr * fl * fl + fl * ja);
    }
    w->s_active[r] = (unsigned char)active;
    w->s_force[r] = w->efc_D[r] * -ja * (REAL)active + floss_force;
    csum += w->efc_D[r] * ja * ja * (REAL)active;
  }
  for (int d = 0; d < nv; d++) {
    REAL s = 0;
    for (int r = 0; r < nefc; r++) s += w->efc_J[r * nv + d] * w->s_force[r];
    w->s_qfrc[d] = s;
  }
  REAL g = 0;
  for (int d = 0; d < nv; d++) g += (w->s_Ma[d] - w->qfrc_smooth[d]) * (w->s_qacc[d] - w->qacc_smooth[d]);
  c->gauss = (REAL)0.5 * g;
  REAL cost = ((REAL)0.5 * csum + c->gauss) + (fneg + fpos);
  c->prev_cost = c->cost;
  c->cost = cost;
}

static void FN(update_gradient)(const FN(MjoModel) * M, FN(MjoWork) * w) { /* :359-376 */
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nefc = m->nefc;
  for (int d = 0; d < nv; d++) w->s_grad[d] = (w->s_Ma[d] - w->qfrc_smooth[d]) - w->s_qfrc[d];
  if (m->solver == SOL_CG) {
    FN(cholesky_solve)(w->qLD, w->s_grad, w->s_Mgrad, nv, w->tmp_nv2);
  } else {
    for (int i = 0; i < nv; i++)
      for (int j = 0; j < nv; j++) {
        REAL s = 0;
        for (int r = 0; r < nefc; r++) s += (w->efc_J[r * nv + i] * w->efc_D[r] * (REAL)w->s_active[r]) * w->efc_J[r * nv + j];
        w->H[i * nv + j] = w->qM[i * nv + j] + s;
      }
    FN(cholesky)(w->H, w->HL, nv);
    FN(cholesky_solve)(w->HL, w->s_grad, w->s_Mgrad, nv, w->tmp_nv2);
  }
}

static void FN(create_context)(const FN(MjoModel) * M, FN(MjoWork) * w, FN(SolveCtx) * c, const REAL* qacc, int grad_flag) { /* :293-318 */
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nefc = m->nefc;
  for (int d = 0; d < nv; d++) w->s_qacc[d] = qacc[d];
  for (int r = 0; r < nefc; r++) {
    REAL s = 0;
    for (int d = 0; d < nv; d++) s += w->efc_J[r * nv + d] * qacc[d];
    w->s_Jaref[r] = s - w->efc_aref[r];
  }
  for (int i = 0; i < nv; i++) {
    REAL s = 0;
    for (int j = 0; j < nv; j++) s += w->qM[i * nv + j] * qacc[j];
    w->s_Ma[i] = s;
  }
  c->gauss = 0; c->cost = (REAL)INFINITY; c->prev_cost = 0; c->niter = 0;
  for (int d = 0; d < nv; d++) { w->s_grad[d] = 0; w->s_Mgrad[d] = 0; w->s_search[d] = 0; }
  FN(update_constraint)(M, w, c);
  if (grad_flag) {
    FN(update_gradient)(M, w);
    for (int d = 0; d < nv; d++) w->s_search[d] = -w->s_Mgrad[d];
  }
}

static FN(LSPoint) FN(ls_point)(const FN(MjoWork) * w, int nefc, const REAL* qg, REAL alpha) { /* point_fn :396-422 */
  REAL q0 = 0, q1 = 0, q2 = 0;
  REAL f0n = 0, f0p = 0, f1n = 0, f1p = 0;
  for (int r = 0; r < nefc; r++) {
    REAL x = w->s_Jaref[r] + alpha * w->s_jv[r];
    int act = (x < 0) || (r < w->ne_nf);
    if (w->nf > 0) {
      REAL fl = w->efc_frictionloss[r];
      REAL rr = 1 / (w->efc_D[r] + (REAL)(w->efc_D[r] == 0) * (REAL)(float)mjMINVAL);
      REAL rf = rr * fl;
      int ln = (x <= -rf) && (fl > 0), lp = (x >= rf) && (fl > 0);
      f0n += (REAL)ln * fl * ((REAL)-0.5 * rf - w->s_Jaref[r]);
      f0p += (REAL)lp * fl * ((REAL)-0.5 * rf + w->s_Jaref[r]);
      f1n += (REAL)ln * (-fl * w->s_jv[r]);
      f1p += (REAL)lp * (fl * w->s_jv[r]);
      act = act && !ln && !lp;
    }
    REAL a = (REAL)act;
    q0 += w->s_quad[3 * r] * a;
    q1 += w->s_quad[3 * r + 1] * a;
    q2 += w->s_quad[3 * r + 2] * a;
  }
  REAL t0 = (qg[0] + q0) + (f0n + f0p), t1 = (qg[1] + q1) + (f1n + f1p), t2 = (qg[2] + q2) + 0;
  FN(LSPoint) p;
  p.alpha = alpha;
  p.cost = alpha * alpha * t2 + alpha * t1 + t0;
  p.d0 = 2 * alpha * t2 + t1;
  p.d1 = 2 * t2 + (REAL)(t2 == 0) * (REAL)mjMINVAL;
  return p;
}

static inline int FN(ls_swap)(REAL cur, REAL cand, int not_bracketed) { /* _swap :440-449 */
  int in_bracket = ((cur < cand) && (cand < 0)) || ((cur > cand) && (cand > 0));
  return in_bracket || (not_bracketed && (R_FABS(cand) < R_FABS(cur)));
}

static void FN(linesearch)(const FN(MjoModel) * M, FN(MjoWork) * w, FN(SolveCtx) * c, int fixed_iterations) { /* :378-497 */
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nefc = m->nefc;
  REAL scale = (REAL)(m->meaninertia * (double)(nv > 1 ? nv : 1)); /* python float, solver.py:288 */
  REAL smag = FN(norm_n)(w->s_search, nv) * scale;
  REAL gtol = (REAL)(m->tolerance * m->ls_tolerance) * smag;
  for (int i = 0; i < nv; i++) {
    REAL s = 0;
    for (int j = 0; j < nv; j++) s += w->qM[i * nv + j] * w->s_search[j];
    w->s_mv[i] = s;
  }
  for (int r = 0; r < nefc; r++) {
    REAL s = 0;
    for (int d = 0; d < nv; d++) s += w->efc_J[r * nv + d] * w->s_search[d];
    w->s_jv[r] = s;
  }
  REAL sMa = 0, sqs = 0, smv = 0;
  for (int d = 0; d < nv; d++) { sMa += w->s_search[d] * w->s_Ma[d]; sqs += w->s_search[d] * w->qfrc_smooth[d]; smv += w->s_search[d] * w->s_mv[d]; }
  REAL qg[3] = {c->gauss, sMa - sqs, (REAL)0.5 * smv};
  for (int r = 0; r < nefc; r++) {
    REAL ja = w->s_Jaref[r], jv = w->s_jv[r], D = w->efc_D[r];
    w->s_quad[3 * r] = ((REAL)0.5 * ja * ja) * D;
    w->s_quad[3 * r + 1] = (jv * ja) * D;
    w->s_quad[3 * r + 2] = ((REAL)0.5 * jv * jv) * D;
  }
  FN(LSPoint) p0 = FN(ls_point)(w, nefc, qg, 0);
  FN(LSPoint) p1 = FN(ls_point)(w, nefc, qg, p0.alpha - p0.d0 / p0.d1);
  int early = R_FABS(p1.d0) < gtol;
  FN(LSPoint) lo, hi;
  if (p1.d0 < p0.d0) { hi = p0; lo = p1; } else { hi = p1; lo = p0; }
  int swap = !early, ls_iter = 0;
  for (;;) {
    if (fixed_iterations) { if (ls_iter >= m->ls_iterations) break; }
    else {
      int done = ls_iter >= m->ls_iterations;
      done |= !swap;
      done |= (lo.d0 < 0) && (lo.d0 > -gtol);
      done |= (hi.d0 > 0) && (hi.d0 < gtol);
      if (done) break;
    }
    w->stat_ls++;
    FN(LSPoint) lo_next = FN(ls_point)(w, nefc, qg, lo.alpha - lo.d0 / lo.d1);
    FN(LSPoint) hi_next = FN(ls_point)(w, nefc, qg, hi.alpha - hi.d0 / hi.d1);
    FN(LSPoint) mid = FN(ls_point)(w, nefc, qg, (REAL)0.5 * (lo.alpha + hi.alpha));
    {
      /* knife-edge bookkeeping (test diagnostics; natural behaviour when knife_policy < 0) */
#ifdef REAL_IS_FLOAT
      REAL noise = (REAL)g_knife_band_f32 * (R_FABS(p0.d0) + (REAL)1e-30);
#else
      REAL noise = (REAL)g_knife_band_f64 * (R_FABS(p0.d0) + (REAL)1e-300);
#endif
      FN(LSPoint)* cands[3] = {&lo_next, &hi_next, &mid};
      for (int q = 0; q < 3; q++) {
        FN(LSPoint)* cd = cands[q];
        mjo_knife_hist_add((double)cd->d0, (double)p0.d0, cd->alpha != lo.alpha && cd->alpha != hi.alpha);
        if (R_FABS(cd->d0) < noise && cd->alpha != lo.alpha && cd->alpha != hi.alpha) {
          if (w->knife_policy >= (1 << 30)) {
            /* "every Newton candidate that lands on the root rounds to exactly zero": only derivatives at the rounding floor of their own sum (a few ulp of
               |d0(0)|), not the wider noise band -- bisection points that approach the root stay what they are */
#ifdef REAL_IS_FLOAT
            if (R_FABS(cd->d0) < (REAL)4e-6 * R_FABS(p0.d0)) cd->d0 = 0;
#else
            if (R_FABS(cd->d0) < (REAL)1e-14 * R_FABS(p0.d0)) cd->d0 = 0;
#endif
          } else if (w->knife_policy >= 0) {
            if (w->knife < w->knife_policy) cd->d0 = 0;                       /* "rounded to exactly zero": rejected */
            else if (w->knife == w->knife_policy && cd->d0 == 0) cd->d0 = -noise * (REAL)1e-6; /* "not exactly zero": accepted */
          }
          w->knife++;
        }
      }
    }
    int nb = (lo.d0 < 0) == (hi.d0 < 0);
    int s1 = FN(ls_swap)(lo.d0, lo_next.d0, nb); if (s1) lo = lo_next;
    int s2 = FN(ls_swap)(lo.d0, mid.d0, nb); if (s2) lo = mid;
    int s3 = FN(ls_swap)(lo.d0, hi_next.d0, nb); if (s3) lo = hi_next;
    int s4 = FN(ls_swap)(hi.d0, hi_next.d0, nb); if (s4) hi = hi_next;
    int s5 = FN(ls_swap)(hi.d0, mid.d0, nb); if (s5) hi = mid;
    int s6 = FN(ls_swap)(hi.d0, lo_next.d0, nb); if (s6) hi = lo_next;
    swap = s1 | s2 | s3 | s4 | s5 | s6;
    if (mjo_trace_on()) fprintf(stderr, "[mjo]     ls %d: lo_next (a %.17g d0 %.6e) hi_next (a %.17g d0 %.6e) mid (a %.17g d0 %.6e) swaps %d%d%d%d%d%d -> lo (a %.17g d0 %.6e) hi (a %.17g d0 %.6e)\n", ls_iter, (double)lo_next.alpha, (double)lo_next.d0, (double)hi_next.alpha, (double)hi_next.d0, (double)mid.alpha, (double)mid.d0, s1, s2, s3, s4, s5, s6, (double)lo.alpha, (double)lo.d0, (double)hi.alpha, (double)hi.d0);
    ls_iter++;
  }
  REAL improved = (REAL)((lo.cost < p0.cost) || (hi.cost < p0.cost));
  REAL alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  if (mjo_trace_on()) fprintf(stderr, "[mjo]   ls: iters %d early %d p0 (d0 %.6e d1 %.6e cost %.17g) p1 (a %.17g d0 %.6e) lo (a %.17g d0 %.6e cost %.17g) hi (a %.17g d0 %.6e cost %.17g) gtol %.3e alpha %.17g improved %g\n", ls_iter, early, (double)p0.d0, (double)p0.d1, (double)p0.cost, (double)p1.alpha, (double)p1.d0, (double)lo.alpha, (double)lo.d0, (double)lo.cost, (double)hi.alpha, (double)hi.d0, (double)hi.cost, (double)gtol, (double)alpha, (double)improved);
  for (int d = 0; d < nv; d++) {
    w->s_qacc[d] = w->s_qacc[d] + improved * w->s_search[d] * alpha;
    w->s_Ma[d] = w->s_Ma[d] + improved * w->s_mv[d] * alpha;
  }
  for (int r = 0; r < nefc; r++) w->s_Jaref[r] = w->s_Jaref[r] + improved * w->s_jv[r] * alpha;
}

static void FN(solve)(const FN(MjoModel) * M, FN(MjoWork) * w, int fixed_iterations) {
  const mjhModelDesc* m = M->d;
  int nv = m->nv, nefc = m->nefc;
  REAL scale = (REAL)(m->meaninertia * (double)(nv > 1 ? nv : 1));
  FN(SolveCtx) c;
  const REAL* start = w->qacc_smooth;
  if (!(m->disableflags & DSBL_WARMSTART)) { /* :526-531 */
    FN(create_context)(M, w, &c, w->qacc_warmstart, 0);
    REAL warm_cost = c.cost;
    FN(create_context)(M, w, &c, w->qacc_smooth, 0);
    REAL smth_cost = c.cost;
    start = (warm_cost < smth_cost) ? w->qacc_warmstart : w->qacc_smooth;
  }
  for (int d = 0; d < nv; d++) w->tmp_nv[d] = start[d];
  FN(create_context)(M, w, &c, w->tmp_nv, 1);
  w->stat_solves++;
  for (int r = 0; r < nefc; r++) w->stat_rows += w->efc_D[r] != 0 && w->efc_aref[r] != 0;
  for (int it = 0;; it++) {
    if (m->iterations == 1) { if (it >= 1) break; }
    else if (fixed_iterations) { if (it >= m->iterations) break; }
    else { /* cond :501-508 */
      REAL improvement = (c.prev_cost - c.cost) / scale;
      REAL gradient = FN(norm_n)(w->s_grad, nv) / scale;
      /* torch.norm has no all-zero special case but agrees: sqrt(0) = 0 */
      int done = c.niter >= m->iterations;
      done |= improvement < (REAL)m->tolerance;
      done |= gradient < (REAL)m->tolerance;
      if (mjo_trace_on()) fprintf(stderr, "[mjo] it %d cost %.17g improvement %.6e gradient %.6e done %d qacc0 %.17g\n", c.niter, (double)c.cost, (double)improvement, (double)gradient, done, (double)w->s_qacc[0]);
      if (done) break;
    }
    /* body :510-524 */
    w->stat_niter++;
    FN(linesearch)(M, w, &c, fixed_iterations);
    for (int d = 0; d < nv; d++) { w->s_prev_grad[d] = w->s_grad[d]; w->s_prev_Mgrad[d] = w->s_Mgrad[d]; }
    FN(update_constraint)(M, w, &c);
    FN(update_gradient)(M, w);
    if (m->solver == SOL_NEWTON) {
      for (int d = 0; d < nv; d++) w->s_search[d] = -w->s_Mgrad[d];
    } else {
      REAL num = 0, den = 0;
      for (int d = 0; d < nv; d++) { num += w->s_grad[d] * (w->s_Mgrad[d] - w->s_prev_Mgrad[d]); den += w->s_prev_grad[d] * w->s_prev_Mgrad[d]; }
      REAL beta = num / (den > (REAL)mjMINVAL ? den : (REAL)mjMINVAL);
      beta = beta > 0 ? beta : 0;
      for (int d = 0; d < nv; d++) w->s_search[d] = -w->s_Mgrad[d] + beta * w->s_search[d];
    }
    c.niter++;
  }
  for (int d = 0; d < nv; d++) { w->qacc[d] = w->s_qacc[d]; w->qacc_warmstart[d] = w->s_qacc[d]; w->qfrc_constraint[d] = w->s_qfrc[d]; }
  for (int r = 0; r < nefc; r++) w->efc_force[r] = w->s_force[r];
}

/* ---- forward (forward.py:373-401) ----------------------------------------------------------------- */
/* ---- sensors (sensor.py:56-440, ray.py:28-373) ------------------------------------------------------- */
/* Ray intersections run in the Data dtype.  The reference keeps the geom sizes of its ray tables in float64 (ray.py:317); with float32 Data its own call
   raises on the mixed dot products, so float32 + rangefinder has no reference result: this build's choice is float32 throughout (mjh_sensor.h). */
static REAL FN(ray_safe_div)(REAL num, REAL den) { return num / (den + (den == 0 ? (REAL)(float)mjMINVAL : (REAL)0)); }
static void FN(ray_quad)(REAL a, REAL b, REAL c, REAL* x0, REAL* x1) { /* :28-40 */
  REAL det = b * b - a * c, det2 = R_SQRT(det);
  REAL r0 = FN(ray_safe_div)(-b - det2, a), r1 = FN(ray_safe_div)(-b + det2, a);
  *x0 = ((det < (REAL)mjMINVAL) || (r0 < 0)) ? INFINITY : r0;
  *x1 = ((det < (REAL)mjMINVAL) || (r1 < 0)) ? INFINITY : r1;
}
static REAL FN(ray_dot3)(const REAL* a, const REAL* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static REAL FN(ray_geom)(int type, const REAL* size, const REAL* pnt, const REAL* vec) {
  if (type == 0) { /* plane :43-57 */
    REAL x = -FN(ray_safe_div)(pnt[2], vec[2]);
    int valid = (vec[2] <= -(REAL)mjMINVAL) && (x >= 0);
    for (int i = 0; i < 2; i++) { REAL p = pnt[i] + x * vec[i]; valid = valid && ((size[i] <= 0) || (R_FABS(p) <= size[i])); }
    return valid ? x : INFINITY;
  }
  if (type == 2) { /* sphere :60-69 */
    REAL x0, x1;
    FN(ray_quad)(FN(ray_dot3)(vec, vec), FN(ray_dot3)(vec, pnt), FN(ray_dot3)(pnt, pnt) - size[0] * size[0], &x0, &x1);
    return isinf(x0) ? x1 : x0;
  }
  if (type == 3) { /* capsule :72-106 */
    REAL a = vec[0] * vec[0] + vec[1] * vec[1], b = vec[0] * pnt[0] + vec[1] * pnt[1], c = (pnt[0] * pnt[0] + pnt[1] * pnt[1]) - size[0] * size[0];
    REAL x0, x1;
    FN(ray_quad)(a, b, c, &x0, &x1);
    REAL x = isinf(x0) ? x1 : x0;
    x = (R_FABS(pnt[2] + x * vec[2]) <= size[1]) ? x : INFINITY;
    for (int cap = 0; cap < 2; cap++) {
      REAL dif[3] = {pnt[0], pnt[1], cap == 0 ? pnt[2] - size[1] : pnt[2] + size[1]};
      FN(ray_quad)(FN(ray_dot3)(vec, vec), FN(ray_dot3)(vec, dif), FN(ray_dot3)(dif, dif) - size[0] * size[0], &x0, &x1);
      if (cap == 0) {
        if ((pnt[2] + x0 * vec[2] >= size[1]) && (x0 < x)) x = x0;
        if ((pnt[2] + x1 * vec[2] >= size[1]) && (x1 < x)) x = x1;
      } else {
        if ((pnt[2] + x0 * vec[2] <= -size[1]) && (x0 < x)) x = x0;
        if ((pnt[2] + x1 * vec[2] <= -size[1]) && (x1 < x)) x = x1;
      }
    }
    return x;
  }
  if (type == 4) { /* ellipsoid :109-129 */
    REAL s[3], sv[3], sp[3];
    for (int i = 0; i < 3; i++) { s[i] = FN(ray_safe_div)(1, size[i] * size[i]); sv[i] = s[i] * vec[i]; sp[i] = s[i] * pnt[i]; }
    REAL x0, x1;
    FN(ray_quad)(FN(ray_dot3)(sv, vec), FN(ray_dot3)(sv, pnt), FN(ray_dot3)(sp, pnt) - 1, &x0, &x1);
    return isinf(x0) ? x1 : x0;
  }
  if (type == 5) { /* cylinder :235-268 */
    REAL a = vec[0] * vec[0] + vec[1] * vec[1], b = vec[0] * pnt[0] + vec[1] * pnt[1], c = (pnt[0] * pnt[0] + pnt[1] * pnt[1]) - size[0] * size[0];
    REAL x0, x1;
    FN(ray_quad)(a, b, c, &x0, &x1);
    REAL x = isinf(x0) ? x1 : x0;
    x = (R_FABS(pnt[2] + x * vec[2]) <= size[1]) ? x : INFINITY;
    for (int cap = 0; cap < 2; cap++) {
      REAL t = FN(ray_safe_div)((cap == 0 ? size[1] : -size[1]) - pnt[2], vec[2]);
      REAL p0 = pnt[0] + t * vec[0], p1 = pnt[1] + t * vec[1];
      if ((t >= 0) && (p0 * p0 + p1 * p1 <= size[0] * size[0]) && (t < x)) x = t;
    }
    return x;
  }
  if (type == 6) { /* box :132-161 */
    static const int iface[6][2] = {{1, 2}, {0, 2}, {0, 1}, {1, 2}, {0, 2}, {0, 1}};
    REAL best = INFINITY;
    for (int f = 0; f < 6; f++) {
      int ax = f % 3;
      REAL x = f < 3 ? FN(ray_safe_div)(size[ax] - pnt[ax], vec[ax]) : -FN(ray_safe_div)(size[ax] + pnt[ax], vec[ax]);
      REAL p0 = pnt[iface[f][0]] + x * vec[iface[f][0]], p1 = pnt[iface[f][1]] + x * vec[iface[f][1]];
      int valid = (R_FABS(p0) <= size[iface[f][0]]) && (R_FABS(p1) <= size[iface[f][1]]) && (x >= 0);
      if (valid && x < best) best = x;
    }
    return best;
  }
  return INFINITY;
}


static REAL FN(sensor_cut)(REAL v, REAL cutoff, int datatype) { /* _apply_cutoff :41-53 */
  if (!(cutoff > 0)) return v;
  if (datatype == 0) return v < -cutoff ? -cutoff : (v > cutoff ? cutoff : v);
  if (datatype == 1) return v < cutoff ? v : cutoff;
  return v;
}
/* frame of an object of a frame sensor (sensor.py:62-74): position and orientation by mjtObj */
static void FN(sns_frame)(const FN(MjoWork) * w, int objtype, int id, const REAL** pos, const REAL** mat) {
  static const REAL zero3[3] = {0, 0, 0}, eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  switch (objtype) {
    case 1: *pos = w->xipos + 3 * id; *mat = w->ximat + 9 * id; break;
    case 2: *pos = w->xpos + 3 * id; *mat = w->xmat + 9 * id; break;
    case 5: *pos = w->geom_xpos + 3 * id; *mat = w->geom_xmat + 9 * id; break;
    case 6: *pos = w->site_xpos + 3 * id; *mat = w->site_xmat + 9 * id; break;
    case 7: *pos = w->cam_xpos + 3 * id; *mat = w->cam_xmat + 9 * id; break;
    default: *pos = zero3; *mat = eye; break;
  }
}
/* orientation of such an object as a quaternion (sensor.py:164-181) */
static void FN(sns_quat)(const FN(MjoModel) * M, const FN(MjoWork) * w, int objtype, int id, int body, REAL* q) {
  switch (objtype) {
    case 2: for (int i = 0; i < 4; i++) q[i] = w->xquat[4 * id + i]; break;
    case 1: FN(quat_mul)(w->xquat + 4 * id, M->body_iquat + 4 * id, q); break;
    case 5: FN(quat_mul)(w->xquat + 4 * body, M->geom_quat + 4 * id, q); break;
    case 6: FN(quat_mul)(w->xquat + 4 * body, M->site_quat + 4 * id, q); break;
    case 7: FN(quat_mul)(w->xquat + 4 * body, M->cam_quat + 4 * id, q); break;
    default: q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; break;
  }
}
/* value of sensor s, component comp */
static REAL FN(sensor_value)(const FN(MjoModel) * M, const FN(MjoWork) * w, int s, int comp) {
  const mjhModelDesc* m = M->d;
  int type = m->sns_type[s], obj = m->sns_objid[s], body = m->sns_bodyid[s], root = m->sns_rootid[s];
  if (type == 9) return w->qpos[obj];  /* jointpos */
  if (type == 10) return w->qvel[obj]; /* jointvel */
  if (type == 11) return w->ten_length[obj];         /* tendonpos :111-112 */
  if (type == 12) return w->ten_velocity[obj];       /* tendonvel :254-255 */
  if (type == 13) return w->actuator_length[obj];    /* actuatorpos :113-114 */
  if (type == 14) return w->actuator_velocity[obj];  /* actuatorvel :256-257 */
  if (type == 15) return w->actuator_force[obj];     /* actuatorfrc :417-418 */
  if (type == 16) return w->qfrc_actuator[obj];      /* jointactuatorfrc :419-420 */
  if (type == 17) { /* tendonactuatorfrc :421-423: force_mask @ actuator_force, the mask picks the actuators acting on the tendon */
    REAL acc = 0;
    for (int i = 0; i < m->nu; i++) acc += (REAL)(m->act_trntype[i] == 3 && m->act_trnid[i] == obj) * w->actuator_force[i];
    return acc;
  }
  if (type == 18) { /* ballquat :115-118 */
    REAL q[4] = {w->qpos[obj], w->qpos[obj + 1], w->qpos[obj + 2], w->qpos[obj + 3]};
    FN(normalize_n)(q, 4);
    return q[comp];
  }
  if (type == 19) return w->qvel[obj + comp]; /* ballangvel :258-260 */
  if (type == 35) return w->subtree_com[3 * obj + comp]; /* subtreecom :211-213 */
  if (type == 36) return w->x_subtree_linvel ? w->x_subtree_linvel[3 * obj + comp] : (REAL)0; /* :261-263: a leaf no stage writes (smooth.subtree_vel does not exist in the reference) */
  if (type == 37) return w->x_subtree_angmom ? w->x_subtree_angmom[3 * obj + comp] : (REAL)0; /* :264-266 */
  if (type == 45) return w->time[0]; /* clock :214-215 */
#define ROT_T(R_, v, o) for (int i_ = 0; i_ < 3; i_++) (o)[i_] = (R_)[i_] * (v)[0] + (R_)[3 + i_] * (v)[1] + (R_)[6 + i_] * (v)[2];
  if (type >= 26 && type <= 32) { /* frame sensors: object (objtype, obj) seen from the reference object (reftype, refid) or, without one, from the world */
    int ot = m->sns_objtype[s], rt = m->sns_reftype[s], rid = m->sns_refid[s], rbody = m->sns_refbodyid[s], rroot = m->sns_refrootid[s];
    const REAL *xpos, *xmat, *rpos, *rmat;
    FN(sns_frame)(w, ot, obj, &xpos, &xmat);
    FN(sns_frame)(w, rid >= 0 ? rt : 0, rid >= 0 ? rid : 0, &rpos, &rmat);
    if (type == 26) { /* framepos :119-138 */
      if (rid < 0) return xpos[comp];
      REAL d3[3] = {xpos[0] - rpos[0], xpos[1] - rpos[1], xpos[2] - rpos[2]}, o[3];
      ROT_T(rmat, d3, o)
      return o[comp];
    }
    if (type >= 28 && type <= 30) { /* frame{x,y,z}axis :139-160 */
      int k = type - 28;
      REAL axis[3] = {xmat[k], xmat[3 + k], xmat[6 + k]}, o[3];
      if (rid < 0) return axis[comp];
      ROT_T(rmat, axis, o)
      return o[comp];
    }
    if (type == 27) { /* framequat :161-199 */
      REAL q[4], r[4], ri[4], o[4];
      FN(sns_quat)(M, w, ot, obj, body, q);
      if (rid < 0) return q[comp];
      FN(sns_quat)(M, w, rt, rid, rbody, r);
      ri[0] = r[0] * (REAL)1; ri[1] = r[1] * (REAL)-1; ri[2] = r[2] * (REAL)-1; ri[3] = r[3] * (REAL)-1; /* quat_inv :264-273 */
      FN(quat_mul)(ri, q, o);
      return o[comp];
    }
    /* framelinvel / frameangvel :267-328 */
    const REAL *cv = w->cvel + 6 * body, *cvr = w->cvel + 6 * rbody;
    if (type == 32) {
      if (rid < 0) return cv[comp];
      REAL rel[3] = {cv[0] - cvr[0], cv[1] - cvr[1], cv[2] - cvr[2]}, o[3];
      ROT_T(rmat, rel, o)
      return o[comp];
    }
    const REAL *sc = w->subtree_com + 3 * root, *scr = w->subtree_com + 3 * rroot;
    REAL off[3] = {xpos[0] - sc[0], xpos[1] - sc[1], xpos[2] - sc[2]}, c[3], xl[3];
    FN(cross3)(off, cv, c);
    for (int i = 0; i < 3; i++) xl[i] = cv[3 + i] - c[i];
    if (rid < 0) return xl[comp];
    REAL offr[3] = {rpos[0] - scr[0], rpos[1] - scr[1], rpos[2] - scr[2]}, cr[3], xlr[3], rvec[3] = {xpos[0] - rpos[0], xpos[1] - rpos[1], xpos[2] - rpos[2]}, cw[3], rel[3], o[3];
    FN(cross3)(offr, cvr, cr);
    for (int i = 0; i < 3; i++) xlr[i] = cvr[3 + i] - cr[i];
    FN(cross3)(rvec, cvr, cw);
    for (int i = 0; i < 3; i++) rel[i] = (xl[i] - xlr[i]) + cw[i];
    ROT_T(rmat, rel, o)
    return o[comp];
  }
  const REAL* rot = w->site_xmat + 9 * obj;
  const REAL* pos = w->site_xpos + 3 * obj;
  if (type == 6) { /* magnetometer :92-94 */
    REAL mg[3] = {(REAL)m->magnetic_x, (REAL)m->magnetic_y, (REAL)m->magnetic_z}, o[3];
    ROT_T(rot, mg, o)
    return o[comp];
  }
  if (type == 7) { /* rangefinder: ray along the site's z axis (sensor.py:94-108, ray.py:327-372) */
    REAL vec[3] = {rot[2], rot[5], rot[8]};
    double best = INFINITY;
    for (int q = m->sns_rfadr[s]; q < m->sns_rfadr[s + 1]; q++) {
      int g = m->rf_geom[q];
      const REAL *gm = w->geom_xmat + 9 * g, *gp = w->geom_xpos + 3 * g;
      REAL d3[3] = {pos[0] - gp[0], pos[1] - gp[1], pos[2] - gp[2]}, lp[3], lv[3];
      for (int i = 0; i < 3; i++) { lp[i] = gm[i] * d3[0] + gm[3 + i] * d3[1] + gm[6 + i] * d3[2]; lv[i] = gm[i] * vec[0] + gm[3 + i] * vec[1] + gm[6 + i] * vec[2]; }
      REAL size[3] = {M->geom_size[3 * g], M->geom_size[3 * g + 1], M->geom_size[3 * g + 2]};
      double x = (double)FN(ray_geom)(m->geom_type[g], size, lp, lv);
      if (x < best) best = x;
    }
    return isinf(best) ? (REAL)-1 : (REAL)best;
  }
  const REAL* sc = w->subtree_com + 3 * root;
  REAL dif[3] = {pos[0] - sc[0], pos[1] - sc[1], pos[2] - sc[2]};
  if (type == 4 || type == 5) { /* force :399-406, torque :407-416: from Data.cfrc_int, which no stage of the reference writes (smooth.rne_postconstraint does not exist there) */
    REAL fz[6] = {0, 0, 0, 0, 0, 0}, o[3];
    const REAL* cf = w->x_cfrc_int ? w->x_cfrc_int + 6 * body : fz;
    if (type == 4) { ROT_T(rot, cf + 3, o) return o[comp]; }
    REAL c[3], v[3];
    FN(cross3)(dif, cf + 3, c);
    for (int i = 0; i < 3; i++) v[i] = cf[i] - c[i];
    ROT_T(rot, v, o)
    return o[comp];
  }
  const REAL* cvel = w->cvel + 6 * body;
  if (type == 3) { REAL o[3]; ROT_T(rot, cvel, o) return o[comp]; } /* gyro :246-251 */
  REAL c[3], v[3], lin[3];
  FN(cross3)(dif, cvel, c);
  for (int i = 0; i < 3; i++) v[i] = cvel[3 + i] - c[i];
  ROT_T(rot, v, lin)
  if (type == 2) return lin[comp]; /* velocimeter :235-245 */
  /* accelerometer :379-399 with Data.cacc, which no stage of the reference ever writes (the caller's leaf: zeros from make_data) */
  REAL ang[3], zero[6] = {0, 0, 0, 0, 0, 0}, ca[3], av[3], acc[3], corr[3];
  const REAL* cacc = w->x_cacc ? w->x_cacc + 6 * body : zero;
  ROT_T(rot, cvel, ang)
  FN(cross3)(dif, cacc, ca);
  for (int i = 0; i < 3; i++) av[i] = cacc[3 + i] - ca[i];
  ROT_T(rot, av, acc)
  FN(cross3)(ang, lin, corr);
#undef ROT_T
  return (acc[comp] + corr[comp]) + 0; /* + gravity term, zero for mujoco >= 3.3.7 (sensor.py:36-38) */
}
static void FN(sensors)(const FN(MjoModel) * M, FN(MjoWork) * w) {
  const mjhModelDesc* m = M->d;
  for (int k = 0; k < m->nsensordata; k++) {
    int s = m->slot_sensor[k];
    if (s < 0) continue; /* slot keeps the caller's value */
    w->sensordata[k] = FN(sensor_cut)(FN(sensor_value)(M, w, s, k - m->sns_adr[s]), M->sns_cutoff[s], m->sns_datatype[s]);
  }
}

static void FN(forward_env)(const FN(MjoModel) * M, FN(MjoWork) * w, int stages, int flags, int with_cams) {
  const mjhModelDesc* m = M->d;
  if (stages & 0x7f) { FN(kinematics)(M, w, with_cams); FN(com_pos)(M, w); if (m->ntendon > 0) FN(tendon)(M, w); }
  if (stages & 0x7e) FN(crb_factor)(M, w);
  if (stages & 0x7c) { if (m->ncon > 0) FN(collision)(M, w); }
  if (stages & 0x78) FN(make_constraint)(M, w);
  if (stages & 0x70) FN(velocity)(M, w);
  if (stages & 0x60) FN(actuation)(M, w);
  if (stages & 0x40) {
    if (m->nefc == 0) { for (int d = 0; d < m->nv; d++) w->qacc[d] = w->qacc_smooth[d]; }
    else FN(solve)(M, w, flags & MJH_FLAG_FIXED_ITERATIONS);
    if (with_cams && m->nsensor > 0) FN(sensors)(M, w); /* the returned Data carries the sensors of its own forward pass (RK4: stage 0) */
  }
}

/* ---- integrators (forward.py:231-370) ---------------------------------------------------------------- */
static void FN(integrate_pos)(const FN(MjoModel) * M, const REAL* qpos, const REAL* qvel, REAL dt, REAL* out) { /* :231-252 */
  const mjhModelDesc* m = M->d;
  for (int j = 0; j < m->njnt; j++) {
    int t = m->jnt_type[j], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (t == JNT_FREE) {
      for (int i = 0; i < 3; i++) out[qa + i] = qpos[qa + i] + dt * qvel[da + i];
      FN(quat_integrate)(qpos + qa + 3, qvel + da + 3, dt, out + qa + 3);
    } else if (t == JNT_BALL) {
      FN(quat_integrate)(qpos + qa, qvel + da, dt, out + qa);
    } else {
      out[qa] = qpos[qa] + dt * qvel[da];
    }
  }
}

static void FN(advance)(const FN(MjoModel) * M, FN(MjoWork) * w, const REAL* qpos0, const REAL* qvel0, const REAL* act0, REAL time0,
                        const REAL* act_dot, const REAL* qacc, const REAL* qvel_for_pos) { /* _advance :255-310 */
  const mjhModelDesc* m = M->d;
  REAL dt = M->timestep;
  for (int i = 0; i < m->nu; i++) {
    int dyn = m->act_dyntype[i];
    if (dyn == DYN_NONE) continue;
    int a = m->act_actadr[i];
    REAL act = act0[a];
    if (dyn == DYN_FILTEREXACT) {
      REAL tau = M->act_dynprm[3 * i];
      tau = tau > (REAL)mjMINVAL ? tau : (REAL)mjMINVAL;
      act = act + act_dot[a] * tau * (1 - R_EXP(-dt / tau));
    } else {
      act = act + act_dot[a] * dt;
    }
    if (m->act_actlimited[i]) {
      REAL lo = M->act_actrange[2 * i], hi = M->act_actrange[2 * i + 1];
      act = act < lo ? lo : (act > hi ? hi : act);
    }
    w->act[a] = act;
  }
  for (int d = 0; d < m->nv; d++) w->tmp_nv[d] = qvel0[d] + qacc[d] * dt;
  const REAL* vp = qvel_for_pos ? qvel_for_pos : w->tmp_nv;
  FN(integrate_pos)(M, qpos0, vp, dt, w->tmp_nefc); /* tmp_nefc is sized >= nq */
  for (int i = 0; i < m->nq; i++) w->qpos[i] = w->tmp_nefc[i];
  for (int d = 0; d < m->nv; d++) w->qvel[d] = w->tmp_nv[d];
  w->time[0] = time0 + dt;
}
