// mjh_kernels.h -- the per-environment step pipeline (one wavefront = one environment, five phase kernels).
//
// Pipeline per environment (reference mujoco_torch/_src/forward.py:373-401, 463-496):
//   PH_KIN  check_state -> kinematics -> com_pos
//   PH_CRB  crb -> make_m -> factor_m
//   PH_CON  collision -> make_constraint
//   PH_VEL  transmission/_velocity (com_vel, passive, rne) -> _actuation -> _acceleration
//   PH_SOL  solve -> Euler advance / RK4 stage bookkeeping
// Each phase is its own launch of the same kernel template: it loads the Data leaves it consumes from HBM/L2
// (they are leaves the previous phases had to write anyway) into a small LDS arena, computes, and streams the
// leaves it produces back as contiguous batch-major rows.  Splitting by phase keeps every arena small
// (6..30 KB per environment instead of 51 KB for the fused pipeline), which is what buys occupancy: the
// per-environment work is a chain of dependent small ops, so waves-in-flight per SIMD hide the latency.
// Data-parallel axes: bodies / joints / dofs / geoms / contact pairs / constraint rows -> lanes;
// tree recursions -> each lane walks its own ancestor chain; dot products -> per-lane partials + wave all-reduce.
#pragma once
#include <cstring>
#include "mjh_device.h"

// view of the arena: offsets (in REALs) come with the launch (kernarg, scalar loads); the view only carries the
// base pointer, so it stays in registers.
template <typename REAL>
struct LdsView {
  REAL* base;
  const LdsOff* off;
#define X(n, c, p) __device__ __forceinline__ REAL* n() const { return base + off->n; }
  MJH_LDS_ARRAYS(X, _)
#undef X
};

// Leaves nobody reads again inside the step (frames, spatial inertias, contact geometry and constants, zero rows of efc_J, forces ...) are stored with the non-temporal hint: written as
// ordinary stores the ant's stage-0 constraint phase (287 MB of outputs) pushed what the NEXT kernels read out of L2 / the memory-side cache -- kernel 13 of RK4 stage 1 ran 64.8 us
// against 50.5 in stages 2 / 3; with the hint 57.1 (ant 23.9 -> 24.5 M env-steps/s; profiles/r05/notes.md).  -DMJH_NO_NT turns the hint off.
#ifndef MJH_NO_NT
#define MJH_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define MJH_NT_STORE(v, p) (*(p) = (v))
#endif
// carve the arena of one phase on the host; returns the number of REALs used.
template <typename MM>
inline int lds_carve(const MM& m, int phase_bit, LdsOff& o, int* kv_defer_ok = nullptr) {
  int off = 0;
#define X(n, c, p) o.n = off; if ((p) & phase_bit) off += (((c) + 1) & ~1);
  MJH_LDS_ARRAYS(X, m)
#undef X
  if (phase_bit == PH_CON) {
    // the geom frames are only read by the narrow phase, the dense efc_J only written after it: they share storage
    // (unless the frames are the larger of the two, then they get their own)
    const int n3 = ((3 * m.ngeom + 1) & ~1), n9 = ((9 * m.ngeom + 1) & ~1), nj = (((m.con_general ? m.nefc * m.nv : (m.con_direct ? 0 : (m.nefc - m.nl) * m.nv)) + 1) & ~1);
    if (n3 + n9 <= nj) { o.geom_xpos = o.efc_J; o.geom_xmat = o.efc_J + n3; }
    else { o.geom_xpos = off; o.geom_xmat = off + n3; off += n3 + n9; }
  }
  if (phase_bit == PH_SOL2) {
    // [parked state | solve arrays]; the integrator tail starts once the solve is over: its arrays are carved again over the solve arrays
    LdsOff a, t;
    int p0 = 0;
#define X(n, c, p) a.n = p0; if ((p) & PH_SOL2P) p0 += (((c) + 1) & ~1);
    MJH_LDS_ARRAYS(X, m)
#undef X
    int o1 = p0, o2 = p0;
#define X(n, c, p) if ((p) & PH_SOL2P) o.n = a.n; else { o.n = o1; if ((p) & PH_SOL2) o1 += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
#define X(n, c, p) t.n = o2; if ((p) & PH_SOL2T) o2 += (((c) + 1) & ~1);
    MJH_LDS_ARRAYS(X, m)
#undef X
#define X(n, c, p) if ((p) & PH_SOL2T) o.n = t.n;
    MJH_LDS_ARRAYS(X, m)
#undef X
    off = o1 > o2 ? o1 : o2;
  }
  if (phase_bit == PH_KINVEL) {
    // fused kinematics + velocity kernel: what both phases use first (it stays resident across the two), then the frames only the kinematics
    // touches and the spatial quantities only the velocity phase touches from one common offset -- the former are stored and dead when the
    // latter are first written (a wave-wide sync separates the phases)
    int s0 = 0;
#define X(n, c, p) if (((p) & PH_KIN) && ((p) & PH_VEL)) { o.n = s0; s0 += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    // Within the overlaid part: the kinematics stage's scratch arrays (never stored) lead its side, the arrays the velocity stage writes FIRST (its inputs and the
    // transmission's outputs) lead the other -- so the frame leaves stay intact in LDS until the velocity stage stores them in front of its first sweep (velocity(), kv_defer)
    int ka = s0, va = s0;
    auto early_k = [](const char* n) { return !strcmp(n, "jquat") || !strcmp(n, "sub_mass") || !strcmp(n, "sub_pos"); };
    auto early_v = [](const char* n) { return !strcmp(n, "qvel") || !strcmp(n, "act") || !strcmp(n, "act_length") || !strcmp(n, "act_velocity") || !strcmp(n, "act_rot"); };
#define X(n, c, p) if (((p) & PH_KIN) && !((p) & PH_VEL) && early_k(#n)) { o.n = ka; ka += (((c) + 1) & ~1); } else if (!((p) & PH_KIN) && ((p) & PH_VEL) && early_v(#n)) { o.n = va; va += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    if (kv_defer_ok) *kv_defer_ok = (va <= ka) ? 1 : 0;
#define X(n, c, p) if (((p) & PH_KIN) && !((p) & PH_VEL) && !early_k(#n)) { o.n = ka; ka += (((c) + 1) & ~1); } else if (!((p) & PH_KIN) && ((p) & PH_VEL) && !early_v(#n)) { o.n = va; va += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    off = ka > va ? ka : va;
  }
  if (phase_bit == PH_KCV) {
    // kinematics -> crb / factor -> velocity in one kernel: the crb stage runs between the two others on the arrays they share (cinert, cdof); its own
    // arrays, the kinematics-only frames (stored by then) and the velocity-only quantities (not yet written) start from one common offset.  The factor is
    // written from registers over the crb stage's own arrays (n <= 32), past the shared part
    int s0 = 0;
#define X(n, c, p) if (((p) & PH_KIN) && ((p) & PH_VEL)) { o.n = s0; s0 += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    int ka = s0, ca = s0, va = s0;
#define X(n, c, p) if (((p) & PH_KIN) && !((p) & PH_VEL)) { o.n = ka; ka += (((c) + 1) & ~1); } else if (!((p) & PH_KIN) && ((p) & PH_VEL)) { o.n = va; va += (((c) + 1) & ~1); } else if (((p) & PH_CRB) && !((p) & (PH_KIN | PH_VEL))) { o.n = ca; ca += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    const int nn = ((m.nv * m.nv + 1) & ~1);
    if (m.nv <= 32) { o.qLD = s0; if (ca < s0 + nn) ca = s0 + nn; }
    else { o.qLD = ca; ca += nn; }
    off = ka > va ? ka : va;
    if (ca > off) off = ca;
  }
  if (phase_bit == PH_KCV2) {
    // two-wave kinematics -> {velocity || crb / factor}: [what kinematics and velocity share | kinematics-only OVER velocity-only (as PH_KINVEL) | crb-only arrays, the factor over them]
    // -- the crb wave reads cinert / cdof of the shared part and writes nothing outside its own region, so the velocity wave's arrays are never under it
    int s0 = 0;
#define X(n, c, p) if (((p) & PH_KIN) && ((p) & PH_VEL)) { o.n = s0; s0 += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    int ka = s0, va = s0;
#define X(n, c, p) if (((p) & PH_KIN) && !((p) & PH_VEL)) { o.n = ka; ka += (((c) + 1) & ~1); } else if (!((p) & PH_KIN) && ((p) & PH_VEL)) { o.n = va; va += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    const int c0 = ka > va ? ka : va;
    int ca = c0;
#define X(n, c, p) if (((p) & PH_CRB) && !((p) & (PH_KIN | PH_VEL))) { o.n = ca; ca += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    const int nn = ((m.nv * m.nv + 1) & ~1);
    if (m.nv <= 32) { o.qLD = c0; if (ca < c0 + nn) ca = c0 + nn; }
    else { o.qLD = ca; ca += nn; }
    off = ca;
  }
  if (phase_bit == PH_CS) {
    // Fused constraint + register-solver kernel (plain rows, one contact condim, one dense-row slot per lane, no tiers: the humanoid).  Laid out by hand:
    //   [qvel | act | act_dot | con_dist | i_con_act | i_crow_act]     live through both stages (the two int tables: active contacts, contact -> compact slot)
    //   [efc_Jc]                                                       the rows of the ACTIVE contacts, built in compact order by the constraint stage (the geom frames of the
    //                                                                  narrow phase are dead before the rows are built and sit under them, as in PH_CON)
    //   constraint-only [subtree_com cdof con_pos con_frame]           OVER
    //   solver-only     [efc_D efc_aref | r_vs r_vs2 r_pg r_fs qpos]   (efc_D / efc_aref of the active contacts' rows, compact order: staged by the aref loop, which no longer reads what lies under them)
    //   integrator tail (PH_SOL2T) over the rows, as in PH_SOL2.
    // The single-column limit rows travel in registers (lane r <-> row r in both stages).  Returns -1 when the overlays do not work out (the caller keeps the two launches).
    auto ev = [](int c) { return (c + 1) & ~1; };
    const int nd = m.nefc - m.nl;
    off = 0;
    o.qvel = off; off += ev(m.nv);
    o.act = off; off += ev(m.na);
    o.act_dot = off; off += ev(m.na);
    o.con_dist = off; off += ev(m.ncand);
    o.i_con_act = off; off += ev(m.ncon);   // (ints: one REAL each is enough in both dtypes)
    o.i_crow_act = off; off += ev(m.ncon);
    const int rows0 = off, nj = ev(nd * m.nv), n3 = ev(3 * m.ngeom), n9 = ev(9 * m.ngeom);
    o.efc_J = o.efc_Jc = rows0;
    if (n3 + n9 > nj) return -1;
    o.geom_xpos = rows0; o.geom_xmat = rows0 + n3;
    off += nj;
    const int ov = off;
    int ca = ov;
    o.subtree_com = ca; ca += ev(3 * m.nbody);
    o.cdof = ca; ca += ev(6 * m.nv);
    o.con_pos = ca; ca += ev(3 * m.ncand);
    o.con_frame = ca; ca += ev(9 * m.ncand);
    int sa = ov;
    o.efc_D = sa; sa += ev(nd);
    o.efc_aref = sa; sa += ev(nd);
    o.r_vs = sa; sa += ev(m.nv);
    o.r_vs2 = sa; sa += ev(m.nv);
    o.r_pg = sa; sa += ev(2 * m.nv);
    o.r_fs = sa; sa += ev(nd + m.nl);
    const int qpos_at = sa;
    o.qpos = sa; sa += ev(m.nq);
    off = ca > sa ? ca : sa;
    int ta = rows0;
#define X(n, c, p) if ((p) & PH_SOL2T) { o.n = ta; ta += (((c) + 1) & ~1); }
    MJH_LDS_ARRAYS(X, m)
#undef X
    if (ta > qpos_at) return -1;  // (the tail must leave the parked qpos alone)
  }
  if (phase_bit == PH_CRB) {
    // the Cholesky factor is produced from registers after every other array of the phase is dead: it is written
    // over them (n <= 32, register factorisation); the in-LDS factorisation of larger models gets its own space
    const int nn = ((m.nv * m.nv + 1) & ~1);
    if (m.nv <= 32) { o.qLD = 0; if (off < nn) off = nn; }
    else { o.qLD = off; off += nn; }
  }
  return off;
}

template <typename REAL>
struct DevData {  // typed view of mjhData
#define X(n) REAL* n;
  MJH_DATA_REALS(X)
#undef X
#define X(n) int32_t* n;
  MJH_DATA_I32(X)
#undef X
#define X(n) int64_t* n;
  MJH_DATA_I64(X)
#undef X
#define X(n) const REAL* n;
  MJH_DATA_EXTRA_IN(X)
#undef X
};

// per-environment RK4 bookkeeping rows kept in the caller's workspace ([B, n] each)
template <typename REAL>
struct RkWork {
  REAL *qvel0, *act0, *kqvel, *sum_qvel, *sum_qacc, *sum_actdot;
};

// All launch parameters travel as ONE by-value kernel argument.  Device code reads them through the kernarg
// segment pointer (constant address space => scalar loads, nothing is copied to scratch).  That pointer only
// exists inside the kernel function itself, so everything below must inline into the kernel (build.sh checks
// the ISA for calls).
// the state leaves an advance writes (the only leaves of the "next stage" / "returned" Data the kernels touch)
template <typename REAL>
struct StatePtrs {
  REAL *qpos, *qvel, *act, *time;
};
template <typename REAL>
inline StatePtrs<REAL> state_of(const DevData<REAL>& d) { return StatePtrs<REAL>{d.qpos, d.qvel, d.act, d.time}; }

template <typename REAL>
struct KArgs {
  DevModel<REAL> M;
  LdsOff off;          // arena of this launch's phase
  LdsOff off2;         // whole-pass kernel (mjh_sol2_kernel<.., 34>): the arena layout of its second half (constraint stage + solver: PH_CS); `off` is its first half's (PH_KCV)
  LdsOff off3;         // stage kernel (mjh_sol2_kernel<.., 18>, one launch per RK4 stage of a small model): `off` = kernel 13's arena (four environments per wavefront), `off2` = the constraint phase's (kernel 8, two per wavefront), `off3` = the register solver's first tier
  int lds_reals2, lds_reals3;  // ... and the REALs between the arenas of a wavefront's environments in its second and third part
  int xswap_k;                 // whole-pass kernel, out of lockstep in the kinematics stage: workgroups of odd parity under this mask of their index run com_pos before the geom / site / camera frames (0: one order)
  int xswap_c;                 // stage kernel, out of lockstep in stage 0: workgroups of odd parity under this mask store the model-constant contact leaves (a third of what the constraint phase of stage 0 writes) at the kernel's HEAD instead of behind the constraint rows (0: all behind)
  int stage_parts;             // ... and which parts this launch runs: 1 = kinematics + crb / factor + velocity, 2 = collision + constraint rows, 4 = solver tier + integrator (7: a whole RK4 stage; 6: the tail of a pass behind the convex narrow phase)
  DevData<REAL> in;    // the caller's Data: external inputs (ctrl, applied forces, warm start) and stage-0 state
  DevData<REAL> cur;   // the Data being computed: `out` for a forward / RK stage 0, the workspace Data for RK stages 1..3
  StatePtrs<REAL> nxt; // where an RK stage writes the next stage's state (workspace Data)
  StatePtrs<REAL> fin; // the returned Data (final advance)
  RkWork<REAL> W;
  int64_t B;
  int64_t env_begin, env_count;  // environments of THIS launch (a packed launch covers an even count, a second one the odd tail)
  int lds_reals;                 // REALs between the arenas of the environments sharing a wavefront
  int flags;
  int stages;          // MJH_STAGE_* prefix mask
  int do_step;
  int rk_stage;        // -1: Euler / forward only; 0..3: RK4 stage
  int state_from_cur;  // qpos/qvel/act of this pass come from `cur` (RK stages >= 1) instead of `in`
  int sns_epw;          // sensor kernel: environments per wavefront
  REAL* hs;             // small models with one contact condim (DevModel::crow_by_con): per-environment hand-over of the constraint phase to the register solver (workspace, hs_reals each): [nda | contact -> compact slot (ncon) | efc_D (nd) | efc_aref (nd) | efc_J rows (nd * nv)] of the ACTIVE contacts' rows in compact order, so that the solver's loads are ONE round of fixed addresses instead of contact_dist -> compaction -> D / aref gather -> row gather (four dependent trips: 45 % of the ant's solver kernel).  NULL: the solver reads the leaves
  int hs_reals;
  REAL* cand;           // max_contact_points over convex pairs: candidate contacts of the convex narrow phase, [B, ncand] dist | [B, ncand, 3] pos | [B, ncand, 9] frame (workspace)
  const REAL* warm_src; // [B, nv] warm start of this pass: the caller's, or the previous RK stage's solution
  int it_cap, ls_cap;   // > 0: the register solver leaves an environment to the fallback launch (LDS solver, one environment per wavefront) once its solve has run it_cap Newton iterations or ls_cap line-search iterations: the long solves of a batch are few, and inside a shared wavefront every one of them holds three other environments' lanes
  int mark_leftover;    // register solver, first tier with a second one behind it: an environment with more active rows than this tier keeps gets mjh_bail_mark in out.qacc
  int scan_marks;       // register solver, second tier: waves scan 64 environments' marks each and serve the marked ones (instead of one wave per environment pair counting rows)
  int fallback_only;    // LDS solver launch: serve only the environments the register solver flagged (mjh_bail_mark in out.qacc)
  int row_lo, row_hi;   // register solver tiers: this launch serves the environments with row_lo < (dense rows of their active contacts) <= row_hi
  const int* sol_perm;  // register solver, four environments per wavefront: environment served by each (wave, lane group) slot, sorted by the previous step's iteration counts (NULL: identity)
  int* sol_key;         // ... and where this step's count of an environment goes (NULL: not recorded)
  unsigned long long* stamps;  // diagnostic builds (-DMJH_STAMPS): [B, 128] s_memtime stamps, else unused
};
template <typename REAL>
__device__ __forceinline__ const KArgs<REAL>& kargs() {
  typedef const KArgs<REAL> __attribute__((address_space(4))) * KPtr;
  return *(const KArgs<REAL>*)(KPtr)__builtin_amdgcn_kernarg_segment_ptr();
}
#define M (kargs<REAL>().M)
#define in (kargs<REAL>().in)
#define out (kargs<REAL>().cur)
#define KA (kargs<REAL>())
// In-kernel stamps (diagnostic build only, never in the shipped library): lane 0 records the shader clock at
// section boundaries into a buffer of its own; tools/stamps.py turns them into a per-section cycle profile.
#ifdef MJH_SOL2_CAPS
#define MJH_SOL2_CAPS_ON 1
#else
#define MJH_SOL2_CAPS_ON 0
#endif
#ifdef MJH_STAMPS
#ifdef MJH_STAMPS_STAGE  /* -DMJH_STAMPS_STAGE=2: only the launches of that RK4 stage are recorded */
#define MJH_STAMPS_STAGE_OK (KA.rk_stage == MJH_STAMPS_STAGE)
#else
#define MJH_STAMPS_STAGE_OK true
#endif
// each STAMP adds the shader-clock time since the previous STAMP of this environment's phase to its slot: sections inside loops
// accumulate over the iterations
#define STAMP(slot)                                                                                  \
  do {                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();                                            \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                              \
    if (KA.stamps && lane() == 0 && MJH_STAMPS_STAGE_OK) KA.stamps[e * 128 + (slot)] += t_ - stamp_prev; \
    stamp_prev = __builtin_amdgcn_s_memtime();                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)
#define STAMP0() do { stamp_prev = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP0() do {} while (0)
#endif

// the value is only known from here on, as far as the optimiser can tell (an environment index is wave-uniform for a whole-wave environment)
template <int W>
__device__ __forceinline__ void late_bind(int64_t& v) {
  if constexpr (W == MJH_WAVE) asm volatile("" : "+s"(v));
  else asm volatile("" : "+v"(v));
}
// ---- helpers: coalesced row copy between LDS and the environment's global row -------------------------------------------
// Four independent transfers are issued per trip so one HBM/L2 (or LDS) round trip covers 256 elements.
template <int W, bool NT = false, typename REAL>
__device__ __forceinline__ void row_store(REAL* g, const REAL* l, int n, int64_t env) {
  if (!g) return;
  // (the row's address is formed HERE: left to the optimiser, the address arithmetic of every leaf store of a phase is hoisted to the kernel's head,
  // spilled, and reloaded in front of the store -- and a scratch reload queues behind the stores already in flight: vmcnt is in order on gfx9)
  late_bind<W>(env);
  REAL* dst = g + env * n;
  int i = sub_lane<W>();
  asm volatile("" : "+v"(i));
  for (; i + 3 * W < n; i += 4 * W) {
    const REAL a = l[i], b = l[i + W], c = l[i + 2 * W], d = l[i + 3 * W];
    if (NT) { MJH_NT_STORE(a, &dst[i]); MJH_NT_STORE(b, &dst[i + W]); MJH_NT_STORE(c, &dst[i + 2 * W]); MJH_NT_STORE(d, &dst[i + 3 * W]); }
    else { dst[i] = a; dst[i + W] = b; dst[i + 2 * W] = c; dst[i + 3 * W] = d; }
  }
  for (; i < n; i += W) { if (NT) MJH_NT_STORE(l[i], &dst[i]); else dst[i] = l[i]; }
}
// global -> global copy of a model constant into a leaf (the source is L2-resident): eight requests in flight per trip
template <int W, typename REAL>
__device__ __forceinline__ void row_copy_const(REAL* g, const REAL* c, int n, int64_t env) {
  if (!g) return;
  late_bind<W>(env);
  REAL* dst = g + env * n;
  int i = sub_lane<W>();
  asm volatile("" : "+v"(i));
  for (; i + 15 * W < n; i += 16 * W) {  // (sixteen reads, then sixteen stores: a read behind a store waits for the store to land)
    REAL t[16];
#pragma unroll
    for (int q = 0; q < 16; q++) t[q] = c[i + q * W];
#pragma unroll
    for (int q = 0; q < 16; q++) MJH_NT_STORE(t[q], &dst[i + q * W]);
  }
  for (; i + 7 * W < n; i += 8 * W) {
    REAL t[8];
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = c[i + q * W];
#pragma unroll
    for (int q = 0; q < 8; q++) MJH_NT_STORE(t[q], &dst[i + q * W]);
  }
  for (; i + 3 * W < n; i += 4 * W) {
    const REAL a = c[i], b = c[i + W], cc = c[i + 2 * W], d = c[i + 3 * W];
    MJH_NT_STORE(a, &dst[i]); MJH_NT_STORE(b, &dst[i + W]); MJH_NT_STORE(cc, &dst[i + 2 * W]); MJH_NT_STORE(d, &dst[i + 3 * W]);
  }
  for (; i < n; i += W) dst[i] = c[i];
}
// K model constants into K leaves: the first T * W elements of every array are read before the first store (longer arrays finish with row_copy_const)
template <int W, int K, int T, typename REAL>
__device__ __forceinline__ void multi_copy_const(REAL* const (&dst)[K], const REAL* const (&src)[K], const int (&n)[K], int64_t env) {
  late_bind<W>(env);
  int l = sub_lane<W>();
  asm volatile("" : "+v"(l));
  REAL v[K][T];
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int t = 0; t < T; t++) { const int i = l + t * W; v[k][t] = (dst[k] && i < n[k]) ? src[k][i] : (REAL)0; }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int t = 0; t < T; t++) { const int i = l + t * W; if (dst[k] && i < n[k]) MJH_NT_STORE(v[k][t], &dst[k][env * n[k] + i]); }
  }
#pragma unroll
  for (int k = 0; k < K; k++)
    if (dst[k] && n[k] > T * W) {
      REAL* d = dst[k] + env * n[k];
      int i = l + T * W;
      for (; i + 3 * W < n[k]; i += 4 * W) {
        const REAL a = src[k][i], b = src[k][i + W], c = src[k][i + 2 * W], e2 = src[k][i + 3 * W];
        d[i] = a; d[i + W] = b; d[i + 2 * W] = c; d[i + 3 * W] = e2;
      }
      for (; i < n[k]; i += W) d[i] = src[k][i];
    }
}
template <int W, typename REAL>
__device__ __forceinline__ void row_load(REAL* l, const REAL* g, int n, int64_t env) {
  if (!g) { for (int i = sub_lane<W>(); i < n; i += W) l[i] = 0; return; }
  late_bind<W>(env);  // (as for the stores: addresses formed here, not hoisted to the kernel's head and spilled)
  const REAL* src = g + env * n;
  int i = sub_lane<W>();
  asm volatile("" : "+v"(i));
  for (; i + 3 * W < n; i += 4 * W) {
    const REAL a = src[i], b = src[i + W], c = src[i + 2 * W], d = src[i + 3 * W];
    l[i] = a; l[i + W] = b; l[i + 2 * W] = c; l[i + 3 * W] = d;
  }
  for (; i < n; i += W) l[i] = src[i];
}

// Several leaves of one environment into LDS with ALL their loads in flight before the first LDS store: consecutive row_load calls
// are separate loops, so each one pays a full L2 round trip before the next starts -- at a phase start that was a quarter of the
// phase.  The first T * W elements of each of the K arrays are requested up front (predicated), then stored; longer tails loop.
template <int W, int K, int T, typename REAL>
__device__ __forceinline__ void multi_load(REAL* const (&dst)[K], const REAL* const (&src)[K], const int (&n)[K], int64_t env) {
  late_bind<W>(env);
  int l = sub_lane<W>();
  asm volatile("" : "+v"(l));
  REAL v[K][T];
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int i = l + t * W;
      v[k][t] = (src[k] && i < n[k]) ? src[k][env * n[k] + i] : (REAL)0;
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int i = l + t * W;
      if (i < n[k]) dst[k][i] = v[k][t];
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++)
    for (int i = l + T * W; i < n[k]; i += W) dst[k][i] = src[k] ? src[k][env * n[k] + i] : (REAL)0;
}

// =====================================================================================================================
// dense Cholesky in LDS (math.small_cholesky :87-129).  Right-looking: once column j is final, every row
// subtracts its rank-1 contribution from the trailing columns.  Each element still receives its updates in
// the order k = 0, 1, ... -- the subtraction order of the reference's unrolled loop -- but the updates of one
// column are independent, so their LDS traffic pipelines instead of forming one long dependent chain.
// =====================================================================================================================
// element (i, k), k <= i, of a lower-triangular factor stored as a full n x n matrix or as packed rows
template <bool PACKED>
__device__ __forceinline__ int tri_at(int i, int k, int n) { return PACKED ? (i * (i + 1)) / 2 + k : i * n + k; }

// inverse of the packed index: w = i (i + 1) / 2 + j, 0 <= j <= i
// ---- 16x16x4 matrix-core tiles (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64) ------------------------------------------------------------
// The Newton Hessian H = M + J^T diag(D) J (solver.py:366-370) and qfrc_constraint = J^T efc_force are the only contractions of the step with a
// long summed index (the constraint rows); for nv <= 16 one tile holds the whole result.  Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] of a
// 16x4 . 4x16 block, i.e. ONE element J[row 4 b + (l >> 4)][column l & 15] feeds both operands.  The instruction accumulates in k order with one
// rounding per term (a fused-multiply-add chain): the rows are summed in index order like the scalar loops they replace.
// Float64 instantiations only.  In float32 the Newton iteration of the mesh scene is sensitive to the last bit of H and of J^T f (its stopping
// test works in rounding noise): the fused chain moved the HIP step 5.8e-5 -> 2.4e-3 away from the oracle's unfused sums on BASELINE config 5
// (profiles/r02/notes.md), for 4 % of that solver phase -- float32 keeps the counted VALU loops, whose operation order is the oracle's.
typedef float mjh_f32x4 __attribute__((ext_vector_type(4)));
typedef double mjh_f64x4 __attribute__((ext_vector_type(4)));
template <typename REAL> struct MfmaTile;
template <> struct MfmaTile<float> {
  typedef mjh_f32x4 Acc;
  static __device__ __forceinline__ Acc mac(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }  // C/D: column = lane & 15
};
template <> struct MfmaTile<double> {
  typedef mjh_f64x4 Acc;
  static __device__ __forceinline__ Acc mac(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }  // the f64 form has its own C/D map
};

__device__ __forceinline__ void tri_unpack(int w, int& i, int& j) {
  int r = (int)((__builtin_sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
  if ((r * (r + 1)) / 2 > w) r--;
  if (((r + 1) * (r + 2)) / 2 <= w) r++;
  i = r;
  j = w - (r * (r + 1)) / 2;
}

template <int W, bool APACKED, typename REAL>
__device__ __forceinline__ void chol_factor_lds(const REAL* A, REAL* L, int n) {
  const int i = sub_lane<W>();
  for (int w = i; w < n * n; w += W) {
    const int r = w / n, c = w - n * r;
    L[w] = (c <= r) ? A[tri_at<APACKED>(r, c, n)] : (REAL)0;
  }
  wave_sync();
  const bool big = n > INLINE_CHOL_MAX;
  for (int j = 0; j < n; j++) {
    REAL s = L[j * n + j];
    if (big) s = s + (REAL)1e-10;  // torch.linalg.cholesky(A + 1e-10 I), math.py:108-113
    const REAL d = big ? r_sqrt<REAL>(s) : r_sqrt<REAL>(s > (REAL)1e-12 ? s : (REAL)1e-12);
    if (n > W) {  // more rows than lanes (models beyond 64 dofs): each lane scales the column entries of ITS rows in place -- only the pivot is shared
      wave_sync();  // everyone has read the pivot before its owner overwrites it
      for (int r = i; r < n; r += W) {
        if (r > j) L[r * n + j] = L[r * n + j] / d;
        else if (r == j) L[j * n + j] = d;
      }
      wave_sync();
    } else {
    REAL lij = 0;
    if (i > j && i < n) { lij = L[i * n + j] / d; }
    wave_sync();  // everyone has read column j before it is overwritten
    if (i > j && i < n) L[i * n + j] = lij;
    if (i == j) L[j * n + j] = d;
    wave_sync();
    }
    {  // trailing update L[r][c] -= L[r][j] * L[c][j] for j < c <= r < n, spread over all lanes (2D index)
      const int m = n - j - 1;
      const float inv_m = 1.0f / (float)(m > 0 ? m : 1);
      const int total = m * m;
      for (int t0 = 0; t0 < total; t0 += 2 * W) {
        const int ta = t0 + i, tb = t0 + W + i;
        const int ra = (int)(((float)ta + 0.5f) * inv_m), rb = (int)(((float)tb + 0.5f) * inv_m);
        const int ca = ta - ra * m, cb = tb - rb * m;
        const bool oka = ta < total && ca <= ra, okb = tb < total && cb <= rb;
        const int ia = (j + 1 + ra) * n, ka = j + 1 + ca, ib = (j + 1 + rb) * n, kb = j + 1 + cb;
        REAL va = 0, vb = 0, la = 0, lb = 0, ua = 0, ub = 0;
        if (oka) { va = L[ia + ka]; la = L[ia + j]; ua = L[ka * n + j]; }
        if (okb) { vb = L[ib + kb]; lb = L[ib + j]; ub = L[kb * n + j]; }
        if (oka) L[ia + ka] = va - la * ua;
        if (okb) L[ib + kb] = vb - lb * ub;
      }
    }
    wave_sync();
  }
}


// reciprocal diagonal of a Cholesky factor (one lane per row); consumed by chol_solve
template <int W, bool PACKED, typename REAL>
__device__ __forceinline__ void chol_inv_diag(const REAL* L, REAL* inv, int n) {
  for (int k = sub_lane<W>(); k < n; k += W) inv[k] = 1 / L[tri_at<PACKED>(k, k, n)];
}

// x = (L L^T)^-1 b  (math.small_cholesky_solve :132-168).  Column-oriented substitution: lane i carries its
// running right-hand side; the value solved at step k is broadcast.  Per element the operation order equals
// the reference's row loops (k ascending forward, descending backward); the divisions by L[k][k] are
// multiplications by the precomputed reciprocal (<= 1 ulp per step off the reference's quotient).
// ... with more rows than lanes (models beyond 64 dofs; a capacity path, not a tuned one): the running right-hand side lives in x (LDS), every
// lane updates the rows it owns, two barriers per step.  Per element the same operations in the same order as below.
template <int W, bool PACKED, typename REAL>
__device__ __forceinline__ void chol_solve_lds_big(const REAL* L, const REAL* inv, const REAL* b, REAL* x, int n) {
  const int i = sub_lane<W>();
  for (int r = i; r < n; r += W) x[r] = b[r];
  wave_sync();
  for (int k = 0; k < n; k++) {
    const REAL yk = x[k] * inv[k];
    wave_sync();
    for (int r = i; r < n; r += W) {
      if (r == k) x[r] = yk;
      else if (r > k) x[r] = x[r] - L[tri_at<PACKED>(r, k, n)] * yk;
    }
    wave_sync();
  }
  for (int k = n - 1; k >= 0; k--) {
    const REAL xk = x[k] * inv[k];
    wave_sync();
    for (int r = i; r < n; r += W) {
      if (r == k) x[r] = xk;
      else if (r < k) x[r] = x[r] - L[tri_at<PACKED>(k, r, n)] * xk;
    }
    wave_sync();
  }
}

template <int W, bool PACKED, typename REAL>
__device__ __forceinline__ void chol_solve_lds(const REAL* L, const REAL* inv, const REAL* b, REAL* x, int n) {
  if (n > W) { chol_solve_lds_big<W, PACKED>(L, inv, b, x, n); return; }
  const int i = sub_lane<W>();
  REAL s = (i < n) ? b[i] : (REAL)0;
  const REAL myinv = (i < n) ? inv[i] : (REAL)0;
  // forward: eight columns of this lane's row are fetched at once, then consumed by eight dependent steps
  for (int k0 = 0; k0 < n; k0 += 8) {
    REAL r[8];
#pragma unroll
    for (int t = 0; t < 8; t++) r[t] = (i > k0 + t && i < n && k0 + t < n) ? L[tri_at<PACKED>(i, k0 + t, n)] : (REAL)0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int k = k0 + t;
      if (k < n) {
        const REAL yk = sub_read<W>(s * myinv, k);
        if (i == k) s = yk;
        else if (i > k && i < n) s = s - r[t] * yk;
      }
    }
  }
  for (int k1 = n - 1; k1 >= 0; k1 -= 8) {
    REAL r[8];
#pragma unroll
    for (int t = 0; t < 8; t++) r[t] = (k1 - t >= 0 && i < k1 - t) ? L[tri_at<PACKED>(k1 - t, i, n)] : (REAL)0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int k = k1 - t;
      if (k >= 0) {
        const REAL xk = sub_read<W>(s * myinv, k);
        if (i == k) s = xk;
        else if (i < k) s = s - r[t] * xk;
      }
    }
  }
  if (i < n) x[i] = s;
  wave_sync();
}

// =====================================================================================================================
// Register-resident triangular kernels for n <= NMAX (8 / 16 / 32).  Lane i owns row i of the matrix in
// registers; everything another lane needs from it travels by v_readlane (a scalar broadcast), so the
// factorisation and the substitutions run without LDS round trips or barriers.  Loops are fully unrolled
// over NMAX so every register index is a compile-time constant.  Operation order per element is the
// reference's (math.small_cholesky :117-127, small_cholesky_solve :152-166).
// =====================================================================================================================
template <typename REAL, int NMAX>
struct TriReg {
  REAL row[NMAX];  // row[k] = L[i][k], k <= i
  REAL col[NMAX];  // col[k] = L[k][i], k >= i  (row i of L^T)
  REAL inv;        // 1 / L[i][i]
};

template <int W, bool PACKED, typename REAL, int NMAX>
__device__ __forceinline__ void tri_load(TriReg<REAL, NMAX>& T, const REAL* L, int n) {
  const int i = sub_lane<W>();
#pragma unroll
  for (int k = 0; k < NMAX; k++) {
    T.row[k] = (k < n && i < n && k <= i) ? L[tri_at<PACKED>(i, k, n)] : (REAL)0;
    T.col[k] = (k < n && i < n && k >= i) ? L[tri_at<PACKED>(k, i, n)] : (REAL)0;
  }
  T.inv = (i < n) ? 1 / L[tri_at<PACKED>(i, i, n)] : (REAL)0;
}

// x = (L L^T)^-1 b with b, x distributed one element per lane
template <int W, typename REAL, int NMAX>
__device__ __forceinline__ REAL tri_solve(const TriReg<REAL, NMAX>& T, REAL bi, int n) {
  const int i = sub_lane<W>();
  REAL s = (i < n) ? bi : (REAL)0;
  // no k < n guards: lanes >= n carry s = 0 and inv = 0, rows / columns >= n are zero, so those steps change nothing
#pragma unroll
  for (int k = 0; k < NMAX; k++) {
    const REAL yk = dof_read<W, NMAX>(s * T.inv, k);
    if (i == k) s = yk;
    else if (i > k) s = s - T.row[k] * yk;
  }
#pragma unroll
  for (int k = NMAX - 1; k >= 0; k--) {
    const REAL xk = dof_read<W, NMAX>(s * T.inv, k);
    if (i == k) s = xk;
    else if (i < k) s = s - T.col[k] * xk;
  }
  return s;
}

// Row i AND column i of a lower-triangular factor in ONE register array: lane i only ever needs L[i][k] for k <= i (forward
// substitution) and L[k][i] for k >= i (backward), and the two index ranges meet at the diagonal -- t[k] = k <= i ? L[i][k] : L[k][i].
template <typename REAL, int NMAX>
struct TriPack {
  REAL t[NMAX];
  REAL inv;  // 1 / L[i][i]
};
// x = (L L^T)^-1 b, one element per lane (same step order as tri_solve / math.small_cholesky_solve :132-168)
template <int W, typename REAL, int NMAX>
__device__ __forceinline__ REAL tri_solve(const TriPack<REAL, NMAX>& T, REAL bi, int n) {
  const int i = sub_lane<W>();
  REAL s = (i < n) ? bi : (REAL)0;
#pragma unroll
  for (int k = 0; k < NMAX; k++) {
    const REAL yk = dof_read<W, NMAX>(s * T.inv, k);
    if (i == k) s = yk;
    else if (i > k) s = s - T.t[k] * yk;
  }
#pragma unroll
  for (int k = NMAX - 1; k >= 0; k--) {
    const REAL xk = dof_read<W, NMAX>(s * T.inv, k);
    if (i == k) s = xk;
    else if (i < k) s = s - T.t[k] * xk;
  }
  return s;
}

// Cholesky of the symmetric matrix A (LDS, packed lower rows, n <= NMAX <= 16) straight into a TriPack: lane i starts from row i of A; the
// column scale L[k][j] of step j is broadcast to every lane for the trailing update anyway, and lane j keeps it as its column entry -- row AND
// column of the factor end up in registers without the round trip through an n x n LDS image (store, reciprocal diagonal, reload) that a separate
// factorisation and substitution pay.  math.small_cholesky :117-127 (pivots clamped at 1e-12), same operation order as chol_factor_reg.
template <int W, typename REAL, int NMAX>
__device__ __forceinline__ void chol_factor_pack(const REAL* A, TriPack<REAL, NMAX>& T, int n) {
  const int i = sub_lane<W>();
  const bool valid = i < n;
  if constexpr (sizeof(REAL) == 4) {  // row i of A is contiguous: one base address, every lane reads a valid one (the guard selects afterwards -- a guarded read is an exec-masked round trip of its own)
    const REAL* row = A + (valid ? (i * (i + 1)) / 2 : 0);
#pragma unroll
    for (int k = 0; k < NMAX; k++) { const bool c = k < n && valid && k <= i; const REAL a = row[c ? k : 0]; T.t[k] = c ? a : (REAL)0; }
  } else {  // (double: the extra values in flight cost the LDS solver's float64 instantiation its second wave per SIMD, 213 -> 256 VGPRs)
#pragma unroll
    for (int k = 0; k < NMAX; k++) T.t[k] = (k < n && valid && k <= i) ? A[(i * (i + 1)) / 2 + k] : (REAL)0;
  }
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) {
      const REAL sj = dof_read<W, NMAX>(T.t[j], j);
      const REAL dj = r_sqrt<REAL>(sj > (REAL)1e-12 ? sj : (REAL)1e-12);
      const REAL lij = (i == j) ? dj : ((i > j && valid) ? T.t[j] / dj : (REAL)0);
      if (i >= j) T.t[j] = lij;
#pragma unroll
      for (int k = j + 1; k < NMAX; k++) {
        const REAL lkj = dof_read<W, NMAX>(lij, k);  // L[k][j]: lane j keeps it (its column), lanes below update their rows
        T.t[k] = (i == j) ? lkj : ((i > j) ? T.t[k] - lij * lkj : T.t[k]);
      }
    }
  }
  REAL dg = 1;
#pragma unroll
  for (int k = 0; k < NMAX; k++) dg = (k == i) ? T.t[k] : dg;
  T.inv = valid ? 1 / dg : (REAL)0;
}
// Cholesky, 16 < n <= NMAX <= 32, of the symmetric matrix whose row i (entries k <= i) lane i holds in T.t[k]: the torch.linalg.cholesky(A + 1e-10 I) branch of
// math.small_cholesky (:108-113), right-looking with every element updated in step order -- per element the operations of chol_factor_lds, bit for bit (plain
// square root and division).  Column j of the factor reaches every lane through a two-column LDS buffer (`colbuf`, 2 * NMAX reals: uniform-address
// broadcast reads, as in chol_factor_reg) and lane j keeps it: row AND column of the factor end up in the TriPack without an n x n image.
template <int W, typename REAL, int NMAX>
__device__ __forceinline__ void chol_pack_big(TriPack<REAL, NMAX>& T, REAL* colbuf, int n) {
  static_assert(NMAX > 16 && NMAX <= 32 && NMAX <= W, "one lane per row");
  const int i = sub_lane<W>();
  const bool valid = i < n;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) {
      const REAL sj = sub_read<W>(T.t[j], j) + (REAL)1e-10;
      const REAL dj = r_sqrt<REAL>(sj);
      const REAL lij = (i == j) ? dj : ((i > j && valid) ? T.t[j] / dj : (REAL)0);
      if (i >= j) T.t[j] = lij;
      REAL* col = colbuf + (j & 1) * NMAX;
      if (i < NMAX) col[i] = lij;  // (no barrier: one wavefront's LDS operations complete in order; the two buffers alternate, so step j + 1's stores do not pass step j's reads)
      constexpr int CH = 8;
#pragma unroll
      for (int k0 = 0; k0 < NMAX; k0 += CH) {
        if (k0 + CH - 1 > j) {
          REAL c[CH];
#pragma unroll
          for (int t = 0; t < CH; t++) c[t] = (k0 + t > j && k0 + t < NMAX) ? col[k0 + t] : (REAL)0;
#pragma unroll
          for (int t = 0; t < CH; t++) {
            if (k0 + t > j && k0 + t < NMAX) {
              T.t[k0 + t] = (i == j) ? c[t] : ((i > j) ? T.t[k0 + t] - lij * c[t] : T.t[k0 + t]);
              asm volatile("" : "+v"(T.t[k0 + t]));  // (pinned here: left alone the scheduler sinks the update to step k and keeps every column read so far in registers, see chol_factor_reg)
            }
          }
        }
      }
    }
  }
  REAL dg = 1;
#pragma unroll
  for (int k = 0; k < NMAX; k++) dg = (k == i) ? T.t[k] : dg;
  T.inv = valid ? 1 / dg : (REAL)0;
}

// x = (A)^-1 b for a symmetric positive matrix in LDS (packed lower rows), n <= 16: factor and substitute in registers
template <int W, typename REAL, int NMAX>
__device__ __forceinline__ void chol_factor_solve_n(const REAL* A, const REAL* b, REAL* x, int n) {
  TriPack<REAL, NMAX> T;
  chol_factor_pack<W, REAL, NMAX>(A, T, n);
  const int i = sub_lane<W>();
  const REAL xi = tri_solve<W, REAL, NMAX>(T, i < n ? b[i] : (REAL)0, n);
  if (i < n) x[i] = xi;
  wave_sync();
}
template <int W, typename REAL, int NLO = 0, int NHI = 16>
__device__ __forceinline__ void chol_factor_solve(const REAL* A, const REAL* b, REAL* x, int n) {
  // NLO <= n <= NHI is known to the caller at compile time (an instantiation of the register solver serves one range of nv): the variants outside it are not compiled
  if (NLO <= 8 && (NHI <= 8 || n <= 8)) { if constexpr (NLO <= 8) chol_factor_solve_n<W, REAL, 8>(A, b, x, n); }
  else if (NLO <= 12 && NHI > 8 && (NHI <= 12 || n <= 12)) { if constexpr (NLO <= 12 && NHI > 8) chol_factor_solve_n<W, REAL, 12>(A, b, x, n); }
  else { if constexpr (NHI > 12) chol_factor_solve_n<W, REAL, 16>(A, b, x, n); }
}

// Cholesky of the symmetric matrix A (LDS, n x n) into L (LDS, lower triangle, zeros above)
template <int W, bool APACKED, typename REAL, int NMAX>
__device__ __forceinline__ void chol_factor_reg(const REAL* A, REAL* L, int n) {
  const int i = sub_lane<W>();
  REAL row[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) row[k] = (k < n && i < n && k <= i) ? A[tri_at<APACKED>(i, k, n)] : (REAL)0;
  wave_sync();  // L may alias A (and anything else that is dead by now): every lane has its row in registers
  const bool big = n > INLINE_CHOL_MAX;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) {
      REAL s = dof_read<W, NMAX>(row[j], j);           // A[j][j] - sum_k L[j][k]^2, accumulated by the updates below
      if (big) s = s + (REAL)1e-10;                  // torch.linalg.cholesky(A + 1e-10 I), math.py:108-113
      REAL d, lij;
      if (big && sizeof(REAL) == 8) {
        // n > 16 is LAPACK's blocked factorisation in the reference: its operation order is not part of the algorithm, so the
        // column scale is taken as a refined reciprocal square root (hardware estimate + two Newton steps, ~1 ulp) instead
        // of a square-root sequence followed by a divide sequence -- the two longest dependent chains of the CRB phase
        const double sd = (double)s, hs = 0.5 * sd;
        double y = __builtin_amdgcn_rsq(sd);
        y = y * (1.5 - hs * y * y);
        y = y * (1.5 - hs * y * y);
        d = (REAL)(sd * y);
        lij = (i == j) ? d : ((i < n) ? row[j] * (REAL)y : (REAL)0);
      } else {
        d = big ? r_sqrt<REAL>(s) : r_sqrt<REAL>(s > (REAL)1e-12 ? s : (REAL)1e-12);
        lij = (i == j) ? d : ((i < n) ? row[j] / d : (REAL)0);  // lanes i < j hold zeros: harmless; lanes >= n stay zero
      }
      row[j] = lij;
      if constexpr (NMAX > 16) {
        // The trailing update needs column j of the factor -- lane k's lij -- in EVERY lane.  Broadcast lane by lane that is a v_readlane (two for a
        // double) per value, half of the instructions of the whole factorisation at n = 27; instead the column goes through LDS once: each lane
        // writes its entry, and every lane reads the column back with uniform addresses (LDS broadcast reads, two doubles / four floats per
        // instruction).  Two alternating buffers at the head of L (dead until the factor is stored below): one barrier per column.  The products and
        // differences are the same expressions in the same order.
        REAL* col = L + (j & 1) * NMAX;
        if (i < NMAX) col[i] = lij;  // (no barrier: one wavefront's LDS operations complete in order, and the compiler keeps the store ahead of the reads of the same buffer)
        constexpr int CH = 8;  // column entries per batch of reads (adjacent ones merge into two-element LDS reads): 16 VGPRs of a double factorisation
#pragma unroll
        for (int k0 = 0; k0 < NMAX; k0 += CH) {
          if (k0 + CH - 1 > j) {
            REAL c[CH];
#pragma unroll
            for (int t = 0; t < CH; t++) c[t] = (k0 + t > j && k0 + t < NMAX) ? col[k0 + t] : (REAL)0;
#pragma unroll
            for (int t = 0; t < CH; t++) {
              if (k0 + t > j && k0 + t < NMAX) {
                row[k0 + t] = row[k0 + t] - lij * c[t];
                // the update is only consumed at step k: left alone the scheduler sinks it there and keeps every column read so far in registers
                // (a left-looking factorisation holding n^2 / 2 values: 256 VGPRs, 256 AGPRs and 500 bytes of scratch).  Pin the value here.
                asm volatile("" : "+v"(row[k0 + t]));
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int k = j + 1; k < NMAX; k++) {           // no k < n guard: lanes / columns >= n carry exact zeros, the update is a no-op there
          const REAL lkj = dof_read<W, NMAX>(lij, k);  // L[k][j]
          row[k] = row[k] - lij * lkj;                 // only k <= i is ever read back
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NMAX; k++)
    if (k < n && i < n) L[i * n + k] = (k <= i) ? row[k] : (REAL)0;
  wave_sync();
}

template <int W, bool PACKED, typename REAL, int NMAX>
__device__ __forceinline__ void chol_solve_reg(const REAL* L, const REAL* b, REAL* x, int n) {
  TriReg<REAL, NMAX> T;
  tri_load<W, PACKED>(T, L, n);
  const int i = sub_lane<W>();
  const REAL xi = tri_solve<W>(T, (i < n) ? b[i] : (REAL)0, n);
  if (i < n) x[i] = xi;
  wave_sync();
}
template <int W, bool PACKED, typename REAL>
__device__ __forceinline__ void chol_solve(const REAL* L, const REAL* inv, const REAL* b, REAL* x, int n) {
  // register variants only up to 16: at 32 the two register triangles push the solver phases to 256 VGPRs
  // (one wave per SIMD), which costs more than the substitution saves
  if (n <= 8) chol_solve_reg<W, PACKED, REAL, 8>(L, b, x, n);
  else if (n <= 16) chol_solve_reg<W, PACKED, REAL, 16>(L, b, x, n);
  else chol_solve_lds<W, PACKED>(L, inv, b, x, n);
}

// A: symmetric matrix, full n x n or (APACKED) packed lower rows; L: full n x n, may alias A when n <= 32
template <int W, typename REAL, int MAXREG, bool APACKED>
__device__ __forceinline__ void chol_factor(const REAL* A, REAL* L, int n) {
  if (n <= 8) chol_factor_reg<W, APACKED, REAL, 8>(A, L, n);
  else if (n <= 16) chol_factor_reg<W, APACKED, REAL, 16>(A, L, n);
  else if (MAXREG >= 24 && n <= 24) { if constexpr (MAXREG >= 24) chol_factor_reg<W, APACKED, REAL, 24>(A, L, n); }
  else if (MAXREG >= 28 && n <= 28) { if constexpr (MAXREG >= 28) chol_factor_reg<W, APACKED, REAL, 28>(A, L, n); }
  else if (MAXREG >= 32 && n <= 32) { if constexpr (MAXREG >= 32) chol_factor_reg<W, APACKED, REAL, 32>(A, L, n); }
  else chol_factor_lds<W, APACKED>(A, L, n);
}

// sum_k a[k * sa] * b[k * sb] accumulated in index order (the reference's reduction order for its explicit
// loops); the loads of four terms are issued together so one LDS round trip covers four multiply-adds.
template <typename REAL>
__device__ __forceinline__ REAL dot_seq(const REAL* a, int sa, const REAL* b, int sb, int n) {
  REAL s = 0;
  int k = 0;
  for (; k + 4 <= n; k += 4) {
    const REAL a0 = a[k * sa], a1 = a[(k + 1) * sa], a2 = a[(k + 2) * sa], a3 = a[(k + 3) * sa];
    const REAL b0 = b[k * sb], b1 = b[(k + 1) * sb], b2 = b[(k + 2) * sb], b3 = b[(k + 3) * sb];
    s += a0 * b0; s += a1 * b1; s += a2 * b2; s += a3 * b3;
  }
  for (; k < n; k++) s += a[k * sa] * b[k * sb];
  return s;
}

// "left to the fallback launch": a quiet NaN with a payload, written to the first element of the environment's qacc row by the register
// solver and tested (bit pattern) by the LDS solver's fallback launch.  It lives in the call's own output storage, so concurrent calls on
// other streams / other Data never see each other's marks; a solve that legitimately ends in NaN never produces this payload.
__device__ __forceinline__ float mjh_bail_mark(float) { return __builtin_bit_cast(float, 0x7fc5eed5u); }
__device__ __forceinline__ double mjh_bail_mark(double) { return __builtin_bit_cast(double, 0x7ff80005eed5eed5ull); }
__device__ __forceinline__ bool mjh_is_bail_mark(float x) { return __builtin_bit_cast(unsigned, x) == 0x7fc5eed5u; }
__device__ __forceinline__ bool mjh_is_bail_mark(double x) { return __builtin_bit_cast(unsigned long long, x) == 0x7ff80005eed5eed5ull; }

// what the register solver reads from leaves that earlier LAUNCHES produced (crb / factor, velocity, the caller's inputs): in the fused constraint + solver
// kernel these loads are issued at the head of the constraint stage and arrive under its arithmetic
template <typename REAL, int NMAX>
struct Sol2Pre {
  REAL f;                   // qfrc_smooth[l]
  TriPack<REAL, NMAX> T;    // row and column l of the factor of M (inv is set by the solver)
  REAL qp[2];               // qpos[l], qpos[l + 32]
  REAL ac, ad, warm;        // act[l], act_dot[l], qacc_warmstart[l]
};

// ... and what the constraint stage of that kernel hands to the solver in registers: lane r's single-column limit row, the dense rows of the active contacts
template <typename REAL>
struct Sol2Con {
  REAL jl, Dl, arl;  // the row's one Jacobian entry, efc_D, efc_aref
  int ldof;          // its column
  int nda;           // dense rows of the active contacts (compact, in S.efc_Jc())
  int nact;          // active contacts
};

#ifndef MJH_HROWS
#define MJH_HROWS 4
#endif
#ifndef MJH_INCR_H_PERIOD
#define MJH_INCR_H_PERIOD 8  /* Newton, incremental Hessian: every so many builds of a solve the Hessian is rebuilt from all rows (bounds the drift of the add / subtract updates) */
#endif
// lane c of every quad (4 consecutive lanes) broadcast to the quad's four lanes (DPP quad_perm [c, c, c, c])
template <int C> __device__ __forceinline__ float quad_bcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C * 0x55, 0xf, 0xf, false));
}

// =====================================================================================================================
// FRIC: the general constraint / solver instantiations (equality, frictionloss, dense limit rows; also max_contact_points);
// DIRECT: the constraint phase of small models that writes its contact rows straight to the efc_J leaf (kernel 8)
template <typename REAL, int W = 64, bool FRIC = false, bool DIRECT = false>
struct Env {
  __device__ __forceinline__ static int lane() { return sub_lane<W>(); }
  // the lane index of a stage function: opaque to the optimiser, so the address arithmetic that hangs off it is formed inside the stage -- hoisted to the
  // kernel's head (every stage is inlined into one function) the addresses of ALL stages are live, or spilled, across the whole kernel
  __device__ __forceinline__ static int lane_here() { int l = sub_lane<W>(); asm volatile("" : "+v"(l)); __builtin_assume(l >= 0 && l < W); return l; }
  // dof-frictionloss rows of the SOLVER phase: its frictionloss-free instantiation (kernel 4) carries none of their code,
  // models that have such rows run kernel 6
  __device__ __forceinline__ static int nf_() { return FRIC ? M.nf : 0; }
  // equality rows (always in the quadratic set).  Inside the solver phase the rows are kept in the order
  // [frictionloss | joint limits | equality | contacts]: single-column rows first, dense rows after them; the Data leaves
  // use the reference's order [equality | frictionloss | limits | contacts] and ext_row() maps one to the other.
  __device__ __forceinline__ static int ne_() { return FRIC ? M.ne : 0; }
  // tendon-frictionloss rows (constraint.py:230-234): dense rows with the frictionloss cost; they lead the solver's dense block
  __device__ __forceinline__ static int nft_() { return FRIC ? M.nft : 0; }
  __device__ __forceinline__ static bool is_dfric(int r) { return FRIC && (unsigned)(r - (nf_() + M.nl)) < (unsigned)M.nft; }
  __device__ __forceinline__ static bool is_fric(int r) { return FRIC && (r < nf_() || is_dfric(r)); }
  __device__ __forceinline__ static int fric_index(int r) { return r < nf_() ? r : r - M.nl; }  // slot of a friction row in efc_fl: dof rows, then tendon rows
  __device__ __forceinline__ static int nlim_rows() { return FRIC ? M.nlb + M.nlt : 0; }
  __device__ __forceinline__ static bool is_eq_row(int r) { return FRIC && (unsigned)(r - (nf_() + M.nl + nft_())) < (unsigned)M.ne; }
  // dense limit rows (ball joints, tendons) sit between the equality rows and the contacts in the solver's dense block:
  //   solver order  [dof frictionloss | slide-hinge limits || tendon frictionloss | equality | ball limits | tendon limits | contacts]
  //   Data order    [equality | dof frictionloss | tendon frictionloss | ball limits | slide-hinge limits | tendon limits | contacts]
  __device__ __forceinline__ static int ext_row(int r) {
    if (!FRIC) return r;
    const int nf = nf_(), nft = M.nft, s1 = nf + M.nl, s2 = s1 + nft, ne = M.ne, nlb = M.nlb;
    if (r < nf) return ne + r;
    if (r < s1) return ne + nft + nlb + r;              // ne + nf + nft + nlb + (r - nf)
    if (r < s2) return ne + nf + (r - s1);
    if (r < s2 + ne) return r - s2;
    if (r < s2 + ne + nlb) return r - s2 + nf + nft;    // ne + nf + nft + (r - s2 - ne)
    return r;                                           // tendon limits and contacts sit at the same index in both orders
  }
  // S.i_crow_act of the small-model constraint phase has two meanings: per dense ROW a 0 / 1 activity flag, or -- DevModel::crow_by_con: every contact has con_rows rows and
  // dense row q belongs to contact q / con_rows -- per CONTACT 0 = inactive, else its place in the compact list of active contacts + 1.  Every reader goes through these two:
  __device__ __forceinline__ static int crow_entry(const int* row_act, int q, float inv_rows) {  // non-zero iff dense row q is a row of an active contact
    int qc = q, sub_;
    if (M.crow_by_con) split_index(q, M.con_rows, inv_rows, qc, sub_);
    return row_act[qc];
  }
  __device__ __forceinline__ static int crow_slot_of_contact(const int* row_act, int c) { return row_act[c] - 1; }  // crow_by_con only: compact slot of contact c, -1 = inactive
  LdsView<REAL> S;
  bool consts_done_ = false;  // stage kernel: this workgroup stored the model-constant contact leaves at the kernel's head (KArgs::xswap_c)
  int64_t e;      // environment index
  REAL ho_g[12];  // whole-pass kernel: this lane's geom frame (position, matrix) as the kinematics formed it, handed to the constraint stage in registers (lane g <-> geom g, ngeom <= W)
  // from here on the environment index is news to the optimiser: the address arithmetic of a store section (e * leaf width, one 64-bit value per leaf) is
  // formed where the stores are instead of at the kernel's head, where it sat in registers -- or in scratch -- across the whole phase
  __device__ __forceinline__ void rebind() { late_bind<W>(e); }
  int flags;
#ifdef MJH_STAMPS
  unsigned long long stamp_prev = 0;
#endif

  __device__ __forceinline__ Env(REAL* lds, int64_t env, int fl) : S{lds, &KA.off}, e(env), flags(fl) {}

  // every phase streams the leaves it produces to the Data being computed (`out` == KArgs::cur)
  template <typename T>
  __device__ __forceinline__ void put(T* g, const REAL* l, int n) { row_store<W>(g, l, n, e); }
  template <typename T>
  __device__ __forceinline__ void putnt(T* g, const REAL* l, int n) { row_store<W, true>(g, l, n, e); }  // a leaf no later kernel of the step reads: non-temporal

  // ---- state loads (+ _check_state, forward.py:44-59, on the caller's state) ------------------------------------------------
  __device__ __forceinline__ REAL checked(REAL x, REAL fallback) const {
    return (!r_finite(x) || r_abs(x) > (REAL)mjMAXVAL) ? fallback : x;
  }
  // raw: the caller's qpos (kinematics normalises it); otherwise the normalised qpos this pass already wrote
  __device__ __forceinline__ void load_qpos(bool raw) {
    const bool from_in = raw && !KA.state_from_cur;
    const REAL* src = (raw ? (KA.state_from_cur ? KA.cur.qpos : in.qpos) : KA.cur.qpos) + e * M.nq;
    const bool check = from_in && KA.do_step;
    for (int i = lane(); i < M.nq; i += W) S.qpos()[i] = check ? checked(src[i], M.qpos0[i]) : src[i];
  }
  __device__ __forceinline__ void load_qvel() {
    const bool from_in = !KA.state_from_cur;
    const REAL* src = (from_in ? in.qvel : KA.cur.qvel) + e * M.nv;
    const bool check = from_in && KA.do_step;
    for (int i = lane(); i < M.nv; i += W) S.qvel()[i] = check ? checked(src[i], (REAL)0) : src[i];
  }
  __device__ __forceinline__ void load_act() {
    const REAL* src = KA.state_from_cur ? KA.cur.act : in.act;
    row_load<W>(S.act(), src, M.na, e);
  }

  __device__ __forceinline__ void frame_stores() {
    putnt(out.xpos, S.xpos(), 3 * M.nbody); putnt(out.xquat, S.xquat(), 4 * M.nbody); if (!M.lds_diet) putnt(out.xmat, S.xmat(), 9 * M.nbody);  // (lds_diet: the two matrices went out from the registers that formed them)
    putnt(out.xipos, S.xipos(), 3 * M.nbody); if (!M.lds_diet) putnt(out.ximat, S.ximat(), 9 * M.nbody);
    putnt(out.xanchor, S.xanchor(), 3 * M.njnt); putnt(out.xaxis, S.xaxis(), 3 * M.njnt);
  }
  __device__ __forceinline__ void com_stores() {
    if (M.nt_all) { putnt(out.subtree_com, S.subtree_com(), 3 * M.nbody); putnt(out.cdof, S.cdof(), 6 * M.nv); }
    else { put(out.subtree_com, S.subtree_com(), 3 * M.nbody); put(out.cdof, S.cdof(), 6 * M.nv); }
    putnt(out.cinert, S.cinert(), 10 * M.nbody);
  }
  // ---- kinematics (smooth.py:34-207): each lane walks world -> its body along the ancestor chain ---------------------------------
  // com_first (whole-pass kernel, out of lockstep: KArgs::xswap_k): com_pos() runs HERE, between the body frames and the geom / site / camera / light frames -- both only read the body frames --
  // instead of behind this function (the caller then skips it): the same operations on the same inputs, while the other workgroups are in the other section
  template <bool DEFER = false, bool KEEPG = false, bool XK = KEEPG>
  __device__ __forceinline__ void kinematics(bool with_cams, bool com_first = false) {
    const int l = lane_here();
    // joint-local rotations first, one lane per joint: the trigonometry and the quaternion normalisations leave the
    // serial ancestor walk below (same expressions, smooth.py:85-120)
    for (int j = l; j < M.njnt; j += W) {
      const int t = M.jnt_type[j], qa = M.jnt_qposadr[j];
      REAL q[4] = {0, 0, 0, 0};
      if (t == JNT_FREE || t == JNT_BALL) {
        const int o = (t == JNT_FREE) ? qa + 3 : qa;
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = S.qpos()[o + i];
        normalize_n<REAL, 4>(q);
      } else if (t == JNT_HINGE) {
        axis_angle_to_quat(M.jnt_axis + 3 * j, S.qpos()[qa] - M.qpos0[qa], q);
      } else {
        q[0] = S.qpos()[qa] - M.qpos0[qa];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) S.jquat()[4 * j + i] = q[i];
    }
    wave_sync();
    STAMP(8);
    // Pointer jumping (DevModel::kin_tab; models with one body per lane).  The serial walk below has every lane compose its whole ancestor chain -- ~7 levels and ~10 joints of
    // dependent float64 rotations for the humanoid, 35 k of the kernel's 309 k cycles, the same chain recomputed by every lane.  Here each lane first composes the frame of ITS body
    // relative to its parent (body offset, then its own joints: smooth.py:85-120 in the parent's frame), then ceil(log2(depth)) rounds replace "relative to the ancestor 2^r levels
    // up" by "relative to the ancestor 2^(r+1) levels up" (frame(a) o frame(b), through the xpos / xquat arrays of the arena), and the joints' anchors and axes, formed in the
    // parent's frame, are carried to the world by one more rotation.  The same compositions as the walk in a different association: results agree to rounding (1e-16 relative),
    // not bit for bit -- the parity bounds of the leaves upstream of the solver are 1e-9 (float64) / 2e-4 (float32).
    // one joint of a body (smooth.py:85-120): its anchor and axis in the frame (pos, quat) reached so far, then that frame moved by the joint
    auto joint_step = [&](int j, REAL* pos, REAL* quat, bool keep) {
      const int t = M.jnt_type[j], qa = M.jnt_qposadr[j];
      const REAL jpos[3] = {M.jnt_pos[3 * j], M.jnt_pos[3 * j + 1], M.jnt_pos[3 * j + 2]};
      const REAL jaxis[3] = {M.jnt_axis[3 * j], M.jnt_axis[3 * j + 1], M.jnt_axis[3 * j + 2]};
      const REAL ql[4] = {S.jquat()[4 * j], S.jquat()[4 * j + 1], S.jquat()[4 * j + 2], S.jquat()[4 * j + 3]};
      REAL anchor[3], axis[3];
      if (t == JNT_FREE) {
#pragma unroll
        for (int i = 0; i < 3; i++) { anchor[i] = S.qpos()[qa + i]; pos[i] = S.qpos()[qa + i]; }
        axis[0] = 0; axis[1] = 0; axis[2] = 1;
#pragma unroll
        for (int i = 0; i < 4; i++) quat[i] = ql[i];
      } else {
        REAL r[3];
        rotate(jpos, quat, r);
#pragma unroll
        for (int i = 0; i < 3; i++) anchor[i] = r[i] + pos[i];
        rotate(jaxis, quat, axis);
        if (t == JNT_BALL || t == JNT_HINGE) {
          quat_mul(quat, ql, quat);
          rotate(jpos, quat, r);
#pragma unroll
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - r[i];
        } else {
          const REAL dq = ql[0];
#pragma unroll
          for (int i = 0; i < 3; i++) pos[i] = pos[i] + axis[i] * dq;
        }
      }
      if (keep) {
#pragma unroll
        for (int i = 0; i < 3; i++) { S.xanchor()[3 * j + i] = anchor[i]; S.xaxis()[3 * j + i] = axis[i]; }
      }
    };
    // Level sweep (DevModel::kin_lvl; round 6; models with one body per lane).  The serial walk below has every lane recompute its whole ancestor chain, and -- what costs the
    // time -- reads the constants of every level and of every joint on it (chain id -> body offset -> joint type / address / anchor / axis: dependent table reads through L2) INSIDE
    // that dependent chain: 35 k of the humanoid kernel's 278 k cycles for ~10 joints of arithmetic.  Here lane b owns body b: it reads ITS constants once, up front and all at once
    // (nothing of them depends on another lane), then the levels of the tree are swept in order -- at level L the lanes whose body sits L below the world take their parent's
    // finished frame from the arena, apply their own offset and joints, and put their frame down.  Every body's frame is formed by exactly the operations of the walk in exactly its
    // order (the walk recomputes the parent's frame with the same operations on the same inputs): bit-identical leaves, unlike the pointer-jumping form below.
    // (Compiled into the whole-pass kernel only -- KEEPG -- for now: in the stand-alone kinematics kernels the constants' registers set the allocation, 80 -> 150 VGPRs in float32.)
    const bool lvl = KEEPG && M.kin_lvl != 0 && M.nbody <= W && M.kin_tab == nullptr;
    if constexpr (KEEPG) if (lvl) {
      const int nb = M.nbody, md = M.max_depth;
      const int b = l;
      const bool body = b > 0 && b < nb;
      int depth = 0, par = 0, jn = 0, j0 = 0;
      REAL bp[3] = {0, 0, 0}, bq[4] = {1, 0, 0, 0};
      constexpr int KJ = 3;  // joints of a body whose constants ride in registers (more: read at use)
      int jt[KJ], jqa[KJ];
      REAL jp[KJ][3], jx[KJ][3];
      if (body) {
        depth = M.body_depth[b]; par = M.body_parentid[b]; jn = M.body_jntnum[b]; j0 = M.body_jntadr[b];
#pragma unroll
        for (int i = 0; i < 3; i++) bp[i] = M.body_pos[3 * b + i];
#pragma unroll
        for (int i = 0; i < 4; i++) bq[i] = M.body_quat[4 * b + i];
      }
#pragma unroll
      for (int jj = 0; jj < KJ; jj++) {
        const int j = (body && jj < jn) ? j0 + jj : 0;
        jt[jj] = M.jnt_type[j]; jqa[jj] = M.jnt_qposadr[j];
#pragma unroll
        for (int i = 0; i < 3; i++) { jp[jj][i] = M.jnt_pos[3 * j + i]; jx[jj][i] = M.jnt_axis[3 * j + i]; }
      }
      if (b == 0) {  // the world body: the frame the walk starts from
#pragma unroll
        for (int i = 0; i < 3; i++) S.xpos()[i] = M.body_pos[i];
#pragma unroll
        for (int i = 0; i < 4; i++) S.xquat()[i] = M.body_quat[i];
      }
      wave_sync();
      // one joint with its constants in registers: joint_step's operations, in its order
      auto joint_reg = [&](int j, int t, int qa, const REAL* jpos, const REAL* jaxis, REAL* pos, REAL* quat) {
        const REAL ql[4] = {S.jquat()[4 * j], S.jquat()[4 * j + 1], S.jquat()[4 * j + 2], S.jquat()[4 * j + 3]};
        REAL anchor[3], axis[3];
        if (t == JNT_FREE) {
#pragma unroll
          for (int i = 0; i < 3; i++) { anchor[i] = S.qpos()[qa + i]; pos[i] = S.qpos()[qa + i]; }
          axis[0] = 0; axis[1] = 0; axis[2] = 1;
#pragma unroll
          for (int i = 0; i < 4; i++) quat[i] = ql[i];
        } else {
          REAL r[3];
          rotate(jpos, quat, r);
#pragma unroll
          for (int i = 0; i < 3; i++) anchor[i] = r[i] + pos[i];
          rotate(jaxis, quat, axis);
          if (t == JNT_BALL || t == JNT_HINGE) {
            quat_mul(quat, ql, quat);
            rotate(jpos, quat, r);
#pragma unroll
            for (int i = 0; i < 3; i++) pos[i] = anchor[i] - r[i];
          } else {
            const REAL dq = ql[0];
#pragma unroll
            for (int i = 0; i < 3; i++) pos[i] = pos[i] + axis[i] * dq;
          }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) { S.xanchor()[3 * j + i] = anchor[i]; S.xaxis()[3 * j + i] = axis[i]; }
      };
      for (int L = 1; L <= md; L++) {  // (uniform trip count)
        if (body && depth == L) {
          REAL pos[3], quat[4];
#pragma unroll
          for (int i = 0; i < 3; i++) pos[i] = S.xpos()[3 * par + i];
#pragma unroll
          for (int i = 0; i < 4; i++) quat[i] = S.xquat()[4 * par + i];
          {
            REAL r[3];
            rotate(bp, quat, r);
#pragma unroll
            for (int i = 0; i < 3; i++) pos[i] = pos[i] + r[i];
            quat_mul(quat, bq, quat);
          }
#pragma unroll
          for (int jj = 0; jj < KJ; jj++) if (jj < jn) joint_reg(j0 + jj, jt[jj], jqa[jj], jp[jj], jx[jj], pos, quat);
          for (int jj = KJ; jj < jn; jj++) joint_step(j0 + jj, pos, quat, true);
#pragma unroll
          for (int i = 0; i < 3; i++) S.xpos()[3 * b + i] = pos[i];
#pragma unroll
          for (int i = 0; i < 4; i++) S.xquat()[4 * b + i] = quat[i];
        }
        wave_sync();
      }
      STAMP(9);
    }
    const bool jump = M.kin_tab != nullptr && M.nbody <= W;
    const int* const kin_anc = M.kin_tab;
    if (jump) {
      const int nb = M.nbody, md = M.max_depth;
      int R = 0;
      while ((1 << R) < md) R++;
      const REAL* const kin_start = reinterpret_cast<const REAL*>(reinterpret_cast<const unsigned char*>(M.kin_tab) + 8 * (((size_t)R * nb + 1) / 2));
      const int b = l;
      const bool body = b > 0 && b < nb;
      REAL pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
      int anc_r = 0;
      if (body) {
        anc_r = R > 0 ? kin_anc[b] : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) pos[i] = kin_start[7 * b + i];
#pragma unroll
        for (int i = 0; i < 4; i++) quat[i] = kin_start[7 * b + 3 + i];
        const int jn = M.body_jntnum[b], j0 = M.body_jntadr[b];
        for (int jj = 0; jj < jn; jj++) joint_step(j0 + jj, pos, quat, true);  // (anchors / axes in the parent's frame: carried to the world below)
      }
      if (b == 0) {  // the world body: the frame the walk starts from
#pragma unroll
        for (int i = 0; i < 3; i++) pos[i] = M.body_pos[i];
#pragma unroll
        for (int i = 0; i < 4; i++) quat[i] = M.body_quat[i];
      }
      if (b < nb) {
#pragma unroll
        for (int i = 0; i < 3; i++) S.xpos()[3 * b + i] = pos[i];
#pragma unroll
        for (int i = 0; i < 4; i++) S.xquat()[4 * b + i] = quat[i];
      }
      wave_sync();
      for (int r = 0; r < R; r++) {  // (uniform trip count)
        const int a = anc_r;
        REAL pa[3] = {0, 0, 0}, qa4[4] = {1, 0, 0, 0};
        if (a > 0) {
#pragma unroll
          for (int i = 0; i < 3; i++) pa[i] = S.xpos()[3 * a + i];
#pragma unroll
          for (int i = 0; i < 4; i++) qa4[i] = S.xquat()[4 * a + i];
        }
        if (r + 1 < R) anc_r = body ? kin_anc[(r + 1) * nb + b] : 0;
        wave_sync();  // every lane has read its ancestor's frame of this round
        if (a > 0) {
          REAL rr[3];
          rotate(pos, qa4, rr);
#pragma unroll
          for (int i = 0; i < 3; i++) pos[i] = pa[i] + rr[i];
          quat_mul(qa4, quat, quat);
#pragma unroll
          for (int i = 0; i < 3; i++) S.xpos()[3 * b + i] = pos[i];
#pragma unroll
          for (int i = 0; i < 4; i++) S.xquat()[4 * b + i] = quat[i];
        }
        wave_sync();
      }
    }
    for (int b = l; b < M.nbody; b += W) {
      REAL pos[3] = {M.body_pos[0], M.body_pos[1], M.body_pos[2]};
      REAL quat[4] = {M.body_quat[0], M.body_quat[1], M.body_quat[2], M.body_quat[3]};
      const int depth = M.body_depth[b], md = M.max_depth;
      if (jump || lvl) {
#pragma unroll
        for (int i = 0; i < 3; i++) pos[i] = S.xpos()[3 * b + i];
#pragma unroll
        for (int i = 0; i < 4; i++) quat[i] = S.xquat()[4 * b + i];
      } else {
      // the constants of level k + 1 are requested while level k is computed: the chain ids depend on (b, k) only
      int c_n = depth > 0 ? M.body_chain[b * md] : 0;
      REAL bp_n[3] = {M.body_pos[3 * c_n], M.body_pos[3 * c_n + 1], M.body_pos[3 * c_n + 2]};
      REAL bq_n[4] = {M.body_quat[4 * c_n], M.body_quat[4 * c_n + 1], M.body_quat[4 * c_n + 2], M.body_quat[4 * c_n + 3]};
      int jn_n = M.body_jntnum[c_n], j0_n = M.body_jntadr[c_n];
      for (int k = 0; k < md; k++) {  // uniform trip count, lanes past their depth idle
        const int jn = jn_n, j0 = j0_n;
        const REAL bp[3] = {bp_n[0], bp_n[1], bp_n[2]}, bq[4] = {bq_n[0], bq_n[1], bq_n[2], bq_n[3]};
        if (k + 1 < md) {
          c_n = (k + 1 < depth) ? M.body_chain[b * md + k + 1] : 0;
#pragma unroll
          for (int i = 0; i < 3; i++) bp_n[i] = M.body_pos[3 * c_n + i];
#pragma unroll
          for (int i = 0; i < 4; i++) bq_n[i] = M.body_quat[4 * c_n + i];
          jn_n = M.body_jntnum[c_n]; j0_n = M.body_jntadr[c_n];
        }
        if (k >= depth) continue;
        const bool own = (k == depth - 1);
        {
          REAL r[3];
          rotate(bp, quat, r);
#pragma unroll
          for (int i = 0; i < 3; i++) pos[i] = pos[i] + r[i];
          quat_mul(quat, bq, quat);
        }
        for (int jj = 0; jj < jn; jj++) joint_step(j0 + jj, pos, quat, own);
      }
      }
      if (M.nmocap > 0) {  // mocap bodies take the caller's pose after the tree pass (smooth.py:105-113); children of the world, no joints
        const int k = M.body_mocapid[b];
        if (k >= 0) {
          const REAL* mp = in.mocap_pos + (e * M.nmocap + k) * 3;
          const REAL* mq = in.mocap_quat + (e * M.nmocap + k) * 4;
#pragma unroll
          for (int i = 0; i < 3; i++) pos[i] = mp[i];
#pragma unroll
          for (int i = 0; i < 4; i++) quat[i] = mq[i];
          normalize_n<REAL, 4>(quat);
        }
      }
#pragma unroll
      for (int i = 0; i < 3; i++) S.xpos()[3 * b + i] = pos[i];
#pragma unroll
      for (int i = 0; i < 4; i++) S.xquat()[4 * b + i] = quat[i];
      if (M.lds_diet) {  // the two 3 x 3 frames are not staged: stored from here, formed again from xquat where com_pos reads them (the same operations on the same inputs)
        REAL xm[9], xim[9];
        quat_to_mat(quat, xm);
        local_to_global(pos, quat, M.body_ipos + 3 * b, M.body_iquat + 4 * b, S.xipos() + 3 * b, xim);
        if (out.xmat) {
#pragma unroll
          for (int i = 0; i < 9; i++) MJH_NT_STORE(xm[i], &out.xmat[(e * M.nbody + b) * 9 + i]);
        }
        if (out.ximat) {
#pragma unroll
          for (int i = 0; i < 9; i++) MJH_NT_STORE(xim[i], &out.ximat[(e * M.nbody + b) * 9 + i]);
        }
      } else {
      quat_to_mat(quat, S.xmat() + 9 * b);
      local_to_global(pos, quat, M.body_ipos + 3 * b, M.body_iquat + 4 * b, S.xipos() + 3 * b, S.ximat() + 9 * b);
      }
    }
    wave_sync();
    if (jump) {  // anchors and axes of the joints of bodies below the first level: from the parent's frame to the world (the mocap override above touches no parent: mocap bodies have no children here)
      for (int j = l; j < M.njnt; j += W) {
        const int p = kin_anc[M.jnt_bodyid[j]];  // (round 0 of the table: the parent, 0 = the world; one level deep models have no table rows and no such joints)
        if (M.max_depth > 1 && p > 0) {
          const REAL* pq = S.xquat() + 4 * p;
          const REAL* pp = S.xpos() + 3 * p;
          REAL al[3], xl[3], r[3], x[3];
#pragma unroll
          for (int i = 0; i < 3; i++) { al[i] = S.xanchor()[3 * j + i]; xl[i] = S.xaxis()[3 * j + i]; }
          rotate(al, pq, r);
          rotate(xl, pq, x);
#pragma unroll
          for (int i = 0; i < 3; i++) { S.xanchor()[3 * j + i] = pp[i] + r[i]; S.xaxis()[3 * j + i] = x[i]; }
        }
      }
      wave_sync();
    }
    STAMP(2);
    // the frames go out now, ahead of the geom / site / camera loops: a phase's leaf stores are bursts of tens of MB issued by every wave at the same
    // moment, and the first table read behind one waits until L2 has taken it (vmcnt is in order) -- several smaller bursts with arithmetic between them drain
    // in the background where one large one does not
    // cameras that track / target a subtree's centre of mass read the CALLER's subtree_com (previous step, smooth.py:162-166): requested here, ahead of this stage's stores -- behind
    // them the read waited for 25 leaf stores to land (vmcnt is in order)
    REAL cam_com[3] = {0, 0, 0};
    if (with_cams && l < M.ncam && in.subtree_com) {
      const int mode = M.cam_mode[l], tgt = M.cam_targetbodyid[l];
      const int sb = mode == CAM_TRACKCOM ? M.cam_bodyid[l] : ((mode == CAM_TARGETBODYCOM && tgt >= 0) ? tgt : -1);
      if (sb >= 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) cam_com[i] = in.subtree_com[(e * M.nbody + sb) * 3 + i];
      }
    }
    if (!(DEFER && W > 16 && M.kv_defer)) frame_stores();  // (fused with the velocity stage: they go out in front of its LDS-only sweep, see velocity())
    if constexpr (XK) if (com_first) { com_pos<DEFER>(); wave_sync(); }
    // normalised free / ball quaternions are written back into qpos (smooth.py:60-70); one lane per joint
    for (int j = l; j < M.njnt; j += W) {
      const int t = M.jnt_type[j], qa = M.jnt_qposadr[j];
      if (t == JNT_FREE || t == JNT_BALL) {
        const int o = (t == JNT_FREE) ? qa + 3 : qa;
#pragma unroll
        for (int i = 0; i < 4; i++) S.qpos()[o + i] = S.jquat()[4 * j + i];
      }
    }
    for (int g = l; g < M.ngeom; g += W) {
      const int b = M.geom_bodyid[g];
      REAL p[3], mat[9];
      local_to_global(S.xpos() + 3 * b, S.xquat() + 4 * b, M.geom_pos + 3 * g, M.geom_quat + 4 * g, p, mat);
      if constexpr (KEEPG) if (g == l) {
#pragma unroll
        for (int i = 0; i < 3; i++) ho_g[i] = p[i];
#pragma unroll
        for (int i = 0; i < 9; i++) ho_g[3 + i] = mat[i];
      }
      if (M.nt_all) {  // (whole-pass kernel: nothing reads the geom frames back -- they cross the seam in registers)
        if (out.geom_xpos) for (int i = 0; i < 3; i++) MJH_NT_STORE(p[i], &out.geom_xpos[(e * M.ngeom + g) * 3 + i]);
        if (out.geom_xmat) for (int i = 0; i < 9; i++) MJH_NT_STORE(mat[i], &out.geom_xmat[(e * M.ngeom + g) * 9 + i]);
      } else {
      if (out.geom_xpos) for (int i = 0; i < 3; i++) out.geom_xpos[(e * M.ngeom + g) * 3 + i] = p[i];
      if (out.geom_xmat) for (int i = 0; i < 9; i++) out.geom_xmat[(e * M.ngeom + g) * 9 + i] = mat[i];
      }
    }
    {
      if (out.site_xpos || out.site_xmat)  // (RK4 stages 1..3 keep no site frames: nothing reads them there)
      for (int s = l; s < M.nsite; s += W) {
        const int b = M.site_bodyid[s];
        REAL p[3], mat[9];
        local_to_global(S.xpos() + 3 * b, S.xquat() + 4 * b, M.site_pos + 3 * s, M.site_quat + 4 * s, p, mat);
        if (out.site_xpos) for (int i = 0; i < 3; i++) out.site_xpos[(e * M.nsite + s) * 3 + i] = p[i];
        if (out.site_xmat) for (int i = 0; i < 9; i++) out.site_xmat[(e * M.nsite + s) * 9 + i] = mat[i];
      }
      if (with_cams) {
        for (int c = l; c < M.ncam; c += W) {  // smooth.py:139-198
          const int b = M.cam_bodyid[c], mode = M.cam_mode[c], tgt = M.cam_targetbodyid[c];
          REAL cp[3], cm[9];
          local_to_global(S.xpos() + 3 * b, S.xquat() + 4 * b, M.cam_pos + 3 * c, M.cam_quat + 4 * c, cp, cm);
          if (mode == CAM_TRACK) {
            for (int i = 0; i < 3; i++) cp[i] = S.xpos()[3 * b + i] + M.cam_pos0[3 * c + i];
            for (int i = 0; i < 9; i++) cm[i] = M.cam_mat0[9 * c + i];
          } else if (mode == CAM_TRACKCOM) {
            // subtree_com of the CALLER's Data (previous step), smooth.py:162-166
            REAL r[3];
            rotate(M.cam_pos + 3 * c, S.xquat() + 4 * b, r);
            for (int i = 0; i < 3; i++) cp[i] = (c == l ? cam_com[i] : (in.subtree_com ? in.subtree_com[(e * M.nbody + b) * 3 + i] : (REAL)0)) + r[i];
          } else if ((mode == CAM_TARGETBODY || mode == CAM_TARGETBODYCOM) && tgt >= 0) {
            REAL tp[3];
            for (int i = 0; i < 3; i++)
              tp[i] = (mode == CAM_TARGETBODY) ? S.xpos()[3 * tgt + i] : (c == l ? cam_com[i] : (in.subtree_com ? in.subtree_com[(e * M.nbody + tgt) * 3 + i] : (REAL)0));
            REAL f[3] = {tp[0] - cp[0], tp[1] - cp[1], tp[2] - cp[2]};
            normalize_n<REAL, 3>(f);
            REAL up_hint[3] = {0, 0, 1}, right[3], up[3];
            cross3(f, up_hint, right);
            normalize_n<REAL, 3>(right);
            cross3(right, f, up);
            for (int i = 0; i < 3; i++) { cm[3 * i + 0] = right[i]; cm[3 * i + 1] = up[i]; cm[3 * i + 2] = -f[i]; }
          }
          if (out.cam_xpos) for (int i = 0; i < 3; i++) out.cam_xpos[(e * M.ncam + c) * 3 + i] = cp[i];
          if (out.cam_xmat) for (int i = 0; i < 9; i++) out.cam_xmat[(e * M.ncam + c) * 9 + i] = cm[i];
        }
        for (int q = l; q < M.nlight; q += W) {  // :200-204
          const int b = M.light_bodyid[q];
          REAL r[3], dir[3];
          rotate(M.light_pos + 3 * q, S.xquat() + 4 * b, r);
          rotate(M.light_dir + 3 * q, S.xquat() + 4 * b, dir);
          if (out.light_xpos) for (int i = 0; i < 3; i++) out.light_xpos[(e * M.nlight + q) * 3 + i] = S.xpos()[3 * b + i] + r[i];
          if (out.light_xdir) for (int i = 0; i < 3; i++) out.light_xdir[(e * M.nlight + q) * 3 + i] = dir[i];
        }
      }
    }
    wave_sync();
    STAMP(3);
    put(out.qpos, S.qpos(), M.nq);
    STAMP(4);
  }

  // ---- com_pos (smooth.py:210-288) --------------------------------------------------------------------------------------------------------------
  template <bool DEFER = false>
  __device__ __forceinline__ void com_pos() {
    const int l = lane_here();
    const int nb = M.nbody;
    // subtree mass / mass-weighted position: bodies are in DFS order, a subtree is a contiguous id range
    // the per-body terms first (one global read of the mass per term, staged where cinert will be written later), so the
    // subtree loops below run on LDS only
    REAL* term = S.cinert();
    for (int w = l; w < nb * 4; w += W) {
      const int b = w >> 2, k = w & 3;
      const REAL mass = M.body_mass[b];
      term[w] = (k < 3) ? S.xipos()[3 * b + k] * mass : mass;
    }
    wave_sync();
    for (int w = l; w < nb * 4; w += W) {
      const int b = w >> 2, k = w & 3;
      const int end = M.body_subtree_end[b];
      REAL acc = 0;
      for (int d = end - 1; d >= b; d--) acc += term[4 * d + k];
      if (k < 3) S.sub_pos()[3 * b + k] = acc; else S.sub_mass()[b] = acc;
    }
    wave_sync();
    for (int w = l; w < nb * 3; w += W) {
      const int b = w / 3;
      const REAL ms = S.sub_mass()[b];
      const REAL den = ms > (REAL)MINVAL_CACHED ? ms : (REAL)MINVAL_CACHED;
      S.subtree_com()[w] = (ms < (REAL)mjMINVAL) ? S.xipos()[w] : S.sub_pos()[w] / den;
    }
    wave_sync();
    STAMP(5);
    for (int b = l; b < nb; b += W) {  // inert_com :236-243
      const REAL* rc = S.subtree_com() + 3 * M.body_rootid[b];
      const REAL off[3] = {S.xipos()[3 * b] - rc[0], S.xipos()[3 * b + 1] - rc[1], S.xipos()[3 * b + 2] - rc[2]};
      const REAL mass = M.body_mass[b];
      REAL xi[9];  // (a register copy either way: a pointer that may address a private array OR the arena would put the array in scratch memory)
      if (M.lds_diet) { REAL q_[4]; quat_mul(S.xquat() + 4 * b, M.body_iquat + 4 * b, q_); quat_to_mat(q_, xi); }  // support.local_to_global's orientation half, as the kinematics formed it
      else {
#pragma unroll
        for (int i = 0; i < 9; i++) xi[i] = S.ximat()[9 * b + i];
      }
      const REAL* inr = M.body_inertia + 3 * b;
      const REAL h[3][3] = {{0, -off[2], off[1]}, {off[2], 0, -off[0]}, {-off[1], off[0], 0}};
      REAL I[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          REAL s = 0, hh = 0;
#pragma unroll
          for (int k = 0; k < 3; k++) s += (xi[3 * i + k] * inr[k]) * xi[3 * j + k];
#pragma unroll
          for (int k = 0; k < 3; k++) hh += h[i][k] * h[j][k];
          I[i][j] = s + hh * mass;
        }
      REAL* ci = S.cinert() + 10 * b;
      ci[0] = I[0][0]; ci[1] = I[1][1]; ci[2] = I[2][2]; ci[3] = I[0][1]; ci[4] = I[0][2]; ci[5] = I[1][2];
      ci[6] = off[0] * mass; ci[7] = off[1] * mass; ci[8] = off[2] * mass; ci[9] = mass;
    }
    for (int j = l; j < M.njnt; j += W) {  // cdof_fn :250-273
      const int b = M.jnt_bodyid[j], t = M.jnt_type[j];
      int d = M.jnt_dofadr[j];
      const REAL* rc = S.subtree_com() + 3 * M.body_rootid[b];
      const REAL off[3] = {rc[0] - S.xanchor()[3 * j], rc[1] - S.xanchor()[3 * j + 1], rc[2] - S.xanchor()[3 * j + 2]};
      if (t == JNT_FREE || t == JNT_BALL) {
        if (t == JNT_FREE) {
          for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) S.cdof()[6 * (d + r) + k] = (k == 3 + r) ? (REAL)1 : (REAL)0;
          d += 3;
        }
        REAL xmb[9];
        if (M.lds_diet) quat_to_mat(S.xquat() + 4 * b, xmb);
        else {
#pragma unroll
          for (int i = 0; i < 9; i++) xmb[i] = S.xmat()[9 * b + i];
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const REAL a[3] = {xmb[r], xmb[3 + r], xmb[6 + r]};
          REAL c[3];
          cross3(a, off, c);
          for (int k = 0; k < 3; k++) { S.cdof()[6 * (d + r) + k] = a[k]; S.cdof()[6 * (d + r) + 3 + k] = c[k]; }
        }
      } else if (t == JNT_HINGE) {
        REAL c[3];
        cross3(S.xaxis() + 3 * j, off, c);
        for (int k = 0; k < 3; k++) { S.cdof()[6 * d + k] = S.xaxis()[3 * j + k]; S.cdof()[6 * d + 3 + k] = c[k]; }
      } else {
        for (int k = 0; k < 3; k++) { S.cdof()[6 * d + k] = 0; S.cdof()[6 * d + 3 + k] = S.xaxis()[3 * j + k]; }
      }
    }
    wave_sync();
    STAMP(6);
    if (!(DEFER && W > 16 && M.kv_defer)) com_stores();
    STAMP(7);
  }

  // ---- crb + make_m + factor_m (smooth.py:291-332, support.make_m :50-80) ------------------------------------------------------------
  // MAXN: the largest nv the calling instantiation serves (the register Cholesky variants for more rows are not compiled: they would set the kernel's register allocation without ever running)
  template <bool FUSED = false, int MAXN = (W == 16 ? 16 : 32)>
  __device__ __forceinline__ void crb_factor() {
    const int l = lane_here();
    const int nb = M.nbody, nv = M.nv;
    STAMP0();
    if (!FUSED) {  // (fused behind the kinematics: cinert and cdof are still in the arena)
      REAL* const dst[2] = {S.cinert(), S.cdof()};
      const REAL* const src[2] = {out.cinert, out.cdof};
      const int cnt[2] = {10 * nb, 6 * nv};
      multi_load<W, 2, 3>(dst, src, cnt, e);
    }
    wave_sync();
    STAMP(11);
    for (int w = l; w < nb * 10; w += W) {
      const int b = w / 10, k = w - 10 * b;
      REAL acc = 0;
      if (b > 0) {
        const int end = M.body_subtree_end[b];
        for (int d = end - 1; d >= b; d--) acc += S.cinert()[10 * d + k];
      }
      S.crb()[w] = acc;  // crb[0] = 0 (smooth.py:300-301)
    }
    wave_sync();
    STAMP(12);
    for (int d = l; d < nv; d += W) inert_mul(S.crb() + 10 * M.dof_bodyid[d], S.cdof() + 6 * d, S.crb_cdof() + 6 * d);
    wave_sync();
    STAMP(13);
    // qM (support.make_m :50-80) is zero except for dof pairs on one ancestor path: one lane per structurally non-zero
    // lower-triangle entry computes it into the packed copy, then the full symmetric row is written out through a
    // slot table (each address stored exactly once, coalesced).
    for (int w = l; w < (nv * (nv + 1)) / 2; w += W) S.qMp()[w] = 0;
    wave_sync();
    for (int w = l; w < M.nqmpair; w += W) {
      const int pk = M.qm_pair[w], i = pk >> 8, j = pk & 0xff;  // j <= i, and j is an ancestor-or-self dof of i
      REAL s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) s += S.crb_cdof()[6 * i + k] * S.cdof()[6 * j + k];
      if (i == j) s = s + M.dof_armature[i];
      S.qMp()[tri_at<true>(i, j, nv)] = s;
    }
    wave_sync();
    if (M.has_ten_armature) {  // smooth.tendon_armature :500-522: qM += J^T diag(armature) J, a constant for fixed tendons (every entry, not only the tree's)
      for (int w = l; w < (nv * (nv + 1)) / 2; w += W) S.qMp()[w] = S.qMp()[w] + M.ten_JTAJ[w];
      wave_sync();
    }
    if (out.qM) {
      REAL* gM = out.qM + e * nv * nv;
      // (entry (i, j) is slot (max, min) of the packed copy, whose structurally zero entries hold the +0 they were initialised with: no table read per
      // pass -- behind the previous pass's stores each one waited for them to land, twelve times for the humanoid)
      for (int w = l; w < nv * nv; w += W) {
        int i, j;
        split_index(w, nv, M.inv_nv, i, j);
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        gM[w] = S.qMp()[(hi * (hi + 1)) / 2 + lo];
      }
    }
    STAMP(14);
    putnt(out.crb, S.crb(), 10 * nb);
    wave_sync();
    STAMP(15);
    chol_factor<W, REAL, MAXN, true>(S.qMp(), S.qLD(), nv);            // S.qLD() overlays the arrays above (lds_carve).  (Four environments per wavefront: nv <= 16 -- the 24 / 28 / 32-row register variants are not compiled in)
    STAMP(16);
    put(out.qLD, S.qLD(), nv * nv);
    STAMP(17);
  }

  // ---- collision (collision_driver.py:800-875, collision_primitive.py, math.py:506-569): one lane per static geom pair ------------------
  __device__ __forceinline__ static void plane_sphere_(const REAL* n, const REAL* ppos, const REAL* spos, REAL r, REAL& dist, REAL* pos) {
    const REAL d[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
    dist = dot3(d, n) - r;
#pragma unroll
    for (int i = 0; i < 3; i++) pos[i] = spos[i] - n[i] * (r + (REAL)0.5 * dist);
  }
  __device__ __forceinline__ static void sphere_sphere_(const REAL* p1, REAL r1, const REAL* p2, REAL r2, REAL& dist, REAL* pos, REAL* n) {
#pragma unroll
    for (int i = 0; i < 3; i++) n[i] = p2[i] - p1[i];
    REAL d = normalize_n<REAL, 3>(n);
    if (d == 0) { n[0] = 1; n[1] = 0; n[2] = 0; }
    d = d - (r1 + r2);
#pragma unroll
    for (int i = 0; i < 3; i++) pos[i] = p1[i] + n[i] * (r1 + d * (REAL)0.5);
    dist = d;
  }
  __device__ __forceinline__ static void closest_segment_point(const REAL* a, const REAL* b, const REAL* pt, REAL* o) {
    const REAL ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const REAL pa[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
    REAL t = dot3(pa, ab) / (dot3(ab, ab) + (REAL)1e-6);
    t = t < 0 ? (REAL)0 : (t > 1 ? (REAL)1 : t);
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a[i] + t * ab[i];
  }
  __device__ __forceinline__ static void closest_segment_to_segment(const REAL* a0, const REAL* a1, const REAL* b0, const REAL* b1, REAL* best_a, REAL* best_b) {
    REAL dir_a[3], dir_b[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { dir_a[i] = a1[i] - a0[i]; dir_b[i] = b1[i] - b0[i]; }
    const REAL len_a = normalize_n<REAL, 3>(dir_a), len_b = normalize_n<REAL, 3>(dir_b);
    const REAL hla = len_a * (REAL)0.5, hlb = len_b * (REAL)0.5;
    REAL a_mid[3], b_mid[3], trans[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { a_mid[i] = a0[i] + dir_a[i] * hla; b_mid[i] = b0[i] + dir_b[i] * hlb; trans[i] = a_mid[i] - b_mid[i]; }
    const REAL dadb = dot3(dir_a, dir_b), dat = dot3(dir_a, trans), dbt = dot3(dir_b, trans);
    const REAL denom = 1 - dadb * dadb;
    const REAL ota = (-dat + dadb * dbt) / (denom + (REAL)1e-6);
    const REAL otb = dbt + ota * dadb;
    const REAL ta = ota < -hla ? -hla : (ota > hla ? hla : ota);
    const REAL tb = otb < -hlb ? -hlb : (otb > hlb ? hlb : otb);
#pragma unroll
    for (int i = 0; i < 3; i++) { best_a[i] = a_mid[i] + dir_a[i] * ta; best_b[i] = b_mid[i] + dir_b[i] * tb; }
    REAL new_a[3], new_b[3];
    closest_segment_point(a0, a1, best_b, new_a);
    closest_segment_point(b0, b1, best_a, new_b);
    REAL d1 = 0, d2 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { REAL x = best_b[i] - new_a[i]; d1 += x * x; }
#pragma unroll
    for (int i = 0; i < 3; i++) { REAL x = best_a[i] - new_b[i]; d2 += x * x; }
    if (d1 < d2) { for (int i = 0; i < 3; i++) best_a[i] = new_a[i]; }
    else { for (int i = 0; i < 3; i++) best_b[i] = new_b[i]; }
  }

  bool con_inputs_loaded_ = false;  // collision() already fetched what make_constraint() reads (plain instantiation)
  bool ho_on_ = false;              // whole-pass kernel: this model's constraint-stage inputs travel on chip (DevModel::all_handoff)
  // HANDOFF (whole-pass kernel): the geom frames, qvel, subtree_com and cdof are already in the arena -- the first half of the kernel handed them over on chip
  // DEFER_CONST: the caller copies the model-constant contact leaves itself, behind the rows (contact_const_stores)
  template <int PRE_NMAX = 0, bool HANDOFF = false, bool DEFER_CONST = false>
  __device__ __forceinline__ void collision(Sol2Pre<REAL, (PRE_NMAX > 0 ? PRE_NMAX : 1)>* pre = nullptr) {
    const int l = lane_here();
    if (HANDOFF && ho_on_) {
      con_inputs_loaded_ = true;
    } else
    if (!FRIC && (KA.stages & 0x78) && M.nefc > 0) {
      // plain instantiation with the rows to follow: qvel, subtree_com and cdof ride in the same round trip as the geom frames -- loaded at the
      // head of make_constraint() they were a second, fully exposed trip behind the narrow phase (16 k of the phase's 77 k cycles on the humanoid)
      const bool from_in = !KA.state_from_cur;
      REAL* const dst[5] = {S.geom_xpos(), S.geom_xmat(), S.qvel(), S.subtree_com(), S.cdof()};
      const REAL* const src[5] = {out.geom_xpos, out.geom_xmat, from_in ? in.qvel : KA.cur.qvel, out.subtree_com, out.cdof};
      const int cnt[5] = {3 * M.ngeom, 9 * M.ngeom, M.nv, 3 * M.nbody, 6 * M.nv};
      multi_load<W, 5, 3>(dst, src, cnt, e);
      con_inputs_loaded_ = true;
    } else {
      REAL* const dst[2] = {S.geom_xpos(), S.geom_xmat()};
      const REAL* const src[2] = {out.geom_xpos, out.geom_xmat};
      const int cnt[2] = {3 * M.ngeom, 9 * M.ngeom};
      multi_load<W, 2, 3>(dst, src, cnt, e);
    }
    if (M.ncvxpair > 0) {  // box / mesh pairs were narrow-phased by mjh_convex_kernel (mjh_convex.h) into their contact slots
      if (FRIC && M.topk) {  // ... or, with max_contact_points, into the candidate arrays of the workspace (selection below)
        const int64_t B = KA.B, nc = M.ncand;
        row_load<W>(S.con_dist(), KA.cand, (int)nc, e);
        row_load<W>(S.con_pos(), KA.cand + B * nc, 3 * (int)nc, e);
        row_load<W>(S.con_frame(), KA.cand + 4 * B * nc, 9 * (int)nc, e);
      } else {
        row_load<W>(S.con_dist(), out.contact_dist, M.ncon, e);
        row_load<W>(S.con_pos(), out.contact_pos, 3 * M.ncon, e);
        row_load<W>(S.con_frame(), out.contact_frame, 9 * M.ncon, e);
      }
    }
#ifndef MJH_CS_PF
#define MJH_CS_PF 3  /* where the fused constraint + solver kernel requests the solver's inputs from the leaves of earlier launches: 0 behind the narrow phase's own loads (the narrow phase then waits for them, vmcnt is in order: kernel 78.0 - 83.6 us), 1 behind the narrow phase (78.0), 2 behind the contact rows (76.0), 3 behind the whole constraint stage (75.8: requested early they only hold registers -- the constraint stage has little arithmetic to hide them under) */
#endif
#if MJH_CS_PF == 0
    if constexpr (PRE_NMAX > 0) sol2_prefetch<PRE_NMAX>(*pre);  // fused constraint + solver kernel: requested BEHIND this stage's own inputs (vmcnt is in order: the narrow phase does not wait for them)
#endif
    wave_sync();
    STAMP(22);
    // Workspace Data of an RK4 stage (1..3): nothing of a contact leaves the launch but whether it is active (dist < includemargin) and, for the
    // active ones, the rows built from pos / frame.  A pair whose bounding spheres are further apart than the margin cannot be active, whatever its
    // exact distance: those pairs get the sphere gap as their dist (inactive, like the exact value) and the rest are packed into one dense list, so
    // the lanes run the narrow phase over the near pairs only (the ant has 56 capsule pairs per environment and a handful of them anywhere near
    // each other: two passes of the lanes became one).  Stage 0 and Euler steps publish dist / pos / frame of every contact and take no part in this.
    const bool cull = !FRIC && DIRECT && M.pair_cull_on && KA.rk_stage > 0 && M.ncvxpair == 0 && M.con_rows > 0;  // (the consumer builds rows of ACTIVE contacts only: pos / frame of the others are never read)
    int npair_run = M.npair;
    int* const near_list = reinterpret_cast<int*>(S.i_con_act());  // (free until make_constraint's compaction; ncon >= npair entries)
    if (cull) {
      int nnear = 0;
      for (int p0 = 0; p0 < M.npair; p0 += W) {
        const int p = p0 + l;
        bool near = false;
        if (p < M.npair) {
          // DevModel::pair_cull: [reach^2, r1 + r2] per pair -- reach = 1.01 (r1 + r2 + margin) + 1e-3 over the bounding spheres (a hundred thousand
          // times the rounding of this test), negative for the pairs that are always narrow-phased (planes, hulls)
          const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p], k = M.pair_ncon[p], c0 = M.pair_dst[p * MJH_MAX_PAIR_CONTACTS], c1 = M.pair_dst[p * MJH_MAX_PAIR_CONTACTS + 1];
          const REAL reach2 = M.pair_cull[2 * p], rsum = M.pair_cull[2 * p + 1];
          const REAL *p1 = S.geom_xpos() + 3 * g1, *p2 = S.geom_xpos() + 3 * g2;
          const REAL d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
          const REAL d2 = dot3(d, d);
          near = !(reach2 >= 0 && d2 > reach2);
          if (!near) {
            const REAL gap = r_sqrt<REAL>(d2) - rsum;  // <= the exact distance, > includemargin: inactive like the exact value
            S.con_dist()[c0] = gap;
            if (k > 1) S.con_dist()[c1] = gap;
          }
        }
        int tot;
        const int at = sub_prefix_count<W>(near, tot) + nnear;
        if (near) near_list[at] = p;
        nnear += tot;
      }
      npair_run = nnear;
      wave_sync();
    }
    STAMP(24);
    for (int it = l; it < npair_run; it += W) {
      const int p = cull ? near_list[it] : it;
      const int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p], fn = M.pair_fn[p], k = M.pair_ncon[p];
      if (fn >= MJH_FN_PLANE_CONVEX) continue;
      const REAL *p1 = S.geom_xpos() + 3 * g1, *m1 = S.geom_xmat() + 9 * g1, *s1 = M.geom_size + 3 * g1;
      const REAL *p2 = S.geom_xpos() + 3 * g2, *m2 = S.geom_xmat() + 9 * g2, *s2 = M.geom_size + 3 * g2;
      REAL dist[2], pos[2][3], frame[9];
      if (fn == MJH_FN_PLANE_SPHERE) {
        const REAL n[3] = {m1[2], m1[5], m1[8]};
        plane_sphere_(n, p1, p2, s2[0], dist[0], pos[0]);
        make_frame(n, frame);
      } else if (fn == MJH_FN_PLANE_CAPSULE) {  // collision_primitive.py:48-74
        const REAL n[3] = {m1[2], m1[5], m1[8]}, axis[3] = {m2[2], m2[5], m2[8]};
        const REAL na = dot3(n, axis);
        REAL b[3];
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = axis[i] - n[i] * na;
        const REAL bn = normalize_n<REAL, 3>(b);
        if (bn < (REAL)0.5) {
          b[0] = 0; b[1] = 0; b[2] = 0;
          if ((REAL)-0.5 < n[1] && n[1] < (REAL)0.5) b[1] = 1; else b[2] = 1;
        }
        REAL c[3];
        cross3(n, b, c);
#pragma unroll
        for (int i = 0; i < 3; i++) { frame[i] = n[i]; frame[3 + i] = b[i]; frame[6 + i] = c[i]; }
        const REAL seg[3] = {axis[0] * s2[1], axis[1] * s2[1], axis[2] * s2[1]};
#pragma unroll
        for (int q = 0; q < 2; q++) {
          REAL sp[3];
#pragma unroll
          for (int i = 0; i < 3; i++) sp[i] = p2[i] + (q == 0 ? seg[i] : -seg[i]);
          plane_sphere_(n, p1, sp, s2[0], dist[q], pos[q]);
        }
      } else if (fn == MJH_FN_SPHERE_SPHERE) {
        REAL n[3];
        sphere_sphere_(p1, s1[0], p2, s2[0], dist[0], pos[0], n);
        make_frame(n, frame);
      } else if (fn == MJH_FN_SPHERE_CAPSULE) {  // :195-201
        const REAL axis[3] = {m2[2], m2[5], m2[8]};
        REAL a[3], b[3], pt[3], n[3];
#pragma unroll
        for (int i = 0; i < 3; i++) { REAL sg = axis[i] * s2[1]; a[i] = p2[i] - sg; b[i] = p2[i] + sg; }
        closest_segment_point(a, b, p1, pt);
        sphere_sphere_(p1, s1[0], pt, s2[0], dist[0], pos[0], n);
        make_frame(n, frame);
      } else if (fn == MJH_FN_CAPSULE_CAPSULE) {  // :204-221
        const REAL ax1[3] = {m1[2], m1[5], m1[8]}, ax2[3] = {m2[2], m2[5], m2[8]};
        REAL a0[3], a1[3], b0[3], b1[3], pt1[3], pt2[3], n[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          REAL sg1 = ax1[i] * s1[1], sg2 = ax2[i] * s2[1];
          a0[i] = p1[i] - sg1; a1[i] = p1[i] + sg1; b0[i] = p2[i] - sg2; b1[i] = p2[i] + sg2;
        }
        closest_segment_to_segment(a0, a1, b0, b1, pt1, pt2);
        sphere_sphere_(pt1, s1[0], pt2, s2[0], dist[0], pos[0], n);
        make_frame(n, frame);
      } else {
        dist[0] = dist[1] = 1;
        for (int i = 0; i < 3; i++) pos[0][i] = pos[1][i] = 0;
        for (int i = 0; i < 9; i++) frame[i] = 0;
      }
      for (int q = 0; q < k && q < 2; q++) {
        const int c = M.pair_dst[p * MJH_MAX_PAIR_CONTACTS + q];
        S.con_dist()[c] = dist[q];
        for (int i = 0; i < 3; i++) S.con_pos()[3 * c + i] = pos[q][i];
        for (int i = 0; i < 9; i++) S.con_frame()[9 * c + i] = frame[i];
      }
    }
    wave_sync();
    STAMP(20);
    if (FRIC && M.topk) {  // (models with max_contact_points run the general constraint kernel)
      // max_contact_points (collision_driver.py:822-840): keep the ncon candidates with the smallest dist -- torch.topk(-dist): closest
      // first; equal distances are ordered by candidate index here (torch leaves that order to its partial sort) -- then the static
      // permutation of the argsort over their (equal) condims.  One candidate per lane, rank by counting.
      const int nk = M.ncon, ncand = M.ncand;
      for (int q = l; q < ncand; q += W) {
        const REAL dq = S.con_dist()[q];
        int rank = 0;
        for (int q2 = 0; q2 < ncand; q2++) { const REAL d2 = S.con_dist()[q2]; rank += (d2 < dq) || (d2 == dq && q2 < q); }
        if (rank < nk) con_src_lds()[M.topk_slot[rank]] = q;
      }
      wave_sync();
      for (int c = l; c < nk; c += W) {  // every contact leaf is per environment now: gathered through the kept candidate
        const int q = con_src_lds()[c];
        if (out.contact_dist) out.contact_dist[e * nk + c] = S.con_dist()[q];
        for (int i = 0; i < 3; i++) if (out.contact_pos) out.contact_pos[(e * nk + c) * 3 + i] = S.con_pos()[3 * q + i];
        for (int i = 0; i < 9; i++) if (out.contact_frame) out.contact_frame[(e * nk + c) * 9 + i] = S.con_frame()[9 * q + i];
        if (out.contact_includemargin) out.contact_includemargin[e * nk + c] = M.con_includemargin[q];
        for (int i = 0; i < 5; i++) if (out.contact_friction) out.contact_friction[(e * nk + c) * 5 + i] = M.con_friction[5 * q + i];
        for (int i = 0; i < 2; i++) if (out.contact_solref) out.contact_solref[(e * nk + c) * 2 + i] = M.con_solref[2 * q + i];
        for (int i = 0; i < 2; i++) if (out.contact_solreffriction) out.contact_solreffriction[(e * nk + c) * 2 + i] = M.con_solreffriction[2 * q + i];
        for (int i = 0; i < 5; i++) if (out.contact_solimp) out.contact_solimp[(e * nk + c) * 5 + i] = M.con_solimp[5 * q + i];
        if (out.contact_dim) out.contact_dim[e * nk + c] = M.con_dim[q];
        if (out.contact_geom1) out.contact_geom1[e * nk + c] = M.con_geom1[q];
        if (out.contact_geom2) out.contact_geom2[e * nk + c] = M.con_geom2[q];
        if (out.contact_geom) { out.contact_geom[(e * nk + c) * 2] = M.con_geom1[q]; out.contact_geom[(e * nk + c) * 2 + 1] = M.con_geom2[q]; }
        if (out.contact_efc_address) out.contact_efc_address[e * nk + c] = M.con_efc_address[c];
      }
      return;
    }
    {
      const int nc = M.ncon;
      if (M.nt_all) putnt(out.contact_dist, S.con_dist(), nc); else put(out.contact_dist, S.con_dist(), nc);
      putnt(out.contact_pos, S.con_pos(), 3 * nc); putnt(out.contact_frame, S.con_frame(), 9 * nc);
      STAMP(21);
      if constexpr (PRE_NMAX > 0 || DEFER_CONST) return;  // fused constraint + solver kernel: the constant leaves are copied at the kernel's end (contact_const_stores), off the path to the solve
      contact_const_stores();
    }
  }
  __device__ __forceinline__ void contact_const_stores() {
    const int l = lane_here();
    // (RK4 stages 1..3 keep none of these leaves: without this test the integer tables were still READ there -- 4.5 us of the ant's constraint phase per stage)
    if (!(out.contact_includemargin || out.contact_solref || out.contact_solreffriction || out.contact_friction || out.contact_solimp || out.contact_dim || out.contact_geom1 ||
          out.contact_geom2 || out.contact_geom || out.contact_efc_address)) return;
    {
      const int nc = M.ncon;
      // model-constant contact leaves (collision_driver.py:553-568 / :691-793)
      // (all the reads first, then the stores: a read between two stores waits for the first store to land -- vmcnt is in order -- and the optimiser
      // may not move it up past a store it cannot prove distinct)
      // in three groups, each with all of its reads in flight before its first store (the ant has 60 contacts: 900 constants per environment, ten passes of 32 lanes
      // for the wide leaves -- copied pass by pass, every pass's reads waited for the previous pass's stores: 64 us of the ant's stage-0 launch)
      {
        REAL* const dst[3] = {out.contact_includemargin, out.contact_solref, out.contact_solreffriction};
        const REAL* const src[3] = {M.con_includemargin, M.con_solref, M.con_solreffriction};
        const int cnt[3] = {nc, 2 * nc, 2 * nc};
        multi_copy_const<W, 3, 4>(dst, src, cnt, e);
      }
      {
        REAL* const dst[1] = {out.contact_friction};
        const REAL* const src[1] = {M.con_friction};
        const int cnt[1] = {5 * nc};
        multi_copy_const<W, 1, 10>(dst, src, cnt, e);
      }
      {
        REAL* const dst[1] = {out.contact_solimp};
        const REAL* const src[1] = {M.con_solimp};
        const int cnt[1] = {5 * nc};
        multi_copy_const<W, 1, 10>(dst, src, cnt, e);
      }
      for (int c0 = l; c0 < nc; c0 += 2 * W) {  // two passes of the lanes per trip: their reads first
        const int c1 = c0 + W;
        const bool h1 = c1 < nc;
        const int dim0 = M.con_dim[c0], g10 = M.con_geom1[c0], g20 = M.con_geom2[c0], adr0 = M.con_efc_address[c0];
        const int dim1 = h1 ? M.con_dim[c1] : 0, g11 = h1 ? M.con_geom1[c1] : 0, g21 = h1 ? M.con_geom2[c1] : 0, adr1 = h1 ? M.con_efc_address[c1] : 0;
        if (out.contact_dim) { out.contact_dim[e * nc + c0] = dim0; if (h1) out.contact_dim[e * nc + c1] = dim1; }
        if (out.contact_geom1) { out.contact_geom1[e * nc + c0] = g10; if (h1) out.contact_geom1[e * nc + c1] = g11; }
        if (out.contact_geom2) { out.contact_geom2[e * nc + c0] = g20; if (h1) out.contact_geom2[e * nc + c1] = g21; }
        if (out.contact_geom) {
          out.contact_geom[(e * nc + c0) * 2] = g10; out.contact_geom[(e * nc + c0) * 2 + 1] = g20;
          if (h1) { out.contact_geom[(e * nc + c1) * 2] = g11; out.contact_geom[(e * nc + c1) * 2 + 1] = g21; }
        }
        if (out.contact_efc_address) { out.contact_efc_address[e * nc + c0] = adr0; if (h1) out.contact_efc_address[e * nc + c1] = adr1; }
      }
    }
  }
  __device__ __forceinline__ int* con_src_lds() const { return reinterpret_cast<int*>(S.i_con_src()); }
  // candidate behind contact slot c: the slot itself unless max_contact_points selected per environment
  __device__ __forceinline__ int con_src(int c) const { return (FRIC && M.topk) ? con_src_lds()[c] : c; }

  // ---- constraint rows (constraint.py:600-768) ---------------------------------------------------------------------------------------------------
  // support.jac :138-153 restricted to one dof: jacp / jacr of `point` on `body`, masked to ancestor dofs
  // BIG: the instantiation may serve models with more than 64 dofs (masks of several words); the headline instantiations keep the one-word read
  template <bool BIG = false>
  __device__ __forceinline__ void jac_dof(const REAL* point, int body, int dof, REAL* jp, REAL* jr) const {
    const REAL on = BIG ? (REAL)((M.body_dofmask[body * M.mask_words + (dof >> 6)] >> (dof & 63)) & 1ull) : (REAL)((M.body_dofmask[body] >> dof) & 1ull);
    const REAL* rc = S.subtree_com() + 3 * M.body_rootid[body];
    const REAL off[3] = {point[0] - rc[0], point[1] - rc[1], point[2] - rc[2]};
    const REAL* cd = S.cdof() + 6 * dof;
    REAL c[3];
    cross3(cd, off, c);
#pragma unroll
    for (int i = 0; i < 3; i++) { jp[i] = (cd[3 + i] + c[i]) * on; jr[i] = cd[i] * on; }
  }

  // the same with the body's root and its dof-mask bit already looked up (DevModel::con_body / con_dmask)
  __device__ __forceinline__ void jac_dof_root(const REAL* point, int root, REAL on, int dof, REAL* jp, REAL* jr) const {
    const REAL* rc = S.subtree_com() + 3 * root;
    const REAL off[3] = {point[0] - rc[0], point[1] - rc[1], point[2] - rc[2]};
    const REAL* cd = S.cdof() + 6 * dof;
    REAL c[3];
    cross3(cd, off, c);
#pragma unroll
    for (int i = 0; i < 3; i++) { jp[i] = (cd[3 + i] + c[i]) * on; jr[i] = cd[i] * on; }
  }

  __device__ __forceinline__ void kbi(const REAL* solref, const REAL* solimp, REAL pos, REAL& k, REAL& b, REAL& imp) const {  // :69-113
    REAL timeconst = solref[0], dampratio = solref[1];
    if (!(M.disableflags & DSBL_REFSAFE)) {
      const REAL t2 = 2 * M.timestep;
      timeconst = (timeconst > t2 ? timeconst : t2) * (REAL)(timeconst > 0);
    }
    REAL dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    auto clampf = [](REAL x, REAL lo, REAL hi) { return x < lo ? lo : (x > hi ? hi : x); };
    dmin = clampf(dmin, (REAL)mjMINIMP, (REAL)mjMAXIMP);
    dmax = clampf(dmax, (REAL)mjMINIMP, (REAL)mjMAXIMP);
    width = width > (REAL)MINVAL_CACHED ? width : (REAL)MINVAL_CACHED;
    mid = clampf(mid, (REAL)mjMINIMP, (REAL)mjMAXIMP);
    power = power > 1 ? power : (REAL)1;
    REAL kk = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    REAL bb = 2 / (dmax * timeconst);
    if (dampratio <= 0) kk = -dampratio / (dmax * dmax);
    if (timeconst <= 0) bb = -timeconst / dmax;
    const REAL imp_x = r_abs(pos) / width;
    // solimp's default power is 2: x^1 and x^2 are exact products, no need for the general pow (four calls per row otherwise)
    const bool sq = (power == (REAL)2);
    const REAL om = 1 - mid, ox = 1 - imp_x;
    const REAL pm = sq ? mid : r_pow<REAL>(mid, power - 1), pom = sq ? om : r_pow<REAL>(om, power - 1);
    const REAL px = sq ? imp_x * imp_x : r_pow<REAL>(imp_x, power), pox = sq ? ox * ox : r_pow<REAL>(ox, power);
    const REAL imp_a = (1 / pm) * px;
    const REAL imp_b = 1 - (1 / pom) * pox;
    const REAL imp_y = imp_x < mid ? imp_a : imp_b;
    REAL im = dmin + imp_y * (dmax - dmin);
    im = clampf(im, dmin, dmax);
    if (imp_x > 1) im = dmax;
    k = kk; b = bb; imp = im;
  }

  __device__ __forceinline__ void make_constraint() {
    const int l = lane_here();
    // FRIC = false: the plain instantiation (slide / hinge limits and contacts only); equality, frictionloss, ball- and tendon-limit
    // rows compile away and cost the headline kernel no registers
    const int nv = M.nv, nefc = M.nefc, nl = M.nl, nf = FRIC ? M.nf : 0, nft = FRIC ? M.nft : 0, nfa = nf + nft, ne = FRIC ? M.ne : 0, nlb = FRIC ? M.nlb : 0, nlt = FRIC ? M.nlt : 0;
    if (nefc == 0) return;
    // plain instantiation: the slide / hinge limit rows have one non-zero each -- they go straight to global memory and the LDS copy
    // of efc_J holds the contact rows only (4.5 KB less for the humanoid); qpos is read from global by the few lanes that need it
    const int jrow0 = FRIC ? 0 : nl;                       // first efc row held in S.efc_J()
    const REAL* gq = KA.cur.qpos + e * M.nq;               // the normalised qpos of this pass
    if (FRIC) { for (int i = l; i < M.nq; i += W) S.qpos_con()[i] = gq[i]; }
    {
      const bool from_in = !KA.state_from_cur;
      if (!con_inputs_loaded_) {
        REAL* const dst[3] = {S.qvel(), S.subtree_com(), S.cdof()};
        const REAL* const src[3] = {from_in ? in.qvel : KA.cur.qvel, out.subtree_com, out.cdof};
        const int cnt[3] = {nv, 3 * M.nbody, 6 * nv};
        multi_load<W, 3, 3>(dst, src, cnt, e);
      }
      if (from_in && KA.do_step) for (int i = l; i < nv; i += W) S.qvel()[i] = checked(S.qvel()[i], (REAL)0);  // _check_state (same lane wrote it)
    }
    if (FRIC) for (int w = l; w < (ne + nfa + nlb + nl + nlt) * nv; w += W) S.efc_J()[w] = 0;
    wave_sync();
    STAMP(23);
    // equality rows (constraint.py:116-212, 254-296): one lane per (constraint, dof) column of a connect / weld, one lane per
    // joint coupling.  Body frames come from global memory (this pass's kinematics output): few values, read once per lane.
    for (int w = l; w < (FRIC ? M.neqtab : 0) * nv; w += W) {
      int q, d;
      split_index(w, nv, M.inv_nv, q, d);
      const int kind = M.eq_kind[q], id = M.eq_id[q], id1 = M.eq_obj1[q], id2 = M.eq_obj2[q], row = M.eq_row[q];
      const REAL* data = M.eq_data + 11 * id;
      const REAL active = (REAL)in.eq_active[e * M.neq + id];
      if (kind == 2) {  // _instantiate_equality_joint :254-296
        if (d != 0) continue;
        const int* ja = M.eq_jadr + 4 * q;  // dofadr1, dofadr2, qposadr1, qposadr2
        const REAL has2 = (REAL)(id2 > -1);
        const REAL pos1 = S.qpos_con()[ja[2]], pos2 = S.qpos_con()[ja[3]] * has2;
        const REAL ref1 = M.qpos0[ja[2]], ref2 = M.qpos0[ja[3]] * has2;
        const REAL dif = pos2 - ref2;
        REAL pw[5];
#pragma unroll
        for (int i = 0; i < 5; i++) pw[i] = r_pow<REAL>(dif, (REAL)i);
        REAL deriv = 0, poly = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) deriv += data[1 + i] * pw[i] * (REAL)(i + 1);
#pragma unroll
        for (int i = 0; i < 5; i++) poly += data[i] * pw[i];
        S.efc_J()[row * nv + ja[0]] = 1 * active;
        S.efc_J()[row * nv + ja[1]] = -deriv * active;  // second scatter wins on a shared dof; without a second joint it lands on the last joint's dof, as in the reference
        const REAL pos = (pos1 - ref1 - poly) * active;
        S.efc_pos()[row] = pos;
        S.efc_pos_norm()[row] = pos;
        S.efc_invweight()[row] = M.dof_invweight0[ja[0]] + M.dof_invweight0[ja[1]] * has2;
        continue;
      }
      const int nb = M.nbody;
      const REAL *xm1 = out.xmat + (e * nb + id1) * 9, *xm2 = out.xmat + (e * nb + id2) * 9;
      const REAL *xp1 = out.xpos + (e * nb + id1) * 3, *xp2 = out.xpos + (e * nb + id2) * 3;
      // connect: data[0:3] rides on body1 and data[3:6] on body2; weld: the point on body1 is data[3:6], on body2 data[0:3]
      const REAL* a1 = kind == 0 ? data : data + 3;
      const REAL* a2 = kind == 0 ? data + 3 : data;
      REAL pos1[3], pos2[3], cpos[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        pos1[i] = ((xm1[3 * i] * a1[0] + xm1[3 * i + 1] * a1[1]) + xm1[3 * i + 2] * a1[2]) + xp1[i];
        pos2[i] = ((xm2[3 * i] * a2[0] + xm2[3 * i + 1] * a2[1]) + xm2[3 * i + 2] * a2[2]) + xp2[i];
        cpos[i] = pos1[i] - pos2[i];
      }
      REAL jp1[3], jr1[3], jp2[3], jr2[3];
      jac_dof<FRIC>(pos1, id1, d, jp1, jr1);
      jac_dof<FRIC>(pos2, id2, d, jp2, jr2);
#pragma unroll
      for (int i = 0; i < 3; i++) S.efc_J()[(row + i) * nv + d] = (jp1[i] - jp2[i]) * active;
      const REAL iwt = M.body_invweight0[id1] + M.body_invweight0[id2];
      if (kind == 0) {  // _instantiate_equality_connect :116-157
        if (d == 0) {
          const REAL nrm = norm_n<REAL, 3>(cpos);
#pragma unroll
          for (int i = 0; i < 3; i++) { S.efc_pos()[row + i] = cpos[i] * active; S.efc_pos_norm()[row + i] = nrm * active; S.efc_invweight()[row + i] = iwt; }
        }
        continue;
      }
      // _instantiate_equality_weld :160-212
      const REAL torquescale = data[10];
      const REAL *xq1 = out.xquat + (e * nb + id1) * 4, *xq2 = out.xquat + (e * nb + id2) * 4;
      const REAL q1[4] = {xq1[0], xq1[1], xq1[2], xq1[3]}, rel[4] = {data[6], data[7], data[8], data[9]};
      REAL quat[4], qd[4];
      quat_mul(q1, rel, quat);
      const REAL quat1[4] = {xq2[0], xq2[1] * (REAL)-1, xq2[2] * (REAL)-1, xq2[3] * (REAL)-1};
      const REAL ax[3] = {(jr1[0] - jr2[0]) * torquescale, (jr1[1] - jr2[1]) * torquescale, (jr1[2] - jr2[2]) * torquescale};
      const REAL t[4] = {-quat1[1] * ax[0] - quat1[2] * ax[1] - quat1[3] * ax[2], quat1[0] * ax[0] + quat1[2] * ax[2] - quat1[3] * ax[1],
                         quat1[0] * ax[1] + quat1[3] * ax[0] - quat1[1] * ax[2], quat1[0] * ax[2] + quat1[1] * ax[1] - quat1[2] * ax[0]};
      REAL o[4];
      quat_mul(t, quat, o);
#pragma unroll
      for (int i = 0; i < 3; i++) S.efc_J()[(row + 3 + i) * nv + d] = ((REAL)0.5 * o[1 + i]) * active;
      if (d == 0) {
        quat_mul(quat1, quat, qd);
        const REAL pos6[6] = {cpos[0], cpos[1], cpos[2], qd[1] * torquescale, qd[2] * torquescale, qd[3] * torquescale};
        const REAL nrm = norm_n<REAL, 6>(pos6);
        const REAL iwr = M.body_invweight0_rot[id1] + M.body_invweight0_rot[id2];
#pragma unroll
        for (int i = 0; i < 6; i++) { S.efc_pos()[row + i] = pos6[i] * active; S.efc_pos_norm()[row + i] = nrm * active; S.efc_invweight()[row + i] = i < 3 ? iwt : iwr; }
      }
    }
    for (int r0 = l; r0 < nf; r0 += W) {  // _instantiate_friction :215-251 (dof rows)
      const int r = ne + r0;
      const int da = M.fric_dof[r0];
      S.efc_J()[r * nv + da] = 1;
      S.efc_pos()[r] = 0;
      S.efc_pos_norm()[r] = 0;
      S.efc_invweight()[r] = M.dof_invweight0[da];
    }
    for (int r0 = l; r0 < nft; r0 += W) {  // _instantiate_friction :215-251 (tendon rows: J = ten_J[t], a model constant for fixed tendons)
      const int r = ne + nf + r0;
      const int t = M.fric_tendon[r0];
      for (int d = 0; d < nv; d++) S.efc_J()[r * nv + d] = M.ten_J0[t * nv + d];
      S.efc_pos()[r] = 0;
      S.efc_pos_norm()[r] = 0;
      S.efc_invweight()[r] = M.tendon_invweight0[t];
    }
    for (int r0 = l; r0 < nlb; r0 += W) {  // _instantiate_limit_ball :299-335
      const int r = ne + nfa + r0;
      const int j = M.lim_ball_jnt[r0], qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
      const REAL q[4] = {S.qpos_con()[qa], S.qpos_con()[qa + 1], S.qpos_con()[qa + 2], S.qpos_con()[qa + 3]};
      REAL axis[3], angle;
      quat_to_axis_angle(q, axis, angle);
      const REAL r0_ = M.jnt_range[2 * j], r1_ = M.jnt_range[2 * j + 1];
      const REAL pos = ((r0_ > r1_ ? r0_ : r1_) - angle) - M.jnt_margin[j];
      const REAL active = (REAL)(pos < 0);
#pragma unroll
      for (int k = 0; k < 3; k++) S.efc_J()[r * nv + da + k] = (-axis[k]) * active;
      S.efc_pos()[r] = pos * active;
      S.efc_pos_norm()[r] = pos * active;
      S.efc_invweight()[r] = M.dof_invweight0[da];
    }
    for (int r0 = l; r0 < nlt; r0 += W) {  // _instantiate_limit_tendon :375-405
      const int r = ne + nfa + nlb + nl + r0;
      const int t = M.lim_tendon[r0];
      REAL len = 0;
      for (int q = M.ten_adr[t]; q < M.ten_adr[t + 1]; q++) len += M.ten_coef[q] * S.qpos_con()[M.ten_qposadr[q]];
      const REAL dist_min = len - M.tendon_range[2 * t], dist_max = M.tendon_range[2 * t + 1] - len;
      const REAL pos = (dist_min < dist_max ? dist_min : dist_max) - M.tendon_margin[t];
      const REAL active = (REAL)(pos < 0);
      const REAL sign = ((REAL)(dist_min < dist_max) * 2 - 1) * active;
      for (int d = 0; d < nv; d++) S.efc_J()[r * nv + d] = M.ten_J0[t * nv + d] * sign;
      S.efc_pos()[r] = pos * active;
      S.efc_pos_norm()[r] = pos * active;
      S.efc_invweight()[r] = M.tendon_invweight0[t];
    }
    for (int r0 = l; r0 < nl; r0 += W) {  // _instantiate_limit_slide_hinge :338-372
      const int r = ne + nfa + nlb + r0;
      const int j = M.lim_jnt[r0], qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
      const REAL q = FRIC ? S.qpos_con()[qa] : gq[qa];
      const REAL dist_min = q - M.jnt_range[2 * j], dist_max = M.jnt_range[2 * j + 1] - q;
      const REAL val = (REAL)(dist_min < dist_max) * 2 - 1;
      const REAL pos = (dist_min < dist_max ? dist_min : dist_max) - M.jnt_margin[j];
      const REAL active = (REAL)(pos < 0);
      if (FRIC) S.efc_J()[r * nv + da] = val * active;
      else S.efc_jl()[r0] = val * active;
      S.efc_pos()[r] = pos * active;
      if (FRIC) S.efc_pos_norm()[r] = pos * active;
      S.efc_invweight()[r] = M.dof_invweight0[da];
    }
    const bool elliptic = M.cone == CONE_ELLIPTIC;
    int nact_contacts = 0;  // small models: active contacts of this environment (compact list in S.i_con_act())
    REAL* hs = nullptr;     // ... and their hand-over to the register solver (KArgs::hs), when this launch has one
    REAL* hsJ = nullptr;
    if constexpr (!FRIC && DIRECT) {
      // Small models (rows straight to the leaf).  (A) one lane per contact: which contacts are active, as a compact list and as a per-row flag; (B) the rows of the
      // inactive contacts are zeroed by a straight loop over the block (no table reads); (C) one lane per (ACTIVE contact, dof) forms the
      // Jacobian entries, everything it needs about the contact coming from two table reads indexed by the contact.  Walking all
      // (contact, dof) pairs cost every lane a chain of dependent table reads per pair -- 15 trips per lane for the ant, which keeps 4 - 8 of
      // its 60 contacts active.  (The float64 instantiation that keeps the rows in LDS -- the humanoid's -- is slower with this loop
      // structure, 43 -> 56 us: its register allocation sits at the 128-VGPR bound; it keeps the (contact, dof) walk below.)
      int* const act_list = reinterpret_cast<int*>(S.i_con_act());
      int* const row_act = reinterpret_cast<int*>(S.i_crow_act());
      const int ncon = M.ncon, nd = nefc - nl;
      int nact = 0;
      for (int base = 0; base < ncon; base += W) {
        const int c = base + l;
        bool act = false;
        int rows = 0, row0 = 0;
        if (c < ncon) {
          const int dim = M.con_dim[c];
          rows = dim == 1 ? 1 : (elliptic ? dim : 2 * (dim - 1));
          row0 = M.con_efc_address[c] - nl;
          act = (S.con_dist()[c] - M.con_includemargin[c]) < 0;
        }
        int tot;
        const int before = sub_prefix_count<W>(act, tot);
        if (act) act_list[nact + before] = c;
        if (M.crow_by_con) { if (c < ncon) row_act[c] = act ? nact + before + 1 : 0; }  // (dense row q is contact q / con_rows: one entry per contact -- 0 = inactive, else its place in the compact list + 1)
        else for (int r = 0; r < rows; r++) row_act[row0 + r] = act ? 1 : 0;
        nact += tot;
      }
      nact_contacts = nact;
      wave_sync();
      // hand-over to the register solver (KArgs::hs): the count, then contact -> compact slot
      if (KA.hs && M.crow_by_con) hs = KA.hs + e * KA.hs_reals;
      if (hs) {
        if (l == 0) hs[0] = (REAL)(nact * M.con_rows);
        for (int c = l; c < ncon; c += W) hs[1 + c] = (REAL)crow_slot_of_contact(row_act, c);
      }
      if (hs) hsJ = hs + 1 + ncon + 2 * nd;
      REAL* const Jdst = out.efc_J + (e * nefc + nl) * nv;  // row 0 = first contact row
      // RK4 stages 1..3 write a private workspace Data whose only reader is this stage's solver phase, and that gathers the rows of the ACTIVE
      // contacts only (load_solver_inputs / run_sol2): the zero rows, and further down D / aref of the inactive rows, are not written there
      const bool scratch_stage = KA.rk_stage > 0;
      const float inv_rows_ = 1.0f / (float)(M.con_rows > 0 ? M.con_rows : 1);
      if (!scratch_stage) {
        constexpr int VW = 16 / (int)sizeof(REAL);  // elements per 16-byte store
        typedef REAL zvec __attribute__((ext_vector_type(VW)));
        if (nv % VW == 0 && (reinterpret_cast<uintptr_t>(Jdst) & 15) == 0) {
          // rows are whole 16-byte groups: a quarter (float) / half (double) of the passes of the element-wise loop below (the ant: 45 -> 12)
          const int gpr = nv / VW;  // groups per row
          zvec z;
#pragma unroll
          for (int i = 0; i < VW; i++) z[i] = 0;
          for (int w = l; w < nd * gpr; w += W) {
            int q, g;
            split_index(w, gpr, 1.0f / (float)gpr, q, g);
            if (!crow_entry(row_act, q, inv_rows_)) MJH_NT_STORE(z, &reinterpret_cast<zvec*>(Jdst)[w]);  // inactive contact: every entry is (something) * 0 in the reference -- the Jacobians are not formed
          }
        } else
        for (int w = l; w < nd * nv; w += W) {
          int q, d;
          split_index(w, nv, M.inv_nv, q, d);
          if (!crow_entry(row_act, q, inv_rows_)) Jdst[w] = 0;
        }
      }
      for (int w = l; w < nact * nv; w += W) {
        int a, d;
        split_index(w, nv, M.inv_nv, a, d);
        const int c = act_list[a];
        const int dim = M.con_dim[c], row0 = M.con_efc_address[c] - nl;
        const int* cb = M.con_body + 4 * c;
        const int root1 = cb[2], root2 = cb[3];
        const REAL on1 = (REAL)((M.con_dmask[2 * c] >> d) & 1ull), on2 = (REAL)((M.con_dmask[2 * c + 1] >> d) & 1ull);
        const REAL* fr = S.con_frame() + 9 * c;
        const REAL* cpos = S.con_pos() + 3 * c;
        const REAL* fric = M.con_friction + 5 * c;
        REAL jp1[3], jr1[3], jp2[3], jr2[3];
        jac_dof_root(cpos, root2, on2, d, jp2, jr2);
        jac_dof_root(cpos, root1, on1, d, jp1, jr1);
        const REAL dp[3] = {jp2[0] - jp1[0], jp2[1] - jp1[1], jp2[2] - jp1[2]};
        const REAL dr[3] = {jr2[0] - jr1[0], jr2[1] - jr1[1], jr2[2] - jr1[2]};
        REAL diff[6];
#pragma unroll
        for (int r = 0; r < 3; r++) {
          diff[r] = fr[3 * r] * dp[0] + fr[3 * r + 1] * dp[1] + fr[3 * r + 2] * dp[2];
          diff[3 + r] = fr[3 * r] * dr[0] + fr[3 * r + 1] * dr[1] + fr[3 * r + 2] * dr[2];
        }
        // RK4 stages 1..3 with a hand-over: the solver reads the compact copy only -- the workspace leaf is not written
        REAL* const dstA = (hsJ && scratch_stage && !MJH_SOL2_CAPS_ON) ? nullptr : Jdst + row0 * nv + d;  // (a build with the iteration caps hands capped solves to the LDS solver, which reads the leaves)
        REAL* const dstB = hsJ ? hsJ + (a * M.con_rows) * nv + d : nullptr;
        if (dim == 1) {
          if (dstA) dstA[0] = diff[0];
          if (dstB) dstB[0] = diff[0];
        } else if (!elliptic) {  // _instantiate_contact_pyramidal :454-516
          const int nedge = 2 * (dim - 1);
          for (int ed = 0; ed < nedge; ed++) {
            const REAL f = fric[ed >> 1] * ((ed & 1) ? (REAL)-1 : (REAL)1);
            const REAL v = diff[0] + diff[1 + (ed >> 1)] * f;
            if (dstA) { if (dstB) MJH_NT_STORE(v, &dstA[ed * nv]); else dstA[ed * nv] = v; }  // (with a hand-over the solver reads THAT copy: the leaf is output only)
            if (dstB) dstB[ed * nv] = v;
          }
        } else {  // _instantiate_contact_elliptic :519-583
          for (int r = 0; r < dim; r++) { if (dstA) { if (dstB) MJH_NT_STORE(diff[r], &dstA[r * nv]); else dstA[r * nv] = diff[r]; } if (dstB) dstB[r * nv] = diff[r]; }
        }
      }
    } else
    // contact rows: one lane per (contact, dof) column entry; all rows of the contact for that dof
    for (int w = l; w < M.ncon * nv; w += W) {
      int c, d;
      split_index(w, nv, M.inv_nv, c, d);
      const int cq = con_src(c);  // candidate behind the slot (the per-contact tables and the narrow-phase results are indexed by it)
      // small models (con_direct): the rows go straight to the efc_J leaf (lanes of one contact write runs of nv elements) and the
      // aref loop below reads them back through L2 -- without the dense copy this phase's arena fits two environments per wavefront
      constexpr bool direct = DIRECT;
      REAL* const Jdst = direct ? out.efc_J + e * nefc * nv : S.efc_J();
      const int dim = M.con_dim[cq], row0 = M.con_efc_address[c] - (direct ? 0 : jrow0);
      const int b1 = M.geom_bodyid[M.con_geom1[cq]], b2 = M.geom_bodyid[M.con_geom2[cq]];
      const REAL* fr = S.con_frame() + 9 * cq;
      const REAL* cpos = S.con_pos() + 3 * cq;
      const REAL* fric = M.con_friction + 5 * cq;
      const REAL dist = S.con_dist()[cq] - M.con_includemargin[cq];
      if (!(dist < 0)) {  // inactive contact: every entry is (something) * 0 in the reference -- the Jacobians are not formed
        const int rows = dim == 1 ? 1 : (elliptic ? dim : 2 * (dim - 1));
        for (int r = 0; r < rows; r++) Jdst[(row0 + r) * nv + d] = 0;
        continue;
      }
      REAL jp1[3], jr1[3], jp2[3], jr2[3];
      jac_dof<FRIC>(cpos, b2, d, jp2, jr2);
      jac_dof<FRIC>(cpos, b1, d, jp1, jr1);
      const REAL dp[3] = {jp2[0] - jp1[0], jp2[1] - jp1[1], jp2[2] - jp1[2]};
      const REAL dr[3] = {jr2[0] - jr1[0], jr2[1] - jr1[1], jr2[2] - jr1[2]};
      REAL diff[6];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        diff[r] = fr[3 * r] * dp[0] + fr[3 * r + 1] * dp[1] + fr[3 * r + 2] * dp[2];
        diff[3 + r] = fr[3 * r] * dr[0] + fr[3 * r + 1] * dr[1] + fr[3 * r + 2] * dr[2];
      }
      if (dim == 1) {
        Jdst[row0 * nv + d] = diff[0];
      } else if (!elliptic) {  // _instantiate_contact_pyramidal :454-516
        const int nedge = 2 * (dim - 1);
        for (int ed = 0; ed < nedge; ed++) {
          const REAL f = fric[ed >> 1] * ((ed & 1) ? (REAL)-1 : (REAL)1);
          Jdst[(row0 + ed) * nv + d] = diff[0] + diff[1 + (ed >> 1)] * f;
        }
      } else {  // _instantiate_contact_elliptic :519-583
        for (int r = 0; r < dim; r++) Jdst[(row0 + r) * nv + d] = diff[r];
      }
    }
    if constexpr (!FRIC && DIRECT) wave_sync_global();  // the aref loop below reads the rows back through L2: the stores must have landed
    else wave_sync();
    STAMP(25);
    const int ns = ne + nfa + nlb + nl + nlt;  // efc_pos / efc_pos_norm / efc_invweight only hold the equality / frictionloss / limit rows
    // RK4 stages 1..3 of a small model (their Data is a private workspace, see above): only the limit rows and the rows of the ACTIVE contacts are
    // visited -- one pass of the lanes for the ant (8 + 3 x (4 .. 8) rows) where the walk over all 188 rows took six dependent passes
    const bool compact_rows = !FRIC && DIRECT && KA.rk_stage > 0 && M.con_rows > 0;
    const int nvisit = compact_rows ? ns + nact_contacts * M.con_rows : nefc;
    // Small models, several passes of the lanes over the rows (the ant: 188 rows, six passes): aref and D are staged in LDS over the contact positions / frames (dead by
    // now) and stored in one go behind the loop.  Stored inside it, every pass's table reads and row read-backs waited for the previous pass's stores to land (vmcnt is in order):
    // 20 of the ant's 83 us per launch.
    REAL* const ad_stage = S.con_pos();
    const bool stage_ad = !FRIC && DIRECT && !compact_rows && nvisit > W && 2 * nefc <= (int)((S.con_frame() + 9 * M.ncand) - S.con_pos()) && out.efc_aref && out.efc_D;
    if (stage_ad) wave_sync();  // (the row loops above read the frames)
    for (int idx = l; idx < nvisit; idx += W) {  // :683-693
      int r = idx;
      if (compact_rows && idx >= ns) {
        int a, k;
        split_index(idx - ns, M.con_rows, 1.0f / (float)M.con_rows, a, k);
        r = M.con_efc_address[reinterpret_cast<const int*>(S.i_con_act())[a]] + k;
      }
      REAL solref[2], solimp[5];
      REAL pos = 0, pos_norm = 0, invweight = 0;
      bool con_row_active = false;
      int hs_row = -1;
      if (r < ns) { pos = S.efc_pos()[r]; pos_norm = FRIC ? S.efc_pos_norm()[r] : pos; invweight = S.efc_invweight()[r]; }
      if (r < ne) {
        const int id = M.eq_id[M.efc_row_eq[r]];
        solref[0] = M.eq_solref[2 * id]; solref[1] = M.eq_solref[2 * id + 1];
        for (int i = 0; i < 5; i++) solimp[i] = M.eq_solimp[5 * id + i];
      } else if (r < ne + nf) {
        const int da = M.fric_dof[r - ne];
        solref[0] = M.dof_solref[2 * da]; solref[1] = M.dof_solref[2 * da + 1];
        for (int i = 0; i < 5; i++) solimp[i] = M.dof_solimp[5 * da + i];
      } else if (r < ne + nfa) {
        const int t = M.fric_tendon[r - ne - nf];
        solref[0] = M.tendon_solref_fri[2 * t]; solref[1] = M.tendon_solref_fri[2 * t + 1];
        for (int i = 0; i < 5; i++) solimp[i] = M.tendon_solimp_fri[5 * t + i];
      } else if (r < ne + nfa + nlb + nl) {
        const int j = r < ne + nfa + nlb ? M.lim_ball_jnt[r - ne - nfa] : M.lim_jnt[r - ne - nfa - nlb];
        solref[0] = M.jnt_solref[2 * j]; solref[1] = M.jnt_solref[2 * j + 1];
        for (int i = 0; i < 5; i++) solimp[i] = M.jnt_solimp[5 * j + i];
      } else if (r < ns) {
        const int t = M.lim_tendon[r - ne - nfa - nlb - nl];
        solref[0] = M.tendon_solref_lim[2 * t]; solref[1] = M.tendon_solref_lim[2 * t + 1];
        for (int i = 0; i < 5; i++) solimp[i] = M.tendon_solimp_lim[5 * t + i];
      } else if (!FRIC && DIRECT) {  // contact row, small models: every static scalar of the row from ONE table column read indexed by the row
        const int q = r - ns, ncr = M.ncrow;
        const int info = M.crow_info[q];
        const REAL* P = M.crow_par + q;
        solref[0] = P[0]; solref[1] = P[ncr];
#pragma unroll
        for (int i = 0; i < 5; i++) solimp[i] = P[(2 + i) * ncr];
        invweight = P[7 * ncr];
        const int c = info & 0xffff, sub = (info >> 16) & 0xff;
        const REAL dist = S.con_dist()[c] - P[8 * ncr];
        const REAL active = (REAL)(dist < 0);
        con_row_active = dist < 0;
        if (KA.rk_stage > 0 && !con_row_active) continue;  // workspace Data of an RK4 stage: nobody reads D / aref of an inactive row (see above)
        if (hs && con_row_active) hs_row = crow_slot_of_contact(reinterpret_cast<const int*>(S.i_crow_act()), c) * M.con_rows + sub;  // this row's place in the hand-over
        if (!(info >> 24)) { pos = dist * active; pos_norm = dist * active; }
        else { pos = (sub == 0 ? dist : (REAL)0) * active; pos_norm = dist; }
      } else {  // contact row: its scalars are functions of the contact (constraint.py:440-451, 480-487, 547-561), recomputed here
        const int cs = M.efc_row_con[r], sub = r - M.con_efc_address[cs];
        const int c = con_src(cs);
        const REAL* sr = M.con_solref + 2 * c;
        solref[0] = sr[0]; solref[1] = sr[1];
        const int dim = M.con_dim[c];
        if (elliptic && dim > 1 && sub > 0) {
          const REAL* sf = M.con_solreffriction + 2 * c;
          const REAL none = (REAL)(!((sf[0] != 0) || (sf[1] != 0)));
          solref[0] = sf[0] + sr[0] * none; solref[1] = sf[1] + sr[1] * none;
        }
        for (int i = 0; i < 5; i++) solimp[i] = M.con_solimp[5 * c + i];
        const REAL* fric = M.con_friction + 5 * c;
        const REAL dist = S.con_dist()[c] - M.con_includemargin[c];
        const REAL active = (REAL)(dist < 0);
        con_row_active = dist < 0;
        const REAL t = M.body_invweight0[M.geom_bodyid[M.con_geom1[c]]] + M.body_invweight0[M.geom_bodyid[M.con_geom2[c]]];
        if (dim == 1) {
          pos = dist * active; pos_norm = dist * active; invweight = t;
        } else if (!elliptic) {
          const REAL mu = fric[0];
          pos = dist * active; pos_norm = dist * active; invweight = (t + mu * mu * t) * 2 * mu * mu / M.impratio;
        } else {
          const REAL iwf = t / M.impratio;
          pos = (sub == 0 ? dist : (REAL)0) * active;
          pos_norm = dist;
          invweight = (sub == 0) ? t : (sub == 1 ? iwf : iwf * ((fric[0] * fric[0]) / (fric[sub - 1] * fric[sub - 1])));
        }
      }
      REAL k, b, imp;
      kbi(solref, solimp, pos_norm, k, b, imp);
      REAL rr = invweight * (1 - imp) / imp;
      rr = rr > (REAL)MINVAL_CACHED ? rr : (REAL)MINVAL_CACHED;
      REAL jv;
      if (FRIC) jv = dot_seq(S.efc_J() + r * nv, 1, S.qvel(), 1, nv);
      else if (r < nl) jv = 0 + S.efc_jl()[r] * S.qvel()[M.jnt_dofadr[M.lim_jnt[r]]];
      else if (!DIRECT) jv = dot_seq(S.efc_J() + (r - jrow0) * nv, 1, S.qvel(), 1, nv);
      else if (!con_row_active) jv = 0;  // the row is all zeros
      else {
        // this wave stored the row above and the barrier drained the stores to L2.  Agent-scope loads read L2 past the CU's L1,
        // where a line shared with a neighbouring environment's rows could have been cached before this wave's stores landed.
        const REAL* jr = (hs_row >= 0 && !MJH_SOL2_CAPS_ON) ? hsJ + hs_row * nv : out.efc_J + (e * nefc + r) * nv;  // (RK4 stages 1..3 with a hand-over keep the row there only)
        REAL s = 0;
        int k = 0;
        for (; k + 4 <= nv; k += 4) {
          const REAL a0 = __hip_atomic_load(jr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(jr + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const REAL a2 = __hip_atomic_load(jr + k + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3 = __hip_atomic_load(jr + k + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s += a0 * S.qvel()[k]; s += a1 * S.qvel()[k + 1]; s += a2 * S.qvel()[k + 2]; s += a3 * S.qvel()[k + 3];
        }
        for (; k < nv; k++) s += __hip_atomic_load(jr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * S.qvel()[k];
        jv = s;
      }
      const REAL aref_r = -b * jv - k * imp * pos, D_r = 1 / rr;
      if (hs_row >= 0) { hs[1 + M.ncon + hs_row] = D_r; hs[1 + M.ncon + (nefc - nl) + hs_row] = aref_r; }
      if (hs_row >= 0 && KA.rk_stage > 0 && !MJH_SOL2_CAPS_ON) continue;  // (stages 1..3: the solver reads the hand-over, nobody the workspace leaves of a contact row)
      if (stage_ad) { ad_stage[r] = aref_r; ad_stage[nefc + r] = D_r; }
      else {
        if (out.efc_aref) out.efc_aref[e * nefc + r] = aref_r;  // lane r <-> row r: coalesced, no staging
        if (out.efc_D) out.efc_D[e * nefc + r] = D_r;
      }
    }
    if (stage_ad) {
      wave_sync();
      put(out.efc_aref, ad_stage, nefc); put(out.efc_D, ad_stage + nefc, nefc);
    }
    STAMP(26);
    if (FRIC) put(out.efc_J, S.efc_J(), nefc * nv);
    else if (out.efc_J) {
      rebind();
      REAL* gJ = out.efc_J + e * nefc * nv;
      if (nl <= W) {
        // single-column rows: zeros and the one entry, written once, coalesced.  Lane r holds the column of row r; an element looks its row's column up
        // across the lanes (LDS crossbar) -- a table read per element queued behind the stores already in flight (vmcnt is in order) and made every
        // pass of this loop wait for the previous pass's stores to land
        const int mycol = l < nl ? M.lim_dof[M.nf + l] : -1;
        for (int w = l; w - l < nl * nv; w += W) {
          int r, d;
          split_index(w < nl * nv ? w : 0, nv, M.inv_nv, r, d);
          const int col = __shfl(mycol, (int)(lane_id() & ~(W - 1)) + r, MJH_WAVE);
          if (w < nl * nv) gJ[w] = (d == col) ? S.efc_jl()[r] : (REAL)0;
        }
      } else
      for (int w = l; w < nl * nv; w += W) {
        int r, d;
        split_index(w, nv, M.inv_nv, r, d);
        gJ[w] = (d == M.jnt_dofadr[M.lim_jnt[r]]) ? S.efc_jl()[r] : (REAL)0;
      }
      if (!DIRECT) for (int w = l; w < (nefc - nl) * nv; w += W) gJ[nl * nv + w] = S.efc_J()[w];
    }
    STAMP(27);
    if (out.efc_frictionloss) for (int r = l; r < nefc; r += W) out.efc_frictionloss[e * nefc + r] = (r >= ne && r < ne + nf) ? M.dof_frictionloss[M.fric_dof[r - ne]] : ((r >= ne + nf && r < ne + nfa) ? M.tendon_frictionloss[M.fric_tendon[r - ne - nf]] : (REAL)0);
  }

  // sum over the dofs [d, d + width) of the products cdof * qvel staged in cdof_dot, in dof order (width 1 or 3)
  __device__ __forceinline__ void dof_sum(int d, int width, REAL* s) {
    const REAL* p = S.cdof_dot() + 6 * d;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      REAL a = p[k];
      if (width == 3) a = (a + p[6 + k]) + p[12 + k];
      s[k] = a;
    }
  }
  // v += the contribution of one joint (pk = (type + 1) | dofadr << 8); a free joint adds its translational sum, then its rotational one
  __device__ __forceinline__ void cvel_add_joint(int pk, REAL* v) {
    const int t = (pk & 0xff) - 1, d = pk >> 8;
    REAL s[6];
    dof_sum(d, t == JNT_HINGE || t == JNT_SLIDE ? 1 : 3, s);
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = v[k] + s[k];
    if (t == JNT_FREE) {
      dof_sum(d + 3, 3, s);
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = v[k] + s[k];
    }
  }

  // ---- transmission + _velocity (smooth.py:535-591, forward.py:87-99, smooth.com_vel :385-424, passive.py:80-200, smooth.rne :427-467) ----------
  template <bool FLUID, bool FUSED = false, bool DEFER = false>
  __device__ __forceinline__ void velocity() {
    const int l = lane_here();
    const int nv = M.nv, nb = M.nbody, nu = M.nu;
    if (!FUSED) load_qpos(false);
    load_qvel(); load_act();
    if (!FUSED) {  // (fused with the kinematics: qpos, cdof, cinert, subtree_com and xipos are still in the arena)
      row_load<W>(S.cdof(), out.cdof, 6 * nv, e);      // the two long arrays batch four loads per trip on their own; this phase has no
      row_load<W>(S.cinert(), out.cinert, 10 * nb, e);  // registers to spare for holding more of them in flight
      REAL* const dst[2] = {S.subtree_com(), S.xipos()};
      const REAL* const src[2] = {out.subtree_com, out.xipos};
      const int cnt[2] = {3 * nb, 3 * nb};
      multi_load<W, 2, 2>(dst, src, cnt, e);
    }
    wave_sync();
    STAMP(31);
    if (FLUID && M.ntendon > 0) {  // smooth.tendon :470-497 and forward._velocity :93-94 for fixed tendons: one lane per tendon
      const int nt = M.ntendon;
      for (int t = l; t < nt; t += W) {
        REAL len = 0, vel = 0;
        for (int q = M.ten_adr[t]; q < M.ten_adr[t + 1]; q++) len += M.ten_coef[q] * S.qpos()[M.ten_qposadr[q]];
        for (int d = 0; d < nv; d++) vel += M.ten_J0[t * nv + d] * S.qvel()[d];
        S.ten_len()[t] = len;
        if (out.ten_length) out.ten_length[e * nt + t] = len;
        if (out.ten_velocity) out.ten_velocity[e * nt + t] = vel;
        // tendon-level spring (slack between the two rest lengths) and damper, passive.py:119-132
        const REAL below = M.tendon_lengthspring[2 * t] - len, above = M.tendon_lengthspring[2 * t + 1] - len;
        REAL fs = below > 0 ? M.tendon_stiffness[t] * below : (REAL)0;
        fs = above < 0 ? M.tendon_stiffness[t] * above : fs;
        S.ten_frc()[t] = fs + (-M.tendon_damping[t] * vel);
      }
      row_store<W>(out.ten_J, M.ten_J0, nt * nv, e);
      wave_sync();
    }
    if (!M.act_simple) row_copy_const<W>(out.actuator_moment, M.act_moment, nu * nv, e);  // the constant part of the moment matrix (smooth.py:535-591); all of it for act_simple models: velocity_stores()
    if (M.act_simple) {  // every transmission is a slide / hinge joint: one constant non-zero per moment row
      for (int i = l; i < nu; i += W) {
        const REAL gear = M.act_gear[6 * i];
        S.act_length()[i] = S.qpos()[M.act_qposadr[i]] * gear;
        S.act_velocity()[i] = gear * S.qvel()[M.act_dofadr[i]];
      }
    } else
    for (int i = l; i < nu; i += W) {
      const REAL* gear = M.act_gear + 6 * i;
      const int jt = M.act_jnttype[i], qa = M.act_qposadr[i];
      REAL len = 0;
      if (FLUID && M.act_trntype[i] == 3) {  // tendon transmission :558-561 (its moment row is a model constant)
        len = S.ten_len()[M.act_trnid[i]] * gear[0];
      } else if (jt == JNT_SLIDE || jt == JNT_HINGE) {
        len = S.qpos()[qa] * gear[0];
      } else {  // ball / free joints (:565-583)
        const bool inparent = M.act_trntype[i] == 1;
        const REAL* qp = S.qpos() + qa + (jt == JNT_FREE ? 3 : 0);
        const REAL qn[4] = {qp[0], qp[1] * (REAL)-1, qp[2] * (REAL)-1, qp[3] * (REAL)-1};
        REAL ga[3] = {gear[0], gear[1], gear[2]};
        if (inparent) {
          rotate(gear + (jt == JNT_FREE ? 3 : 0), qn, ga);
          if (M.act_has_rot) { S.act_rot()[3 * i] = ga[0]; S.act_rot()[3 * i + 1] = ga[1]; S.act_rot()[3 * i + 2] = ga[2]; }
        }
        if (jt == JNT_BALL) {
          const REAL q[4] = {qp[0], qp[1], qp[2], qp[3]};
          REAL axis[3], angle;
          quat_to_axis_angle(q, axis, angle);
          len = ((axis[0] * angle) * ga[0] + (axis[1] * angle) * ga[1]) + (axis[2] * angle) * ga[2];
        }
      }
      REAL vel = 0;  // moment . qvel over the row's non-zeros, in dof order
      for (int q = M.act_ent_adr[i]; q < M.act_ent_adr[i + 1]; q++) {
        const int rot = M.act_ent_rot[q], dd = M.act_ent_dof[q];
        const REAL coef = rot < 0 ? M.act_ent_coef[q] : S.act_rot()[3 * i + rot];
        vel += coef * S.qvel()[dd];
        if (rot >= 0 && out.actuator_moment) out.actuator_moment[(e * nu + i) * nv + dd] = coef;  // after the row_store above: same wave, program order
      }
      S.act_length()[i] = len;
      S.act_velocity()[i] = vel;
    }
    STAMP(32);
    if constexpr (W <= 16) {  // small models (four environments per wavefront): shallow trees, where the plain walk down the ancestor chain is shorter than the sweep's passes
    // com_vel: lane b accumulates cvel along its ancestor chain, in the reference's per-body order
    for (int b = l; b < nb; b += W) {
      REAL cvel[6] = {0, 0, 0, 0, 0, 0};
      const int depth = M.body_depth[b], md = M.max_depth, mj = M.max_jnt;
      for (int kk = 0; kk < md; kk++) {  // uniform trip counts; the joint list of every level comes from one flat table
        if (kk >= depth) continue;      // (its addresses depend on (b, level) only, so the loads run ahead of the arithmetic)
        const bool own = (kk == depth - 1);
        for (int jj = 0; jj < mj; jj++) {
          const int pk = M.chain_jnt[(b * md + kk) * mj + jj];
          if (pk == 0) continue;
          const int t = (pk & 0xff) - 1, d = pk >> 8;
          if (t == JNT_FREE) {
            REAL s[6];
#pragma unroll
            for (int k = 0; k < 6; k++) s[k] = (S.cdof()[6 * d + k] * S.qvel()[d] + S.cdof()[6 * (d + 1) + k] * S.qvel()[d + 1]) + S.cdof()[6 * (d + 2) + k] * S.qvel()[d + 2];
#pragma unroll
            for (int k = 0; k < 6; k++) cvel[k] = cvel[k] + s[k];
            if (own) {
              for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) S.cdof_dot()[6 * (d + r) + k] = 0;
              for (int r = 3; r < 6; r++) motion_cross(cvel, S.cdof() + 6 * (d + r), S.cdof_dot() + 6 * (d + r));
            }
#pragma unroll
            for (int k = 0; k < 6; k++) s[k] = (S.cdof()[6 * (d + 3) + k] * S.qvel()[d + 3] + S.cdof()[6 * (d + 4) + k] * S.qvel()[d + 4]) + S.cdof()[6 * (d + 5) + k] * S.qvel()[d + 5];
#pragma unroll
            for (int k = 0; k < 6; k++) cvel[k] = cvel[k] + s[k];
          } else {
            const int width = (t == JNT_BALL) ? 3 : 1;
            if (own) for (int r = 0; r < width; r++) motion_cross(cvel, S.cdof() + 6 * (d + r), S.cdof_dot() + 6 * (d + r));
            REAL s[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
              s[k] = S.cdof()[6 * d + k] * S.qvel()[d];
              for (int r = 1; r < width; r++) s[k] = s[k] + S.cdof()[6 * (d + r) + k] * S.qvel()[d + r];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) cvel[k] = cvel[k] + s[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 6; k++) S.cvel()[6 * b + k] = cvel[k];
    }
    } else {
    // com_vel (smooth.py:385-424): cvel[b] = cvel[parent] + the sums of b's own joints, in joint order -- the reference's scan over the tree,
    // level by level.  What does not depend on the parent is taken off the serial part: the products cdof * qvel are formed by one lane per
    // entry first (staged in cdof_dot, which nobody has written yet), and cdof_dot = cvel-before-the-joint x cdof is formed for all bodies at once
    // after the sweep (a lane rebuilds its body's partial sums from the parent's cvel: the same additions in the same order).  The sweep itself
    // is six additions per joint and one LDS round trip per level (it used to walk every lane down its whole ancestor chain: depth x joints
    // dependent table reads and multiply-adds per lane).
    {
      const int md = M.max_depth, mj = M.max_jnt;
      // the descriptors of the first chunk of bodies are the sweep's only table reads: requested first, because ...
      const int depth0 = l < nb ? M.body_depth[l] : 0, par0 = l < nb ? M.body_parentid[l] : 0;
      const int* const own0 = M.chain_jnt + ((size_t)(l < nb ? l : 0) * md + (depth0 > 0 ? depth0 - 1 : 0)) * mj;
      int pkj0[MJH_CHAIN_PRE];
#pragma unroll
      for (int jj = 0; jj < MJH_CHAIN_PRE; jj++) pkj0[jj] = (depth0 > 0 && jj < mj) ? own0[jj] : 0;
      // ... the frame leaves of the kinematics stage go out HERE (fused kernel): everything from here to the end of the sweep works on LDS and registers, so the burst (21 MB
      // for the humanoid batch) drains behind ~8 us of arithmetic.  Stored at the end of the kinematics stage, the next table read -- any vector load, vmcnt is in order -- waited for all of it.
      // (The arrays the velocity stage has written so far -- qvel, act, actuator length / velocity -- overlay the kinematics stage's scratch arrays, not the frames: lds_carve.)
      if (DEFER && M.kv_defer) frame_stores();
      for (int w = l; w < 6 * nv; w += W) S.cdof_dot()[w] = S.cdof()[w] * S.qvel()[w / 6];
      if (l < 6) S.cvel()[l] = 0;  // the world body
      wave_sync();
      for (int b0 = 0; b0 < nb; b0 += W) {  // parents precede their children: a chunk of W bodies only needs earlier chunks and its own lower levels
        const int b = b0 + l;
        const bool has = b < nb;
        const int depth = b0 == 0 ? depth0 : (has ? M.body_depth[b] : 0), par = b0 == 0 ? par0 : (has ? M.body_parentid[b] : 0);
        const int* own = b0 == 0 ? own0 : M.chain_jnt + ((size_t)(has ? b : 0) * md + (depth > 0 ? depth - 1 : 0)) * mj;
        int pkj[MJH_CHAIN_PRE];
#pragma unroll
        for (int jj = 0; jj < MJH_CHAIN_PRE; jj++) pkj[jj] = b0 == 0 ? pkj0[jj] : ((depth > 0 && jj < mj) ? own[jj] : 0);
        for (int kk = 0; kk < md; kk++) {
          if (depth == kk + 1) {
            REAL v[6];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = S.cvel()[6 * par + k];
#pragma unroll
            for (int jj = 0; jj < MJH_CHAIN_PRE; jj++) if (pkj[jj]) cvel_add_joint(pkj[jj], v);
            for (int jj = MJH_CHAIN_PRE; jj < mj; jj++) { const int pk = own[jj]; if (pk) cvel_add_joint(pk, v); }
#pragma unroll
            for (int k = 0; k < 6; k++) S.cvel()[6 * b + k] = v[k];
          }
          wave_sync();
        }
        if (depth > 0) {  // cdof_dot of the body's own dofs (each lane touches its own body's slots only: the staged products are read before they are overwritten)
          REAL v[6];
#pragma unroll
          for (int k = 0; k < 6; k++) v[k] = S.cvel()[6 * par + k];
          auto own_joint = [&](int pk) {
            if (pk == 0) return;
            const int t = (pk & 0xff) - 1, d = pk >> 8;
            if (t == JNT_FREE) {
              REAL tr[6], rot[6];
              dof_sum(d, 3, tr);
              dof_sum(d + 3, 3, rot);  // read before its slots are overwritten
#pragma unroll
              for (int k = 0; k < 6; k++) v[k] = v[k] + tr[k];
              for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) S.cdof_dot()[6 * (d + r) + k] = 0;
              for (int r = 3; r < 6; r++) motion_cross(v, S.cdof() + 6 * (d + r), S.cdof_dot() + 6 * (d + r));
#pragma unroll
              for (int k = 0; k < 6; k++) v[k] = v[k] + rot[k];
            } else {
              const int width = (t == JNT_BALL) ? 3 : 1;
              REAL sj[6];
              dof_sum(d, width, sj);
              for (int r = 0; r < width; r++) motion_cross(v, S.cdof() + 6 * (d + r), S.cdof_dot() + 6 * (d + r));
#pragma unroll
              for (int k = 0; k < 6; k++) v[k] = v[k] + sj[k];
            }
          };
#pragma unroll
          for (int jj = 0; jj < MJH_CHAIN_PRE; jj++) own_joint(pkj[jj]);
          for (int jj = MJH_CHAIN_PRE; jj < mj; jj++) own_joint(own[jj]);
        }
        wave_sync();
      }
    }
    }
    STAMP(33);
    // passive forces
    if (M.disableflags & (DSBL_SPRING | DSBL_DAMPER)) {
      for (int d = l; d < nv; d += W) S.qfrc_passive()[d] = 0;
      if (FLUID && M.has_gravcomp) for (int d = l; d < nv; d += W) { S.qfrc_gravcomp()[d] = 0; if (out.qfrc_gravcomp) out.qfrc_gravcomp[e * nv + d] = 0; }  // passive.py:178-183
    } else {
      for (int j = l; j < M.njnt; j += W) {
        const int t = M.jnt_type[j], qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
        const REAL k = M.jnt_stiffness[j];
        if (t == JNT_FREE) {
          for (int i = 0; i < 3; i++) S.qfrc_passive()[da + i] = -k * (S.qpos()[qa + i] - M.qpos_spring[qa + i]);
          REAL r[3];
          quat_sub(S.qpos() + qa + 3, M.qpos_spring + qa + 3, r);
          for (int i = 0; i < 3; i++) S.qfrc_passive()[da + 3 + i] = -k * r[i];
        } else if (t == JNT_BALL) {
          REAL r[3];
          quat_sub(S.qpos() + qa, M.qpos_spring + qa, r);
          for (int i = 0; i < 3; i++) S.qfrc_passive()[da + i] = -k * r[i];
        } else {
          S.qfrc_passive()[da] = -k * (S.qpos()[qa] - M.qpos_spring[qa]);
        }
      }
      wave_sync();
      for (int d = l; d < nv; d += W) S.qfrc_passive()[d] = (0 + S.qfrc_passive()[d]) - M.dof_damping[d] * S.qvel()[d];
      if (FLUID && M.ntendon > 0) {  // qfrc += ten_J^T (spring + damper), tendons in order (passive.py:134-143)
        for (int d = l; d < nv; d += W) {
          REAL acc = 0;
          for (int t = 0; t < M.ntendon; t++) acc += M.ten_J0[t * nv + d] * S.ten_frc()[t];
          S.qfrc_passive()[d] = S.qfrc_passive()[d] + acc;
        }
      }
      if (FLUID && M.has_gravcomp && (M.disableflags & DSBL_GRAVITY)) {  // gravity off: the caller's leaf is carried (passive.py:190-194) and still feeds the actuator term
        for (int d = l; d < nv; d += W) {
          const REAL v = in.qfrc_gravcomp ? in.qfrc_gravcomp[e * nv + d] : (REAL)0;
          S.qfrc_gravcomp()[d] = v;
          if (out.qfrc_gravcomp) out.qfrc_gravcomp[e * nv + d] = v;
        }
      }
      if (FLUID && M.has_gravcomp && !(M.disableflags & DSBL_GRAVITY)) {  // passive._gravcomp :148-156: -gravity * mass * gravcomp at every body's inertial origin
        wave_sync();
        for (int d = l; d < nv; d += W) {
          REAL acc = 0;
          for (int b = 0; b < nb; b++) {
            const REAL mg = M.body_mass[b] * M.body_gravcomp[b];
            const REAL f[3] = {-M.gravity[0] * mg, -M.gravity[1] * mg, -M.gravity[2] * mg};
            REAL jp[3], jr[3];
            jac_dof<FLUID>(S.xipos() + 3 * b, b, d, jp, jr);
            acc += dot3(jp, f);
          }
          S.qfrc_gravcomp()[d] = acc;
          if (out.qfrc_gravcomp) out.qfrc_gravcomp[e * nv + d] = acc;
          S.qfrc_passive()[d] = S.qfrc_passive()[d] + acc * (REAL)(1 - M.jnt_actgravcomp[M.dof_jntid[d]]);
        }
      }
      if (FLUID && M.has_fluid) {  // (own kernel instantiation: its registers must not weigh on fluid-free models) passive._fluid :158-173 with _inertia_box_fluid_model :31-78: one lane per body, wrench staged in cfrc
        for (int b = l; b < nb; b += W) {
          const REAL pi = (REAL)3.14159265358979323846;
          const REAL* inr = M.body_inertia + 3 * b;
          const REAL mass = M.body_mass[b];
          REAL box[3];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            REAL s3 = (inr[0] * (i == 0 ? (REAL)-1 : (REAL)1) + inr[1] * (i == 1 ? (REAL)-1 : (REAL)1)) + inr[2] * (i == 2 ? (REAL)-1 : (REAL)1);
            s3 = s3 > (REAL)1e-12 ? s3 : (REAL)1e-12;
            const REAL mm = mass > (REAL)(float)1e-12 ? mass : (REAL)(float)1e-12;
            box[i] = r_sqrt<REAL>(((REAL)6.0 * s3) / mm) * (REAL)(mass > 0);
          }
          REAL xi[9];
#pragma unroll
          for (int i = 0; i < 9; i++) xi[i] = out.ximat[(e * nb + b) * 9 + i];
          const REAL* cv = S.cvel() + 6 * b;
          const REAL* rc = S.subtree_com() + 3 * M.body_rootid[b];
          const REAL off[3] = {S.xipos()[3 * b] - rc[0], S.xipos()[3 * b + 1] - rc[1], S.xipos()[3 * b + 2] - rc[2]};
          REAL c[3], v3[3], lvel[6], lwind[3];
          cross3(off, cv, c);  // math.transform_motion :437-452
#pragma unroll
          for (int i = 0; i < 3; i++) v3[i] = cv[3 + i] - c[i];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            lvel[3 + i] = xi[i] * v3[0] + xi[3 + i] * v3[1] + xi[6 + i] * v3[2];
            lvel[i] = xi[i] * cv[0] + xi[3 + i] * cv[1] + xi[6 + i] * cv[2];
            lwind[i] = xi[i] * M.wind[0] + xi[3 + i] * M.wind[1] + xi[6 + i] * M.wind[2];
          }
#pragma unroll
          for (int i = 0; i < 3; i++) lvel[3 + i] = lvel[3 + i] + (-lwind[i]);
          const REAL diam = ((box[0] + box[1]) + box[2]) / 3;
          const REAL d3 = diam * diam * diam;
          REAL fa[3], fv[3];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            fa[i] = lvel[i] * -pi * d3 * M.viscosity;
            fv[i] = lvel[3 + i] * (REAL)-3.0 * pi * diam * M.viscosity;
          }
          const REAL sv[3] = {box[1] * box[2], box[0] * box[2], box[0] * box[1]};
          const REAL b4[3] = {r_pow<REAL>(box[0], (REAL)4), r_pow<REAL>(box[1], (REAL)4), r_pow<REAL>(box[2], (REAL)4)};
          const REAL sa[3] = {box[0] * (b4[1] + b4[2]), box[1] * (b4[0] + b4[2]), box[2] * (b4[0] + b4[1])};
#pragma unroll
          for (int i = 0; i < 3; i++) {
            fv[i] = fv[i] - (REAL)0.5 * M.density * sv[i] * r_abs(lvel[3 + i]) * lvel[3 + i];
            fa[i] = fa[i] - ((REAL)1.0 * M.density * sa[i] * r_abs(lvel[i]) * lvel[i] / (REAL)64.0);
          }
#pragma unroll
          for (int i = 0; i < 3; i++) {
            S.cfrc()[6 * b + i] = xi[3 * i] * fv[0] + xi[3 * i + 1] * fv[1] + xi[3 * i + 2] * fv[2];        // force
            S.cfrc()[6 * b + 3 + i] = xi[3 * i] * fa[0] + xi[3 * i + 1] * fa[1] + xi[3 * i + 2] * fa[2];    // torque
          }
        }
        wave_sync();
        for (int d = l; d < nv; d += W) {  // support.apply_ft :169-181, summed over bodies in order
          REAL acc = 0;
          for (int b = 0; b < nb; b++) {
            REAL jp[3], jr[3];
            jac_dof<FLUID>(S.xipos() + 3 * b, b, d, jp, jr);
            acc += dot3(jp, S.cfrc() + 6 * b) + dot3(jr, S.cfrc() + 6 * b + 3);
          }
          S.qfrc_passive()[d] = S.qfrc_passive()[d] + acc;
        }
        wave_sync();
      }
    }
    wave_sync();
    STAMP(35);
    if constexpr (W <= 16) {
    // rne: cacc along the ancestor chain (needs cdof_dot of ancestors: written above, visible after the sync)
    for (int b = l; b < nb; b += W) {
      REAL cacc[6];
      const bool nograv = M.disableflags & DSBL_GRAVITY;
#pragma unroll
      for (int k = 0; k < 3; k++) { cacc[k] = 0; cacc[3 + k] = nograv ? (REAL)0 : -M.gravity[k]; }
      const int depth = M.body_depth[b], md = M.max_depth;
      for (int kk = 0; kk < md; kk++) {
        if (kk >= depth) continue;
        const int pk = M.chain_dof[b * md + kk];
        const int d0 = pk & 0xffff, nd = pk >> 16;
        if (nd > 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) {
            REAL s = S.cdof_dot()[6 * d0 + k] * S.qvel()[d0];
            for (int r = 1; r < nd; r++) s = s + S.cdof_dot()[6 * (d0 + r) + k] * S.qvel()[d0 + r];
            cacc[k] = cacc[k] + s;
          }
        }
      }
      REAL f1[6], f2[6], f3[6];
      inert_mul(S.cinert() + 10 * b, cacc, f1);
      inert_mul(S.cinert() + 10 * b, S.cvel() + 6 * b, f2);
      motion_cross_force(S.cvel() + 6 * b, f2, f3);
#pragma unroll
      for (int k = 0; k < 6; k++) S.cacc()[6 * b + k] = f1[k] + f3[k];  // local cfrc (cacc itself is not a Data output)
    }
    } else {
    // rne (smooth.py:427-467): cacc[b] = cacc[parent] + sum over b's dofs of cdof_dot * qvel -- the same level sweep: the sums are formed for all
    // bodies at once, the sweep is one addition per component and level, the local forces follow for all bodies at once
    {
      const bool nograv = M.disableflags & DSBL_GRAVITY;
      if (l < 6) S.cacc()[l] = (l < 3 || nograv) ? (REAL)0 : -M.gravity[l - 3];  // the world body
      const int md = M.max_depth;
      const int depth0 = l < nb ? M.body_depth[l] : 0, par0 = l < nb ? M.body_parentid[l] : 0;
      const int d00 = l < nb ? M.body_dofadr[l] : 0, nd0 = (l < nb && depth0 > 0) ? M.body_dofnum[l] : 0;
      if (DEFER && M.kv_defer) {  // (same reasoning as in front of the com_vel sweep: the rest of the kinematics stage's leaves and the two velocity leaves that are final by now)
        com_stores();
        put(out.cvel, S.cvel(), 6 * nb); putnt(out.cdof_dot, S.cdof_dot(), 6 * nv);
      }
      for (int b0 = 0; b0 < nb; b0 += W) {
        const int b = b0 + l;
        const bool has = b < nb;
        const int depth = b0 == 0 ? depth0 : (has ? M.body_depth[b] : 0), par = b0 == 0 ? par0 : (has ? M.body_parentid[b] : 0);
        const int d0 = b0 == 0 ? d00 : (has ? M.body_dofadr[b] : 0), nd = b0 == 0 ? nd0 : ((has && depth > 0) ? M.body_dofnum[b] : 0);
        REAL vm[6] = {0, 0, 0, 0, 0, 0};
        if (nd > 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) {
            REAL s = S.cdof_dot()[6 * d0 + k] * S.qvel()[d0];
            for (int r = 1; r < nd; r++) s = s + S.cdof_dot()[6 * (d0 + r) + k] * S.qvel()[d0 + r];
            vm[k] = s;
          }
        }
        wave_sync();  // (first chunk: the world body's row; later chunks: nothing pending)
        for (int kk = 0; kk < md; kk++) {
          if (depth == kk + 1) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
              const REAL c = S.cacc()[6 * par + k];
              S.cacc()[6 * b + k] = nd > 0 ? c + vm[k] : c;
            }
          }
          wave_sync();
        }
      }
      for (int b = l; b < nb; b += W) {
        REAL cacc[6], f1[6], f2[6], f3[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cacc[k] = S.cacc()[6 * b + k];
        inert_mul(S.cinert() + 10 * b, cacc, f1);
        inert_mul(S.cinert() + 10 * b, S.cvel() + 6 * b, f2);
        motion_cross_force(S.cvel() + 6 * b, f2, f3);
#pragma unroll
        for (int k = 0; k < 6; k++) S.cacc()[6 * b + k] = f1[k] + f3[k];  // local cfrc, over the body's own cacc (cacc itself is not a Data output)
      }
    }
    }
    wave_sync();
    STAMP(36);
    if (M.lds_diet) {  // no cfrc array: the lane of a dof forms its body's six subtree sums itself (same terms, same order) and projects them
      for (int d = l; d < nv; d += W) {
        const int b = M.dof_bodyid[d], end = M.body_subtree_end[b];
        REAL s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          REAL acc = 0;
          for (int bb = end - 1; bb >= b; bb--) acc += S.cacc()[6 * bb + k];
          s += S.cdof()[6 * d + k] * acc;
        }
        S.qfrc_bias()[d] = s;
      }
    } else {
    for (int w = l; w < nb * 6; w += W) {  // subtree sums of the body forces
      const int b = w / 6, k = w - 6 * b;
      const int end = M.body_subtree_end[b];
      REAL acc = 0;
      for (int d = end - 1; d >= b; d--) acc += S.cacc()[6 * d + k];
      S.cfrc()[w] = acc;
    }
    wave_sync();
    STAMP(37);
    for (int d = l; d < nv; d += W) {
      REAL s = 0;
      const REAL* cf = S.cfrc() + 6 * M.dof_bodyid[d];
#pragma unroll
      for (int k = 0; k < 6; k++) s += S.cdof()[6 * d + k] * cf[k];
      S.qfrc_bias()[d] = s;
    }
    }
    wave_sync();
    STAMP(38);
    STAMP(39);
  }
  // the leaves of the velocity stage go out at the very end of the phase, behind _actuation: nothing waits for a store that has no read behind it
  // (in front of _actuation, its first table read waited for all of them to land)
  template <bool DEFER = false>
  __device__ __forceinline__ void velocity_stores() {
    const int nv = M.nv, nb = M.nbody, nu = M.nu;
    put(out.actuator_length, S.act_length(), nu); put(out.actuator_velocity, S.act_velocity(), nu);
    if (!(DEFER && W > 16 && M.kv_defer)) { put(out.cvel, S.cvel(), 6 * nb); putnt(out.cdof_dot, S.cdof_dot(), 6 * nv); }  // (else: in front of the rne sweep)
    putnt(out.qfrc_passive, S.qfrc_passive(), nv); putnt(out.qfrc_bias, S.qfrc_bias(), nv);
    if (M.act_simple) row_copy_const<W>(out.actuator_moment, M.act_moment, nu * nv, e);  // the constant moment matrix (smooth.py:535-591): 4.5 KB per humanoid
  }

  // ---- muscle actuators (support.py:197-296) ---------------------------------------------------------------------------------------------
  static __device__ __forceinline__ REAL clamp_min(REAL x, REAL lo) { return x > lo ? x : lo; }
  static __device__ __forceinline__ REAL sq(REAL x) { return x * x; }
  static __device__ __forceinline__ REAL muscle_sigmoid(REAL x) {  // :197-202
    REAL sol = x * x * x * (3 * x * (2 * x - 5) + 10);
    sol = x <= 0 ? (REAL)0 : sol;
    return x >= 1 ? (REAL)1 : sol;
  }
  static __device__ __forceinline__ REAL muscle_dynamics(REAL ctrl, REAL act, const REAL* prm) {  // :205-232
    const REAL ctrlclamp = ctrl < 0 ? (REAL)0 : (ctrl > 1 ? (REAL)1 : ctrl), actclamp = act < 0 ? (REAL)0 : (act > 1 ? (REAL)1 : act);
    const REAL tau_act = prm[0] * ((REAL)0.5 + (REAL)1.5 * actclamp), tau_deact = prm[1] / ((REAL)0.5 + (REAL)1.5 * actclamp), width = prm[2];
    const REAL dctrl = ctrlclamp - act;
    const REAL tau_hard = dctrl > 0 ? tau_act : tau_deact;
    const REAL q = dctrl / (width + (width == 0 ? (REAL)(float)mjMINVAL : (REAL)0));  // math.safe_div
    const REAL tau_smooth = tau_deact + (tau_act - tau_deact) * muscle_sigmoid(q + (REAL)0.5);
    const REAL tau = width < (REAL)mjMINVAL ? tau_hard : tau_smooth;
    return dctrl / clamp_min(tau, (REAL)mjMINVAL);
  }
  static __device__ __forceinline__ REAL muscle_gain_length(REAL len, REAL lmin, REAL lmax) {  // :235-249
    const REAL a = (REAL)0.5 * (lmin + 1), b = (REAL)0.5 * (1 + lmax);
    const REAL out0 = (REAL)0.5 * sq((len - lmin) / clamp_min(a - lmin, (REAL)mjMINVAL));
    const REAL out1 = 1 - (REAL)0.5 * sq((1 - len) / clamp_min(1 - a, (REAL)mjMINVAL));
    const REAL out2 = 1 - (REAL)0.5 * sq((len - 1) / clamp_min(b - 1, (REAL)mjMINVAL));
    const REAL out3 = (REAL)0.5 * sq((lmax - len) / clamp_min(lmax - b, (REAL)mjMINVAL));
    REAL o = len <= b ? out2 : out3;
    o = len <= 1 ? out1 : o;
    o = len <= a ? out0 : o;
    return (lmin <= len && len <= lmax) ? o : (REAL)0;
  }
  static __device__ __forceinline__ REAL muscle_gain(REAL len, REAL vel, const REAL* lr, REAL acc0, const REAL* prm) {  // :252-278
    REAL force = prm[2];
    const REAL scale = prm[3], lmin = prm[4], lmax = prm[5], vmax = prm[6], fvmax = prm[8];
    force = force < 0 ? scale / clamp_min(acc0, (REAL)mjMINVAL) : force;
    const REAL L0 = (lr[1] - lr[0]) / clamp_min(prm[1] - prm[0], (REAL)mjMINVAL);
    const REAL L = prm[0] + (len - lr[0]) / clamp_min(L0, (REAL)mjMINVAL);
    const REAL V = vel / clamp_min(L0 * vmax, (REAL)mjMINVAL);
    const REAL FL = muscle_gain_length(L, lmin, lmax);
    const REAL y = fvmax - 1;
    REAL FV = V <= y ? fvmax - sq(y - V) / clamp_min(y, (REAL)mjMINVAL) : fvmax;
    FV = V <= 0 ? sq(V + 1) : FV;
    FV = V <= -1 ? (REAL)0 : FV;
    return -force * FL * FV;
  }
  static __device__ __forceinline__ REAL muscle_bias(REAL len, const REAL* lr, REAL acc0, const REAL* prm) {  // :281-296
    REAL force = prm[2];
    const REAL scale = prm[3], lmax = prm[5], fpmax = prm[7];
    force = force < 0 ? scale / clamp_min(acc0, (REAL)mjMINVAL) : force;
    const REAL L0 = (lr[1] - lr[0]) / clamp_min(prm[1] - prm[0], (REAL)mjMINVAL);
    const REAL L = prm[0] + (len - lr[0]) / clamp_min(L0, (REAL)mjMINVAL);
    const REAL b = (REAL)0.5 * (1 + lmax);
    const REAL out1 = -force * fpmax * (REAL)0.5 * sq((L - 1) / clamp_min(b - 1, (REAL)mjMINVAL));
    const REAL out2 = -force * fpmax * ((REAL)0.5 + (L - b) / clamp_min(b - 1, (REAL)mjMINVAL));
    const REAL o = L <= b ? out1 : out2;
    return L <= 1 ? (REAL)0 : o;
  }

  // ---- _actuation + _acceleration (forward.py:102-228, support.xfrc_accumulate :184-194) --------------------------------------------------
  template <bool FLUID>
  __device__ __forceinline__ void actuation() {
    const int l = lane_here();
    const int nv = M.nv, nu = M.nu;
    const bool off = (nu == 0) || (M.disableflags & DSBL_ACTUATION);
    if (off) {
      for (int i = l; i < M.na; i += W) S.act_dot()[i] = 0;
    } else {
      for (int i = l; i < nu; i += W) {
        // every table entry of the actuator is requested up front, unconditionally: behind a flag each one was a round trip of its own (flag, branch, value)
        REAL ctrl = in.ctrl ? in.ctrl[e * nu + i] : (REAL)0;
        const int ctrllim = M.act_ctrllimited[i], dyn = M.act_dyntype[i], gt = M.act_gaintype[i], bt = M.act_biastype[i], frclim = M.act_forcelimited[i];
        const REAL clo = M.act_ctrlrange[2 * i], chi = M.act_ctrlrange[2 * i + 1], flo = M.act_forcerange[2 * i], fhi = M.act_forcerange[2 * i + 1];
        const REAL* gp = M.act_gainprm + 9 * i;
        const REAL* bp = M.act_biasprm + 9 * i;
        const REAL gp0 = gp[0], gp1 = gp[1], gp2 = gp[2], bp0 = bp[0], bp1 = bp[1], bp2 = bp[2];
        if (!(M.disableflags & DSBL_CLAMPCTRL) && ctrllim) {
          ctrl = ctrl > clo ? ctrl : clo;
          ctrl = ctrl < chi ? ctrl : chi;
        }
        REAL ctrl_act = ctrl;
        if (dyn != DYN_NONE) {
          const int a = M.act_actadr[i];
          const REAL act = S.act()[a];
          if (dyn == DYN_INTEGRATOR) S.act_dot()[a] = ctrl;
          else if (dyn == DYN_MUSCLE) S.act_dot()[a] = muscle_dynamics(ctrl, act, M.act_dynprm + 3 * i);
          else {
            REAL tau = M.act_dynprm[3 * i];
            tau = tau > (REAL)mjMINVAL ? tau : (REAL)mjMINVAL;
            S.act_dot()[a] = (ctrl - act) / tau;
          }
          ctrl_act = S.act()[a + M.act_actnum[i] - 1];
        }
        const REAL len = S.act_length()[i], vel = S.act_velocity()[i];
        REAL gain = (gt == GAIN_FIXED) ? gp0 : gp0 + gp1 * len + gp2 * vel;
        REAL bias = (bt == BIAS_AFFINE) ? bp0 + bp1 * len + bp2 * vel : (REAL)0;
        if (gt == GAIN_MUSCLE) gain = muscle_gain(len, vel, M.act_lengthrange + 2 * i, M.act_acc0[i], gp);
        if (bt == BIAS_MUSCLE) bias = muscle_bias(len, M.act_lengthrange + 2 * i, M.act_acc0[i], bp);
        REAL force = gain * ctrl_act + bias;
        if (frclim) force = force < flo ? flo : (force > fhi ? fhi : force);
        S.act_force()[i] = force;
      }
    }
    wave_sync();
    STAMP(40);
    // xfrc_accumulate visits every body for every dof even when no force is applied (support.py:184-194).  With an
    // all-zero xfrc_applied every term is an exact +-0 and the sum is +0: the body loop is skipped in that case.
    bool any_xfrc = false;
    if (in.xfrc_applied) {
      bool nzf = false;
      for (int i = l; i < 6 * M.nbody; i += W) nzf = nzf || (in.xfrc_applied[e * M.nbody * 6 + i] != 0);
      any_xfrc = sub_any<W>(nzf);
    }
    for (int d = l; d < nv; d += W) {
      REAL s = 0;
      // (the dof's table entries and the applied force in one round trip: they used to hang off each other -- dof -> joint -> flag -> range, entry -> actuator -> gear)
      const int q0 = M.dof_act_adr[d], q1 = M.dof_act_adr[d + 1], flim = M.dof_frc_lim[d];
      const REAL flo = M.dof_frc_range[2 * d], fhi = M.dof_frc_range[2 * d + 1];
      const REAL qapp = in.qfrc_applied ? in.qfrc_applied[e * nv + d] : (REAL)0;
      if (!off) {
        // moment^T force: only the actuators on this dof have a non-zero moment entry (actuator order kept)
        for (int q = q0; q < q1; q++) {
          const int i = M.dof_act_id[q];
          const REAL cq = M.dof_act_coef[q];  // (a simple transmission's gear is the entry's constant coefficient)
          const REAL coef = M.act_simple ? cq : ((M.act_has_rot && M.dof_act_rot[q] >= 0) ? S.act_rot()[3 * i + M.dof_act_rot[q]] : cq);
          s += coef * S.act_force()[i];
        }
        if (FLUID && M.has_gravcomp) s = s + S.qfrc_gravcomp()[d] * (REAL)M.jnt_actgravcomp[M.dof_jntid[d]];  // forward.py:206-207 (the leaf is zero while gravity is disabled)
        if (flim) s = s < flo ? flo : (s > fhi ? fhi : s);
      }
      S.qfrc_actuator()[d] = s;
      // xfrc_accumulate: sum over bodies of jacp^T f + jacr^T tau at the body's inertial origin
      REAL acc = 0;
      if (any_xfrc) {
        for (int b = 0; b < M.nbody; b++) {
          const REAL* f = in.xfrc_applied + (e * M.nbody + b) * 6;
          REAL jp[3], jr[3];
          jac_dof<FLUID>(S.xipos() + 3 * b, b, d, jp, jr);
          const REAL ff[3] = {f[0], f[1], f[2]}, tt[3] = {f[3], f[4], f[5]};
          acc += dot3(jp, ff) + dot3(jr, tt);
        }
      }
      const REAL applied = qapp + acc;
      S.qfrc_smooth()[d] = ((S.qfrc_passive()[d] - S.qfrc_bias()[d]) + s) + applied;
    }
    wave_sync();
    STAMP(41);
    // qacc_smooth = M^-1 qfrc_smooth (forward.py:222-228) is the first thing the solver phase computes: the factor is
    // resident there anyway, and without it this phase's arena is small enough for two environments per wavefront
    if (!off) put(out.actuator_force, S.act_force(), nu);
    else if (out.actuator_force && in.actuator_force)  // actuation disabled: the reference returns before touching the leaf (forward.py:102-108), the caller's values stay
      for (int i = l; i < nu; i += W) out.actuator_force[e * nu + i] = in.actuator_force[e * nu + i];
    else if (out.actuator_force)
      for (int i = l; i < nu; i += W) out.actuator_force[e * nu + i] = 0;
    put(out.act_dot, S.act_dot(), M.na);
    put(out.qfrc_actuator, S.qfrc_actuator(), nv); put(out.qfrc_smooth, S.qfrc_smooth(), nv);
    STAMP(42);
  }

  // ---- solver (solver.py:244-553) -----------------------------------------------------------------------------------------------------------------------------
  struct LSPoint { REAL alpha, cost, d0, d1; };
  struct Ctx { REAL gauss, cost, prev_cost; int niter; };

  // rows the solver iterates on: every non-contact row + the rows of the contacts that are ACTIVE in this environment.  A contact
  // with dist >= includemargin has an all-zero Jacobian row and aref == 0 (constraint.py:440-451, :507-515, :552-561), so Jaref, jv,
  // force and every sum it enters stay exact zeros: those rows are left out of the arena instead of being carried through every
  // product, reduction and line-search point (the ant keeps 4 - 8 of its 60 contacts active)
  int nrow_;
  __device__ __forceinline__ int* row_src_lds() const { return reinterpret_cast<int*>(S.i_row_src()); }   // compact contact row -> Data row
  __device__ __forceinline__ int* row_dst_lds() const { return reinterpret_cast<int*>(S.i_row_dst()); }   // Data contact row - first contact row -> compact row, -1 = inactive
  __device__ __forceinline__ int* lim_dof_lds() const { return reinterpret_cast<int*>(S.i_lim_dof()); }
  __device__ __forceinline__ int* dof_limrow_lds() const { return reinterpret_cast<int*>(S.i_dof_limrow()); }

  // (dense_M * v).sum(-1).  qM stays in global memory (L2-resident: the CRB phase just wrote it): it is exactly
  // symmetric, so row i is read as column i -- lane i takes element i of every row, a coalesced load per term --
  // and the terms are still accumulated in column order.
  __device__ __forceinline__ void mul_M(const REAL* v, REAL* o) {
    const int nv = M.nv;
    if (M.sol_qm_lds) {
      for (int i = lane(); i < nv; i += W) o[i] = dot_seq(S.qMs() + i * nv, 1, v, 1, nv);
      return;
    }
    const REAL* g = out.qM + e * nv * nv;
    for (int i = lane(); i < nv; i += W) {
      REAL s = 0;
      int k = 0;
      for (; k + 14 <= nv; k += 14) {  // 14 rows in flight per round trip (two trips cover the humanoid's 27)
        REAL a[14];
#pragma unroll
        for (int t = 0; t < 14; t++) a[t] = g[(k + t) * nv + i];
#pragma unroll
        for (int t = 0; t < 14; t++) s += a[t] * v[k + t];
      }
      for (; k + 4 <= nv; k += 4) {
        REAL a[4];
#pragma unroll
        for (int t = 0; t < 4; t++) a[t] = g[(k + t) * nv + i];
#pragma unroll
        for (int t = 0; t < 4; t++) s += a[t] * v[k + t];
      }
      for (; k < nv; k++) s += g[k * nv + i] * v[k];
      o[i] = s;
    }
  }
  // efc_J @ v (- sub).  A joint-limit row has one non-zero (column lim_dof[r]); the zeros of the dense product add
  // exact zeros, so the single term is the same value.
  __device__ __forceinline__ void mul_J(const REAL* v, REAL* o, const REAL* sub) {
    const int nv = M.nv, nl = nf_() + M.nl;  // single-column rows (frictionloss, joint limits) come first
    for (int r = lane(); r < nrow_; r += W) {
      const REAL s = r < nl ? S.efc_Jl()[r] * v[lim_dof_lds()[r]] : dot_seq(S.efc_Jc() + (r - nl) * nv, 1, v, 1, nv);
      o[r] = sub ? s - sub[r] : s;
    }
  }

  __device__ __forceinline__ void update_constraint(Ctx& c) {  // :320-357
    const int l = lane_here();
    const int nv = M.nv, nefc = nrow_;
    REAL part = 0, fneg = 0, fpos = 0;
    const int nf = nf_();
    for (int r = l; r < nefc; r += W) {
      const REAL ja = S.s_Jaref()[r];
      bool act = (ja < 0) || is_eq_row(r);
      REAL floss_force = 0;
      if (is_fric(r)) {
        act = true;  // frictionloss row: quadratic inside |Jaref| < R f, linear (saturated force) outside (solver.py:326-342)
        const REAL fl = S.efc_fl()[fric_index(r)], D = S.efc_D()[r];
        const REAL rr = 1 / (D + (REAL)(D == 0) * (REAL)(float)mjMINVAL);
        const bool lin_neg = (ja <= -rr * fl) && (fl > 0), lin_pos = (ja >= rr * fl) && (fl > 0);
        act = act && !lin_neg && !lin_pos;
        floss_force = lin_neg ? fl : (lin_pos ? -fl : (REAL)0);
        fneg = (REAL)lin_neg * ((REAL)-0.5 * rr * fl * fl - fl * ja);
        fpos = (REAL)lin_pos * ((REAL)-0.5 * rr * fl * fl + fl * ja);
      }
      const REAL active = (REAL)act;
      S.s_force()[r] = S.efc_D()[r] * -ja * active + floss_force;
      part += S.efc_D()[r] * ja * ja * active;
    }
    REAL floss_cost = 0;
    if (nf > 0 || nft_() > 0) {  // the two frictionloss cost sums, rows in index order (every friction row sits in the first 64 rows: one per lane)
      REAL sn = 0, sp = 0;
      for (int r = 0; r < nf; r++) { sn += read_lane(fneg, r); sp += read_lane(fpos, r); }
      for (int r = nf + M.nl; r < nf + M.nl + nft_(); r++) { sn += read_lane(fneg, r); sp += read_lane(fpos, r); }
      floss_cost = sn + sp;
    }
    REAL gpart = 0;
    for (int d = l; d < nv; d += W) gpart += (S.s_Ma()[d] - S.qfrc_smooth()[d]) * (S.s_qacc()[d] - S.qacc_smooth()[d]);
    const REAL csum = wave_sum(part);
    const REAL g = wave_sum(gpart);
    c.gauss = (REAL)0.5 * g;
    c.prev_cost = c.cost;
    c.cost = ((REAL)0.5 * csum + c.gauss) + floss_cost;
    wave_sync();
  }
  // whether the dense (tendon) frictionloss row r is in its quadratic zone
  __device__ __forceinline__ bool dfric_quadratic(int r) const {
    const REAL ja = S.s_Jaref()[r], fl = S.efc_fl()[fric_index(r)], D = S.efc_D()[r];
    const REAL rr = 1 / (D + (REAL)(D == 0) * (REAL)(float)mjMINVAL);
    return !((ja <= -rr * fl) && (fl > 0)) && !((ja >= rr * fl) && (fl > 0));
  }
  // whether row r (a single-column row: frictionloss first, then joint limits) is in the quadratic (active) set
  __device__ __forceinline__ bool crow_active(int r) const {
    const REAL ja = S.s_Jaref()[r];
    if (r >= nf_()) return ja < 0;
    const REAL fl = S.efc_fl()[r], D = S.efc_D()[r];
    const REAL rr = 1 / (D + (REAL)(D == 0) * (REAL)(float)mjMINVAL);
    return !((ja <= -rr * fl) && (fl > 0)) && !((ja >= rr * fl) && (fl > 0));
  }
  // second half of _update_constraint: qfrc_constraint = J^T efc_force.  Only needed once a context is iterated on or
  // returned -- the cost-only contexts of the warm-start choice (:526-531) never read it.
  __device__ __forceinline__ void constraint_qfrc() {
    const int l = lane_here();
    const int nv = M.nv, nefc = nrow_;
    const int nl = nf_() + M.nl;
    if constexpr (W == 64 && sizeof(REAL) == 8) {
      if (nv <= 16) {  // one matrix-core tile: C[i][*] = sum_rows J[row][i] * force[row], rows in index order (every column of C carries the same vector)
        typedef MfmaTile<REAL> MT;
        const int c = l & 15, kq = l >> 4, nd = nefc - nl;
        typename MT::Acc acc = {0, 0, 0, 0};
        if (nl > 0) {  // the single-column rows come first; at most two of them (frictionloss, limit) touch dof i
          const int nslot = nf_() > 0 ? 2 : 1;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = MT::row(l, r);
            REAL s0 = 0;
            for (int q = 0; q < nslot; q++) {
              const int lr = i < nv ? dof_limrow_lds()[2 * i + q] : -1;
              if (lr >= 0) { const REAL f = S.s_force()[lr]; if (f != 0) s0 += S.efc_Jl()[lr] * f; }
            }
            acc[r] = s0;
          }
        }
        // out-of-range lanes read a clamped address and select 0 afterwards: unconditional LDS reads that pipeline across the unrolled blocks
        const REAL* jc = S.efc_Jc() + (c < nv ? c : 0);
        const REAL* fr = S.s_force() + nl;
        for (int k0 = 0; k0 < nd; k0 += 16) {  // four tiles per trip, unrolled by hand (the compiler does not unroll a loop around the intrinsic): eight reads in flight
          REAL jl[4], fl[4];
#pragma unroll
          for (int t = 0; t < 4; t++) { const int row = k0 + 4 * t + kq, rc = row < nd ? row : nd - 1; jl[t] = jc[rc * nv]; fl[t] = fr[rc]; }
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int row = k0 + 4 * t + kq;
            acc = MT::mac((row < nd && c < nv) ? jl[t] : (REAL)0, row < nd ? fl[t] : (REAL)0, acc);  // a tile past the last row adds exact zeros
          }
        }
        if (c == 0) {
#pragma unroll
          for (int r = 0; r < 4; r++) { const int i = MT::row(l, r); if (i < nv) S.s_qfrc()[i] = acc[r]; }
        }
        wave_sync();
        return;
      }
    }
    for (int c = l; c < nv; c += W) {  // rows in index order; one dof per lane (a second pass only beyond 64 dofs)
      REAL s = 0;
      if (nl > 0) {  // the single-column rows come first; at most two of them (frictionloss, limit) touch column c
        const int nslot = nf_() > 0 ? 2 : 1;  // without frictionloss rows the second slot of every dof is empty
        for (int q = 0; q < nslot; q++) {
          const int lr = dof_limrow_lds()[2 * c + q];
          if (lr >= 0) { const REAL f = S.s_force()[lr]; if (f != 0) s += S.efc_Jl()[lr] * f; }
        }
      }
      // a counted loop over the compacted rows: the loads of consecutive rows pipeline (the force is one broadcast read), where
      // picking the non-zero forces out of a ballot made every row a dependent round trip; rows with a zero force add +-0
      {
        const REAL* jc = S.efc_Jc() + c;
        const REAL* fr = S.s_force();
#pragma unroll 8
        for (int r = nl; r < nefc; r++) s += jc[(r - nl) * nv] * fr[r];
      }
      S.s_qfrc()[c] = s;
    }
    wave_sync();
  }

  __device__ __forceinline__ void update_gradient() {  // :359-376
    const int l = lane_here();
    const int nv = M.nv, nefc = nrow_;
    for (int d = l; d < nv; d += W) S.s_grad()[d] = (S.s_Ma()[d] - S.qfrc_smooth()[d]) - S.s_qfrc()[d];
    wave_sync();
    if (M.solver == SOL_CG) {
      chol_solve<W, true>(S.qLDp(), S.qLD_inv(), S.s_grad(), S.s_Mgrad(), nv);
    } else {
      // H = M + J^T diag(D active) J (solver.py:366-370); inactive rows contribute exact zeros and are skipped
      // only the lower triangle is ever read by the factorisation: one lane per packed entry (i, j <= i)
      const int np = (nv * (nv + 1)) / 2, nl = nf_() + M.nl;
      // weight of every dense row: D where the row is in the quadratic set, else 0 (the line search's quad buffer is dead here).
      // The row loop below is then a plain counted loop whose LDS reads pipeline; an inactive row adds (J * 0) * J = +-0.
      REAL* hw = S.s_quad();
      for (int r = nl + l; r < nefc; r += W) hw[r] = (is_dfric(r) ? dfric_quadratic(r) : (S.s_Jaref()[r] < 0 || is_eq_row(r))) ? S.efc_D()[r] : (REAL)0;
      wave_sync();
      bool tiled = false;
      if constexpr (W == 64 && sizeof(REAL) == 8) {
        if (nv <= 16) {  // one matrix-core tile: C = J^T diag(hw) J over the dense rows, started from the single-column rows' diagonal terms
          typedef MfmaTile<REAL> MT;
          tiled = true;
          const int c = l & 15, kq = l >> 4, nd = nefc - nl;
          typename MT::Acc acc = {0, 0, 0, 0};
          if (nl > 0) {
            const int nslot = nf_() > 0 ? 2 : 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int i = MT::row(l, r);
              REAL s0 = 0;
              if (i == c && i < nv) {
                for (int q = 0; q < nslot; q++) {
                  const int lr = dof_limrow_lds()[2 * i + q];
                  if (lr >= 0 && crow_active(lr)) { const REAL jl = S.efc_Jl()[lr]; s0 += (jl * S.efc_D()[lr] * (REAL)1) * jl; }
                }
              }
              acc[r] = s0;
            }
          }
          const REAL* jc = S.efc_Jc() + (c < nv ? c : 0);  // clamped addresses, selected to 0 afterwards: the LDS reads pipeline
          const REAL* hd = hw + nl;
          for (int k0 = 0; k0 < nd; k0 += 16) {  // four tiles per trip, unrolled by hand: eight reads in flight
            REAL jl[4], dl[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { const int row = k0 + 4 * t + kq, rc = row < nd ? row : nd - 1; jl[t] = jc[rc * nv]; dl[t] = hd[rc]; }
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const int row = k0 + 4 * t + kq;
              const REAL jv = (row < nd && c < nv) ? jl[t] : (REAL)0;
              acc = MT::mac(jv * (row < nd ? dl[t] : (REAL)0) * (REAL)1, jv, acc);  // a tile past the last row adds exact zeros
            }
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = MT::row(l, r);
            if (i < nv && c <= i) S.H()[(i * (i + 1)) / 2 + c] = (M.sol_qm_lds ? S.qMs()[i * nv + c] : out.qM[e * nv * nv + i * nv + c]) + acc[r];
          }
        }
      }
      if (!tiled)
      for (int w0 = 0; w0 < np; w0 += 2 * W) {  // two packed entries per lane and pass (nv = 12: 78 entries, one pass)
        const int wa = w0 + l, wb = w0 + W + l;
        int ia, ja, ib, jb;
        tri_unpack(wa < np ? wa : 0, ia, ja);
        tri_unpack(wb < np ? wb : 0, ib, jb);
        REAL sa = 0, sb = 0;
        if (nl > 0) {  // a single-column row only touches its own diagonal entry, and it precedes the contact rows
          const int nslot = nf_() > 0 ? 2 : 1;
          for (int q = 0; q < nslot; q++) {
            if (ia == ja) {
              const int lr = dof_limrow_lds()[2 * ia + q];
              if (lr >= 0 && crow_active(lr)) { const REAL jl = S.efc_Jl()[lr]; sa += (jl * S.efc_D()[lr] * (REAL)1) * jl; }
            }
            if (ib == jb) {
              const int lr = dof_limrow_lds()[2 * ib + q];
              if (lr >= 0 && crow_active(lr)) { const REAL jl = S.efc_Jl()[lr]; sb += (jl * S.efc_D()[lr] * (REAL)1) * jl; }
            }
          }
        }
        const REAL* jr = S.efc_Jc();
#pragma unroll 4
        for (int row = nl; row < nefc; row++, jr += nv) {
          const REAL dw = hw[row];
          sa += (jr[ia] * dw * (REAL)1) * jr[ja];
          sb += (jr[ib] * dw * (REAL)1) * jr[jb];
        }
        if (wa < np) S.H()[wa] = (M.sol_qm_lds ? S.qMs()[ia * nv + ja] : out.qM[e * nv * nv + ia * nv + ja]) + sa;
        if (wb < np) S.H()[wb] = (M.sol_qm_lds ? S.qMs()[ib * nv + jb] : out.qM[e * nv * nv + ib * nv + jb]) + sb;
      }
      wave_sync();
      STAMP(63);
      if (nv <= 16) {
        chol_factor_solve<W, REAL>(S.H(), S.s_grad(), S.s_Mgrad(), nv);
        STAMP(64);
      } else {
        chol_factor<W, REAL, 16, true>(S.H(), S.HL(), nv);
        chol_inv_diag<W, false>(S.HL(), S.HL_inv(), nv);
        wave_sync();
        STAMP(64);
        chol_solve<W, false>(S.HL(), S.HL_inv(), S.s_grad(), S.s_Mgrad(), nv);
        STAMP(65);
      }
    }
  }

  __device__ __forceinline__ LSPoint ls_point(const REAL* qg, REAL alpha) {  // point_fn :396-422
    REAL q0 = 0, q1 = 0, q2 = 0;
    REAL f0n = 0, f0p = 0, f1n = 0, f1p = 0;
    const int nf = nf_();
    for (int r = lane(); r < nrow_; r += W) {
      const REAL ja = S.s_Jaref()[r], jv = S.s_jv()[r];
      const REAL x = ja + alpha * jv;
      bool act = (x < 0) || is_eq_row(r);
      if (is_fric(r)) {  // frictionloss row (solver.py:404-416): active unless in a linear zone
        act = true;
        const REAL fl = S.efc_fl()[fric_index(r)], D = S.efc_D()[r];
        const REAL rf = (1 / (D + (REAL)(D == 0) * (REAL)(float)mjMINVAL)) * fl;
        const bool ln = (x <= -rf) && (fl > 0), lp = (x >= rf) && (fl > 0);
        f0n = (REAL)ln * fl * ((REAL)-0.5 * rf - ja);
        f0p = (REAL)lp * fl * ((REAL)-0.5 * rf + ja);
        f1n = (REAL)ln * (-fl * jv);
        f1p = (REAL)lp * (fl * jv);
        act = act && !ln && !lp;
      }
      const REAL a = (REAL)act;
      q0 += S.s_quad()[3 * r] * a;
      q1 += S.s_quad()[3 * r + 1] * a;
      q2 += S.s_quad()[3 * r + 2] * a;
    }
    q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2);
    REAL fa0 = 0, fa1 = 0;
    if (nf > 0 || nft_() > 0) {  // frictionloss adjustments, rows in index order (one row per lane)
      REAL s0n = 0, s0p = 0, s1n = 0, s1p = 0;
      for (int r = 0; r < nf; r++) { s0n += read_lane(f0n, r); s0p += read_lane(f0p, r); s1n += read_lane(f1n, r); s1p += read_lane(f1p, r); }
      for (int r = nf + M.nl; r < nf + M.nl + nft_(); r++) { s0n += read_lane(f0n, r); s0p += read_lane(f0p, r); s1n += read_lane(f1n, r); s1p += read_lane(f1p, r); }
      fa0 = s0n + s0p; fa1 = s1n + s1p;
    }
    const REAL t0 = (qg[0] + q0) + fa0, t1 = (qg[1] + q1) + fa1, t2 = (qg[2] + q2) + 0;
    LSPoint p;
    p.alpha = alpha;
    p.cost = alpha * alpha * t2 + alpha * t1 + t0;
    p.d0 = 2 * alpha * t2 + t1;
    p.d1 = 2 * t2 + (REAL)(t2 == 0) * (REAL)mjMINVAL;
    return p;
  }
  // the three candidates of a line-search iteration in ONE pass over the rows (each row's Jaref / jv / quad read once, nine
  // interleaved reductions): per candidate the same per-lane partial sums in the same order as ls_point, so the same bits.
  // Used by the float32 instantiations (+2.5 % on the ant, where 188 rows make three trips; -1.3 % on the float64 humanoid).
  __device__ __forceinline__ void ls_points3(const REAL* qg, const REAL (&alpha)[3], LSPoint (&p)[3]) {
    REAL q0[3] = {0, 0, 0}, q1[3] = {0, 0, 0}, q2[3] = {0, 0, 0};
    REAL f0n[3] = {0, 0, 0}, f0p[3] = {0, 0, 0}, f1n[3] = {0, 0, 0}, f1p[3] = {0, 0, 0};
    const int nf = nf_();
    for (int r = lane(); r < nrow_; r += W) {
      const REAL ja = S.s_Jaref()[r], jv = S.s_jv()[r];
      const REAL u0 = S.s_quad()[3 * r], u1 = S.s_quad()[3 * r + 1], u2 = S.s_quad()[3 * r + 2];
      const bool eq = is_eq_row(r);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const REAL x = ja + alpha[k] * jv;
        bool act = (x < 0) || eq;
        if (is_fric(r)) {
          act = true;
          const REAL fl = S.efc_fl()[fric_index(r)], D = S.efc_D()[r];
          const REAL rf = (1 / (D + (REAL)(D == 0) * (REAL)(float)mjMINVAL)) * fl;
          const bool ln = (x <= -rf) && (fl > 0), lp = (x >= rf) && (fl > 0);
          f0n[k] = (REAL)ln * fl * ((REAL)-0.5 * rf - ja);
          f0p[k] = (REAL)lp * fl * ((REAL)-0.5 * rf + ja);
          f1n[k] = (REAL)ln * (-fl * jv);
          f1p[k] = (REAL)lp * (fl * jv);
          act = act && !ln && !lp;
        }
        const REAL a = (REAL)act;
        q0[k] += u0 * a; q1[k] += u1 * a; q2[k] += u2 * a;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { q0[k] = wave_sum(q0[k]); q1[k] = wave_sum(q1[k]); q2[k] = wave_sum(q2[k]); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      REAL fa0 = 0, fa1 = 0;
      if (nf > 0 || nft_() > 0) {
        REAL s0n = 0, s0p = 0, s1n = 0, s1p = 0;
        for (int r = 0; r < nf; r++) { s0n += read_lane(f0n[k], r); s0p += read_lane(f0p[k], r); s1n += read_lane(f1n[k], r); s1p += read_lane(f1p[k], r); }
        for (int r = nf + M.nl; r < nf + M.nl + nft_(); r++) { s0n += read_lane(f0n[k], r); s0p += read_lane(f0p[k], r); s1n += read_lane(f1n[k], r); s1p += read_lane(f1p[k], r); }
        fa0 = s0n + s0p; fa1 = s1n + s1p;
      }
      const REAL t0 = (qg[0] + q0[k]) + fa0, t1 = (qg[1] + q1[k]) + fa1, t2 = (qg[2] + q2[k]) + 0;
      p[k].alpha = alpha[k];
      p[k].cost = alpha[k] * alpha[k] * t2 + alpha[k] * t1 + t0;
      p[k].d0 = 2 * alpha[k] * t2 + t1;
      p[k].d1 = 2 * t2 + (REAL)(t2 == 0) * (REAL)mjMINVAL;
    }
  }
  __device__ __forceinline__ static bool ls_swap(REAL cur, REAL cand, bool not_bracketed) {  // _swap :440-449
    const bool in_bracket = ((cur < cand) && (cand < 0)) || ((cur > cand) && (cand > 0));
    return in_bracket || (not_bracketed && (r_abs(cand) < r_abs(cur)));
  }

  __device__ __forceinline__ void linesearch(Ctx& c) {  // :378-497
    const int l = lane_here();
    const int nv = M.nv, nefc = nrow_;
    const REAL scale = (REAL)(M.meaninertia * (double)(nv > 1 ? nv : 1));
    REAL ss = 0;
    bool nz = false;
    for (int d = l; d < nv; d += W) { ss += S.s_search()[d] * S.s_search()[d]; nz = nz || (S.s_search()[d] != 0); }
    ss = wave_sum(ss);
    const REAL snorm = wave_any(nz) ? r_sqrt<REAL>(ss) : (REAL)0;
    const REAL smag = snorm * scale;
    const REAL gtol = (REAL)(M.tolerance * M.ls_tolerance) * smag;
    mul_M(S.s_search(), S.s_mv());
    mul_J(S.s_search(), S.s_jv(), nullptr);
    wave_sync();
    STAMP(57);
    REAL a = 0, b = 0, cc = 0;
    for (int d = l; d < nv; d += W) { a += S.s_search()[d] * S.s_Ma()[d]; b += S.s_search()[d] * S.qfrc_smooth()[d]; cc += S.s_search()[d] * S.s_mv()[d]; }
    a = wave_sum(a); b = wave_sum(b); cc = wave_sum(cc);
    const REAL qg[3] = {c.gauss, a - b, (REAL)0.5 * cc};
    for (int r = l; r < nefc; r += W) {
      const REAL ja = S.s_Jaref()[r], jv = S.s_jv()[r], D = S.efc_D()[r];
      S.s_quad()[3 * r] = ((REAL)0.5 * ja * ja) * D;
      S.s_quad()[3 * r + 1] = (jv * ja) * D;
      S.s_quad()[3 * r + 2] = ((REAL)0.5 * jv * jv) * D;
    }
    wave_sync();
    STAMP(58);
    const LSPoint p0 = ls_point(qg, 0);
    const LSPoint p1 = ls_point(qg, p0.alpha - p0.d0 / p0.d1);
    const bool early = r_abs(p1.d0) < gtol;
    LSPoint lo, hi;
    if (p1.d0 < p0.d0) { hi = p0; lo = p1; } else { hi = p1; lo = p0; }
    bool swap = !early;
    int ls_iter = 0;
    const bool fixed = flags & MJH_FLAG_FIXED_ITERATIONS;
    for (;;) {
      if (fixed) { if (ls_iter >= M.ls_iterations) break; }
      else {
        bool done = ls_iter >= M.ls_iterations;
        done |= !swap;
        done |= (lo.d0 < 0) && (lo.d0 > -gtol);
        done |= (hi.d0 > 0) && (hi.d0 < gtol);
        if (done) break;
      }
      LSPoint lo_next, hi_next, mid;
      if (sizeof(REAL) == 4) {
        const REAL al[3] = {lo.alpha - lo.d0 / lo.d1, hi.alpha - hi.d0 / hi.d1, (REAL)0.5 * (lo.alpha + hi.alpha)};
        LSPoint pp[3];
        ls_points3(qg, al, pp);
        lo_next = pp[0]; hi_next = pp[1]; mid = pp[2];
      } else {
        lo_next = ls_point(qg, lo.alpha - lo.d0 / lo.d1);
        hi_next = ls_point(qg, hi.alpha - hi.d0 / hi.d1);
        mid = ls_point(qg, (REAL)0.5 * (lo.alpha + hi.alpha));
      }
      const bool nb = (lo.d0 < 0) == (hi.d0 < 0);
      const bool s1 = ls_swap(lo.d0, lo_next.d0, nb); if (s1) lo = lo_next;
      const bool s2 = ls_swap(lo.d0, mid.d0, nb); if (s2) lo = mid;
      const bool s3 = ls_swap(lo.d0, hi_next.d0, nb); if (s3) lo = hi_next;
      const bool s4 = ls_swap(hi.d0, hi_next.d0, nb); if (s4) hi = hi_next;
      const bool s5 = ls_swap(hi.d0, mid.d0, nb); if (s5) hi = mid;
      const bool s6 = ls_swap(hi.d0, lo_next.d0, nb); if (s6) hi = lo_next;
      swap = s1 | s2 | s3 | s4 | s5 | s6;
      ls_iter++;
    }
    const REAL improved = (REAL)((lo.cost < p0.cost) || (hi.cost < p0.cost));
    const REAL alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
    for (int d = l; d < nv; d += W) {
      S.s_qacc()[d] = S.s_qacc()[d] + improved * S.s_search()[d] * alpha;
      S.s_Ma()[d] = S.s_Ma()[d] + improved * S.s_mv()[d] * alpha;
    }
    for (int r = l; r < nefc; r += W) S.s_Jaref()[r] = S.s_Jaref()[r] + improved * S.s_jv()[r] * alpha;
    wave_sync();
    STAMP(59);
  }

  // first half of the solver phase's inputs and _acceleration's solve (forward.py:222-228): qacc_smooth = M^-1 qfrc_smooth
  __device__ __forceinline__ void load_factor_and_accelerate(bool solving) {
    const int l = lane(), nv = M.nv;
    if (solving) {  // the small vectors of the phase in one batch of loads (one L2 round trip instead of eight)
      const bool from_in = !KA.state_from_cur;
      REAL* const dst[6] = {S.qfrc_smooth(), S.qpos(), S.qvel(), S.act(), S.act_dot(), S.qacc_warm()};  // (efc_D / efc_aref are gathered row by row in load_solver_inputs)
      const REAL* const src[6] = {out.qfrc_smooth, KA.cur.qpos, from_in ? in.qvel : KA.cur.qvel, KA.state_from_cur ? KA.cur.act : in.act,
                                  out.act_dot, KA.warm_src};
      const int cnt[6] = {nv, M.nq, nv, M.na, M.na, M.nefc > 0 ? nv : 0};
      multi_load<W, 6, 1>(dst, src, cnt, e);
      if (from_in && KA.do_step) for (int i = l; i < nv; i += W) S.qvel()[i] = checked(S.qvel()[i], (REAL)0);  // _check_state (same lane wrote it)
    } else {
      row_load<W>(S.qfrc_smooth(), out.qfrc_smooth, nv, e);
    }
    const REAL* gL = out.qLD + e * nv * nv;
    {  // lower triangle of the factor into packed rows: only the entries that are kept are requested, three loads in flight per trip
      const int np = (nv * (nv + 1)) / 2;
      int p = l;
      for (; p + 2 * W < np; p += 3 * W) {
        int i0, k0, i1, k1, i2, k2;
        tri_unpack(p, i0, k0); tri_unpack(p + W, i1, k1); tri_unpack(p + 2 * W, i2, k2);
        const REAL a = gL[i0 * nv + k0], b = gL[i1 * nv + k1], c = gL[i2 * nv + k2];
        S.qLDp()[p] = a; S.qLDp()[p + W] = b; S.qLDp()[p + 2 * W] = c;
      }
      for (; p < np; p += W) {
        int i, k;
        tri_unpack(p, i, k);
        S.qLDp()[p] = gL[i * nv + k];
      }
    }
    if (solving) load_solver_inputs();  // the constraint Jacobian and the row tables: requested before the first wait too
    wave_sync();
    STAMP(51);
    chol_inv_diag<W, true>(S.qLDp(), S.qLD_inv(), nv);
    wave_sync();
    chol_solve<W, true>(S.qLDp(), S.qLD_inv(), S.qfrc_smooth(), S.qacc_smooth(), nv);
    STAMP(52);
    put(out.qacc_smooth, S.qacc_smooth(), nv);
  }
  // rows `src[r]` (an LDS table of Data row numbers) of the efc_J leaf -> consecutive rows of an LDS block, DEPTH requests per lane in flight per
  // round trip.  The whole copy is two trips for the humanoid's 32 x 27 block; at four per trip it was seven dependent ones, the longest
  // section of the solver phase.  Lanes past the end read a clamped address (no exec-masked branches between the loads) and skip the store.
  template <int DEPTH>
  __device__ __forceinline__ void gather_rows(REAL* dstJ, const REAL* gJ, const int* src, int n) const {
    const int nv = M.nv;
    for (int i = lane(); i < n; i += DEPTH * W) {
      REAL v[DEPTH];
#pragma unroll
      for (int t = 0; t < DEPTH; t++) {
        const int idx = i + t * W, idc = idx < n ? idx : n - 1;
        int r, k;
        split_index(idc, nv, M.inv_nv, r, k);
        v[t] = gJ[src[r] * nv + k];
      }
#pragma unroll
      for (int t = 0; t < DEPTH; t++) { const int idx = i + t * W; if (idx < n) dstJ[idx] = v[t]; }
    }
  }

  __device__ __forceinline__ void load_solver_inputs() {
    const int nv = M.nv, nefc = M.nefc;
    if (M.sol_qm_lds) row_load<W>(S.qMs(), out.qM, nv * nv, e);
    if (nefc > 0) {
      const int l = lane(), nl = nf_() + M.nl;
      const REAL* gJ = out.efc_J + e * nefc * nv;
      const int ne = ne_();
      {  // the three small row tables in ONE loop: their loads overlap instead of queueing behind each other's LDS stores
        const int nfr = nf_(), n2 = nl > 0 ? 2 * nv : 0;
        const int nmax = nl > n2 ? nl : n2;
        for (int r = l; r < nmax; r += W) {
          const bool a = r < nl, b = r < nfr, c = r < n2;
          const int dr = a ? M.lim_dof[r] : 0;
          const int fd = b ? M.fric_dof[r] : 0;
          const int lr = c ? M.dof_limrow[r] : 0;
          const REAL jl = a ? gJ[ext_row(r) * nv + dr] : (REAL)0;
          const REAL fl = b ? M.dof_frictionloss[fd] : (REAL)0;
          if (a) { lim_dof_lds()[r] = dr; S.efc_Jl()[r] = jl; }
          if (b) S.efc_fl()[r] = fl;
          if (c) dof_limrow_lds()[r] = lr;
        }
        for (int j = l; j < nft_(); j += W) S.efc_fl()[nfr + j] = M.tendon_frictionloss[M.fric_tendon[j]];
      }
      const int nlim = FRIC ? M.nlb + M.nlt : 0;  // dense limit rows: gathered row by row
      const int ndense0 = nft_() + ne + nlim;     // dense rows ahead of the contacts: tendon frictionloss, equality, ball / tendon limits
      const int c0 = nl + ndense0;                // first contact row (the contact rows close both row orders)
      int nact = 0;
      {  // active contacts -> compact row tables.  One contact per lane, exclusive prefix sum of the active contacts' row counts.
        const int ncon = M.ncon;
        const bool elliptic = M.cone == CONE_ELLIPTIC;
        for (int base = 0; base < ncon; base += W) {
          const int c = base + l;
          const bool valid = c < ncon;
          int rows = 0, start = 0;
          bool act = false;
          if (valid) {
            const int dim = M.con_dim[c];
            rows = dim == 1 ? 1 : (elliptic ? dim : 2 * (dim - 1));
            start = M.con_efc_address[c] - c0;
            act = (out.contact_dist[e * ncon + c] - (M.topk ? out.contact_includemargin[e * ncon + c] : M.con_includemargin[c])) < 0;
          }
          int excl, tot;
          if (M.con_rows) {  // one condim: the prefix sum of the row counts is a population count times that count
            excl = sub_prefix_count<W>(act, tot) * M.con_rows + nact;
            tot *= M.con_rows;
          } else {
            int x = act ? rows : 0;
            for (int o = 1; o < W; o <<= 1) { const int y = __shfl_up(x, o, W); if (l >= o) x += y; }
            excl = x - (act ? rows : 0) + nact;
            tot = __builtin_amdgcn_readlane(x, W - 1);
          }
          for (int k = 0; k < rows; k++) {
            row_dst_lds()[start + k] = act ? excl + k : -1;
            if (act) row_src_lds()[excl + k] = c0 + start + k;
          }
          nact += tot;
        }
      }
      nrow_ = c0 + nact;
      wave_sync();
      {
        for (int i = l; i < ndense0 * nv; i += W) {
          int k, c;
          split_index(i, nv, M.inv_nv, k, c);
          S.efc_Jc()[i] = gJ[ext_row(nl + k) * nv + c];
        }
        // the rows of the active contacts, in row order (the rows of one contact are adjacent in memory: runs of rows * nv elements)
        gather_rows<8>(S.efc_Jc() + ndense0 * nv, gJ, row_src_lds(), nact * nv);
      }
      for (int r = l; r < nrow_; r += W) {
        const int x = r < c0 ? ext_row(r) : row_src_lds()[r - c0];
        S.efc_D()[r] = out.efc_D[e * nefc + x];
        S.efc_aref()[r] = out.efc_aref[e * nefc + x];
      }
    }
  }

  // solver.solve :244-553 as ONE loop whose body contains each heavy routine exactly once (line search, constraint
  // update, gradient): the kernel is fully inlined, so every extra call site would be another copy of the routine and
  // the iteration loop would no longer fit the instruction cache.  Steps, in order:
  //   P_SMOOTH : context of qacc_smooth (cost only)                       } warm start enabled only (:526-531)
  //   P_WARM   : context of qacc_warmstart (cost only); pick the cheaper  }
  //   P_START  : the chosen context with gradient and search direction (:293-318).  When the warm start won, its
  //              context is still in place and only the gradient is added.
  //   P_ITER   : cond :501-508, then body :509-524 (line search, constraint update, gradient, new direction)
  __device__ __forceinline__ void solve() {
    const int l = lane_here();
    const int nv = M.nv, nefc = M.nefc;
    const REAL scale = (REAL)(M.meaninertia * (double)(nv > 1 ? nv : 1));
    const bool fixed = flags & MJH_FLAG_FIXED_ITERATIONS;
    enum { P_SMOOTH = 0, P_WARM = 1, P_START = 2, P_ITER = 3 };
    Ctx c;
    c.gauss = 0; c.cost = 0; c.prev_cost = 0; c.niter = 0;
    bool use_warm = false;
    REAL smooth_cost = 0;
    int ph = (M.disableflags & DSBL_WARMSTART) ? P_START : P_SMOOTH;
    int it = 0;
    for (;;) {
      bool do_init, do_grad, do_ls;
      const REAL* src = S.qacc_smooth();
      if (ph == P_SMOOTH) { do_init = true; do_grad = false; do_ls = false; }
      else if (ph == P_WARM) { do_init = true; do_grad = false; do_ls = false; src = S.qacc_warm(); }
      else if (ph == P_START) { do_init = !use_warm; do_grad = true; do_ls = false; }
      else {
        if (M.iterations == 1) { if (it >= 1) break; }
        else if (fixed) { if (it >= M.iterations) break; }
        else {  // cond :501-508
          REAL gg = 0;
          for (int d = l; d < nv; d += W) gg += S.s_grad()[d] * S.s_grad()[d];
          gg = wave_sum(gg);
          const REAL improvement = (c.prev_cost - c.cost) / scale;
          const REAL gradient = r_sqrt<REAL>(gg) / scale;
          bool done = c.niter >= M.iterations;
          done |= improvement < (REAL)M.tolerance;
          done |= gradient < (REAL)M.tolerance;
          if (done) break;
        }
        // the gradient (and the next search direction) of the last allowed iteration is never read: cond's
        // `niter >= iterations` ends the loop whatever the gradient norm is
        do_init = false; do_ls = true; do_grad = !(it + 1 >= M.iterations);
      }
      if (do_ls) {
        linesearch(c);
        STAMP(60);
        for (int d = l; d < nv; d += W) { S.s_pgrad()[d] = S.s_grad()[d]; S.s_pMgrad()[d] = S.s_Mgrad()[d]; }
        wave_sync();
      }
      if (do_init) {  // _Context.create :293-318
        for (int d = l; d < nv; d += W) S.s_qacc()[d] = src[d];
        wave_sync();
        mul_J(S.s_qacc(), S.s_Jaref(), S.efc_aref());
        if (ph == P_SMOOTH) {
          // cost-only context at qacc = qacc_smooth: its Gauss term is (Ma - qfrc_smooth) . (qacc - qacc_smooth) = x . 0, an exact zero for
          // any finite Ma -- the M product (27 dependent row reads from L2) is skipped; if this context wins, P_START rebuilds it in full
          for (int d = l; d < nv; d += W) S.s_Ma()[d] = 0;
        } else {
          mul_M(S.s_qacc(), S.s_Ma());
        }
        c.gauss = 0; c.cost = (REAL)INFINITY; c.prev_cost = 0; c.niter = 0;
        for (int d = l; d < nv; d += W) { S.s_grad()[d] = 0; S.s_Mgrad()[d] = 0; S.s_search()[d] = 0; }
        wave_sync();
      }
      STAMP(68);
      if (do_ls || do_init) update_constraint(c);
      STAMP(67);
      if (ph >= P_START) constraint_qfrc();
      STAMP(66);
      if (do_grad) {
        update_gradient();
        if (ph == P_START || M.solver == SOL_NEWTON) {
          for (int d = l; d < nv; d += W) S.s_search()[d] = -S.s_Mgrad()[d];
        } else {  // Polak-Ribiere :519-523
          REAL num = 0, den = 0;
          for (int d = l; d < nv; d += W) { num += S.s_grad()[d] * (S.s_Mgrad()[d] - S.s_pMgrad()[d]); den += S.s_pgrad()[d] * S.s_pMgrad()[d]; }
          num = wave_sum(num); den = wave_sum(den);
          REAL beta = num / (den > (REAL)mjMINVAL ? den : (REAL)mjMINVAL);
          beta = beta > 0 ? beta : (REAL)0;
          for (int d = l; d < nv; d += W) S.s_search()[d] = -S.s_Mgrad()[d] + beta * S.s_search()[d];
        }
        wave_sync();
      }
      if (ph == P_SMOOTH) { smooth_cost = c.cost; ph = P_WARM; }
      else if (ph == P_WARM) { use_warm = __builtin_amdgcn_readfirstlane((int)(c.cost < smooth_cost)) != 0; ph = P_START; STAMP(54); }  // wave-uniform by construction: keep the step flags scalar
      else if (ph == P_START) { ph = P_ITER; STAMP(55); }
      else { c.niter++; it++; }
    }
    for (int d = l; d < nv; d += W) { S.qacc()[d] = S.s_qacc()[d]; S.qacc_warm()[d] = S.s_qacc()[d]; S.qfrc_constraint()[d] = S.s_qfrc()[d]; }
    wave_sync();
    STAMP(61);
    put(out.qacc, S.qacc(), nv); put(out.qacc_warmstart, S.qacc(), nv); put(out.qfrc_constraint, S.qfrc_constraint(), nv);
    if (out.efc_force) {  // Data order; the rows of inactive contacts carry exact zeros
      const int c0 = nf_() + M.nl + nft_() + ne_() + nlim_rows();
      for (int r = l; r < c0; r += W) out.efc_force[e * nefc + ext_row(r)] = S.s_force()[r];
      for (int r = l; r < nefc - c0; r += W) { const int q = row_dst_lds()[r]; out.efc_force[e * nefc + c0 + r] = q >= 0 ? S.s_force()[c0 + q] : (REAL)0; }
    }
    STAMP(62);
  }

  // ---- integrators (forward.py:231-370) ---------------------------------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void integrate_pos(const REAL* qpos, const REAL* qvel, REAL dt, REAL* o) {  // :231-252, one lane per joint
    for (int j = lane(); j < M.njnt; j += W) {
      const int t = M.jnt_type[j], qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
      if (t == JNT_FREE) {
        for (int i = 0; i < 3; i++) o[qa + i] = qpos[qa + i] + dt * qvel[da + i];
        REAL q[4] = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]}, w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]}, r[4];
        quat_integrate(q, w, dt, r);
        for (int i = 0; i < 4; i++) o[qa + 3 + i] = r[i];
      } else if (t == JNT_BALL) {
        REAL q[4] = {qpos[qa], qpos[qa + 1], qpos[qa + 2], qpos[qa + 3]}, w[3] = {qvel[da], qvel[da + 1], qvel[da + 2]}, r[4];
        quat_integrate(q, w, dt, r);
        for (int i = 0; i < 4; i++) o[qa + i] = r[i];
      } else {
        o[qa] = qpos[qa] + dt * qvel[da];
      }
    }
  }

  // act <- act + act_dot * dt (or exact filter), clamped (forward.py:267-294); one lane per actuator
  __device__ __forceinline__ void advance_act(const REAL* act0, const REAL* act_dot, REAL* dst) {
    const REAL dt = M.timestep;
    for (int i = lane(); i < M.nu; i += W) {
      const int dyn = M.act_dyntype[i];
      if (dyn == DYN_NONE) continue;
      const int a = M.act_actadr[i];
      REAL act = act0[a];
      if (dyn == DYN_FILTEREXACT) {
        REAL tau = M.act_dynprm[3 * i];
        tau = tau > (REAL)mjMINVAL ? tau : (REAL)mjMINVAL;
        act = act + act_dot[a] * tau * (1 - r_exp<REAL>(-dt / tau));
      } else {
        act = act + act_dot[a] * dt;
      }
      if (M.act_actlimited[i]) {
        const REAL lo = M.act_actrange[2 * i], hi = M.act_actrange[2 * i + 1];
        act = act < lo ? lo : (act > hi ? hi : act);
      }
      if (dst) dst[e * M.na + a] = act;
    }
  }

  // _advance :255-310 into the RETURNED Data (KArgs::out): qvel += qacc dt, qpos integrated with qvel_for_pos
  // (the new qvel when null), act, time.
  __device__ __forceinline__ void advance(const REAL* qpos0, const REAL* qvel0, const REAL* act0, REAL time0, const REAL* act_dot, const REAL* qacc, const REAL* qvel_for_pos) {
    const int l = lane_here();
    const REAL dt = M.timestep;
    const StatePtrs<REAL>& fin = KA.fin;
    advance_act(act0, act_dot, fin.act);
    for (int d = l; d < M.nv; d += W) S.tmp_nv()[d] = qvel0[d] + qacc[d] * dt;
    wave_sync();
    integrate_pos(qpos0, qvel_for_pos ? qvel_for_pos : S.tmp_nv(), dt, S.tmp_nq());
    wave_sync();
    row_store<W>(fin.qpos, S.tmp_nq(), M.nq, e);
    row_store<W>(fin.qvel, S.tmp_nv(), M.nv, e);
    if (l == 0 && fin.time) fin.time[e] = time0 + dt;
  }

  // ---- phase drivers ------------------------------------------------------------------------------------------------------------------------------------------------
  template <bool DEFER = false, bool KEEPG = false, bool XK = KEEPG>  // XK: the out-of-lockstep order of the kinematics stage is compiled in (whole-pass and stage kernels)
  __device__ __forceinline__ void run_kin() {
    STAMP0();
    load_qpos(true);
    wave_sync();
    STAMP(1);
    const bool com_first = XK && (__builtin_popcount((unsigned)blockIdx.x & (unsigned)KA.xswap_k) & 1) != 0;
    kinematics<DEFER, KEEPG, XK>(KA.rk_stage <= 0, com_first);
    if (!com_first) com_pos<DEFER>();
  }
  __device__ __forceinline__ void run_crb() { crb_factor<false>(); }
  __device__ __forceinline__ void run_con() {
    STAMP0();
    // the model-constant contact leaves (5 KB per environment for the ant's 60 contacts) go out LAST: copied between the narrow phase and the rows, every table read of the
    // rows queued behind that burst (vmcnt is in order) -- at the end nothing of this wave waits for it, and the next wave's arithmetic runs while it drains
    if (M.ncon > 0) collision<0, false, true>();
    if (KA.stages & 0x78) make_constraint();
    if (M.ncon > 0 && !(FRIC && M.topk) && !consts_done_) contact_const_stores();
  }
  // the solver's loads that depend on nothing this kernel computes (fused constraint + solver kernel)
  template <int NMAX>
  __device__ __forceinline__ void sol2_prefetch(Sol2Pre<REAL, NMAX>& P) {
    const int l = lane_here();
    const int nq = M.nq, nv = M.nv, na = M.na;
    const bool dof = l < nv;
    P.f = (dof && out.qfrc_smooth) ? out.qfrc_smooth[e * nv + l] : (REAL)0;
    {
      const REAL* gL = out.qLD + e * nv * nv;
#pragma unroll
      for (int k = 0; k < NMAX; k++) P.T.t[k] = (dof && k < nv) ? (k <= l ? gL[l * nv + k] : gL[k * nv + l]) : (REAL)0;
    }
    P.T.inv = 0;
    const REAL* gq = KA.cur.qpos + e * nq;
#pragma unroll
    for (int j = 0; j < 2; j++) P.qp[j] = l + W * j < nq ? gq[l + W * j] : (REAL)0;
    P.ac = 0; P.ad = 0;
    if (l < na) { const REAL* ga = KA.state_from_cur ? KA.cur.act : in.act; P.ac = ga ? ga[e * na + l] : (REAL)0; P.ad = out.act_dot ? out.act_dot[e * na + l] : (REAL)0; }
    P.warm = (dof && KA.warm_src && M.nefc > 0) ? KA.warm_src[e * nv + l] : (REAL)0;
  }
  // The constraint stage of the fused kernel: make_constraint() for the plain row set (slide / hinge limits + contacts of ONE condim, nl <= W), with
  //  * the limit rows in registers, lane r <-> row r, from the joint's position to efc_D / efc_aref (constraint.py:338-372, 683-693);
  //  * the rows of the ACTIVE contacts only, built in compact order in S.efc_Jc() -- one lane per (active contact, dof), the contact's bodies / dof masks from the
  //    per-contact tables (DevModel::con_body, con_dmask), as the small-model kernel does; an inactive contact's rows are exact zeros in the reference and in the leaf;
  //  * efc_D / efc_aref of every row stored to their leaves here (lane <-> row: coalesced), those of the active contacts' rows also staged in compact order for the solve;
  //  * the efc_J leaf, efc_frictionloss and the constant contact leaves NOT stored here: cs_deferred_stores() writes them behind the solve, where nothing waits for them
  //    (vmcnt is in order: 11 KB of row stores per environment in front of the solve's qM reads made every wave wait for its own stores to land).
  // Arithmetic and operation order per value as in make_constraint().
  template <int NMAX>
  __device__ __forceinline__ void make_constraint_cs(Sol2Pre<REAL, NMAX>& pre, Sol2Con<REAL>& C) {
    const int l = lane_here();
#if MJH_CS_PF >= 2
    {  // (the limit rows read the normalised qpos from the prefetched registers: that part comes first in these variants)
      const REAL* gq = KA.cur.qpos + e * M.nq;
#pragma unroll
      for (int j = 0; j < 2; j++) pre.qp[j] = l + W * j < M.nq ? gq[l + W * j] : (REAL)0;
    }
#endif
    const int nv = M.nv, nefc = M.nefc, nl = M.nl, ncon = M.ncon, rows = M.con_rows;
    const bool elliptic = M.cone == CONE_ELLIPTIC;
    if (!KA.state_from_cur && KA.do_step) for (int i = l; i < nv; i += W) S.qvel()[i] = checked(S.qvel()[i], (REAL)0);  // _check_state (same lane wrote it)
    // ---- limit rows: _instantiate_limit_slide_hinge :338-372, lane r0 <-> row r0 ----
    REAL lpos = 0, linvw = 0;
    int lj = 0;
    C.jl = 0; C.ldof = 0;
    if (l < nl) {
      lj = M.lim_jnt[l];
      C.ldof = M.jnt_dofadr[lj];
    }
    {
      const int qa = l < nl ? M.jnt_qposadr[lj] : 0;
      // the normalised qpos of this pass sits one element per lane in the prefetched registers: element qa comes across the lanes of this environment's group
      const int srcl = (int)(lane_id() & ~(W - 1)) + (qa & (W - 1));
      const REAL q0 = __shfl(pre.qp[0], srcl, MJH_WAVE), q1 = __shfl(pre.qp[1], srcl, MJH_WAVE);
      const REAL q = qa < W ? q0 : q1;
      if (l < nl) {
        const REAL dist_min = q - M.jnt_range[2 * lj], dist_max = M.jnt_range[2 * lj + 1] - q;
        const REAL val = (REAL)(dist_min < dist_max) * 2 - 1;
        const REAL pos = (dist_min < dist_max ? dist_min : dist_max) - M.jnt_margin[lj];
        const REAL active = (REAL)(pos < 0);
        C.jl = val * active;
        lpos = pos * active;
        linvw = M.dof_invweight0[C.ldof];
      }
    }
    // ---- active contacts: compact list and contact -> compact slot ----
    int* const act_list = reinterpret_cast<int*>(S.i_con_act());
    int* const slot = reinterpret_cast<int*>(S.i_crow_act());
    int nact = 0;
    for (int base = 0; base < ncon; base += W) {
      const int c = base + l;
      const bool valid = c < ncon;
      const bool act = valid && (S.con_dist()[valid ? c : 0] - M.con_includemargin[valid ? c : 0]) < 0;
      int tot;
      const int at = sub_prefix_count<W>(act, tot) + nact;
      if (act) act_list[at] = c;
      if (valid) slot[c] = act ? at : -1;
      nact += tot;
    }
    C.nact = nact;
    C.nda = nact * rows;
    wave_sync();
    STAMP(23);
    // ---- rows of the active contacts, compact: _instantiate_contact_* :409-583 ----
    REAL* const Jc = S.efc_Jc();  // (the geom frames under it are dead: the narrow phase is over)
    for (int w = l; w < nact * nv; w += W) {
      int a, d;
      split_index(w, nv, M.inv_nv, a, d);
      const int c = act_list[a];
      const int dim = M.con_dim[c];
      const int* cb = M.con_body + 4 * c;
      const int root1 = cb[2], root2 = cb[3];
      const REAL on1 = (REAL)((M.con_dmask[2 * c] >> d) & 1ull), on2 = (REAL)((M.con_dmask[2 * c + 1] >> d) & 1ull);
      const REAL* fr = S.con_frame() + 9 * c;
      const REAL* cpos = S.con_pos() + 3 * c;
      const REAL* fric = M.con_friction + 5 * c;
      REAL jp1[3], jr1[3], jp2[3], jr2[3];
      jac_dof_root(cpos, root2, on2, d, jp2, jr2);
      jac_dof_root(cpos, root1, on1, d, jp1, jr1);
      const REAL dp[3] = {jp2[0] - jp1[0], jp2[1] - jp1[1], jp2[2] - jp1[2]};
      const REAL dr[3] = {jr2[0] - jr1[0], jr2[1] - jr1[1], jr2[2] - jr1[2]};
      REAL diff[6];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        diff[r] = fr[3 * r] * dp[0] + fr[3 * r + 1] * dp[1] + fr[3 * r + 2] * dp[2];
        diff[3 + r] = fr[3 * r] * dr[0] + fr[3 * r + 1] * dr[1] + fr[3 * r + 2] * dr[2];
      }
      REAL* const dst = Jc + (a * rows) * nv + d;
      if (dim == 1) {
        dst[0] = diff[0];
      } else if (!elliptic) {  // _instantiate_contact_pyramidal :454-516
        const int nedge = 2 * (dim - 1);
        for (int ed = 0; ed < nedge; ed++) {
          const REAL f = fric[ed >> 1] * ((ed & 1) ? (REAL)-1 : (REAL)1);
          dst[ed * nv] = diff[0] + diff[1 + (ed >> 1)] * f;
        }
      } else {  // _instantiate_contact_elliptic :519-583
        for (int r = 0; r < dim; r++) dst[r * nv] = diff[r];
      }
    }
#if MJH_CS_PF == 2
    sol2_prefetch<NMAX>(pre);
#endif
    wave_sync();
    STAMP(25);
    // ---- efc_aref / efc_D of every row (:683-693): lane idx <-> Data row ----
    C.Dl = 0; C.arl = 0;
    for (int r = l; r < nefc; r += W) {
      REAL solref[2], solimp[5];
      REAL pos = 0, pos_norm = 0, invweight = 0, jv = 0;
      int crow = -1;
      if (r < nl) {
        pos = lpos; pos_norm = lpos; invweight = linvw;
        solref[0] = M.jnt_solref[2 * lj]; solref[1] = M.jnt_solref[2 * lj + 1];
#pragma unroll
        for (int i = 0; i < 5; i++) solimp[i] = M.jnt_solimp[5 * lj + i];
        jv = 0 + C.jl * S.qvel()[C.ldof];
      } else {
        const int q = r - nl, ncr = M.ncrow;
        const int info = M.crow_info[q];
        const REAL* P = M.crow_par + q;
        solref[0] = P[0]; solref[1] = P[ncr];
#pragma unroll
        for (int i = 0; i < 5; i++) solimp[i] = P[(2 + i) * ncr];
        invweight = P[7 * ncr];
        const int c = info & 0xffff, sub = (info >> 16) & 0xff;
        const REAL dist = S.con_dist()[c] - P[8 * ncr];
        const REAL active = (REAL)(dist < 0);
        if (!(info >> 24)) { pos = dist * active; pos_norm = dist * active; }
        else { pos = (sub == 0 ? dist : (REAL)0) * active; pos_norm = dist; }
        if (dist < 0) {
          crow = slot[c] * rows + sub;
          jv = dot_seq(Jc + crow * nv, 1, S.qvel(), 1, nv);
        }  // (an inactive contact's row is all zeros: its product with the finite qvel is 0)
      }
      REAL k, b, imp;
      kbi(solref, solimp, pos_norm, k, b, imp);
      REAL rr = invweight * (1 - imp) / imp;
      rr = rr > (REAL)MINVAL_CACHED ? rr : (REAL)MINVAL_CACHED;
      const REAL aref_r = -b * jv - k * imp * pos, D_r = 1 / rr;
      if (out.efc_aref) MJH_NT_STORE(aref_r, &out.efc_aref[e * nefc + r]);  // (fused kernels: the solve takes both from the arena / registers)
      if (out.efc_D) MJH_NT_STORE(D_r, &out.efc_D[e * nefc + r]);
      if (r < nl) { C.Dl = D_r; C.arl = aref_r; }
      else if (crow >= 0) { S.efc_aref()[crow] = aref_r; S.efc_D()[crow] = D_r; }  // (over subtree_com / cdof, which nobody reads any more)
    }
    STAMP(26);
  }
  // the stores the fused kernel keeps for the end: efc_J (single-column rows + the contact rows, the inactive ones as zeros), efc_frictionloss, the constant contact leaves
  __device__ __forceinline__ void cs_deferred_stores(const Sol2Con<REAL>& C) {
    const int l = lane_here();
    const int nv = M.nv, nefc = M.nefc, nl = M.nl, nd = nefc - nl, rows = M.con_rows;
    rebind();
    if (out.efc_J) {
      REAL* gJ = out.efc_J + e * nefc * nv;
      const bool dof = l < nv;  // nv <= W: one Data row per pass of the lanes, lane d <-> column d (no index arithmetic per element)
      // single-column rows: lane r holds row r's entry and column
      for (int r = 0; r < nl; r++) {
        const int col = sub_read<W>(C.ldof, r);
        const REAL v = sub_read<W>(C.jl, r);
        if (dof) MJH_NT_STORE((l == col) ? v : (REAL)0, &gJ[r * nv + l]);
      }
      const int* slot = reinterpret_cast<const int*>(S.i_crow_act());
      const REAL* Jc = S.efc_Jc();
      REAL* gC = gJ + nl * nv;
      for (int c = 0; c < M.ncon; c++) {
        const int at = slot[c];  // (same address in every lane of the environment: a broadcast read)
        for (int sub = 0; sub < rows; sub += 4) {
          REAL v[4];
#pragma unroll
          for (int t = 0; t < 4; t++) v[t] = (at >= 0 && sub + t < rows && dof) ? Jc[(at * rows + sub + t) * nv + l] : (REAL)0;
#pragma unroll
          for (int t = 0; t < 4; t++) if (sub + t < rows && dof) MJH_NT_STORE(v[t], &gC[(c * rows + sub + t) * nv + l]);
        }
      }
    }
    if (out.efc_frictionloss) for (int r = l; r < nefc; r += W) MJH_NT_STORE((REAL)0, &out.efc_frictionloss[e * nefc + r]);
    if (M.ncon > 0) contact_const_stores();
  }
  // constraint stage + register solver + integrator in ONE kernel (two environments per wavefront): the rows of the active contacts, efc_D / efc_aref and the
  // narrow phase's distances never leave the arena between the two (the leaves are still stored -- nothing waits for them), and the factor rows, qfrc_smooth
  // and the state arrive while the constraint stage computes.  Serves models of the plain constraint phase whose dense rows fit one slot per lane.
  template <int NMAX, int RPL, bool ONE = false, bool HANDOFF = false>
  __device__ __forceinline__ void run_con_sol2() {
    static_assert(W == 32 && !FRIC && !DIRECT, "plain constraint stage, two environments per wavefront");
    Sol2Pre<REAL, NMAX> pre;
    Sol2Con<REAL> con;
    STAMP0();
    collision<NMAX, HANDOFF>(&pre);
#if MJH_CS_PF == 1
    sol2_prefetch<NMAX>(pre);
#endif
    make_constraint_cs<NMAX>(pre, con);
#if MJH_CS_PF == 3
    if (HANDOFF && ho_on_) {  // what the solver reads back of the first half's leaves (factor rows, qfrc_smooth, qpos, act_dot; qM later): its stores landed while the constraint stage ran
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    sol2_prefetch<NMAX>(pre);
#endif
    wave_sync();
    run_sol2<NMAX, RPL, false, true, ONE>(&pre, &con);
  }
  template <bool FLUID, bool FUSED = false, bool DEFER = false>
  __device__ __forceinline__ void run_vel() {
    STAMP0();
    velocity<FLUID, FUSED, DEFER>();
    if (KA.stages & 0x60) actuation<FLUID>();
    velocity_stores<DEFER>();
  }

  // solve, then (when stepping) the integrator: _euler :313-328, or one stage of _rungekutta4 :331-370.
  __device__ __forceinline__ void run_sol() {
    const int l = lane_here();
    const int nq = M.nq, nv = M.nv, na = M.na;
    STAMP0();
    if (KA.fallback_only) {  // second launch behind the iteration-capped register solver: only the environments it flagged (wave-uniform: one environment per wavefront)
      const REAL q0 = __hip_atomic_load(out.qacc + e * nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!mjh_is_bail_mark(q0)) return;
    }
    load_factor_and_accelerate((KA.stages & 0x40) != 0);
    if (!(KA.stages & 0x40)) return;  // forward() with a stage prefix that ends at _acceleration
    STAMP(53);
    if (M.nefc == 0) {
      for (int d = l; d < nv; d += W) S.qacc()[d] = S.qacc_smooth()[d];
      wave_sync();
      put(out.qacc, S.qacc(), nv);
    } else {
      solve();
    }
    integrate_tail();
  }

  // the integrator on the solved accelerations: _euler :313-328, or one stage of _rungekutta4 :331-370.  Reads S.qacc / qpos / qvel /
  // act / act_dot / qfrc_smooth / qfrc_constraint from the arena (both solver kernels leave them there).
  // NLO <= nv <= NHI: what the calling instantiation knows about the model it serves (the register solver: one range of nv per NMAX)
  // rows of qM for the implicit-damping solve of the register solver's tail (16 < nv <= NMAX), as integrate_tail() takes them: lane i holds row i (entries k <= i)
  template <int NMAX>
  __device__ __forceinline__ void load_damping_rows(TriPack<REAL, NMAX>& Hh) {
    const int l = lane_here();
    const int nv = M.nv;
    const REAL* gM = out.qM + e * nv * nv;
#pragma unroll
    for (int k = 0; k < NMAX; k++) Hh.t[k] = (l < nv && k <= l) ? gM[k * nv + l] : (REAL)0;
  }
  template <int NLO = 0, int NHI = (1 << 30)>
  __device__ __forceinline__ void integrate_tail() {
    const int l = lane_here();
    const int nq = M.nq, nv = M.nv, na = M.na;
    if (!KA.do_step) return;
    rebind();
    const REAL dt = M.timestep;
    const REAL time0 = in.time ? in.time[e] : (REAL)0;
    const int rk = KA.rk_stage;
    if (rk < 0) {  // Euler
      const REAL* qacc = S.qacc();
      constexpr bool REGSOL = NHI <= 32;  // called by the register solver (its arena: H2 / chol_col); the LDS solvers pass no bounds and keep H / HL / HL_inv
      if (!(M.disableflags & DSBL_EULERDAMP)) {
        if constexpr (REGSOL && NLO > 16) {
          // 16 < nv <= NHI: M + dt D factorised and substituted in registers.  Lane i takes row i of the (exactly symmetric) qM leaf as column i -- one coalesced
          // load per term -- and the right-hand side is its own element; no packed copy, no n x n image of the factor, none of chol_factor_lds's three barriers
          // per column.  Same operations per element, in the same order, as the LDS path the other solver kernels run.
          TriPack<REAL, NHI> Hh;
          load_damping_rows<NHI>(Hh);
          const REAL dd = l < nv ? M.timestep * M.dof_damping[l] : (REAL)0;
          const REAL rhs = l < nv ? S.qfrc_smooth()[l] + (M.nefc ? S.qfrc_constraint()[l] : (in.qfrc_constraint ? in.qfrc_constraint[e * nv + l] : (REAL)0)) : (REAL)0;
#pragma unroll
          for (int k = 0; k < NHI; k++) Hh.t[k] = (k == l) ? Hh.t[k] + dd : Hh.t[k];
          chol_pack_big<W, REAL, NHI>(Hh, S.chol_col(), nv);
          const REAL x = tri_solve<W, REAL, NHI>(Hh, rhs, nv);
          if (l < nv) S.s_Mgrad()[l] = x;
          wave_sync();
        } else {
        REAL* const Hp = REGSOL ? S.H2() : S.H();
        for (int w = l; w < (nv * (nv + 1)) / 2; w += W) {
          int i, j;
          tri_unpack(w, i, j);
          const REAL mw = out.qM[e * nv * nv + i * nv + j];
          Hp[w] = (i == j) ? mw + M.timestep * M.dof_damping[i] : mw;
        }
        for (int d = l; d < nv; d += W) S.s_grad()[d] = S.qfrc_smooth()[d] + (M.nefc ? S.qfrc_constraint()[d] : (in.qfrc_constraint ? in.qfrc_constraint[e * nv + d] : (REAL)0));
        wave_sync();
        if (NLO <= 16 && (NHI <= 16 || nv <= 16)) {
          if constexpr (NLO <= 16) chol_factor_solve<W, REAL, NLO, (NHI < 16 ? NHI : 16)>(Hp, S.s_grad(), S.s_Mgrad(), nv);
        } else {
          if constexpr (NHI > 16 && !REGSOL) {
            chol_factor<W, REAL, 16, true>(S.H(), S.HL(), nv);
            chol_inv_diag<W, false>(S.HL(), S.HL_inv(), nv);
            wave_sync();
            chol_solve<W, false>(S.HL(), S.HL_inv(), S.s_grad(), S.s_Mgrad(), nv);
          }
        }
        }
        qacc = S.s_Mgrad();
      }
      advance(S.qpos(), S.qvel(), S.act(), time0, S.act_dot(), qacc, nullptr);
      return;
    }
    // RK4: the tableau is a float32 literal tensor up-cast to the data dtype (forward.py:63-70, math.py:34-45)
    const REAL A[3] = {(REAL)0.5, (REAL)0.5, (REAL)1.0};
    const REAL Bt[4] = {(REAL)(float)(1.0 / 6.0), (REAL)(float)(1.0 / 3.0), (REAL)(float)(1.0 / 3.0), (REAL)(float)(1.0 / 6.0)};
    const RkWork<REAL>& RW = KA.W;
    const REAL b = Bt[rk];
    // this stage's state: S.qvel() is kqvel_s, S.qacc() / S.act_dot() are this pass's results
    for (int i = l; i < nv; i += W) {
      const REAL kq = S.qvel()[i], qa = S.qacc()[i];
      if (rk == 0) { RW.qvel0[e * nv + i] = kq; RW.sum_qvel[e * nv + i] = b * kq; RW.sum_qacc[e * nv + i] = b * qa; }
      else { RW.sum_qvel[e * nv + i] = RW.sum_qvel[e * nv + i] + b * kq; RW.sum_qacc[e * nv + i] = RW.sum_qacc[e * nv + i] + b * qa; }
    }
    for (int i = l; i < na; i += W) {
      const REAL ad = S.act_dot()[i];
      if (rk == 0) { RW.act0[e * na + i] = S.act()[i]; RW.sum_actdot[e * na + i] = b * ad; }
      else RW.sum_actdot[e * na + i] = RW.sum_actdot[e * na + i] + b * ad;
    }
    // d_t0's state: the stage-0 normalised qpos lives in the returned Data until the final advance overwrites it
    const REAL* qpos0g = KA.fin.qpos + e * nq;
    for (int i = l; i < nq; i += W) S.tmp_nq()[i] = (rk == 0) ? S.qpos()[i] : qpos0g[i];
    for (int i = l; i < nv; i += W) S.s_pgrad()[i] = (rk == 0) ? S.qvel()[i] : RW.qvel0[e * nv + i];  // qvel0
    for (int i = l; i < na; i += W) S.act()[i] = (rk == 0) ? S.act()[i] : RW.act0[e * na + i];         // act0
    wave_sync();
    if (rk < 3) {  // state of the next stage (forward.py:356-362)
      const REAL a = A[rk];
      for (int i = l; i < nv; i += W) S.tmp_nv2()[i] = a * S.qvel()[i];
      wave_sync();
      for (int i = l; i < nq; i += W) S.qpos()[i] = S.tmp_nq()[i];  // qpos0 (integrate_pos output goes to tmp_nq)
      wave_sync();
      integrate_pos(S.qpos(), S.tmp_nv2(), dt, S.tmp_nq());
      for (int i = l; i < na; i += W) KA.nxt.act[e * na + i] = S.act()[i] + (a * S.act_dot()[i]) * dt;
      for (int i = l; i < nv; i += W) KA.nxt.qvel[e * nv + i] = S.s_pgrad()[i] + (a * S.qacc()[i]) * dt;
      wave_sync();
      row_store<W>(KA.nxt.qpos, S.tmp_nq(), nq, e);
      return;
    }
    // final _advance(d_t0, act_dot_sum, qacc_sum, qvel_sum)
    for (int i = l; i < nv; i += W) { S.s_mv()[i] = RW.sum_qacc[e * nv + i]; S.tmp_nv2()[i] = RW.sum_qvel[e * nv + i]; }
    for (int i = l; i < na; i += W) S.act_dot()[i] = RW.sum_actdot[e * na + i];
    for (int i = l; i < nq; i += W) S.qpos()[i] = S.tmp_nq()[i];
    wave_sync();
    advance(S.qpos(), S.s_pgrad(), S.act(), time0, S.act_dot(), S.s_mv(), S.tmp_nv2());
  }

  // =====================================================================================================================================================
  // Register solver (mjh_sol2_kernel): the solver phase of CG models with nv <= NMAX <= 28 dofs, slide / hinge limit rows and contact rows only, TWO
  // ENVIRONMENTS PER WAVEFRONT (32 lanes each).  What the LDS solver above keeps in ~40 small arena arrays lives in registers here:
  //   lane d < nv      every nv-vector (one element per lane), row d and column d of the Cholesky factor (TriReg);
  //   lane r < nl      the single-column limit row r: its one Jacobian entry, D, aref, Jaref, jv, force;
  //   lane l, slot j   the dense (contact) row l + 32 j, j < RPL: D, aref, Jaref, jv, force.
  // LDS holds only the contact rows of efc_J and three small staging vectors (a vector that every lane must read in full is written once and
  // read back as broadcasts), 7.8 KB per humanoid environment instead of 18.6 -- with two environments per wavefront a CU keeps 16 environments
  // in flight, the whole B = 4096 batch in one round (the LDS solver: 8, two rounds).  The arithmetic follows solver.py:244-553 like the LDS
  // solver; summation orders differ (register partials), which the 1e-8 parity bar leaves free.  The two environments of a wavefront branch
  // independently (warm-start pick, line-search and iteration counts): plain divergent control flow, every cross-lane primitive used here
  // (sub_sum / sub_read / sub_any) stays inside a 32-lane half.
  // =====================================================================================================================================================
  template <int NMAX>
  __device__ __forceinline__ REAL tri_solve2(const TriPack<REAL, NMAX>& T, REAL bi) const {
    return tri_solve<W, REAL, NMAX>(T, bi, M.nv);
  }

  template <int NMAX, int RPL, bool NEWTON_ONLY = false, bool CS = false, bool ONE = false>
  __device__ __forceinline__ void run_sol2(const Sol2Pre<REAL, NMAX>* pre = nullptr, const Sol2Con<REAL>* con = nullptr) {
    static_assert(W == 32 || W == 16, "two or four environments per wavefront");
    static_assert(!CS || (W == 32 && !NEWTON_ONLY), "the fused constraint + solver kernel runs two environments per wavefront");
    constexpr bool NEWT = NMAX <= 16;  // the Newton direction needs H = M + J^T D J factorised per iteration: register Cholesky, n <= 16 (math.py:84)
    const int l = lane_here();
    const int nq = M.nq, nv = M.nv, na = M.na, nefc = M.nefc, nl = M.nl, nd = nefc - nl;
    const bool dof = l < nv, lim = l < nl;
    const bool solving = (KA.stages & 0x40) != 0;
    const bool from_in = !KA.state_from_cur;
    const bool newton = NEWTON_ONLY ? NEWT : (NEWT && M.solver == SOL_NEWTON);  // (NEWTON_ONLY: a constant -- the CG state, the factor of M past the first solve and the Polak-Ribiere pair, is compiled out)
    STAMP0();
    if (KA.row_lo >= 0 && !KA.scan_marks) {
      // Second (full-width) tier: almost every environment was served by the first launch.  Count the rows of the active contacts before
      // anything else is requested and leave -- the full prologue (factor rows, state, limit rows) cost the ant 21 us per launch for nothing.
      int rows_active = 0;
      if (solving && nefc > 0) {
        const int ncon = M.ncon;
        const bool elliptic = M.cone == CONE_ELLIPTIC;
        for (int base = 0; base < ncon; base += W) {
          const int c = base + l;
          int rows = 0;
          if (c < ncon) {
            const int dim = M.con_dim[c];
            const bool act = (out.contact_dist[e * ncon + c] - (M.topk ? out.contact_includemargin[e * ncon + c] : M.con_includemargin[c])) < 0;
            rows = act ? (dim == 1 ? 1 : (elliptic ? dim : 2 * (dim - 1))) : 0;
          }
          rows_active += (int)sub_sum<W>((float)rows);  // small integers: exact in float
        }
      }
      if (!(rows_active > KA.row_lo && rows_active <= KA.row_hi)) return;
    }
    // ---- every global load of the phase, issued before the first wait -------------------------------------------------------------------------
    // (the one table read that ADDRESSES a later load -- the dof of this lane's limit row, for its single Jacobian entry -- goes first: it is back while the rest is being requested)
    int ldof_t = 0;
    if constexpr (!CS) if (solving && nefc > 0 && lim) ldof_t = M.lim_dof[l];
    const REAL f = CS ? pre->f : ((dof && out.qfrc_smooth) ? out.qfrc_smooth[e * nv + l] : (REAL)0);              // qfrc_smooth
    TriPack<REAL, NMAX> T;
    if constexpr (CS) {
#pragma unroll
      for (int k = 0; k < NMAX; k++) T.t[k] = pre->T.t[k];
    } else {
      // (staging the leaf through LDS with contiguous loads was measured: the strided section shrinks 21 k -> 3 k cycles but the wait only moves
      // to the next load -- this phase's loads are bound by the bytes, 79 MB at B = 4096 -- and the extra registers cost the float32 tier its third wave)
      const REAL* gL = out.qLD + e * nv * nv;
#pragma unroll
      for (int k = 0; k < NMAX; k++) T.t[k] = (dof && k < nv) ? (k <= l ? gL[l * nv + k] : gL[k * nv + l]) : (REAL)0;
    }
    // (MROW for the one-iteration instantiations of the big CG models too -- to save the two passes over qM through L2 -- was tried in round 5: 153 VGPRs -> 256 + 324 B of scratch)
    constexpr bool MROW = NEWT;
    REAL mrow[MROW ? NMAX : 1];  // row d of qM (Newton models: nv <= 16): M products and the Hessian start from registers
    if (MROW) {  // qM is exactly symmetric: row d is read as column d, one coalesced load per term
      const REAL* gM = out.qM + e * nv * nv;
#pragma unroll
      for (int k = 0; k < (MROW ? NMAX : 1); k++) mrow[k] = (dof && k < nv && solving && nefc > 0) ? gM[k * nv + l] : (REAL)0;
    }
    STAMP(80);
    constexpr int NQS = 64 / W;  // qpos slots per lane (nq <= 64)
    REAL qp[NQS], qv = 0, ac = 0, ad = 0, warm = 0;
#pragma unroll
    for (int j = 0; j < NQS; j++) qp[j] = 0;
    REAL Dl = 0, arl = 0, Jl = 0, Dd[RPL], ard[RPL];
    int ldof = 0, limrow = -1;
    int nda = 0;  // dense rows of the ACTIVE contacts of this environment (the rows of inactive contacts are exact zeros throughout, see Env::nrow_)
    const REAL* hsp = nullptr;  // this environment's hand-over from the constraint phase (KArgs::hs), when there is one
#pragma unroll
    for (int j = 0; j < RPL; j++) { Dd[j] = 0; ard[j] = 0; }
    int* rsrc = reinterpret_cast<int*>(S.r_src());  // compact dense row -> Data row
    unsigned short* rdst = reinterpret_cast<unsigned short*>(S.r_dst());  // Data row - nl -> compact dense row, 0xffff = inactive
    const int ndc = nd < KA.row_hi ? nd : KA.row_hi;  // dense rows this tier's arena keeps (r_src, r_fs and efc_Jc are carved for them)
    if (solving) {
      if constexpr (CS) {  // the constraint stage left qvel (checked) in the arena; the rest was requested at its head
#pragma unroll
        for (int j = 0; j < NQS; j++) qp[j] = pre->qp[j];
        if (dof) qv = S.qvel()[l];
        ac = pre->ac; ad = pre->ad;
      } else {
      const REAL* gq = KA.cur.qpos + e * nq;
#pragma unroll
      for (int j = 0; j < NQS; j++) qp[j] = l + W * j < nq ? gq[l + W * j] : (REAL)0;
      if (dof) qv = (from_in ? in.qvel : KA.cur.qvel)[e * nv + l];
      if (l < na) { const REAL* ga = KA.state_from_cur ? KA.cur.act : in.act; ac = ga ? ga[e * na + l] : (REAL)0; ad = out.act_dot ? out.act_dot[e * na + l] : (REAL)0; }
      }
      // (the check of qvel and the copies of the state into the arena wait for these loads -- and, vmcnt being in order, for every load requested before them: they
      // follow the rest of the requests, below.  Round 5 stamps of the ant's solver launch: 24 k of its 55 k cycles were SIX dependent round trips of this prologue.)
      constexpr int NSLOT = 4;   // contact -> slot entries of the hand-over held in registers (contacts l + W j, j < NSLOT; more contacts than that: the loop below)
      REAL slotv[NSLOT];
      REAL nda_f = 0;
#pragma unroll
      for (int j = 0; j < NSLOT; j++) slotv[j] = -1;
      if constexpr (!CS) if (nefc > 0 && KA.hs && M.crow_by_con) hsp = KA.hs + e * KA.hs_reals;
      if (nefc > 0) {
        const REAL* gJ = out.efc_J + e * nefc * nv;
        if constexpr (CS) {
          warm = pre->warm;
          if (lim) { ldof = con->ldof; Dl = con->Dl; arl = con->arl; Jl = con->jl; }
        } else {
        if (dof && KA.warm_src) warm = KA.warm_src[e * nv + l];
        if (lim) { Dl = out.efc_D[e * nefc + l]; arl = out.efc_aref[e * nefc + l]; }
        if (hsp) {
          // The constraint phase handed the active contacts' rows over in compact order (KArgs::hs): every address is fixed, so the whole solver input is ONE round of loads --
          // the count, the contact -> slot table (for the Data-order efc_force store at the end), D / aref of this lane's rows; the rows themselves follow behind the tier test.
          const int ncon = M.ncon;
          nda_f = hsp[0];
#pragma unroll
          for (int j = 0; j < RPL; j++) {
            const int r = l + W * j;
            if (r < ndc) { Dd[j] = hsp[1 + ncon + r]; ard[j] = hsp[1 + ncon + nd + r]; }
          }
#pragma unroll
          for (int j = 0; j < NSLOT; j++) { const int c = l + W * j; if (c < ncon) slotv[j] = hsp[1 + c]; }
        }
        if (lim) { ldof = ldof_t; Jl = gJ[l * nv + ldof]; }  // (waits for the table read at the head only)
        }
        if (dof) limrow = M.dof_limrow[2 * l];
      }
      // ---- first uses ----
      if constexpr (!CS) if (from_in && KA.do_step) qv = checked(qv, (REAL)0);  // _check_state
      // the state only feeds the integrator tail: parked in the arena (its own slots, not under the constraint rows)
#pragma unroll
      for (int j = 0; j < NQS; j++) if (l + W * j < nq) S.qpos()[l + W * j] = qp[j];
      if (dof) S.qvel()[l] = qv;
      if (l < na) { S.act()[l] = ac; S.act_dot()[l] = ad; }
      if (nefc > 0) {
        STAMP(81);
        if constexpr (CS) nda = con->nda;  // (the constraint stage built the rows in compact order)
        else if (hsp) {
          const int ncon = M.ncon, rows = M.con_rows;
          nda = (int)nda_f;
#pragma unroll
          for (int j = 0; j < NSLOT; j++) {
            const int c = l + W * j;
            if (c < ncon) {
              const int at = (int)slotv[j];
              const int start = c * rows;
              for (int k = 0; k < rows; k++) rdst[start + k] = at >= 0 ? (unsigned short)(at * rows + k) : (unsigned short)0xffff;
            }
          }
          for (int c = l + W * NSLOT; c < ncon; c += W) {
            const int at = (int)hsp[1 + c];
            const int start = c * rows;
            for (int k = 0; k < rows; k++) rdst[start + k] = at >= 0 ? (unsigned short)(at * rows + k) : (unsigned short)0xffff;
          }
#pragma unroll
          for (int j = 0; j < RPL; j++) if (!(l + W * j < nda)) { Dd[j] = 0; ard[j] = 0; }  // (past the count: whatever an earlier step left there)
        } else
        {  // active contacts -> compact row tables: one contact per lane, exclusive prefix sum of the active contacts' row counts
          const int ncon = M.ncon;
          const bool elliptic = M.cone == CONE_ELLIPTIC;
          for (int base = 0; base < ncon; base += W) {
            const int c = base + l;
            const bool valid = c < ncon;
            int rows = 0, start = 0;
            bool act = false;
            if (valid) {
              const int dim = M.con_dim[c];
              rows = dim == 1 ? 1 : (elliptic ? dim : 2 * (dim - 1));
              start = M.con_efc_address[c] - nl;
              act = (out.contact_dist[e * ncon + c] - (M.topk ? out.contact_includemargin[e * ncon + c] : M.con_includemargin[c])) < 0;
            }
            int excl, tot;
            if (M.con_rows) {  // one condim: the prefix sum of the row counts is a population count times that count
              excl = sub_prefix_count<W>(act, tot) * M.con_rows + nda;
              tot *= M.con_rows;
            } else {
              int x = act ? rows : 0;
              for (int o = 1; o < W; o <<= 1) { const int y = __shfl_up(x, o, W); if (l >= o) x += y; }
              excl = x - (act ? rows : 0) + nda;
              tot = sub_read<W>(x, W - 1);
            }
            for (int k = 0; k < rows; k++) {
              rdst[start + k] = act ? (unsigned short)(excl + k) : (unsigned short)0xffff;
              if (act && excl + k < ndc) rsrc[excl + k] = nl + start + k;  // (an environment with more rows than this tier keeps leaves below)
            }
            nda += tot;
          }
        }
      }
    }
    // Tiers: the first launch is the instantiation with ONE row slot per lane (32 dense rows: fewer registers -- three waves per SIMD instead of
    // two -- and none of the eight-slot loops) and serves the environments whose active contacts fit it; a second launch of the full-width
    // instantiation picks up the rest (the ant keeps 4 - 8 of its 60 contacts active: almost none).  Each environment is integrated by exactly one.
    STAMP(82);
    if (!(nda > KA.row_lo && nda <= KA.row_hi)) {
      if (((MJH_SOL2_CAPS_ON && KA.it_cap > 0) || KA.mark_leftover) && nda > KA.row_hi && l == 0) out.qacc[e * nv] = mjh_bail_mark((REAL)0);  // the next launch (wider tier / LDS-solver fallback) finds the environment by this mark
      return;
    }
    if (solving) {
      if (nefc > 0) {
        const REAL* gJ = out.efc_J + e * nefc * nv;
        wave_sync();
        if constexpr (CS) {
#pragma unroll
          for (int j = 0; j < RPL; j++) {
            const int r = l + W * j;
            if (r < nda) { Dd[j] = S.efc_D()[r]; ard[j] = S.efc_aref()[r]; }
          }
          STAMP(83);
        } else if (hsp) {
          STAMP(83);
          // the rows of the active contacts, already compact: a contiguous copy, sixteen loads in flight per trip
          const REAL* hJ = hsp + 1 + M.ncon + 2 * nd;
          REAL* dJ = S.efc_Jc();
          const int n = nda * nv;
          for (int i0 = 0; i0 < n; i0 += 16 * W) {
            REAL v[16];
#pragma unroll
            for (int t = 0; t < 16; t++) { const int i = i0 + t * W + l; v[t] = i < n ? hJ[i] : (REAL)0; }
#pragma unroll
            for (int t = 0; t < 16; t++) { const int i = i0 + t * W + l; if (i < n) dJ[i] = v[t]; }
          }
        } else {
#pragma unroll
        for (int j = 0; j < RPL; j++) {
          const int r = l + W * j;
          if (r < nda) { const int x = rsrc[r]; Dd[j] = out.efc_D[e * nefc + x]; ard[j] = out.efc_aref[e * nefc + x]; }
        }
        STAMP(83);
        // rows of the active contacts of efc_J -> LDS
        gather_rows<16>(S.efc_Jc(), gJ, rsrc, nda * nv);
        }
      }
    }
    STAMP(70);
    {  // T.t[l] with a lane-dependent index would spill the triangle: pick the diagonal with a compile-time scan instead
      REAL dg = 1;
#pragma unroll
      for (int k = 0; k < NMAX; k++) dg = (k == l) ? T.t[k] : dg;
      T.inv = dof ? 1 / dg : (REAL)0;
    }
    // ---- _acceleration: qacc_smooth = M^-1 qfrc_smooth (forward.py:222-228) -----------------------------------------------------------------------
    const REAL qs = tri_solve2<NMAX>(T, f);
    if (dof && out.qacc_smooth) out.qacc_smooth[e * nv + l] = qs;
    STAMP(71);
    if (!solving) return;
    REAL qacc = qs, qfrc = 0;
    if (nefc > 0) {
      const REAL* gM = out.qM + e * nv * nv;
      const REAL scale = (REAL)(M.meaninertia * (double)(nv > 1 ? nv : 1));
      const bool fixed = flags & MJH_FLAG_FIXED_ITERATIONS;
      REAL* vs = S.r_vs();
      REAL* vs2 = S.r_vs2();
      REAL* fs = S.r_fs();
      const REAL* Jc = S.efc_Jc();
      // o1 = M a, o2 = M b for vectors staged in LDS (lane d: element d).  Newton models: row d of qM is in registers; otherwise qM is
      // read from L2 -- it is symmetric, so row d is read as column d, coalesced -- nine rows in flight per trip.  Terms in column order.
      auto mul_M2 = [&](const REAL* a, const REAL* b, REAL& o1, REAL& o2, bool two) {
        REAL s1 = 0, s2 = 0;
        if (MROW) {
#pragma unroll
          for (int k = 0; k < (MROW ? NMAX : 1); k++) {  // (no `k < nv` branch around the reads: columns past nv read a clamped address and meet mrow = 0)
            const int kc = k < nv ? k : 0;
            s1 += mrow[k] * a[kc]; s2 += mrow[k] * b[kc];  // callers with one vector pass it twice: the second read is the same address
          }
        } else if (dof) {
          int k = 0;
          for (; k + 9 <= nv; k += 9) {
            REAL m[9];
#pragma unroll
            for (int t = 0; t < 9; t++) m[t] = gM[(k + t) * nv + l];
#pragma unroll
            for (int t = 0; t < 9; t++) { s1 += m[t] * a[k + t]; if (two) s2 += m[t] * b[k + t]; }
          }
          for (; k + 4 <= nv; k += 4) {
            REAL m[4];
#pragma unroll
            for (int t = 0; t < 4; t++) m[t] = gM[(k + t) * nv + l];
#pragma unroll
            for (int t = 0; t < 4; t++) { s1 += m[t] * a[k + t]; if (two) s2 += m[t] * b[k + t]; }
          }
          for (; k < nv; k++) { const REAL m = gM[k * nv + l]; s1 += m * a[k]; if (two) s2 += m * b[k]; }
        }
        o1 = dof ? s1 : (REAL)0; o2 = dof ? s2 : (REAL)0;
      };
      // dense rows: (J a)[r], (J b)[r] for the rows of this lane, terms in column order
      auto mul_J2 = [&](const REAL* a, const REAL* b, REAL (&o1)[RPL], REAL (&o2)[RPL], bool two) {
#pragma unroll
        for (int j = 0; j < RPL; j++) {
          const int r = l + W * j;
          REAL s1 = 0, s2 = 0;
          if (r < nda) {
            const REAL* row = Jc + r * nv;
            int k = 0;
            for (; k + 4 <= nv; k += 4) {
              const REAL j0 = row[k], j1 = row[k + 1], j2 = row[k + 2], j3 = row[k + 3];
              const REAL a0 = a[k], a1 = a[k + 1], a2 = a[k + 2], a3 = a[k + 3];
              s1 += j0 * a0; s1 += j1 * a1; s1 += j2 * a2; s1 += j3 * a3;
              if (two) { s2 += j0 * b[k]; s2 += j1 * b[k + 1]; s2 += j2 * b[k + 2]; s2 += j3 * b[k + 3]; }
            }
            for (; k < nv; k++) { s1 += row[k] * a[k]; if (two) s2 += row[k] * b[k]; }
          }
          o1[j] = s1; o2[j] = s2;
        }
      };
      // _update_constraint :320-357 on register rows: forces, cost; returns (cost sum over rows, gauss term)
      REAL frl = 0, frd[RPL];
      auto constraint_cost = [&](REAL jal, const REAL (&jad)[RPL], REAL Ma, REAL qa, REAL& gauss) -> REAL {
        REAL part = 0;
        {
          const REAL act = (REAL)(lim && jal < 0);
          frl = Dl * -jal * act;
          part += Dl * jal * jal * act;
        }
#pragma unroll
        for (int j = 0; j < RPL; j++) {
          const REAL act = (REAL)((l + W * j < nda) && jad[j] < 0);
          frd[j] = Dd[j] * -jad[j] * act;
          part += Dd[j] * jad[j] * jad[j] * act;
        }
        const REAL gpart = dof ? (Ma - f) * (qa - qs) : (REAL)0;
        const REAL csum = sub_sum<W>(part), g = sub_sum<W>(gpart);
        gauss = (REAL)0.5 * g;
        return ((REAL)0.5 * csum + gauss) + 0;
      };
      // qfrc_constraint = J^T efc_force (rows in index order: the single-column rows first), from the forces in frl / frd
      auto constraint_qfrc = [&]() -> REAL {
#pragma unroll
        for (int j = 0; j < RPL; j++) if (l + W * j < nda) fs[l + W * j] = frd[j];
        if (lim) fs[ndc + l] = Jl * frl;
        wave_sync();
        REAL s = 0;
        if (dof) {
          if (limrow >= 0) s += fs[ndc + limrow];
          int r = 0;
          for (; r + 8 <= nda; r += 8) {
            REAL jj[8], ff[8];
#pragma unroll
            for (int t = 0; t < 8; t++) { jj[t] = Jc[(r + t) * nv + l]; ff[t] = fs[r + t]; }
#pragma unroll
            for (int t = 0; t < 8; t++) s += jj[t] * ff[t];
          }
          for (; r < nda; r++) s += Jc[r * nv + l] * fs[r];
        }
        wave_sync();
        return s;
      };
      // ---- contexts of qacc_smooth and qacc_warmstart in one pass over efc_J and qM (solver.py:293-318, :526-531) ---------------------------------
      const bool warm_on = !(M.disableflags & DSBL_WARMSTART);
      if (dof) { vs[l] = qs; vs2[l] = warm; }
      wave_sync();
      REAL Ma_s, Ma_w, jal_s, jal_w, jad_s[RPL], jad_w[RPL];
      mul_M2(vs, vs2, Ma_s, Ma_w, warm_on);
      mul_J2(vs, vs2, jad_s, jad_w, warm_on);
      {
        const REAL a = lim ? vs[ldof] : (REAL)0, b = lim ? vs2[ldof] : (REAL)0;
        jal_s = Jl * a - arl; jal_w = Jl * b - arl;
      }
#pragma unroll
      for (int j = 0; j < RPL; j++) { jad_s[j] = jad_s[j] - ard[j]; jad_w[j] = jad_w[j] - ard[j]; }
      wave_sync();
      REAL Ma = Ma_s, jal = jal_s, jad[RPL];
#pragma unroll
      for (int j = 0; j < RPL; j++) jad[j] = jad_s[j];
      REAL gauss = 0, cost = 0, prev_cost = (REAL)INFINITY;
      if (warm_on) {
        REAL g_s, g_w;
        const REAL cost_s = constraint_cost(jal_s, jad_s, Ma_s, qs, g_s);
        const REAL cost_w = constraint_cost(jal_w, jad_w, Ma_w, warm, g_w);
        const bool use_warm = cost_w < cost_s;
        if (use_warm) {
          qacc = warm; Ma = Ma_w; jal = jal_w;
#pragma unroll
          for (int j = 0; j < RPL; j++) jad[j] = jad_w[j];
        }
      }
      cost = constraint_cost(jal, jad, Ma, qacc, gauss);  // leaves the forces of the chosen context in frl / frd
      qfrc = constraint_qfrc();
      STAMP(72);
      // M^-1 grad (CG) or H^-1 grad with H = M + J^T diag(D active) J factorised in registers (Newton, solver.py:359-376)
      // Newton, incremental Hessian (round 5): between two iterations of a solve only the rows whose `active` flag flipped change
      // H = M + J^T diag(D active) J (solver.py:359-376).  The un-factored row of H stays in registers (Hraw) with the activity bits of this lane's rows at the last build (hact);
      // later builds add / subtract the flipped rows only (a compact list of them in the arena, the same lane-group accumulation as the full build) and every MJH_INCR_H_PERIOD-th build is
      // a full one again.  Tolerance-level, not bit-level, parity: the sums differ from the reference's full rebuild in their last bits (as the matrix-core build already does).
      // (instantiations for more than 8 dofs only: the 8-dof tier -- the ant's -- sits at its 128-VGPR bound and would spill 80 B more for it: 52 -> 132 B)
      constexpr bool INCR = NEWT && NMAX > 8;
      REAL Hraw[INCR ? NMAX : 1];
      unsigned hact = 0;
      int hbuilds = 0;  // builds of this solve so far (the same in every environment of a wave that is still iterating)
      const bool incr_on = INCR && M.sol2_incr != 0;
      auto precondition = [&](REAL grad) -> REAL {
        if (!newton) return tri_solve2<NMAX>(T, grad);
        TriPack<REAL, NMAX> H;
        unsigned hnow = 0;
        if (INCR) {
#pragma unroll
          for (int j = 0; j < RPL; j++) if (l + W * j < nda && jad[j] < 0) hnow |= 1u << j;
          if (lim && jal < 0) hnow |= 1u << 31;
        }
        const bool incr = incr_on && hbuilds > 0 && (hbuilds % MJH_INCR_H_PERIOD) != 0;
        if (INCR && incr) {
          const unsigned flip = hnow ^ hact;
          int* const flist = reinterpret_cast<int*>(S.r_src());  // (the row-source table is dead once the solver's inputs are loaded)
          int nfl = 0;
#pragma unroll
          for (int j = 0; j < RPL; j++) {
            const int r = l + W * j;
            const bool f = r < nda && ((flip >> j) & 1u);
            if (r < nda) fs[r] = f ? ((hnow >> j) & 1u ? Dd[j] : -Dd[j]) : (REAL)0;  // the row's weight CHANGE
            int tot;
            const int at = sub_prefix_count<W>(f, tot) + nfl;
            if (f) flist[at] = r;
            nfl += tot;
          }
          if (lim) fs[ndc + l] = (flip >> 31) ? ((hnow >> 31) ? (Jl * Dl * (REAL)1) * Jl : -((Jl * Dl * (REAL)1) * Jl)) : (REAL)0;
          wave_sync();
          REAL acc[NMAX];
#pragma unroll
          for (int k = 0; k < NMAX; k++) acc[k] = 0;
          if (dof && limrow >= 0) {
#pragma unroll
            for (int k = 0; k < NMAX; k++) if (k == l) acc[k] += fs[ndc + limrow];
          }
          constexpr int G = NEWT ? W / NMAX : 1;
          const int hg = NEWT ? l / NMAX : 0, hi = NEWT ? l - hg * NMAX : l;
          for (int q = hg; q < nfl; q += G) {
            const int r = flist[q];
            const REAL* jr = Jc + r * nv;
            const REAL ji = (hi < nv ? jr[hi] : (REAL)0) * fs[r] * (REAL)1;
            REAL jk[NMAX];
#pragma unroll
            for (int k = 0; k < NMAX; k++) jk[k] = jr[k < nv ? k : 0];
#pragma unroll
            for (int k = 0; k < NMAX; k++) acc[k] += ji * jk[k];
          }
#pragma unroll
          for (int o = NMAX; o < NMAX * G; o <<= 1) {
#pragma unroll
            for (int k = 0; k < NMAX; k++) acc[k] += __shfl_xor(acc[k], o, W);
          }
#pragma unroll
          for (int k = 0; k < (INCR ? NMAX : 1); k++) { Hraw[k] = (k <= l) ? Hraw[k] + acc[k] : (REAL)0; H.t[k] = Hraw[k]; }
          wave_sync();
        } else {
        // weights of the rows in the quadratic set, staged for broadcast reads (fs is rewritten by the next constraint_qfrc)
#pragma unroll
        for (int j = 0; j < RPL; j++) if (l + W * j < nda) fs[l + W * j] = jad[j] < 0 ? Dd[j] : (REAL)0;
        if (lim) fs[ndc + l] = jal < 0 ? (Jl * Dl * (REAL)1) * Jl : (REAL)0;
        wave_sync();
#pragma unroll
        for (int k = 0; k < NMAX; k++) H.t[k] = (NEWT && k <= l) ? mrow[NEWT ? k : 0] : (REAL)0;
        REAL acc[NMAX];
#pragma unroll
        for (int k = 0; k < NMAX; k++) acc[k] = 0;
        // (measured, MI355X: mesh scene, nv 12, up to 80 rows: solver phase 206 -> 195 us; ant, nv 8, <= 32 rows: 48.3 -> 49.4 us -- eight-dof models keep the vector path)
        constexpr bool MFMA_H = sizeof(REAL) == 4 && NEWT && (NMAX % 4) == 0 && NMAX >= 12;
        bool by_mfma = false;
        if constexpr (MFMA_H) by_mfma = M.sol2_hs != 0;
        if (dof && limrow >= 0 && !by_mfma) {  // a single-column row only touches its own diagonal entry, and it precedes the contact rows
#pragma unroll
          for (int k = 0; k < NMAX; k++) if (k == l) acc[k] += fs[ndc + limrow];
        }
        // The 32 lanes of the environment split into G = 32 / NMAX groups of NMAX: lane (g, i) sums row i of J^T diag(w) J over the rows
        // r = g, g + G, ... (a counted loop whose LDS reads pipeline; a row outside the quadratic set has weight 0 and adds (J * 0) * J = +-0),
        // then the groups' partial sums are added across lanes.  One lane per dof walking every row was a quarter of the lanes doing
        // four times the trips.
        if constexpr (MFMA_H) {
          if (by_mfma) {
            // J^T diag(w) J on the matrix cores: v_mfma_f32_4x4x1 is 16 independent 4 x 4 outer-product accumulations, block b fed by lanes 4 b .. 4 b + 3 -- an
            // environment's W lanes own W / 4 blocks and nothing of another environment's lanes enters them, so the instruction sits in this per-environment
            // (divergent) code like any other (checked on the device: lanes outside EXEC keep their destination registers, tools/micro/mfma_exec.hip).  Block q owns
            // rows 4 q .. 4 q + 3 of H: lane p of the block supplies A[p] = J[r][4 q + p] * w[r] once per row and B[p] = J[r][4 tj + p] for each column tile tj, and
            // receives column 4 tj + p of the block's four rows (register t = row 4 q + t); rows r in order, NMAX / 4 instructions per row.  Lane i = 4 q + p needs
            // row i, i.e. register p of every lane of its own block: a 4 x 4 transpose inside each quad, done with quad_perm broadcasts and selects -- nothing goes
            // through LDS (a staging square cost the mesh scene one of its seven workgroups per CU).  Against the vector path (one lane per dof walking every row:
            // 1 + 2 NMAX vector instructions and NMAX + 1 LDS reads per row) this is one multiply, 2 + NMAX / 4 LDS reads and NMAX / 4 matrix instructions per row.  The
            // products are fused multiply-adds here and separate multiplies and adds there: the Hessian differs in its last bits (it only steers the search direction).
            typedef float f4v __attribute__((ext_vector_type(4)));
            constexpr int TS = NMAX / 4;
            const int q = l >> 2, p = l & 3;
            const bool qon = q < TS;
            const int ia = qon ? 4 * q + p : nv, ca = ia < nv ? ia : 0;  // (columns past nv: a clamped address and a zero after the read -- no branch around an LDS read, as in the vector path)
            int cb[TS];
            bool bon[TS];
            f4v hacc[TS];
#pragma unroll
            for (int tj = 0; tj < TS; tj++) { bon[tj] = 4 * tj + p < nv; cb[tj] = bon[tj] ? 4 * tj + p : 0; hacc[tj] = (f4v){0.f, 0.f, 0.f, 0.f}; }
            // MJH_HROWS rows per trip, their reads requested together (a row past nda: row 0 with weight 0); the accumulation stays in row order
            for (int r = 0; r < nda; r += MJH_HROWS) {
              float ja[MJH_HROWS], jb[MJH_HROWS][TS], w[MJH_HROWS];
#pragma unroll
              for (int u = 0; u < MJH_HROWS; u++) {
                const bool row_in = r + u < nda;
                const int rr = row_in ? r + u : 0;
                const REAL* jr = Jc + rr * nv;
                const float wr = (float)fs[rr];
                w[u] = row_in ? wr : 0.f;
                ja[u] = (float)jr[ca];
#pragma unroll
                for (int tj = 0; tj < TS; tj++) jb[u][tj] = (float)jr[cb[tj]];
              }
#pragma unroll
              for (int u = 0; u < MJH_HROWS; u++) {
                const float a = (ia < nv ? ja[u] : 0.f) * w[u] * 1.f;
#pragma unroll
                for (int tj = 0; tj < TS; tj++) hacc[tj] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bon[tj] ? jb[u][tj] : 0.f, hacc[tj], 0, 0, 0);
              }
            }
#define MJH_QUAD_COL(tj, c)                                                                                                                                         \
            {                                                                                                                                                       \
              const float x0 = quad_bcast<c>(hacc[tj][0]), x1 = quad_bcast<c>(hacc[tj][1]), x2 = quad_bcast<c>(hacc[tj][2]), x3 = quad_bcast<c>(hacc[tj][3]);       \
              const float v = p == 0 ? x0 : (p == 1 ? x1 : (p == 2 ? x2 : x3));                                                                                      \
              acc[4 * tj + c] = (dof && 4 * tj + c <= l) ? (REAL)v : (REAL)0;                                                                                        \
            }
#pragma unroll
            for (int tj = 0; tj < TS; tj++) { MJH_QUAD_COL(tj, 0) MJH_QUAD_COL(tj, 1) MJH_QUAD_COL(tj, 2) MJH_QUAD_COL(tj, 3) }
#undef MJH_QUAD_COL
            if (dof && limrow >= 0) {
#pragma unroll
              for (int k = 0; k < NMAX; k++) if (k == l) acc[k] += fs[ndc + limrow];
            }
          }
        }
        if (!by_mfma) {
          constexpr int G = NEWT ? W / NMAX : 1;
          static_assert(G == 1 || (NMAX & (NMAX - 1)) == 0, "several lane groups need a power-of-two NMAX");
          const int hg = NEWT ? l / NMAX : 0, hi = NEWT ? l - hg * NMAX : l;  // (lanes past the last group run the loop on rows nobody reads)
          const REAL* jr = Jc + hg * nv;
#pragma unroll 2
          for (int r = hg; r < nda; r += G, jr += G * nv) {
            // no `k < nv` branches around the reads: a scalar branch per column made every LDS read its own round trip (the compiler waits at each
            // block boundary) -- 14 dependent trips per row on the mesh scene.  Columns past nv read a clamped address; their sums are never read back
            // (only k <= l < nv is), and lanes past nv multiply by ji = 0.
            const REAL ji = (hi < nv ? jr[hi] : (REAL)0) * fs[r] * (REAL)1;
            REAL jk[NMAX];
#pragma unroll
            for (int k = 0; k < NMAX; k++) jk[k] = jr[k < nv ? k : 0];
#pragma unroll
            for (int k = 0; k < NMAX; k++) acc[k] += ji * jk[k];  // only k <= l is read back
          }
#pragma unroll
          for (int o = NMAX; o < NMAX * G; o <<= 1) {
#pragma unroll
            for (int k = 0; k < NMAX; k++) acc[k] += __shfl_xor(acc[k], o, W);
          }
        }
#pragma unroll
        for (int k = 0; k < NMAX; k++) H.t[k] = (k <= l) ? H.t[k] + acc[k] : (REAL)0;
        if (INCR && incr_on) {
#pragma unroll
          for (int k = 0; k < (INCR ? NMAX : 1); k++) Hraw[k] = H.t[k];
        }
        wave_sync();
        }
        hact = hnow; hbuilds++;
        STAMP(73);
        // register Cholesky (math.small_cholesky :117-127, pivots clamped at 1e-12) leaving row AND column l of L in lane l
#pragma unroll
        for (int j = 0; j < NMAX; j++) {
          if (j < nv) {
            const REAL sj = dof_read<W, NMAX>(H.t[j], j);
            const REAL dj = r_sqrt<REAL>(sj > (REAL)1e-12 ? sj : (REAL)1e-12);
            const REAL lij = (l == j) ? dj : ((l > j && dof) ? H.t[j] / dj : (REAL)0);
            if (l >= j) H.t[j] = lij;
#pragma unroll
            for (int k = j + 1; k < NMAX; k++) {
              const REAL lkj = dof_read<W, NMAX>(lij, k);
              H.t[k] = (l == j) ? lkj : ((l > j) ? H.t[k] - lij * lkj : H.t[k]);
            }
          }
        }
        {
          REAL dg = 1;
#pragma unroll
          for (int k = 0; k < NMAX; k++) dg = (k == l) ? H.t[k] : dg;
          H.inv = dof ? 1 / dg : (REAL)0;
        }
        STAMP(74);
        const REAL hx = tri_solve2<NMAX>(H, grad);
        STAMP(75);
        return hx;
      };
      // The preconditioned gradient (M^-1 grad, or the Newton direction: Hessian build + factorisation + substitution) is only ever consumed by the
      // NEXT line search: cond (:501-508) looks at the plain gradient and the cost.  The reference computes it inside _update_gradient at the
      // end of every iteration (and once in _Context.create); here it is computed at the head of an iteration, after cond has decided that
      // there is one -- one Hessian factorisation less per solve (the mesh scene builds 3.4 per solve otherwise), same numbers in every output.
      REAL grad = dof ? (Ma - f) - qfrc : (REAL)0;
      REAL Mgrad = 0, search = 0;
      bool first_dir = true;
      REAL* pg = S.r_pg();  // previous gradient / M^-1 gradient of the Polak-Ribiere step
      int it = 0, niter = 0, ls_total = 0;
      bool bail = false;
      // (iteration caps: measured slower than letting the packed tier finish its solves, profiles/r03/notes.md -- compiled in only with -DMJH_SOL2_CAPS, so the kernels do not carry their bookkeeping)
      const int it_cap = MJH_SOL2_CAPS_ON ? KA.it_cap : 0, ls_cap = MJH_SOL2_CAPS_ON ? KA.ls_cap : 0;
      for (;;) {
        if (ONE || M.iterations == 1) { if (it >= 1) break; }  // (ONE: the launch has checked opt.iterations == 1 -- a compile-time trip count of one)
        else if (fixed) { if (it >= M.iterations) break; }
        else {  // cond :501-508
          const REAL gg = sub_sum<W>(grad * grad);
          const REAL improvement = (prev_cost - cost) / scale;
          const REAL gradient = r_sqrt<REAL>(gg) / scale;
          bool done = niter >= M.iterations;
          done |= improvement < (REAL)M.tolerance;
          done |= gradient < (REAL)M.tolerance;
          if (done) break;
        }
        if (it_cap > 0 && (it >= it_cap || ls_total >= ls_cap)) { bail = true; break; }  // another iteration is due: the fallback launch redoes this solve from its inputs
        const bool need_grad = ONE ? false : !(it + 1 >= M.iterations);
        {  // search direction of this iteration: -H^-1 grad (Newton), -M^-1 grad at the start, Polak-Ribiere afterwards (CG, :519-523)
          const REAL Mg = precondition(grad);
          if (first_dir || newton) {
            search = -Mg;
          } else {
            const REAL pgrad = dof ? pg[l] : (REAL)0, pMgrad = dof ? pg[nv + l] : (REAL)0;  // written by this lane
            const REAL num = sub_sum<W>(grad * (Mg - pMgrad)), den = sub_sum<W>(pgrad * pMgrad);
            REAL beta = num / (den > (REAL)mjMINVAL ? den : (REAL)mjMINVAL);
            beta = beta > 0 ? beta : (REAL)0;
            search = -Mg + beta * search;
          }
          Mgrad = Mg;
          first_dir = false;
        }
        // ---- _linesearch :378-497 -----------------------------------------------------------------------------------------------------------------
        {
          if (dof) vs[l] = search;
          wave_sync();
          REAL mv, unused, jvd[RPL], unused_d[RPL];
          mul_M2(vs, vs, mv, unused, false);
          mul_J2(vs, vs, jvd, unused_d, false);
          const REAL jvl = lim ? Jl * vs[ldof] : (REAL)0;
          wave_sync();
          STAMP(76);
          const REAL ss = sub_sum<W>(dof ? search * search : (REAL)0);
          const bool nz = sub_any<W>(dof && search != 0);
          const REAL snorm = nz ? r_sqrt<REAL>(ss) : (REAL)0;
          const REAL gtol = (REAL)(M.tolerance * M.ls_tolerance) * (snorm * scale);
          const REAL a = sub_sum<W>(dof ? search * Ma : (REAL)0), b = sub_sum<W>(dof ? search * f : (REAL)0), cc = sub_sum<W>(dof ? search * mv : (REAL)0);
          const REAL qg0 = gauss, qg1 = a - b, qg2 = (REAL)0.5 * cc;
          // per-row quadratic coefficients (:386-394)
          const REAL ql0 = ((REAL)0.5 * jal * jal) * Dl, ql1 = (jvl * jal) * Dl, ql2 = ((REAL)0.5 * jvl * jvl) * Dl;
          REAL qd0[RPL], qd1[RPL], qd2[RPL];
#pragma unroll
          for (int j = 0; j < RPL; j++) { qd0[j] = ((REAL)0.5 * jad[j] * jad[j]) * Dd[j]; qd1[j] = (jvd[j] * jad[j]) * Dd[j]; qd2[j] = ((REAL)0.5 * jvd[j] * jvd[j]) * Dd[j]; }
          auto point = [&](REAL alpha) -> LSPoint {  // point_fn :396-422
            REAL q0 = 0, q1 = 0, q2 = 0;
            {
              const REAL x = jal + alpha * jvl;
              const REAL act = (REAL)(lim && x < 0);
              q0 += ql0 * act; q1 += ql1 * act; q2 += ql2 * act;
            }
#pragma unroll
            for (int j = 0; j < RPL; j++) {
              const REAL x = jad[j] + alpha * jvd[j];
              const REAL act = (REAL)((l + W * j < nda) && x < 0);
              q0 += qd0[j] * act; q1 += qd1[j] * act; q2 += qd2[j] * act;
            }
            q0 = sub_sum<W>(q0); q1 = sub_sum<W>(q1); q2 = sub_sum<W>(q2);
            const REAL t0 = (qg0 + q0) + 0, t1 = (qg1 + q1) + 0, t2 = (qg2 + q2) + 0;
            LSPoint p;
            p.alpha = alpha;
            p.cost = alpha * alpha * t2 + alpha * t1 + t0;
            p.d0 = 2 * alpha * t2 + t1;
            p.d1 = 2 * t2 + (REAL)(t2 == 0) * (REAL)mjMINVAL;
            return p;
          };
          const LSPoint p0 = point(0);
          const LSPoint p1 = point(p0.alpha - p0.d0 / p0.d1);
          const bool early = r_abs(p1.d0) < gtol;
          LSPoint lo, hi;
          if (p1.d0 < p0.d0) { hi = p0; lo = p1; } else { hi = p1; lo = p0; }
          bool swap = !early;
          int ls_iter = 0;
          // Wide tiers (RPL >= 4: the mesh scene's 80-row tier, ~10 iterations per search): the three candidates of an iteration are evaluated WITHOUT their cost -- it is only ever read of
          // the two bracket ends the search finishes with (:491-495) -- and carry the two sums the cost needs (t1, t2); the cost of a surviving candidate is formed at the end from the same
          // expression on the same values (one more pass over the rows for the constant term): every output bit-identical, a third of the row work and of the reductions of an iteration less.
          constexpr bool LAZY = RPL >= 4;
          if constexpr (LAZY) {
            struct LP { REAL alpha, cost, d0, d1, t1, t2; bool hc; };
            auto point_nc = [&](REAL alpha) -> LP {
              REAL q1 = 0, q2 = 0;
              {
                const REAL x = jal + alpha * jvl;
                const REAL act = (REAL)(lim && x < 0);
                q1 += ql1 * act; q2 += ql2 * act;
              }
#pragma unroll
              for (int j = 0; j < RPL; j++) {
                const REAL x = jad[j] + alpha * jvd[j];
                const REAL act = (REAL)((l + W * j < nda) && x < 0);
                q1 += qd1[j] * act; q2 += qd2[j] * act;
              }
              q1 = sub_sum<W>(q1); q2 = sub_sum<W>(q2);
              const REAL t1 = (qg1 + q1) + 0, t2 = (qg2 + q2) + 0;
              LP p;
              p.alpha = alpha; p.cost = 0; p.t1 = t1; p.t2 = t2; p.hc = false;
              p.d0 = 2 * alpha * t2 + t1;
              p.d1 = 2 * t2 + (REAL)(t2 == 0) * (REAL)mjMINVAL;
              return p;
            };
            auto cost_of = [&](const LP& p) -> REAL {
              REAL q0 = 0;
              {
                const REAL x = jal + p.alpha * jvl;
                const REAL act = (REAL)(lim && x < 0);
                q0 += ql0 * act;
              }
#pragma unroll
              for (int j = 0; j < RPL; j++) {
                const REAL x = jad[j] + p.alpha * jvd[j];
                const REAL act = (REAL)((l + W * j < nda) && x < 0);
                q0 += qd0[j] * act;
              }
              q0 = sub_sum<W>(q0);
              const REAL t0 = (qg0 + q0) + 0;
              return p.alpha * p.alpha * p.t2 + p.alpha * p.t1 + t0;
            };
            LP lo2{lo.alpha, lo.cost, lo.d0, lo.d1, 0, 0, true}, hi2{hi.alpha, hi.cost, hi.d0, hi.d1, 0, 0, true};
            for (;;) {
              if (fixed) { if (ls_iter >= M.ls_iterations) break; }
              else {
                bool done = ls_iter >= M.ls_iterations;
                done |= !swap;
                done |= (lo2.d0 < 0) && (lo2.d0 > -gtol);
                done |= (hi2.d0 > 0) && (hi2.d0 < gtol);
                if (done) break;
              }
              const LP lo_next = point_nc(lo2.alpha - lo2.d0 / lo2.d1);
              const LP hi_next = point_nc(hi2.alpha - hi2.d0 / hi2.d1);
              const LP mid = point_nc((REAL)0.5 * (lo2.alpha + hi2.alpha));
              const bool nb = (lo2.d0 < 0) == (hi2.d0 < 0);
              const bool s1 = ls_swap(lo2.d0, lo_next.d0, nb); if (s1) lo2 = lo_next;
              const bool s2 = ls_swap(lo2.d0, mid.d0, nb); if (s2) lo2 = mid;
              const bool s3 = ls_swap(lo2.d0, hi_next.d0, nb); if (s3) lo2 = hi_next;
              const bool s4 = ls_swap(hi2.d0, hi_next.d0, nb); if (s4) hi2 = hi_next;
              const bool s5 = ls_swap(hi2.d0, mid.d0, nb); if (s5) hi2 = mid;
              const bool s6 = ls_swap(hi2.d0, lo_next.d0, nb); if (s6) hi2 = lo_next;
              swap = s1 | s2 | s3 | s4 | s5 | s6;
              ls_iter++;
              if (it_cap > 0 && ls_total + ls_iter >= ls_cap) { bail = true; break; }
            }
            if (!bail) {  // (the bracket ends are the same in every lane of the environment: uniform branches around the reductions)
              if (!lo2.hc) lo2.cost = cost_of(lo2);
              if (!hi2.hc) hi2.cost = cost_of(hi2);
            }
            lo.alpha = lo2.alpha; lo.cost = lo2.cost; lo.d0 = lo2.d0; lo.d1 = lo2.d1;
            hi.alpha = hi2.alpha; hi.cost = hi2.cost; hi.d0 = hi2.d0; hi.d1 = hi2.d1;
          } else
          for (;;) {
            if (fixed) { if (ls_iter >= M.ls_iterations) break; }
            else {
              bool done = ls_iter >= M.ls_iterations;
              done |= !swap;
              done |= (lo.d0 < 0) && (lo.d0 > -gtol);
              done |= (hi.d0 > 0) && (hi.d0 < gtol);
              if (done) break;
            }
            const LSPoint lo_next = point(lo.alpha - lo.d0 / lo.d1);
            const LSPoint hi_next = point(hi.alpha - hi.d0 / hi.d1);
            const LSPoint mid = point((REAL)0.5 * (lo.alpha + hi.alpha));
            const bool nb = (lo.d0 < 0) == (hi.d0 < 0);
            const bool s1 = ls_swap(lo.d0, lo_next.d0, nb); if (s1) lo = lo_next;
            const bool s2 = ls_swap(lo.d0, mid.d0, nb); if (s2) lo = mid;
            const bool s3 = ls_swap(lo.d0, hi_next.d0, nb); if (s3) lo = hi_next;
            const bool s4 = ls_swap(hi.d0, hi_next.d0, nb); if (s4) hi = hi_next;
            const bool s5 = ls_swap(hi.d0, mid.d0, nb); if (s5) hi = mid;
            const bool s6 = ls_swap(hi.d0, lo_next.d0, nb); if (s6) hi = lo_next;
            swap = s1 | s2 | s3 | s4 | s5 | s6;
            ls_iter++;
            if (it_cap > 0 && ls_total + ls_iter >= ls_cap) { bail = true; break; }
          }
          ls_total += ls_iter;
          if (bail) break;
          const REAL improved = (REAL)((lo.cost < p0.cost) || (hi.cost < p0.cost));
          const REAL alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
          qacc = qacc + improved * search * alpha;
          Ma = Ma + improved * mv * alpha;
          jal = jal + improved * jvl * alpha;
#pragma unroll
          for (int j = 0; j < RPL; j++) jad[j] = jad[j] + improved * jvd[j] * alpha;
          STAMP(77);
        }
        if (need_grad && !newton && dof) { pg[l] = grad; pg[nv + l] = Mgrad; }
        prev_cost = cost;
        cost = constraint_cost(jal, jad, Ma, qacc, gauss);
        qfrc = constraint_qfrc();
        STAMP(78);
        if (need_grad) grad = dof ? (Ma - f) - qfrc : (REAL)0;  // _update_gradient :359-376, the part cond reads (its preconditioned half: next loop head)
        niter++; it++;
      }
      if (W == 16 && KA.sol_key && l == 0) { const int k = 6 * niter + ls_total; KA.sol_key[e] = k < 63 ? k : 63; }  // ~ the solve's cost: a Newton iteration (Hessian build + factorisation) weighs about six line-search iterations
      if (bail) {  // nothing of this environment's solve is kept: the mark hands it to the fallback launch, which writes every output of the phase
        if (l == 0) out.qacc[e * nv] = mjh_bail_mark((REAL)0);
        return;
      }
      rebind();
      if (newton) {  // the Hessian weights overwrote the staged forces: stage the final ones for the row-order store below
#pragma unroll
        for (int j = 0; j < RPL; j++) if (l + W * j < nda) fs[l + W * j] = frd[j];
        wave_sync();
      }
      if (dof) {
        if (out.qacc) out.qacc[e * nv + l] = qacc;
        if (out.qacc_warmstart) out.qacc_warmstart[e * nv + l] = qacc;
        if (out.qfrc_constraint) out.qfrc_constraint[e * nv + l] = qfrc;
      }
      if (out.efc_force) {  // Data order; the rows of inactive contacts carry exact zeros
        if (lim) MJH_NT_STORE(frl, &out.efc_force[e * nefc + l]);
        if constexpr (CS) {
          const int* slot = reinterpret_cast<const int*>(S.i_crow_act());
          const int rows = M.con_rows;
          const float inv_rows = 1.0f / (float)rows;
          for (int r = l; r < nd; r += W) {
            int c, sub;
            split_index(r, rows, inv_rows, c, sub);
            const int at = slot[c];
            MJH_NT_STORE(at >= 0 ? fs[at * rows + sub] : (REAL)0, &out.efc_force[e * nefc + nl + r]);
          }
        } else
        for (int r = l; r < nd; r += W) { const int q = rdst[r]; MJH_NT_STORE(q != 0xffff ? fs[q] : (REAL)0, &out.efc_force[e * nefc + nl + r]); }
      }
      // (requesting the tail's rows of qM ahead of these stores, into the registers the factor has just left, was measured: 116 -> 240 B of scratch, kernel 78.0 -> 83.6 us)
      if constexpr (CS) cs_deferred_stores(*con);
    } else {
      if (dof && out.qacc) out.qacc[e * nv + l] = qacc;
    }
    STAMP(79);
    if (!KA.do_step) return;
    // ---- integrator tail on arena arrays carved over the (dead) constraint rows ----------------------------------------------------------------
    wave_sync();
    if (dof) { S.qacc()[l] = qacc; S.qfrc_smooth()[l] = f; S.qfrc_constraint()[l] = qfrc; }
    wave_sync();
    integrate_tail<(NMAX == 8 ? 0 : (NMAX <= 16 ? 9 : 17)), NMAX>();  // (this instantiation serves nv in that range: the variants of the implicit-damping solve for other sizes are not compiled in)
  }
};

#undef M
#undef in
#undef out
#undef KA
#undef STAMP
#undef STAMP0

#ifndef MJH_CRB32P_WAVES
#define MJH_CRB32P_WAVES 4  /* packed float32 CRB (two / four environments per wavefront): at four waves per SIMD the ant runs it in 27.1 us (three: 32.2), the mesh scene in 23.0 (20.7) */
#endif
#ifndef MJH_SOL2_T1_WAVES
#define MJH_SOL2_T1_WAVES 4  /* float32 register solver, NMAX = 8, one row slot per lane (the ant's first tier): 128 VGPRs + 80 B of scratch at four waves per SIMD, 94 us; 164 VGPRs at three: 100.7 us */
#endif
#ifndef MJH_SOL2_W16_WAVES
#define MJH_SOL2_W16_WAVES 4  /* ... the same tier at four environments per wavefront */
#endif
#ifndef MJH_CON32D_WAVES
#define MJH_CON32D_WAVES 4  /* float32 small-model constraint phase */
#endif
#ifndef MJH_SOL32_WAVES
#define MJH_SOL32_WAVES 3  /* float32 LDS solver: 168 VGPRs + ~96 B of scratch; measured on the mesh scene: 2 waves (173 VGPRs, no scratch) 374 us, 3 waves 338 us, 4 waves (128 + 156 B) 362 us */
#endif
#ifndef MJH_KV32_WAVES
#define MJH_KV32_WAVES 2  /* float32 fused kinematics + velocity kernel, packed: a lower bound -- the kernel needs ~122 VGPRs since the stage functions stopped sharing hoisted address arithmetic (round 2: 256) and runs four waves per SIMD: the ant's B = 16384 is one round.  Requesting the caller's cold ctrl / applied-force rows at the kernel's head instead of at their use: 129 VGPRs, humanoid -0.7 us, ant +2.4 us: not kept */
#endif
#ifndef MJH_KCV32_WAVES
#define MJH_KCV32_WAVES 4  /* round 5: 137 VGPRs once the crb stage of the four-per-wavefront instantiation stopped compiling the 24 / 28 / 32-row register Cholesky variants it can never run (nv <= 16): 128 + 24 B at four waves per SIMD.  Rounds 1 - 4, when those variants set the allocation: float32 fused kinematics + crb + velocity kernel, packed: 173 VGPRs; at three waves per SIMD (168 + 20 B of scratch) the ant ran it in 99.7 us instead of 95.7, at four (128 + 112 B) in 104.5 */
#endif
#ifndef MJH_KCV2_WAVES
#define MJH_KCV2_WAVES 4  /* two-wave kernel 13 of packed float32 models: two workgroups of two waves per SIMD pair -- the point of the kernel is to be at four waves per SIMD where kernel 13 (169 VGPRs) sits at two */
#endif
#ifndef MJH_CON64_WAVES
#define MJH_CON64_WAVES 4  /* float64 plain constraint phase: 128 VGPRs + ~108 B of scratch buys the fourth wave per SIMD (16 environments per CU) */
#endif
// Occupancy bounds (second launch-bound argument = waves per SIMD the allocator must fit): float64 kernels are left alone --
// capping them at 128 VGPRs was measured 3-11 % slower than ~180-230 VGPRs at 2 waves/SIMD, their spills are twice as wide
// (profiles/r01/notes.md).  The float32 solver, packed kinematics / velocity and CRB kernels sit just above an occupancy
// step (197 / 181 / 185 -> 168 VGPRs = 3 waves, 132 -> 128 = 4 waves) and spill only 5-24 dwords to get under it; with
// 32-64 environments per CU (ant B = 16384, mesh B = 8192) the extra wave in flight is worth +13 % / +12 % end to end.
// The register solver's kernel: two environments per wavefront; an odd last environment leaves the second half of its wave idle.
// WT: lanes per environment (32 or 16); WT = 17 is the four-per-wavefront kernel of NEWTON models (16 lanes, Newton-only code: 128 VGPRs + 116 B of scratch instead of + 160 B
// for the ant's first tier, 227 instead of 243 VGPRs for the mesh scene's; ant 53.6 -> 52.2 us per launch, mesh scene 223.7 -> 219.1 us)
#ifndef MJH_SOL2_F64_WAVES
#define MJH_SOL2_F64_WAVES 1  /* float64 instantiations for <= 16 dofs (the twins of configs 3 / 5): at two waves per SIMD they spilled 80 - 600 B per lane to scratch memory; at ONE the unified register file holds the overflow in AGPRs (24 - 150 of them, no scratch).  Measured (MI355X, solver phase): mesh scene float64 B = 4096 190.6 -> 186.9 us, B = 16384 501 -> 463 us; ant float64 29.3 -> 28.3 / 54.9 -> 53.8 us */
#endif
template <typename REAL, int NMAX, int RPL, int WT>
__global__ void __launch_bounds__(MJH_WAVE, (sizeof(REAL) == 4 && NMAX == 8 && WT < 33 && RPL * ((WT == 17 || WT == 18) ? 16 : WT) == 32) ? (WT != 32 ? MJH_SOL2_W16_WAVES : MJH_SOL2_T1_WAVES) : ((sizeof(REAL) == 8 && NMAX <= 16) ? MJH_SOL2_F64_WAVES : 2)) mjh_sol2_kernel(KArgs<REAL> args) {
  constexpr int W = (WT == 17 || WT == 18) ? 16 : (WT >= 33 ? 32 : WT);
  constexpr bool NEWTON_ONLY = WT == 17 || WT == 18;
  constexpr bool STAGE = WT == 18;  // WT = 18: one RK4 stage of a small Newton model in ONE launch -- kernel 13's stages, the constraint phase and the solver's first tier behind one another (round 6)
  constexpr bool ONE = WT == 35 || WT == 36;             // WT = 35: WT = 33 for models with opt.iterations == 1 (the humanoid benchmark, solver.py:534-535): the solver loop's body is straight-line code -- the factor of M is dead behind the one preconditioning step instead of crossing the line search in 56 VGPRs
  constexpr bool CS = WT >= 33;  // WT = 33: two environments per wavefront, the constraint stage in front of the solve (Env::run_con_sol2)
  constexpr bool ALL = WT == 34 || WT == 36;  // WT = 34 (36: its one-iteration form, as 35 is to 33): ... and kinematics + crb / factor + velocity in front of that: the whole forward pass + integrator of an environment in one kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const KArgs<REAL>& K = kargs<REAL>();
  constexpr int NSUB = MJH_WAVE / W;  // environments per wavefront: two (32 lanes each) or, for nv <= 16, four (16 lanes each)
  const int sub = (int)(threadIdx.x / W);
  REAL* lds = reinterpret_cast<REAL*>(lds_raw) + sub * K.lds_reals;
  if constexpr (RPL > 1 && W == 32) if (K.scan_marks) {  // (only the full-width instantiations ever run as a second tier: the others carry none of this)
    // Second tier behind a first one that marked what it left (mjh_bail_mark in out.qacc): each wave reads the marks of 64 environments with
    // one load and serves the marked ones NSUB at a time.  A launch of one wave per environment pair, each counting its active rows only to
    // leave, cost the ant 11.5 us per RK4 stage for a tier that serves (almost) nobody.
    for (int64_t base = (int64_t)blockIdx.x * MJH_WAVE; base < K.env_count; base += (int64_t)gridDim.x * MJH_WAVE) {
      const int64_t mine = base + threadIdx.x;
      bool marked = false;
      if (mine < K.env_count) marked = mjh_is_bail_mark(__hip_atomic_load(K.cur.qacc + (K.env_begin + mine) * K.M.nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      unsigned long long todo = __ballot(marked);
      while (todo) {
        int pick = -1;
#pragma unroll
        for (int q = 0; q < NSUB; q++) {
          int nxt = -1;
          if (todo) { nxt = __ffsll((long long)todo) - 1; todo &= todo - 1; }
          if (q == sub) pick = nxt;
        }
        if (pick >= 0) {
          Env<REAL, W, false> E(lds, K.env_begin + base + pick, K.flags);
          E.template run_sol2<NMAX, RPL, NEWTON_ONLY>();
        }
        wave_sync();
      }
    }
    return;
  }
  // One workgroup per NSUB environments, no grid-stride loop (the host launches per 2^20 workgroups): nothing is live across the body for a next trip -- with the loop the whole-pass
  // kernel took 256 VGPRs + 48 B of scratch, without it 153 (see mjh_phase_kernel)
  const int64_t blk = blockIdx.x;
  if constexpr (STAGE) {
    // One RK4 stage of a small model used to be three launches (kernel 13: kinematics + crb / factor + velocity at four environments per wavefront; kernel 8: collision +
    // constraint rows at two; the register solver's first tier at four) with a device-wide barrier, a launch and a burst of input loads between each pair -- 45 % of the solver
    // launch's cycles were its input loads (in-kernel stamps, profiles/r06/notes.md).  Here a wavefront takes its four environments through all three parts: the parts are the SAME
    // stage functions on the same leaves and workspace blocks (bit-identical results), what a part reads of the previous one it reads back through this CU's write-through L1 / L2
    // behind a release / acquire pair at workgroup scope (as the first whole-pass kernel of round 4 did), the constraint phase runs its two-per-wavefront code twice (environments
    // 0, 1 then 2, 3 of the wave), and the arena is carved three times.  Environments the first tier leaves are marked for the second tier's launch as before.
    const int64_t idx = blk * NSUB + sub;
    // (the constant contact leaves depend on nothing: for half of the workgroups they leave at the head of the kernel, while the others' go out behind their constraint rows as before)
    const bool consts_early = (K.stage_parts & 1) && (__builtin_popcount((unsigned)blockIdx.x & (unsigned)K.xswap_c) & 1) != 0;
    if (K.stage_parts & 1) {
    if (idx < K.env_count) {
      Env<REAL, 16, false> A(lds, K.env_begin + idx, K.flags);
      if (consts_early && K.M.ncon > 0 && !K.M.topk) A.contact_const_stores();
      A.template run_kin<false, false, true>(); wave_sync();
      // (out of lockstep, as in the whole-pass kernel: workgroups of odd parity under the mask of flags bits 16..27 run the velocity stage before crb / factor)
      if (!((__builtin_popcount((unsigned)blockIdx.x & (((unsigned)K.flags >> 16) & 0xfffu)) & 1) != 0)) { A.template crb_factor<true>(); wave_sync(); A.template run_vel<false, true>(); }
      else { A.template run_vel<false, true>(); wave_sync(); A.template crb_factor<true>(); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    }
    {
      const int sub2 = (int)(threadIdx.x >> 5);
      REAL* lds2 = reinterpret_cast<REAL*>(lds_raw) + sub2 * K.lds_reals2;
      for (int p = 0; p < 2; p++) {  // (uniform trip count)
        const int64_t i2 = blk * NSUB + 2 * p + sub2;
        if (i2 < K.env_count) {
          Env<REAL, 32, false, true> C(lds2, K.env_begin + i2, K.flags);
          C.S.off = &K.off2;
          C.consts_done_ = consts_early;
          C.run_con();
        }
        wave_sync();
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    if (idx < K.env_count) {
      REAL* lds3 = reinterpret_cast<REAL*>(lds_raw) + sub * K.lds_reals3;
      Env<REAL, 16, false> E(lds3, K.env_begin + idx, K.flags);
      E.S.off = &K.off3;
      E.template run_sol2<NMAX, RPL, NEWTON_ONLY>();
    }
  } else
  {
    const int64_t idx = blk * NSUB + sub;
    if (idx < K.env_count) {
      // W = 16: the four environments of a wave run until the slowest has converged.  They are drawn from a list sorted by the previous step's iteration counts
      // (mjh_sort_kernel), so that long solves share waves; which environments share a wave changes nothing in any of them.
      const int64_t env = (W == 16 && K.sol_perm) ? (int64_t)K.sol_perm[idx] : idx;
      Env<REAL, W, false> E(lds, K.env_begin + env, K.flags);
      if constexpr (ALL) {
        // Out of lockstep (round 6).  A batch of one round of waves starts every wave at the same moment on the same program: all of them are in an arithmetic section, then all of them in
        // a store burst (a lone wave runs the kernel in 108 us, 2048 of them in 142; delaying half of the workgroups by 19 us cost 4 us: profiles/r06/notes.md).  crb / factor and the
        // velocity stage both depend on the kinematics only (forward.py:73-99), so the workgroups whose index has odd parity under a mask (flags bits 16..27; host: MJH_XSWAP) run the
        // velocity stage FIRST: the same operations on the same inputs -- every leaf bit-identical -- while half of the chip is in one stage and half in the other.
        const bool swap_cv = (__builtin_popcount((unsigned)blockIdx.x & (((unsigned)K.flags >> 16) & 0xfffu)) & 1) != 0;
        // The two halves of the pass used to be two launches (kernels 13 and 14): a device-wide barrier between them -- the slowest wave of the first, the launch, and
        // every wave of the second requesting its inputs at the same moment.  Here an environment's wave goes straight on: what the second half reads of the first
        // (geom frames, subtree_com, cdof, the factor, qM, qfrc_smooth, the normalised qpos) it reads back from the leaves THIS wave has just stored -- behind a
        // release / acquire pair at the scope of its own CU -- through the arena layout of the second half.
        E.template run_kin<false, true>();
        wave_sync();
        REAL h_qv_early = 0;  // (velocity first: qvel sits in the overlay the crb stage writes over -- lifted before it)
        if (!swap_cv) {
          E.template crb_factor<true, NMAX>();
          wave_sync();
          E.template run_vel<false, true>();
        } else {
          E.template run_vel<false, true>();
          wave_sync();
          { const int l_ = (int)(threadIdx.x & (W - 1)); h_qv_early = l_ < K.M.nv ? E.S.qvel()[l_] : (REAL)0; }
          wave_sync();
          E.template crb_factor<true, NMAX>();
          wave_sync();
        }
        if (K.M.all_handoff) {
          // Round 5: the constraint stage's inputs -- geom frames, qvel, subtree_com, cdof -- cross the seam ON CHIP.  Read back from the leaves (round 4) the stage's first loads queued behind
          // the velocity stage's 25 KB of stores (vmcnt is in order): ~15 k of the environment's 307 k cycles.  The geom frames ride in registers from the kinematics (lane g <-> geom g), the
          // three arena arrays are lifted into registers under the first layout and put down under the second; the release / acquire pair for everything ELSE the second half reads back
          // (factor rows, qfrc_smooth, qpos, qM) sits behind the constraint stage, where the stores have long landed (Env::run_con_sol2).
          const int l = (int)(threadIdx.x & (W - 1));
          const int nv6 = 6 * K.M.nv, nb3 = 3 * K.M.nbody, ng = K.M.ngeom;
          REAL h_cd[6], h_sc[3], h_qv;
#pragma unroll
          for (int k = 0; k < 6; k++) h_cd[k] = l + W * k < nv6 ? E.S.cdof()[l + W * k] : (REAL)0;
#pragma unroll
          for (int k = 0; k < 3; k++) h_sc[k] = l + W * k < nb3 ? E.S.subtree_com()[l + W * k] : (REAL)0;
          h_qv = swap_cv ? h_qv_early : (l < K.M.nv ? E.S.qvel()[l] : (REAL)0);
          wave_sync();
          E.S.off = &K.off2;
#pragma unroll
          for (int k = 0; k < 6; k++) if (l + W * k < nv6) E.S.cdof()[l + W * k] = h_cd[k];
#pragma unroll
          for (int k = 0; k < 3; k++) if (l + W * k < nb3) E.S.subtree_com()[l + W * k] = h_sc[k];
          if (l < K.M.nv) E.S.qvel()[l] = h_qv;
          if (l < ng) {
#pragma unroll
            for (int i = 0; i < 3; i++) E.S.geom_xpos()[3 * l + i] = E.ho_g[i];
#pragma unroll
            for (int i = 0; i < 9; i++) E.S.geom_xmat()[9 * l + i] = E.ho_g[3 + i];
          }
          E.ho_on_ = true;
          wave_sync();
        } else {
        // (workgroup scope = this CU: its waves share ONE write-through L1, so a wave's loads see its own landed stores; agent scope would write back and invalidate the
        //  XCD's L2 on every wave -- measured: 228.9 us against 167 us for the two launches)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        wave_sync();
        E.S.off = &K.off2;
        }
        E.template run_con_sol2<NMAX, RPL, ONE, true>();
      } else
      if constexpr (CS) E.template run_con_sol2<NMAX, RPL, ONE>();
      else E.template run_sol2<NMAX, RPL, NEWTON_ONLY>();
    }
  }
}

template <typename REAL, int PHASE, int W>
__global__ void __launch_bounds__((PHASE == 17 ? 2 * MJH_WAVE : MJH_WAVE), ((sizeof(REAL) == 4 && PHASE == 17 && W < 64) ? MJH_KCV2_WAVES : (sizeof(REAL) == 4 && PHASE == 8) ? MJH_CON32D_WAVES : (sizeof(REAL) == 4 && PHASE == 4) ? MJH_SOL32_WAVES : (sizeof(REAL) == 4 && (PHASE == 6 || ((PHASE == 0 || PHASE == 3) && W < 64))) ? 3 : (sizeof(REAL) == 4 && PHASE == 13 && W < 64) ? MJH_KCV32_WAVES : (sizeof(REAL) == 4 && PHASE == 12 && W < 64) ? MJH_KV32_WAVES : ((sizeof(REAL) == 4 && PHASE == 1) ? (W < 64 ? MJH_CRB32P_WAVES : 4) : ((sizeof(REAL) == 8 && PHASE == 2) ? (W < 64 ? 2 : MJH_CON64_WAVES) : ((sizeof(REAL) == 8 && (PHASE == 12 || PHASE == 13)) ? 2 : ((sizeof(REAL) == 8 && PHASE == 1 && W == 64) ? 4 : 1)))))) mjh_phase_kernel(KArgs<REAL> args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const KArgs<REAL>& K = kargs<REAL>();
  constexpr int NSUB = MJH_WAVE / W;  // environments per wavefront: W lanes each, their own LDS arena each
  const int sub = (W == MJH_WAVE) ? 0 : (int)((threadIdx.x & (MJH_WAVE - 1)) / W);  // (PHASE 17: two wavefronts per workgroup share the arenas of the same NSUB environments)  // folded away for a whole-wave environment: everything stays scalar
  REAL* lds = reinterpret_cast<REAL*>(lds_raw) + sub * K.lds_reals;
  // One workgroup per NSUB environments, no grid-stride loop (the host cuts batches past 2^20 workgroups into several launches): with a loop around the body the optimiser keeps
  // whatever is loop-invariant -- addresses, kernarg-derived values of EVERY stage -- in registers across the whole body (round 5: the whole-pass kernel went 256 VGPRs + 48 B -> 153 without it)
  {
    const int64_t blk = blockIdx.x;  // env_count is a multiple of NSUB and the grid covers it exactly (host)
    Env<REAL, W, PHASE == 6 || PHASE == 7, PHASE == 8> E(lds, K.env_begin + blk * NSUB + sub, K.flags);
    if (PHASE == 0) E.template run_kin<false>();
    else if (PHASE == 1) E.run_crb();
    else if (PHASE == 2 || PHASE == 7 || PHASE == 8) E.run_con();  // 8: plain constraint phase of small models, contact rows straight to the leaf  // 7: constraint phase of models with equality / frictionloss / ball- or tendon-limit rows
    else if (PHASE == 3) E.template run_vel<false>();
    else if (PHASE == 5) E.template run_vel<true>();  // velocity phase of models with fluid forces (density / viscosity / wind)
    else if (PHASE == 13) { E.template run_kin<false>(); wave_sync(); E.template crb_factor<true>(); wave_sync(); E.template run_vel<false, true>(); }  // ... and the crb / factor stage between them (small models: the three stages of one RK4 stage are one launch)
    else if (PHASE == 17) {
      // Kernel 13 on TWO wavefronts per workgroup (small float32 models): crb / factor and velocity both depend on the kinematics only (forward.py:73-99), so behind
      // the kinematics stage the first wave goes on with the velocity stage while a second one -- idle at the barrier until then -- runs crb / factor on cinert / cdof in the shared
      // arena (PH_KCV2: its own arrays are a region of their own).  The serial chain of an environment is KIN + max(VEL, CRB) instead of KIN + CRB + VEL.
      const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      if (role == 0) E.template run_kin<false>();
      wave_sync(); __builtin_amdgcn_s_barrier(); wave_sync();  // (an LDS-only fence on both sides: __syncthreads() would also wait for the kinematics stage's leaf stores to land)
      if (role == 0) E.template run_vel<false, true>();
      else E.template crb_factor<true>();
    }
    else if (PHASE == 12) { E.template run_kin<true>(); wave_sync(); E.template run_vel<false, true, true>(); }  // kinematics + velocity in one launch (the velocity phase needs nothing of CRB / CON)
    else E.run_sol();                                 // 4: solver phase; 6: solver phase of models with dof-frictionloss rows
    wave_sync();
  }
}
