// mjhip.hip -- C ABI of the MI355X-native stepper (see include/mjhip.h) and the host side of the launches.
//
// Host work per call: fill the kernarg struct of each pipeline phase and enqueue it on the caller's stream
// (5 launches per forward pass; 20 for an RK4 step).  No allocation, no synchronisation, no PyTorch types:
// the binding (ctypes) passes raw device pointers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mjh_kernels.h"
#include "mjh_convex.h"
#include "mjh_sensor.h"
#include "mjh_reset.h"
#include "mjh_io.h"
#include "mjh_instances.h"

// the kernels are compiled in their own translation units (mjh_inst.hip, one per build group): this file is the host side only
#define X_(R, P, W) extern template __global__ void mjh_phase_kernel<R, P, W>(KArgs<R>);
#define S_(R, N, RPL, W) extern template __global__ void mjh_sol2_kernel<R, N, RPL, W>(KArgs<R>);
#define C_(R) extern template __global__ void mjh_convex_kernel<R>(KArgs<R>);
#define N_(R) extern template __global__ void mjh_sensor_kernel<R, 0>(KArgs<R>); extern template __global__ void mjh_sensor_kernel<R, 1>(KArgs<R>);
MJH_INST_ALL(X_, S_, C_, N_, double)
MJH_INST_ALL(X_, S_, C_, N_, float)
#undef X_
#undef S_
#undef C_
#undef N_

static thread_local std::string g_err;
static unsigned long long* g_stamps = nullptr;  // diagnostic builds only (mjh_debug_set_stamps)
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

// ---- per-launch timing (mjh_debug_phase_timing): HIP events on the launch stream around every kernel of a call ----------------
#define MJH_TIMING_MAX 96
static struct {
  bool on = false;
  int n = 0;                          // launches recorded by the last call
  int id[MJH_TIMING_MAX];             // 0..8 phase-kernel ids, 9 register solver, 10 convex narrow phase, 11 sensors, 12 / 13 fused kinematics (+ crb) + velocity, 14 fused constraint + register solver
  hipEvent_t ev[MJH_TIMING_MAX + 1];  // ev[i] .. ev[i + 1] brackets launch i
} g_timing;
static inline void timing_begin(hipStream_t s) { if (g_timing.on) { g_timing.n = 0; (void)hipEventRecord(g_timing.ev[0], s); } }
static inline void timing_mark(hipStream_t s, int id) {
  if (g_timing.on && g_timing.n < MJH_TIMING_MAX) { g_timing.id[g_timing.n] = id; g_timing.n++; (void)hipEventRecord(g_timing.ev[g_timing.n], s); }
}
#define HIP_TRY(x)                                                                                         \
  do {                                                                                                     \
    hipError_t e_ = (x);                                                                                   \
    if (e_ != hipSuccess) return fail(-5, std::string(#x) + ": " + hipGetErrorString(e_));                 \
  } while (0)

struct mjhModel {
  int dtype;
  void* blob;          // device allocation holding every table
  size_t blob_bytes;
  LdsOff off[MJH_NARENA];
  int lds_bytes[MJH_NARENA];
  int sol2_nmax = 0, sol2_rpl = 0;         // register solver (mjh_sol2_kernel) instantiation serving this model, 0 = not eligible
  int fuse_kv = 0;                         // kinematics + velocity phases run as ONE kernel (12) from an arena of their own
  LdsOff off_kv;
  int lds_kv = 0;
  int fuse_kcv = 0;                        // ... and the crb / factor stage between them: ONE kernel (13) for the three (models whose crb stage packs like the other two)
  LdsOff off_kcv;
  int lds_kcv = 0;
  LdsOff off_kcv2;                         // ... and of its two-wave form (mjh_phase_kernel<.., 17, W>, timing id 17: PH_KCV2)
  int lds_kcv2 = 0;
  int kcv2 = 0;                            // a step launches the two-wave form instead of kernel 13 ...
  int64_t kcv2_max_envs = 0;               // ... forced on (MJH_KCV2=1): any batch; otherwise 0 and the limit is kcv2_limit() of the LAUNCH's device
  int kcv2_per_wg = 0, kcv2_envs_per_wg = 0; // arena bytes and environments of one of its workgroups
  int64_t kcv_max_envs = 0;                // ... while the batch is ONE round of that kernel's waves (it needs more registers than either of its parts: two waves per SIMD)
  int sol2_tiers = 0;                      // 1: a first launch with ONE row slot per lane serves the environments whose active contacts fit 32 dense rows
  int sol2_w16_rpl = 0;                    // > 0: that first launch runs FOUR environments per wavefront (16 lanes each, nv <= 16) with this many row slots per lane
  int sol2_w16_nmax = 0;                   // ... instantiated for 8, 12 or 16 dofs
  int sol2_it_cap = 0, sol2_ls_cap = 0;    // > 0: that launch leaves long solves (Newton iterations / line-search iterations beyond the caps) to a fallback launch of the LDS solver
  LdsOff off_tier;                         // ... from an arena of its own (32 rows of efc_J instead of all of them)
  int lds_tier = 0;
  int fuse_all = 0;                        // ... and the whole pass as ONE kernel (mjh_sol2_kernel<.., 34>, timing id 16): kernel 13's stages in front of kernel 14's, one arena of max(lds_kcv, lds_cs)
  int lds_all = 0;
  int fuse_stage = 0;                      // RK4 stages 1..3 of a small Newton model run as ONE launch each (mjh_sol2_kernel<.., 18>, timing id 18): kernel 13's stages, the constraint phase (kernel 8) and the register solver's first tier
  int lds_stage = 0;                       // ... dynamic LDS of one of its four-environment workgroups
  int fuse_tail = 0;                       // the tail of a pass -- constraint phase + the solver's first tier + integrator -- as ONE launch of the same kernel (parts 2 | 4): small Newton models whose pass cannot be one launch (convex narrow phase or sensors between the parts, Euler)
  int lds_tail = 0;
  int fuse_cs = 0;                         // constraint stage + register solver + integrator run as ONE kernel (mjh_sol2_kernel<.., 33>, timing id 14) from an arena of its own
  LdsOff off_cs;
  int lds_cs = 0;
  int pack2[MJH_NPHASE];                   // phase runs two environments per wavefront
  int pack4[MJH_NPHASE];                   // ... or four (16 lanes each): small models only
  // hipGraph replay: the launch sequence of a (buffers, batch, flags) combination is captured once on a private stream and
  // replayed with one hipGraphLaunch on the caller's stream -- a step is 6 launches (24 with RK4) with ~3.6 KB of kernel
  // arguments each, which costs more host time than a small batch takes on the device
  struct GraphEntry { unsigned long long key; hipGraphExec_t exec; unsigned long long last_use; };
  mutable std::vector<GraphEntry> graphs;
  mutable std::mutex graph_mutex;
  mutable hipStream_t capture_stream = nullptr;
  mutable unsigned long long graph_clock = 0;
  mutable std::mutex split_mutex;          // MJH_SPLIT: internal streams the slices of a batch run on, fork / join events
  mutable bool split_ready = false;
  mutable hipStream_t split_stream[4] = {nullptr, nullptr, nullptr, nullptr};
  mutable hipEvent_t split_done[4] = {nullptr, nullptr, nullptr, nullptr};
  mutable hipEvent_t split_fork = nullptr;
  mutable std::mutex dag_mutex;            // MJH_DAG=1 (experiment): crb / factor and velocity kernels beside the constraint phase on two internal streams
  mutable bool dag_ready = false;
  mutable hipStream_t dag_stream[2] = {nullptr, nullptr};
  mutable hipEvent_t dag_fork = nullptr, dag_done[2] = {nullptr, nullptr};
  // the sensor kernel needs nothing of CRB / CON / SOL: it runs on a stream of its own beside them (forked behind the velocity stage, joined at the end of the pass)
  int cvx_lds_bytes;                       // LDS scratch of one (environment, convex pair) wave
  int64_t work_reals;                      // per-environment REALs of workspace: RK4 stage Data and sums + the convex candidates of max_contact_points (0 for most Euler models)
  int64_t sort_reals = 0;                  // ... of which the register solver's environment list and iteration-count keys (two ints per environment, at the very head)
  int64_t hs_reals = 0;                    // ... of which the constraint phase's hand-over to the register solver (KArgs::hs): small models with one contact condim
  int64_t cand_reals = 0;                  // ... of which the candidate contacts (at the HEAD of the workspace: its first B * cand_reals reals; the RK4 stage Data and sums follow)
  std::vector<int64_t> leaf_count;         // per-env element count of every real Data leaf, ABI order
  DevModel<double> m64;
  DevModel<float> m32;
};

namespace {

struct BlobBuilder {
  std::vector<unsigned char> host;
  size_t add(const void* p, size_t bytes) {
    size_t off = (host.size() + 15) & ~size_t(15);
    host.resize(off + bytes);
    if (bytes) memcpy(host.data() + off, p, bytes);
    return off;
  }
};

std::vector<int64_t> leaf_counts(const mjhModelDesc* m) {
  const int64_t nq = m->nq, nv = m->nv, nu = m->nu, na = m->na, nb = m->nbody, nj = m->njnt, ng = m->ngeom;
  const int64_t ncon = m->ncon, nefc = m->nefc;
  std::vector<int64_t> v;
#define F(n, c) v.push_back(c);
  F(time, 1) F(qpos, nq) F(qvel, nv) F(act, na) F(qacc_warmstart, nv) F(ctrl, nu) F(qfrc_applied, nv)
  F(xfrc_applied, nb * 6) F(mocap_pos, m->nmocap * 3) F(mocap_quat, m->nmocap * 4) F(qacc, nv) F(act_dot, na)
  F(xpos, nb * 3) F(xquat, nb * 4) F(xmat, nb * 9) F(xipos, nb * 3) F(ximat, nb * 9) F(xanchor, nj * 3) F(xaxis, nj * 3)
  F(geom_xpos, ng * 3) F(geom_xmat, ng * 9) F(site_xpos, m->nsite * 3) F(site_xmat, m->nsite * 9)
  F(cam_xpos, m->ncam * 3) F(cam_xmat, m->ncam * 9) F(light_xpos, m->nlight * 3) F(light_xdir, m->nlight * 3)
  F(subtree_com, nb * 3) F(cdof, nv * 6) F(cinert, nb * 10) F(crb, nb * 10) F(ten_length, m->ntendon) F(ten_J, m->ntendon * nv) F(ten_velocity, m->ntendon) F(actuator_length, nu)
  F(actuator_moment, nu * nv) F(qM, nv * nv) F(qLD, nv * nv) F(contact_dist, ncon) F(contact_pos, ncon * 3)
  F(contact_frame, ncon * 9) F(contact_includemargin, ncon) F(contact_friction, ncon * 5) F(contact_solref, ncon * 2)
  F(contact_solreffriction, ncon * 2) F(contact_solimp, ncon * 5) F(sensordata, m->nsensordata) F(efc_J, nefc * nv) F(efc_frictionloss, nefc)
  F(efc_D, nefc) F(efc_aref, nefc) F(efc_force, nefc) F(actuator_velocity, nu) F(cvel, nb * 6) F(cdof_dot, nv * 6)
  F(qfrc_bias, nv) F(qfrc_passive, nv) F(qfrc_gravcomp, nv) F(actuator_force, nu) F(qfrc_actuator, nv) F(qfrc_smooth, nv)
  F(qacc_smooth, nv) F(qfrc_constraint, nv)
#undef F
  return v;
}

// leaves an RK stage pass needs as storage between its phases (everything a forward pass writes except
// frames that no later phase reads)
const char* const kStageLeaves[] = {
    "qpos", "qvel", "act", "qacc_warmstart", "qacc", "act_dot", "xipos", "geom_xpos", "geom_xmat", "subtree_com", "cdof",
    "cinert", "qM", "qLD", "efc_J", "efc_D", "efc_aref", "qfrc_smooth", "qacc_smooth", "qfrc_constraint",
    "contact_dist" /* the solver phase picks the active contacts' rows by it */};
// ... plus, for models with convex pairs, the contact leaves the convex kernel hands to the constraint phase
const char* const kConvexStageLeaves[] = {"contact_dist", "contact_pos", "contact_frame"};
const char* const kEqStageLeaves[] = {"xpos", "xquat", "xmat"};  // body frames read by the equality rows (constraint.py:116-212)
bool is_stage_leaf(const char* name, bool has_convex, bool has_fluid, bool has_eq, bool topk) {
  if (topk && !strcmp(name, "contact_includemargin")) return true;  // per environment once max_contact_points selects: the solver phase reads it
  if (!strcmp(name, "qfrc_gravcomp") || !strncmp(name, "ten_", 4)) return false;  // tendon quantities are recomputed from qpos where they are used  // staged in LDS inside the velocity phase; the leaf itself is written by stage 0 only
  if (has_eq) for (const char* s : kEqStageLeaves) if (!strcmp(s, name)) return true;
  for (const char* s : kStageLeaves) if (!strcmp(s, name)) return true;
  if (has_convex) for (const char* s : kConvexStageLeaves) if (!strcmp(s, name)) return true;
  if (has_fluid && !strcmp(name, "ximat")) return true;  // the fluid model needs the inertial frames of the stage (passive.py:31-78)
  return false;
}

template <typename REAL>
int build(const mjhModelDesc* d, mjhModel* out, DevModel<REAL>& M) {
  BlobBuilder bb;
  std::vector<std::pair<const void**, size_t>> fix;  // (pointer slot, offset)
  memset(&M, 0, sizeof(M));
#define X(n) M.n = d->n;
  MJH_MODEL_INTS(X)
#undef X
  M.timestep = (REAL)d->timestep;
  M.impratio = (REAL)d->impratio;
  M.density = (REAL)d->density; M.viscosity = (REAL)d->viscosity;
  M.wind[0] = (REAL)d->wind_x; M.wind[1] = (REAL)d->wind_y; M.wind[2] = (REAL)d->wind_z;
  M.magnetic[0] = (REAL)d->magnetic_x; M.magnetic[1] = (REAL)d->magnetic_y; M.magnetic[2] = (REAL)d->magnetic_z;
  M.has_fluid = (d->density > 0) || (d->viscosity > 0) || (d->wind_x != 0) || (d->wind_y != 0) || (d->wind_z != 0);
  M.has_gravcomp = 0;
  M.con_general = (d->nf > 0 || d->nft > 0 || d->ne > 0 || d->nlb > 0 || d->nlt > 0 || d->topk) ? 1 : 0;
  if (d->nft > 0 && d->nf + d->nl + d->nft > 64) return fail(-38, "frictionloss rows beyond the first 64 solver rows are not supported");
  for (int b = 0; b < d->nbody; b++) if (d->body_gravcomp[b] != 0) M.has_gravcomp = 1;
  {
    // measured on MI355X (profiles/r02/notes.md): ant (nv 8) constraint phase 127 -> 112 us, mesh scene 58 -> 54 us; the humanoid (nv 27) is
    // slower that way (43 -> 59 us: every (contact, dof) lane re-reads 27-wide rows through L2) and keeps the dense copy in LDS
    static const int direct_nv = [] { const char* e = getenv("MJH_CON_DIRECT_NV"); return e ? atoi(e) : 16; }();
    M.con_direct = (d->nv <= direct_nv && !(d->nf > 0 || d->nft > 0 || d->ne > 0 || d->nlb > 0 || d->nlt > 0 || d->topk)) ? 1 : 0;
  }
  M.gravity[0] = (REAL)d->gravity_x; M.gravity[1] = (REAL)d->gravity_y; M.gravity[2] = (REAL)d->gravity_z;
  M.meaninertia = d->meaninertia; M.tolerance = d->tolerance; M.ls_tolerance = d->ls_tolerance;
#define X(n) fix.push_back({(const void**)&M.n, bb.add(d->n, sizeof(int32_t) * (size_t)d->len_##n)});
  MJH_MODEL_INT_ARRAYS(X)
#undef X
#define X(n)                                                                                   \
  {                                                                                            \
    std::vector<REAL> tmp((size_t)d->len_##n);                                                 \
    for (size_t i = 0; i < tmp.size(); i++) tmp[i] = (REAL)d->n[i];                            \
    fix.push_back({(const void**)&M.n, bb.add(tmp.data(), sizeof(REAL) * tmp.size())});        \
  }
  MJH_MODEL_REAL_ARRAYS(X)
#undef X
  // ---- derived topology tables ----
  const int nb = d->nbody, nv = d->nv;
  if (nv > 256) return fail(-22, "nv must be <= 256 (inertia-matrix pair table packs two 8-bit dof indices)");
  const int NW = nv > 64 ? (nv + 63) / 64 : 1;  // 64-bit words per dof mask
  M.mask_words = NW;
  M.big = nv > 64 ? 1 : 0;
  {
    static const bool ho_off = [] { const char* e = getenv("MJH_ALL_HANDOFF"); return e && e[0] == '0'; }();
    M.all_handoff = (!ho_off && d->ngeom <= 32 && d->nbody <= 32 && nv <= 32) ? 1 : 0;
  }
  {  // small plain models (the ones that run four environments per wavefront, no optional physics): a leaner arena (DevModel::lds_diet).  MJH_LDS_DIET=0: off
    static const bool diet_off = [] { const char* e = getenv("MJH_LDS_DIET"); return e && e[0] == '0'; }();
    M.lds_diet = (!diet_off && d->nbody <= 16 && d->njnt <= 16 && nv <= 16 && !M.has_fluid && !M.has_gravcomp && d->ntendon == 0) ? 1 : 0;
  }
  if (M.big) M.con_general = 1;  // the general constraint kernel (7) reads multi-word masks; so does the optional-physics velocity kernel (5)
  if (M.big) M.con_direct = 0;
  std::vector<int> depth(nb, 0), sub_end(nb, 0);
  int max_depth = 1;
  for (int b = 1; b < nb; b++) {
    if (d->body_parentid[b] >= b) return fail(-22, "bodies must be ordered parent-before-child");
    depth[b] = depth[d->body_parentid[b]] + 1;
    if (depth[b] > max_depth) max_depth = depth[b];
  }
  if (max_depth > MJH_MAX_DEPTH) return fail(-22, "kinematic tree deeper than MJH_MAX_DEPTH");
  std::vector<int> chain((size_t)nb * max_depth, 0);
  for (int b = 1; b < nb; b++) {
    int c = b;
    for (int k = depth[b] - 1; k >= 0; k--) { chain[(size_t)b * max_depth + k] = c; c = d->body_parentid[c]; }
  }
  for (int b = nb - 1; b >= 0; b--) {
    if (sub_end[b] < b + 1) sub_end[b] = b + 1;
    if (b > 0) { int p = d->body_parentid[b]; if (sub_end[p] < sub_end[b]) sub_end[p] = sub_end[b]; }
  }
  for (int b = 1; b < nb; b++)  // DFS order check: every body in [b, sub_end[b]) must descend from b
    for (int c = b + 1; c < sub_end[b]; c++) {
      int a = c;
      while (a > b) a = d->body_parentid[a];
      if (a != b) return fail(-22, "bodies are not in depth-first order");
    }
  std::vector<unsigned long long> body_dofmask((size_t)nb * NW, 0ull), dof_ancmask((size_t)nv * NW, 0ull);
  auto anc = [&](int i, int j) { return (dof_ancmask[(size_t)i * NW + (j >> 6)] >> (j & 63)) & 1ull; };
  for (int dd = 0; dd < nv; dd++) {
    int a = dd;
    while (a >= 0) { dof_ancmask[(size_t)dd * NW + (a >> 6)] |= 1ull << (a & 63); a = d->dof_parentid[a]; }
  }
  for (int b = 0; b < nb; b++) {
    int a = b;
    while (a > 0) {
      for (int dd = 0; dd < nv; dd++) if (d->dof_bodyid[dd] == a) body_dofmask[(size_t)b * NW + (dd >> 6)] |= 1ull << (dd & 63);
      a = d->body_parentid[a];
    }
  }
  {  // lower-triangle entries (i, j) of qM whose dofs are on one ancestor path (support.make_m :50-80 masks the rest to zero)
    std::vector<int> pairs;
    for (int i = 0; i < nv; i++)
      for (int j = 0; j <= i; j++)
        if (anc(i, j)) pairs.push_back((i << 8) | j);
    M.nqmpair = (int)pairs.size();
    std::vector<int> slot((size_t)nv * nv, -1);
    for (int pk : pairs) {
      const int i = pk >> 8, j = pk & 0xff, sl = (i * (i + 1)) / 2 + j;
      slot[(size_t)i * nv + j] = sl;
      slot[(size_t)j * nv + i] = sl;
    }
    {  // tendon armature couples dofs of different branches: every entry of qM can be non-zero then
      bool arm = false;
      for (int t = 0; t < d->ntendon; t++) arm = arm || d->tendon_armature[t] != 0;
      if (arm)
        for (int i = 0; i < nv; i++)
          for (int j = 0; j <= i; j++) { slot[(size_t)i * nv + j] = (i * (i + 1)) / 2 + j; slot[(size_t)j * nv + i] = (i * (i + 1)) / 2 + j; }
    }
    fix.push_back({(const void**)&M.qm_slot, bb.add(slot.data(), sizeof(int) * slot.size())});
    fix.push_back({(const void**)&M.qm_pair, bb.add(pairs.data(), sizeof(int) * pairs.size())});
  }
  std::vector<int> row_con((size_t)d->nefc, -1);
  for (int c = 0; c < d->ncon; c++) {
    int dim = d->con_dim[c];
    int rows = dim == 1 ? 1 : (d->cone == CONE_ELLIPTIC ? dim : 2 * (dim - 1));
    for (int r = 0; r < rows; r++) {
      int row = d->con_efc_address[c] + r;
      if (row < 0 || row >= d->nefc) return fail(-22, "contact row address out of range");
      row_con[row] = c;
    }
  }
  {  // per-row / per-contact tables of the plain constraint phase (DevModel::crow_par ...).  The arithmetic is the kernel's own, in REAL, in
     // the same order (constraint.py:440-451, 480-487, 547-561): the values are bit-identical to what each lane used to recompute.
    const int first = d->nefc - [&] { int n = 0; for (int r = 0; r < d->nefc; r++) n += row_con[r] >= 0; return n; }();
    int ncrow = d->nefc - first;
    for (int r = first; r < d->nefc; r++) if (row_con[r] < 0) ncrow = -1;  // contact rows must close the row order
    if (ncrow < 0 || d->topk) ncrow = 0;
    M.ncrow = ncrow;
    M.con_rows = 0;
    for (int c = 0; c < d->ncon; c++) {
      const int dim = d->con_dim[c], rows = dim == 1 ? 1 : (d->cone == CONE_ELLIPTIC ? dim : 2 * (dim - 1));
      if (c == 0) M.con_rows = rows; else if (rows != M.con_rows) { M.con_rows = 0; break; }
    }
    M.crow_by_con = (M.con_rows > 0 && ncrow == d->ncon * M.con_rows) ? 1 : 0;
    for (int c = 0; c < d->ncon && M.crow_by_con; c++) if (d->con_efc_address[c] != first + c * M.con_rows) M.crow_by_con = 0;
    const bool elliptic = d->cone == CONE_ELLIPTIC;
    std::vector<REAL> par((size_t)9 * (ncrow > 0 ? ncrow : 1), (REAL)0);
    std::vector<int> info((size_t)(ncrow > 0 ? ncrow : 1), 0);
    for (int q = 0; q < ncrow; q++) {
      const int c = row_con[first + q], sub = first + q - d->con_efc_address[c], dim = d->con_dim[c];
      REAL sr0 = (REAL)d->con_solref[2 * c], sr1 = (REAL)d->con_solref[2 * c + 1];
      if (elliptic && dim > 1 && sub > 0) {
        const REAL sf0 = (REAL)d->con_solreffriction[2 * c], sf1 = (REAL)d->con_solreffriction[2 * c + 1];
        const REAL none = (REAL)(!((sf0 != 0) || (sf1 != 0)));
        sr0 = sf0 + sr0 * none; sr1 = sf1 + sr1 * none;
      }
      REAL fric[5];
      for (int i = 0; i < 5; i++) fric[i] = (REAL)d->con_friction[5 * c + i];
      const REAL t = (REAL)d->body_invweight0[d->geom_bodyid[d->con_geom1[c]]] + (REAL)d->body_invweight0[d->geom_bodyid[d->con_geom2[c]]];
      REAL invweight;
      if (dim == 1) invweight = t;
      else if (!elliptic) { const REAL mu = fric[0]; invweight = (t + mu * mu * t) * 2 * mu * mu / M.impratio; }
      else { const REAL iwf = t / M.impratio; invweight = (sub == 0) ? t : (sub == 1 ? iwf : iwf * ((fric[0] * fric[0]) / (fric[sub - 1] * fric[sub - 1]))); }
      par[(size_t)0 * ncrow + q] = sr0; par[(size_t)1 * ncrow + q] = sr1;
      for (int i = 0; i < 5; i++) par[(size_t)(2 + i) * ncrow + q] = (REAL)d->con_solimp[5 * c + i];
      par[(size_t)7 * ncrow + q] = invweight;
      par[(size_t)8 * ncrow + q] = (REAL)d->con_includemargin[c];
      info[q] = c | (sub << 16) | ((elliptic && dim > 1) ? (1 << 24) : 0);
    }
    fix.push_back({(const void**)&M.crow_par, bb.add(par.data(), sizeof(REAL) * par.size())});
    fix.push_back({(const void**)&M.crow_info, bb.add(info.data(), sizeof(int) * info.size())});
    std::vector<int> cbody((size_t)4 * (d->ncon > 0 ? d->ncon : 1), 0);
    std::vector<unsigned long long> cmask((size_t)2 * (d->ncon > 0 ? d->ncon : 1), 0ull);
    for (int c = 0; c < d->ncon; c++) {
      const int b1 = d->geom_bodyid[d->con_geom1[c]], b2 = d->geom_bodyid[d->con_geom2[c]];
      cbody[4 * c] = b1; cbody[4 * c + 1] = b2; cbody[4 * c + 2] = d->body_rootid[b1]; cbody[4 * c + 3] = d->body_rootid[b2];
      cmask[2 * c] = body_dofmask[(size_t)b1 * NW]; cmask[2 * c + 1] = body_dofmask[(size_t)b2 * NW];  // (first word: the table serves the small-model kernel only)
    }
    fix.push_back({(const void**)&M.con_body, bb.add(cbody.data(), sizeof(int) * cbody.size())});
    fix.push_back({(const void**)&M.con_dmask, bb.add(cmask.data(), sizeof(unsigned long long) * cmask.size())});
    // bounding-sphere reach of the sphere / capsule pairs (collision(): the cull ahead of the narrow phase in RK4 stages 1..3)
    std::vector<REAL> pcull((size_t)2 * (d->npair > 0 ? d->npair : 1), (REAL)-1);
    for (int p = 0; p < d->npair; p++) {
      const int fn = d->pair_fn[p], g1 = d->pair_geom1[p], g2 = d->pair_geom2[p];
      if (fn != MJH_FN_SPHERE_SPHERE && fn != MJH_FN_SPHERE_CAPSULE && fn != MJH_FN_CAPSULE_CAPSULE) continue;
      const double r1 = d->geom_size[3 * g1] + (fn == MJH_FN_CAPSULE_CAPSULE ? d->geom_size[3 * g1 + 1] : 0.0);
      const double r2 = d->geom_size[3 * g2] + (fn == MJH_FN_SPHERE_SPHERE ? 0.0 : d->geom_size[3 * g2 + 1]);
      double margin = 0;
      for (int q = 0; q < d->pair_ncon[p] && q < MJH_MAX_PAIR_CONTACTS; q++) margin = std::max(margin, (double)d->con_includemargin[d->pair_dst[p * MJH_MAX_PAIR_CONTACTS + q]]);
      const double reach = (r1 + r2 + margin) * 1.01 + 1e-3;
      pcull[2 * p] = (REAL)(reach * reach); pcull[2 * p + 1] = (REAL)(r1 + r2);
    }
    fix.push_back({(const void**)&M.pair_cull, bb.add(pcull.data(), sizeof(REAL) * pcull.size())});
    static const bool cull_off = [] { const char* e = getenv("MJH_PAIR_CULL"); return e && e[0] == '0'; }();  // MJH_PAIR_CULL=0: RK4 stages 1..3 narrow-phase every pair like stage 0
    M.pair_cull_on = cull_off ? 0 : 1;
  }
  {
    int max_jnt = 1;
    for (int b = 0; b < nb; b++) if (d->body_jntnum[b] > max_jnt) max_jnt = d->body_jntnum[b];
    std::vector<int> chain_dof((size_t)nb * max_depth, 0), chain_jnt((size_t)nb * max_depth * max_jnt, 0);
    for (int b = 1; b < nb; b++)
      for (int k = 0; k < depth[b]; k++) {
        const int c = chain[(size_t)b * max_depth + k];
        chain_dof[(size_t)b * max_depth + k] = d->body_dofadr[c] < 0 ? 0 : (d->body_dofadr[c] | (d->body_dofnum[c] << 16));
        for (int jj = 0; jj < d->body_jntnum[c]; jj++) {
          const int j = d->body_jntadr[c] + jj;
          chain_jnt[((size_t)b * max_depth + k) * max_jnt + jj] = (d->jnt_type[j] + 1) | (d->jnt_dofadr[j] << 8);
        }
      }
    M.max_jnt = max_jnt;
    {  // kinematics by pointer jumping (Env::kinematics): ancestors at distance 2^r and the bodies' start frames
      // Round 6: OPT-IN.  MJH_KIN_JUMP=1: pointer jumping for every tree of two levels or more; unset / 0: the serial walk, whose association of the
      // frame compositions is the reference's (smooth.py:85-113) and the oracle's.  The two forms agree to rounding (1e-13 of each leaf's scale), not bit for bit; as round 5's default
      // for deep trees it bought 2.4 % on the humanoid (142.0 -> 138.7 us) and cost config 2 its parity margin (pre-solver 2.8e-14 -> 6.9e-12, contact_dist element-wise 8.9e-10 against
      // a 1e-8 bound), and a solver run to convergence amplified the last-bit difference past that bound in one stress case -- which round 5 hid behind a gate on opt.iterations, an
      // unrelated solver setting.  The selection is now structural only (depth, nbody, no mocap body with children) and never changes with solver options.
      static const int mode = [] { const char* e = getenv("MJH_KIN_JUMP"); return e ? (e[0] == '0' ? 0 : 1) : 0; }();
      bool mocap_parent = false;  // a mocap body with a child: the reference overrides mocap frames AFTER its scan (smooth.py:85-113), so children hang off the STATIC body_pos /
      //                             body_quat chain; the jump form's anchor / axis pass reads the parent's frame after the override -- such models keep the walk
      for (int b = 1; b < nb; b++) if (d->body_parentid[b] > 0 && d->body_mocapid[d->body_parentid[b]] >= 0) mocap_parent = true;
      const bool on = mode == 1 && max_depth >= 2 && !mocap_parent;
      M.kin_tab = nullptr;
      {  // the level sweep (Env::kinematics): same operations per body as the walk, bit-identical; MJH_KIN_LEVEL=0 keeps the walk (A / B runs)
        static const int lv = [] { const char* e = getenv("MJH_KIN_LEVEL"); return e ? (e[0] != '0') : 1; }();
        M.kin_lvl = (lv && nb > 1 && nb <= 32 && !on) ? 1 : 0;  // (nb <= 32: one body per lane in EVERY instantiation the model can run at, as for the jump form below)
      }
      int R = 0;
      while ((1 << R) < max_depth) R++;
      if (on && nb > 1 && nb <= 32) {  // (one body per lane in EVERY instantiation the model can run at -- 16 lanes per environment only with nbody <= 16, else 32 or 64 -- so that which kernel a batch size or an odd tail selects never changes a bit)
        const size_t anc_bytes = 8 * (((size_t)R * nb + 1) / 2);
        std::vector<unsigned char> tab(anc_bytes + sizeof(REAL) * (size_t)nb * 7, 0);
        int* anc = reinterpret_cast<int*>(tab.data());
        REAL* start = reinterpret_cast<REAL*>(tab.data() + anc_bytes);
        const double wq[4] = {d->body_quat[0], d->body_quat[1], d->body_quat[2], d->body_quat[3]}, wp[3] = {d->body_pos[0], d->body_pos[1], d->body_pos[2]};
        const bool world_identity = wq[0] == 1 && wq[1] == 0 && wq[2] == 0 && wq[3] == 0 && wp[0] == 0 && wp[1] == 0 && wp[2] == 0;
        for (int b = 1; b < nb; b++) {
          for (int r = 0; r < R; r++) { const int k = depth[b] - 1 - (1 << r); anc[(size_t)r * nb + b] = k >= 0 ? chain[(size_t)b * max_depth + k] : 0; }
          for (int i = 0; i < 3; i++) start[(size_t)b * 7 + i] = (REAL)d->body_pos[3 * b + i];
          for (int i = 0; i < 4; i++) start[(size_t)b * 7 + 3 + i] = (REAL)d->body_quat[4 * b + i];
          if (depth[b] == 1 && !world_identity) {  // (never the case for a compiled MuJoCo model; kept equal to the walk's arithmetic, in REAL)
            REAL p[3] = {(REAL)wp[0], (REAL)wp[1], (REAL)wp[2]}, q[4] = {(REAL)wq[0], (REAL)wq[1], (REAL)wq[2], (REAL)wq[3]};
            const REAL bp[3] = {(REAL)d->body_pos[3 * b], (REAL)d->body_pos[3 * b + 1], (REAL)d->body_pos[3 * b + 2]};
            const REAL bq[4] = {(REAL)d->body_quat[4 * b], (REAL)d->body_quat[4 * b + 1], (REAL)d->body_quat[4 * b + 2], (REAL)d->body_quat[4 * b + 3]};
            const REAL sc = q[0], *u = q + 1;
            const REAL uv = u[0] * bp[0] + u[1] * bp[1] + u[2] * bp[2], uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
            const REAL c[3] = {u[1] * bp[2] - u[2] * bp[1], u[2] * bp[0] - u[0] * bp[2], u[0] * bp[1] - u[1] * bp[0]};
            for (int i = 0; i < 3; i++) start[(size_t)b * 7 + i] = p[i] + ((2 * (uv * u[i]) + (sc * sc - uu) * bp[i]) + 2 * sc * c[i]);
            start[(size_t)b * 7 + 3] = q[0] * bq[0] - q[1] * bq[1] - q[2] * bq[2] - q[3] * bq[3];
            start[(size_t)b * 7 + 4] = q[0] * bq[1] + q[1] * bq[0] + q[2] * bq[3] - q[3] * bq[2];
            start[(size_t)b * 7 + 5] = q[0] * bq[2] - q[1] * bq[3] + q[2] * bq[0] + q[3] * bq[1];
            start[(size_t)b * 7 + 6] = q[0] * bq[3] + q[1] * bq[2] - q[2] * bq[1] + q[3] * bq[0];
          }
        }
        fix.push_back({(const void**)&M.kin_tab, bb.add(tab.data(), tab.size())});
      }
    }
    fix.push_back({(const void**)&M.chain_dof, bb.add(chain_dof.data(), sizeof(int) * chain_dof.size())});
    fix.push_back({(const void**)&M.chain_jnt, bb.add(chain_jnt.data(), sizeof(int) * chain_jnt.size())});
  }
  fix.push_back({(const void**)&M.body_depth, bb.add(depth.data(), sizeof(int) * nb)});
  fix.push_back({(const void**)&M.body_chain, bb.add(chain.data(), sizeof(int) * chain.size())});
  fix.push_back({(const void**)&M.body_subtree_end, bb.add(sub_end.data(), sizeof(int) * nb)});
  fix.push_back({(const void**)&M.body_dofmask, bb.add(body_dofmask.data(), sizeof(unsigned long long) * body_dofmask.size())});
  fix.push_back({(const void**)&M.efc_row_con, bb.add(row_con.data(), sizeof(int) * row_con.size())});
  {
    const int nrf = d->nsensor > 0 ? d->sns_rfadr[d->nsensor] : 0;
    std::vector<int> owner((size_t)nrf + 1, 0);
    for (int sidx = 0; sidx < d->nsensor; sidx++)
      for (int q = d->sns_rfadr[sidx]; q < d->sns_rfadr[sidx + 1]; q++) owner[q] = sidx;
    M.nrfq = nrf;
    fix.push_back({(const void**)&M.rf_sensor, bb.add(owner.data(), sizeof(int) * owner.size())});
    std::vector<int> rsite((size_t)nrf + 1, 0), rtype((size_t)nrf + 1, 0);
    std::vector<REAL> rsize((size_t)3 * nrf + 3, (REAL)0);
    for (int q = 0; q < nrf; q++) {
      const int g = d->rf_geom[q];
      rsite[q] = d->sns_objid[owner[q]]; rtype[q] = d->geom_type[g];
      for (int i = 0; i < 3; i++) rsize[3 * q + i] = (REAL)d->geom_size[3 * g + i];
    }
    fix.push_back({(const void**)&M.rf_site, bb.add(rsite.data(), sizeof(int) * rsite.size())});
    fix.push_back({(const void**)&M.rf_gtype, bb.add(rtype.data(), sizeof(int) * rtype.size())});
    fix.push_back({(const void**)&M.rf_gsize, bb.add(rsize.data(), sizeof(REAL) * rsize.size())});
    M.sns_full = 0;
    for (int sidx = 0; sidx < d->nsensor; sidx++) { const int t = d->sns_type[sidx]; if (!(t == 1 || t == 2 || t == 3 || t == 7 || t == 9 || t == 10)) M.sns_full = 1; }
  }
  {
    std::vector<int> row_eq((size_t)d->ne + 1, 0);
    for (int q = 0; q < d->neqtab; q++) {
      const int width = d->eq_kind[q] == 0 ? 3 : (d->eq_kind[q] == 1 ? 6 : 1);
      for (int k = 0; k < width && d->eq_row[q] + k < d->ne; k++) row_eq[d->eq_row[q] + k] = q;
    }
    fix.push_back({(const void**)&M.efc_row_eq, bb.add(row_eq.data(), sizeof(int) * row_eq.size())});
  }
  {
    // transmission tables (smooth.transmission :535-591): the non-zeros of every actuator's moment row, actuator-major (length /
    // velocity of an actuator) and dof-major (qfrc_actuator of a dof).  A coefficient is a model constant (gear component) except
    // for JOINTINPARENT on ball / free joints, where it is a component of the gear axis rotated into the child frame (rot >= 0).
    std::vector<REAL> tenJ((size_t)d->ntendon * nv + 1, (REAL)0);
    for (int t = 0; t < d->ntendon; t++)
      for (int q = d->ten_adr[t]; q < d->ten_adr[t + 1]; q++) tenJ[(size_t)t * nv + d->ten_dof[q]] = (REAL)d->ten_coef[q];
    fix.push_back({(const void**)&M.ten_J0, bb.add(tenJ.data(), sizeof(REAL) * tenJ.size())});
    {  // tendon armature: J^T diag(armature) J (products and sums in REAL, tendons in index order)
      std::vector<REAL> jtaj((size_t)(nv * (nv + 1)) / 2 + 1, (REAL)0);
      M.has_ten_armature = 0;
      for (int t = 0; t < d->ntendon; t++) if (d->tendon_armature[t] != 0) M.has_ten_armature = 1;
      if (M.has_ten_armature)
        for (int i = 0; i < nv; i++)
          for (int j = 0; j <= i; j++) {
            REAL sacc = 0;
            for (int t = 0; t < d->ntendon; t++) sacc += tenJ[(size_t)t * nv + i] * (tenJ[(size_t)t * nv + j] * (REAL)d->tendon_armature[t]);
            jtaj[(size_t)(i * (i + 1)) / 2 + j] = sacc;
          }
      fix.push_back({(const void**)&M.ten_JTAJ, bb.add(jtaj.data(), sizeof(REAL) * jtaj.size())});
    }
    std::vector<REAL> moment((size_t)d->nu * nv, (REAL)0);
    std::vector<int> a_adr((size_t)d->nu + 1, 0), a_dof, a_rot;
    std::vector<REAL> a_coef;
    int has_rot = 0;
    for (int i = 0; i < d->nu; i++) {
      a_adr[i] = (int)a_dof.size();
      const int jt = d->act_jnttype[i], da = d->act_dofadr[i];
      const bool inparent = d->act_trntype[i] == 1;
      if (d->act_trntype[i] == 3) {  // tendon transmission: moment row = ten_J[t] * gear[0] (smooth.py:558-561), dense row in dof order
        const int t = d->act_trnid[i];
        for (int dd = 0; dd < nv; dd++) {
          const REAL c = tenJ[(size_t)t * nv + dd] * (REAL)d->act_gear[6 * i];
          moment[(size_t)i * nv + dd] = c;
          if (tenJ[(size_t)t * nv + dd] != 0) { a_dof.push_back(dd); a_coef.push_back(c); a_rot.push_back(-1); }
        }
        continue;
      }
      const int width = jt == JNT_FREE ? 6 : (jt == JNT_BALL ? 3 : 1);
      for (int k = 0; k < width; k++) {
        const bool rot = inparent && ((jt == JNT_BALL) || (jt == JNT_FREE && k >= 3));
        a_dof.push_back(da + k);
        a_coef.push_back((REAL)d->act_gear[6 * i + k]);
        a_rot.push_back(rot ? (jt == JNT_FREE ? k - 3 : k) : -1);
        if (rot) has_rot = 1; else moment[(size_t)i * nv + da + k] = (REAL)d->act_gear[6 * i + k];
      }
    }
    a_adr[d->nu] = (int)a_dof.size();
    std::vector<int> adr((size_t)nv + 1, 0), ids, d_rot;
    std::vector<REAL> d_coef;
    for (int dd = 0; dd < nv; dd++) {
      adr[dd] = (int)ids.size();
      for (int i = 0; i < d->nu; i++)
        for (int q = a_adr[i]; q < a_adr[i + 1]; q++)
          if (a_dof[q] == dd) { ids.push_back(i); d_coef.push_back(a_coef[q]); d_rot.push_back(a_rot[q]); }
    }
    adr[nv] = (int)ids.size();
    M.act_has_rot = has_rot;
    M.act_simple = 1;
    for (int i = 0; i < d->nu; i++) if ((d->act_jnttype[i] != JNT_SLIDE && d->act_jnttype[i] != JNT_HINGE) || d->act_trntype[i] == 3) M.act_simple = 0;
    a_dof.push_back(0); a_rot.push_back(-1); a_coef.push_back(0); ids.push_back(0); d_rot.push_back(-1); d_coef.push_back(0);  // never empty
    fix.push_back({(const void**)&M.act_moment, bb.add(moment.data(), sizeof(REAL) * moment.size())});
    fix.push_back({(const void**)&M.dof_act_adr, bb.add(adr.data(), sizeof(int) * adr.size())});
    fix.push_back({(const void**)&M.dof_act_id, bb.add(ids.data(), sizeof(int) * ids.size())});
    fix.push_back({(const void**)&M.dof_act_coef, bb.add(d_coef.data(), sizeof(REAL) * d_coef.size())});
    fix.push_back({(const void**)&M.dof_act_rot, bb.add(d_rot.data(), sizeof(int) * d_rot.size())});
    fix.push_back({(const void**)&M.act_ent_adr, bb.add(a_adr.data(), sizeof(int) * a_adr.size())});
    fix.push_back({(const void**)&M.act_ent_dof, bb.add(a_dof.data(), sizeof(int) * a_dof.size())});
    fix.push_back({(const void**)&M.act_ent_coef, bb.add(a_coef.data(), sizeof(REAL) * a_coef.size())});
    fix.push_back({(const void**)&M.act_ent_rot, bb.add(a_rot.data(), sizeof(int) * a_rot.size())});
    {
      std::vector<int> flim((size_t)nv + 1, 0);
      std::vector<REAL> frange((size_t)2 * nv + 2, (REAL)0);
      for (int dd = 0; dd < nv; dd++) {
        const int j = d->dof_jntid[dd];
        flim[dd] = d->jnt_actfrclimited[j];
        frange[2 * dd] = (REAL)d->jnt_actfrcrange[2 * j]; frange[2 * dd + 1] = (REAL)d->jnt_actfrcrange[2 * j + 1];
      }
      fix.push_back({(const void**)&M.dof_frc_lim, bb.add(flim.data(), sizeof(int) * flim.size())});
      fix.push_back({(const void**)&M.dof_frc_range, bb.add(frange.data(), sizeof(REAL) * frange.size())});
    }
    M.inv_nv = nv > 0 ? 1.0f / (float)nv : 0.0f;
  }
  // single-column rows: dof-frictionloss rows (J = e_dof) first, then slide/hinge limit rows (J = +-e_dof or 0)
  const int ncr = d->nf + d->nl;
  std::vector<int> lim_dof((size_t)ncr, 0);
  for (int r = 0; r < d->nf; r++) lim_dof[r] = d->fric_dof[r];
  for (int r = 0; r < d->nl; r++) lim_dof[d->nf + r] = d->jnt_dofadr[d->lim_jnt[r]];
  fix.push_back({(const void**)&M.lim_dof, bb.add(lim_dof.data(), sizeof(int) * lim_dof.size())});
  std::vector<int> dof_limrow((size_t)2 * nv, -1);
  for (int r = 0; r < ncr; r++) {
    const int dd = lim_dof[r];
    if (dof_limrow[2 * dd] < 0) dof_limrow[2 * dd] = r;
    else if (dof_limrow[2 * dd + 1] < 0) dof_limrow[2 * dd + 1] = r;
    else return fail(-22, "more than two single-column rows on one dof");
  }
  fix.push_back({(const void**)&M.dof_limrow, bb.add(dof_limrow.data(), sizeof(int) * dof_limrow.size())});
  M.max_depth = max_depth;
  M.sol_qm_lds = (d->nefc > 0 && d->iterations > 4) ? 1 : 0;  // one pass over qM per solver iteration: from LDS when there are many
  {  // register solver, Newton, float32: J^T diag(w) J on the matrix cores (v_mfma_f32_4x4x1, 16 independent 4 x 4 blocks: each environment's lanes feed only their own blocks).  MJH_SOL2_MFMA=0: vector path
    static const bool mfma_off = [] { const char* e = getenv("MJH_SOL2_MFMA"); return e && e[0] == '0'; }();
    M.sol2_hs = (!mfma_off && sizeof(REAL) == 4 && d->solver == SOL_NEWTON && d->nv <= 16 && d->nefc > 0) ? 1 : 0;
    static const bool incr_off = [] { const char* e = getenv("MJH_SOL2_INCR"); return e && e[0] == '0'; }();
    M.sol2_incr = (!incr_off && d->solver == SOL_NEWTON && d->nv <= 16 && d->nefc > 0) ? 1 : 0;
  }
  // convex pairs and the LDS scratch their wave needs (layout in mjh_convex.h)
  std::vector<int> cvx_pairs;
  int cvx_reals = 0;
  for (int p = 0; p < d->npair; p++) {
    const int fn = d->pair_fn[p];
    if (fn < MJH_FN_PLANE_CONVEX) continue;
    if (fn > MJH_FN_CONVEX_CONVEX) return fail(-38, "pair function not implemented");
    const int c2 = d->geom_convexid[d->pair_geom2[p]], c1 = d->geom_convexid[d->pair_geom1[p]];
    if (c2 < 0 || c2 >= d->nconvex || (fn == MJH_FN_CONVEX_CONVEX && (c1 < 0 || c1 >= d->nconvex))) return fail(-22, "convex pair without convex tables");
    int need = 0;
    if (fn == MJH_FN_PLANE_CONVEX) need = 5 * d->convex_nvert[c2];
    else if (fn == MJH_FN_CAPSULE_CONVEX) need = 6 * d->convex_nfv[c2];
    else if (fn == MJH_FN_CONVEX_CONVEX) {
      const int K = d->convex_nfv[c1] > d->convex_nfv[c2] ? d->convex_nfv[c1] : d->convex_nfv[c2];
      need = 3 * (d->convex_nvert[c1] + d->convex_nface[c1] + d->convex_nvert[c2] + d->convex_nface[c2]) + 43 * K;
    }
    if (need > cvx_reals) cvx_reals = need;
    cvx_pairs.push_back(p);
  }
  fix.push_back({(const void**)&M.cvx_pairs, bb.add(cvx_pairs.data(), sizeof(int) * cvx_pairs.size())});
  M.ncvxpair = (int)cvx_pairs.size();
  out->cvx_lds_bytes = (cvx_reals + 8) * (int)sizeof(REAL);
  if (out->cvx_lds_bytes > 64 * 1024) return fail(-12, "convex hull too large for the pair kernel's LDS scratch");

  for (int p = 0; p < MJH_NARENA; p++) {
    out->lds_bytes[p] = lds_carve(M, 1 << p, out->off[p]) * (int)sizeof(REAL);
    if (p == 4 && out->lds_bytes[p] > 160 * 1024 && M.sol_qm_lds) {  // a large model: qM stays in global memory (L2) for the solver's products
      M.sol_qm_lds = 0;
      out->lds_bytes[p] = lds_carve(M, 1 << p, out->off[p]) * (int)sizeof(REAL);
    }
    if (p < MJH_NPHASE && out->lds_bytes[p] > 160 * 1024) return fail(-12, "model does not fit the 160 KiB LDS of one CU");
  }
  {  // register solver: CG, slide / hinge limit rows + contact rows only, one dof per lane of a 32-lane half
    static const bool off = [] { const char* e = getenv("MJH_SOL2"); return e && e[0] == '0'; }();
    const int nd = d->nefc - d->nf - d->nl;
    const bool general = d->nf > 0 || d->nft > 0 || d->ne > 0 || d->nlb > 0 || d->nlt > 0;
    out->sol2_nmax = out->sol2_rpl = 0;
    const int nmax = d->nv <= 8 ? 8 : (d->nv <= 16 ? 16 : 28), rpl = nd <= 32 ? 1 : (nd <= 64 ? 2 : (nd <= 128 ? 4 : 8));
    // Newton (Hessian built and factorised in registers): two environments per wavefront run until BOTH have converged, which pays
    // when the solves are short -- measured on MI355X: ant (nv 8, 1 - 3 iterations) solver phase 229 -> 175 us, mesh scene (nv 12, up to
    // 100 iterations x 50 line-search steps, very uneven across environments) 548 -> 691 us.  MJH_SOL2_NEWTON_NV moves the cut.
    static const int newton_nv = [] { const char* e = getenv("MJH_SOL2_NEWTON_NV"); return e ? atoi(e) : 8; }();
    // (... with FOUR environments per wavefront -- the first tier of models whose nv, na and single-column rows fit 16 lanes, see below -- the shared
    // nv-sized work outweighs the wait: the mesh scene's solver phase 338 -> see profiles/r03/notes.md)
    static const bool w16_off = [] { const char* e = getenv("MJH_SOL2_W16"); return e && e[0] == '0'; }();
    const bool w16 = !w16_off && nmax <= 16 && d->nv <= 16 && d->na <= 16 && d->nl <= 16;
    const bool solver_ok = d->solver == SOL_CG || (d->solver == SOL_NEWTON && nmax <= 16 && (d->nv <= newton_nv || w16));
    if (!off && solver_ok && !general && d->nv <= 28 && nd <= 32 * (nmax <= 16 ? 8 : 2) && d->nl <= 32 && d->na <= 32 && d->nq <= 64 && 2 * out->lds_bytes[5] <= 64 * 1024) {
      out->sol2_nmax = nmax;
      out->sol2_rpl = rpl;
      // measured on MI355X (profiles/r02/notes.md): ant solver phase 124 -> 114 us (the narrow tier runs three waves per SIMD).  MJH_SOL2_TIERS=0 keeps the single full-width launch.
      static const bool tiers_off = [] { const char* e = getenv("MJH_SOL2_TIERS"); return e && e[0] == '0'; }();
      // Small models (nv, na and the single-column rows fit 16 lanes): the first tier packs FOUR environments per wavefront.  The solver's nv-sized
      // work (substitutions, Cholesky of the Newton Hessian, M products) costs a wave the same whatever it carries, so its cost per environment
      // halves, and a CU keeps twice the environments in flight.  Row slots of that tier: 2 per lane (32 rows) for contacts of up to 4 rows,
      // 5 (80 rows) for wider ones (condim 6 pyramidal: 10 rows per contact).  MJH_SOL2_W16=0 keeps two per wavefront; MJH_SOL2_W16_RPL picks the slots.
      static const int w16_rpl_env = [] { const char* e = getenv("MJH_SOL2_W16_RPL"); return e ? atoi(e) : 0; }();
      const int w16_rpl = (w16_rpl_env == 2 || w16_rpl_env == 5) ? w16_rpl_env : ((M.con_rows > 0 && M.con_rows <= 4) || nd <= 32 ? 2 : 5);
      if (w16) {
        DevModel<REAL> Mc = M;
        Mc.sol2_row_cap = 16 * w16_rpl;
        out->lds_tier = lds_carve(Mc, PH_SOL2, out->off_tier) * (int)sizeof(REAL);
        if (4 * out->lds_tier <= 64 * 1024) {
          out->sol2_tiers = 1; out->sol2_w16_rpl = w16_rpl;
          out->sol2_w16_nmax = d->nv <= 8 ? 8 : (d->nv <= 12 ? 12 : 16);
          // Opt-in (MJH_SOL2_ITCAP / MJH_SOL2_LSCAP > 0): long Newton solves leave the packed tier once they pass the caps and are redone by the LDS
          // solver, which gives ONE environment all 64 lanes.  Measured on the mesh scene (B = 8192, profiles/r03/notes.md): solver
          // phase 257 us without caps, 323 / 346 / 377 / 407 us with caps of 2/6, 3/8, 4/12, 5/20 iterations / line-search iterations -- redoing a long
          // solve from its inputs at one environment per wavefront costs more than the three lane groups it frees, so the default is no caps.
          static const int it_env = [] { const char* e = getenv("MJH_SOL2_ITCAP"); return e ? atoi(e) : -1; }();
          static const int ls_env = [] { const char* e = getenv("MJH_SOL2_LSCAP"); return e ? atoi(e) : -1; }();
          if (it_env >= 0) out->sol2_it_cap = it_env;
          if (ls_env >= 0) out->sol2_ls_cap = ls_env;
          if (out->sol2_it_cap <= 0 || out->sol2_ls_cap <= 0) out->sol2_it_cap = out->sol2_ls_cap = 0;
          if (d->solver != SOL_NEWTON || (d->nf > 0 || d->nft > 0 || d->ne > 0 || d->nlb > 0 || d->nlt > 0)) out->sol2_it_cap = out->sol2_ls_cap = 0;  // the fallback is kernel 4
        }
      }
      if (!out->sol2_tiers && rpl > 1 && !tiers_off) {
        DevModel<REAL> Mc = M;
        Mc.sol2_row_cap = 32;
        out->lds_tier = lds_carve(Mc, PH_SOL2, out->off_tier) * (int)sizeof(REAL);
        out->sol2_tiers = 1;
      }
    }
  }
  {  // constraint stage + register solver in one kernel: plain rows, one dense-row slot per lane, no tiers, nv in (16, 28] (the instantiation that is built).
     // Measured on the humanoid (MI355X, B = 4096): profiles/r04/notes.md.  MJH_FUSE_CS=0 keeps the two launches.
    static const bool off = [] { const char* e = getenv("MJH_FUSE_CS"); return e && e[0] == '0'; }();
    bool mono = true;  // contact rows in contact order (the in-place row compaction walks them upwards)
    for (int c = 1; c < d->ncon; c++) mono = mono && d->con_efc_address[c] > d->con_efc_address[c - 1];
    out->fuse_cs = 0;
    if (!off && out->sol2_nmax == 28 && out->sol2_rpl == 1 && !out->sol2_tiers && !M.con_direct && !M.con_general && !M.topk && d->ncon > 0 && d->nefc > 0 && mono &&
        M.con_rows > 0 && M.ncrow == d->nefc - d->nl && d->nf == 0 && d->nv <= 32 && d->nq <= 64 && d->nl <= 32 && d->na <= 32) {
      const int reals = lds_carve(M, PH_CS, out->off_cs);
      if (reals > 0 && 2 * reals * (int)sizeof(REAL) <= 64 * 1024) {
        out->lds_cs = reals * (int)sizeof(REAL);
        out->fuse_cs = 1;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, 28, 1, 33>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_cs));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, 28, 1, 35>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_cs));
      }
    }
  }
  out->leaf_count = leaf_counts(d);
  out->work_reals = 0;
  if (d->integrator == INT_RK4) {
    const char* names[] = {
#define X(n) #n,
        MJH_DATA_REALS(X)
#undef X
    };
    for (size_t i = 0; i < out->leaf_count.size(); i++)
      if (is_stage_leaf(names[i], M.ncvxpair > 0, M.has_fluid != 0, M.ne > 0, M.topk != 0)) out->work_reals += out->leaf_count[i];
    out->work_reals += 4 * (int64_t)d->nv + 2 * (int64_t)d->na;  // qvel0, kqvel(unused), sum_qvel, sum_qacc, act0, sum_actdot
  }
  {  // hand-over of the small-model constraint phase (kernel 8) to the register solver: [count | contact -> slot | D | aref | rows] of the active contacts' rows, compact.  MJH_HANDOVER=0: off
    static const bool off = [] { const char* e = getenv("MJH_HANDOVER"); return e && e[0] == '0'; }();
    const int64_t nd = (int64_t)d->nefc - d->nl;
    // Measured (MI355X, profiles/r05/notes.md): ant (RK4: stages 1..3 write the hand-over INSTEAD of their workspace leaves) solver phase 44.7 -> 42.3 us per stage, constraint phase unchanged;
    // mesh scene (Euler: the leaves AND the hand-over) solver 197.9 -> 194.4 us but constraint phase 47.2 -> 52.0: RK4 models only.
    // (a -DMJH_SOL2_CAPS build bypasses the hand-over in the solver: it is not written there either -- ADVICE r05)
    out->hs_reals = (!off && !MJH_SOL2_CAPS_ON && d->integrator == INT_RK4 && M.con_direct && M.crow_by_con && out->sol2_nmax && !out->fuse_cs && nd > 0) ? ((1 + (int64_t)d->ncon + 2 * nd + nd * d->nv + 3) & ~(int64_t)3) : 0;
    out->work_reals += out->hs_reals;
  }
  out->cand_reals = (d->topk && M.ncvxpair > 0) ? 13 * (int64_t)d->ncand : 0;  // candidate contacts of the convex narrow phase (dist, pos, frame)
  out->work_reals += out->cand_reals;
  {  // register solver at four environments per wavefront with solves of uneven length (Newton): two ints per environment at the head of the workspace -- the solver's
     // slot -> environment list of this step and the iteration-count key each environment leaves for the next one.
    // Measured (MI355X, profiles/r04/notes.md): the mesh scene's solver phase 203.1 us with the list against 203.9 us without -- last step's counts do not predict this
    // step's (which environments take the long line searches changes from step to step) -- and the one-workgroup sort costs 14.6 us at B = 8192, 30.7 us at B = 16384:
    // opt-in (MJH_SOL2_SORT=1), off by default.
    static const bool on = [] { const char* e = getenv("MJH_SOL2_SORT"); return e && e[0] == '1'; }();
    out->sort_reals = (on && out->sol2_w16_rpl && d->solver == SOL_NEWTON) ? (int64_t)(8 / sizeof(REAL)) : 0;
    out->work_reals += out->sort_reals;
  }

  void* dev = nullptr;
  HIP_TRY(hipMalloc(&dev, bb.host.size() + 16));
  HIP_TRY(hipMemcpy(dev, bb.host.data(), bb.host.size(), hipMemcpyHostToDevice));
  for (auto& f : fix) *f.first = (const unsigned char*)dev + f.second;
  out->blob = dev;
  out->blob_bytes = bb.host.size();
#define SET_ATTR(P) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, P, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_bytes[P]));
  SET_ATTR(0) SET_ATTR(1) SET_ATTR(2) SET_ATTR(3) SET_ATTR(4)
#undef SET_ATTR
  // phases that are register-bound (not LDS-bound) run two environments per wavefront, 32 lanes each: the same
  // VGPR budget then keeps twice as many environments in flight (mjh_kernels.h, Env<REAL, W>)
  for (int p = 0; p < MJH_NPHASE; p++) out->pack2[p] = out->pack4[p] = 0;
#define SET_PACK(P)                                                                                                          \
  if (2 * out->lds_bytes[P] <= 64 * 1024) {                                                                                  \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, P, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_bytes[P])); \
    out->pack2[P] = 1;                                                                                                       \
  }
  SET_PACK(0) SET_PACK(3)
  {  // fused kinematics + velocity kernel (plain velocity phase only).  Measured on MI355X: profiles/r02/notes.md.  MJH_FUSE_KV=0 keeps two launches.
    static const bool fuse_off = [] { const char* e = getenv("MJH_FUSE_KV"); return e && e[0] == '0'; }();
    int defer_ok = 0;
    out->lds_kv = lds_carve(M, PH_KINVEL, out->off_kv, &defer_ok) * (int)sizeof(REAL);
    static const bool defer_off = [] { const char* e = getenv("MJH_KV_DEFER"); return e && e[0] == '0'; }();
    M.kv_defer = (defer_ok && !defer_off) ? 1 : 0;
    out->fuse_kv = (!fuse_off && !(M.has_fluid || M.has_gravcomp || M.ntendon > 0 || M.big) && out->lds_kv <= 160 * 1024) ? 1 : 0;
    if (out->fuse_kv) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 12, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_kv));
      if (2 * out->lds_kv <= 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 12, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_kv));
      if (4 * out->lds_kv <= 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 12, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * out->lds_kv));
    }
  }
  static const bool con_pack = [] { const char* e = getenv("MJH_CON_PACK"); return !(e && e[0] == '0'); }();
  if (con_pack && M.con_direct && 2 * out->lds_bytes[2] <= 64 * 1024) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 8, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_bytes[2]));
    out->pack2[2] = 1;
  }
  {  // experiment (MJH_CON2_PACK=1): the plain constraint phase of a mid-size model at two environments per wavefront
    static const bool con2 = [] { const char* e = getenv("MJH_CON2_PACK"); return e && e[0] == '1'; }();
    if (con2 && !M.con_direct && !M.con_general && 2 * out->lds_bytes[2] <= 64 * 1024) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 2, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_bytes[2]));
      out->pack2[2] = 1;
    }
  }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 8, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_bytes[2]));  // plain constraint phase: its contact rows go straight to the leaf, the arena is small enough for two per wavefront
  // CRB: the register Cholesky keeps one matrix row per lane, so nv <= 32 would fit a 32-lane half too; measured on the float64
  // humanoid (nv 27) two per wavefront is SLOWER, 50.2 vs 39.9 us (132 VGPRs: 3 waves / SIMD instead of 4, and every broadcast of the
  // factorisation becomes two v_readlane + a select): opt-in only (MJH_CRB_PACK=1).  float32 small models: +13 % on the ant (round 1).
  // (round 4, after the register work of round 3: 35.1 vs 36.5 us, and it is what lets kernel 13 serve the humanoid: default on, MJH_CRB_PACK=0 turns it off)
  static const bool crb_pack = [] { const char* e = getenv("MJH_CRB_PACK"); return !(e && e[0] == '0'); }();
  if ((sizeof(REAL) == 4 && d->nv <= 16) || (crb_pack && d->nv <= 32 && d->nv > 16)) { SET_PACK(1) }
#undef SET_PACK
  // small models (every per-body / per-joint / per-dof loop fits 16 lanes): FOUR environments per wavefront in the packed phases.
  // These phases are bound by the number of dependent chains a SIMD can interleave (registers cap the waves), so the same waves
  // carrying twice the environments is close to twice the throughput -- measured on the ant (B = 16384, 64 environments per CU):
  // see profiles/r02/notes.md.  MJH_PACK4=0 turns it off.
  static const bool pack4 = [] { const char* e = getenv("MJH_PACK4"); return !(e && e[0] == '0'); }();
  if (pack4 && d->nbody <= 16 && d->njnt <= 16 && d->nv <= 16) {
#define SET_PACK4(P, Q)                                                                                                      \
    if (out->pack2[Q] && 4 * out->lds_bytes[Q] <= 64 * 1024) {                                                               \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, P, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * out->lds_bytes[Q])); \
      if (P == Q) out->pack4[Q] = 1;                                                                                         \
    }
    SET_PACK4(0, 0) SET_PACK4(1, 1) SET_PACK4(3, 3) SET_PACK4(5, 3)
#undef SET_PACK4
  }
  {  // kinematics + crb / factor + velocity as ONE kernel (13): when the crb stage packs environments per wavefront the way the other two do (float32 models with
     // nv <= 16; MJH_CRB_PACK for the rest).  Saves the crb kernel's loads of cinert / cdof, its launch and the wait between the launches.  MJH_FUSE_CRB=0: off.
    static const bool off = [] { const char* e = getenv("MJH_FUSE_CRB"); return e && e[0] == '0'; }();
    out->lds_kcv = lds_carve(M, PH_KCV, out->off_kcv) * (int)sizeof(REAL);
    const bool same_pack = out->pack2[1] == (out->pack2[0] && out->pack2[3] && 2 * out->lds_kv <= 64 * 1024) && out->pack4[1] == (out->pack4[0] && out->pack4[3] && 4 * out->lds_kv <= 64 * 1024);
    out->fuse_kcv = (!off && out->fuse_kv && same_pack && out->pack2[1] && out->lds_kcv <= 160 * 1024 && 2 * out->lds_kcv <= 64 * 1024 && (!out->pack4[1] || 4 * out->lds_kcv <= 64 * 1024)) ? 1 : 0;
    if (out->fuse_kcv) {
      // measured (MI355X, profiles/r03/notes.md): mesh scene, B = 8192 = one round of 2048 four-environment waves at two per SIMD: 45.3 us against 33.7 + 20.4 in two launches;
      // ant, B = 16384 = two rounds: 95.7 us against 76.4 + 20.9 -- the separate kernels fit four waves per SIMD.  Larger batches keep the separate launches.
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      out->kcv_max_envs = (int64_t)cus * 4 /* SIMDs */ * 2 /* waves per SIMD */ * (out->pack4[1] ? 4 : 2);
      out->kcv_max_envs = (int64_t)1 << 62;  // round 4: at every batch size -- the humanoid (two per wavefront) at B = 32768: 652 us against 452 + 218; the ant (four per wavefront) at B = 16384, two rounds: 93.6 us against 75.8 + 21.8 (profiles/r04/notes.md)
      static const long long kcv_env = [] { const char* e = getenv("MJH_KCV_MAX_ENVS"); return e ? atoll(e) : -1ll; }();  // experiments: the batch bound of kernel 13
      if (kcv_env >= 0) out->kcv_max_envs = kcv_env;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 13, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_kcv));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 13, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_kcv));
      if (out->pack4[1]) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 13, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * out->lds_kcv));
    }
  }
  {  // kernel 13 on two wavefronts per workgroup (velocity beside crb / factor): packed float32 models.  MJH_KCV2=0 / 1 forces it off / on (also for float64, two per wavefront).
    static const int sw = [] { const char* e = getenv("MJH_KCV2"); return !e ? -1 : (e[0] == '1' ? 1 : 0); }();
    out->kcv2 = 0;
    if (out->fuse_kcv && sw != 0 && (sw == 1 || (sizeof(REAL) == 4 && out->pack4[1]))) {
      out->lds_kcv2 = lds_carve(M, PH_KCV2, out->off_kcv2) * (int)sizeof(REAL);
      const int per_wg = (out->pack4[1] ? 4 : 2) * out->lds_kcv2;
      if (per_wg <= 64 * 1024) {
        out->kcv2 = 1;
        // The two-wave form pays while the whole batch is resident at once: two wavefronts per workgroup at four per SIMD are 8 workgroups per CU (and their arenas must fit the
        // CU's 160 KB).  Past that it runs more rounds of waves than kernel 13 (which fits 16 one-wave workgroups) and loses.  Measured (MI355X, profiles/r05/notes.md): mesh scene
        // B = 8192 37.5 us against 46.0; ant B = 8192 46.1 against 54.5; ant B = 16384 (two rounds against one) 101 against 62.
        // The limit belongs to the device a LAUNCH runs on, not to the one that was current when the model was built: kcv2_limit() below, cached per device.
        out->kcv2_per_wg = per_wg;
        out->kcv2_envs_per_wg = out->pack4[1] ? 4 : 2;
        out->kcv2_max_envs = sw == 1 ? ((int64_t)1 << 62) : 0;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 17, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_kcv2));
        if (out->pack4[1]) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 17, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * out->lds_kcv2));
      }
    }
  }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 6, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_bytes[4]));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 7, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_bytes[2]));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 5, MJH_WAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_bytes[3]));
  if (out->pack2[3]) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_phase_kernel<REAL, 5, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_bytes[3]));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_convex_kernel<REAL>), hipFuncAttributeMaxDynamicSharedMemorySize, out->cvx_lds_bytes));
  if (out->sol2_nmax) {
#define SET_SOL2(N, R) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, N, R, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_bytes[5]));
    SET_SOL2(8, 1) SET_SOL2(8, 2) SET_SOL2(8, 4) SET_SOL2(8, 8) SET_SOL2(16, 1) SET_SOL2(16, 2) SET_SOL2(16, 4) SET_SOL2(16, 8) SET_SOL2(28, 1) SET_SOL2(28, 2)
#undef SET_SOL2
    if (out->sol2_w16_rpl) {
#define SET_SOL2W(N, R) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, N, R, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * out->lds_tier)); HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, N, R, 17>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * out->lds_tier));
      SET_SOL2W(8, 2) SET_SOL2W(8, 5) SET_SOL2W(12, 2) SET_SOL2W(12, 5) SET_SOL2W(16, 2) SET_SOL2W(16, 5)
#undef SET_SOL2W
    }
  }
  {  // the whole pass in one kernel: models both fused kernels serve at two environments per wavefront, without convex pairs or sensors (their kernels sit between the two halves).
    // Measured (MI355X, humanoid B = 4096, three A / B pairs in one call, profiles/r04/notes.md): 166.8 - 170.5 us in one launch against 91.7 - 93.1 + 76.4 - 76.7 in two
    // (23.5 - 24.3 M against 24.1 - 24.6 M env-steps/s): the one function is allocated 256 VGPRs + 304 B of scratch where the halves take 198 + 0 and 256 + 120 B, and that
    // costs what the missing device-wide barrier saves.  Opt-in (MJH_FUSE_ALL=1).
    // Round 5: for models with opt.iterations == 1 (the humanoid benchmark) the one-iteration instantiation (W = 36: the solver loop is straight-line code, the factor of M
    // dies behind its one preconditioning step) takes 256 VGPRs + 52 B where the generic one takes 256 + 304 B, and the single launch wins at every batch size: B = 4096
    // 155.0 - 155.9 us against 91.0 + 71.0 - 71.6 in two launches (25.5 - 25.7 M against 25.0 - 25.2 M env-steps/s), B = 32768 1162 us against 657 + 577 (27.8 - 28.1 M against
    // 26.4 M; profiles/r05/notes.md).  Default on for those models (MJH_FUSE_ALL=0: off); still opt-in (MJH_FUSE_ALL=1) for the others.
    static const int sw = [] { const char* e = getenv("MJH_FUSE_ALL"); return !e ? -1 : (e[0] == '1' ? 1 : 0); }();
    const bool on = sw != 0;  // (round 5, second half: without the kernels' grid-stride loops the generic instantiation takes 228 VGPRs and no scratch either: default on for every model both fused kernels serve)
    out->fuse_all = 0;
    if (on && out->fuse_cs && out->fuse_kcv && out->pack2[1] && !out->pack4[1] && M.ncvxpair == 0 && d->nsensor == 0) {
      out->lds_all = out->lds_kcv > out->lds_cs ? out->lds_kcv : out->lds_cs;
      if (2 * out->lds_all <= 64 * 1024) {
        out->fuse_all = 1;
        M.nt_all = M.all_handoff;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, 28, 1, 34>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_all));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, 28, 1, 36>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * out->lds_all));
      }
    }
  }
  {  // One launch per RK4 stage (stages 1..3) for small Newton models: kernel 13 at four environments per wavefront, the direct constraint phase (kernel 8, two per wavefront: run twice
     // per wave) and the solver's packed Newton tier at four, behind one another in mjh_sol2_kernel<.., 18>.  MJH_FUSE_STAGE=0: the three launches.
    static const bool off = [] { const char* e = getenv("MJH_FUSE_STAGE"); return e && e[0] == '0'; }();
    out->fuse_stage = 0;
    if (!off && d->integrator == INT_RK4 && d->solver == SOL_NEWTON && out->fuse_kcv && out->pack4[1] && M.con_direct && out->pack2[2] && out->sol2_tiers && out->sol2_w16_nmax == 8 &&
        (out->sol2_w16_rpl == 2 || out->sol2_w16_rpl == 5) && M.ncvxpair == 0 && out->sort_reals == 0 && d->nefc > 0 && !(MJH_SOL2_CAPS_ON && out->sol2_it_cap > 0)) {
      int need = 4 * out->lds_kcv;
      if (2 * out->lds_bytes[2] > need) need = 2 * out->lds_bytes[2];
      if (4 * out->lds_tier > need) need = 4 * out->lds_tier;
      if (need <= 64 * 1024) {
        out->fuse_stage = 1;
        out->lds_stage = need;
      }
    }
    // ... and the tail of a pass alone (MJH_FUSE_TAIL=0: the two launches)
    // Measured (MI355X, profiles/r06/notes.md): mesh scene (12 dofs: the tier's instantiation takes 227 VGPRs, two waves per SIMD, and the constraint phase inherits that) 219.6 us
    // against 167.9 + 49.2 in two launches -- default for 8-dof tiers only, whose kernel sits at the constraint phase's own four waves per SIMD; MJH_FUSE_TAIL=1 / 0 forces it on / off.
    static const int tail_sw = [] { const char* e = getenv("MJH_FUSE_TAIL"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    const bool tail_off = tail_sw == 0 || (tail_sw < 0 && out->sol2_w16_nmax != 8);
    out->fuse_tail = 0;
    if (!tail_off && d->solver == SOL_NEWTON && M.con_direct && out->pack2[2] && out->sol2_tiers && out->sol2_w16_rpl > 0 && out->sort_reals == 0 && d->nefc > 0 && !M.topk &&
        !(MJH_SOL2_CAPS_ON && out->sol2_it_cap > 0)) {
      int need = 2 * out->lds_bytes[2];
      if (4 * out->lds_tier > need) need = 4 * out->lds_tier;
      if (need <= 64 * 1024) { out->fuse_tail = 1; out->lds_tail = need; }
    }
    if (out->fuse_stage || out->fuse_tail) {
      const int need = out->lds_stage > out->lds_tail ? out->lds_stage : out->lds_tail;
#define SET_ST(N, R) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_sol2_kernel<REAL, N, R, 18>), hipFuncAttributeMaxDynamicSharedMemorySize, need));
      SET_ST(8, 2) SET_ST(8, 5) SET_ST(12, 2) SET_ST(12, 5) SET_ST(16, 2) SET_ST(16, 5)
#undef SET_ST
    }
  }
  return 0;
}

// Workgroups per launch.  The kernels run ONE unit of work per workgroup (no grid-stride loop), so a batch past this many is cut into several launches.  2^20 by default;
// MJH_MAX_GRID_LOG2 lowers it so that the tests reach the multi-launch paths at small batches.
static int64_t max_grid() {
  static const int64_t g = [] { const char* e = getenv("MJH_MAX_GRID_LOG2"); int v = e ? atoi(e) : 20; return (int64_t)1 << (v < 0 ? 0 : (v > 20 ? 20 : v)); }();
  return g;
}

template <typename REAL, int P, int W>
int launch_range(const mjhModel* m, KArgs<REAL>& a, int64_t begin, int64_t count, hipStream_t stream) {
  if (count <= 0) return 0;
  constexpr int NSUB = MJH_WAVE / W;
  constexpr int A = P == 5 ? 3 : (P == 6 ? 4 : ((P == 7 || P == 8) ? 2 : (P == 17 ? 0 : P)));  // kernels 5 / 6: velocity phase with fluid forces, solver phase with frictionloss rows
  const int arena_bytes = P == 17 ? m->lds_kcv2 : (P == 13 ? m->lds_kcv : (P == 12 ? m->lds_kv : m->lds_bytes[(P == 12 || P == 13) ? 0 : A]));
  a.off = P == 17 ? m->off_kcv2 : (P == 13 ? m->off_kcv : (P == 12 ? m->off_kv : m->off[(P == 12 || P == 13) ? 0 : A]));
  a.env_begin = begin; a.env_count = count;
  a.lds_reals = arena_bytes / (int)sizeof(REAL);
  const int64_t blocks = count / NSUB;
  for (int64_t b0 = 0; b0 < blocks; b0 += max_grid()) {  // the kernels have no grid-stride loop: one launch per 2^20 workgroups
    const int64_t grid = blocks - b0 < max_grid() ? blocks - b0 : max_grid();
    a.env_begin = begin + b0 * NSUB; a.env_count = grid * NSUB;
    hipLaunchKernelGGL((mjh_phase_kernel<REAL, P, W>), dim3((unsigned)grid), dim3(P == 17 ? 2 * MJH_WAVE : MJH_WAVE), (size_t)(NSUB * arena_bytes), stream, a);  // (17: a second wavefront per workgroup runs the crb / factor stage)
    HIP_TRY(hipGetLastError());
  }
  timing_mark(stream, P);
  return 0;
}
template <typename REAL, int P>
int launch_phase(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream) {
  constexpr bool PACKABLE = (P == 0 || P == 1 || P == 2 || P == 3 || P == 5 || P == 8 || P == 12 || P == 13 || P == 17);
  constexpr int PI = !PACKABLE ? 0 : (P == 5 ? 3 : (P == 8 ? 2 : ((P == 12 || P == 13 || P == 17) ? 0 : P)));  // index into the per-phase packing flags
  const bool can2 = !PACKABLE ? false : ((P == 13 || P == 17) ? (bool)m->pack2[1] : (P == 12 ? (m->pack2[0] && m->pack2[3] && 2 * m->lds_kv <= 64 * 1024) : (bool)m->pack2[PI]));  // (13: fuse_kcv holds only when the crb stage packs like the other two)
  const bool can4 = !PACKABLE ? false : ((P == 13 || P == 17) ? (bool)m->pack4[1] : (P == 12 ? (m->pack4[0] && m->pack4[3] && 4 * m->lds_kv <= 64 * 1024) : (P != 8 && P != 2 && m->pack4[PI])));
  if (PACKABLE && can2 && a.B >= 2) {  // groups of four / pairs of environments, then the odd one on its own
    int64_t done = 0;
    int rc = 0;
    if (can4 && a.B >= 4) {
      done = a.B & ~(int64_t)3;
      if ((rc = launch_range<REAL, P, ((PACKABLE && P != 8 && P != 2) ? 16 : MJH_WAVE)>(m, a, 0, done, stream))) return rc;
    }
    const int64_t even = (a.B - done) & ~(int64_t)1;
    if ((rc = launch_range<REAL, P, (PACKABLE ? 32 : MJH_WAVE)>(m, a, done, even, stream))) return rc;
    return launch_range<REAL, (P == 17 ? 13 : P), MJH_WAVE>(m, a, done + even, a.B - done - even, stream);  // (the odd environment: one per wavefront -- the two-wave form only exists packed, kernel 13 serves it)
  }
  return launch_range<REAL, (P == 17 ? 13 : P), MJH_WAVE>(m, a, 0, a.B, stream);
}

// MJH_XSWAP=mask (12 bits of the workgroup index; 0: off): in the whole-pass kernel and the stage kernel, workgroups of odd parity under the mask run the velocity stage before
// crb / factor (mjh_sol2_kernel: "out of lockstep").  Default 0x100: workgroups 256 apart, i.e. different wave slots of the same CUs under the round-robin placement.  Measured
// (MI355X, humanoid B = 4096, profiles/r06/notes.md): 141.5 - 144.0 us -> 137.0 - 137.6 us whatever the mask (0x1 .. 0xfff); B = 32768 (eight rounds: the waves drift apart by themselves) unchanged.
// Launches of more than 4096 workgroups keep one order: B = 32768 (eight rounds, the waves drift apart by themselves) measured 1015 us in one order against 1033 us with the swap.
static int xswap_flags(int64_t workgroups) {
  static const int f = [] { const char* e = getenv("MJH_XSWAP"); const long v = e ? strtol(e, nullptr, 0) : 0x100; return (int)((v & 0xfff) << 16); }();
  static const bool forced = getenv("MJH_XSWAP") != nullptr;  // (an explicit mask applies at every size: the bit-identity tests)
  return (forced || workgroups <= 4096) ? f : 0;
}

// MJH_XSWAP_K=mask (default 0x200; 0: off): the same idea inside the kinematics stage -- workgroups of odd parity under this mask run com_pos before the geom / site / camera frames
// (Env::kinematics, com_first).  Measured (MI355X, humanoid B = 4096, profiles/r06/notes.md): 138.8 -> 136.0 us on top of MJH_XSWAP; bit-identical.
static int xswap_k_mask(int64_t workgroups) {
  static const int f = [] { const char* e = getenv("MJH_XSWAP_K"); return e ? (int)(strtol(e, nullptr, 0) & 0xfff) : 0x200; }();
  static const bool forced = getenv("MJH_XSWAP_K") != nullptr;
  return (forced || workgroups <= 4096) ? f : 0;
}

// the solver phase through the register solver: two (or, first tier of a small model, four) environments per wavefront
template <typename REAL>
int launch_sol2(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream, bool first_done = false) {  // first_done: the stage kernel has run the first tier (and marked what it left): only the second tier is launched
  a.env_begin = 0; a.env_count = a.B;
  const int64_t blocks = (a.B + 1) / 2;
  const int64_t grid = blocks < (int64_t)1 << 20 ? blocks : (int64_t)1 << 20;
  // (the kernels have no grid-stride loop: one launch per 2^20 workgroups of NS environments each; the marks-scanning second tier keeps its own walk)
#define CHUNKED(NS, LAUNCH) do { if (a.scan_marks) { a.env_begin = 0; a.env_count = a.B; const int64_t g_ = grid; (void)g_; LAUNCH(grid); } else for (int64_t e0_ = 0; e0_ < a.B; e0_ += NS * max_grid()) { const int64_t n_ = a.B - e0_ < NS * max_grid() ? a.B - e0_ : NS * max_grid(); a.env_begin = e0_; a.env_count = n_; LAUNCH((n_ + NS - 1) / NS); } a.env_begin = 0; a.env_count = a.B; } while (0)
#define GO(N, R) do { auto L_ = [&](int64_t g) { hipLaunchKernelGGL((mjh_sol2_kernel<REAL, N, R, 32>), dim3((unsigned)g), dim3(MJH_WAVE), lds, stream, a); }; CHUNKED(2, L_); } while (0)
#define GOW(N, R) do { auto L_ = [&](int64_t g) { if (a.M.solver == SOL_NEWTON) hipLaunchKernelGGL((mjh_sol2_kernel<REAL, N, R, 17>), dim3((unsigned)g), dim3(MJH_WAVE), lds, stream, a); else hipLaunchKernelGGL((mjh_sol2_kernel<REAL, N, R, 16>), dim3((unsigned)g), dim3(MJH_WAVE), lds, stream, a); }; CHUNKED(4, L_); } while (0)  /* 17: the Newton-only code of the four-per-wavefront tier */
  const int nd = a.M.nefc - a.M.nf - a.M.nl;
  bool second = true;
  const bool marks = m->sol2_tiers && a.cur.qacc != nullptr;  // the first tier marks what it leaves, the second scans the marks (needs the qacc leaf)
  a.mark_leftover = marks ? 1 : 0;
  if (m->sol2_tiers) {  // first tier: fewer row slots per lane, its own (smaller) arena
    a.off = m->off_tier;
    a.lds_reals = m->lds_tier / (int)sizeof(REAL);
    if (m->sol2_w16_rpl) {
      const int64_t blocks4 = (a.B + 3) / 4;
      const int64_t grid4 = blocks4 < (int64_t)1 << 20 ? blocks4 : (int64_t)1 << 20;
      a.row_lo = -1; a.row_hi = 16 * m->sol2_w16_rpl;
      const size_t lds = (size_t)(4 * m->lds_tier);
      const bool capped = MJH_SOL2_CAPS_ON && m->sol2_it_cap > 0 && !(a.flags & MJH_FLAG_FIXED_ITERATIONS) && a.cur.qacc;  // (a build with -DMJH_SOL2_CAPS; MJH_SOL2_ITCAP / MJH_SOL2_LSCAP then set the caps)
      a.it_cap = capped ? m->sol2_it_cap : 0; a.ls_cap = capped ? m->sol2_ls_cap : 0;
      if (first_done) {}
      else if (m->sol2_w16_nmax == 8) { if (m->sol2_w16_rpl == 2) GOW(8, 2); else GOW(8, 5); }
      else if (m->sol2_w16_nmax == 12) { if (m->sol2_w16_rpl == 2) GOW(12, 2); else GOW(12, 5); }
      else { if (m->sol2_w16_rpl == 2) GOW(16, 2); else GOW(16, 5); }
      a.it_cap = a.ls_cap = 0;
      if (capped) {  // the LDS solver takes what the packed tier left: more rows than it keeps, or a solve past the caps
        HIP_TRY(hipGetLastError());
        a.fallback_only = 1;
        const int rc = launch_phase<REAL, 4>(m, a, stream);
        a.fallback_only = 0;
        if (rc) return rc;
        g_timing.n -= (g_timing.on && g_timing.n > 0) ? 1 : 0;  // (launch_phase marked kernel 4: both launches belong under the solver phase's one mark below)
        second = false;
      } else {
        second = nd > a.row_hi;  // every environment fits the first tier otherwise
      }
    } else {
      a.row_lo = -1; a.row_hi = 32;
      const size_t lds = (size_t)(2 * m->lds_tier);
      if (m->sol2_nmax == 8) GO(8, 1); else if (m->sol2_nmax == 16) GO(16, 1); else GO(28, 1);
    }
    HIP_TRY(hipGetLastError());
    a.row_lo = a.row_hi;
  } else {
    a.row_lo = -1;
  }
  a.mark_leftover = 0;
  if (second) {
    a.row_hi = 0x7fffffff;
    a.off = m->off[5];
    a.lds_reals = m->lds_bytes[5] / (int)sizeof(REAL);
    a.scan_marks = marks ? 1 : 0;
    const int64_t sblocks = (a.B + MJH_WAVE - 1) / MJH_WAVE;
    const int64_t grid = marks ? (sblocks < 4096 ? sblocks : 4096) : (blocks < (int64_t)1 << 20 ? blocks : (int64_t)1 << 20);
    const size_t lds = (size_t)(2 * m->lds_bytes[5]);
    if (m->sol2_nmax == 8) { if (m->sol2_rpl == 1) GO(8, 1); else if (m->sol2_rpl == 2) GO(8, 2); else if (m->sol2_rpl == 4) GO(8, 4); else GO(8, 8); }
    else if (m->sol2_nmax == 16) { if (m->sol2_rpl == 1) GO(16, 1); else if (m->sol2_rpl == 2) GO(16, 2); else if (m->sol2_rpl == 4) GO(16, 4); else GO(16, 8); }
    else { if (m->sol2_rpl == 1) GO(28, 1); else GO(28, 2); }
    HIP_TRY(hipGetLastError());
    a.scan_marks = 0;
  }
#undef GO
#undef GOW
#undef CHUNKED
  if (!(first_done && !second)) timing_mark(stream, 9);  // both tiers under one mark: the solver phase (behind the stage kernel: the second tier alone, when there is one)
  return 0;
}

// one RK4 stage of a small Newton model in one launch (mjhModel::fuse_stage), then the solver's second tier for what the first left
template <typename REAL>
int launch_stage(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream, int parts = 7) {  // parts 6: the tail of a pass (constraint phase + first solver tier) behind launches of its own for the rest
  a.stage_parts = parts;
  const int keep_flags_ = a.flags;
  a.flags |= xswap_flags(a.B / 4);
  struct RestoreF_ { KArgs<REAL>& a; int f; ~RestoreF_() { a.flags = f; } } restore_f_{a, keep_flags_};
  a.xswap_k = xswap_k_mask(a.B / 4);
  static const int xswap_c = [] { const char* e = getenv("MJH_XSWAP_C"); return e ? (int)(strtol(e, nullptr, 0) & 0xfff) : 0x400; }();  // MJH_XSWAP_C=mask (0: off; see KArgs::xswap_c): measured ant 30.9 -> 31.1 M env-steps/s
  a.xswap_c = a.rk_stage == 0 ? xswap_c : 0;
  a.off = m->off_kcv; a.lds_reals = m->lds_kcv / (int)sizeof(REAL);
  a.off2 = m->off[2]; a.lds_reals2 = m->lds_bytes[2] / (int)sizeof(REAL);
  a.off3 = m->off_tier; a.lds_reals3 = m->lds_tier / (int)sizeof(REAL);
  a.row_lo = -1; a.row_hi = 16 * m->sol2_w16_rpl;
  a.mark_leftover = (m->sol2_tiers && a.cur.qacc != nullptr) ? 1 : 0;
  a.scan_marks = 0; a.it_cap = a.ls_cap = 0; a.fallback_only = 0;
  for (int64_t e0 = 0; e0 < a.B; e0 += 4 * max_grid()) {  // (no grid-stride loop in the kernels: one launch per 2^20 workgroups)
    const int64_t n = a.B - e0 < 4 * max_grid() ? a.B - e0 : 4 * max_grid();
    a.env_begin = e0; a.env_count = n;
    const size_t lds = (size_t)(parts == 7 ? m->lds_stage : m->lds_tail);
    const dim3 g((unsigned)(n / 4)), b(MJH_WAVE);
#define GOS(N, R) hipLaunchKernelGGL((mjh_sol2_kernel<REAL, N, R, 18>), g, b, lds, stream, a)
    if (m->sol2_w16_nmax == 8) { if (m->sol2_w16_rpl == 2) GOS(8, 2); else GOS(8, 5); }
    else if (m->sol2_w16_nmax == 12) { if (m->sol2_w16_rpl == 2) GOS(12, 2); else GOS(12, 5); }
    else { if (m->sol2_w16_rpl == 2) GOS(16, 2); else GOS(16, 5); }
#undef GOS
    HIP_TRY(hipGetLastError());
  }
  a.env_begin = 0; a.env_count = a.B;
  timing_mark(stream, parts == 7 ? 18 : 19);
  return launch_sol2<REAL>(m, a, stream, true);
}

// constraint stage + register solver + integrator in one launch (mjhModel::fuse_cs)
template <typename REAL>
int launch_cs(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream) {
  a.env_begin = 0; a.env_count = a.B;
  a.off = m->off_cs;
  a.lds_reals = m->lds_cs / (int)sizeof(REAL);
  a.row_lo = -1; a.row_hi = 0x7fffffff;
  a.mark_leftover = 0; a.scan_marks = 0; a.it_cap = a.ls_cap = 0;
  static const bool one_off = [] { const char* e = getenv("MJH_CS_ONE"); return e && e[0] == '0'; }();  // (A / B switch: MJH_CS_ONE=0 launches the generic instantiation for one-iteration models too)
  for (int64_t e0 = 0; e0 < a.B; e0 += 2 * max_grid()) {  // (no grid-stride loop in the kernels: one launch per 2^20 workgroups)
    const int64_t n = a.B - e0 < 2 * max_grid() ? a.B - e0 : 2 * max_grid(), grid = (n + 1) / 2;
    a.env_begin = e0; a.env_count = n;
    if (a.M.iterations == 1 && !one_off) hipLaunchKernelGGL((mjh_sol2_kernel<REAL, 28, 1, 35>), dim3((unsigned)grid), dim3(MJH_WAVE), (size_t)(2 * m->lds_cs), stream, a);  // opt.iterations == 1 (solver.py:534-535: the loop body runs exactly once): straight-line solver code
    else hipLaunchKernelGGL((mjh_sol2_kernel<REAL, 28, 1, 33>), dim3((unsigned)grid), dim3(MJH_WAVE), (size_t)(2 * m->lds_cs), stream, a);
  }
  a.env_begin = 0; a.env_count = a.B;
  HIP_TRY(hipGetLastError());
  timing_mark(stream, 14);
  return 0;
}

// the whole pass in one launch (mjhModel::fuse_all)
template <typename REAL>
int launch_all(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream) {
  a.env_begin = 0; a.env_count = a.B;
  a.off = m->off_kcv; a.off2 = m->off_cs;
  a.lds_reals = m->lds_all / (int)sizeof(REAL);
  a.row_lo = -1; a.row_hi = 0x7fffffff;
  a.mark_leftover = 0; a.scan_marks = 0; a.it_cap = a.ls_cap = 0;
  const int64_t blocks = (a.B + 1) / 2;
  const int64_t grid = blocks < (int64_t)1 << 20 ? blocks : (int64_t)1 << 20;
  const int keep_flags_ = a.flags;
  a.flags |= xswap_flags(grid);
  struct RestoreF_ { KArgs<REAL>& a; int f; ~RestoreF_() { a.flags = f; } } restore_f_{a, keep_flags_};
  a.xswap_k = xswap_k_mask(grid);
  if (a.M.iterations == 1) hipLaunchKernelGGL((mjh_sol2_kernel<REAL, 28, 1, 36>), dim3((unsigned)grid), dim3(MJH_WAVE), (size_t)(2 * m->lds_all), stream, a);
  else hipLaunchKernelGGL((mjh_sol2_kernel<REAL, 28, 1, 34>), dim3((unsigned)grid), dim3(MJH_WAVE), (size_t)(2 * m->lds_all), stream, a);
  HIP_TRY(hipGetLastError());
  timing_mark(stream, 16);
  return 0;
}

// convex narrow phase: one wavefront per (environment, pair); sensors: one per environment.  No grid-stride loops in the kernels: one launch per 2^22 items / 2^20 environments
template <typename REAL>
int launch_convex(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream) {
  const int64_t items = a.B * a.M.ncvxpair;
  for (int64_t i0 = 0; i0 < items; i0 += 4 * max_grid()) {
    const int64_t grid = items - i0 < 4 * max_grid() ? items - i0 : 4 * max_grid();
    a.env_begin = i0;  // (first ITEM of this launch)
    hipLaunchKernelGGL((mjh_convex_kernel<REAL>), dim3((unsigned)grid), dim3(MJH_WAVE), (size_t)m->cvx_lds_bytes, stream, a);
    HIP_TRY(hipGetLastError());
  }
  a.env_begin = 0;
  return 0;
}
template <typename REAL>
int launch_sensor_kernel(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream) {
  (void)m;
  // environments per wavefront: two when their slots fit the 64 lanes of the slot phase (MJH_SENSOR_EPW overrides).  Measured (MI355X, ant B = 16384, 27 slots and 104 ray tests per environment):
  // one 36.8 - 37.1 us, two 28.8 - 29.0, four 35.6 - 35.7 (a quarter of the workgroups: too few waves to hide the kernel's two dependent load chains) -- profiles/r05/notes.md
  static const int epw_env = [] { const char* e = getenv("MJH_SENSOR_EPW"); return e ? atoi(e) : 0; }();
  int epw = epw_env > 0 ? epw_env : (a.M.nsensordata > 0 ? MJH_WAVE / a.M.nsensordata : 1);
  epw = epw < 1 ? 1 : (epw_env > 0 ? (epw > 8 ? 8 : epw) : (epw > 2 ? 2 : epw));
  a.sns_epw = epw;
  const size_t lds = sizeof(double) * (size_t)(a.M.nrfq + 1) * (size_t)epw;
  for (int64_t e0 = 0; e0 < a.B; e0 += epw * max_grid()) {
    const int64_t n = a.B - e0 < epw * max_grid() ? a.B - e0 : epw * max_grid(), grid = (n + epw - 1) / epw;
    a.env_begin = e0; a.env_count = n;
    if (a.M.sns_full) hipLaunchKernelGGL((mjh_sensor_kernel<REAL, 1>), dim3((unsigned)grid), dim3(MJH_WAVE), lds, stream, a);
    else hipLaunchKernelGGL((mjh_sensor_kernel<REAL, 0>), dim3((unsigned)grid), dim3(MJH_WAVE), lds, stream, a);
    HIP_TRY(hipGetLastError());
  }
  a.env_begin = 0; a.env_count = a.B;
  return 0;
}

// Largest batch the two-wave form of kernel 13 serves in ONE round of its workgroups on the device `stream` belongs to: two wavefronts per workgroup at four per SIMD are 8
// workgroups per CU, and their arenas must fit the CU's LDS.  (CUs, LDS per CU) are read once per device (a model built while another device was current, or stepped
// under bench --gpus N before device selection, kept the wrong one-round threshold: 101 us against 62 for the ant).
static int64_t kcv2_limit(const mjhModel* m, hipStream_t stream) {
  if (m->kcv2_max_envs) return m->kcv2_max_envs;
  static int cus_of[64], lds_of[64];  // 0 = not read yet (plain ints: a race writes the same values twice)
  int dev = 0;
  if (hipStreamGetDevice(stream, &dev) != hipSuccess && hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cus_of[dev]) {
    int cus = 256, lds = 160 * 1024;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds < 64 * 1024) lds = 160 * 1024;
    lds_of[dev] = lds;
    cus_of[dev] = cus > 0 ? cus : 256;
  }
  int wgs = lds_of[dev] / (m->kcv2_per_wg > 0 ? m->kcv2_per_wg : 1);
  if (wgs > 8) wgs = 8;
  return (int64_t)cus_of[dev] * wgs * m->kcv2_envs_per_wg;
}

// one forward pass = the phases selected by `stages`
template <typename REAL>
int forward_pass(const mjhModel* m, KArgs<REAL>& a, hipStream_t stream) {
  int rc = 0;
  const int st = a.stages;
  if (m->fuse_all && (st & 0x7f) == 0x7f && a.B >= 2 && (a.B & 1) == 0 && a.B <= 2 * max_grid() /* one workgroup per pair: the kernel has no grid-stride loop */ && a.cur.efc_J && a.cur.efc_D && a.cur.efc_aref && a.cur.qM && a.cur.qLD) return launch_all<REAL>(m, a, stream);
  // MJH_DAG=1 (experiment): kinematics -> {crb / factor || velocity (+ sensors) || convex + constraint phase} -> solver, the two side branches on internal
  // streams forked from and joined into the caller's.  Needs the stand-alone kernels: run with MJH_FUSE_KV=0 (and so no kernel 13 / whole-pass kernel).  profiles/r05/notes.md has the numbers.
  // MJH_FUSE_STAGE0=0: stage 0 (which writes the returned Data in full) keeps its three launches.  Its sensors (they read leaves of the kinematics and velocity stages only) follow the
  // stage kernel: in an RK4 step nothing of stage 0 writes the returned state -- the final advance is stage 3's -- so the jointpos / ballquat readers of out.qpos race nothing.
  static const bool stage0_off = [] { const char* e = getenv("MJH_FUSE_STAGE0"); return e && e[0] == '0'; }();
  if (m->fuse_stage && a.rk_stage >= (stage0_off ? 1 : 0) && (st & 0x7f) == 0x7f && a.B >= 4 && (a.B & 3) == 0 && !a.sol_perm && a.cur.qM && a.cur.qLD && a.cur.qfrc_smooth && a.cur.contact_dist &&
      a.cur.efc_J && a.cur.efc_D && a.cur.efc_aref) {
    if ((rc = launch_stage<REAL>(m, a, stream))) return rc;
    if (a.M.nsensor > 0 && a.rk_stage == 0 && a.cur.sensordata) {
      if ((rc = launch_sensor_kernel<REAL>(m, a, stream))) return rc;
      timing_mark(stream, 11);
    }
    return 0;
  }
  static const bool dag = [] { const char* e = getenv("MJH_DAG"); return e && e[0] == '1'; }();
  if (dag && !m->fuse_kv && (st & 0x7f) == 0x7f && !g_timing.on && !g_stamps) {
    {
      std::lock_guard<std::mutex> lock(m->dag_mutex);
      if (!m->dag_ready) {
        for (int k = 0; k < 2; k++) { HIP_TRY(hipStreamCreateWithFlags(&m->dag_stream[k], hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&m->dag_done[k], hipEventDisableTiming)); }
        HIP_TRY(hipEventCreateWithFlags(&m->dag_fork, hipEventDisableTiming));
        m->dag_ready = true;
      }
    }
    if ((rc = launch_phase<REAL, 0>(m, a, stream))) return rc;
    HIP_TRY(hipEventRecord(m->dag_fork, stream));
    HIP_TRY(hipStreamWaitEvent(m->dag_stream[0], m->dag_fork, 0));
    HIP_TRY(hipStreamWaitEvent(m->dag_stream[1], m->dag_fork, 0));
    if ((rc = launch_phase<REAL, 1>(m, a, m->dag_stream[0]))) return rc;
    HIP_TRY(hipEventRecord(m->dag_done[0], m->dag_stream[0]));
    if ((rc = (a.M.has_fluid || a.M.has_gravcomp || a.M.ntendon > 0 || a.M.big) ? launch_phase<REAL, 5>(m, a, m->dag_stream[1]) : launch_phase<REAL, 3>(m, a, m->dag_stream[1]))) return rc;
    if (a.M.nsensor > 0 && a.rk_stage <= 0 && a.cur.sensordata) {
      if ((rc = launch_sensor_kernel<REAL>(m, a, m->dag_stream[1]))) return rc;
    }
    HIP_TRY(hipEventRecord(m->dag_done[1], m->dag_stream[1]));
    if (a.M.ncvxpair > 0) {
      if ((rc = launch_convex<REAL>(m, a, stream))) return rc;
    }
    if ((a.M.ncon > 0 || a.M.nefc > 0) && (rc = a.M.con_general ? launch_phase<REAL, 7>(m, a, stream) : (a.M.con_direct ? launch_phase<REAL, 8>(m, a, stream) : launch_phase<REAL, 2>(m, a, stream)))) return rc;
    HIP_TRY(hipStreamWaitEvent(stream, m->dag_done[0], 0));
    HIP_TRY(hipStreamWaitEvent(stream, m->dag_done[1], 0));
    if (m->sol2_nmax) return launch_sol2<REAL>(m, a, stream);
    return (a.M.nf > 0 || a.M.nft > 0 || a.M.ne > 0 || a.M.nlb > 0 || a.M.nlt > 0) ? launch_phase<REAL, 6>(m, a, stream) : launch_phase<REAL, 4>(m, a, stream);
  }
  const bool fused_kv = m->fuse_kv && (st & 0x70);  // the velocity phase is asked for: it rides with the kinematics (it needs nothing of CRB / CON)
  const bool fused_kcv = fused_kv && m->fuse_kcv && (st & 0x7e) && a.B <= m->kcv_max_envs;  // ... and so does the crb / factor stage (small models)
  if ((st & 0x7f) && (rc = fused_kcv ? ((m->kcv2 && a.B <= kcv2_limit(m, stream)) ? launch_phase<REAL, 17>(m, a, stream) : launch_phase<REAL, 13>(m, a, stream)) : (fused_kv ? launch_phase<REAL, 12>(m, a, stream) : launch_phase<REAL, 0>(m, a, stream)))) return rc;
  if ((st & 0x7c) && a.M.ncvxpair > 0) {  // convex narrow phase: one wave per (environment, pair); needs only the geom frames of PH_KIN
    if ((rc = launch_convex<REAL>(m, a, stream))) return rc;
    timing_mark(stream, 10);
  }
  if ((st & 0x7e) && !fused_kcv && (rc = launch_phase<REAL, 1>(m, a, stream))) return rc;
  const bool fused_cs = m->fuse_cs && (st & 0x7c) && (st & 0x40) && a.cur.efc_J && a.cur.efc_D && a.cur.efc_aref;  // the whole tail of the pass is asked for: constraint stage and solve share a kernel
  const bool fused_tail = !fused_cs && m->fuse_tail && (st & 0x7c) == 0x7c && a.B >= 4 && (a.B & 3) == 0 && !a.sol_perm && a.cur.qM && a.cur.qLD && a.cur.qfrc_smooth && a.cur.contact_dist &&
                          a.cur.efc_J && a.cur.efc_D && a.cur.efc_aref && !(a.M.has_fluid || a.M.has_gravcomp || a.M.ntendon > 0 || a.M.big) && fused_kv;  // (the velocity stage has run: it rode with the kinematics)
  if (fused_tail) {  // small Newton models: constraint phase + first solver tier + integrator in one launch (their sensors read leaves of the kinematics / velocity stages only: first)
    if ((st & 0x40) && a.M.nsensor > 0 && a.rk_stage <= 0 && a.cur.sensordata) {
      if ((rc = launch_sensor_kernel<REAL>(m, a, stream))) return rc;
      timing_mark(stream, 11);
    }
    return launch_stage<REAL>(m, a, stream, 6);
  }
  if (!fused_cs && (st & 0x7c) && (a.M.ncon > 0 || a.M.nefc > 0) &&
      (rc = a.M.con_general ? launch_phase<REAL, 7>(m, a, stream) : (a.M.con_direct ? launch_phase<REAL, 8>(m, a, stream) : launch_phase<REAL, 2>(m, a, stream)))) return rc;
  if ((st & 0x70) && !fused_kv && (rc = (a.M.has_fluid || a.M.has_gravcomp || a.M.ntendon > 0 || a.M.big) ? launch_phase<REAL, 5>(m, a, stream) : launch_phase<REAL, 3>(m, a, stream))) return rc;
  const bool want_sensors = (st & 0x40) && a.M.nsensor > 0 && a.rk_stage <= 0 && a.cur.sensordata;  // needs only the leaves of KIN and VEL
  auto launch_sensors = [&](hipStream_t s_) -> int {
#ifdef MJH_SENSOR_ABLATE
    static const int abl = [] { const char* e = getenv("MJH_SENSOR_ABLATE"); return e ? atoi(e) : 0; }();
    const int keep_flags = a.flags;
    a.flags |= abl << 8;
    struct Restore { KArgs<REAL>& a; int f; ~Restore() { a.flags = f; } } restore_{a, keep_flags};
#endif
    return launch_sensor_kernel<REAL>(m, a, s_);
  };
  if (want_sensors) {  // (round 4's opt-in side stream for this launch is gone: it raced the integrator tail's write of out.qpos, which jointpos / ballquat sensors read, and measured no gain -- profiles/r04/notes.md)
    if ((rc = launch_sensors(stream))) return rc;
    timing_mark(stream, 11);
  }
  if (fused_cs) return launch_cs<REAL>(m, a, stream);
  if ((st & 0x60) && m->sol2_nmax) return launch_sol2<REAL>(m, a, stream);
  if ((st & 0x60) && (rc = (a.M.nf > 0 || a.M.nft > 0 || a.M.ne > 0 || a.M.nlb > 0 || a.M.nlt > 0) ? launch_phase<REAL, 6>(m, a, stream) : launch_phase<REAL, 4>(m, a, stream))) return rc;  // 0x20: _acceleration's solve lives at the head of the solver phase
  return 0;
}

template <typename REAL>
int run_launches_one(const mjhModel* m, const DevModel<REAL>& M, const mjhData* in, mjhData* out, void* work, int64_t B, int flags, int do_step, int stages, void* stream) {
  if (B <= 0) return 0;
  static_assert(sizeof(DevData<REAL>) == sizeof(mjhData), "DevData must mirror mjhData");
  static_assert(sizeof(KArgs<REAL>) <= 4096, "kernel arguments exceed the 4 KiB kernarg segment");
  KArgs<REAL> a;
  memset(&a, 0, sizeof(a));
  timing_begin((hipStream_t)stream);
  a.M = M;
  memcpy(&a.in, in, sizeof(a.in));
  DevData<REAL> fin;
  memcpy(&fin, out, sizeof(fin));
  a.fin = state_of(fin);
  a.cur = fin;
  a.B = B; a.flags = flags; a.do_step = do_step; a.stages = do_step ? MJH_STAGE_ALL : stages;
  a.rk_stage = -1; a.state_from_cur = 0;
  a.warm_src = a.in.qacc_warmstart;
  a.stamps = g_stamps;
  if (!a.in.qpos || !a.in.qvel) return fail(-22, "in.qpos and in.qvel are required");
  if ((a.stages & 0x60) && !(fin.qM && fin.qLD && fin.qfrc_smooth)) return fail(-22, "the solver phase reads out.qM / out.qLD: both leaves are required");
  if (M.ncon > 0 && M.nefc > 0 && (a.stages & 0x40) && !fin.contact_dist) return fail(-22, "the solver phase reads out.contact_dist (active-contact row compaction): the leaf is required");
  if (M.ncvxpair > 0 && (a.stages & 0x7c) && !(fin.contact_dist && fin.contact_pos && fin.contact_frame && fin.geom_xpos && fin.geom_xmat))
    return fail(-22, "models with convex pairs need out.geom_xpos/geom_xmat and out.contact_dist/pos/frame");
  hipStream_t s = (hipStream_t)stream;
  REAL* w = (REAL*)work;
  if (m->sort_reals > 0) {
    if (work) {
      int* perm = (int*)work;
      int* key = perm + B;
      if ((a.stages & 0x40) && B >= 8) {  // this step's list from last step's keys (a forward pass without a solve leaves both alone)
        hipLaunchKernelGGL(mjh_sort_kernel, dim3(1), dim3(1024), 0, s, (const int*)key, perm, (long long)B);
        HIP_TRY(hipGetLastError());
        timing_mark(s, 15);
        if (B <= 4 * max_grid()) { a.sol_perm = perm; a.sol_key = key; }  // (the list indexes the whole batch: one launch of the packed tier)
      }
    }
    w += m->sort_reals * B;
  }
  if (m->hs_reals > 0) {  // (without a workspace the solver reads the leaves, as it always did)
    if (work) { a.hs = w; a.hs_reals = (int)m->hs_reals; }
    w += m->hs_reals * B;
  }
  if (m->cand_reals > 0 && (a.stages & 0x7c)) {  // max_contact_points over box / mesh pairs: the candidates live at the head of the workspace
    if (!work) return fail(-22, "max_contact_points with box / mesh pairs needs a workspace of mjh_model_work_bytes(m) * B bytes");
    a.cand = w;
    w += m->cand_reals * B;
  }
  if (!do_step || M.integrator == INT_EULER) return forward_pass<REAL>(m, a, s);

  // ---- RK4 (forward.py:331-370): stage 0 computes the returned Data; stages 1..3 run in the workspace Data ----
  if (!work) return fail(-22, "RK4 needs a workspace of mjh_model_work_bytes(m) * B bytes");
  DevData<REAL> scr;
  memset(&scr, 0, sizeof(scr));
  {
    REAL** slots = reinterpret_cast<REAL**>(&scr);
    const char* names[] = {
#define X(n) #n,
        MJH_DATA_REALS(X)
#undef X
    };
    for (size_t i = 0; i < m->leaf_count.size(); i++)
      if (is_stage_leaf(names[i], M.ncvxpair > 0, M.has_fluid != 0, M.ne > 0, M.topk != 0)) {
        // (kernel 13 keeps cinert / xipos in its arena from the kinematics to the crb and velocity stages: no later LAUNCH of a stage reads their workspace copies -- left NULL, the stores are skipped)
        if (!(m->fuse_kcv && B <= m->kcv_max_envs && (!strcmp(names[i], "cinert") || !strcmp(names[i], "xipos")))) slots[i] = w;
        w += m->leaf_count[i] * B;
      }
  }
  a.W.qvel0 = w; w += (int64_t)M.nv * B;
  a.W.kqvel = w; w += (int64_t)M.nv * B;
  a.W.sum_qvel = w; w += (int64_t)M.nv * B;
  a.W.sum_qacc = w; w += (int64_t)M.nv * B;
  a.W.act0 = w; w += (int64_t)M.na * B;
  a.W.sum_actdot = w; w += (int64_t)M.na * B;
  a.nxt = state_of(scr);
  int rc = 0;
  for (int stage = 0; stage < 4; stage++) {
    a.rk_stage = stage;
    if (stage == 0) {
      a.cur = fin; a.state_from_cur = 0; a.warm_src = a.in.qacc_warmstart;
    } else {
      a.cur = scr; a.state_from_cur = 1;
      a.warm_src = (stage == 1) ? fin.qacc_warmstart : scr.qacc_warmstart;  // solver.py:547-552 writes it every pass
    }
    if ((rc = forward_pass<REAL>(m, a, s))) return rc;
  }
  return 0;
}

int split_ways() {
  // MJH_SPLIT=n: the batch is cut into n contiguous slices whose launch sequences run on n internal streams, forked from and
  // joined back into the caller's stream with events.  The phases of different slices then overlap on the CUs (a register-bound
  // phase of one slice fills the wave slots an LDS-bound phase of another leaves idle).  Default 1 (profiles/r01/notes.md).
  static const int n = [] { const char* e = getenv("MJH_SPLIT"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  return n;
}

// the Data view of environments [begin, ...): every non-null leaf pointer advanced by begin * (elements per environment)
template <typename REAL>
void offset_data(const mjhModel* m, const DevModel<REAL>& M, const mjhData* src, mjhData* dst, int64_t begin) {
  memcpy(dst, src, sizeof(*dst));
  unsigned char** p = reinterpret_cast<unsigned char**>(dst);
  const size_t nreal = m->leaf_count.size();
  for (size_t i = 0; i < nreal; i++) if (p[i]) p[i] += (size_t)begin * (size_t)m->leaf_count[i] * sizeof(REAL);
  const int64_t ncon = M.ncon, neq = M.neq;
  const int64_t int_bytes[] = {4 * ncon, 4 * neq, 8 * ncon, 8 * ncon, 16 * ncon, 8 * ncon};  // contact_dim, eq_active | geom1, geom2, geom, efc_address
  for (size_t k = 0; k < 6; k++) if (p[nreal + k]) p[nreal + k] += (size_t)begin * (size_t)int_bytes[k];
  const int64_t extra_reals[] = {6 * (int64_t)M.nbody, 6 * (int64_t)M.nbody, 3 * (int64_t)M.nbody, 3 * (int64_t)M.nbody};  // MJH_DATA_EXTRA_IN: cacc, cfrc_int, subtree_linvel, subtree_angmom (input-only, may be NULL)
  static_assert(sizeof(mjhData) / sizeof(void*) >= 10, "mjhData: leaves + six integer leaves + four input-only leaves");
  for (size_t k = 0; k < 4; k++) if (p[nreal + 6 + k]) p[nreal + 6 + k] += (size_t)begin * (size_t)extra_reals[k] * sizeof(REAL);
}

template <typename REAL>
int run_launches(const mjhModel* m, const DevModel<REAL>& M, const mjhData* in, mjhData* out, void* work, int64_t B, int flags, int do_step, int stages, void* stream) {
  const int ways = split_ways();
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (ways <= 1 || B < 64 * ways || capturing) return run_launches_one<REAL>(m, M, in, out, work, B, flags, do_step, stages, stream);
  std::lock_guard<std::mutex> lock(m->split_mutex);
  if (!m->split_ready) {
    for (int k = 0; k < 4; k++) {
      HIP_TRY(hipStreamCreateWithFlags(&m->split_stream[k], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&m->split_done[k], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&m->split_fork, hipEventDisableTiming));
    m->split_ready = true;
  }
  HIP_TRY(hipEventRecord(m->split_fork, s));
  const int64_t per = ((B + ways - 1) / ways + 1) & ~(int64_t)1;  // even slices keep the two-environments-per-wave phases paired
  const size_t work_env_bytes = (size_t)m->work_reals * sizeof(REAL);
  int rc = 0;
  for (int k = 0; k < ways; k++) {
    const int64_t begin = (int64_t)k * per, count = (begin + per <= B) ? per : B - begin;
    if (count <= 0) break;
    mjhData in_k, out_k;
    offset_data<REAL>(m, M, in, &in_k, begin);
    offset_data<REAL>(m, M, out, &out_k, begin);
    HIP_TRY(hipStreamWaitEvent(m->split_stream[k], m->split_fork, 0));
    void* work_k = work ? (void*)((unsigned char*)work + (size_t)begin * work_env_bytes) : nullptr;
    if ((rc = run_launches_one<REAL>(m, M, &in_k, &out_k, work_k, count, flags, do_step, stages, (void*)m->split_stream[k]))) break;
    HIP_TRY(hipEventRecord(m->split_done[k], m->split_stream[k]));
    HIP_TRY(hipStreamWaitEvent(s, m->split_done[k], 0));
  }
  return rc;
}

unsigned long long fnv1a(unsigned long long h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

bool graphs_enabled() {
  // opt-in (MJH_GRAPHS=1): measured on MI355X the replay saves ~3 % of wall time at B <= 1024 (the step is bound by the
  // latency of its dependent kernels there, not by the host) and costs 1-2 % at B = 4096 (profiles/r01/notes.md)
  static const bool on = [] { const char* e = getenv("MJH_GRAPHS"); return e && e[0] == '1'; }();
  return on && !g_stamps && !g_timing.on;  // per-launch events cannot be recorded into a replayed graph
}

template <typename REAL>
int run(const mjhModel* m, const DevModel<REAL>& M, const mjhData* in, mjhData* out, void* work, int64_t B, int flags, int do_step, int stages, void* stream) {
  if (B <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (!graphs_enabled() || (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone))
    return run_launches<REAL>(m, M, in, out, work, B, flags, do_step, stages, stream);  // the caller is capturing a graph of their own
  unsigned long long key = 1469598103934665603ull;
  key = fnv1a(key, in, sizeof(*in));
  key = fnv1a(key, out, sizeof(*out));
  key = fnv1a(key, &work, sizeof(work));
  const long long scal[4] = {(long long)B, flags, do_step, stages};
  key = fnv1a(key, scal, sizeof(scal));
  std::lock_guard<std::mutex> lock(m->graph_mutex);
  for (auto& g : m->graphs)
    if (g.key == key) {
      g.last_use = ++m->graph_clock;
      HIP_TRY(hipGraphLaunch(g.exec, s));
      return 0;
    }
  if (!m->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&m->capture_stream, hipStreamNonBlocking));
  if (hipStreamBeginCapture(m->capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return run_launches<REAL>(m, M, in, out, work, B, flags, do_step, stages, stream);
  }
  const int rc = run_launches<REAL>(m, M, in, out, work, B, flags, do_step, stages, (void*)m->capture_stream);
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(m->capture_stream, &graph);
  if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  hipGraphExec_t exec = nullptr;
  if (ec != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    return run_launches<REAL>(m, M, in, out, work, B, flags, do_step, stages, stream);
  }
  (void)hipGraphDestroy(graph);
  if (m->graphs.size() >= 16) {  // ping-pong loops need two entries; keep a handful and evict the least recently used
    size_t lru = 0;
    for (size_t i = 1; i < m->graphs.size(); i++) if (m->graphs[i].last_use < m->graphs[lru].last_use) lru = i;
    (void)hipGraphExecDestroy(m->graphs[lru].exec);
    m->graphs.erase(m->graphs.begin() + (long)lru);
  }
  m->graphs.push_back({key, exec, ++m->graph_clock});
  HIP_TRY(hipGraphLaunch(exec, s));
  return 0;
}

}  // namespace

extern "C" {

int mjh_model_create(const mjhModelDesc* desc, int dtype, mjhModel** out) {
  if (!desc || !out) return fail(-22, "null argument");
  if (desc->abi_version != MJH_ABI_VERSION) return fail(-22, "mjhModelDesc.abi_version mismatch");
  if (dtype != MJH_F64 && dtype != MJH_F32) return fail(-22, "dtype must be MJH_F64 or MJH_F32");
  if (desc->integrator != INT_EULER && desc->integrator != INT_RK4) return fail(-38, "integrator not implemented");
  if (desc->solver != SOL_CG && desc->solver != SOL_NEWTON) return fail(-38, "solver not implemented");
  mjhModel* m = new mjhModel();
  m->dtype = dtype;
  int rc = dtype == MJH_F64 ? build<double>(desc, m, m->m64) : build<float>(desc, m, m->m32);
  if (rc != 0) { delete m; return rc; }
  *out = m;
  return 0;
}

void mjh_model_destroy(mjhModel* m) {
  if (!m) return;
  for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.exec);
  if (m->capture_stream) (void)hipStreamDestroy(m->capture_stream);
  if (m->dag_ready) { for (int k = 0; k < 2; k++) { (void)hipStreamDestroy(m->dag_stream[k]); (void)hipEventDestroy(m->dag_done[k]); } (void)hipEventDestroy(m->dag_fork); }
  if (m->split_ready) {
    for (int k = 0; k < 4; k++) { (void)hipStreamDestroy(m->split_stream[k]); (void)hipEventDestroy(m->split_done[k]); }
    (void)hipEventDestroy(m->split_fork);
  }
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

int mjh_forward(const mjhModel* m, const mjhData* in, mjhData* out, void* work, int64_t B, int stages, int flags, void* stream) {
  if (!m || !in || !out) return fail(-22, "null argument");
  return m->dtype == MJH_F64 ? run<double>(m, m->m64, in, out, work, B, flags, 0, stages, stream)
                             : run<float>(m, m->m32, in, out, work, B, flags, 0, stages, stream);
}

int mjh_step(const mjhModel* m, const mjhData* in, mjhData* out, void* work, int64_t B, int flags, void* stream) {
  if (!m || !in || !out) return fail(-22, "null argument");
  return m->dtype == MJH_F64 ? run<double>(m, m->m64, in, out, work, B, flags, 1, MJH_STAGE_ALL, stream)
                             : run<float>(m, m->m32, in, out, work, B, flags, 1, MJH_STAGE_ALL, stream);
}

int mjh_reset_where(const mjhModel* m, mjhData* d, const mjhData* d0, const unsigned char* mask, const void* qpos_rows,
                    const void* qvel_rows, int64_t B, void* stream) {
  if (!m || !d || !d0 || !mask) return fail(-22, "null argument");
  if (B <= 0) return 0;
  const int rw = m->dtype == MJH_F64 ? 2 : 1;  // 4-byte words per real
  ResetArgs a;
  memset(&a, 0, sizeof(a));
  a.mask = mask;
  void* const* dp = reinterpret_cast<void* const*>(d);
  void* const* sp = reinterpret_cast<void* const*>(d0);
  const char* names[] = {
#define X(n) #n,
      MJH_DATA_REALS(X) MJH_DATA_I32(X) MJH_DATA_I64(X)
#undef X
  };
  const size_t nreal = m->leaf_count.size();
  const int64_t ncon = m->dtype == MJH_F64 ? m->m64.ncon : m->m32.ncon;
  const int64_t neq = m->dtype == MJH_F64 ? m->m64.neq : m->m32.neq;
  const int64_t int_words[] = {ncon /* contact_dim */, neq /* eq_active */, 2 * ncon, 2 * ncon, 4 * ncon, 2 * ncon /* geom1, geom2, geom, efc_address (int64) */};
  const size_t nall = sizeof(names) / sizeof(names[0]);
  if (nall != nreal + 6 || nall > MJH_RESET_MAX_LEAVES) return fail(-22, "mjh_reset_where: leaf table out of sync with mjhData");
  for (size_t i = 0; i < nall; i++) {
    const int64_t words = i < nreal ? m->leaf_count[i] * rw : int_words[i - nreal];
    if (!dp[i] || words == 0) continue;  // leaf not carried by the caller's Data
    if (!sp[i]) return fail(-22, std::string("mjh_reset_where: d0.") + names[i] + " is null but d." + names[i] + " is not");
    ResetLeaf& L = a.leaf[a.nleaf++];
    L.dst = (unsigned*)dp[i]; L.src = (const unsigned*)sp[i]; L.words = (int)words;
    const void* rows = !strcmp(names[i], "qpos") ? qpos_rows : !strcmp(names[i], "qvel") ? qvel_rows : nullptr;
    if (rows) { L.src = (const unsigned*)rows; L.src_stride = (int)words; }
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mjh_reset_kernel, dim3((unsigned)B), dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

/* measurement aid (bench.py): while enabled, every kernel launch of mjh_step / mjh_forward is bracketed by HIP events recorded
 * on the launch stream; mjh_debug_phase_times() waits for the last call and returns the elapsed time of each of its launches.
 * Not for concurrent callers; costs one hipEventRecord per launch while on. */
int mjh_debug_phase_timing(int enable) {
  if (enable && !g_timing.on) {
    for (int i = 0; i <= MJH_TIMING_MAX; i++) HIP_TRY(hipEventCreate(&g_timing.ev[i]));
    g_timing.on = true; g_timing.n = 0;
  } else if (!enable && g_timing.on) {
    for (int i = 0; i <= MJH_TIMING_MAX; i++) (void)hipEventDestroy(g_timing.ev[i]);
    g_timing.on = false; g_timing.n = 0;
  }
  return 0;
}
int mjh_debug_phase_times(float* ms, int* ids, int max) {
  if (!g_timing.on) return fail(-22, "mjh_debug_phase_timing(1) has not been called");
  const int n = g_timing.n < max ? g_timing.n : max;
  if (n > 0) HIP_TRY(hipEventSynchronize(g_timing.ev[g_timing.n]));
  for (int i = 0; i < n; i++) { HIP_TRY(hipEventElapsedTime(&ms[i], g_timing.ev[i], g_timing.ev[i + 1])); ids[i] = g_timing.id[i]; }
  return n;
}

/* diagnostic (-DMJH_STAMPS builds): device buffer [B, 128] of uint64 receiving in-kernel clock stamps */
void mjh_debug_set_stamps(void* dev_ptr) { g_stamps = (unsigned long long*)dev_ptr; }

int mjh_model_leaf_counts(const mjhModel* m, int64_t* counts, int max) {
  if (!m || !counts) return fail(-22, "null argument");
  const int64_t ncon = m->dtype == MJH_F64 ? m->m64.ncon : m->m32.ncon, neq = m->dtype == MJH_F64 ? m->m64.neq : m->m32.neq;
  std::vector<int64_t> all(m->leaf_count);
  const int64_t ints[] = {ncon /* contact_dim */, neq /* eq_active */, ncon, ncon, 2 * ncon, ncon /* geom1, geom2, geom, efc_address */};
  all.insert(all.end(), ints, ints + 6);
  for (int i = 0; i < (int)all.size() && i < max; i++) counts[i] = all[i];
  return (int)all.size();
}

static thread_local bool g_io_inner = false;
static thread_local int g_io_stage = 0;  // 2: the accounts asked for are those of RK4 stages 1..3 alone (the stage kernel's parts)
static int mjh_model_kernel_io_impl(const mjhModel* m, int kernel, int64_t* read_write_bytes) {
  const bool prev = g_io_inner;
  g_io_inner = true;
  const int rc = mjh_model_kernel_io(m, kernel, read_write_bytes);
  g_io_inner = prev;
  return rc;
}
int mjh_model_kernel_io(const mjhModel* m, int kernel, int64_t* read_write_bytes) {
  if (!m || !read_write_bytes) return fail(-22, "null argument");
  if ((kernel == 9) != (m->sol2_nmax != 0) && (kernel == 9 || kernel == 4 || kernel == 6)) return -2;  // the solver phase runs as ONE of kernels 4 / 6 / 9
  if (kernel == 17) { if (!m->kcv2) return -2; const bool prev = g_io_inner; g_io_inner = true; const int rc = mjh_model_kernel_io(m, 13, read_write_bytes); g_io_inner = prev; return rc; }  // the two-wave form of kernel 13 moves the same bytes
  if (kernel == 13 && !m->fuse_kcv) return -2;  // (a model with kernel 13 reports both accounts: which one a step launches depends on the batch)
  if ((kernel == 12) != (m->fuse_kv != 0) && (kernel == 12 || kernel == 0 || kernel == 3)) return -2;  // kinematics + velocity: ONE kernel (12) or two (0, 3)
  const bool f64 = m->dtype == MJH_F64;
  const bool rk4 = (f64 ? m->m64.integrator : m->m32.integrator) == INT_RK4;
  auto io_of = [&](int k, int64_t* a) -> int {  // RK4: the mean over the four stage launches of a step (the sensor kernel runs once)
    int64_t b[2] = {0, 0};
    int rc = f64 ? mjh_kernel_io<double>(m->m64, k, 1, &a[0], &a[1]) : mjh_kernel_io<float>(m->m32, k, 1, &a[0], &a[1]);
    if (rc == 0 && rk4 && k != 11) {
      rc = f64 ? mjh_kernel_io<double>(m->m64, k, 2, &b[0], &b[1]) : mjh_kernel_io<float>(m->m32, k, 2, &b[0], &b[1]);
      if (g_io_stage == 2) { a[0] = b[0]; a[1] = b[1]; }                      // the stage kernel: launches of stages 1..3 only
      else if (m->fuse_stage) {}                                               // ... whose model launches kernels 13 / 8 / 9 in stage 0 only
      else { a[0] = (a[0] + 3 * b[0]) / 4; a[1] = (a[1] + 3 * b[1]) / 4; }
    }
    return rc;
  };
  int64_t a[2] = {0, 0};
  if (kernel == 16) {  // the whole pass in one launch: the accounts of kernels 13 and 14 (what the second half reads of the first it still reads from the leaves)
    if (!m->fuse_all) return -2;
    int64_t a13[2] = {0, 0}, a14[2] = {0, 0};
    const mjhModel* mm = m;
    if (mjh_model_kernel_io_impl(mm, 13, a13) != 0 || mjh_model_kernel_io_impl(mm, 14, a14) != 0) return -2;
    read_write_bytes[0] = a13[0] + a14[0]; read_write_bytes[1] = a13[1] + a14[1];
    return 0;
  }
  if (kernel == 18) {  // one RK4 stage in one launch: the stage accounts of kernel 13, the constraint phase and the register solver (what a part reads of the previous one it still reads from the workspace leaves)
    if (!m->fuse_stage) return -2;
    int64_t a13[2] = {0, 0}, a8[2] = {0, 0}, a9[2] = {0, 0};
    const int prev = g_io_stage;
    g_io_stage = 2;
    const int rc = (mjh_model_kernel_io_impl(m, 13, a13) != 0 || mjh_model_kernel_io_impl(m, 8, a8) != 0 || mjh_model_kernel_io_impl(m, 9, a9) != 0) ? -2 : 0;
    g_io_stage = prev;
    if (rc) return rc;
    read_write_bytes[0] = a13[0] + a8[0] + a9[0]; read_write_bytes[1] = a13[1] + a8[1] + a9[1];
    return 0;
  }
  if (kernel == 19) {  // the tail of a pass in one launch: the accounts of the constraint phase and the register solver
    if (!m->fuse_tail) return -2;
    int64_t a8[2] = {0, 0}, a9[2] = {0, 0};
    if (mjh_model_kernel_io_impl(m, 8, a8) != 0 || mjh_model_kernel_io_impl(m, 9, a9) != 0) return -2;
    read_write_bytes[0] = a8[0] + a9[0]; read_write_bytes[1] = a8[1] + a9[1];
    return 0;
  }
  if ((kernel == 13 || kernel == 14) && m->fuse_all && !g_io_inner) return -2;
  if (kernel == 14) {  // constraint stage + register solver: the two accounts minus what stays in the arena between them -- the dense rows of efc_J, efc_D / efc_aref, one entry per single-column row, qvel
    if (!m->fuse_cs) return -2;
    int64_t k2[2] = {0, 0}, k4[2] = {0, 0};
    if (io_of(2, k2) != 0 || io_of(4, k4) != 0) return -2;
    const int64_t R = f64 ? 8 : 4, nv = f64 ? m->m64.nv : m->m32.nv, nefc = f64 ? m->m64.nefc : m->m32.nefc, nl = f64 ? m->m64.nl : m->m32.nl;
    read_write_bytes[0] = k2[0] + k4[0] - ((nefc - nl) * nv + 2 * nefc + nl + nv) * R;
    read_write_bytes[1] = k2[1] + k4[1];
    return 0;
  }
  if ((kernel == 2 || kernel == 9) && m->fuse_cs) return -2;  // (a full pass of this model launches kernel 14 instead)
  if (kernel == 12 || kernel == 13) {  // the accounts of the fused stages minus what stays in the arena between them: qpos, cdof, cinert, subtree_com, xipos are not read back (13: nor cinert, cdof by the crb stage)
    int64_t k0[2] = {0, 0}, k3[2] = {0, 0}, k1[2] = {0, 0};
    if (io_of(0, k0) != 0 || io_of(3, k3) != 0 || (kernel == 13 && io_of(1, k1) != 0)) return -2;
    const int64_t R = f64 ? 8 : 4, nq = f64 ? m->m64.nq : m->m32.nq, nv = f64 ? m->m64.nv : m->m32.nv, nb = f64 ? m->m64.nbody : m->m32.nbody;
    read_write_bytes[0] = k0[0] + k3[0] - (nq + 6 * nv + 10 * nb + 3 * nb + 3 * nb) * R + (kernel == 13 ? k1[0] - (10 * nb + 6 * nv) * R : 0);
    read_write_bytes[1] = k0[1] + k3[1] + (kernel == 13 ? k1[1] : 0);
    return 0;
  }
  const int k = kernel == 9 ? 4 : kernel;  // the register solver moves the same leaves as the plain LDS solver
  if (io_of(k, a) != 0) return -2;  // -2: this model's step does not launch that kernel
  read_write_bytes[0] = a[0]; read_write_bytes[1] = a[1];
  return 0;
}

int64_t mjh_model_work_bytes(const mjhModel* m) { return m ? m->work_reals * (m->dtype == MJH_F64 ? 8 : 4) : 0; }
int mjh_model_lds_bytes(const mjhModel* m, int phase) {  // phase 5: the register solver's arena; 16 / 17 / 18: its packed first tier, the fused kinematics + crb + velocity kernel, the fused constraint + solver kernel (per environment)
  if (!m) return 0;
  if (phase == 16) return m->lds_tier;
  if (phase == 17) return m->lds_kcv;
  if (phase == 18) return m->lds_cs;
  if (phase == 19) return m->lds_kcv2;
  return (phase >= 0 && phase < MJH_NARENA) ? m->lds_bytes[phase] : 0;
}
const char* mjh_last_error(void) { return g_err.c_str(); }
int mjh_abi_version(void) { return MJH_ABI_VERSION; }

#define STR_(x) #x
const char* mjh_data_fields(void) {
  static const char s[] =
#define X(n) STR_(n) ","
      MJH_DATA_REALS(X) MJH_DATA_I32(X) MJH_DATA_I64(X)
#undef X
      ;
  static std::string t = std::string(s).substr(0, sizeof(s) - 2);
  return t.c_str();
}
const char* mjh_data_extra_fields(void) {  // the input-only leaves that trail mjhData (MJH_DATA_EXTRA_IN)
  static const char s[] =
#define X(n) STR_(n) ","
      MJH_DATA_EXTRA_IN(X)
#undef X
      ;
  static std::string t = std::string(s).substr(0, sizeof(s) - 2);
  return t.c_str();
}
int mjh_sizeof_data(void) { return (int)sizeof(mjhData); }
const char* mjh_model_fields(void) {
  static const char s[] =
#define X(n) STR_(n) ","
      MJH_MODEL_INTS(X) MJH_MODEL_REALS(X) MJH_MODEL_INT_ARRAYS(X) MJH_MODEL_REAL_ARRAYS(X)
#undef X
      ;
  static std::string t = std::string(s).substr(0, sizeof(s) - 2);
  return t.c_str();
}

}  // extern "C"
