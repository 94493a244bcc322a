// mjhip.hip -- C ABI of the MI355X-native stepper (see include/mjhip.h) and the host side of the launch.
//
// Host work per call: fill two kernarg structs and one hipLaunchKernelGGL on the caller's stream.  No
// allocation, no synchronisation, no PyTorch types: the binding (ctypes) passes raw device pointers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mjh_kernels.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(x)                                                                                         \
  do {                                                                                                     \
    hipError_t e_ = (x);                                                                                   \
    if (e_ != hipSuccess) return fail(-5, std::string(#x) + ": " + hipGetErrorString(e_));                 \
  } while (0)

struct mjhModel {
  int dtype;
  void* blob;          // device allocation holding every table
  size_t blob_bytes;
  int lds_bytes;
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
  if (nb > 64 || nv > 64) return fail(-22, "this build keeps ancestor sets in 64-bit masks: nbody and nv must be <= 64");
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
  // DFS order check: every body in [b, sub_end[b]) must descend from b
  for (int b = 1; b < nb; b++)
    for (int c = b + 1; c < sub_end[b]; c++) {
      int a = c;
      while (a > b) a = d->body_parentid[a];
      if (a != b) return fail(-22, "bodies are not in depth-first order");
    }
  std::vector<unsigned long long> body_dofmask(nb, 0ull), dof_ancmask(nv, 0ull);
  for (int dd = 0; dd < nv; dd++) {
    int a = dd;
    while (a >= 0) { dof_ancmask[dd] |= 1ull << a; a = d->dof_parentid[a]; }
  }
  for (int b = 0; b < nb; b++) {
    int a = b;
    while (a > 0) {
      for (int dd = 0; dd < nv; dd++) if (d->dof_bodyid[dd] == a) body_dofmask[b] |= 1ull << dd;
      a = d->body_parentid[a];
    }
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
  fix.push_back({(const void**)&M.body_depth, bb.add(depth.data(), sizeof(int) * nb)});
  fix.push_back({(const void**)&M.body_chain, bb.add(chain.data(), sizeof(int) * chain.size())});
  fix.push_back({(const void**)&M.body_subtree_end, bb.add(sub_end.data(), sizeof(int) * nb)});
  fix.push_back({(const void**)&M.body_dofmask, bb.add(body_dofmask.data(), sizeof(unsigned long long) * nb)});
  fix.push_back({(const void**)&M.dof_ancmask, bb.add(dof_ancmask.data(), sizeof(unsigned long long) * nv)});
  fix.push_back({(const void**)&M.efc_row_con, bb.add(row_con.data(), sizeof(int) * row_con.size())});
  M.max_depth = max_depth;
  for (int p = 0; p < d->npair; p++)
    if (d->pair_fn[p] > MJH_FN_CAPSULE_CAPSULE) return fail(-38, "convex (box/mesh) pair functions are not built yet");

  M.lds_reals = lds_carve(M, M.off);
  out->lds_bytes = M.lds_reals * (int)sizeof(REAL);
  if (out->lds_bytes > 160 * 1024) return fail(-12, "model does not fit the 160 KiB LDS arena of one CU");

  void* dev = nullptr;
  HIP_TRY(hipMalloc(&dev, bb.host.size() + 16));
  HIP_TRY(hipMemcpy(dev, bb.host.data(), bb.host.size(), hipMemcpyHostToDevice));
  for (auto& f : fix) *f.first = (const unsigned char*)dev + f.second;
  out->blob = dev;
  out->blob_bytes = bb.host.size();
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mjh_step_kernel<REAL>), hipFuncAttributeMaxDynamicSharedMemorySize, out->lds_bytes));
  return 0;
}

template <typename REAL>
int launch(const mjhModel* m, const DevModel<REAL>& M, const mjhData* in, mjhData* out, int64_t B, int flags, int do_step, int stages, void* stream) {
  if (B <= 0) return 0;
  static_assert(sizeof(DevData<REAL>) == sizeof(mjhData), "DevData must mirror mjhData");
  static_assert(sizeof(KArgs<REAL>) <= 4096, "kernel arguments exceed the 4 KiB kernarg segment");
  KArgs<REAL> args;
  args.M = M;
  memcpy(&args.in, in, sizeof(args.in));
  memcpy(&args.out, out, sizeof(args.out));
  args.B = B; args.flags = flags; args.do_step = do_step; args.stages = stages;
  if (!args.in.qpos || !args.in.qvel) return fail(-22, "in.qpos and in.qvel are required");
  const int64_t grid = B < (int64_t)1 << 20 ? B : (int64_t)1 << 20;
  hipLaunchKernelGGL(mjh_step_kernel<REAL>, dim3((unsigned)grid), dim3(MJH_WAVE), (size_t)m->lds_bytes, (hipStream_t)stream, args);
  HIP_TRY(hipGetLastError());
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
  if (desc->ne || desc->nf) return fail(-38, "equality / frictionloss rows not implemented");
  mjhModel* m = new mjhModel();
  m->dtype = dtype;
  int rc = dtype == MJH_F64 ? build<double>(desc, m, m->m64) : build<float>(desc, m, m->m32);
  if (rc != 0) { delete m; return rc; }
  *out = m;
  return 0;
}

void mjh_model_destroy(mjhModel* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

int mjh_forward(const mjhModel* m, const mjhData* in, mjhData* out, int64_t B, int stages, int flags, void* stream) {
  if (!m || !in || !out) return fail(-22, "null argument");
  return m->dtype == MJH_F64 ? launch<double>(m, m->m64, in, out, B, flags, 0, stages, stream)
                             : launch<float>(m, m->m32, in, out, B, flags, 0, stages, stream);
}

int mjh_step(const mjhModel* m, const mjhData* in, mjhData* out, int64_t B, int flags, void* stream) {
  if (!m || !in || !out) return fail(-22, "null argument");
  return m->dtype == MJH_F64 ? launch<double>(m, m->m64, in, out, B, flags, 1, MJH_STAGE_ALL, stream)
                             : launch<float>(m, m->m32, in, out, B, flags, 1, MJH_STAGE_ALL, stream);
}

int mjh_model_lds_bytes(const mjhModel* m) { return m ? m->lds_bytes : 0; }
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
