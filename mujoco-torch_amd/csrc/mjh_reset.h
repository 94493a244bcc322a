// mjh_reset.h -- masked in-place reset of environments, the env caller's `self._dx[mask] = self._make_batch(n)`
// (reference zoo/base.py:266-273 _make_batch; :289-293 partial reset; :327-331 fused auto-reset).
//
// The reference gathers n fresh environments (dx0 broadcast + uniform noise on qpos / qvel) and scatters them leaf by leaf
// with a boolean-mask index_put: ~70 leaves x (nonzero + index_put) launches plus a host sync for the count.  Here it is one
// launch and no sync: one workgroup per environment, workgroups of unmasked environments exit at once, the others stream
// every leaf of the single-environment `d0` (qpos / qvel: of the caller's per-environment rows) over their slice of `d`.
// Pure HBM traffic: bytes(Data of one env) written per reset environment.
#pragma once
#include <hip/hip_runtime.h>

#define MJH_RESET_MAX_LEAVES 80

struct ResetLeaf {
  unsigned* dst;        // [B, words]
  const unsigned* src;  // [words] (one environment, stride 0) or [B, words] (stride = words)
  int words;            // 4-byte words per environment
  int src_stride;
};

struct ResetArgs {
  const unsigned char* mask;  // [B], non-zero = reset
  int nleaf;
  int pad_;
  ResetLeaf leaf[MJH_RESET_MAX_LEAVES];
};

__global__ __launch_bounds__(256) void mjh_reset_kernel(ResetArgs a) {
  const long long e = blockIdx.x;
  if (!a.mask[e]) return;
  for (int l = 0; l < a.nleaf; l++) {
    const ResetLeaf L = a.leaf[l];
    unsigned* dst = L.dst + e * L.words;
    const unsigned* src = L.src + e * L.src_stride;
    for (int i = threadIdx.x; i < L.words; i += blockDim.x) dst[i] = src[i];
  }
}


// Counting sort of the environments by the iteration-count key the register solver left last step (any int: clipped to 0..63), slowest first: `perm[slot]` = environment
// served by that slot of the solver's four-per-wavefront launch.  One workgroup; the order inside a bucket is whatever the atomics give -- which environments share a wave
// has no effect on any environment's result.  Garbage keys (a fresh workspace) still give a valid permutation.
__global__ __launch_bounds__(1024) void mjh_sort_kernel(const int* key, int* perm, long long B) {
  __shared__ int hist[64];
  __shared__ int base[64];
  const int t = threadIdx.x;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (long long e = t; e < B; e += 1024) { const unsigned k = (unsigned)key[e]; atomicAdd(&hist[k > 63u ? 63 : (int)k], 1); }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 63; k >= 0; k--) { base[k] = acc; acc += hist[k]; } }
  __syncthreads();
  for (long long e = t; e < B; e += 1024) { const unsigned k = (unsigned)key[e]; perm[atomicAdd(&base[k > 63u ? 63 : (int)k], 1)] = (int)e; }
}
