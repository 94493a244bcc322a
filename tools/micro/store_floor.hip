// Pure-store floor of a step's output leaves: the byte-per-environment sizes of the leaves one step writes are given on the command line; every leaf is a
// batch-major [B, size] array of its own, and one wavefront writes the rows of four consecutive environments of every leaf (a contiguous span: the most favourable pattern the
// kernels could have) with 4-byte lanes -- plain stores, then with the non-temporal hint, then 16-byte lanes.  No loads, no arithmetic: what is left is what the memory system
// takes for the stores themselves when every wave of a one-round launch issues them at the same moment.
//   hipcc -O3 --offload-arch=gfx950 store_floor.hip -o store_floor && ./store_floor B size1 size2 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Leaves { float* p[64]; int bytes[64]; int n; };
template <int MODE>
__global__ void __launch_bounds__(64) k(Leaves L, float v) {
  const long long e0 = (long long)blockIdx.x * 4;
  for (int i = 0; i < L.n; i++) {
    const int n = L.bytes[i];  // bytes per environment: four environments = one contiguous span of n bytes x 4
    if (MODE == 2 && (n % 16) == 0) {
      typedef float v4 __attribute__((ext_vector_type(4)));
      v4* dst = reinterpret_cast<v4*>(reinterpret_cast<char*>(L.p[i]) + e0 * n);
      for (int w = threadIdx.x; w < n / 4; w += 64) __builtin_nontemporal_store(v4{v, v, v, v}, &dst[w]);
    } else {
      float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(L.p[i]) + e0 * n);
      for (int w = threadIdx.x; w < n; w += 64) { if (MODE == 0) dst[w] = v; else __builtin_nontemporal_store(v, &dst[w]); }
    }
  }
}
int main(int argc, char** argv) {
  const long long B = atoll(argv[1]);
  Leaves L{}; L.n = argc - 2;
  double total = 0;
  for (int i = 0; i < L.n; i++) { L.bytes[i] = atoi(argv[2 + i]); total += (double)L.bytes[i] * B; hipMalloc(&L.p[i], (size_t)L.bytes[i] * B + 256); }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[3] = {"plain 4-byte stores", "non-temporal 4-byte stores", "non-temporal 16-byte stores (leaves whose rows are whole 16-byte groups)"};
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9, sum = 0;
    for (int it = 0; it < 30; it++) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((unsigned)(B / 4)), dim3(64), 0, 0, L, 1.0f + it);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3((unsigned)(B / 4)), dim3(64), 0, 0, L, 1.0f + it);
      else hipLaunchKernelGGL(k<2>, dim3((unsigned)(B / 4)), dim3(64), 0, 0, L, 1.0f + it);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (it >= 10) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-78s %lld environments x %.0f B = %.1f MB: mean %.1f us (best %.1f) = %.2f TB/s\n", names[mode], B, total / B, total / 1e6, 1e3 * sum / 20, 1e3 * best, total / (1e-3 * sum / 20) / 1e12);
  }
  return 0;
}
