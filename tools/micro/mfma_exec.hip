// Does v_mfma_f32_4x4x1_16b_f32 honour EXEC?  (1) layout check with every lane active; (2) the same instruction inside a divergent branch taken by lanes 0..15 only: are the
// destination registers of lanes 16..63 left alone, and do lanes 0..15 get the products of their own blocks?      hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_exec.hip -o /tmp/mfma_exec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
  const int l = threadIdx.x;
  const float a = 1.0f + l, b = 100.0f + l;
  f4 c = {-1.f, -2.f, -3.f, -4.f};
  if (mode == 0) {
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  } else {
    if (l < 16) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  }
  for (int t = 0; t < 4; t++) out[l * 4 + t] = c[t];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
  float h[256];
  for (int mode = 0; mode < 2; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l : {0, 1, 5, 15, 16, 17, 40, 63}) printf("  lane %2d: %g %g %g %g\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
