// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS library's access shape (MI355X_MICROARCH.md: "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  The phase kernels move Data rows with one
// element per lane: 8 B / lane (float64) or 4 B / lane (float32), a wave covering one contiguous run.  Each kernel below streams a
// 1 GiB buffer (beyond the 256 MiB Infinity Cache) exactly once with that shape; the script compares the counters with the bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ void read_k(const T* __restrict__ p, size_t n, T* sink) {
  T acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == (T)123456789) *sink = acc;  // never true: keeps the loads
}
__global__ void read16_k(const double2* __restrict__ p, size_t n, double* sink) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 123456789.0) *sink = acc;
}
template <typename T>
__global__ void write_k(T* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (T)i;
}
// the phase kernels' shape exactly: one 64-thread workgroup per "environment", rows of ROW elements, row r of environment e at e * ROW
template <typename T, int ROW>
__global__ void rows_k(const T* __restrict__ p, size_t nenv, T* __restrict__ q) {
  for (size_t e = blockIdx.x; e < nenv; e += gridDim.x)
    for (int i = threadIdx.x; i < ROW; i += 64) q[e * ROW + i] = p[e * ROW + i] * (T)2;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *b;
  double* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 8);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  const int grid = 256 * 32;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(read_k<double>, dim3(grid), dim3(256), 0, 0, (const double*)a, bytes / 8, sink);
    hipLaunchKernelGGL(read_k<float>, dim3(grid), dim3(256), 0, 0, (const float*)b, bytes / 4, (float*)sink);
    hipLaunchKernelGGL(read16_k, dim3(grid), dim3(256), 0, 0, (const double2*)a, bytes / 16, sink);
    hipLaunchKernelGGL(write_k<double>, dim3(grid), dim3(256), 0, 0, (double*)b, bytes / 8);
    hipLaunchKernelGGL(write_k<float>, dim3(grid), dim3(256), 0, 0, (float*)a, bytes / 4);
    hipLaunchKernelGGL((rows_k<double, 1431>), dim3(4096), dim3(64), 0, 0, (const double*)a, (size_t)(bytes / 8 / 1431), (double*)b);   // efc_J row of the humanoid: 53 x 27
    hipLaunchKernelGGL((rows_k<float, 1504>), dim3(4096), dim3(64), 0, 0, (const float*)b, (size_t)(bytes / 4 / 1504), (float*)a);      // efc_J row of the ant: 188 x 8
  }
  hipDeviceSynchronize();
  printf("bytes per read / write kernel: %zu; rows_k<double,1431>: %zu each way; rows_k<float,1504>: %zu each way\n", bytes,
         (bytes / 8 / 1431) * 1431 * 8, (bytes / 4 / 1504) * 1504 * 4);
  return 0;
}
