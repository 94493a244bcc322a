// host cost of hipLaunchKernelGGL against the size of a by-value kernel argument (GPU box: hipcc --offload-arch=gfx950 launch_cost.hip -o launch_cost && ./launch_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Arg { int v[N]; int* out; };
template <int N> __global__ void k(Arg<N> a) { if (a.out && threadIdx.x == 0 && a.v[N - 1] == 12345) a.out[0] = a.v[0]; }
template <int N> void run(hipStream_t s) {
  Arg<N> a; for (int i = 0; i < N; i++) a.v[i] = i; a.out = nullptr;
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k<N>, dim3(4096), dim3(64), 1024, s, a);
  hipStreamSynchronize(s);
  const int n = 4000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) { hipLaunchKernelGGL(k<N>, dim3(4096), dim3(64), 1024, s, a); (void)hipGetLastError(); }
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("arg %5zu B: host %.2f us per launch (%.2f incl. drain)\n", sizeof(a), 1e6 * std::chrono::duration<double>(t1 - t0).count() / n, 1e6 * std::chrono::duration<double>(t2 - t0).count() / n);
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  run<8>(s); run<128>(s); run<450>(s); run<900>(s); run<1000>(s);
  return 0;
}
