#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[3600]; };
struct Small { char b[64]; };
template <typename T> __global__ void k(T a) { if (a.b[0] == 77) printf("x"); }
template <typename T> double run(int lds, int n) {
  T a{}; hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k<T>, dim3(64), dim3(64), lds, s, a);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k<T>, dim3(64), dim3(64), lds, s, a);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("  issue %.1f us/launch, end-to-end %.1f us/launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
  return 0;
}
int main() {
  hipFuncSetAttribute((const void*)k<Big>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("64 B kernarg, no LDS:\n"); run<Small>(0, 2000);
  printf("3600 B kernarg, no LDS:\n"); run<Big>(0, 2000);
  printf("3600 B kernarg, 20 KB dynamic LDS:\n"); run<Big>(20000, 2000);
  return 0;
}
