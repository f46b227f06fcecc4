// in-graph cost of dependent tiny kernels on one stream (MI355X): empty kernel / tiny streaming kernel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void empty_k(float* p) { if (p == nullptr) return; }
__global__ void small_k(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  float* d; hipMalloc(&d, 64 << 20);
  hipStream_t s; hipStreamCreate(&s);
  for (int mode = 0; mode < 4; ++mode) {
    const int K = 500;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < K; ++i) {
      if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, d);
      if (mode == 1) hipLaunchKernelGGL(empty_k, dim3(512), dim3(256), 0, s, d);
      if (mode == 2) hipLaunchKernelGGL(small_k, dim3(64), dim3(256), 0, s, d, 64 * 256);
      if (mode == 3) hipLaunchKernelGGL(small_k, dim3(4096), dim3(256), 0, s, d, 4096 * 256);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t1 = std::chrono::high_resolution_clock::now();
    double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (5.0 * K);
    printf("mode %d: %.2f us per kernel in graph\n", mode, us);
    // eager
    t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    t1 = std::chrono::high_resolution_clock::now();
    if (mode == 0) printf("eager empty: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / K);
  }
  return 0;
}
