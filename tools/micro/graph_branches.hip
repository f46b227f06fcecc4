// do independent hipGraph branches (captured with fork/join across two streams) overlap on MI355X?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin_k(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  float *a, *b; hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t fork, join; hipEventCreate(&fork); hipEventCreate(&join);
  const int K = 50, IT = 4000, BL = 64;
  for (int mode = 0; mode < 3; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal);
    if (mode == 0) {          // serial: 2K kernels on one stream
      for (int i = 0; i < K; ++i) { hipLaunchKernelGGL(spin_k, dim3(BL), dim3(256), 0, s1, a, IT); hipLaunchKernelGGL(spin_k, dim3(BL), dim3(256), 0, s1, b, IT); }
    } else if (mode == 1) {   // two long branches
      hipEventRecord(fork, s1); hipStreamWaitEvent(s2, fork, 0);
      for (int i = 0; i < K; ++i) { hipLaunchKernelGGL(spin_k, dim3(BL), dim3(256), 0, s1, a, IT); hipLaunchKernelGGL(spin_k, dim3(BL), dim3(256), 0, s2, b, IT); }
      hipEventRecord(join, s2); hipStreamWaitEvent(s1, join, 0);
    } else {                  // fork/join around every pair (fine-grained)
      for (int i = 0; i < K; ++i) {
        hipEventRecord(fork, s1); hipStreamWaitEvent(s2, fork, 0);
        hipLaunchKernelGGL(spin_k, dim3(BL), dim3(256), 0, s1, a, IT); hipLaunchKernelGGL(spin_k, dim3(BL), dim3(256), 0, s2, b, IT);
        hipEventRecord(join, s2); hipStreamWaitEvent(s1, join, 0);
      }
    }
    hipStreamEndCapture(s1, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s1); hipStreamSynchronize(s1);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s1);
    hipStreamSynchronize(s1);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("mode %d: %.1f us per graph (%d kernels)\n", mode, std::chrono::duration<double, std::micro>(t1 - t0).count() / 5.0, 2 * K);
  }
  return 0;
}
