// what does the BN statistics prologue cost, and which load pattern is cheapest?  (fp64 slot accumulators filled by atomics)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define SLOTS 16
__global__ void fill_k(double* acc, int C) {  // emulates the conv epilogue: block = tile, 128 threads add per-channel sums
  const int c = threadIdx.x;
  if (c < C) {
    double* sp = acc + ((size_t)(blockIdx.x % SLOTS) * C + c) * 2;
    atomicAdd(sp, 1.0 + c);
    atomicAdd(sp + 1, 2.0 + c);
  }
}
template <int V>
__global__ __launch_bounds__(256) void cons_k(const double* __restrict__ acc, int C, float* out, int iters) {
  __shared__ float s_sc[1024];
  if (V == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) { s1 += acc[((size_t)k * C + c) * 2]; s2 += acc[((size_t)k * C + c) * 2 + 1]; }
      s_sc[c] = (float)(s1 / (s2 + 1.0));
    }
  } else if (V == 1) {  // 16-byte loads
    for (int c = threadIdx.x; c < C; c += 256) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) { const double2 v = *(const double2*)(acc + ((size_t)k * C + c) * 2); s1 += v.x; s2 += v.y; }
      s_sc[c] = (float)(s1 / (s2 + 1.0));
    }
  } else if (V == 2) {  // all 256 threads: (channel, slot-half) then LDS combine
    __shared__ double part[2][1024][2];
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      const int c = i % C, hf = i / C;
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < SLOTS / 2; ++k) { const double2 v = *(const double2*)(acc + ((size_t)(hf * (SLOTS / 2) + k) * C + c) * 2); s1 += v.x; s2 += v.y; }
      part[hf][c][0] = s1; part[hf][c][1] = s2;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) s_sc[c] = (float)((part[0][c][0] + part[1][c][0]) / (part[0][c][1] + part[1][c][1] + 1.0));
  } else if (V == 5) {  // V1 + the fp64 rsqrt / products of the real kernel
    for (int c = threadIdx.x; c < C; c += 256) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) { const double2 v = *(const double2*)(acc + ((size_t)k * C + c) * 2); s1 += v.x; s2 += v.y; }
      const double mean = s1 * 1e-5;
      double var = s2 * 1e-5 - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + 1e-3);
      s_sc[c] = (float)(1.5 * invstd) + (float)(0.5 - mean * 1.5 * invstd);
    }
  } else if (V == 3) {  // no prologue (reference)
    for (int c = threadIdx.x; c < C; c += 256) s_sc[c] = 1.f;
  } else if (V == 4) {  // float math only after the sums (no fp64 divide)
    for (int c = threadIdx.x; c < C; c += 256) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) { const double2 v = *(const double2*)(acc + ((size_t)k * C + c) * 2); s1 += v.x; s2 += v.y; }
      s_sc[c] = (float)s1 * __builtin_amdgcn_rcpf((float)s2 + 1.0f);
    }
  }
  __syncthreads();
  float a = 0;
  for (int i = 0; i < iters; ++i) a += s_sc[(threadIdx.x + i) % C];
  out[blockIdx.x * 256 + threadIdx.x] = a;
}
int main() {
  const int C = 128, NT = 800;
  double* acc; float* out;
  hipMalloc(&acc, SLOTS * 1024 * 16); hipMalloc(&out, 4 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 6; ++v) {
    float tot = 0;
    for (int rep = 0; rep < 20; ++rep) {
      hipMemsetAsync(acc, 0, SLOTS * 1024 * 16, 0);
      hipLaunchKernelGGL(fill_k, dim3(NT), dim3(128), 0, 0, acc, C);
      hipEventRecord(e0, 0);
      if (v == 0) hipLaunchKernelGGL(cons_k<0>, dim3(800), dim3(256), 0, 0, acc, C, out, 8);
      if (v == 1) hipLaunchKernelGGL(cons_k<1>, dim3(800), dim3(256), 0, 0, acc, C, out, 8);
      if (v == 2) hipLaunchKernelGGL(cons_k<2>, dim3(800), dim3(256), 0, 0, acc, C, out, 8);
      if (v == 3) hipLaunchKernelGGL(cons_k<3>, dim3(800), dim3(256), 0, 0, acc, C, out, 8);
      if (v == 5) hipLaunchKernelGGL(cons_k<5>, dim3(800), dim3(256), 0, 0, acc, C, out, 8);
      if (v == 4) hipLaunchKernelGGL(cons_k<4>, dim3(800), dim3(256), 0, 0, acc, C, out, 8);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (rep >= 5) tot += ms;
    }
    printf("variant %d: %.2f us\n", v, tot / 15 * 1e3);
  }
  return 0;
}
