// Cost of a grid-wide barrier on MI355X, by construction (hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier).
// Variants: flat counter / two-level tree with G groups; relaxed polling; with and without 4096 fp64 atomics before it
// (the BatchNorm sums) and the 32 sc1 loads per channel after it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned* gen_w(unsigned* bar, int g) { return bar + 64 * (1 + g); }
__device__ __forceinline__ unsigned* cnt_w(unsigned* bar, int g) { return bar + 64 * (1 + 64 + g); }

template <int SLEEP>
__device__ __forceinline__ void barrier(unsigned* bar, unsigned gen0, int bid, int nb, int GG) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int G = nb < GG ? nb : GG;
    const int g = bid % G;
    const unsigned gsize = (unsigned)((nb - g + G - 1) / G);
    bool released = false;
    if (__hip_atomic_fetch_add(cnt_w(bar, g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
      __hip_atomic_store(cnt_w(bar, g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u) {
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 0; q < G; ++q) __hip_atomic_fetch_add(gen_w(bar, q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        released = true;
      }
    }
    if (!released) {
      int spins = 0;
      while (__hip_atomic_load(gen_w(bar, g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0) {
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1 << 22)) break;
      }
    }
  }
  __syncthreads();
}

// mode bits: 1 = fp64 atomics before (C channels x 2, slot = bid % 16), 2 = slot loads after, 4 = barrier
template <int SLEEP>
__global__ __launch_bounds__(256, 2) void k(unsigned* bar, double* acc, float* out, int C, int GG, int mode) {
  const int bid = blockIdx.x, nb = gridDim.x;
  unsigned gen0 = 0;
  if (threadIdx.x == 0) {
    const int G = nb < GG ? nb : GG;
    gen0 = __hip_atomic_load(gen_w(bar, bid % G), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (mode & 1) {
    double* slot = acc + (size_t)(bid % 16) * C * 2;
    for (int j = threadIdx.x; j < C * 2; j += 256) atomicAdd(slot + j, 1.0);
  }
  if (mode & 4) barrier<SLEEP>(bar, gen0, bid, nb, GG);
  if (mode & 2) {
    for (int c = threadIdx.x; c < C; c += 256) {
      double t = 0;
      double v[32];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        v[2 * s] = __hip_atomic_load(acc + ((size_t)s * C + c) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[2 * s + 1] = __hip_atomic_load(acc + ((size_t)s * C + c) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) t += v[s];
      if (bid == 0) out[c] = (float)t;
    }
  }
}

template <int SLEEP>
static float run(unsigned* bar, double* acc, float* out, int nb, int C, int GG, int mode, int iters) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<SLEEP>, dim3(nb), dim3(256), 0, 0, bar, acc, out, C, GG, mode);
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<SLEEP>, dim3(nb), dim3(256), 0, 0, bar, acc, out, C, GG, mode);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / iters;
}

int main() {
  unsigned* bar; double* acc; float* out;
  CHECK(hipMalloc(&bar, 64 * 200 * 4)); CHECK(hipMemset(bar, 0, 64 * 200 * 4));
  CHECK(hipMalloc(&acc, 16 * 2048 * 2 * 8)); CHECK(hipMemset(acc, 0, 16 * 2048 * 2 * 8));
  CHECK(hipMalloc(&out, 2048 * 4));
  const int iters = 200;
  printf("empty kernel (mode 0), 512 blocks: %.2f us\n", run<8>(bar, acc, out, 512, 128, 16, 0, iters));
  for (int nb : {64, 256, 512}) {
    for (int GG : {1, 8, 16, 32, 64}) {
      printf("nb %3d groups %2d: barrier only %.2f us (sleep 8) %.2f us (sleep 1) %.2f us (no sleep)\n", nb, GG,
             run<8>(bar, acc, out, nb, 128, GG, 4, iters), run<1>(bar, acc, out, nb, 128, GG, 4, iters),
             run<0>(bar, acc, out, nb, 128, GG, 4, iters));
    }
  }
  for (int C : {64, 128, 256, 512}) {
    printf("C %3d nb 512: atomics only %.2f  loads only %.2f  atomics+barrier %.2f  atomics+barrier+loads %.2f us\n", C,
           run<8>(bar, acc, out, 512, C, 16, 1, iters), run<8>(bar, acc, out, 512, C, 16, 2, iters),
           run<8>(bar, acc, out, 512, C, 16, 5, iters), run<8>(bar, acc, out, 512, C, 16, 7, iters));
  }
  return 0;
}
