// What does one 32-pixel sub-step of the weight-gradient kernels cost in isolation?  8 transposed fragment reads
// (ds_read_b64_tr_b16 pairs: 4 A + 4 B fragments) + 16 v_mfma_f32_16x16x32_bf16 per wave, from a resident LDS image, no
// loader, no barrier.  Variants: 0 reads then MFMAs (the compiler's order in wgrad2_body), 1 MFMAs only, 2 reads only,
// 3 plain ds_read_b64 in place of the transposed ones, 4 fragments of sub-step s + 1 read before the MFMAs of sub-step s.
// hipcc --offload-arch=gfx950 -O3 tr_mfma.hip -o tr_mfma && ./tr_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool TR>
__device__ __forceinline__ bf16x8 rd2(const char* b0, const char* b1) {
  s16x4 lo, hi;
  if (TR) {
    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)b0);
    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)b1);
  } else {
    lo = *(const s16x4*)b0;
    hi = *(const s16x4*)b1;
  }
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 64 rows x 256 B (dy) + 64 rows x 256 B (x): one stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, t = lane & 15;
  for (int i = tid; i < 32768 / 4; i += 256) ((unsigned*)smem)[i] = 0x3f803f80u + i;
  __syncthreads();
  const int wco = wave >> 1, wci = wave & 1;
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane row bases as in wgrad2_body (row P = ks * 32 + 16 e + 4 g + (t >> 2), 256-byte rows, 32-byte group swizzle)
  int pA[2][2];
  for (int ks = 0; ks < 2; ++ks) for (int e = 0; e < 2; ++e) pA[ks][e] = (ks * 32 + 16 * e + 4 * g + (t >> 2)) * 256 + (t & 3) * 8;
  const char* dyB = smem;
  const char* xB = smem + 16384;
  auto frag = [&](int ks, bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    const int f0 = (ks * 32 + 4 * g + (t >> 2)) & 7, f1 = (ks * 32 + 16 + 4 * g + (t >> 2)) & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = rd2<V != 3>(dyB + pA[ks][0] + (((wco * 4 + i) ^ f0) * 32), dyB + pA[ks][1] + (((wco * 4 + i) ^ f1) * 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = rd2<V != 3>(xB + pA[ks][0] + (((wci * 4 + j) ^ f0) * 32), xB + pA[ks][1] + (((wci * 4 + j) ^ f1) * 32));
  };
  auto mul = [&](bf16x8 (&a)[4], bf16x8 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  __syncthreads();
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  if (V == 4) {
    bf16x8 a[2][4], b[2][4];
    frag(0, a[0], b[0]);
    for (int it = 0; it < iters; ++it) {
      frag(1, a[1], b[1]);
      mul(a[0], b[0]);
      frag(0, a[0], b[0]);
      mul(a[1], b[1]);
    }
  } else {
    bf16x8 a[4], b[4];
    if (V == 1) frag(0, a, b);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (V != 1) frag(ks, a, b);
        if (V != 2) mul(a, b);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc[i][0][0] += (float)a[i][0]; acc[0][i][1] += (float)b[i][0]; }
        }
        if (V == 1) asm volatile("" ::: "memory");
      }
    }
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { cyc[blockIdx.x * 2] = c1 - c0; cyc[blockIdx.x * 2 + 1] = t1 - t0; }
}

template <int V>
static void run(const char* name, int blocks, int iters) {
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, blocks * 256 * 4)); CHECK(hipMalloc(&cyc, blocks * 16));
  CHECK(hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 65536, 0, out, cyc, iters);
  CHECK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 65536, 0, out, cyc, iters);
  hipEventRecord(e1);
  CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  printf("%-44s blocks %4d: %7.1f us  = %6.1f ns per 64-pixel step (2 sub-steps: 32 MFMAs + 32 reads per wave); block 0: %lld shader clocks, %lld wall ticks (100 MHz) per step x100\n",
         name, blocks, ms * 1e3, ms * 1e6 / iters, h[0] / iters, h[1] * 100 / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 2000;
  for (int blocks : {1, 256, 512}) {
    run<0>("reads -> MFMAs (compiler order)", blocks, it);
    run<1>("MFMAs only", blocks, it);
    run<2>("transposed reads only", blocks, it);
    run<3>("plain ds_read_b64 -> MFMAs", blocks, it);
    run<4>("next sub-step's reads before the MFMAs", blocks, it);
  }
  return 0;
}
