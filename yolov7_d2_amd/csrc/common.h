// Shared device/host helpers for libmi355det (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mi355_det.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) double f64x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MI_WAVE 64

extern thread_local char g_mi_err[512];
#define MI_FAIL(code, ...)                              \
  do {                                                  \
    snprintf(g_mi_err, sizeof(g_mi_err), __VA_ARGS__);  \
    return (code);                                      \
  } while (0)
#define MI_REQUIRE(cond, ...)              \
  do {                                     \
    if (!(cond)) MI_FAIL(MI_EINVAL, __VA_ARGS__); \
  } while (0)
#define MI_CHECK_LAUNCH(name)                                                     \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) MI_FAIL(MI_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline int mi_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t mi_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }

// unpack 8 bf16 (one 16-byte vector) to floats
__device__ __forceinline__ void unpack8(const bf16x8& v, float* f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ bf16x8 pack8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
  return v;
}

// 1-ulp hardware reciprocal instead of the ~10-instruction IEEE division: the consumers round to bf16 anyway
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// counter-based random bits: dropout masks are pure functions of (seed, element index), so a backward kernel recomputes the
// mask of its forward instead of loading it.  Two stages: mi_rng_key(seed) - one splitmix64 round, uniform over a launch
// (scalar ALU, once per thread) - and mi_rng32k(key, idx), the per-element part in 32-bit arithmetic (lowbias32 mixer: two
// 32-bit multiplies; the high index word only enters through xor / rotate).  Round 4: the first form ran splitmix64 - two
// 64-bit multiplies, ~8 quarter-rate VALU multiplies - PER ELEMENT, and attention-weight dropout made the encoder's
// attention kernels 4x slower than without it (80 us vs 19 us forward at L = 1 050: 35 M scores per layer).
__device__ __forceinline__ unsigned long long mi_rng_key(unsigned long long seed) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned mi_rng32k(unsigned long long key, unsigned long long idx) {
  const unsigned hi = (unsigned)(idx >> 32);
  unsigned x = (unsigned)idx ^ (unsigned)key ^ hi ^ ((hi << 16) | (hi >> 16));
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x ^ (unsigned)(key >> 32);
}
__device__ __forceinline__ unsigned mi_rng32(unsigned long long seed, unsigned long long idx) {
  return mi_rng32k(mi_rng_key(seed), idx);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
