// forward-mode dual numbers with 4 partials (value + d/d(box x, y, w, h)): the box-loss kernels carry the gradient
// along the forward expression, so forward and backward are one pass and cannot drift apart
#pragma once
#include "common.h"

struct D4 {  // value and d/d(pred x, y, w, h)  (or x1, y1, x2, y2)
  float v, d[4];
};
__device__ __forceinline__ D4 dconst(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float c, int i) { D4 r = dconst(c); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  D4 r; r.v = a.v / b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
  return r;
}
__device__ __forceinline__ D4 operator+(D4 a, float c) { a.v += c; return a; }
__device__ __forceinline__ D4 operator-(D4 a, float c) { a.v -= c; return a; }
__device__ __forceinline__ D4 operator*(D4 a, float c) { a.v *= c; for (int i = 0; i < 4; ++i) a.d[i] *= c; return a; }
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v <= b.v ? a : b; }   // torch.min / max: gradient to the selected
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 dclamp0(D4 a) { return a.v > 0.f ? a : dconst(0.f); }  // clamp(0): zero gradient below
__device__ __forceinline__ D4 dabs(D4 a) { return a.v >= 0.f ? a : a * -1.f; }
__device__ __forceinline__ D4 dchain(D4 a, float fv, float fd) { D4 r; r.v = fv; for (int i = 0; i < 4; ++i) r.d[i] = fd * a.d[i]; return r; }
__device__ __forceinline__ D4 dsqr(D4 a) { return dchain(a, a.v * a.v, 2.f * a.v); }
__device__ __forceinline__ D4 dsqrt(D4 a) { const float s = sqrtf(a.v); return dchain(a, s, 0.5f / s); }
__device__ __forceinline__ D4 datan(D4 a) { return dchain(a, atanf(a.v), 1.f / (1.f + a.v * a.v)); }
__device__ __forceinline__ D4 dexp(D4 a) { const float e = expf(a.v); return dchain(a, e, e); }
__device__ __forceinline__ D4 dpow4(D4 a) { const float a2 = a.v * a.v; return dchain(a, a2 * a2, 4.f * a2 * a.v); }

