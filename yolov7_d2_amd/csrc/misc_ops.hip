// Data-movement ops of the YOLOX path (NHWC bf16): Focus space-to-depth, nearest x2 upsample,
// SPP max-pools, strided copy, column sums, and the fused SGD-momentum update.  All HBM-bound.
#include <stdlib.h>
#include "common.h"

static int ew_blocks(int64_t total, int cap = 4096) {
  int64_t b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- Focus (yolov7/modeling/backbone/layers/wrappers.py:202-220): channel order TL, BL, TR, BR x (c0,c1,c2)
__global__ __launch_bounds__(256) void focus_pack_kernel(const float* __restrict__ img, int N, int H, int W,
                                                         __bf16* out, int ldo) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % Wo);
    const int64_t r = idx / Wo;
    const int oy = (int)(r % Ho), n = (int)(r / Ho);
    float f[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dy = q & 1, dx = q >> 1;  // q: 0 TL, 1 BL, 2 TR, 3 BR
#pragma unroll
      for (int c = 0; c < 3; ++c)
        f[q * 3 + c] = img[(((int64_t)n * 3 + c) * H + (2 * oy + dy)) * W + 2 * ox + dx];
    }
    f[12] = f[13] = f[14] = f[15] = 0.f;
    __bf16* op = out + idx * ldo;
    *(bf16x8*)op = pack8(f);
    *(bf16x8*)(op + 8) = pack8(f + 8);
  }
}

// The uint8 image a data loader hands over (yolox.py:96-99 converts to float on the device; values 0..255 are exact in
// bf16) goes through the same kernel: 4x less input traffic than the float image and no conversion pass.
// one output pixel per thread and trip, four trips in flight: per (channel, row) the two horizontally adjacent source
// pixels are ONE 8-byte (fp32) / 2-byte (uint8) load, neighbouring lanes read neighbouring pairs and write neighbouring
// 32-byte output pixels.  (Four adjacent pixels per thread read wider but scatter the stores: 16 bytes out of every 128
// per instruction - measured slower.)  T = float or uint8_t.
template <class T>
__global__ __launch_bounds__(256) void focus_pack4_kernel(const T* __restrict__ img, int N, int H, int W, __bf16* out,
                                                          int ldo) {
  const int Ho = H / 2, Wo = W / 2;
  // 32-bit pixel indices (the launchers check N * Ho * Wo < 2^31): two 64-bit divisions per pixel cost more than its loads
  const unsigned total = (unsigned)N * (unsigned)Ho * (unsigned)Wo;
  const unsigned stride = gridDim.x * 256u;
  for (unsigned idx0 = blockIdx.x * 256u + threadIdx.x; idx0 < total; idx0 += 4 * stride) {
    float f[4][16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned idx = idx0 + j * stride;
      if (idx >= total) break;
      const unsigned r = idx / (unsigned)Wo;
      const int ox = (int)(idx - r * (unsigned)Wo);
      const unsigned nn = r / (unsigned)Ho;
      const int oy = (int)(r - nn * (unsigned)Ho), n = (int)nn;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const T* src = img + (((int64_t)n * 3 + c) * H + (2 * oy + dy)) * W + 2 * ox;
          if constexpr (sizeof(T) == 4) {
            const float2 v = *(const float2*)src;
            f[j][dy * 3 + c] = v.x;
            f[j][(2 + dy) * 3 + c] = v.y;
          } else {
            const unsigned short v = *(const unsigned short*)src;
            f[j][dy * 3 + c] = (float)(v & 0xff);
            f[j][(2 + dy) * 3 + c] = (float)(v >> 8);
          }
        }
      f[j][12] = f[j][13] = f[j][14] = f[j][15] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned idx = idx0 + j * stride;
      if (idx >= total) break;
      __bf16* op = out + (size_t)idx * ldo;
      *(bf16x8*)op = pack8(f[j]);
      *(bf16x8*)(op + 8) = pack8(f[j] + 8);
    }
  }
}

// The uint8 image, one BLOCK per pair of output rows: the six (channel, row parity) source rows of an output row arrive as
// 16-byte loads (one per lane: W / 16 pieces per row) in LDS, then every thread assembles output pixels from six 2-byte LDS
// reads and writes 32 contiguous bytes.  focus_pack4_kernel's global loads are 2 bytes per lane - 128 bytes per wave
// instruction, 24 instructions per thread: 48.8 us for 72 MB in the captured step (1.5 TB/s).  Needs W % 16 == 0 and a
// 16-byte aligned image (MI_FOCUS_ROWS=0: the per-pixel form).
#define FOCUS_ROWS 2
__global__ __launch_bounds__(256) void focus_pack_u8_rows_kernel(const uint8_t* __restrict__ img, int N, int H, int W,
                                                                 __bf16* out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) uint8_t srow[];      // [FOCUS_ROWS][3 channels][2 parities][W]
  const int Ho = H / 2, Wo = W / 2;
  const int rb = (Ho + FOCUS_ROWS - 1) / FOCUS_ROWS;
  const int n = blockIdx.x / rb, oy0 = (blockIdx.x % rb) * FOCUS_ROWS;
  const int per = W / 16;
  for (int i = threadIdx.x; i < FOCUS_ROWS * 6 * per; i += 256) {
    const int piece = i % per, rowi = i / per;        // rowi = (r * 3 + c) * 2 + dy
    const int dy = rowi & 1, c = (rowi >> 1) % 3, r = rowi / 6;
    if (oy0 + r < Ho)
      *(uint4*)(srow + (size_t)rowi * W + piece * 16) =
          *(const uint4*)(img + (((int64_t)n * 3 + c) * H + 2 * (oy0 + r) + dy) * W + piece * 16);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FOCUS_ROWS * Wo; i += 256) {
    const int r = i / Wo, ox = i - r * Wo;
    if (oy0 + r >= Ho) break;
    float f[16];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const unsigned short v = *(const unsigned short*)(srow + (size_t)((r * 3 + c) * 2 + dy) * W + 2 * ox);
        f[dy * 3 + c] = (float)(v & 0xff);             // q: 0 TL, 1 BL, 2 TR, 3 BR (wrappers.py:202-220)
        f[(2 + dy) * 3 + c] = (float)(v >> 8);
      }
    f[12] = f[13] = f[14] = f[15] = 0.f;
    __bf16* op = out + ((size_t)(n * Ho + oy0 + r) * Wo + ox) * ldo;
    *(bf16x8*)op = pack8(f);
    *(bf16x8*)(op + 8) = pack8(f + 8);
  }
}

// The same block geometry with the STORES laid out for whole lines (round 6): a lane PAIR writes one 32-byte output pixel,
// even lane the first 16 bytes (TL c0..2, BL c0..2, TR c0..1), odd lane the second (TR c2, BR c0..2, four zero pads), so a
// wave's store instruction covers 1 KB of consecutive addresses (ldo = 16) instead of 16 bytes out of every 32 twice
// (focus_pack_u8_rows_kernel: 40 us for 52 MB of output in the captured step, 1.3 TB/s).  FOCUS_ROWS2 output rows per block.
template <int FOCUS_ROWS2>
__global__ __launch_bounds__(256) void focus_pack_u8_pairs_kernel(const uint8_t* __restrict__ img, int N, int H, int W,
                                                                  __bf16* out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) uint8_t srow[];      // [FOCUS_ROWS2][3 channels][2 parities][W]
  const int Ho = H / 2, Wo = W / 2;
  const int rb = (Ho + FOCUS_ROWS2 - 1) / FOCUS_ROWS2;
  const int n = blockIdx.x / rb, oy0 = (blockIdx.x % rb) * FOCUS_ROWS2;
  const int per = W / 16;
  for (int i = threadIdx.x; i < FOCUS_ROWS2 * 6 * per; i += 256) {
    const int piece = i % per, rowi = i / per;        // rowi = (r * 3 + c) * 2 + dy
    const int dy = rowi & 1, c = (rowi >> 1) % 3, r = rowi / 6;
    if (oy0 + r < Ho)
      *(uint4*)(srow + (size_t)rowi * W + piece * 16) =
          *(const uint4*)(img + (((int64_t)n * 3 + c) * H + 2 * (oy0 + r) + dy) * W + piece * 16);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FOCUS_ROWS2 * Wo * 2; i += 256) {
    const int half = i & 1, p = i >> 1;
    const int r = p / Wo, ox = p - r * Wo;
    if (oy0 + r >= Ho) break;
    const uint8_t* base = srow + (size_t)(r * 6) * W + 2 * ox;        // row (c, dy) at base + (c * 2 + dy) * W
    unsigned short v[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) v[c][dy] = *(const unsigned short*)(base + (size_t)(c * 2 + dy) * W);
    float f[8];
    if (half == 0) {      // q 0 TL (dx 0, dy 0), q 1 BL (dx 0, dy 1), first two of q 2 TR (dx 1, dy 0)  (wrappers.py:202-220)
      f[0] = (float)(v[0][0] & 0xff); f[1] = (float)(v[1][0] & 0xff); f[2] = (float)(v[2][0] & 0xff);
      f[3] = (float)(v[0][1] & 0xff); f[4] = (float)(v[1][1] & 0xff); f[5] = (float)(v[2][1] & 0xff);
      f[6] = (float)(v[0][0] >> 8);   f[7] = (float)(v[1][0] >> 8);
    } else {              // TR c2, q 3 BR (dx 1, dy 1), pads
      f[0] = (float)(v[2][0] >> 8);
      f[1] = (float)(v[0][1] >> 8);   f[2] = (float)(v[1][1] >> 8);   f[3] = (float)(v[2][1] >> 8);
      f[4] = f[5] = f[6] = f[7] = 0.f;
    }
    *(bf16x8*)(out + ((size_t)(n * Ho + oy0 + r) * Wo + ox) * ldo + half * 8) = pack8(f);
  }
}

extern "C" int mi_focus_pack_u8(const uint8_t* img, int N, int H, int W, void* out, int ldo, mi_stream_t st) {
  MI_REQUIRE(img && out && H % 2 == 0 && W % 2 == 0 && ldo % 8 == 0 && ldo >= 16 && ((uintptr_t)img & 1) == 0,
             "focus_pack_u8: args");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2);
  MI_REQUIRE(total < (1LL << 31) - (1 << 24), "focus_pack_u8: %lld output pixels exceed the 32-bit index", (long long)total);
  // MI_FOCUS_ROWS: 2 (default) lane-pair stores, 1 the round-5 row kernel, 0 the per-pixel kernel (read per call: the
  // equality test runs all three in one process)
  const int rows_mode = getenv("MI_FOCUS_ROWS") ? atoi(getenv("MI_FOCUS_ROWS")) : 2;
  if (rows_mode >= 2 && W % 16 == 0 && ((uintptr_t)img & 15) == 0 && W <= 4096) {
    static const int th = getenv("MI_FOCUS_TH") ? atoi(getenv("MI_FOCUS_TH")) : 4;     // output rows per block (A/B knob: 1, 2, 4, 8)
    const int R = th == 1 ? 1 : th == 2 ? 2 : th == 8 ? 8 : 4;
    const int rb = (H / 2 + R - 1) / R;
    const dim3 g((unsigned)(N * rb));
    const size_t lds = (size_t)R * 6 * W;
    if (R == 1) hipLaunchKernelGGL(focus_pack_u8_pairs_kernel<1>, g, dim3(256), lds, (hipStream_t)st, img, N, H, W, (__bf16*)out, ldo);
    else if (R == 2) hipLaunchKernelGGL(focus_pack_u8_pairs_kernel<2>, g, dim3(256), lds, (hipStream_t)st, img, N, H, W, (__bf16*)out, ldo);
    else if (R == 8) hipLaunchKernelGGL(focus_pack_u8_pairs_kernel<8>, g, dim3(256), lds, (hipStream_t)st, img, N, H, W, (__bf16*)out, ldo);
    else hipLaunchKernelGGL(focus_pack_u8_pairs_kernel<4>, g, dim3(256), lds, (hipStream_t)st, img, N, H, W, (__bf16*)out, ldo);
    MI_CHECK_LAUNCH("focus_pack_u8 (pairs)");
    return MI_OK;
  }
  if (rows_mode && W % 16 == 0 && ((uintptr_t)img & 15) == 0 && W <= 4096) {
    const int rb = (H / 2 + FOCUS_ROWS - 1) / FOCUS_ROWS;
    hipLaunchKernelGGL(focus_pack_u8_rows_kernel, dim3((unsigned)(N * rb)), dim3(256), (size_t)FOCUS_ROWS * 6 * W, (hipStream_t)st, img,
                       N, H, W, (__bf16*)out, ldo);
    MI_CHECK_LAUNCH("focus_pack_u8 (rows)");
    return MI_OK;
  }
  hipLaunchKernelGGL(focus_pack4_kernel<uint8_t>, dim3(ew_blocks((total + 3) / 4)), dim3(256), 0, (hipStream_t)st, img, N, H,
                     W, (__bf16*)out, ldo);
  MI_CHECK_LAUNCH("focus_pack_u8");
  return MI_OK;
}

extern "C" int mi_focus_pack(const float* img, int N, int H, int W, void* out, int ldo, mi_stream_t st) {
  MI_REQUIRE(img && out && H % 2 == 0 && W % 2 == 0 && ldo % 8 == 0 && ldo >= 16, "focus_pack: args");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2);
  MI_REQUIRE(total < (1LL << 31) - (1 << 24), "focus_pack: %lld output pixels exceed the 32-bit index", (long long)total);
  if (((uintptr_t)img & 7) == 0)
    hipLaunchKernelGGL(focus_pack4_kernel<float>, dim3(ew_blocks((total + 3) / 4)), dim3(256), 0, (hipStream_t)st, img, N, H,
                       W, (__bf16*)out, ldo);
  else
    hipLaunchKernelGGL(focus_pack_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, img, N, H, W,
                       (__bf16*)out, ldo);
  MI_CHECK_LAUNCH("focus_pack");
  return MI_OK;
}

// ---- MaxPool2d(3, stride 2, padding 1): detectron2 BasicStem (ResNet-50 of the DETR / SparseInst configurations)
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const __bf16* __restrict__ x, int ldx, __bf16* y, int ldy,
                                                               int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const bf16x8 v = *(const bf16x8*)(x + (((int64_t)n * H + iy) * W + ix) * ldx + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
      }
    }
    *(bf16x8*)(y + (((int64_t)n * Ho + oy) * Wo + ox) * ldy + c8 * 8) = pack8(m);
  }
}
// gather form of the backward pass: an input pixel collects dy of every window (<= 2 x 2) whose FIRST maximum it is
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const __bf16* __restrict__ x, int ldx,
                                                               const __bf16* __restrict__ dy, int lddy, __bf16* dx,
                                                               int lddx, int accumulate, int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    const __bf16* xp = x + ((int64_t)n * H * W) * ldx + c8 * 8;
    const bf16x8 me = *(const bf16x8*)(xp + ((int64_t)iy * W + ix) * ldx);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {   // windows with 2*oy-1 <= iy <= 2*oy+1: one (iy even) or two (odd)
      if (oy < 0 || oy >= Ho) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        const bf16x8 g = *(const bf16x8*)(dy + (((int64_t)n * Ho + oy) * Wo + ox) * lddy + c8 * 8);
        // am I the first maximum of this window (row-major scan)?
        bool first[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) first[e] = true;
        for (int wy = 0; wy < 3; ++wy) {
          const int yy = 2 * oy - 1 + wy;
          if ((unsigned)yy >= (unsigned)H) continue;
          for (int wx = 0; wx < 3; ++wx) {
            const int xx = 2 * ox - 1 + wx;
            if ((unsigned)xx >= (unsigned)W || (yy == iy && xx == ix)) continue;
            const bf16x8 v = *(const bf16x8*)(xp + ((int64_t)yy * W + xx) * ldx);
            const bool before = (yy < iy) || (yy == iy && xx < ix);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = (float)v[e], b = (float)me[e];
              if (before ? (a >= b) : (a > b)) first[e] = false;
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (first[e]) acc[e] += (float)g[e];
      }
    }
    __bf16* dp = dx + (((int64_t)n * H + iy) * W + ix) * lddx + c8 * 8;
    if (accumulate) {
      const bf16x8 o = *(const bf16x8*)dp;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)o[e];
    }
    *(bf16x8*)dp = pack8(acc);
  }
}
// the same pooling with the position (0..8, row-major in the window) of each output's FIRST maximum recorded - one byte per
// output element - and the backward that reads it: an input pixel looks at its <= 2 x 2 windows and takes dy where the
// recorded position is its own.  (The backward above re-derives "first maximum" from x: 36 loads and ~1300 compares per
// 8-channel item - 1.45 ms for the 320 x 320 stem map of a SparseInst step; this one reads dy and the codes once.)
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_idx_kernel(const __bf16* __restrict__ x, int ldx, __bf16* y, int ldy,
                                                                   uint8_t* __restrict__ code, int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m[8];
    unsigned long long cd = 0ull;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const bf16x8 v = *(const bf16x8*)(x + (((int64_t)n * H + iy) * W + ix) * ldx + c8 * 8);
        const unsigned long long pos = (unsigned long long)(dy * 3 + dx);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          if (f > m[e]) {      // strictly greater: the FIRST maximum in row-major order keeps the window
            m[e] = f;
            cd = (cd & ~(0xffull << (8 * e))) | (pos << (8 * e));
          }
        }
      }
    }
    *(bf16x8*)(y + (((int64_t)n * Ho + oy) * Wo + ox) * ldy + c8 * 8) = pack8(m);
    *(unsigned long long*)(code + ((((int64_t)n * Ho + oy) * Wo + ox) * C8 + c8) * 8) = cd;
  }
}
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_idx_kernel(const uint8_t* __restrict__ code, const __bf16* __restrict__ dy,
                                                                   int lddy, __bf16* dx, int lddx, int accumulate, int N, int H,
                                                                   int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy >= Ho) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox >= Wo) continue;
        const int64_t o = ((int64_t)n * Ho + oy) * Wo + ox;
        const unsigned long long cd = *(const unsigned long long*)(code + (o * C8 + c8) * 8);
        const bf16x8 g = *(const bf16x8*)(dy + o * lddy + c8 * 8);
        const unsigned mine = (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)));
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((unsigned)(cd >> (8 * e)) & 0xffu) == mine) acc[e] += (float)g[e];
      }
    }
    __bf16* dp = dx + (((int64_t)n * H + iy) * W + ix) * lddx + c8 * 8;
    if (accumulate) {
      const bf16x8 o = *(const bf16x8*)dp;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)o[e];
    }
    *(bf16x8*)dp = pack8(acc);
  }
}
extern "C" int mi_maxpool3x3s2_fwd_idx(const void* x, int ldx, void* y, int ldy, uint8_t* code, int N, int H, int W, int C,
                                       mi_stream_t st) {
  MI_REQUIRE(x && y && code && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && N > 0 && H > 0 && W > 0 && ((uintptr_t)code & 7) == 0,
             "maxpool3x3s2_fwd_idx: args");
  const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_fwd_idx_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)x, ldx,
                     (__bf16*)y, ldy, code, N, H, W, C / 8);
  MI_CHECK_LAUNCH("maxpool3x3s2_fwd_idx");
  return MI_OK;
}
extern "C" int mi_maxpool3x3s2_bwd_idx(const uint8_t* code, const void* dy, int lddy, void* dx, int lddx, int accumulate, int N,
                                       int H, int W, int C, mi_stream_t st) {
  MI_REQUIRE(code && dy && dx && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ((uintptr_t)code & 7) == 0, "maxpool3x3s2_bwd_idx: args");
  const int64_t total = (int64_t)N * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_idx_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, code, (const __bf16*)dy,
                     lddy, (__bf16*)dx, lddx, accumulate, N, H, W, C / 8);
  MI_CHECK_LAUNCH("maxpool3x3s2_bwd_idx");
  return MI_OK;
}
extern "C" int mi_maxpool3x3s2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, mi_stream_t st) {
  MI_REQUIRE(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && N > 0 && H > 0 && W > 0, "maxpool3x3s2_fwd: args");
  const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)x,
                     ldx, (__bf16*)y, ldy, N, H, W, C / 8);
  MI_CHECK_LAUNCH("maxpool3x3s2_fwd");
  return MI_OK;
}
extern "C" int mi_maxpool3x3s2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int accumulate,
                                   int N, int H, int W, int C, mi_stream_t st) {
  MI_REQUIRE(x && dy && dx && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "maxpool3x3s2_bwd: args");
  const int64_t total = (int64_t)N * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)x,
                     ldx, (const __bf16*)dy, lddy, (__bf16*)dx, lddx, accumulate, N, H, W, C / 8);
  MI_CHECK_LAUNCH("maxpool3x3s2_bwd");
  return MI_OK;
}

// ---- nearest x2 upsample (yolov7/modeling/neck/yolo_pafpn.py:28,96,101)
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const __bf16* __restrict__ x, int ldx, __bf16* y, int ldy,
                                                             int N, int H, int W, int C8) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const bf16x8 v = *(const bf16x8*)(x + (((int64_t)n * H + oy / 2) * W + ox / 2) * ldx + c8 * 8);
    *(bf16x8*)(y + (((int64_t)n * Ho + oy) * Wo + ox) * ldy + c8 * 8) = v;
  }
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const __bf16* __restrict__ dy, int lddy, __bf16* dx,
                                                             int lddx, int accumulate, int N, int H, int W, int C8) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % C8);
    int64_t r = idx / C8;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bf16x8 v = *(const bf16x8*)(dy + (((int64_t)n * Ho + 2 * iy + a) * Wo + 2 * ix + b) * lddy + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
      }
    __bf16* dp = dx + (((int64_t)n * H + iy) * W + ix) * lddx + c8 * 8;
    if (accumulate) {
      const bf16x8 o = *(const bf16x8*)dp;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)o[e];
    }
    *(bf16x8*)dp = pack8(acc);
  }
}

extern "C" int mi_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C,
                                 mi_stream_t st) {
  MI_REQUIRE(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "upsample_fwd: args");
  const int64_t total = (int64_t)N * 4 * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)x,
                     ldx, (__bf16*)y, ldy, N, H, W, C / 8);
  MI_CHECK_LAUNCH("upsample_fwd");
  return MI_OK;
}
extern "C" int mi_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int accumulate, int N, int H, int W,
                                 int C, mi_stream_t st) {
  MI_REQUIRE(dy && dx && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "upsample_bwd: args");
  const int64_t total = (int64_t)N * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)st, (const __bf16*)dy,
                     lddy, (__bf16*)dx, lddx, accumulate, N, H, W, C / 8);
  MI_CHECK_LAUNCH("upsample_bwd");
  return MI_OK;
}

// ---- SPP max-pools k = 5, 9, 13, stride 1, "same" padding (wrappers.py:150-153), separable with argmax.
// One block owns the H x W plane of 8 channels of one image (20x20 at 640).  max over a (2r+1)^2 window =
// vertical max of horizontal maxes; the argmax is kept as two codes so the backward pass is two gathers
// (54 checks per pixel instead of 169..507):
//   dxc[k][pixel][c] : dx+6 of the first maximum of row segment [x-r, x+r]            (horizontal pass)
//   dyc[k][pixel][c] : dy+6 of the first row whose horizontal max equals the window max (vertical pass)
// "first" = the element F.max_pool2d / the brute-force row-major scan would pick (strict > while scanning up).
// idx buffer: [dyc 3 planes][dxc 3 planes], each N*H*W*C bytes.
// SPP_T threads per plane: at 640x640 a plane is 20x20 = 400 pixels - one pixel per thread in a single round (with 256
// threads the second round ran 144 of them), and 16 waves per CU instead of 8 for the same 2 blocks
#define SPP_T 512
__global__ __launch_bounds__(SPP_T) void spp_fwd_plane_kernel(const __bf16* __restrict__ x, int ldx, __bf16* y5,
                                                            __bf16* y9, __bf16* y13, int ldy, uint8_t* idx, int N,
                                                            int H, int W, int C8) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W, C = C8 * 8;
  bf16x8* Xp = (bf16x8*)smem;          // [HW] input plane
  bf16x8* Hv = Xp + HW;                // [HW] horizontal max values of the current k
  uint64_t* Hc = (uint64_t*)(Hv + HW); // [HW] their dx codes
  // a block touches 16 bytes of every 128-byte line of its plane; the 8 blocks that share those lines (adjacent c8) must sit
  // on the SAME XCD to meet in its L2 - block ids go round-robin over the 8 XCDs, so give each XCD a contiguous range of
  // (image, channel group) pairs (measured before: every line fetched / written back by 8 different L2s, 47 us)
  const int lb = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int c8 = lb % C8, n = lb / C8;
  const int64_t npix = (int64_t)N * HW;
  const __bf16* xb = x + ((int64_t)n * HW) * ldx + c8 * 8;
  for (int p = threadIdx.x; p < HW; p += SPP_T) Xp[p] = *(const bf16x8*)(xb + (int64_t)p * ldx);
  __syncthreads();
  // gridDim.y == 3 (round 6): the three window sizes are three BLOCKS of the plane (each loads it: 3.3 MB in all) - the
  // kernel is bound by the serial chain inside a block (window loads, barrier, window loads per size), and a third of the
  // chain on three times as many resident blocks hides it better than one block walking all three
  const int k_lo = gridDim.y == 3 ? (int)blockIdx.y : 0, k_hi = gridDim.y == 3 ? (int)blockIdx.y + 1 : 3;
  for (int k = k_lo; k < k_hi; ++k) {
    const int r = 2 + 2 * k;
    for (int p = threadIdx.x; p < HW; p += SPP_T) {
      const int py = p / W, px = p - py * W;
      float m[8];
      uint8_t am[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; am[e] = 6; }
      for (int dx = -r; dx <= r; ++dx) {
        const int xx = px + dx;
        if (xx < 0 || xx >= W) continue;
        const bf16x8 v = Xp[py * W + xx];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          if (f > m[e]) { m[e] = f; am[e] = (uint8_t)(dx + 6); }
        }
      }
      Hv[p] = pack8(m);
      uint64_t pk = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) pk |= (uint64_t)am[e] << (8 * e);
      Hc[p] = pk;
      *(uint64_t*)(idx + ((int64_t)(3 + k) * npix + (int64_t)n * HW + p) * C + c8 * 8) = pk;
    }
    __syncthreads();
    __bf16* yk = k == 0 ? y5 : (k == 1 ? y9 : y13);
    for (int p = threadIdx.x; p < HW; p += SPP_T) {
      const int py = p / W, px = p - py * W;
      float m[8];
      uint8_t am[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; am[e] = 6; }
      for (int dy = -r; dy <= r; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
        const bf16x8 v = Hv[yy * W + px];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          if (f > m[e]) { m[e] = f; am[e] = (uint8_t)(dy + 6); }
        }
      }
      const int64_t pix = (int64_t)n * HW + p;
      *(bf16x8*)(yk + pix * ldy + c8 * 8) = pack8(m);
      uint64_t pk = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) pk |= (uint64_t)am[e] << (8 * e);
      *(uint64_t*)(idx + ((int64_t)k * npix + pix) * C + c8 * 8) = pk;
    }
    __syncthreads();
  }
}

// backward: gH_k[q] = sum over outputs o = q - (dy,0) with dyc_k[o] == dy of g_k[o];  gx[p] = sum_k sum over
// q = p - (0,dx) with dxc_k[q] == dx of gH_k[q].  Gathers in a fixed order: deterministic.
__global__ __launch_bounds__(SPP_T) void spp_bwd_plane_kernel(const __bf16* __restrict__ d5, const __bf16* __restrict__ d9,
                                                            const __bf16* __restrict__ d13, int lddy,
                                                            const uint8_t* __restrict__ idx, __bf16* dx, int lddx,
                                                            int accumulate, int N, int H, int W, int C8) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W, C = C8 * 8;
  bf16x8* Gp = (bf16x8*)smem;            // [HW] out-gradient of pool k
  uint64_t* Dy = (uint64_t*)(Gp + HW);   // [HW] vertical codes
  uint64_t* Dx = Dy + HW;                // [HW] horizontal codes
  float* Gh = (float*)(Dx + HW);         // [HW][8] gradient w.r.t. the horizontal maxes
  // a block touches 16 bytes of every 128-byte line of its plane; the 8 blocks that share those lines (adjacent c8) must sit
  // on the SAME XCD to meet in its L2 - block ids go round-robin over the 8 XCDs, so give each XCD a contiguous range of
  // (image, channel group) pairs (measured before: every line fetched / written back by 8 different L2s, 47 us)
  const int lb = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int c8 = lb % C8, n = lb / C8;
  const int64_t npix = (int64_t)N * HW;
  float acc[2][8];  // up to 2 * SPP_T pixels per plane
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  for (int k = 0; k < 3; ++k) {
    const int r = 2 + 2 * k;
    const __bf16* dk = k == 0 ? d5 : (k == 1 ? d9 : d13);
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += SPP_T) {
      const int64_t pix = (int64_t)n * HW + p;
      Gp[p] = *(const bf16x8*)(dk + pix * lddy + c8 * 8);
      Dy[p] = *(const uint64_t*)(idx + ((int64_t)k * npix + pix) * C + c8 * 8);
      Dx[p] = *(const uint64_t*)(idx + ((int64_t)(3 + k) * npix + pix) * C + c8 * 8);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < HW; q += SPP_T) {
      const int qy = q / W, qx = q - qy * W;
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = 0.f;
      for (int dy = -r; dy <= r; ++dy) {
        const int oy = qy - dy;
        if (oy < 0 || oy >= H) continue;
        const int o = oy * W + qx;
        const uint64_t pk = Dy[o];
        const bf16x8 gv = Gp[o];
        const uint8_t code = (uint8_t)(dy + 6);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((uint8_t)(pk >> (8 * e)) == code) g[e] += (float)gv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) Gh[q * 8 + e] = g[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = threadIdx.x + i * SPP_T;
      if (p >= HW) continue;
      const int py = p / W, px = p - py * W;
      for (int dxx = -r; dxx <= r; ++dxx) {
        const int qx = px - dxx;
        if (qx < 0 || qx >= W) continue;
        const int q = py * W + qx;
        const uint64_t pk = Dx[q];
        const uint8_t code = (uint8_t)(dxx + 6);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((uint8_t)(pk >> (8 * e)) == code) acc[i][e] += Gh[q * 8 + e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = threadIdx.x + i * SPP_T;
    if (p >= HW) continue;
    __bf16* op_ = dx + ((int64_t)n * HW + p) * lddx + c8 * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = acc[i][e];
    if (accumulate) {
      const bf16x8 ov = *(const bf16x8*)op_;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += (float)ov[e];
    }
    *(bf16x8*)op_ = pack8(o);
  }
}

extern "C" int mi_spp_pool_fwd(const void* x, int ldx, void* y5, void* y9, void* y13, int ldy, uint8_t* idx, int N,
                               int H, int W, int C, mi_stream_t st) {
  MI_REQUIRE(x && y5 && y9 && y13 && idx && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "spp_fwd: args");
  const size_t lds = (size_t)H * W * 40;
  MI_REQUIRE(lds <= 160 * 1024, "spp_fwd: plane %dx%d too large for the LDS kernel", H, W);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)spp_fwd_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)spp_bwd_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const char* sk = getenv("MI_SPP_SPLIT");        // (read per call: the tests compare the two forms in one process)
  const int split = sk ? atoi(sk) : 1;
  hipLaunchKernelGGL(spp_fwd_plane_kernel, dim3(N * (C / 8), split ? 3 : 1), dim3(SPP_T), lds, (hipStream_t)st, (const __bf16*)x, ldx,
                     (__bf16*)y5, (__bf16*)y9, (__bf16*)y13, ldy, idx, N, H, W, C / 8);
  MI_CHECK_LAUNCH("spp_fwd");
  return MI_OK;
}
extern "C" int mi_spp_pool_bwd(const void* dy5, const void* dy9, const void* dy13, int lddy, const uint8_t* idx,
                               void* dx, int lddx, int accumulate, int N, int H, int W, int C, mi_stream_t st) {
  MI_REQUIRE(dy5 && dy9 && dy13 && idx && dx && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "spp_bwd: args");
  MI_REQUIRE(H * W <= 2 * SPP_T, "spp_bwd: plane %dx%d > %d pixels (register accumulators)", H, W, 2 * SPP_T);
  const size_t lds = (size_t)H * W * 64;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)spp_fwd_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)spp_bwd_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(spp_bwd_plane_kernel, dim3(N * (C / 8)), dim3(SPP_T), lds, (hipStream_t)st, (const __bf16*)dy5,
                     (const __bf16*)dy9, (const __bf16*)dy13, lddy, idx, (__bf16*)dx, lddx, accumulate, N, H, W, C / 8);
  MI_CHECK_LAUNCH("spp_bwd");
  return MI_OK;
}

// ---- strided copy / accumulate
__global__ __launch_bounds__(256) void copy_bf16_kernel(const __bf16* __restrict__ src, int lds_, __bf16* dst, int ldd,
                                                        int accumulate, int64_t npix, int C8) {
  const int64_t total = npix * C8;
  for (int64_t idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t pix = idx / C8;
    const int c8 = (int)(idx - pix * C8);
    const bf16x8 v = *(const bf16x8*)(src + pix * lds_ + c8 * 8);
    __bf16* dp = dst + pix * ldd + c8 * 8;
    if (accumulate) {
      const bf16x8 o = *(const bf16x8*)dp;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (float)o[e] + (float)v[e];
      *(bf16x8*)dp = pack8(f);
    } else {
      *(bf16x8*)dp = v;
    }
  }
}
extern "C" int mi_copy_bf16(const void* src, int lds_, void* dst, int ldd, int accumulate, int64_t npix, int C,
                            mi_stream_t st) {
  MI_REQUIRE(src && dst && C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "copy_bf16: args");
  hipLaunchKernelGGL(copy_bf16_kernel, dim3(ew_blocks(npix * (C / 8))), dim3(256), 0, (hipStream_t)st,
                     (const __bf16*)src, lds_, (__bf16*)dst, ldd, accumulate, npix, C / 8);
  MI_CHECK_LAUNCH("copy_bf16");
  return MI_OK;
}

// ---- column sums (bias gradients of the prediction convs): two stages, fixed summation order
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const __bf16* __restrict__ x, int ldx, int64_t npix,
                                                            int C8N, float* __restrict__ part) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int c8 = tid % C8N, pl = tid / C8N, PL = 256 / C8N;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (pl < PL)
    for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < npix; p += (int64_t)gridDim.x * PL) {
      const bf16x8 v = *(const bf16x8*)(x + p * ldx + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = (pl < PL) ? s[e] : 0.f;
  __syncthreads();
  const int C = C8N * 8;
  if (tid < C) {
    const int cc8 = tid >> 3, e = tid & 7;
    float a = 0.f;
    for (int q = 0; q < PL; ++q) a += red[(q * C8N + cc8) * 8 + e];
    part[(size_t)blockIdx.x * C + tid] = a;
  }
}
__global__ __launch_bounds__(128) void colsum_stage2_kernel(const float* __restrict__ part, int nblk, int CP, int C,
                                                            float* out, int accumulate) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int b = 0; b < nblk; ++b) a += part[(size_t)b * CP + c];
  out[c] = accumulate ? out[c] + a : a;
}
extern "C" int mi_colsum_bf16(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws,
                              mi_stream_t st) {
  MI_REQUIRE(x && out && ws && C > 0 && C <= 128, "colsum: C %d (<=128)", C);
  const int CP = (C + 7) / 8 * 8;  // channels readable (the map is padded to a multiple of 8)
  const int C8N = CP / 8;
  MI_REQUIRE(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ldx >= CP, "colsum: alignment / ld");
  hipStream_t s = (hipStream_t)st;
  const int PL = 256 / C8N;
  int nblk = (int)((npix + PL * 8 - 1) / (PL * 8));
  if (nblk > 128) nblk = 128;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3(nblk), dim3(256), 0, s, (const __bf16*)x, ldx, npix, C8N, ws);
  MI_CHECK_LAUNCH("colsum1");
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3(1), dim3(128), 0, s, ws, nblk, CP, C, out, accumulate);
  MI_CHECK_LAUNCH("colsum2");
  return MI_OK;
}

// any channel count in ONE launch pair (bias gradients of the transformer's Linear layers: 2048 channels were 32 x 2
// launches of the 64-channel form above - 1100 launches per DETR step): grid.y = 64-channel chunk
__global__ __launch_bounds__(256) void colsum_wide_stage1_kernel(const __bf16* __restrict__ x, int ldx, int64_t npix, int CP,
                                                                 float* __restrict__ part, int sq) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  const int C8N = (CP - c0 >= 64) ? 8 : (CP - c0) / 8, PL = 256 / C8N;
  const int c8 = tid % C8N, pl = tid / C8N;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (pl < PL)
    for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < npix; p += (int64_t)gridDim.x * PL) {
      const bf16x8 v = *(const bf16x8*)(x + p * ldx + c0 + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += sq ? (float)v[e] * (float)v[e] : (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = (pl < PL) ? s[e] : 0.f;
  __syncthreads();
  if (tid < C8N * 8) {
    const int cc8 = tid >> 3, e = tid & 7;
    float a = 0.f;
    for (int q = 0; q < PL; ++q) a += red[(q * C8N + cc8) * 8 + e];
    part[(size_t)blockIdx.x * CP + c0 + tid] = a;
  }
}
__global__ __launch_bounds__(128) void colsum_wide_stage2_kernel(const float* __restrict__ part, int nblk, int CP, int C,
                                                                 float* out, int accumulate) {
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int b = 0; b < nblk; ++b) a += part[(size_t)b * CP + c];
  out[c] = accumulate ? out[c] + a : a;
}
// one launch: stage 1 as above, then the LAST block of each 64-channel chunk to arrive (a counter per chunk) sums the chunk's
// partials in block order - deterministic, and no second launch (143 bias gradients per DETR step).  The partials cross
// blocks as agent-scope atomic stores / loads (performed memory-side: no cache maintenance needed, cf. bn_grid_barrier).
// The counters are a library-wide static: launches of this kernel must not overlap (they are issued on one stream).
#define COLSUM_MAX_CHUNKS 64
static __device__ unsigned g_colsum_cnt[COLSUM_MAX_CHUNKS];
__global__ __launch_bounds__(256) void colsum_wide_fused_kernel(const __bf16* __restrict__ x, int ldx, int64_t npix, int CP, int C,
                                                                float* __restrict__ part, float* out, int accumulate, int sq) {
  __shared__ float red[256 * 8];
  __shared__ int s_last;
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  const int C8N = (CP - c0 >= 64) ? 8 : (CP - c0) / 8, PL = 256 / C8N;
  const int c8 = tid % C8N, pl = tid / C8N;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (pl < PL)
    for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < npix; p += (int64_t)gridDim.x * PL) {
      const bf16x8 v = *(const bf16x8*)(x + p * ldx + c0 + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += sq ? (float)v[e] * (float)v[e] : (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = (pl < PL) ? s[e] : 0.f;
  __syncthreads();
  if (tid < C8N * 8) {
    const int cc8 = tid >> 3, e = tid & 7;
    float a = 0.f;
    for (int q = 0; q < PL; ++q) a += red[(q * C8N + cc8) * 8 + e];
    __hip_atomic_store(part + (size_t)blockIdx.x * CP + c0 + tid, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // every storing wave waits for ITS stores to be acknowledged (the barrier alone orders the waves, not their memory
  // traffic: thread 0's counter increment below must not become visible before another wave's partials have landed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
    s_last = __hip_atomic_fetch_add(&g_colsum_cnt[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(&g_colsum_cnt[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // four threads per channel, each summing every fourth block's partial, four loads in flight (a single chain of up to 128
  // memory-side loads was 8 us); the order of the additions is fixed by (slice, block), not by arrival
  {
    const int c = tid & 63, q = tid >> 6, nbk = (int)gridDim.x;
    float a = 0.f;
    if (c < C8N * 8) {
      int b = q;
      for (; b + 12 < nbk; b += 16) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __hip_atomic_load(part + (size_t)(b + 4 * u) * CP + c0 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a += (v[0] + v[1]) + (v[2] + v[3]);
      }
      for (; b < nbk; b += 4) a += __hip_atomic_load(part + (size_t)b * CP + c0 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    red[q * 64 + c] = a;
    __syncthreads();
    if (tid < C8N * 8 && c0 + tid < C) {
      const float t = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
      out[c0 + tid] = accumulate ? out[c0 + tid] + t : t;
    }
  }
}
extern "C" int64_t mi_colsum_wide_ws_bytes(int C) { return (int64_t)128 * ((C + 7) / 8 * 8) * 4; }
static int colsum_wide_launch(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws, int sq,
                              mi_stream_t st) {
  MI_REQUIRE(x && out && ws && C > 0, "colsum_wide: null / C %d", C);
  const int CP = (C + 7) / 8 * 8;
  MI_REQUIRE(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ldx >= CP, "colsum_wide: alignment / ld");
  hipStream_t s = (hipStream_t)st;
  int nblk = (int)((npix + 255) / 256);
  if (nblk > 128) nblk = 128;
  if (nblk < 1) nblk = 1;
  const char* two_e = getenv("MI_COLSUM_TWO_STAGE");   // (read per call: the tests compare the two forms in one process)
  const int two = two_e ? atoi(two_e) : 0;
  if ((CP + 63) / 64 <= COLSUM_MAX_CHUNKS && !two) {
    hipLaunchKernelGGL(colsum_wide_fused_kernel, dim3(nblk, (CP + 63) / 64), dim3(256), 0, s, (const __bf16*)x, ldx, npix, CP, C, ws,
                       out, accumulate, sq);
    MI_CHECK_LAUNCH("colsum_wide");
    return MI_OK;
  }
  hipLaunchKernelGGL(colsum_wide_stage1_kernel, dim3(nblk, (CP + 63) / 64), dim3(256), 0, s, (const __bf16*)x, ldx, npix, CP, ws, sq);
  MI_CHECK_LAUNCH("colsum_wide1");
  hipLaunchKernelGGL(colsum_wide_stage2_kernel, dim3((C + 127) / 128), dim3(128), 0, s, ws, nblk, CP, C, out, accumulate);
  MI_CHECK_LAUNCH("colsum_wide2");
  return MI_OK;
}
extern "C" int mi_colsum_bf16_wide(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws,
                                   mi_stream_t st) {
  return colsum_wide_launch(x, ldx, npix, C, out, accumulate, ws, 0, st);
}
// column sums of SQUARES (fp32 products of the bf16 values): the ||sigmoid(mask)||^2 term of the dice scores of SparseInst's
// matcher / IAM normalisers over 25 600 pixels - a torch reduction of that shape splits across blocks, and that form was
// measured NOT replay-safe inside a captured hipGraph on this stack (tools/si_graph_debug3.py)
extern "C" int mi_colsumsq_bf16_wide(const void* x, int ldx, int64_t npix, int C, float* out, int accumulate, float* ws,
                                     mi_stream_t st) {
  return colsum_wide_launch(x, ldx, npix, C, out, accumulate, ws, 1, st);
}

// ---- out[r][:] = g[r][:] * scale[r]  (fp32; the weight gradient of a convolution whose weight image carried a folded per-Cout
// factor: d/dW = scale[co] * d/dW').  rowlen % 4 == 0 takes the 16-byte path.
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ g, const float* __restrict__ sc, float* out,
                                                         int rows, int rowlen) {
  const int64_t n = (int64_t)rows * rowlen;
  if ((rowlen & 3) == 0) {
    const int64_t n4 = n >> 2;
    const int rl4 = rowlen >> 2;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float f = sc[i / rl4];
      float4 v = ((const float4*)g)[i];
      v.x *= f; v.y *= f; v.z *= f; v.w *= f;
      ((float4*)out)[i] = v;
    }
  } else {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = g[i] * sc[i / rowlen];
  }
}
extern "C" int mi_scale_rows_f32(const float* g, const float* scale, float* out, int rows, int rowlen, mi_stream_t st) {
  MI_REQUIRE(g && scale && out && rows > 0 && rowlen > 0, "scale_rows: args");
  MI_REQUIRE(((uintptr_t)g % 16) == 0 && ((uintptr_t)out % 16) == 0, "scale_rows: alignment");
  int64_t blocks = ((int64_t)rows * rowlen / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)st, g, scale, out, rows, rowlen);
  MI_CHECK_LAUNCH("scale_rows");
  return MI_OK;
}

// ---- SGD with momentum + weight decay over a flat arena (torch.optim.SGD semantics: dampening 0, no nesterov)
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* __restrict__ g, float* m,
                                                  const mi_sgd_seg* __restrict__ segs, float momentum,
                                                  float grad_scale, int first_step) {
  const mi_sgd_seg sg = segs[blockIdx.x];
  const float wd = sg.weight_decay, lr = sg.lr;
  // segments start 16-byte aligned (the arena pads every parameter to 4 floats): float4 body, 2 vectors per trip
  const int64_t n4 = ((sg.offset & 3) == 0) ? (sg.count >> 2) : 0;
  f32x4* p4 = (f32x4*)(p + sg.offset);
  const f32x4* g4 = (const f32x4*)(g + sg.offset);
  f32x4* m4 = (f32x4*)(m + sg.offset);
  for (int64_t i = threadIdx.x; i < n4; i += 512) {
    const bool two = i + 256 < n4;
    const f32x4 pa = p4[i], ga = g4[i], ma = first_step ? f32x4{0.f, 0.f, 0.f, 0.f} : m4[i];
    f32x4 pb = {0.f, 0.f, 0.f, 0.f}, gb = pb, mb = pb;
    if (two) {
      pb = p4[i + 256];
      gb = g4[i + 256];
      if (!first_step) mb = m4[i + 256];
    }
    f32x4 da = ga * grad_scale + pa * wd;
    f32x4 ba = first_step ? da : ma * momentum + da;
    m4[i] = ba;
    p4[i] = pa - ba * lr;
    if (two) {
      f32x4 db = gb * grad_scale + pb * wd;
      f32x4 bb = first_step ? db : mb * momentum + db;
      m4[i + 256] = bb;
      p4[i + 256] = pb - bb * lr;
    }
  }
  for (int64_t i = n4 * 4 + threadIdx.x; i < sg.count; i += 256) {
    const int64_t k = sg.offset + i;
    const float pv = p[k];
    const float d = g[k] * grad_scale + wd * pv;
    const float b = first_step ? d : momentum * m[k] + d;
    m[k] = b;
    p[k] = pv - lr * b;
  }
}
extern "C" int mi_sgd_momentum_step(float* params, const float* grads, float* momentum_buf,
                                    const mi_sgd_seg* segs_dev, int nseg, float momentum, float grad_scale,
                                    int first_step, mi_stream_t st) {
  MI_REQUIRE(params && grads && momentum_buf && segs_dev && nseg > 0, "sgd: args");
  hipLaunchKernelGGL(sgd_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)st, params, grads, momentum_buf, segs_dev,
                     momentum, grad_scale, first_step);
  MI_CHECK_LAUNCH("sgd");
  return MI_OK;
}
