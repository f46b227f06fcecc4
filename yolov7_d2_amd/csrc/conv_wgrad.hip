// Convolution weight gradient on the gfx950 matrix cores: split-K over pixel tiles, no atomics.
//
//   g[co][ci][tap] = sum_pixels dy[p][co] * x[p*stride + tap][ci]
//
// replaces the conv wgrad ATen/cuDNN reaches from BaseConv (layers/wrappers.py:60-83) and the
// prediction convs (head/yolox_head.py:103-129) of the reference.
//
// Design (MI355X):
//   * GEMM view: M = cout, N = cin (x taps), K = pixels.  K is the slow dimension of both NHWC
//     operands, so both MFMA fragments come out of row-major [pixel][channel] LDS tiles through
//     the hardware transpose read (ds_read_b64_tr_b16) into v_mfma_f32_16x16x32_bf16.
//   * a block owns a (BCO x BCI x all taps) slab of the gradient and a contiguous range of
//     128-pixel spatial tiles (split-K).  Per tile the dy tile and the x halo tile are streamed
//     HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging), double-buffered: the loads of
//     tile t+1 are in flight while tile t is multiplied.  Rows are 32-byte-group XOR-swizzled on the
//     SOURCE address (the LDS image of an LDS-DMA is lane-linear) so the transpose reads of 8
//     consecutive rows hit 8 distinct bank groups.
//   * each wave holds (16*MI cout) x (16*NJ cin) x taps of fp32 accumulators for the whole pixel
//     range, then stores them ONCE, in fragment order (one 16-byte store per lane per 16x16 tile),
//     to a split-K workspace; a second small kernel sums the splits in a fixed order (deterministic)
//     and scatters into the OIHW fp32 gradient.  The earlier atomicAdd version spent ~19 M L2
//     atomics per launch (270 us per 3x3 layer regardless of size).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"

struct Wg2K {
  const __bf16* x;
  const __bf16* dy;
  float* part;
  int ldx, lddy, N, H, W, outH, outW, is;
  int TH, TW, tilesY, tilesX, ntiles, tps, nsplit;
  int dymin, dxmin, haloW, npixh, nqx, stage, ns;
  int toff[MI_MAX_TAPS];
  int nco, nci;
  int nrx;            // v3 layout: rows per channel group of the x tile = max(npixh, 32)
  int xmap;           // 1: XCD-aware block order (blocks sharing a pixel range are 8 ids apart: same XCD, dispatched together)
  unsigned mTW, mHW;  // ceil(65536 / TW), ceil(65536 / haloW): row / d == (row * m) >> 20 for row * d < 65536
  long long V;        // float4 vectors per split slab
  float* bpart;       // NULL, or [nsplit][bld] fp32: the bias gradient's split partials (column sums of dy), written by the cib == 0 blocks
  int bld, fix;       // fix: the blocks of an output tile sum the tile's splits themselves (wg_fixup; grouped launches, MI_WG_FIXUP=1)
  // fix-up operands (the split-K reduction's: Wg2R)
  float* g;
  const float* row_scale;
  int Cout, Cin, accumulate, pad_;
  int ra, rb;         // xmap 2: the splits that do not fill a group of 8 hand every XCD a compact ra x rb block of (cout, cin) tiles (0: off)
  long long cnt_rel;  // byte offset from the group's job array to this job's tile counters (64 words per output tile)
};

__device__ uint4 g_mi_zero_page[4];

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_read2(const char* base0, const char* base1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// 16-byte LDS-DMA: LDS[lds_off + lane*16 .. +16) = *g   (lds_off wave-uniform).  Issued from inline asm so the
// compiler neither tracks it on vmcnt nor fences later ds_reads with vmcnt(0): the loop below waits explicitly.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory", "m0");
}

// wait until at most n LDS-DMA loads of this wave are outstanding (n wave-uniform; loads retire in order).
// s_waitcnt takes an immediate, hence the uniform branch tree; a smaller count than asked is always safe.
__device__ __forceinline__ void wait_vmcnt(int n) {
#define MI_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n < 0 ? 0 : (n > 30 ? 30 : n)) {
    MI_VMC(0) MI_VMC(1) MI_VMC(2) MI_VMC(3) MI_VMC(4) MI_VMC(5) MI_VMC(6) MI_VMC(7) MI_VMC(8) MI_VMC(9) MI_VMC(10)
    MI_VMC(11) MI_VMC(12) MI_VMC(13) MI_VMC(14) MI_VMC(15) MI_VMC(16) MI_VMC(17) MI_VMC(18) MI_VMC(19) MI_VMC(20)
    MI_VMC(21) MI_VMC(22) MI_VMC(23) MI_VMC(24) MI_VMC(25) MI_VMC(26) MI_VMC(27) MI_VMC(28) MI_VMC(29) MI_VMC(30)
  }
#undef MI_VMC
}

template <int NG>
__device__ __forceinline__ int swz32(int row) {  // 32-byte group permutation of a row
  if (NG == 1) return 0;
  if (NG == 2) return (row >> 2) & 1;
  if (NG == 4) return (row >> 1) & 3;
  return row & 7;  // NG == 8
}

// Bias gradient inside the weight-gradient launch (mi_wgrad_desc.gbias): the blocks of input-channel tile 0 add up the dy
// rows of their pixel range - read again from global memory right after the tile's LDS-DMA asked for them (L2 hits; the
// LDS images of the two kernels are swizzled differently, this read is layout-free) - and leave one partial row per split;
// the split-K reduction adds the splits in a fixed order.  Thread (chunk = tid % (BCO / 8), row lane = tid / (BCO / 8)).
template <int NTHR, int BCO>
struct WgBias {
  static constexpr int NCH = BCO / 8, RL = NTHR / NCH;
  float s[8];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
  }
  template <int TP>
  __device__ __forceinline__ void tile(const Wg2K& p, const int img, const int ty0, const int tx0, const int co0, const int TPv) {
    const int tid = threadIdx.x, ch = tid % NCH, rl = tid / NCH;
    if (rl >= RL) return;
    const __bf16* const dyb = p.dy + ((size_t)img * p.outH * p.outW) * (size_t)p.lddy + co0 + ch * 8;
    constexpr int NR = (TP + RL - 1) / RL;     // rows of a tile per thread: all loads in flight before the first add
    bf16x8 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = rl + k * RL;
      const int ty = (int)(((unsigned)row * p.mTW) >> 20);
      const int tx = row - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      const bool ok = (row < TPv) & (oy < p.outH) & (ox < p.outW);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = (__bf16)0.f;
      if (ok) v[k] = *(const bf16x8*)(dyb + (size_t)(oy * p.outW + ox) * p.lddy);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[k][e];
  }
  // after the block's last barrier: fold the row lanes through LDS (fixed order) and write this split's partial row
  __device__ __forceinline__ void finish(const Wg2K& p, float* red, const int split, const int co0) {
    const int tid = threadIdx.x, ch = tid % NCH, rl = tid / NCH;
    __syncthreads();
    if (rl < RL) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rl * BCO + ch * 8 + e] = s[e];
    }
    __syncthreads();
    if (tid < BCO) {
      float a = 0.f;
      for (int q = 0; q < RL; ++q) a += red[q * BCO + tid];
      p.bpart[(size_t)split * p.bld + co0 + tid] = a;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Stream-K style fix-up (MI_WG_FIXUP=1, grouped launches): the split-K reduction inside the producing launch.  The nsplit
// blocks of one (cout, cin) output tile leave their slabs as agent-scope (write-through) stores, meet on the tile's arrival
// counter, and then EVERY one of them sums 1 / nsplit of the tile over all splits - in wgrad2_reduce_body<4>'s order, bit
// for bit - and scatters it into the OIHW gradient: the reduce grid and its launch boundary go, the partials are read while
// the memory-side cache still holds them, and the read runs as wide as the launch (a LAST-ARRIVER fix-up would read
// nsplit x 147 KB from one CU: slower than the reduce grid).  The wait needs every block of a tile resident: the grouped
// plan sizes each grid to one round of resident blocks (mi_conv2d_wgrad_group_plan); a wait that does not end in
// WG_FIX_SPIN_LIMIT polls gives up and writes NaN into its share of the gradient instead of hanging the GPU.
// No cache maintenance (see conv_bn.h): slabs and counters are written / read with agent-scope accesses (sc1), the stores
// are acknowledged (s_waitcnt vmcnt(0)) before the arrival is counted.  The counters reset themselves: the block that
// finishes a tile's reduction last clears them for the next replay.
#define WG_FIX_SPIN_LIMIT (1 << 20)
#define WG_FIX_U 3
// 16-byte agent-scope accesses to a layer's slabs: raw buffer instructions with the sc1 cache-policy bit (aux 16 on gfx940+:
// what the compiler emits for a relaxed agent-scope atomic, 128 bits wide), the descriptor over the layer's workspace, byte
// offsets < 2^31 (checked by the plan).  Compiler-tracked: waits and store-data hazards are its business.
#define WG_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_slab_rsrc(const float* part) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st_agent16(__amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff, const f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, WG_SC1);
}
__device__ __forceinline__ f32x4 ld_agent16(__amdgpu_buffer_rsrc_t r, const unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, WG_SC1));
}

// fragment-order float4 index v of a layer's slab -> 4 couts x 1 cin x 1 tap of the OIHW gradient (the split-K reduction's
// epilogue, shared by the reduce kernels and the fix-up)
__device__ __forceinline__ void wg_scatter(float* g, const float* row_scale, const int accumulate, const int NT, const int MI,
                                           const int NJ, const int WCO, const int WCI, const int nci, const int Cout,
                                           const int Cin, const long long v, const f32x4 sum) {
  const int lane = (int)(v & 63);
  long long r = v >> 6;
  const int j = (int)(r % NJ); r /= NJ;
  const int i = (int)(r % MI); r /= MI;
  const int tap = (int)(r % NT); r /= NT;
  const int NW = WCO * WCI;
  const int wave = (int)(r % NW); r /= NW;
  const int cib = (int)(r % nci);
  const int cob = (int)(r / nci);
  const int wco = wave / WCI, wci = wave % WCI;
  const int ci = cib * (16 * NJ * WCI) + (wci * NJ + j) * 16 + (lane & 15);
  const int cobase = cob * (16 * MI * WCO) + (wco * MI + i) * 16 + 4 * (lane >> 4);
  if (ci >= Cin) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int co = cobase + e;
    if (co < Cout) {
      float* dst = g + ((size_t)co * Cin + ci) * NT + tap;
      // (__fmul_rn / __fadd_rn: never contracted into an fma - the reduce kernels and the fix-up must round alike)
      const float val = row_scale ? __fmul_rn(sum[e], row_scale[co]) : sum[e];
      *dst = accumulate ? __fadd_rn(*dst, val) : val;
    }
  }
}

template <int NT, int MI, int NJ, int WCO, int WCI>
__device__ __forceinline__ void wg_fixup(const Wg2K& p, unsigned* const cnt, const int s, const int pl, char* const smem) {
  constexpr int NW = WCO * WCI, NTH = NW * 64, OUTS = NTH / 4, U = WG_FIX_U;
  constexpr int VT = NW * NT * MI * NJ * 64;          // float4s of one output tile in a slab
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this block's slab stores are acknowledged
  __syncthreads();                                    // ... all waves'; the tile buffers in LDS are dead from here
  unsigned* const A = cnt + (size_t)pl * 64;          // arrivals of this tile; A + 32: finished reductions
  int* const flag = (int*)smem;
  f32x4* const red = (f32x4*)(smem + 16);             // [3][U][OUTS]
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(A, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1, spins = 0;
    while (__hip_atomic_load(A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.nsplit) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > WG_FIX_SPIN_LIMIT) {   // a block of this tile never became resident
        ok = 0;
        // sticky, never cleared by the device: the host finds it (Plan.check_bn_barriers reads word 33 of every tile record)
        __hip_atomic_store(A + 33, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    *flag = ok;
  }
  __syncthreads();
  const bool ok = *flag != 0;
  // this block's share of the tile: `per` consecutive float4s (a multiple of OUTS)
  const int per = ((VT + p.nsplit - 1) / p.nsplit + OUTS - 1) / OUTS * OUTS;
  const int beg = s * per, end = min(VT, beg + per);
  const int o = threadIdx.x % OUTS, sl = threadIdx.x / OUTS;
  const long long vbase = (long long)pl * VT;
  const __amdgpu_buffer_rsrc_t rs = wg_slab_rsrc(p.part);
  const unsigned Vb = (unsigned)(p.V * 16);           // bytes between consecutive splits of one float4
  const int nsplit = p.nsplit;
  for (int b0 = beg; b0 < end; b0 += U * OUTS) {
    f32x4 s0[U], s1[U], s2[U], s3[U];
    bool valid[U];
    unsigned src[U];                                  // byte offset of split 0 of this thread's float4
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = b0 + u * OUTS + o;
      valid[u] = q < end;
      src[u] = (unsigned)((vbase + (valid[u] ? q : beg)) * 16);
      s0[u] = s1[u] = s2[u] = s3[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // wgrad2_reduce_body<4>: lane sl takes splits sl, sl + 4, ... - four at a time into s0..s3, the rest into s0
    int k = sl;
    for (; k + 12 < nsplit; k += 16) {
      f32x4 a[U], b[U], c[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = ld_agent16(rs, src[u] + (unsigned)k * Vb);
        b[u] = ld_agent16(rs, src[u] + (unsigned)(k + 4) * Vb);
        c[u] = ld_agent16(rs, src[u] + (unsigned)(k + 8) * Vb);
        d[u] = ld_agent16(rs, src[u] + (unsigned)(k + 12) * Vb);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { s0[u] += a[u]; s1[u] += b[u]; s2[u] += c[u]; s3[u] += d[u]; }
    }
    for (; k < nsplit; k += 4) {
      f32x4 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = ld_agent16(rs, src[u] + (unsigned)k * Vb);
#pragma unroll
      for (int u = 0; u < U; ++u) s0[u] += a[u];
    }
    f32x4 sum[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sum[u] = (s0[u] + s1[u]) + (s2[u] + s3[u]);
      if (sl > 0) red[((sl - 1) * U + u) * OUTS + o] = sum[u];
    }
    __syncthreads();
    if (sl == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int q = 0; q < 3; ++q) sum[u] += red[(q * U + u) * OUTS + o];
        if (!ok) sum[u] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        if (valid[u])
          wg_scatter(p.g, p.row_scale, p.accumulate, NT, MI, NJ, WCO, WCI, p.nci, p.Cout, p.Cin,
                     vbase + b0 + u * OUTS + o, sum[u]);
      }
    }
    __syncthreads();   // red is rewritten by the next pass
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // every block of the tile has passed its wait (or given up: it still counts, so that a tile whose blocks all arrived
    // late re-arms instead of growing without bound) once it counts here: the last one re-arms the counters
    if (__hip_atomic_fetch_add(A + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsplit - 1u) {
      __hip_atomic_store(A + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// block id -> (split s, output tile pair).  Workgroup b runs on XCD b % 8 (observed placement, speed only): with the
// XCD-aware order the nco*nci blocks that read the SAME pixel range (dy re-read per cin block, x per cout block) sit
// 8 ids apart - same XCD, dispatched together, so the second reader finds the tile in that XCD's L2 instead of HBM.
// That covers the splits in whole groups of 8.  The remaining r < 8 splits (ALL of them for the layers of a per-block group:
// 1 - 4 splits of 32 - 128 tile pairs) were dealt pair by pair over the XCDs - every XCD fetched every operand tile: counters
// on the DETR-R50 step, 680 MB fetched by a launch whose operands are 125 MB (profiles/r06_wgrad_xcd_rect.txt).  xmap 2:
// the r x npairs remaining items are ordered so that the 8 residue classes of the block id own CONTIGUOUS ranges of them,
// and a range is a few ra x rb rectangles of the (cout, cin) tile grid: ra + rb operand tiles per ra * rb blocks.
template <class PK>
__device__ __forceinline__ void wg_decode(const PK& p, const int bid, int& s, int& pair) {
  const int npairs = p.nco * p.nci;
  const int s8 = p.xmap ? (p.nsplit & ~7) : 0;  // splits covered by full groups of 8
  const int full = s8 * npairs;
  if (bid < full) {
    const int k = bid >> 3;
    pair = k % npairs;
    s = (k / npairs) * 8 + (bid & 7);
    return;
  }
  const int rem = bid - full, r = p.nsplit - s8;
  if (p.ra > 0) {
    const int m = (r * npairs) >> 3;                   // items per XCD (host: r * npairs % 8 == 0, ra * rb | gcd(m, npairs))
    const int idx = (rem & 7) * m + (rem >> 3);
    const int so = idx / npairs, pi = idx - so * npairs;
    const int rsz = p.ra * p.rb, c = pi / rsz, w = pi - c * rsz;
    const int nca = p.nco / p.ra;
    const int ca = c % nca, cb = c / nca;
    const int da = w % p.ra, db = w / p.ra;
    s = s8 + so;
    pair = (ca * p.ra + da) + (cb * p.rb + db) * p.nco;
    return;
  }
  s = s8 + rem % r;
  pair = rem / r;
}

#ifdef MI_WG_TIMELINE
// diagnostic build (tools/build_variant.sh tl "-DMI_WG_TIMELINE" conv_wgrad; tools/wgrad_timeline.py): wave 0 of block 0 stamps
// the phases of its first 200 steps: [step][0] loop top, [1] tile landed (counted wait), [2] barrier passed, [3] refill
// issued, [4] multiplied
__device__ long long g_wg_tl[200 * 5];
extern "C" int mi_debug_wg_timeline(long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_tl), sizeof(g_wg_tl)) == hipSuccess ? MI_OK : MI_EINVAL;
}
#define WG_TL(k) do { if (bid == 0 && threadIdx.x == 0 && it < 200) g_wg_tl[it * 5 + (k)] = wall_clock64(); } while (0)
#else
#define WG_TL(k) do { } while (0)
#endif
template <int NT, int MI, int NJ, int WCO, int WCI, int TP, bool FIX = false>
__device__ __forceinline__ void wgrad2_body(const Wg2K& p, const int bid, unsigned* const cnt = nullptr) {
  constexpr int NW = WCO * WCI, BCO = 16 * MI * WCO, BCI = 16 * NJ * WCI;
  constexpr int RDY = BCO * 2, RX = BCI * 2, KS = TP / 32;
  constexpr int CPR_DY = RDY / 16, RPI_DY = 64 / CPR_DY, NQ_DY = TP / RPI_DY, QW_DY = NQ_DY / NW;
  constexpr int CPR_X = RX / 16, RPI_X = 64 / CPR_X;
  constexpr int NG_DY = RDY / 32, NG_X = RX / 32;
  static_assert(NQ_DY % NW == 0, "dy loader split");
  static_assert(NG_DY <= 8 && NG_X <= 8, "row width");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte offset

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, t = lane & 15;
  const int wco = wave / WCI, wci = wave % WCI;

  int s, pair;
  wg_decode(p, bid, s, pair);
  const int cob = pair % p.nco, cib = pair / p.nco;
  const int co0 = cob * BCO, ci0 = cib * BCI;
  const int TPv = p.TH * p.TW;

  // ---- per-lane fragment row bases (tile invariant)
  int pA[KS][2], hb[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int P = ks * 32 + 16 * e + 4 * g + (t >> 2);
      pA[ks][e] = P * RDY + (t & 3) * 8;
      const bool v = P < TPv;
      const int ty = v ? (int)(((unsigned)P * p.mTW) >> 20) : 0;
      const int tx = v ? P - ty * p.TW : 0;
      hb[ks][e] = ty * p.is * p.haloW + tx * p.is;
    }

  f32x4 acc[NT][MI][NJ];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tpi = p.tilesY * p.tilesX;
  const int tbeg = s * p.tps;
  const int tend = min(p.ntiles, tbeg + p.tps);
  WgBias<NW * 64, BCO> bias;
  const bool do_bias = p.bpart != nullptr && cib == 0;
  bias.init();

  const char* const zero = (const char*)g_mi_zero_page;
  auto issue = [&](int tile, int st) {
    const int img = tile / tpi;
    const int rem = tile - img * tpi;
    const int tyq = rem / p.tilesX;
    const int ty0 = tyq * p.TH, tx0 = (rem - tyq * p.tilesX) * p.TW;
    const unsigned sbase = lds0 + st * p.stage;
    // dy tile: TP rows of BCO channels (32-bit element offsets: a tensor view is < 2^31 elements)
    const char* const dyb = (const char*)(p.dy + ((size_t)img * p.outH * p.outW) * (size_t)p.lddy + co0);
#pragma unroll
    for (int i = 0; i < QW_DY; ++i) {
      const int q = wave + NW * i;
      const int row = q * RPI_DY + lane / CPR_DY;
      const int chunk = (lane % CPR_DY) ^ (2 * swz32<NG_DY>(row));
      const int ty = (int)(((unsigned)row * p.mTW) >> 20);
      const int tx = row - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      const bool v = (row < TPv) & (oy < p.outH) & (ox < p.outW);
      const unsigned off = (unsigned)(((oy * p.outW + ox) * p.lddy + chunk * 8) * 2);
      glds16(v ? dyb + off : zero, sbase + q * 1024);
    }
    // x halo tile: nqx*RPI_X rows of BCI channels
    const int iy0 = ty0 * p.is + p.dymin, ix0 = tx0 * p.is + p.dxmin;
    const unsigned xbase = sbase + TP * RDY;
    const char* const xb = (const char*)(p.x + ((size_t)img * p.H * p.W) * (size_t)p.ldx + ci0);
    for (int q = wave; q < p.nqx; q += NW) {
      const int row = q * RPI_X + lane / CPR_X;
      const int chunk = (lane % CPR_X) ^ (2 * swz32<NG_X>(row));
      const int hy = (int)(((unsigned)row * p.mHW) >> 20);
      const int hx = row - hy * p.haloW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool v = (row < p.npixh) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const unsigned off = (unsigned)(((iy * p.W + ix) * p.ldx + chunk * 8) * 2);
      glds16(v ? xb + off : zero, xbase + q * 1024);
    }
  };

  // NS-stage ring: tiles it .. it+NS-2 are in flight while tile `it` is multiplied
  const int NS = p.ns;
  const int nload = QW_DY + (p.nqx - wave + NW - 1) / NW;  // LDS-DMA instructions this wave issues per tile
  for (int j = 0; j < NS - 1; ++j)
    if (tbeg + j < tend) issue(tbeg + j, j);
  int it = 0, cur = 0;  // cur = it % NS
  for (int tile = tbeg; tile < tend; ++tile, ++it) {
    const int ahead = min(NS - 2, tend - 1 - tile);
    WG_TL(0);
    wait_vmcnt(ahead * nload);
    WG_TL(1);
    __builtin_amdgcn_s_barrier();  // tile `tile` landed for every wave; the stage of tile-1 is no longer being read
    WG_TL(2);
    {
      const int nxt = tile + NS - 1;
      int st = cur - 1;
      if (st < 0) st += NS;
      if (nxt < tend) issue(nxt, st);
    }
    WG_TL(3);
    if (do_bias) {
      const int img = tile / tpi, rem = tile - img * tpi, tyq = rem / p.tilesX;
      bias.template tile<TP>(p, img, tyq * p.TH, (rem - tyq * p.tilesX) * p.TW, co0, TPv);
    }
    const char* dyB = smem + cur * p.stage;
    if (++cur == NS) cur = 0;
    const char* xB = dyB + TP * RDY;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[MI];
      int h0 = hb[ks][0], h1 = hb[ks][1];
      // keep the per-tap LDS addresses out of the loop-invariant set: hoisting all 2*KS*NT of them costs more
      // registers (spills) than recomputing ~5 VALU per transpose read beside the MFMAs
      asm volatile("" : "+v"(h0), "+v"(h1));
      const int f0 = swz32<NG_DY>(ks * 32 + 4 * g + (t >> 2)), f1 = swz32<NG_DY>(ks * 32 + 16 + 4 * g + (t >> 2));
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int cg = wco * MI + i;
        a[i] = tr_read2(dyB + pA[ks][0] + ((cg ^ f0) * 32), dyB + pA[ks][1] + ((cg ^ f1) * 32));
      }
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int r0 = h0 + p.toff[tap], r1 = h1 + p.toff[tap];
        const int x0 = r0 * RX + (t & 3) * 8, x1 = r1 * RX + (t & 3) * 8;
        const int g0 = swz32<NG_X>(r0), g1 = swz32<NG_X>(r1);
        bf16x8 b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int cg = wci * NJ + j;
          b[j] = tr_read2(xB + x0 + ((cg ^ g0) * 32), xB + x1 + ((cg ^ g1) * 32));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[tap][i][j], 0, 0, 0);
      }
    }
#ifdef MI_WG_TIMELINE
    asm volatile("s_nop 0" : "+v"(acc[0][0][0]) : : "memory");   // (the stamp follows the last MFMA's result)
#endif
    WG_TL(4);
  }
  // ---- split-K slab, fragment order: [split][cob][cib][wave][tap][i][j][lane] x float4
  f32x4* out = (f32x4*)p.part + (size_t)s * (size_t)p.V +
               ((size_t)((cob * p.nci + cib) * NW + wave) * (NT * MI * NJ)) * 64 + lane;
  if (FIX) {                         // fix-up form (its own instantiation): write-through slab, then this block's share of the tile's sum
    const __amdgpu_buffer_rsrc_t rs = wg_slab_rsrc(p.part);
    const unsigned ob = (unsigned)(((size_t)s * (size_t)p.V + ((size_t)((cob * p.nci + cib) * NW + wave) * (NT * MI * NJ)) * 64 + lane) * 16);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) st_agent16(rs, ob, (unsigned)(((tap * MI + i) * NJ + j) * 1024), acc[tap][i][j]);
    wg_fixup<NT, MI, NJ, WCO, WCI>(p, cnt, s, cob * p.nci + cib, smem);
    return;
  }
#pragma unroll
  for (int tap = 0; tap < NT; ++tap)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) out[((tap * MI + i) * NJ + j) * 64] = acc[tap][i][j];
  if (p.bpart != nullptr) {          // (block-uniform; the tile buffers are dead: the partial rows fold through them)
    if (do_bias) bias.finish(p, (float*)smem, s, co0);
  }
}

template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
__global__ __launch_bounds__(WCO* WCI * 64, 2) void wgrad2_kernel(const Wg2K p) {
  wgrad2_body<NT, MI, NJ, WCO, WCI, TP>(p, blockIdx.x);
}
// grouped launch: every layer of the step that uses this tile configuration, in one grid.  starts[j] = first block
// of job j (starts[njobs] = grid size); the job record is fetched with scalar loads (uniform address).
template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
__global__ __launch_bounds__(WCO* WCI * 64, 2) void wgrad2_group_kernel(const Wg2K* __restrict__ jobs,
                                                                         const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && starts[j + 1] <= b) ++j;
  const Wg2K p = jobs[j];
  wgrad2_body<NT, MI, NJ, WCO, WCI, TP>(p, b - starts[j]);
}
// MI_WG_MULTI (default 1): every 1x1 layer of the step in ONE grid, whatever its tile widths - the job carries them (Wg2K.pad_ =
// 16 MI + NJ) and the block branches to that instantiation of the body.  One launch (its ~11 us of fixed cost once) and one
// pool of resident slots for the split choice instead of up to nine grids of one tile configuration each.
template <int TP>
__global__ __launch_bounds__(256, 2) void wgrad2_multi_kernel(const Wg2K* __restrict__ jobs, const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && starts[j + 1] <= b) ++j;
  const Wg2K p = jobs[j];
  const int bid = b - starts[j];
  switch (p.pad_) {
    case 0x11: wgrad2_body<1, 1, 1, 2, 2, TP>(p, bid); break;
    case 0x12: wgrad2_body<1, 1, 2, 2, 2, TP>(p, bid); break;
    case 0x14: wgrad2_body<1, 1, 4, 2, 2, TP>(p, bid); break;
    case 0x21: wgrad2_body<1, 2, 1, 2, 2, TP>(p, bid); break;
    case 0x22: wgrad2_body<1, 2, 2, 2, 2, TP>(p, bid); break;
    case 0x24: wgrad2_body<1, 2, 4, 2, 2, TP>(p, bid); break;
    case 0x41: wgrad2_body<1, 4, 1, 2, 2, TP>(p, bid); break;
    case 0x42: wgrad2_body<1, 4, 2, 2, 2, TP>(p, bid); break;
    default: wgrad2_body<1, 4, 4, 2, 2, TP>(p, bid); break;
  }
}
// the same grid with the split-K reduction inside (wg_fixup): every job of the group carries fix = 1
template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
__global__ __launch_bounds__(WCO* WCI * 64, 2) void wgrad2_group_fix_kernel(const Wg2K* __restrict__ jobs,
                                                                             const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && starts[j + 1] <= b) ++j;
  const Wg2K p = jobs[j];
  wgrad2_body<NT, MI, NJ, WCO, WCI, TP, true>(p, b - starts[j], (unsigned*)((char*)const_cast<Wg2K*>(jobs) + p.cnt_rel));
}


// ---------------------------------------------------------------------------------------------------------------
// v3 LDS layout ("column-major by 32-byte channel group").  PMC of the v2 kernel on the YOLOX-s step (round 2,
// profiles/r02_sq_counters.csv): 6 VALU instructions per MFMA - the XOR-swizzled row-major layout makes every
// transpose read recompute shift / and / xor / add per lane - and the matrix pipe 30 % busy.  Here a tile is stored
// as [channel group of 16][pixel row][32 bytes]: a half-wave of a ds_read_b64_tr_b16 (8 consecutive pixel rows of one
// group) reads 256 contiguous bytes (conflict-free without a swizzle) and the address is LINEAR in the row, so
//   * the dy (A) fragments of all k-steps / cout groups are one VGPR + immediate offsets,
//   * an x (B) fragment of tap t is base[ks][e] + toff[t]*32: one v_add with a scalar per read.
// An LDS-DMA instruction (64 lanes x 16 B, lane-linear in LDS) therefore covers 32 pixel rows of ONE group: each lane
// pair fetches the 32 bytes of its row; the four groups of a 128-byte line are fetched by consecutive instructions
// of the same wave (L2 merges them).  The split-K slab format is unchanged (same reduce kernels).
template <int NT, int MI, int NJ, int WCO, int WCI, int TP, bool FIX = false>
__device__ __forceinline__ void wgrad3_body(const Wg2K& p, const int bid, unsigned* const cnt = nullptr) {
  constexpr int NW = WCO * WCI, BCO = 16 * MI * WCO, BCI = 16 * NJ * WCI;
  constexpr int GD = BCO / 16, GX = BCI / 16, KS = TP / 32, PB = TP / 32;
  constexpr int UD = PB * GD;                 // dy DMA units (32 rows x one group) per tile
  static_assert(UD % NW == 0, "dy loader split");
  constexpr int QW_DY = UD / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, t = lane & 15;
  const int wco = wave / WCI, wci = wave % WCI;

  int s, pair;
  wg_decode(p, bid, s, pair);
  const int cob = pair % p.nco, cib = pair / p.nco;
  const int co0 = cob * BCO, ci0 = cib * BCI;
  const int TPv = p.TH * p.TW;
  const int nrx = p.nrx;                         // x rows per group (npixh, at least 32; the last DMA unit overlaps its predecessor)
  const int nrb = (nrx + 31) >> 5;               // 32-row DMA units per x group

  // ---- per-lane fragment addresses (tile invariant, relative to the stage base)
  // dy: row P = ks*32 + 16*e + 4g + (t>>2), group cg = wco*MI + i  ->  (cg*TP + P)*32 + (t&3)*8
  const int aoff = ((wco * MI) * TP + 4 * g + (t >> 2)) * 32 + (t & 3) * 8;
  // x: halo row of pixel P (+ tap offset), group cg = wci*NJ + j -> (cg*nrx + row)*32 + (t&3)*8
  int hb[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int P = ks * 32 + 16 * e + 4 * g + (t >> 2);
      const bool v = P < TPv;
      const int ty = v ? (int)(((unsigned)P * p.mTW) >> 20) : 0;
      const int tx = v ? P - ty * p.TW : 0;
      hb[ks][e] = ((wci * NJ) * nrx + ty * p.is * p.haloW + tx * p.is) * 32 + (t & 3) * 8 + TP * BCO * 2;
    }
  int toff32[NT];
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) toff32[tap] = p.toff[tap] * 32;
  const int jstride = nrx * 32;

  f32x4 acc[NT][MI][NJ];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tpi = p.tilesY * p.tilesX;
  const int tbeg = s * p.tps;
  const int tend = min(p.ntiles, tbeg + p.tps);
  WgBias<NW * 64, BCO> bias;
  const bool do_bias = p.bpart != nullptr && cib == 0;
  bias.init();

  const char* const zero = (const char*)g_mi_zero_page;
  const int half8 = (lane & 1) * 8;   // this lane's 16-byte half of the 32-byte group (in elements)
  auto issue = [&](int tile, int st) {
    const int img = tile / tpi;
    const int rem = tile - img * tpi;
    const int tyq = rem / p.tilesX;
    const int ty0 = tyq * p.TH, tx0 = (rem - tyq * p.tilesX) * p.TW;
    const unsigned sbase = lds0 + st * p.stage;
    const char* const dyb = (const char*)(p.dy + ((size_t)img * p.outH * p.outW) * (size_t)p.lddy + co0);
#pragma unroll
    for (int i = 0; i < QW_DY; ++i) {
      const int u = wave + NW * i;
      const int pb = u % PB, G = u / PB;
      const int row = pb * 32 + (lane >> 1);
      const int ty = (int)(((unsigned)row * p.mTW) >> 20);
      const int tx = row - ty * p.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      const bool v = (row < TPv) & (oy < p.outH) & (ox < p.outW);
      const unsigned off = (unsigned)(((oy * p.outW + ox) * p.lddy + G * 16 + half8) * 2);
      glds16(v ? dyb + off : zero, sbase + (G * TP + pb * 32) * 32);
    }
    const int iy0 = ty0 * p.is + p.dymin, ix0 = tx0 * p.is + p.dxmin;
    const unsigned xbase = sbase + TP * BCO * 2;
    const char* const xb = (const char*)(p.x + ((size_t)img * p.H * p.W) * (size_t)p.ldx + ci0);
    for (int rb = wave; rb < nrb; rb += NW) {
      const int r0 = min(rb * 32, nrx - 32);   // the last unit re-covers rows of its predecessor (same bytes)
      const int row = r0 + (lane >> 1);
      const int hy = (int)(((unsigned)row * p.mHW) >> 20);
      const int hx = row - hy * p.haloW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool v = (row < p.npixh) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const char* src = xb + (unsigned)(((iy * p.W + ix) * p.ldx + half8) * 2);
#pragma unroll
      for (int G = 0; G < GX; ++G) glds16(v ? src + G * 32 : zero, xbase + (G * nrx + r0) * 32);
    }
  };

  const int NS = p.ns;
  const int nload = QW_DY + ((nrb - wave + NW - 1) / NW) * GX;
  for (int j = 0; j < NS - 1; ++j)
    if (tbeg + j < tend) issue(tbeg + j, j);
  int it = 0, cur = 0;
  for (int tile = tbeg; tile < tend; ++tile, ++it) {
    const int ahead = min(NS - 2, tend - 1 - tile);
    wait_vmcnt(ahead * nload);
    __builtin_amdgcn_s_barrier();
    {
      const int nxt = tile + NS - 1;
      int st = cur - 1;
      if (st < 0) st += NS;
      if (nxt < tend) issue(nxt, st);
    }
    if (do_bias) {
      const int img = tile / tpi, rem = tile - img * tpi, tyq = rem / p.tilesX;
      bias.template tile<TP>(p, img, tyq * p.TH, (rem - tyq * p.tilesX) * p.TW, co0, TPv);
    }
    const char* const sB = smem + cur * p.stage;
    if (++cur == NS) cur = 0;
    const char* const aB = sB + aoff;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 a[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[i] = tr_read2(aB + (i * TP + ks * 32) * 32, aB + (i * TP + ks * 32 + 16) * 32);
      const char* const b0 = sB + hb[ks][0];
      const char* const b1 = sB + hb[ks][1];
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        bf16x8 b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          b[j] = tr_read2(b0 + toff32[tap] + j * jstride, b1 + toff32[tap] + j * jstride);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[tap][i][j], 0, 0, 0);
      }
    }
  }
  f32x4* out = (f32x4*)p.part + (size_t)s * (size_t)p.V +
               ((size_t)((cob * p.nci + cib) * NW + wave) * (NT * MI * NJ)) * 64 + lane;
  if (FIX) {                         // fix-up form (its own instantiation): write-through slab, then this block's share of the tile's sum
    const __amdgpu_buffer_rsrc_t rs = wg_slab_rsrc(p.part);
    const unsigned ob = (unsigned)(((size_t)s * (size_t)p.V + ((size_t)((cob * p.nci + cib) * NW + wave) * (NT * MI * NJ)) * 64 + lane) * 16);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) st_agent16(rs, ob, (unsigned)(((tap * MI + i) * NJ + j) * 1024), acc[tap][i][j]);
    wg_fixup<NT, MI, NJ, WCO, WCI>(p, cnt, s, cob * p.nci + cib, smem);
    return;
  }
#pragma unroll
  for (int tap = 0; tap < NT; ++tap)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) out[((tap * MI + i) * NJ + j) * 64] = acc[tap][i][j];
  if (p.bpart != nullptr) {          // (block-uniform; the tile buffers are dead: the partial rows fold through them)
    if (do_bias) bias.finish(p, (float*)smem, s, co0);
  }
}

template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
__global__ __launch_bounds__(WCO* WCI * 64, 2) void wgrad3_kernel(const Wg2K p) {
  wgrad3_body<NT, MI, NJ, WCO, WCI, TP>(p, blockIdx.x);
}
template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
__global__ __launch_bounds__(WCO* WCI * 64, 2) void wgrad3_group_kernel(const Wg2K* __restrict__ jobs,
                                                                         const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && starts[j + 1] <= b) ++j;
  const Wg2K p = jobs[j];
  wgrad3_body<NT, MI, NJ, WCO, WCI, TP>(p, b - starts[j]);
}
// the same grid with the split-K reduction inside (wg_fixup): every job of the group carries fix = 1
// (MI_WG_MULTI=2, see wgrad2_multi_kernel: the 3x3 layers' three 256-thread tile configurations in one grid; pad_ = 16 MI + WCO)
template <int TP>
__global__ __launch_bounds__(256, 2) void wgrad3_multi_kernel(const Wg2K* __restrict__ jobs, const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && starts[j + 1] <= b) ++j;
  const Wg2K p = jobs[j];
  const int bid = b - starts[j];
  switch (p.pad_) {
    case 0x41: wgrad3_body<9, 4, 1, 1, 4, TP>(p, bid); break;
    case 0x22: wgrad3_body<9, 2, 1, 2, 2, TP>(p, bid); break;
    default: wgrad3_body<9, 1, 1, 2, 2, TP>(p, bid); break;
  }
}
template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
__global__ __launch_bounds__(WCO* WCI * 64, 2) void wgrad3_group_fix_kernel(const Wg2K* __restrict__ jobs,
                                                                             const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && starts[j + 1] <= b) ++j;
  const Wg2K p = jobs[j];
  wgrad3_body<NT, MI, NJ, WCO, WCI, TP, true>(p, b - starts[j], (unsigned*)((char*)const_cast<Wg2K*>(jobs) + p.cnt_rel));
}

struct Wg2R {
  const f32x4* part;
  float* g;
  long long V;
  int nsplit, NT, MI, NJ, WCO, WCI, nco, nci, Cout, Cin, accumulate;
  const float* row_scale;   // NULL or [Cout]
  const float* bpart;       // NULL or [nsplit][bld]: split partials of the bias gradient
  float* gbias;             // [Cout]
  int bld, main_blocks;     // blocks >= main_blocks add up the bias partials (one thread per channel, fixed split order)
};
__device__ __forceinline__ void wgrad2_reduce_bias(const Wg2R& p, const long long blk) {
  const int c = (int)(blk - p.main_blocks) * 256 + (int)threadIdx.x;
  if (c >= p.Cout) return;
  float a = 0.f;
  for (int k = 0; k < p.nsplit; ++k) a += p.bpart[(size_t)k * p.bld + c];
  p.gbias[c] = a;
}

// SL "split lanes": the nsplit partials of one output float4 are summed by SL threads (wave w = splits w, w + SL, ...,
// 4 loads in flight each), combined through LDS in a fixed order.  With one thread per output (SL = 1) a layer with 30-60
// splits is a chain of 8-15 dependent round trips per thread and the launch was latency-bound: 183 us for 219 MB
// (profiles/r02a); four lanes per output and a 4x larger grid cut the chain to a quarter.
template <int SL>
__device__ __forceinline__ void wgrad2_reduce_body(const Wg2R& p, const long long blk) {
  constexpr int OUTS = 256 / SL;                 // output float4s per block
  __shared__ f32x4 red[SL > 1 ? SL - 1 : 1][OUTS];
  const int o = threadIdx.x % OUTS, sl = threadIdx.x / OUTS;
  const long long v = blk * OUTS + o;
  const bool valid = v < p.V;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (valid) {
    const f32x4* src = p.part + v;
    int k = sl;
    for (; k + 3 * SL < p.nsplit; k += 4 * SL) {
      const f32x4 a = src[(size_t)k * p.V], b = src[(size_t)(k + SL) * p.V], c = src[(size_t)(k + 2 * SL) * p.V],
                  d = src[(size_t)(k + 3 * SL) * p.V];
      s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; k < p.nsplit; k += SL) s0 += src[(size_t)k * p.V];
  }
  f32x4 sum = (s0 + s1) + (s2 + s3);
  if (SL > 1) {
    if (sl > 0) red[sl - 1][o] = sum;
    __syncthreads();
    if (sl > 0) return;
#pragma unroll
    for (int q = 0; q < SL - 1; ++q) sum += red[q][o];
  }
  if (!valid) return;
  wg_scatter(p.g, p.row_scale, p.accumulate, p.NT, p.MI, p.NJ, p.WCO, p.WCI, p.nci, p.Cout, p.Cin, v, sum);
}

// 3x3 layers: one block = one 16(cout) x 16(cin) fragment tile, wave w = tap w.  The split sums go through an LDS
// transpose so that the OIHW rows leave as contiguous 576-byte runs (16 cin x 9 taps) instead of 4-byte stores 36 bytes
// apart (measured round 1: the scattered form wrote 175 MB for a 36 MB gradient), with the same read parallelism (one
// thread per partial float4 per tap - summing all taps in one thread was 3x slower).
__device__ __forceinline__ void wgrad2_reduce9_body(const Wg2R& p, const int blk) {
  __shared__ float tile[16 * 16 * 9 + 16];
  const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;
  int r = blk;
  const int j = r % p.NJ; r /= p.NJ;
  const int i = r % p.MI; r /= p.MI;
  const int NW = p.WCO * p.WCI;
  const int wave = r % NW; r /= NW;     // r = cob * nci + cib
  const int cib = r % p.nci, cob = r / p.nci;
  const long long v = ((((long long)(r * NW + wave) * 9 + tap) * p.MI + i) * p.NJ + j) * 64 + lane;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  const f32x4* src = p.part + v;
  int k = 0;
  for (; k + 4 <= p.nsplit; k += 4) {
    const f32x4 a = src[(size_t)k * p.V], b = src[(size_t)(k + 1) * p.V], c = src[(size_t)(k + 2) * p.V],
                d = src[(size_t)(k + 3) * p.V];
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; k < p.nsplit; ++k) s0 += src[(size_t)k * p.V];
  const f32x4 sum = (s0 + s1) + (s2 + s3);
  // fragment element (lane, e): cout 4*(lane>>4)+e, cin lane&15  ->  tile[cout][cin][tap]
#pragma unroll
  for (int e = 0; e < 4; ++e) tile[((4 * (lane >> 4) + e) * 16 + (lane & 15)) * 9 + tap] = sum[e];
  __syncthreads();
  const int wco = wave / p.WCI, wci = wave % p.WCI;
  const int ci0 = cib * (16 * p.NJ * p.WCI) + (wci * p.NJ + j) * 16;
  const int co0 = cob * (16 * p.MI * p.WCO) + (wco * p.MI + i) * 16;
  if (ci0 >= p.Cin) return;
  const int row = tid / 36, c4 = (tid % 36) * 4;            // 16 rows x 36 float4
  const int co = co0 + row;
  const int nval = min(16, p.Cin - ci0) * 9;                 // valid floats of a row
  if (co >= p.Cout || c4 >= nval) return;
  float* dst = p.g + ((size_t)co * p.Cin + ci0) * 9 + c4;
  const float* t = tile + row * 144 + c4;
  const float rs = p.row_scale ? p.row_scale[co] : 1.f;     // (x * 1.f is exact: one code path)
  if (c4 + 4 <= nval && ((uintptr_t)dst & 15) == 0) {
    f32x4 o = {t[0] * rs, t[1] * rs, t[2] * rs, t[3] * rs};
    if (p.accumulate) o += *(const f32x4*)dst;
    *(f32x4*)dst = o;
  } else {
    for (int e = 0; e < 4 && c4 + e < nval; ++e) dst[e] = p.accumulate ? dst[e] + t[e] * rs : t[e] * rs;
  }
}
__global__ __launch_bounds__(576) void wgrad2_reduce9_kernel(const Wg2R p) { wgrad2_reduce9_body(p, blockIdx.x); }
__global__ __launch_bounds__(576) void wgrad2_reduce9_group_kernel(const Wg2R* __restrict__ jobs,
                                                                   const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;  // largest j with starts[j] <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const Wg2R p = jobs[lo];
  wgrad2_reduce9_body(p, b - starts[lo]);
}

template <int SL>
__global__ __launch_bounds__(256) void wgrad2_reduce_kernel(const Wg2R p) {
  if (p.bpart != nullptr && (int)blockIdx.x >= p.main_blocks) {
    wgrad2_reduce_bias(p, blockIdx.x);
    return;
  }
  wgrad2_reduce_body<SL>(p, blockIdx.x);
}
template <int SL>
__global__ __launch_bounds__(256) void wgrad2_reduce_group_kernel(const Wg2R* __restrict__ jobs,
                                                                  const int* __restrict__ starts, int njobs) {
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;  // largest j with starts[j] <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const Wg2R p = jobs[lo];
  if (p.bpart != nullptr && b - starts[lo] >= p.main_blocks) {   // (block-uniform)
    wgrad2_reduce_bias(p, b - starts[lo]);
    return;
  }
  wgrad2_reduce_body<SL>(p, b - starts[lo]);
}

// ---------------------------------------------------------------- host side
#define MI_WG_ALL \
  MI_WG(9, 4, 1, 1, 4, 128) MI_WG(9, 4, 1, 1, 4, 64) \
  MI_WG(9, 2, 1, 2, 2, 128) MI_WG(9, 2, 1, 2, 2, 64) \
  MI_WG(9, 1, 1, 2, 2, 128) MI_WG(9, 1, 1, 2, 2, 64) \
  MI_WG(9, 1, 1, 2, 1, 128) MI_WG(9, 1, 1, 2, 1, 64) \
  MI_WG(16, 1, 1, 2, 1, 128) MI_WG(16, 1, 1, 2, 1, 64) \
  MI_WG(1, 1, 1, 2, 2, 128) MI_WG(1, 1, 2, 2, 2, 128) MI_WG(1, 1, 4, 2, 2, 128) \
  MI_WG(1, 2, 1, 2, 2, 128) MI_WG(1, 2, 2, 2, 2, 128) MI_WG(1, 2, 4, 2, 2, 128) \
  MI_WG(1, 4, 1, 2, 2, 128) MI_WG(1, 4, 2, 2, 2, 128) MI_WG(1, 4, 4, 2, 2, 128) \
  MI_WG(1, 1, 1, 2, 2, 64) MI_WG(1, 1, 2, 2, 2, 64) MI_WG(1, 1, 4, 2, 2, 64) \
  MI_WG(1, 2, 1, 2, 2, 64) MI_WG(1, 2, 2, 2, 2, 64) MI_WG(1, 2, 4, 2, 2, 64) \
  MI_WG(1, 4, 1, 2, 2, 64) MI_WG(1, 4, 2, 2, 2, 64) MI_WG(1, 4, 4, 2, 2, 64)

struct Wg2Cfg {
  int NT, MI, NJ, WCO, WCI, TP;
};

// tuning switches (read once; the defaults are the measured best, the environment overrides are for A/B runs)
static int wg_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}
static int wg_xmap() { return wg_env("MI_WG_XMAP", 1); }      // (read per plan: tests and A/B runs build both orders in one process)
static int wg_units() { static const int v = wg_env("MI_WG_UNITS", 0); return v; }
static int wg_red9() { static const int v = wg_env("MI_WG_RED9", 0); return v; }
// threads per output float4 of the split-K reduction (1 or 4; see wgrad2_reduce_body)
static int wg_redsl() {
  static const int v = [] { const int e = wg_env("MI_WG_REDSL", 4); return (e == 1 || e == 4 || e == 8 || e == 16) ? e : 4; }();
  return v;
}
// at least this much dynamic LDS per block of a grouped launch (e.g. 84000: one block per CU, the rest of the CU's LDS
// stays free for kernels of another stream)
// 1: v3 LDS layout (column-major by channel group, see wgrad3_body) for the 3x3 configurations, 2: for all, 0: v2
static int wg_v3() { static const int v = wg_env("MI_WG_V3", 1); return v; }
static bool wg_use_v3(int NT) { return wg_v3() == 2 || (wg_v3() == 1 && NT == 9); }
// 1: grouped launches sum their split-K partials themselves (wg_fixup) instead of leaving them to the reduce grid
static int wg_fixup_on() { return wg_env("MI_WG_FIXUP", 0); }   // (read per plan: a test builds both forms in one process)
static int wg_min_lds() { static const int v = wg_env("MI_WG_MIN_LDS", 0); return v; }

static void wg_choose_tile(int TP, int gridH, int gridW, int* TH, int* TW) {
  const int cands[] = {gridW, 64, 32, 16, 8, 4};
  long best = -1;
  int bh = 8, bw = 16;
  for (int c : cands) {
    if (c <= 0 || c > TP) continue;
    int tw = c, th = TP / tw;
    if (th > gridH) th = gridH;
    if (th < 1) th = 1;
    long tiles = (long)mi_cdiv(gridH, th) * mi_cdiv(gridW, tw);
    long halo = (long)(th + 2) * (tw + 2);
    long score = tiles * 100000 + halo;
    // a 3x3 halo of more than 184 rows no longer fits two 2-stage blocks per CU (64-channel rows): worth ~2 tiles
    if (halo > 184) score += 200000;
    if (best < 0 || score < best) { best = score; bh = th; bw = tw; }
  }
  *TH = bh; *TW = bw;
}

static int wg_pick_cfg(const mi_wgrad_desc* d, Wg2Cfg* c, bool grouped = false) {
  const int ci = d->CinPad, co = d->CoutPad;
  if (d->ntaps == 16) {   // the 7x7 stride-2 ResNet stem as a 4x4 conv over the space-to-depth image (Cin 12 -> 16)
    c->NT = 16; c->NJ = 1; c->MI = 1; c->WCO = 2; c->WCI = 1; c->TP = 128;
  } else if (d->ntaps == 9) {
    c->NT = 9; c->NJ = 1;
    if (ci % 64 == 0 && co % 64 == 0 && d->stride == 1) { c->MI = 4; c->WCO = 1; c->WCI = 4; c->TP = 128; }
    else if (ci % 32 == 0 && co % 64 == 0) { c->MI = 2; c->WCO = 2; c->WCI = 2; c->TP = d->stride == 1 ? 128 : 64; }
    else if (ci % 32 == 0) { c->MI = 1; c->WCO = 2; c->WCI = 2; c->TP = d->stride == 1 ? 128 : 64; }
    else { c->MI = 1; c->WCO = 2; c->WCI = 1; c->TP = 128; }  // Cin 16 (stem)
  } else {
    c->NT = 1; c->WCO = 2; c->WCI = 2; c->TP = 128;
    c->MI = (co % 128 == 0) ? 4 : (co % 64 == 0) ? 2 : 1;
    c->NJ = (ci % 128 == 0) ? 4 : (ci % 64 == 0) ? 2 : 1;
  }
  if (c->TP == 128) {
    int th, tw;
    wg_choose_tile(128, d->outH, d->outW, &th, &tw);
    const long tiles = (long)d->N * mi_cdiv(d->outH, th) * mi_cdiv(d->outW, tw);
    const int BCO = 16 * c->MI * c->WCO, BCI = 16 * c->NJ * c->WCI;
    const long outt = (long)(d->CoutPad / BCO) * (d->CinPad / BCI);
    if (tiles * outt < 4 * 256 && !grouped) c->TP = 64;  // (a grouped launch is filled by the other layers)
  }
  {  // A/B overrides of the pixel-tile size per tap class (MI_WG_TP9 / MI_WG_TP1 = 64 | 128)
    static const int tp9 = wg_env("MI_WG_TP9", 64), tp1 = wg_env("MI_WG_TP1", 64);
    const int tpe = d->ntaps >= 9 ? tp9 : tp1;
    if (tpe == 64 || tpe == 128) c->TP = tpe;
  }
  if (d->cfg_tp == 64 || d->cfg_tp == 128) c->TP = d->cfg_tp;
  return MI_OK;
}

static int wg_fill(const mi_wgrad_desc* d, Wg2K* k, Wg2Cfg* c, size_t* lds, size_t* ws, bool grouped = false) {
  MI_REQUIRE(d->x && d->dy, "wgrad: null pointer");
  MI_REQUIRE(d->ntaps == 1 || d->ntaps == 9 || d->ntaps == 16, "wgrad: ntaps %d", d->ntaps);
  MI_REQUIRE(d->CoutPad % 32 == 0 && d->CinPad % 16 == 0, "wgrad: pads %d %d", d->CoutPad, d->CinPad);
  MI_REQUIRE(d->ntaps != 1 || d->CinPad % 32 == 0, "wgrad: 1x1 needs CinPad %% 32 (got %d)", d->CinPad);
  MI_REQUIRE(d->Cout > 0 && d->Cout <= d->CoutPad && d->Cin > 0 && d->Cin <= d->CinPad, "wgrad: channels");
  MI_REQUIRE(d->ldx % 8 == 0 && d->ldy % 8 == 0 && ((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->dy % 16) == 0,
             "wgrad: alignment");
  MI_REQUIRE(d->stride == 1 || d->stride == 2, "wgrad: stride");
  wg_pick_cfg(d, c, grouped);
  const int BCO = 16 * c->MI * c->WCO, BCI = 16 * c->NJ * c->WCI;
  MI_REQUIRE(d->CoutPad % BCO == 0 && d->CinPad % BCI == 0, "wgrad: tile %dx%d vs pads %d %d", BCO, BCI, d->CoutPad,
             d->CinPad);
  k->x = (const __bf16*)d->x; k->dy = (const __bf16*)d->dy; k->part = (float*)d->ws;
  k->ldx = d->ldx; k->lddy = d->ldy; k->N = d->N; k->H = d->H; k->W = d->W; k->outH = d->outH; k->outW = d->outW;
  k->is = d->stride;
  int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
  for (int t = 0; t < d->ntaps; ++t) {
    if (d->tap_dy[t] < dymin) dymin = d->tap_dy[t];
    if (d->tap_dy[t] > dymax) dymax = d->tap_dy[t];
    if (d->tap_dx[t] < dxmin) dxmin = d->tap_dx[t];
    if (d->tap_dx[t] > dxmax) dxmax = d->tap_dx[t];
  }
  int TH = d->TH, TW = d->TW;
  if (TH <= 0 || TW <= 0) wg_choose_tile(c->TP, d->outH, d->outW, &TH, &TW);
  MI_REQUIRE(TH * TW <= c->TP && TH >= 1 && TW >= 1, "wgrad: tile %dx%d", TH, TW);
  k->TH = TH; k->TW = TW; k->tilesY = mi_cdiv(d->outH, TH); k->tilesX = mi_cdiv(d->outW, TW);
  k->dymin = dymin; k->dxmin = dxmin;
  const int haloH = (TH - 1) * d->stride + (dymax - dymin) + 1;
  k->haloW = (TW - 1) * d->stride + (dxmax - dxmin) + 1;
  k->npixh = haloH * k->haloW;
  for (int t = 0; t < d->ntaps; ++t) k->toff[t] = (d->tap_dy[t] - dymin) * k->haloW + (d->tap_dx[t] - dxmin);
  const int RPI_X = 64 / (BCI * 2 / 16);
  k->nqx = mi_cdiv(k->npixh, RPI_X);
  MI_REQUIRE((long)k->nqx * RPI_X * k->haloW < (1 << 20) && k->nqx * RPI_X < 4096, "wgrad: tile too large for the row decode");
  k->mTW = ((1u << 20) + TW - 1) / TW;
  k->mHW = ((1u << 20) + k->haloW - 1) / k->haloW;
  k->stage = c->TP * BCO * 2 + k->nqx * 1024;
  if (wg_use_v3(c->NT)) {
    k->nrx = k->npixh < 32 ? 32 : k->npixh;
    k->stage = c->TP * BCO * 2 + k->nrx * BCI * 2;   // [group][row][32 B]
    k->stage = (k->stage + 15) / 16 * 16;
  }
  int ns = d->cfg_ns;
  {  // A/B override of the LDS ring depth per tap class (MI_WG_NS9 / MI_WG_NS1 = 2..4)
    static const int ns9 = wg_env("MI_WG_NS9", 2), ns1 = wg_env("MI_WG_NS1", 2);
    const int nse = d->ntaps >= 9 ? ns9 : ns1;
    if ((ns < 2 || ns > 4) && nse >= 2 && nse <= 4) ns = nse;
  }
  if (ns < 2 || ns > 4) {
    // 3x3: MFMA-heavy, two co-resident blocks overlap each other's barriers -> 2 stages when two blocks fit;
    // 1x1 and oversized stages: pure streams, one block per CU with as many tiles in flight as LDS allows
    ns = (d->ntaps >= 9) ? 2 : (int)((160 * 1024) / k->stage);
    if (ns > 4) ns = 4;
    if (ns < 2) ns = 2;
  }
  while (ns > 2 && (size_t)ns * k->stage > 160 * 1024) --ns;
  k->ns = ns;
  *lds = (size_t)ns * (size_t)k->stage;
  MI_REQUIRE(*lds <= 160 * 1024, "wgrad: LDS %zu too large", *lds);
  k->ntiles = d->N * k->tilesY * k->tilesX;
  k->nco = d->CoutPad / BCO; k->nci = d->CinPad / BCI;
  k->xmap = wg_xmap();
  if (!wg_use_v3(c->NT)) k->nrx = 0;
  int split = d->splitk;
  if (split <= 0) {
    // one resident block per CU, but never fewer than a few pixel tiles per block: each block pays a fixed
    // accumulator-slab write (and the reduce kernel a read) that only a long enough K range amortises.  Single launches:
    // >= 2 tiles (round 4; 4 before): the token-row GEMMs of a transformer have 7-66 pixel tiles in all, with 4 per block a
    // 256 x 256 Linear ran on 64 blocks of a 256-CU chip (DETR-R50 step 238.7 -> 242.7 img/s; 1 tile: 244.3, SparseInst
    // -0.3 %; profiles/r04_wgrad_min_tiles_ab.txt)
    split = 256 / (k->nco * k->nci);
    // in a grouped launch the other layers fill the chip: favour long K ranges (less partial-slab traffic)
    static const int min_tiles = wg_env("MI_WG_MIN_TILES", 2);
    static const int min_tiles_g = wg_env("MI_WG_MIN_TILES_GROUP", 8);
    const int by_tiles = k->ntiles / (grouped ? (min_tiles_g >= 1 ? min_tiles_g : 8) : (min_tiles >= 1 ? min_tiles : 4));
    if (split > by_tiles) split = by_tiles;
    if (split >= 8) split &= ~7;  // multiple of 8: the blocks of one pixel range share an XCD (L2)
    if (split < 1) split = 1;
  }
  if (split > k->ntiles) split = k->ntiles;
  k->tps = mi_cdiv(k->ntiles, split);
  k->nsplit = mi_cdiv(k->ntiles, k->tps);
  k->V = (long long)k->nco * k->nci * (c->WCO * c->WCI) * c->NT * c->MI * c->NJ * 64;
  *ws = (size_t)k->nsplit * (size_t)k->V * 16;
  k->bpart = nullptr; k->bld = 0; k->pad_ = 0;
  k->fix = 0; k->g = d->gw; k->row_scale = d->row_scale; k->Cout = d->Cout; k->Cin = d->Cin; k->accumulate = d->accumulate;
  k->cnt_rel = 0;
  k->ra = k->rb = 0;
  if (k->xmap >= 2) {      // compact (cout, cin) rectangles per XCD for the splits outside the groups of 8 (wg_decode)
    const int s8 = k->nsplit & ~7, r = k->nsplit - s8, np = k->nco * k->nci;
    const long total = (long)r * np;
    if (r > 0 && total % 8 == 0) {
      long a = total / 8, b = np;
      while (b) { const long t_ = a % b; a = b; b = t_; }      // a = gcd(items per XCD, pairs per split)
      int ba = 0, bb = 0;
      for (int ra = 1; ra <= k->nco; ++ra) {
        if (k->nco % ra) continue;
        for (int rb = 1; rb <= k->nci; ++rb) {
          if (k->nci % rb || a % ((long)ra * rb)) continue;
          const long cur = (long)ra * rb, best = (long)ba * bb;
          if (cur > best || (cur == best && abs(ra - rb) < abs(ba - bb))) { ba = ra; bb = rb; }
        }
      }
      if ((long)ba * bb >= 2) { k->ra = ba; k->rb = bb; }
    }
  }
  if (d->gbias) {     // the bias partials behind the slabs (grouped: the group plan points bpart behind the job's slabs)
    k->bld = d->CoutPad;
    k->bpart = (d->ws && !grouped) ? (float*)((char*)d->ws + *ws) : (float*)(uintptr_t)16;   // (planning call without a workspace: non-null marker)
    *ws += (size_t)k->nsplit * (size_t)k->bld * 4;
  }
  return MI_OK;
}

extern "C" int64_t mi_conv2d_wgrad_plan(const mi_wgrad_desc* d) {
  Wg2K k; Wg2Cfg c; size_t lds, ws;
  mi_wgrad_desc t = *d;
  if (!t.x) t.x = (const void*)256;
  if (!t.dy) t.dy = (const void*)256;
  const int rc = wg_fill(&t, &k, &c, &lds, &ws);
  if (rc) return rc;
  return (int64_t)ws;
}

template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
static int wg_launch(const Wg2K& k, size_t lds, hipStream_t s) {
  auto fn = wg_use_v3(NT) ? wgrad3_kernel<NT, MI, NJ, WCO, WCI, TP> : wgrad2_kernel<NT, MI, NJ, WCO, WCI, TP>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)wgrad2_kernel<NT, MI, NJ, WCO, WCI, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad3_kernel<NT, MI, NJ, WCO, WCI, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)(k.nsplit * k.nco * k.nci)), dim3(WCO * WCI * 64), lds, s, k);
  MI_CHECK_LAUNCH("conv_wgrad");
  return MI_OK;
}

extern "C" int mi_conv2d_wgrad(const mi_wgrad_desc* d, mi_stream_t st) {
  Wg2K k; Wg2Cfg c; size_t lds, ws;
  int rc = wg_fill(d, &k, &c, &lds, &ws);
  if (rc) return rc;
  MI_REQUIRE(d->gw && d->ws, "wgrad: null gradient / workspace");
  MI_REQUIRE((uintptr_t)d->ws % 16 == 0 && (size_t)d->ws_bytes >= ws, "wgrad: workspace %lld < %zu bytes",
             (long long)d->ws_bytes, ws);
  // (checked before anything is enqueued.  Note: with gbias the 3x3 layers take the generic reduce kernel, whose
  // summation order differs from wgrad2_reduce9_kernel's: weight gradients with and without the fused bias gradient agree
  // to fp32 rounding, not bit for bit - MI_WGRAD_BIAS_MAXPIX decides per layer)
  MI_REQUIRE(!(d->gbias && d->accumulate), "wgrad: gbias is written, not accumulated");
  hipStream_t s = (hipStream_t)st;
  rc = MI_EINVAL;
#define MI_WG(NTv, MIv, NJv, WCOv, WCIv, TPv)                                                          \
  if (c.NT == NTv && c.MI == MIv && c.NJ == NJv && c.WCO == WCOv && c.WCI == WCIv && c.TP == TPv)     \
    rc = wg_launch<NTv, MIv, NJv, WCOv, WCIv, TPv>(k, lds, s);
  MI_WG_ALL
#undef MI_WG
  if (rc == MI_EINVAL) MI_FAIL(MI_EINVAL, "wgrad: no kernel for cfg NT%d MI%d NJ%d W%dx%d TP%d", c.NT, c.MI, c.NJ, c.WCO, c.WCI, c.TP);
  if (rc) return rc;
  Wg2R r;
  r.part = (const f32x4*)d->ws; r.g = d->gw; r.V = k.V; r.nsplit = k.nsplit;
  r.NT = c.NT; r.MI = c.MI; r.NJ = c.NJ; r.WCO = c.WCO; r.WCI = c.WCI; r.nco = k.nco; r.nci = k.nci;
  r.Cout = d->Cout; r.Cin = d->Cin; r.accumulate = d->accumulate; r.row_scale = d->row_scale;
  r.bpart = nullptr; r.gbias = nullptr; r.bld = 0; r.main_blocks = 0;
  if (c.NT == 9 && wg_red9() && !d->gbias)
    hipLaunchKernelGGL(wgrad2_reduce9_kernel, dim3((unsigned)(k.V / (64 * 9))), dim3(576), 0, s, r);
  else {
    const int outs = 256 / wg_redsl();
    unsigned nblk = (unsigned)((k.V + outs - 1) / outs);
    if (d->gbias) {
      r.bpart = k.bpart; r.gbias = d->gbias; r.bld = k.bld; r.main_blocks = (int)nblk;
      nblk += (unsigned)((d->Cout + 255) / 256);
    }
    const dim3 g(nblk);
    switch (wg_redsl()) {
      case 16: hipLaunchKernelGGL(wgrad2_reduce_kernel<16>, g, dim3(256), 0, s, r); break;
      case 8: hipLaunchKernelGGL(wgrad2_reduce_kernel<8>, g, dim3(256), 0, s, r); break;
      case 4: hipLaunchKernelGGL(wgrad2_reduce_kernel<4>, g, dim3(256), 0, s, r); break;
      default: hipLaunchKernelGGL(wgrad2_reduce_kernel<1>, g, dim3(256), 0, s, r);
    }
  }
  MI_CHECK_LAUNCH("conv_wgrad_reduce");
  return MI_OK;
}

// ---------------------------------------------------------------- grouped launch (all layers of a step)
// Weight gradients are not consumed before the optimizer step, so every layer's wgrad can run at the END of
// backward: with 288 GB of HBM each layer simply keeps its own out-gradient buffer alive, and the ~80 launches of
// a YOLOX step collapse into one grid per tile configuration plus one reduce grid (thousands of blocks, no
// per-layer ramp-up / tail, no per-layer launch latency).
static int wg_cfg_id(const Wg2Cfg& c) { return ((((c.NT * 8 + c.MI) * 8 + c.NJ) * 8 + c.WCO) * 8 + c.WCI) * 256 + c.TP; }
// MI_WG_MULTI: 1 (default) the 1x1 layers of a grouped plan share ONE grid whatever their tile widths (same box: 5.292 -> 5.266
// ms per YOLOX-s step); 2: the 3x3 layers' three 256-thread configurations too (measured: 5.29 -> 5.44 ms, a loss - every block
// of that grid takes the largest configuration's LDS and registers, and its ranges are as long as the cost model is wrong);
// 0: one grid per tile configuration (round 5).  Read per plan.
static int g_wg_multi_ok = 1;   // (set per plan: a mixed grid only for plans with many 1x1 layers, see mi_conv2d_wgrad_group_plan)
static int wg_multi() { return g_wg_multi_ok ? wg_env("MI_WG_MULTI", 1) : 0; }
static bool wg_is_multi(const Wg2Cfg& c) {
  if (wg_multi() < 1 || wg_fixup_on()) return false;
  if (c.NT == 1) return c.WCO == 2 && c.WCI == 2 && !wg_use_v3(1);
  return (wg_multi() == 2 || wg_multi() == 4) && c.NT == 9 && wg_use_v3(9) && c.NJ == 1 && c.WCO * c.WCI == 4 &&
         ((c.MI == 4 && c.WCO == 1) || (c.MI <= 2 && c.WCO == 2));
}
// launch group of a configuration: the 1x1 layers share one (MI_WG_MULTI=1), everything else one per tile configuration
static int wg_gid(const Wg2Cfg& c) {
  if (!wg_is_multi(c)) return wg_cfg_id(c);
  Wg2Cfg t = c;
  t.MI = 0; t.NJ = 0;
  if (c.NT == 9) { t.WCO = 0; t.WCI = 0; }
  return wg_cfg_id(t);
}
// relative time of one pixel-tile step of a configuration (profiles/r06_wgrad_xcd_rect.txt (6): 0.96 us for 1 x 1 fragments per
// wave, 1.77 for 4 x 4): blocks of a mixed grid get pixel ranges of about equal duration
static double wg_step_cost(const Wg2Cfg& c) { return 0.9 + 0.055 * c.NT * c.MI * c.NJ; }

extern "C" int mi_conv2d_wgrad_group_plan(const mi_wgrad_desc* descs, int n, void* ws_base, void* table_host,
                                          int64_t table_cap, mi_wgrad_group* meta) {
  MI_REQUIRE(descs && n > 0 && meta, "wgrad_group_plan: args");
  memset(meta, 0, sizeof(*meta));
  std::vector<Wg2K> ks(n);
  std::vector<Wg2Cfg> cs(n);
  std::vector<size_t> ldss(n), wss(n);
  size_t ws_off = 0;
  // the mixed 1x1 grid (MI_WG_MULTI) pays for the whole-step plans (YOLOX-s: 45 1x1 layers in five to nine configurations:
  // -0.5 .. -0.75 % step time); the per-layer / per-block groups of the eager trees hold 4 - 10 jobs of one or two
  // configurations and measured no gain (DETR-R50 346.4 -> 344.4 images/s): those keep one grid per configuration
  {
    int n1 = 0;
    for (int i = 0; i < n; ++i) n1 += descs[i].ntaps == 1;
    g_wg_multi_ok = n1 >= 16 || wg_env("MI_WG_MULTI", 1) >= 3;     // (3: force, for tests of small groups)
  }
  // pass 1: tile configuration + tile counts per layer
  for (int i = 0; i < n; ++i) {
    mi_wgrad_desc t = descs[i];
    MI_REQUIRE(!(t.gbias && wg_fixup_on()), "wgrad_group_plan: job %d carries a bias gradient (gbias): not with MI_WG_FIXUP", i);
    MI_REQUIRE(!(t.gbias && t.accumulate), "wgrad_group_plan: job %d: gbias is written, not accumulated", i);
    if (!t.x) t.x = (const void*)256;
    if (!t.dy) t.dy = (const void*)256;
    int rc = wg_fill(&t, &ks[i], &cs[i], &ldss[i], &wss[i], true);
    if (rc) return rc;
  }
  // pass 2: split-K per GROUP (= one grid per tile configuration).  Every block of a grid runs the same number T of
  // pixel tiles, so the grid's duration is (rounds of resident blocks) x (block time): the measured round-2 failure
  // mode was a grid of 576-788 blocks on 512 resident slots - a second round at 13-54 % occupancy.  MI_WG_UNITS=0
  // (default): T = the smallest K range for which the whole grid is resident at once (slots = CUs x blocks per CU of
  // this configuration's LDS / register footprint): one full round, no tail.  MI_WG_UNITS=U > 0: T = units / U (A/B).
  std::vector<long> Tsel(n, 0);
  for (int i = 0; i < n; ++i) {
    if (Tsel[i]) continue;
    long units = 0;
    size_t lds = 0;
    const bool multi = wg_is_multi(cs[i]);
    // (a mixed grid: T counts steps of the cheapest configuration; job j runs T / cost_j of its own)
    double cmin = 1e30;
    for (int j = 0; j < n; ++j)
      if (wg_gid(cs[j]) == wg_gid(cs[i]) && wg_step_cost(cs[j]) < cmin) cmin = wg_step_cost(cs[j]);
    auto Tj = [&](int j, long T) -> long {
      if (!multi) return T;
      long t = (long)((double)T * cmin / wg_step_cost(cs[j]) + 0.5);
      return t < 4 ? 4 : t;
    };
    for (int j = 0; j < n; ++j)
      if (wg_gid(cs[j]) == wg_gid(cs[i])) {
        units += (long)ks[j].ntiles * ks[j].nco * ks[j].nci;
        if (ldss[j] > lds) lds = ldss[j];
      }
    auto blocks_for = [&](long T0) {
      long b = 0;
      for (int j = 0; j < n; ++j)
        if (wg_gid(cs[j]) == wg_gid(cs[i])) {
          const long T = Tj(j, T0);
          long split = (ks[j].ntiles + T - 1) / T;
          if (split < 1) split = 1;
          const long tps = (ks[j].ntiles + split - 1) / split;
          b += ((ks[j].ntiles + tps - 1) / tps) * ks[j].nco * ks[j].nci;
        }
      return b;
    };
    long T;
    const long U = wg_units();
    if (U > 0) {
      T = (units + U - 1) / U;
      if (T < 4) T = 4;
    } else {
      const int NW = cs[i].WCO * cs[i].WCI;
      long per_cu = (long)(160 * 1024 / (lds ? lds : 1));
      const long by_waves = 8 / NW;      // __launch_bounds__(NW * 64, 2): two waves per SIMD
      if (per_cu > by_waves) per_cu = by_waves;
      if (per_cu < 1) per_cu = 1;
      const long slots = 256 * per_cu;
      T = (units + slots - 1) / slots;
      if (T < 4) T = 4;
      // per-layer rounding: first T whose grid fits.  (Past the longest layer's tile count every job is one split and the
      // grid no longer shrinks: a group with more (cout, cin) pairs than slots - wide models in one mixed grid - stops there
      // instead of counting to 2^20.)
      long tmax = 4;
      for (int j = 0; j < n; ++j)
        if (wg_gid(cs[j]) == wg_gid(cs[i])) {
          const long need = (long)((double)ks[j].ntiles * wg_step_cost(cs[j]) / cmin) + 2;
          if (need > tmax) tmax = need;
        }
      while (blocks_for(T) > slots && T < tmax) ++T;
    }
    for (int j = 0; j < n; ++j)
      if (wg_gid(cs[j]) == wg_gid(cs[i])) Tsel[j] = Tj(j, T);
  }
  for (int i = 0; i < n; ++i) {
    const long T = Tsel[i];
    mi_wgrad_desc t = descs[i];
    if (!t.x) t.x = (const void*)256;
    if (!t.dy) t.dy = (const void*)256;
    if (t.splitk <= 0) {
      t.splitk = (int)((ks[i].ntiles + T - 1) / T);
      if (t.splitk < 1) t.splitk = 1;
    }
    int rc = wg_fill(&t, &ks[i], &cs[i], &ldss[i], &wss[i], true);
    if (rc) return rc;
    ks[i].part = (float*)((char*)ws_base + ws_off);
    if (descs[i].gbias)      // the bias partials [nsplit][bld] sit behind the job's split slabs (wss[i] counts them)
      ks[i].bpart = (float*)((char*)ws_base + ws_off + (size_t)ks[i].nsplit * (size_t)ks[i].V * 16);
    ws_off += (wss[i] + 255) / 256 * 256;
  }
  meta->ws_bytes = (int64_t)ws_off;
  // group by configuration
  std::vector<int> order;
  size_t off = 0;
  char* tab = (char*)table_host;
  auto put = [&](const void* src, size_t bytes) -> long {
    const size_t o = off;
    off += (bytes + 15) / 16 * 16;
    if (tab && (int64_t)off <= table_cap) memcpy(tab + o, src, bytes);
    return (long)o;
  };
  // MI_WG_FIXUP=1: the tile counters of every job come first in the table (uploaded as zeros; they re-arm themselves), 64
  // words per (cout, cin) output tile: [0] arrivals, [32] finished reductions, each on its own 128-byte line; [33] sticky
  // "a wait of this tile timed out" flag (the blocks that gave up wrote NaN; read by the host, never cleared by the device)
  const bool fix = wg_fixup_on() != 0;
  std::vector<long> cnt_off(n, 0);
  if (fix) {
    size_t words = 0;
    for (int i = 0; i < n; ++i) {
      MI_REQUIRE(wss[i] < ((size_t)1 << 31), "wgrad_group_plan: job %d has %zu bytes of split slabs (the fix-up addresses < 2^31)", i, wss[i]);
      cnt_off[i] = (long)(words * 4);
      words += (size_t)ks[i].nco * ks[i].nci * 64;
    }
    std::vector<unsigned> zeros(words, 0u);
    put(zeros.data(), words * 4);
    off = (off + 255) / 256 * 256;
  }
  std::vector<char> done(n, 0);
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    MI_REQUIRE(meta->ngroups < MI_WGRAD_MAX_GROUPS, "wgrad_group_plan: too many tile configurations");
    auto& g = meta->g[meta->ngroups++];
    g.cfg[0] = cs[i].NT; g.cfg[1] = cs[i].MI; g.cfg[2] = cs[i].NJ; g.cfg[3] = cs[i].WCO; g.cfg[4] = cs[i].WCI;
    g.cfg[5] = cs[i].TP;
    if (wg_is_multi(cs[i])) g.cfg[1] = g.cfg[2] = 0;       // (a mixed group: wgrad2_multi_kernel / wgrad3_multi_kernel)
    std::vector<Wg2K> jobs;
    std::vector<int> starts;
    int blocks = 0;
    size_t lds = 0;
    const long job_off = (long)off;      // (what the put() below returns)
    for (int j = i; j < n; ++j)
      if (!done[j] && wg_gid(cs[j]) == wg_gid(cs[i])) {
        done[j] = 1;
        starts.push_back(blocks);
        ks[j].pad_ = cs[j].NT == 9 ? cs[j].MI * 16 + cs[j].WCO : cs[j].MI * 16 + cs[j].NJ;
        if (fix) {
          ks[j].fix = 1;
          ks[j].cnt_rel = (long long)cnt_off[j] - (long long)job_off;
        }
        jobs.push_back(ks[j]);
        blocks += ks[j].nsplit * ks[j].nco * ks[j].nci;
        if (ldss[j] > lds) lds = ldss[j];
      }
    starts.push_back(blocks);
    if (fix) {   // the fix-up folds its four split lanes through LDS: flag + [3][WG_FIX_U][threads / 4] float4
      const size_t need = 16 + (size_t)3 * WG_FIX_U * (cs[i].WCO * cs[i].WCI * 16) * 16;
      if (lds < need) lds = need;
    }
    g.njobs = (int)jobs.size(); g.nblocks = blocks; g.lds_bytes = (int32_t)lds; g.fixup = fix ? 1 : 0;
    g.job_off = put(jobs.data(), jobs.size() * sizeof(Wg2K));
    MI_REQUIRE(g.job_off == job_off, "wgrad_group_plan: table layout");
    g.starts_off = put(starts.data(), starts.size() * sizeof(int));
  }
  // reduce jobs: one grid for the 1x1 layers (256-thread blocks, 4 fragment tiles each) and one for the 3x3 layers
  // (576-thread blocks = one fragment tile x 9 taps, LDS-transposed rows)
  std::vector<Wg2R> rj, rj9;
  std::vector<int> rs, rs9;
  int rblocks = 0, rblocks9 = 0;
  for (int i = 0; i < n && !fix; ++i) {
    Wg2R r;
    r.part = (const f32x4*)ks[i].part; r.g = descs[i].gw; r.V = ks[i].V; r.nsplit = ks[i].nsplit;
    r.NT = cs[i].NT; r.MI = cs[i].MI; r.NJ = cs[i].NJ; r.WCO = cs[i].WCO; r.WCI = cs[i].WCI;
    r.nco = ks[i].nco; r.nci = ks[i].nci; r.Cout = descs[i].Cout; r.Cin = descs[i].Cin;
    r.accumulate = descs[i].accumulate; r.row_scale = descs[i].row_scale;
    r.bpart = nullptr; r.gbias = nullptr; r.bld = 0; r.main_blocks = 0;
    if (descs[i].gbias) {      // bias gradient: extra blocks of the job in the generic reduce grid add up the split partials
      const int outs = 256 / wg_redsl();
      r.bpart = ks[i].bpart; r.gbias = descs[i].gbias; r.bld = ks[i].bld;
      r.main_blocks = (int)((ks[i].V + outs - 1) / outs);
      rj.push_back(r);
      rs.push_back(rblocks);
      rblocks += r.main_blocks + (descs[i].Cout + 255) / 256;
      continue;
    }
    if (cs[i].NT == 9 && wg_red9()) {
      rj9.push_back(r);
      rs9.push_back(rblocks9);
      rblocks9 += (int)(ks[i].V / (64 * 9));
    } else {
      rj.push_back(r);
      rs.push_back(rblocks);
      const int outs = 256 / wg_redsl();
      rblocks += (int)((ks[i].V + outs - 1) / outs);
    }
  }
  rs.push_back(rblocks);
  rs9.push_back(rblocks9);
  meta->nred = (int)rj.size(); meta->red_blocks = rblocks;
  meta->red_off = put(rj.data(), rj.size() * sizeof(Wg2R));
  meta->red_starts_off = put(rs.data(), rs.size() * sizeof(int));
  meta->nred9 = (int)rj9.size(); meta->red9_blocks = rblocks9;
  meta->red9_off = put(rj9.data(), rj9.size() * sizeof(Wg2R));
  meta->red9_starts_off = put(rs9.data(), rs9.size() * sizeof(int));
  meta->table_bytes = (int64_t)off;
  if (tab) MI_REQUIRE((int64_t)off <= table_cap, "wgrad_group_plan: table needs %zu bytes", off);
  return MI_OK;
}

template <int NT, int MI, int NJ, int WCO, int WCI, int TP>
static int wg_group_launch(const Wg2K* jobs, const int* starts, int njobs, int nblocks, size_t lds, hipStream_t s, bool fix) {
  auto fn = fix ? (wg_use_v3(NT) ? wgrad3_group_fix_kernel<NT, MI, NJ, WCO, WCI, TP> : wgrad2_group_fix_kernel<NT, MI, NJ, WCO, WCI, TP>)
                : (wg_use_v3(NT) ? wgrad3_group_kernel<NT, MI, NJ, WCO, WCI, TP> : wgrad2_group_kernel<NT, MI, NJ, WCO, WCI, TP>);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)wgrad2_group_kernel<NT, MI, NJ, WCO, WCI, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad3_group_kernel<NT, MI, NJ, WCO, WCI, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad2_group_fix_kernel<NT, MI, NJ, WCO, WCI, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad3_group_fix_kernel<NT, MI, NJ, WCO, WCI, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if ((size_t)wg_min_lds() > lds && wg_min_lds() <= 160 * 1024) lds = (size_t)wg_min_lds();
  hipLaunchKernelGGL(fn, dim3((unsigned)nblocks), dim3(WCO * WCI * 64), lds, s, jobs, starts, njobs);
  MI_CHECK_LAUNCH("conv_wgrad_group");
  return MI_OK;
}

extern "C" int mi_conv2d_wgrad_group_run(const mi_wgrad_group* meta, const void* table_dev, mi_stream_t st) {
  MI_REQUIRE(meta && table_dev && meta->ngroups > 0, "wgrad_group_run: args");
  hipStream_t s = (hipStream_t)st;
  const char* tab = (const char*)table_dev;
  for (int gi = 0; gi < meta->ngroups; ++gi) {
    const auto& g = meta->g[gi];
    const Wg2K* jobs = (const Wg2K*)(tab + g.job_off);
    const int* starts = (const int*)(tab + g.starts_off);
    int rc = MI_EINVAL;
#define MI_WG(NTv, MIv, NJv, WCOv, WCIv, TPv)                                                                   \
  if (g.cfg[0] == NTv && g.cfg[1] == MIv && g.cfg[2] == NJv && g.cfg[3] == WCOv && g.cfg[4] == WCIv &&          \
      g.cfg[5] == TPv)                                                                                          \
    rc = wg_group_launch<NTv, MIv, NJv, WCOv, WCIv, TPv>(jobs, starts, g.njobs, g.nblocks, (size_t)g.lds_bytes, s, g.fixup == 1);
    MI_WG_ALL
#undef MI_WG
    if (g.cfg[0] == 1 && g.cfg[1] == 0 && g.cfg[2] == 0 && g.fixup != 1) {
      static bool attr_done = false;
      if (!attr_done) {
        hipFuncSetAttribute((const void*)wgrad2_multi_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)wgrad2_multi_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
      }
      if (g.cfg[5] == 64)
        hipLaunchKernelGGL(wgrad2_multi_kernel<64>, dim3((unsigned)g.nblocks), dim3(256), (size_t)g.lds_bytes, s, jobs, starts, g.njobs);
      else
        hipLaunchKernelGGL(wgrad2_multi_kernel<128>, dim3((unsigned)g.nblocks), dim3(256), (size_t)g.lds_bytes, s, jobs, starts, g.njobs);
      MI_CHECK_LAUNCH("conv_wgrad_multi");
      rc = MI_OK;
    }
    if (g.cfg[0] == 9 && g.cfg[1] == 0 && g.cfg[2] == 0 && g.fixup != 1) {
      static bool attr_done = false;
      if (!attr_done) {
        hipFuncSetAttribute((const void*)wgrad3_multi_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)wgrad3_multi_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
      }
      if (g.cfg[5] == 64)
        hipLaunchKernelGGL(wgrad3_multi_kernel<64>, dim3((unsigned)g.nblocks), dim3(256), (size_t)g.lds_bytes, s, jobs, starts, g.njobs);
      else
        hipLaunchKernelGGL(wgrad3_multi_kernel<128>, dim3((unsigned)g.nblocks), dim3(256), (size_t)g.lds_bytes, s, jobs, starts, g.njobs);
      MI_CHECK_LAUNCH("conv_wgrad3_multi");
      rc = MI_OK;
    }
    if (rc == MI_EINVAL) MI_FAIL(MI_EINVAL, "wgrad_group: no kernel for group %d", gi);
    if (rc) return rc;
  }
  if (meta->red9_blocks > 0) {
    hipLaunchKernelGGL(wgrad2_reduce9_group_kernel, dim3((unsigned)meta->red9_blocks), dim3(576), 0, s,
                       (const Wg2R*)(tab + meta->red9_off), (const int*)(tab + meta->red9_starts_off), meta->nred9);
    MI_CHECK_LAUNCH("conv_wgrad_reduce9_group");
  }
  if (meta->red_blocks > 0) {
    const dim3 g((unsigned)meta->red_blocks);
    const Wg2R* jt = (const Wg2R*)(tab + meta->red_off);
    const int* st_ = (const int*)(tab + meta->red_starts_off);
    switch (wg_redsl()) {
      case 16: hipLaunchKernelGGL(wgrad2_reduce_group_kernel<16>, g, dim3(256), 0, s, jt, st_, meta->nred); break;
      case 8: hipLaunchKernelGGL(wgrad2_reduce_group_kernel<8>, g, dim3(256), 0, s, jt, st_, meta->nred); break;
      case 4: hipLaunchKernelGGL(wgrad2_reduce_group_kernel<4>, g, dim3(256), 0, s, jt, st_, meta->nred); break;
      default: hipLaunchKernelGGL(wgrad2_reduce_group_kernel<1>, g, dim3(256), 0, s, jt, st_, meta->nred);
    }
    MI_CHECK_LAUNCH("conv_wgrad_reduce_group");
  }
  return MI_OK;
}
