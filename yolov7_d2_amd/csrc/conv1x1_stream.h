// Streaming 1x1 convolution for gfx950: persistent blocks, weights stationary in registers.
//
// Replaces the 1x1 nn.Conv2d forward / data gradient of BaseConv (yolov7/modeling/backbone/layers/wrappers.py:60-83:
// CSPLayer conv1 / conv2 / conv3, Bottleneck conv1, SPP / lateral / reduce convs, the head stems of
// yolov7/modeling/head/yolox_head.py:60-72) for K <= 512 input channels.  These layers are HBM-bound (~60 FLOP/B,
// SURVEY 8d) and the tile-per-block implicit-GEMM kernel spent its time on everything but memory: every block
// re-loaded the whole weight matrix into LDS (as many bytes as its pixel tile), staged its output through LDS behind
// two barriers and paid 2 x BN fp64 atomics for the BatchNorm statistics of 128 pixels.  Here
//   * a block is persistent: it walks pixel tiles t = b, b + nb, ... of the flat [N*H*W][K] activation;
//   * each wave keeps the A fragments (32 output channels x all K) of ITS 32-channel slice in VGPRs for the whole
//     launch - loaded once, straight from the packed weight image, no LDS;
//   * the x tile goes HBM -> LDS by LDS-DMA into an NBUF-deep ring, NBUF - 1 tiles ahead, behind COUNTED vmcnt waits
//     (one s_barrier per tile); 16-byte chunks are XOR-permuted on the source address so that the ds_read_b128 of the
//     B fragments is conflict-free;
//   * the epilogue never touches LDS: accumulators -> bf16 -> v_permlane32_swap pairs -> one 16-byte store per lane
//     and 16 channels (cdna guide T21), fire and forget;
//   * BatchNorm (sum, sumsq) of the stored (bf16-rounded) values accumulate in registers over ALL tiles of the block
//     and leave as one set of fp64 atomics per block (256-512 blocks instead of thousands);
//   * a slice table gives every 32-channel slice its own weight image / output view / statistics: the two 1x1 convs
//     of a CSP layer that read the same tensor are ONE launch that reads it once.
// MODE 0: plain store; 1: + BatchNorm statistics; 2: y += result (gradient fan-in; the old values are requested a
// tile's compute ahead of their use and waited for with a counted vmcnt); 3: MODE 1, then - behind a grid barrier - the
// BatchNorm(train) + SiLU (+ residual) of the block's OWN output tiles (conv_bn.h): scale / shift come from the finished
// fp64 sums, every lane re-reads the 16-byte pieces it stored itself (same CU, same L2: no cross-XCD coherence needed) and
// writes the activation.  The whole BaseConv forward (layers/wrappers.py:76-83) in one launch.
// MODE 4 (round 6): plain store with an fp32 bias and an optional ReLU in the epilogue (detectron2 Conv2d + FrozenBatchNorm2d
// (+ ReLU) of the ResNet bottlenecks: the folded shift is the bias); MODE 5: MODE 4's bias, then the second tensor `aux` at
// the output pixels - requested and waited for exactly like MODE 2's old values - as the residual under a ReLU
// (MI_CONV_ADDRELU: conv3 + shortcut + ReLU) or as the ReLU mask of a data gradient (MI_CONV_RELUMASK).  Same arithmetic
// and roundings as the tile kernel's epilogues (conv_igemm_kernel.h).  MODE 6: MODE 2's accumulate, then the ReLU mask of the
// sum from a third tensor (MI_CONV_ACCUM | MI_CONV_RELUMASK: the last data gradient of a ResNet bottleneck's input).
// XF 1 (with MODE 1): x is the RAW output of the producing convolution; its BatchNorm(train) + SiLU runs here, on the x tile
// in LDS, by the wave that fetched the piece (conv_bn.h, BnXf); the blocks of cout tile 0 store the activated tile.
#pragma once
#include "common.h"
#include "conv_bn.h"

#define C1_MAX_SLICES 16
#define C1_MAX_BN 2
struct C1Slice {
  const u32x4* w;   // first row of the slice in the packed image: row(k8, co) = k8 * wld + co (16-byte rows)
  __bf16* y;        // first output channel of the slice, pixel 0
  double* stats;    // fp64 accumulators of the slice's first channel, slot 0 ([slot][sld / 2][2]); MODE 1 only
  int wld, ldy, sld, nslots;
  int bnj, c0;      // MODE 3: the slice's convolution (index into C1K::bn) and its first channel inside that convolution
  const float* bias;   // MODE 4 / 5: fp32 bias of the slice's first channel (NULL: none)
  const __bf16* aux;   // MODE 5: second tensor at the output pixels, the slice's first channel (ReLU mask / residual)
  int ldaux, pad2_;
};
struct C1K {
  const __bf16* x;
  int ldx, ntiles, nco, nb;   // nb: blocks per cout tile (grid = nb * nco)
  int xcd_order, dbg;         // dbg & 1: skip the statistics atomics (timing experiments only).  xcd_order 1: block id = cot * nb + b (cout tiles of a pixel tile share an XCD), 0: b * nco + cot
  C1Slice s[C1_MAX_SLICES];
  CBnFwd bn[C1_MAX_BN];       // MODE 3
  const BnXf* xf;             // XF launches: BatchNorm + activation of the input (device record)
  int xfw, pad_;              // xfw: store the activated input (cout tile 0's blocks do it)
  int relu, epi;              // MODE 4: max(., 0) after the bias; MODE 5: 1 = relu(bf16(conv + bias) + aux), 2 = (aux > 0) ? conv : 0
  int npix, ragged;           // ragged 1 (MODE 0 / 2 / 4 / 5): the last pixel tile holds npix - (ntiles - 1) * TPIX < TPIX pixels
};
struct C1Launch {
  int K, WM, PT, NBUF, MODE, grid, lds, XF;
  C1K k;
};

#define C1_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

static __device__ uint4 g_c1_zero_page[4];
static __device__ __attribute__((aligned(256))) unsigned g_c1_bar[MI_BN_BAR_WORDS];   // grid barrier of the MODE 3 launches (one at a time)

// LDS-DMA, 16 bytes per lane: LDS[lds_off + lane * 16 ..) = *(sbase + voff); sbase / lds_off wave-uniform.
// (s_nop 4: an SGPR base that was produced by v_readfirstlane needs 5 wait states before a VMEM instruction reads it;
//  the same pad covers the M0 write)
__device__ __forceinline__ void c1_glds16(const void* sbase, unsigned voff, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase),
               "s"(lds_off)
               : "memory");
}

template <int K, int WM, int WN, int PT, int NBUF, int MODE, int XF = 0>
__global__ __launch_bounds__(WM* WN * 64) void c1s_kernel(const C1K p) {
  static_assert(!XF || MODE == 1, "the input transform comes with the forward (statistics) mode");
  constexpr int NW = WM * WN, TPIX = WN * PT * 32, KC8 = K / 8, KS = K / 16, R = K * 2;
  constexpr int XB = TPIX * R;              // bytes of one x tile
  constexpr int D = XB / (1024 * NW);       // LDS-DMA instructions per wave and tile
  static_assert(D >= 1 && XB % (1024 * NW) == 0, "tile / DMA split");
  constexpr int RPI = KC8 >= 64 ? 1 : 64 / KC8;              // pixel rows per DMA instruction
  constexpr int RBSH = K >= 128 ? 0 : (K == 64 ? 1 : 2);     // log2(pixel rows per 256-byte bank row)
  constexpr int SWM = KC8 - 1 < 15 ? KC8 - 1 : 15;           // swizzle mask (16-byte slots of a bank row)
  constexpr int S = 2 * PT;                                  // 16-byte stores per wave and tile
  constexpr bool AUXM = MODE == 2 || MODE == 5 || MODE == 6;  // a second (MODE 6: and a third) tensor is read at the output pixels
  constexpr bool RAGGED_OK = (MODE == 0 || MODE == 2 || MODE >= 4) && !XF;   // (no statistics over pad pixels)
  constexpr bool BIASM = MODE == 4 || MODE == 5;
  constexpr int L = MODE == 6 ? 4 * PT : (AUXM ? 2 * PT : 0);   // old-value / aux loads per wave and tile
  // VMEM operations newer than tile i's DMA when iteration i waits for it (queue per iteration: OLD, DMA, ST)
  constexpr int W0 = (NBUF - 2) * D + L;
  constexpr int W1 = NBUF > 2 ? (NBUF - 2) * D + (S + L) + L : (S + L);
  constexpr int W2 = NBUF > 3 ? (NBUF - 2) * D + 2 * (S + L) + L : (NBUF - 2) * D + (NBUF - 1) * (S + L);
  constexpr int W3 = (NBUF - 2) * D + (NBUF - 1) * (S + L);
  static_assert(W3 < 64 && W2 < 64, "vmcnt field");
  // XF, writer blocks: D stores of the activated tile per iteration, issued between the wait and the tile's barrier (queue
  // per iteration: SIDE, DMA, ST).  Newer than tile i's DMA at iteration i: i <= NBUF - 2 (DMA issued in the preamble):
  // (NBUF - 2 - i) D + i (2 D + S); later: (NBUF - 1) S + (NBUF - 2) 2 D
  constexpr int X1 = NBUF > 2 ? (NBUF - 3) * D + (2 * D + S) : S;
  constexpr int X2 = NBUF > 3 ? (NBUF - 4) * D + 2 * (2 * D + S) : (NBUF - 1) * S + (NBUF - 2) * 2 * D;
  constexpr int X3 = (NBUF - 1) * S + (NBUF - 2) * 2 * D;
  static_assert(!XF || (X3 < 64 && X2 < 64 && X1 < 64), "vmcnt field");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int nb = p.nb;
  const int cot = p.xcd_order ? (int)blockIdx.x / nb : (int)blockIdx.x % p.nco;
  const int b = p.xcd_order ? (int)blockIdx.x % nb : (int)blockIdx.x / p.nco;
  const int nt = (p.ntiles - b + nb - 1) / nb;   // tiles of this block (>= 1: nb <= ntiles)
  const C1Slice sl = p.s[cot * WM + wm];
  unsigned gen0 = 0;   // thread 0 only: the barrier generation this launch starts in
  if constexpr (MODE == 3) {
    const int nblk = (int)gridDim.x;
    if (tid == 0)
      gen0 = __hip_atomic_load(bn_bar_gen(g_c1_bar, (int)blockIdx.x % (nblk < BN_BAR_G ? nblk : BN_BAR_G)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  constexpr bool STATS = MODE == 1 || MODE == 3;

  // ---- this wave's weights: A fragments of 32 output channels x K, resident for the whole launch
  bf16x8 a[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = __builtin_bit_cast(bf16x8, sl.w[(size_t)(ks * 2 + h) * sl.wld + l31]);

  // ---- per-lane source offsets of this wave's D DMA instructions (relative to the tile's first pixel)
  const int ldxb = p.ldx * 2;
  unsigned doff[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int q = wave * D + d;
    const int row = q * RPI + (KC8 >= 64 ? 0 : lane / KC8);
    const int c = KC8 >= 64 ? lane : lane % KC8;
    doff[d] = (unsigned)(row * ldxb + ((c ^ ((row >> RBSH) & SWM)) << 4));
  }
  const size_t xtile = (size_t)TPIX * (size_t)ldxb;
  auto issue_x = [&](int i, int buf) {
    // tile i of this block; past the last tile the same number of DMAs is issued from a 16-byte zero page (one cache
    // line for the whole wave), so that the counted waits keep their meaning
    const bool live = i < nt;
    const char* xt = live ? (const char*)p.x + (size_t)(b + i * nb) * xtile : (const char*)g_c1_zero_page;
    if (RAGGED_OK && live && p.ragged && b + i * nb == p.ntiles - 1) {
      // the partial last tile: rows past the end fetch the tile's row 0 instead (their results are never stored, and a
      // pixel's output depends on its own row only)
      const int lim = p.npix - (p.ntiles - 1) * TPIX;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int q = wave * D + d;
        const int row = q * RPI + (KC8 >= 64 ? 0 : lane / KC8);
        const int c = KC8 >= 64 ? lane : lane % KC8;
        c1_glds16(xt, row < lim ? doff[d] : (unsigned)(c << 4), lds0 + buf * XB + (wave * D + d) * 1024);
      }
      return;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) c1_glds16(xt, live ? doff[d] : 0u, lds0 + buf * XB + (wave * D + d) * 1024);
  };

  // ---- B fragment addresses: pixel row r, 16-byte slot (ks * 2 + h) ^ swz(r) = (ks * 2) ^ (h ^ swz(r))
  unsigned brow[PT], bsw[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int r = (wn * PT + j) * 32 + l31;
    brow[j] = (unsigned)(r * R);
    bsw[j] = (unsigned)((h ^ ((r >> RBSH) & SWM)) << 4);
  }
  // ---- output: lane (l31, h) owns pixel r, bytes [h * 16, h * 16 + 16) of each 32-byte channel group pair
  const int ldyb = sl.ldy * 2;
  unsigned yoff[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) yoff[j] = (unsigned)(((wn * PT + j) * 32 + l31) * ldyb + h * 16);
  const size_t ytile = (size_t)TPIX * (size_t)ldyb;
  // MODE 5: the second tensor, addressed like y with its own pixel stride
  const int ldab5 = (MODE == 5 || MODE == 6) ? sl.ldaux * 2 : 0;
  unsigned aoff5[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) aoff5[j] = (unsigned)(((wn * PT + j) * 32 + l31) * ldab5 + h * 16);
  const size_t atile5 = (size_t)TPIX * (size_t)ldab5;
  float bs[BIASM ? 16 : 1];      // bias of this lane's channels 8 q + 4 h + e (accumulator order)
  if constexpr (BIASM) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) bs[4 * q + e] = sl.bias ? sl.bias[8 * q + 4 * h + e] : 0.f;
  }
  const bool relu4 = BIASM && p.relu != 0;

  float s1[16], s2[16];
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
  }

#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) issue_x(i, i);

  // XF: (scale, shift) tables of the K input channels behind the ring; block 0 records the layer's statistics
  float* const s_sc = (float*)(smem + NBUF * XB);
  float* const s_sh = s_sc + K;
  int xf_act = 0, xf_ldab = 0;
  char* xf_a = nullptr;
  if constexpr (XF) {
    bnx_tables(p.xf, s_sc, s_sh, NW * 64, blockIdx.x == 0);
    xf_act = p.xf->bn.act;
    xf_ldab = p.xf->bn.lda * 2;
    xf_a = (p.xfw && cot == 0) ? (char*)p.xf->bn.a : nullptr;
    __syncthreads();
  }

  int buf = 0, nbuf = NBUF - 1;   // ring slots of tile i / tile i + NBUF - 1
  for (int i = 0; i < nt; ++i) {
    char* const yt = (char*)sl.y + (size_t)(b + i * nb) * ytile;
    // pixels of this tile that exist (TPIX except in a ragged launch's last tile); lanes past it read pixel 0's old / aux
    // values (in bounds, discarded) and store nothing
    int lim = TPIX;
    if constexpr (RAGGED_OK) {
      if (p.ragged && b + i * nb == p.ntiles - 1) lim = p.npix - (p.ntiles - 1) * TPIX;
    }
    u32x4 old[2 * PT];
    u32x4 aux6[MODE == 6 ? 2 * PT : 1];
    if constexpr (MODE == 2 || MODE == 6) {
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const unsigned yo = ((wn * PT + j) * 32 + l31) < lim ? yoff[j] : (unsigned)(h * 16);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(old[j * 2]) : "v"(yo), "s"(yt) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(old[j * 2 + 1]) : "v"(yo), "s"(yt) : "memory");
      }
    }
    if constexpr (MODE == 5) {
      const char* const at5 = (const char*)sl.aux + (size_t)(b + i * nb) * atile5;
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const unsigned ao = ((wn * PT + j) * 32 + l31) < lim ? aoff5[j] : (unsigned)(h * 16);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(old[j * 2]) : "v"(ao), "s"(at5) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(old[j * 2 + 1]) : "v"(ao), "s"(at5) : "memory");
      }
    }
    if constexpr (MODE == 6) {   // the ReLU mask of the accumulated sum, beside the old values
      const char* const at5 = (const char*)sl.aux + (size_t)(b + i * nb) * atile5;
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const unsigned ao = ((wn * PT + j) * 32 + l31) < lim ? aoff5[j] : (unsigned)(h * 16);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(aux6[j * 2]) : "v"(ao), "s"(at5) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(aux6[j * 2 + 1]) : "v"(ao), "s"(at5) : "memory");
      }
    }
    if (XF && xf_a) {
      if (i == 0) C1_VMCNT(W0);
      else if (i == 1) C1_VMCNT(X1);
      else if (i == 2) C1_VMCNT(X2);
      else C1_VMCNT(X3);
    } else {
      if (i == 0) C1_VMCNT(W0);
      else if (i == 1) C1_VMCNT(W1);
      else if (i == 2) C1_VMCNT(W2);
      else C1_VMCNT(W3);
    }
    if constexpr (XF) {
      // BatchNorm + SiLU of the pieces this wave fetched itself, in place: lane (row, slot c) holds channel group
      // c ^ swz(row) of pixel row (the source-side permutation of doff[])
      char* const Xw = smem + buf * XB;
      char* const at = xf_a + (size_t)(b + i * nb) * ((size_t)TPIX * (size_t)xf_ldab);
#pragma unroll 2
      for (int d = 0; d < D; ++d) {
        const int q = wave * D + d;
        const int row = q * RPI + (KC8 >= 64 ? 0 : lane / KC8);
        const int c = KC8 >= 64 ? lane : lane % KC8;
        const int cg = c ^ ((row >> RBSH) & SWM);
        const u32x4 o = bnx_apply_lds(Xw + q * 1024 + lane * 16, s_sc, s_sh, cg, xf_act, true);
        if (xf_a) *(u32x4*)(at + (unsigned)(row * xf_ldab + cg * 16)) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // tile i landed (every wave's share); slot nbuf is no longer read by anyone
    issue_x(i + NBUF - 1, nbuf);

    f32x16 acc[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const char* Xs = smem + buf * XB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 bf[PT];
#pragma unroll
      for (int j = 0; j < PT; ++j) bf[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(Xs + brow[j] + ((unsigned)(ks * 32) ^ bsw[j])));
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bf[j], acc[j], 0, 0, 0);
    }

    if constexpr (AUXM) {
      // the old values were requested before this tile's DMA: D newer operations may stay in flight
      if constexpr (MODE == 6) {
        if constexpr (PT == 1)
          asm volatile("s_waitcnt vmcnt(%4)" : "+v"(old[0]), "+v"(old[1]), "+v"(aux6[0]), "+v"(aux6[1]) : "n"(D) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%8)"
                       : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(aux6[0]), "+v"(aux6[1]), "+v"(aux6[2]), "+v"(aux6[3])
                       : "n"(D) : "memory");
      } else if constexpr (PT == 1)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(old[0]), "+v"(old[1]) : "n"(D) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]) : "n"(D) : "memory");
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      unsigned pk[8];   // [q][2 dwords]: channels 8q + 4h + (0..3) of this lane's pixel
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (BIASM) {
            float v = acc[j][4 * q + e] + bs[4 * q + e];       // conv_igemm_kernel.h stage(): + bias, ReLU, then the bf16 rounding
            if (relu4) v = fmaxf(v, 0.f);
            o[e] = (__bf16)v;
          } else {
            o[e] = (__bf16)acc[j][4 * q + e];
          }
        }
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = (float)o[e];
            s1[4 * q + e] += f;
            s2[4 * q + e] = __builtin_fmaf(f, f, s2[4 * q + e]);
          }
        }
        const u32x2 u = __builtin_bit_cast(u32x2, o);
        pk[2 * q] = u[0];
        pk[2 * q + 1] = u[1];
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        // lanes 0-31 end up with channels 16 pr + 0..7 (own 4 | upper half's 4), lanes 32-63 with 16 pr + 8..15
        const int q0 = 2 * pr, q1 = 2 * pr + 1;
        const auto w0 = __builtin_amdgcn_permlane32_swap(pk[2 * q0], pk[2 * q1], false, false);
        const auto w1 = __builtin_amdgcn_permlane32_swap(pk[2 * q0 + 1], pk[2 * q1 + 1], false, false);
        u32x4 v = {w0[0], w1[0], w0[1], w1[1]};
        if constexpr (MODE == 5) {
          const bf16x8 nv = __builtin_bit_cast(bf16x8, v), ov = __builtin_bit_cast(bf16x8, old[j * 2 + pr]);
          bf16x8 rv;
          if (p.epi == 1) {      // relu(bf16(conv + residual)): the rounding of mi_ew_bf16 op 7 / the tile kernel's MI_CONV_ADDRELU
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (__bf16)fmaxf((float)(__bf16)((float)nv[e] + (float)ov[e]), 0.f);
          } else {               // dy * (a > 0): mi_ew_bf16 op 2 / MI_CONV_RELUMASK
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (float)ov[e] > 0.f ? nv[e] : (__bf16)0.f;
          }
          v = __builtin_bit_cast(u32x4, rv);
        }
        if constexpr (MODE == 2 || MODE == 6) {
          // same double rounding as the tile kernel's accumulate path: bf16(result), then bf16(that + old)
          const bf16x8 nv = __builtin_bit_cast(bf16x8, v), ov = __builtin_bit_cast(bf16x8, old[j * 2 + pr]);
          bf16x8 rv;
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = (__bf16)((float)nv[e] + (float)ov[e]);
          if constexpr (MODE == 6) {   // ... then the ReLU mask (MI_CONV_ACCUM | MI_CONV_RELUMASK, the tile kernel's order)
            const bf16x8 mv = __builtin_bit_cast(bf16x8, aux6[j * 2 + pr]);
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (float)mv[e] > 0.f ? rv[e] : (__bf16)0.f;
          }
          v = __builtin_bit_cast(u32x4, rv);
        }
        if (!RAGGED_OK || ((wn * PT + j) * 32 + l31) < lim) *(u32x4*)(yt + yoff[j] + pr * 32) = v;
      }
    }
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    nbuf = nbuf + 1 == NBUF ? 0 : nbuf + 1;
  }

  C1_VMCNT(0);   // the tail DMAs still target this block's LDS
  if constexpr (STATS) {
    // one set of atomics per block: lanes fold over their 32 pixels, waves over the WN pixel groups through LDS
#pragma unroll
    for (int off = 1; off < 32; off <<= 1)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s1[r] += __shfl_xor(s1[r], off, 64);
        s2[r] += __shfl_xor(s2[r], off, 64);
      }
    __syncthreads();
    float* red = (float*)smem;   // [WN][WM * 32][2]
    if (l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = wm * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        red[(wn * WM * 32 + c) * 2 + 0] = s1[r];
        red[(wn * WM * 32 + c) * 2 + 1] = s2[r];
      }
    }
    __syncthreads();
    if (wave < WM && lane < 32) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int q = 0; q < WN; ++q) {
        a1 += red[(q * WM * 32 + wave * 32 + lane) * 2 + 0];
        a2 += red[(q * WM * 32 + wave * 32 + lane) * 2 + 1];
      }
      const C1Slice so = p.s[cot * WM + wave];
      double* sp = so.stats + (size_t)((int)blockIdx.x % so.nslots) * so.sld + lane * 2;
      if (!(p.dbg & 1)) {
        atomicAdd(sp, (double)a1);
        atomicAdd(sp + 1, (double)a2);
      }
    }
  }
  if constexpr (MODE == 3) {
    // ---- phase 2: every block's sums are in; BatchNorm + activation of this block's own tiles
    // (no static LDS: the ring may use the whole dynamic limit; the tile buffers are dead by now, red[] sits below 16 KB)
    int* const s_gave_up = (int*)(smem + 16384);
    bn_grid_barrier(g_c1_bar, gen0, (int)blockIdx.x, (int)gridDim.x, s_gave_up);
    const float poison = *s_gave_up ? __builtin_nanf("") : 0.f;   // (a timed-out wait: the sums are incomplete)
    const CBnFwd& bn = p.bn[sl.bnj];
    float scl, shl;   // of channel c0 + l31 (both halves of the wave compute it; block 0's first pixel group records it)
    cbn_finalize(bn, sl.stats + l31 * 2, sl.sld, sl.nslots, sl.c0 + l31, b == 0 && wn == 0 && h == 0, poison, scl, shl);
    float sc[2][8], sh[2][8];   // this lane's channels 16 pr + 8 h + e
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[pr][e] = __shfl(scl, pr * 16 + h * 8 + e, 64);
        sh[pr][e] = __shfl(shl, pr * 16 + h * 8 + e, 64);
      }
    const int act = bn.act;
    const bool has_res = bn.res != nullptr;
    const int ldab = bn.lda * 2, ldrb = bn.ldres * 2;
    unsigned aoff[PT], roff[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      aoff[j] = (unsigned)(((wn * PT + j) * 32 + l31) * ldab + h * 16);
      roff[j] = (unsigned)(((wn * PT + j) * 32 + l31) * ldrb + h * 16);
    }
    char* const a0 = (char*)(bn.a + sl.c0);
    const char* const r0 = (const char*)(bn.res + sl.c0);
    const size_t atile = (size_t)TPIX * (size_t)ldab, rtile = (size_t)TPIX * (size_t)ldrb;
    for (int i = 0; i < nt; ++i) {
      const size_t t = (size_t)(b + i * nb);
      const char* const yt = (const char*)sl.y + t * ytile;
      char* const at = a0 + t * atile;
      const char* const rt = r0 + t * rtile;
      u32x4 v[2 * PT], r[2 * PT];
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          v[j * 2 + pr] = *(const u32x4*)(yt + yoff[j] + pr * 32);
          r[j * 2 + pr] = has_res ? *(const u32x4*)(rt + roff[j] + pr * 32) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          *(u32x4*)(at + aoff[j] + pr * 32) = cbn_apply8(v[j * 2 + pr], sc[pr], sh[pr], act, has_res, r[j * 2 + pr]);
    }
  }
}

template <int K, int WM, int WN, int PT, int NBUF, int MODE, int XF = 0>
static int c1s_launch_one(const C1Launch& l, hipStream_t s) {
  auto fn = c1s_kernel<K, WM, WN, PT, NBUF, MODE, XF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)l.grid), dim3(WM * WN * 64), (size_t)l.lds, s, l.k);
  MI_CHECK_LAUNCH("conv1x1_stream");
  return MI_OK;
}

// ring depth by tile bytes: >= 48 KB of loads in flight per CU (HBM latency x per-CU share of the bandwidth)
constexpr int c1s_nbuf(int K, int tpix) { return tpix * K * 2 >= 65536 ? 2 : (tpix * K * 2 >= 32768 ? 3 : 4); }

// a (K, pixel tile) pair has a kernel when the tile splits into whole 1 KB DMA instructions per wave and the ring fits LDS
constexpr bool c1s_valid(int K, int tpix) {
  return (tpix * K * 2) % 8192 == 0 && tpix * K * 2 * c1s_nbuf(K, tpix) <= 160 * 1024;
}

template <int K, int MODE, int XF = 0>
static int c1s_launch_k(const C1Launch& l, hipStream_t s) {
  if constexpr (c1s_valid(K, 128)) {
    if (l.WM == 4 && l.PT == 2) return c1s_launch_one<K, 4, 2, 2, c1s_nbuf(K, 128), MODE, XF>(l, s);
    if (l.WM == 2 && l.PT == 1) return c1s_launch_one<K, 2, 4, 1, c1s_nbuf(K, 128), MODE, XF>(l, s);
  }
  if constexpr (c1s_valid(K, 64)) {
    if (l.WM == 4 && l.PT == 1) return c1s_launch_one<K, 4, 2, 1, c1s_nbuf(K, 64), MODE, XF>(l, s);
  }
  if constexpr (c1s_valid(K, 256)) {
    if (l.WM == 1 && l.PT == 1) return c1s_launch_one<K, 1, 8, 1, c1s_nbuf(K, 256), MODE, XF>(l, s);
  }
  MI_FAIL(MI_EINVAL, "conv1x1_stream: no kernel for K %d WM %d PT %d", K, l.WM, l.PT);
}
