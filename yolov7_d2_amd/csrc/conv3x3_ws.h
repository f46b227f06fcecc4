// Weight-stationary 3x3 stride-1 convolution for gfx950 (forward and data gradient of the K -> K BaseConv layers:
// Bottleneck conv2 of CSPDarknet / YOLOPAFPN, the cls / reg towers of YOLOXHead - yolov7/modeling/backbone/layers/
// wrappers.py:105-123, yolov7/modeling/head/yolox_head.py:73-102).
//
// The tile-per-block implicit GEMM streams the whole 3x3 weight set (K*K*18 bytes: 295 KB for 128 channels) through
// LDS for every 128-pixel tile - several times the bytes of the tile's own halo - and pays a barrier + DMA wait per
// (k-chunk, tap group) step.  Here
//   * blocks are persistent (one or two per CU) and each WAVE keeps the A fragments of its 32 output channels for all
//     9 taps x K input channels in registers for the whole launch (K = 128: 288 VGPRs, one wave per SIMD);
//   * the only thing that moves per tile is the input halo (10 x 18 pixels for an 8 x 16 tile): HBM -> LDS by LDS-DMA,
//     double-buffered one tile ahead behind counted vmcnt waits, ONE barrier per tile;
//   * the LDS image is [8-channel group][halo pixel][16 B] (one plane per 16-byte chunk) and the 32 pixel columns of an
//     MFMA are handed to the lanes so that each 16-lane service group of ds_read_b128 ({0-3,12-15,20-27} /
//     {4-11,16-19,28-31}) owns 16 CONSECUTIVE pixels of ONE tile row: every B fragment read is 256 contiguous bytes -
//     conflict-free for every tap without a swizzle - and its address is one per-lane base + an immediate
//     (k-step * plane + tap offset): ZERO VALU instructions per fragment (the XOR-swizzled pixel-major image of the
//     tile kernel needs a VGPR address per (k-step, tap column, pixel group), 6.4 VALU per MFMA);
//   * 9 x K/16 k-steps run back to back without any synchronisation (288 MFMAs per wave and tile at K = 128);
//   * outputs: bf16 -> permlane32_swap -> 16 contiguous bytes per lane, stored during the NEXT tile's main loop; BatchNorm
//     sums per lane, parked in LDS between tiles, one set of fp64 atomics per block.
// A launch carries up to 8 jobs (the head's three levels x two towers run as one launch); a block belongs to one job.
// MODE 0: plain, 1: + BatchNorm statistics, 2: y += result, 3: MODE 1 and then, behind a grid barrier, BatchNorm(train) +
// SiLU (+ residual) of the block's own output tiles (conv_bn.h; see conv1x1_stream.h MODE 3): conv2 + bn + act + shortcut
// of a Bottleneck (layers/wrappers.py:119-123) in one launch.
// XF 1 (with MODE 1): the job's input x is the RAW output of the producing convolution; the BatchNorm(train) + SiLU that
// BaseConv applies to it (wrappers.py:76-83) runs here, on the halo in LDS, by the wave that fetched the piece (conv_bn.h,
// BnXf): no bn_act_fwd launch, y is read once instead of y + a; the writer job stores the activated interior pixels.
#pragma once
#include "common.h"
#include "conv_bn.h"

#define W3_MAX_JOBS 8
#define W3_TW 16
// Geometry of a tile for stride S and tile height TH (output pixels): S = 1: 8 x 16 outputs from a 10 x 18 halo; S = 2 (the
// down-sampling convs of CSPDarknet / the PAFPN bottom-up path, darknetx.py:113-160, yolo_pafpn.py:60-77): TH x 16 outputs
// from a (2 TH + 1) x 33 halo whose COLUMNS are stored de-interleaved - the 17 even columns, then the 16 odd ones - so
// that the pixels 2 ox + dx of 16 consecutive outputs are 16 consecutive LDS positions for every tap: the B fragment reads
// stay 256 contiguous bytes with immediate tap offsets, exactly as for stride 1.
template <int S, int TH>
struct W3Geo {
  static constexpr int HH = S == 1 ? TH + 2 : 2 * TH + 1;      // halo rows
  static constexpr int HW = S == 1 ? W3_TW + 2 : 2 * W3_TW + 1;  // halo columns
  static constexpr int HPIX = HH * HW;
  static constexpr int HROWS = (HPIX + 63) / 64 * 64;            // padded to whole 64-pixel DMA instructions
  // LDS pixel offset of tap (dy, dx in 0..2) relative to the output pixel's base position
  static constexpr int tap(int dy, int dx) { return S == 1 ? dy * HW + dx : dy * HW + (dx & 1) * (W3_TW + 1) + (dx >> 1); }
};

struct W3Job {
  const __bf16* x;
  const u32x4* w;    // packed [tap][K/8][CoutPad][8]
  __bf16* y;
  double* stats;
  int ldx, ldy, N, H, W, tilesY, tilesX, ntiles;   // H x W: the OUTPUT map; ntiles = N * tilesY * tilesX
  int blk0, nblk;                                  // this job's blocks: [blk0, blk0 + nblk)
  int wld, nslots, sld, pad_;
  int tw[9];                                       // weight slab of tap position t = (dy + 1) * 3 + (dx + 1)
  int inH;                                         // input map height (= H for stride 1; input width = inW)
  CBnFwd bn;                                       // MODE 3
  int inW, xfw;                                    // xfw: this job stores the activated input (XF launches; one job per input tensor)
  const BnXf* xf;                                  // XF launches: the BatchNorm + activation of this job's INPUT (device record)
};
struct W3K {
  int njobs, dbg;
  W3Job j[W3_MAX_JOBS];
};
struct W3Launch {
  int K, MODE, grid, lds, S, XF;
  W3K k;
};

#define W3_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

static __device__ uint4 g_w3_zero_page[4];
static __device__ __attribute__((aligned(256))) unsigned g_w3_bar[MI_BN_BAR_WORDS];   // grid barrier of the MODE 3 launches (one at a time)

__device__ __forceinline__ void w3_glds16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory");
}

// A wave owns G pixel groups (32 pixels each) of the 8 x 16 tile and its 32 output channels.  The tile loop is software-
// pipelined through the MFMA stream: while the 9 * K / 16 k-steps of tile i run, the wave issues - a few k-steps apart -
// the LDS-DMA of tile i + 1's halo, the stores of tile i - 1's (already converted, packed) outputs and, in the accumulate
// mode, the loads of tile i's old values.  Issued in one burst at the tile boundary, those 20-28 KB per wave run at the
// chip's HBM rate with every CU in the same phase and the matrix pipes idle (measured: 2 + 2 us per 3.8 us tile).
template <int K, int WM, int WN, int G, int MODE, int S = 1, int TH = 8, int XF = 0>
__global__ __launch_bounds__(WM* WN * 64, (K == 128 || (S == 2 && K == 64)) ? 1 : 2) void w3_kernel(const W3K p) {
  static_assert(!XF || MODE == 1, "the input transform comes with the forward (statistics) mode");
  constexpr int NW = WM * WN, KC8 = K / 8, KS = K / 16;
  using Geo = W3Geo<S, TH>;
  constexpr int W3_TH = TH, W3_HW = Geo::HW, W3_HPIX = Geo::HPIX, W3_HROWS = Geo::HROWS;
  static_assert(WN * G * 32 == W3_TH * W3_TW, "pixel tile");
  constexpr int PLANE = W3_HROWS * 16;           // bytes of one 8-channel plane
  constexpr int NQ = (W3_HROWS / 64) * KC8;      // DMA instructions per tile: (64-pixel block, plane)
  constexpr int D = NQ / NW;                     // per wave
  static_assert(NQ % NW == 0 && D >= 1, "DMA split");
  constexpr int XB = KC8 * PLANE;                // bytes of one halo buffer
  static_assert((KC8 - 1) * PLANE + Geo::tap(2, 2) * 16 < 65536, "ds_read immediate");
  constexpr int NS = 9 * KS, NTS = NS / 3;       // k-steps, triple steps
  // schedule inside the main loop (triple-step index): DMA d at DST d, store s at SST s + SST - 1, old-value loads 2 l, 2 l + 1
  // at SST S + l.  (Two triple steps apart where the tile's k-loop is long enough, every triple step otherwise.)
  constexpr int NST = 2 * G;                     // 16-byte stores (and old-value loads) per wave and tile
  constexpr int DST = 2 * (D - 1) < NTS ? 2 : 1, SST = 2 * NST + NST / 2 <= NTS ? 2 : 1;
  static_assert(DST * (D - 1) < NTS && SST * NST + (MODE == 2 ? NST / 2 : 0) <= NTS, "the memory operations of a tile fit its main loop");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  int ji = 0;
  while (ji + 1 < p.njobs && (int)blockIdx.x >= p.j[ji + 1].blk0) ++ji;
  ji = __builtin_amdgcn_readfirstlane(ji);
  const W3Job& jb = p.j[ji];
  const int b = (int)blockIdx.x - jb.blk0, nb = jb.nblk;
  const int nt = (jb.ntiles - b + nb - 1) / nb;
  constexpr bool STATS = MODE == 1 || MODE == 3;
  unsigned gen0 = 0;   // thread 0 only: the barrier generation this launch starts in
  if constexpr (MODE == 3) {
    const int nblk = (int)gridDim.x;
    if (tid == 0)
      gen0 = __hip_atomic_load(bn_bar_gen(g_w3_bar, (int)blockIdx.x % (nblk < BN_BAR_G ? nblk : BN_BAR_G)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // every job field the tile loop needs, in registers: behind the "memory"-clobbering DMA statements the compiler would
  // re-load them from the argument segment, and each such s_load is followed by lgkmcnt(0) - which also drains the LDS
  // fragment reads in flight (20 pipeline drains per tile)
  const int H = jb.H, Wd = jb.W, tilesX = jb.tilesX, tpi = jb.tilesY * jb.tilesX;   // (H x Wd: the output map)
  const int inH = S == 1 ? jb.H : jb.inH, inW = S == 1 ? jb.W : jb.inW;
  const int ldxb = jb.ldx * 2, ldyb = jb.ldy * 2;
  const char* const xbase = (const char*)jb.x;
  char* const ybase = (char*)jb.y;
  const char* zpage = (const char*)g_w3_zero_page;
  asm volatile("" : "+s"(zpage));   // (pinned: its address otherwise comes from the GOT, one s_load per DMA instruction)
  // XF: (scale, shift) tables of the K input channels behind the halo ring and the per-lane sums; block 0 of the writer job
  // records the layer's statistics.  (The table loads go out before the weight loads below: both wait on L2 together.)
  float* const s_sc = (float*)(smem + 2 * (K / 8) * Geo::HROWS * 16 + 256 * 32 * 4);
  float* const s_sh = s_sc + K;
  int xf_act = 0, xf_ldab = 0;
  char* xf_a = nullptr;
  if constexpr (XF) {
    const BnXf* const xf = jb.xf;
    bnx_tables(xf, s_sc, s_sh, NW * 64, jb.xfw != 0 && b == 0);
    xf_act = xf->bn.act;
    xf_ldab = xf->bn.lda * 2;
    xf_a = jb.xfw ? (char*)xf->bn.a : nullptr;
  }

  // ---- weights: this wave's 32 output channels x 9 taps x K, resident in registers
  bf16x8 a[9][KS];
  {
    const u32x4* wb = jb.w + wm * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int slab = jb.tw[t] * KC8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[t][ks] = __builtin_bit_cast(bf16x8, wb[(size_t)(slab + ks * 2 + h) * jb.wld]);
    }
  }

  // ---- halo DMA, one instruction: q = (64-pixel block pb, plane k8): lane -> halo pixel pb * 64 + lane, 16 bytes =
  // channels 8 k8 .. of that pixel.  The planes of one 128-byte source line are requested by consecutive instructions of
  // one wave (the first misses, the others hit the CU's L1).  Geometry recomputed per instruction (~10 VALU per 1 KB).
  struct Org { int img, ty0, tx0; };   // a tile's image and first output pixel (two integer divisions: once per tile)
  auto tile_origin = [&](int i) {
    const int t = b + i * nb;
    Org o;
    o.img = t / tpi;
    const int rem = t - o.img * tpi, tyq = rem / tilesX;
    o.ty0 = tyq * W3_TH;
    o.tx0 = (rem - tyq * tilesX) * W3_TW;
    return o;
  };
  auto issue_x1 = [&](const Org& o, int buf, int d) {   // instruction d of a tile's halo into buffer buf
    const int iy0 = S * o.ty0 - 1, ix0 = S * o.tx0 - 1;
    const char* xt = xbase + ((size_t)o.img * inH * inW + (ptrdiff_t)iy0 * inW + ix0) * (ptrdiff_t)ldxb;
    const int q = wave * D + d;
    const int pb = q / KC8, k8 = q % KC8;
    const int pix = pb * 64 + lane;
    // LDS position -> halo pixel: row = pix / HW (exact multiply-shift for pix < 384); stride 2 stores the even columns
    // first: position c < 17 is column 2 c, position c >= 17 is column 2 (c - 17) + 1
    const int hy = S == 1 ? (int)(((unsigned)pix * 3641u) >> 16) : (int)(((unsigned)pix * 1986u) >> 16);
    const int hp = pix - hy * W3_HW;
    const int hx = S == 1 ? hp : (hp <= W3_TW ? 2 * hp : 2 * (hp - (W3_TW + 1)) + 1);
    const bool v = (pix < W3_HPIX) & ((unsigned)(iy0 + hy) < (unsigned)inH) & ((unsigned)(ix0 + hx) < (unsigned)inW);
    const char* g = v ? xt + (unsigned)((hy * inW + hx) * ldxb + k8 * 16) : zpage;
    w3_glds16(g, lds0 + buf * XB + k8 * PLANE + pb * 1024);
  };

  // ---- B fragments: MFMA column l31 of pixel group g is pixel (row 2 g' + lg, column lidx) of the tile, where (lg, lidx)
  // is the lane's place in its ds_read_b128 service group; plane (2 ks + h); tap (dy, dx) = + (dy * 18 + dx) pixels
  int lg, lidx;
  {
    const int l = l31;
    lg = ((l >= 4) & (l < 12)) | ((l >= 16) & (l < 20)) | (l >= 28);
    lidx = l - (l < 4 ? 0 : l < 12 ? 4 : l < 20 ? 8 : l < 28 ? 12 : 16);
  }
  unsigned bbase[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bbase[g] = (unsigned)(h * PLANE + (S * ((wn * G + g) * 2 + lg) * W3_HW + lidx) * 16);
  // output pixel of group g: (ty0 + (wn G + g) 2 + lg, tx0 + lidx); address (or null when outside the map)
  auto out_ptr = [&](const Org& o, int g) -> char* {
    const int py = o.ty0 + (wn * G + g) * 2 + lg, px = o.tx0 + lidx;
    char* q = ybase + (((size_t)o.img * H + py) * Wd + px) * (size_t)ldyb + wm * 64 + h * 16;
    return ((py < H) & (px < Wd)) ? q : nullptr;
  };

  // BatchNorm sums: per lane 16 channels x (sum, sumsq), kept in LDS between tiles ([8][threads][16 B] behind the two halo
  // buffers) - 32 more live registers across the main loop do not fit next to the weights
  float* const sacc = (float*)(smem + 2 * XB) + tid * 4;
  constexpr int SAS = NW * 64 * 4;   // floats between the 8 vectors of a lane
  if constexpr (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) *(f32x4*)(sacc + q * SAS) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  Org oc = tile_origin(0), op = oc;   // current / previous tile
#pragma unroll
  for (int d = 0; d < D; ++d) issue_x1(oc, 0, d);

  u32x4 pk[NST];   // packed bf16 outputs of the previous tile: [group][pair of 8-channel halves], stored during this tile
#pragma unroll
  for (int q = 0; q < NST; ++q) pk[q] = u32x4{0u, 0u, 0u, 0u};
  for (int i = 0; i < nt; ++i) {
    W3_VMCNT(0);                    // this wave's share of halo i (issued >= a third of a tile ago)
    if constexpr (XF) {
      // BatchNorm + SiLU of the pieces this wave fetched itself, in place (same q -> (pixel block, plane) map as issue_x1);
      // pixels outside the map stay zero; the writer job stores the tile's interior to the activated tensor
      if (i == 0) __syncthreads();   // the tables
      const int iy0 = S * oc.ty0 - 1, ix0 = S * oc.tx0 - 1;
      char* const Xw = smem + (i & 1) * XB;
      char* const at = xf_a + ((size_t)oc.img * inH * inW + (ptrdiff_t)iy0 * inW + ix0) * (ptrdiff_t)xf_ldab;
#pragma unroll 2
      for (int d = 0; d < D; ++d) {
        const int q = wave * D + d;
        const int pb = q / KC8, k8 = q % KC8;
        const int pix = pb * 64 + lane;
        const int hy = S == 1 ? (int)(((unsigned)pix * 3641u) >> 16) : (int)(((unsigned)pix * 1986u) >> 16);
        const int hp = pix - hy * W3_HW;
        const int hx = S == 1 ? hp : (hp <= W3_TW ? 2 * hp : 2 * (hp - (W3_TW + 1)) + 1);
        const bool v = (pix < W3_HPIX) & ((unsigned)(iy0 + hy) < (unsigned)inH) & ((unsigned)(ix0 + hx) < (unsigned)inW);
        const u32x4 o = bnx_apply_lds(Xw + k8 * PLANE + pb * 1024 + lane * 16, s_sc, s_sh, k8, xf_act, v);
        const bool inner = v & (hy >= 1) & (hy <= S * W3_TH) & (hx >= 1) & (hx <= S * W3_TW);
        if (xf_a && inner) *(u32x4*)(at + (unsigned)((hy * inW + hx) * xf_ldab + k8 * 16)) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // halo i complete; the other buffer is no longer read
    const bool more = (i + 1 < nt) & !(p.dbg & 2);
    const bool prev = i > 0;
    const Org on = tile_origin(i + 1 < nt ? i + 1 : i);
    const char* Xs = smem + (i & 1) * XB;
    u32x4 old[NST];

    f32x16 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    // 9 * KS k-steps, B fragments double-buffered one step ahead.  (Left to itself the compiler - at the register limit -
    // issues read, wait, MFMA, read, wait ... through ONE fragment register: every MFMA then waits a full LDS round trip,
    // and with one wave per SIMD nothing hides it.  The sched_barriers pin: reads of step s + 1, then the MFMAs of step s.)
    auto ldb = [&](int step, bf16x8(&dst)[G]) {
      const int t9 = step / KS, ks = step % KS, dy = t9 / 3, dx = t9 % 3;
#pragma unroll
      for (int g = 0; g < G; ++g)
        dst[g] = __builtin_bit_cast(bf16x8, *(const u32x4*)(Xs + bbase[g] + (2 * ks * PLANE + Geo::tap(dy, dx) * 16)));
    };
    // fragment ring of three: step s multiplies buffer s % 3 while the reads of step s + 2 are in flight (two steps = 256
    // matrix-pipe cycles ahead: with one wave per SIMD nothing else covers the LDS round trip)
    bf16x8 bf[3][G];
    ldb(0, bf[0]);
    ldb(1, bf[1]);
    if (!(p.dbg & 4))   // (dbg 4: timing experiment without the main loop)
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts) {
      // -- the tile's memory traffic: at most one DMA, one store and two old-value loads per triple step
      if (ts % DST == 0 && ts / DST < D) {
        if (more) issue_x1(on, (i + 1) & 1, ts / DST);
      }
      if (ts % SST == SST - 1 && ts / SST < NST) {
        char* q = prev ? out_ptr(op, (ts / SST) / 2) : nullptr;
        if (q) *(u32x4*)(q + ((ts / SST) % 2) * 32) = pk[ts / SST];
      }
      if constexpr (MODE == 2) {
        if (ts >= SST * NST && ts < SST * NST + NST / 2) {
#pragma unroll
          for (int l = 2 * (ts - SST * NST); l < 2 * (ts - SST * NST) + 2; ++l) {
            char* q = out_ptr(oc, l / 2);
            old[l] = u32x4{0u, 0u, 0u, 0u};
            if (q) {
              if (l % 2 == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(old[l]) : "v"(q) : "memory");
              else asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "+v"(old[l]) : "v"(q) : "memory");
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int st = 3 * ts + r;
        __builtin_amdgcn_sched_barrier(0);
        // step st: its MFMAs, interleaved one to one with the fragment reads of step st + 2
        if (st + 2 < NS) ldb(st + 2, bf[(r + 2) % 3]);
#pragma unroll
        for (int g = 0; g < G; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st / KS][st % KS], bf[r][g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // up to two VALU
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- convert: accumulators -> bf16 -> permlane32_swap pairs -> 16 contiguous bytes per lane, kept for the next tile
    if constexpr (MODE == 2) {
      if constexpr (NST == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1])::"memory");
      else if constexpr (NST == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3])::"memory");
      else
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(old[0]), "+v"(old[1]), "+v"(old[2]), "+v"(old[3]), "+v"(old[4]), "+v"(old[5]), "+v"(old[6]), "+v"(old[7])::"memory");
    }
    float s1[16], s2[16];
    if constexpr (STATS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 u = *(const f32x4*)(sacc + q * SAS), w = *(const f32x4*)(sacc + (4 + q) * SAS);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[4 * q + e] = u[e]; s2[4 * q + e] = w[e]; }
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bool pvg = true;
      if constexpr (STATS) pvg = out_ptr(oc, g) != nullptr;
      unsigned pw[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[g][4 * q + e];
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = pvg ? (float)o[e] : 0.f;
            s1[4 * q + e] += f;
            s2[4 * q + e] = __builtin_fmaf(f, f, s2[4 * q + e]);
          }
        }
        const u32x2 u = __builtin_bit_cast(u32x2, o);
        pw[2 * q] = u[0];
        pw[2 * q + 1] = u[1];
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int q0 = 2 * pr, q1 = 2 * pr + 1;
        const auto w0 = __builtin_amdgcn_permlane32_swap(pw[2 * q0], pw[2 * q1], false, false);
        const auto w1 = __builtin_amdgcn_permlane32_swap(pw[2 * q0 + 1], pw[2 * q1 + 1], false, false);
        u32x4 v = {w0[0], w1[0], w0[1], w1[1]};
        if constexpr (MODE == 2) {
          // same double rounding as the tile kernel's accumulate path: bf16(result), then bf16(that + old)
          const bf16x8 nv = __builtin_bit_cast(bf16x8, v), ov = __builtin_bit_cast(bf16x8, old[g * 2 + pr]);
          bf16x8 rv;
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = (__bf16)((float)nv[e] + (float)ov[e]);
          v = __builtin_bit_cast(u32x4, rv);
        }
        pk[g * 2 + pr] = v;
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *(f32x4*)(sacc + q * SAS) = f32x4{s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]};
        *(f32x4*)(sacc + (4 + q) * SAS) = f32x4{s2[4 * q], s2[4 * q + 1], s2[4 * q + 2], s2[4 * q + 3]};
      }
    }
    op = oc;
    oc = on;
  }
  // the last tile's outputs
#pragma unroll
  for (int q = 0; q < NST; ++q) {
    char* o = out_ptr(op, q / 2);
    if (o) *(u32x4*)(o + (q % 2) * 32) = pk[q];
  }

  W3_VMCNT(0);
  if constexpr (STATS) {
    float s1[16], s2[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 u = *(const f32x4*)(sacc + q * SAS), w = *(const f32x4*)(sacc + (4 + q) * SAS);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[4 * q + e] = u[e]; s2[4 * q + e] = w[e]; }
    }
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
    if (tid < WM * 32) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int q = 0; q < WN; ++q) {
        a1 += red[(q * WM * 32 + tid) * 2 + 0];
        a2 += red[(q * WM * 32 + tid) * 2 + 1];
      }
      double* sp = jb.stats + (size_t)((int)blockIdx.x % jb.nslots) * jb.sld + tid * 2;
      if (!(p.dbg & 1)) {
        atomicAdd(sp, (double)a1);
        atomicAdd(sp + 1, (double)a2);
      }
    }
  }
  if constexpr (MODE == 3) {
    // ---- phase 2: every block's sums are in; BatchNorm + activation (+ residual) of this block's own tiles
    // (no static LDS: the ring may use the whole dynamic limit; the tile buffers are dead by now, red[] sits below 16 KB)
    int* const s_gave_up = (int*)(smem + 16384);
    bn_grid_barrier(g_w3_bar, gen0, (int)blockIdx.x, (int)gridDim.x, s_gave_up);
    const float poison = *s_gave_up ? __builtin_nanf("") : 0.f;   // (a timed-out wait: the sums are incomplete)
    const CBnFwd& bn = jb.bn;
    float scl, shl;   // of channel wm * 32 + l31
    cbn_finalize(bn, jb.stats + (wm * 32 + l31) * 2, jb.sld, jb.nslots, wm * 32 + l31, b == 0 && wn == 0 && h == 0, poison, scl, shl);
    float sc[2][8], sh[2][8];   // this lane's channels wm * 32 + 16 pr + 8 h + e
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[pr][e] = __shfl(scl, pr * 16 + h * 8 + e, 64);
        sh[pr][e] = __shfl(shl, pr * 16 + h * 8 + e, 64);
      }
    const int act = bn.act;
    const bool has_res = bn.res != nullptr;
    const size_t ldab = (size_t)bn.lda * 2, ldrb = (size_t)bn.ldres * 2;
    char* const abase = (char*)bn.a + wm * 64 + h * 16;
    const char* const rbase = (const char*)bn.res + wm * 64 + h * 16;
    for (int i = 0; i < nt; ++i) {
      const Org o = tile_origin(i);
      u32x4 v[NST], r[NST];
      size_t pixi[G];
      bool ok[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int py = o.ty0 + (wn * G + g) * 2 + lg, px = o.tx0 + lidx;
        ok[g] = (py < H) & (px < Wd);
        pixi[g] = ((size_t)o.img * H + py) * Wd + px;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          v[g * 2 + pr] = r[g * 2 + pr] = u32x4{0u, 0u, 0u, 0u};
          if (ok[g]) {
            v[g * 2 + pr] = *(const u32x4*)(ybase + pixi[g] * (size_t)ldyb + wm * 64 + h * 16 + pr * 32);
            if (has_res) r[g * 2 + pr] = *(const u32x4*)(rbase + pixi[g] * ldrb + pr * 32);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          if (ok[g]) *(u32x4*)(abase + pixi[g] * ldab + pr * 32) = cbn_apply8(v[g * 2 + pr], sc[pr], sh[pr], act, has_res, r[g * 2 + pr]);
    }
  }
}

template <int K, int WM, int WN, int G, int MODE, int S = 1, int TH = 8, int XF = 0>
static int w3_launch_one(const W3Launch& l, hipStream_t s) {
  auto fn = w3_kernel<K, WM, WN, G, MODE, S, TH, XF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)l.grid), dim3(WM * WN * 64), (size_t)l.lds, s, l.k);
  MI_CHECK_LAUNCH("conv3x3_ws");
  return MI_OK;
}
// output rows of a tile / output channels of a block for (stride, K)
constexpr int w3_tile_h(int S, int K) { return S == 1 ? 8 : (K == 128 ? 2 : 4); }
constexpr int w3_block_cout(int S, int K) { return S == 1 ? K : (K == 32 ? 64 : 128); }
template <int MODE, int XF = 0>
static int w3_launch_mode(const W3Launch& l, hipStream_t s) {
  if (l.S == 2) {
    if constexpr (MODE == 1 || MODE == 0) {
      switch (l.K) {
        case 128: return w3_launch_one<128, 4, 1, 1, MODE, 2, 2, XF>(l, s);
        case 64: return w3_launch_one<64, 4, 1, 2, MODE, 2, 4, XF>(l, s);
        case 32: return w3_launch_one<32, 2, 2, 1, MODE, 2, 4, XF>(l, s);
      }
    }
    MI_FAIL(MI_EINVAL, "conv3x3_ws: stride 2 with K %d / mode %d", l.K, MODE);
  }
  switch (l.K) {
    case 128: return w3_launch_one<128, 4, 1, 4, MODE, 1, 8, XF>(l, s);
    case 64: return w3_launch_one<64, 2, 2, 2, MODE, 1, 8, XF>(l, s);
    case 32: return w3_launch_one<32, 1, 4, 1, MODE, 1, 8, XF>(l, s);
  }
  MI_FAIL(MI_EINVAL, "conv3x3_ws: K %d", l.K);
}
