// Fused multi-head attention core for DETR (config 4), forward + backward, on the gfx950 matrix cores.
//
//   O = softmax(scale * Q K^T + key_padding_mask) V        per (batch, head), head_dim = 32
//
// replaces the attention inside nn.MultiheadAttention as DETR's encoder / decoder layers call it
// (yolov7/modeling/backbone/detr_backbone.py:140,155-157,200-202,222-230): d_model 256 = 8 heads x 32, post-norm,
// key_padding_mask from the padded image batch.  The reference materialises the [B*8, Lq, Lk] fp32 score matrix
// (35 MB per image per encoder layer at 800x1333); here scores never leave registers (online softmax), and the
// backward recomputes them from Q, K and the saved log-sum-exp.
//
// Layouts: Q, K, V, O, dO, dQ, dK, dV are bf16 [L][B][E] (sequence-first, E = H*32, exactly the tensors
// nn.MultiheadAttention produces after its in-projection), i.e. "pixel = l*B + b, channel = h*32 + d" in the NHWC
// convention of the conv kernels, so the in/out projections are 1x1 convolutions of the same library.
// mask: uint8 [B][Lk], 1 = padded key (ignored); lse: fp32 [B][H][Lq].
//
// MFMA plan (v_mfma_f32_16x16x32_bf16, D = head_dim = 32 = ONE k-step for the score products):
//   forward / dQ :  S^T tile = K_tile (A: [16 keys][32 d]) x Q^T (B: [32 d][16 queries])   -> lane (t = query, g)
//                   holds keys 4g..4g+3: the softmax row statistics of a query need only 2 cross-lane steps, and
//                   two such tiles (keys 0-15, 16-31) ARE the A fragment (8 key slots) of the next product
//                   O / dQ += P (A: [16 q][32 key slots]) x V / K (B via ds_read_b64_tr_b16 from row-major LDS).
//   dK, dV      :  S tile = Q_tile (A) x K^T (B) -> lane (t = key, g) holds queries 4g..4g+3, again directly the
//                   A fragment of dV += P^T dO and dK += dS^T Q.
// A wave owns 16 queries (forward, dQ) or 16 keys (dK/dV); a block = 4 waves shares the K/V (or Q/dO) tiles in LDS.
#include <stdlib.h>
#include <string.h>
#include "common.h"

#define MHA_D 32

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_;
__device__ __forceinline__ bf16x8 mha_tr_read2(const char* base0, const char* base1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_)(base0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_)(base1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

struct MhaK {
  const __bf16* q;
  const __bf16* k;
  const __bf16* v;
  const uint8_t* mask;  // [B][Lk] or NULL
  __bf16* o;
  float* o32;           // [Lq][B][E] fp32 copy of o or NULL (forward: written; backward: the operand of delta)
  float* lse;           // [B][H][Lq]
  // backward
  const __bf16* dout;
  const float* delta;   // [B][H][Lq]
  __bf16* dq;
  __bf16* dk;
  __bf16* dv;
  int B, H, Lq, Lk, E;
  int ldq, ldk;        // row strides (elements) of q / dq and of k / dk: E, or 2 E when a self-attention's q and k are the two
                       // halves of ONE [T, 2E] projection (k = q + E); v, o, dout, dv, o32 always have row stride E
  float scale;
  // attention dropout (nn.MultiheadAttention(dropout=p) drops attention WEIGHTS, detr_backbone.py:140,200-202): a
  // weight survives iff mi_rng(seed, ((b*H + h)*Lq + q)*Lk + key) >= drop_thr; survivors are scaled by drop_scale =
  // 1/(1-p).  The mask is a pure function of (seed, index): the backward kernels recompute it, nothing is stored.
  unsigned drop_thr;   // 0 = no dropout
  float drop_scale;
  unsigned long long seed;
  const unsigned long long* seed_off;   // device word added to seed (mi_dropout_seed_offset) or NULL
};

// rk: mha_rng_key(p), taken ONCE at the top of a kernel (the seed word is a uniform load + a 64-bit hash: not per element).
// The element index is the linear index into [B][H][Lq][Lk] (what mi_mha_dropout_mask enumerates): a uniform 64-bit base
// per (b, h) plus a 32-bit offset q * Lk + key (the launchers require Lq * Lk < 2^32)
__device__ __forceinline__ unsigned long long mha_rng_key(const MhaK& p) {
  return p.drop_thr ? mi_rng_key(p.seed + (p.seed_off ? *p.seed_off : 0ull)) : 0ull;
}
__device__ __forceinline__ float mha_keep(const MhaK& p, unsigned long long rk, int b, int h, int q, int key) {
  if (p.drop_thr == 0u) return 1.f;
  const unsigned long long base = (unsigned long long)(b * p.H + h) * (unsigned long long)p.Lq * (unsigned long long)p.Lk;
  const unsigned off = (unsigned)q * (unsigned)p.Lk + (unsigned)key;
  return mi_rng32k(rk, base + off) >= p.drop_thr ? p.drop_scale : 0.f;
}

// LDS tile of 32 rows x 32 d (64-byte rows), 16-byte chunks XOR-swizzled by (row >> 2) & 3 for the direct b128
// fragment reads and -- equivalently for 32-byte groups -- (row >> 2) & 1 for the transpose reads.
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// stage rows [r0, r0+32) of a [L][B][E] tensor (head h, batch b) into a tile; rows >= L are zero
__device__ __forceinline__ void stage_tile(char* T, const __bf16* src, int r0, int L, int B, int E, int b, int h,
                                           int tid) {
  if (tid >= 0 && tid < 128) {
    const int row = tid >> 2, chunk = tid & 3;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (r0 + row < L) val = *(const u32x4*)(src + ((size_t)(r0 + row) * B + b) * E + h * MHA_D + chunk * 8);
    *(u32x4*)(T + tile_off(row, chunk)) = val;
  }
}

// direct fragment: lane (t = row, g) -> 8 consecutive d = 8g..8g+7 of row (r16*16 + t)
__device__ __forceinline__ bf16x8 frag_rows(const char* T, int r16, int t, int g) {
  return __builtin_bit_cast(bf16x8, *(const u32x4*)(T + tile_off(r16 * 16 + t, g)));
}
// transposed fragment for the B operand "rows as k slots": lane (t = d within group j, g) -> rows 4g+{0..3} and
// 16+4g+{0..3}; d group j in {0,1}
__device__ __forceinline__ bf16x8 frag_cols(const char* T, int j, int t, int g) {
  const int r0 = 4 * g + (t >> 2), r1 = 16 + r0;
  // 32-byte group j of a row sits at chunks 2j, 2j+1 -> swizzled by the same XOR (it preserves the pair: the XOR
  // value's low bit moves within the pair only if odd, so build the address per 8-byte piece explicitly)
  const int piece = t & 3;                    // 8-byte piece inside the 32-byte group
  const int c0 = 2 * j + (piece >> 1);        // 16-byte chunk holding the piece
  const int a0 = tile_off(r0, c0) + (piece & 1) * 8;
  const int a1 = tile_off(r1, c0) + (piece & 1) * 8;
  return mha_tr_read2(T + a0, T + a1);
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void mha_fwd_kernel(const MhaK p) {
  const unsigned long long rk = mha_rng_key(p);
  __shared__ __attribute__((aligned(16))) char Ks[2][2048], Vs[2][2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  // this lane's query (B operand of S^T: lane (t = query, g) holds d = 8g..8g+7)
  bf16x8 qf = {};
  const int myq = q0 + t;
  if (myq < p.Lq) qf = *(const bf16x8*)(p.q + ((size_t)myq * p.B + b) * p.ldq + h * MHA_D + g * 8);
  f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // [d group]: lane (t = d, g): queries 4g+r
  float m_run = -INFINITY, l_run = 0.f;  // of query `t` (replicated over g)
  const int ntiles = (p.Lk + 31) / 32;
  stage_tile(Ks[0], p.k, 0, p.Lk, p.B, p.ldk, b, h, tid);
  stage_tile(Vs[0], p.v, 0, p.Lk, p.B, p.E, b, h, tid - 128);
  for (int it = 0; it < ntiles; ++it) {
    __syncthreads();
    const int cur = it & 1;
    if (it + 1 < ntiles) {
      stage_tile(Ks[cur ^ 1], p.k, (it + 1) * 32, p.Lk, p.B, p.ldk, b, h, tid);
      stage_tile(Vs[cur ^ 1], p.v, (it + 1) * 32, p.Lk, p.B, p.E, b, h, tid - 128);
    }
    const int k0 = it * 32;
    // S^T tiles: keys [0,16) and [16,32) of this tile x 16 queries
    f32x4 s[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const bf16x8 kf = frag_rows(Ks[cur], a, t, g);
      s[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // lane (t = query, g): s[a][r] = score of key k0 + 16a + 4g + r
    float sv[8];
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * a + 4 * g + r;
        float x = s[a][r] * p.scale;
        const bool dead = key >= p.Lk || (p.mask && p.mask[(size_t)b * p.Lk + key]);
        x = dead ? -INFINITY : x;
        sv[a * 4 + r] = x;
        mx = fmaxf(mx, x);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);  // (m_run = -inf -> 0)
    float psum = 0.f;
    bf16x8 pf;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pe = (m_new == -INFINITY) ? 0.f : __expf(sv[e] - m_new);
      psum += pe;           // the softmax normaliser is taken BEFORE the dropout, as F.multi_head_attention_forward does
      pf[e] = (__bf16)(pe * mha_keep(p, rk, b, h, myq, k0 + 16 * (e >> 2) + 4 * g + (e & 3)));
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // rescale O: lane (t = d, g) holds queries 4g + r -> alpha of lane (4g + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(alpha, 4 * g + r, 64);
      oacc[0][r] *= ar;
      oacc[1][r] *= ar;
    }
    // O += P (A: lane (t = query, g), 8 key slots) x V (B: transpose read, d group j)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 vf = frag_cols(Vs[cur], j, t, g);
      oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, oacc[j], 0, 0, 0);
    }
  }
  // epilogue: O[q][d] / l ; lane (t = d, g): queries 4g + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = q0 + 4 * g + r;
    const float lr = __shfl(l_run, 4 * g + r, 64);
    const float inv = lr > 0.f ? 1.f / lr : 0.f;
    if (qq < p.Lq) {
      __bf16* op = p.o + ((size_t)qq * p.B + b) * p.E + h * MHA_D;
      op[t] = (__bf16)(oacc[0][r] * inv);
      op[16 + t] = (__bf16)(oacc[1][r] * inv);
      if (p.o32) {
        float* o3 = p.o32 + ((size_t)qq * p.B + b) * p.E + h * MHA_D;
        o3[t] = oacc[0][r] * inv;
        o3[16 + t] = oacc[1][r] * inv;
      }
    }
  }
  if (g == 0 && myq < p.Lq && p.lse)
    p.lse[((size_t)b * p.H + h) * p.Lq + myq] = (l_run > 0.f) ? m_run + __logf(l_run) : -INFINITY;
}

// ------------------------------------------------------------------ forward, version 2
// The first kernel crossed a block barrier and staged 2 + 2 KB for every 32 keys (33 barriers at L = 1050), gave each wave
// 16 queries of a 64-query block (544 blocks of 4 waves) and ran one softmax update - two cross-lane maxima, two sums, four
// broadcasts of the rescale factor - per 4 tiny MFMAs: 92 TFLOP/s kernel-only at B = 4, L = 1050.  With head_dim 32 the
// matrix cores are NOT the limit (128 flops per score = 0.13 SIMD cycles): the kernel is bound by vector instructions per
// score (max, fma, exp2, bf16 pack = 3.5 issue slots of 4 cycles per 64 scores = 0.22 cycles per score at best) and by how
// the 16-query groups fall on the 1024 SIMDs.  Hence
//   * a block is nwv waves x 16 queries, nwv chosen so that the grid is ONE round of blocks (L = 1050, B x H = 32:
//     9 waves = 144 queries, 8 blocks per head, 256 blocks on 256 CUs);
//   * K / V stream through LDS in chunks of 128 keys, double-buffered, ONE barrier per chunk; the next chunk's global
//     loads are issued (unconditionally - no select on the loaded value) before the chunk's compute and written to LDS
//     after it, so that only the LDS store waits for them;
//   * one softmax step per chunk: 32 scores per lane in one basic block.  Per score: half a max3, one fma (scale and
//     reference maximum folded: exp2(s * c - m * c)), one exp2, half a bf16 pack.  The key-padding mask is a 0 / -inf bias
//     row staged with the chunk that enters the score MFMA as its accumulator input; the normaliser l = P x ones is a third
//     MFMA accumulating exactly the bf16 weights that entered O (consistent normalisation; with dropout it is summed on
//     the vector unit before the drop instead);
//   * the reference maximum is lazy (see below): the common step has no cross-lane traffic.
// (Measured and dropped: sharing a 16-query group between four waves by chunk to even out the SIMDs - the waves run in
// lockstep with the chunk barriers, so the split group is still processed serially.)
#define MHA2_KC 128
template <bool DROP>
__global__ __launch_bounds__(768) void mha_fwd2_kernel(const MhaK p) {
  const unsigned long long rk = mha_rng_key(p);
  __shared__ __attribute__((aligned(16))) char Ks[2 * 8192], Vs[2 * 8192];
  __shared__ __attribute__((aligned(16))) float Mb[2][MHA2_KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x;
  const int t = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.x * (nth >> 2) + wave * 16;      // nth / 64 waves x 16 queries
  const int myq = q0 + t;
  bf16x8 qf = {};
  if (myq < p.Lq) qf = *(const bf16x8*)(p.q + ((size_t)myq * p.B + b) * p.ldq + h * MHA_D + g * 8);
  f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 lacc = {0.f, 0.f, 0.f, 0.f};                       // normaliser, same layout as oacc (every column alike)
  float m_run = -INFINITY, neg_m = 0.f, l_part = 0.f;      // reference maximum of query t in raw score units; -m * sc2
  const float sc2 = p.scale * 1.44269504088896341f;        // raw score -> exp2 domain
  const float slack = 8.f / sc2;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.f;
  const int nchunks = (p.Lk + MHA2_KC - 1) / MHA2_KC;
  // this thread's share of a chunk: two 16-byte pieces of K and of V (512 pieces each; a block of more than 256 threads
  // repeats the last piece) and one mask byte.  Loads are unconditional - rows past the last key repeat it (finite
  // values; their bias is -inf, their weight exactly 0) - so that nothing but the LDS store waits for them.
  u32x4 kreg[2], vreg[2];
  unsigned char mreg = 0;
  const uint8_t* const mrow = p.mask ? p.mask + (size_t)b * p.Lk : (const uint8_t*)p.k;
  auto load_chunk = [&](int c) {
    const int k0 = c * MHA2_KC;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + u * nth;
      idx = idx < 511 ? idx : 511;
      int row = k0 + (idx >> 2);
      row = row < p.Lk ? row : p.Lk - 1;
      const size_t rb = (size_t)row * p.B + b;
      const int tail = h * MHA_D + (idx & 3) * 8;
      kreg[u] = *(const u32x4*)(p.k + rb * p.ldk + tail);
      vreg[u] = *(const u32x4*)(p.v + rb * p.E + tail);
    }
    const int key = k0 + (tid & (MHA2_KC - 1));
    mreg = mrow[key < p.Lk ? key : p.Lk - 1];
  };
  auto store_chunk = [&](int c) {
    const int buf = c & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + u * nth;
      idx = idx < 511 ? idx : 511;
      const int row = idx >> 2;
      const int o = buf * 8192 + (row >> 5) * 2048 + tile_off(row & 31, idx & 3);
      *(u32x4*)(Ks + o) = kreg[u];
      *(u32x4*)(Vs + o) = vreg[u];
    }
    const int key = c * MHA2_KC + (tid & (MHA2_KC - 1));
    Mb[buf][tid & (MHA2_KC - 1)] = (key >= p.Lk || (p.mask && mreg)) ? -INFINITY : 0.f;
  };
  load_chunk(0);
  store_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();   // chunk c is visible; the other buffer is no longer read
    const int cur = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    const int k0 = c * MHA2_KC;
    // S^T: eight 16-key x 16-query tiles, the mask bias (0 / -inf) entering as the accumulator input;
    // lane (t = query, g): s[a][r] = raw score of key k0 + 16 a + 4 g + r.  (Keys past Lk: bias -inf, weight 0.)
    f32x4 s[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const bf16x8 kf = frag_rows(Ks + cur * 8192 + (a >> 1) * 2048, a & 1, t, g);
      const f32x4 bias = *(const f32x4*)(&Mb[cur][16 * a + 4 * g]);
      s[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, bias, 0, 0, 0);
    }
    float mx = fmaxf(s[0][0], s[0][1]);
#pragma unroll
    for (int e = 2; e < 32; e += 2) mx = fmaxf(fmaxf(mx, s[e >> 2][e & 3]), s[e >> 2][(e & 3) + 1]);   // v_max3_f32
    // the reference maximum is LAZY: it moves only when some score of the wave exceeds it by more than `slack` (exp2
    // arguments stay <= 8: p <= 256, inside bf16's exponent range and harmless in the fp32 sums); the common step then
    // has no cross-lane traffic at all.  O and l carry the same factor, so the result does not depend on it.
    if (__any(mx > m_run + slack)) {
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * sc2);
      m_run = m_new;
      neg_m = (m_new == -INFINITY) ? 0.f : -m_new * sc2;
      if constexpr (DROP) l_part *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ar = __shfl(alpha, 4 * g + r, 64);
        oacc[0][r] *= ar;
        oacc[1][r] *= ar;
        lacc[r] *= ar;
      }
    }
    // O += P (A: lane (t = query, g), 8 key slots of a 32-key tile) x V (B: transpose read, d group j)
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * t2 + (e >> 2)][e & 3], sc2, neg_m));
        if constexpr (DROP) {
          l_part += pe;    // the softmax normaliser is taken BEFORE the dropout, as F.multi_head_attention_forward does
          pe *= mha_keep(p, rk, b, h, myq, k0 + 32 * t2 + 16 * (e >> 2) + 4 * g + (e & 3));
        }
        pf[e] = (__bf16)pe;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 vf = frag_cols(Vs + cur * 8192 + t2 * 2048, j, t, g);
        oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, oacc[j], 0, 0, 0);
      }
      if constexpr (!DROP) lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, ones, lacc, 0, 0, 0);
    }
    if (c + 1 < nchunks) store_chunk(c + 1);
  }
  if constexpr (DROP) {   // normaliser of query 4g + r in the O layout
    l_part += __shfl_xor(l_part, 16, 64);
    l_part += __shfl_xor(l_part, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) lacc[r] = __shfl(l_part, 4 * g + r, 64);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = q0 + 4 * g + r;
    const float inv = lacc[r] > 0.f ? 1.f / lacc[r] : 0.f;
    if (qq < p.Lq) {
      __bf16* op = p.o + ((size_t)qq * p.B + b) * p.E + h * MHA_D;
      op[t] = (__bf16)(oacc[0][r] * inv);
      op[16 + t] = (__bf16)(oacc[1][r] * inv);
      if (p.o32) {
        float* o3 = p.o32 + ((size_t)qq * p.B + b) * p.E + h * MHA_D;
        o3[t] = oacc[0][r] * inv;
        o3[16 + t] = oacc[1][r] * inv;
      }
    }
  }
  // lse of query t: its normaliser sits in lanes of group t >> 2, register t & 3
  float lq = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = __shfl(lacc[r], (t >> 2) * 16, 64);
    lq = (t & 3) == r ? v : lq;
  }
  if (g == 0 && myq < p.Lq && p.lse)
    p.lse[((size_t)b * p.H + h) * p.Lq + myq] = (lq > 0.f) ? (m_run * sc2 + log2f(lq)) * 0.69314718055994531f : -INFINITY;
}

// ------------------------------------------------------------------ backward
// delta[b][h][q] = sum_d dO[q][d] * O[q][d].  With the bf16 O this is the backward's weak point: dS = P o (dP - delta) is a
// difference of nearly equal numbers whenever the values of a row's keys are alike (a freshly initialised DETR: |delta| is
// 10 - 100 x |dP - delta|), and the 2^-9 rounding of the stored O, coherent over all keys of the row, comes out of the
// difference as a 30 - 80 % error of dq (tools/attn_bwd_error.py on operands dumped from the device: dq rel 0.78 -> 0.0014
// with the fp32 O, every other term unchanged).  The forward therefore also writes O in fp32 (p.o32) for this kernel.
__global__ __launch_bounds__(256) void mha_delta_kernel(const MhaK p, float* delta) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // over B*H*Lq*4 (4 threads of 8 d per row)
  const int part = idx & 3, row = idx >> 2;
  if (row >= p.B * p.H * p.Lq) return;
  const int q = row % p.Lq, bh = row / p.Lq, b = bh / p.H, h = bh % p.H;
  const size_t off = ((size_t)q * p.B + b) * p.E + h * MHA_D + part * 8;
  const bf16x8 a = *(const bf16x8*)(p.dout + off);
  float s = 0.f;
  if (p.o32) {
    const f32x4 c0 = *(const f32x4*)(p.o32 + off), c1 = *(const f32x4*)(p.o32 + off + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += (float)a[e] * c0[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) s += (float)a[4 + e] * c1[e];
  } else {
    const bf16x8 c = *(const bf16x8*)(p.o + off);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (part == 0) delta[row] = s;
}

// dQ: same loop structure as the forward (a wave owns 16 queries, streams K/V tiles)
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(const MhaK p) {
  const unsigned long long rk = mha_rng_key(p);
  __shared__ __attribute__((aligned(16))) char Ks[2][2048], Vs[2][2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int myq = q0 + t;
  bf16x8 qf = {}, dof = {};
  float lse = 0.f, dl = 0.f;
  if (myq < p.Lq) {
    const size_t rb = (size_t)myq * p.B + b;
    qf = *(const bf16x8*)(p.q + rb * p.ldq + h * MHA_D + g * 8);
    dof = *(const bf16x8*)(p.dout + rb * p.E + h * MHA_D + g * 8);
    lse = p.lse[((size_t)b * p.H + h) * p.Lq + myq];
    dl = p.delta[((size_t)b * p.H + h) * p.Lq + myq];
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int ntiles = (p.Lk + 31) / 32;
  stage_tile(Ks[0], p.k, 0, p.Lk, p.B, p.ldk, b, h, tid);
  stage_tile(Vs[0], p.v, 0, p.Lk, p.B, p.E, b, h, tid - 128);
  for (int it = 0; it < ntiles; ++it) {
    __syncthreads();
    const int cur = it & 1;
    if (it + 1 < ntiles) {
      stage_tile(Ks[cur ^ 1], p.k, (it + 1) * 32, p.Lk, p.B, p.ldk, b, h, tid);
      stage_tile(Vs[cur ^ 1], p.v, (it + 1) * 32, p.Lk, p.B, p.E, b, h, tid - 128);
    }
    const int k0 = it * 32;
    bf16x8 dsf;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const bf16x8 kf = frag_rows(Ks[cur], a, t, g), vf = frag_rows(Vs[cur], a, t, g);
      const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * a + 4 * g + r;
        const bool dead = key >= p.Lk || (p.mask && p.mask[(size_t)b * p.Lk + key]) || myq >= p.Lq;
        const float pe = (dead || lse == -INFINITY) ? 0.f : __expf(s[r] * p.scale - lse);
        dsf[a * 4 + r] = (__bf16)(pe * (dp[r] * mha_keep(p, rk, b, h, myq, key) - dl) * p.scale);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 kc = frag_cols(Ks[cur], j, t, g);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, kc, acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = q0 + 4 * g + r;
    if (qq < p.Lq) {
      __bf16* op = p.dq + ((size_t)qq * p.B + b) * p.ldq + h * MHA_D;
      op[t] = (__bf16)acc[0][r];
      op[16 + t] = (__bf16)acc[1][r];
    }
  }
}

// dK, dV: a wave owns 16 keys and streams Q / dO tiles
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(const MhaK p) {
  const unsigned long long rk = mha_rng_key(p);
  __shared__ __attribute__((aligned(16))) char Qs[2][2048], Ds[2][2048];
  __shared__ float Ls[2][32], Dl[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int key0 = blockIdx.x * 64 + wave * 16;
  const int mykey = key0 + t;
  bf16x8 kf = {}, vf = {};
  bool kdead = mykey >= p.Lk;
  if (!kdead) {
    const size_t rb = (size_t)mykey * p.B + b;
    kf = *(const bf16x8*)(p.k + rb * p.ldk + h * MHA_D + g * 8);
    vf = *(const bf16x8*)(p.v + rb * p.E + h * MHA_D + g * 8);
    kdead = p.mask && p.mask[(size_t)b * p.Lk + mykey];
  }
  f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int ntiles = (p.Lq + 31) / 32;
  auto stage_stats = [&](int buf, int r0) {
    if (tid >= 192 && tid < 224) {
      const int row = tid - 192;
      const bool ok = r0 + row < p.Lq;
      float lv = ok ? p.lse[((size_t)b * p.H + h) * p.Lq + r0 + row] : INFINITY;  // exp(s - inf) = 0
      if (lv == -INFINITY) lv = INFINITY;  // fully masked query row: contributes nothing
      Ls[buf][row] = lv;
      Dl[buf][row] = ok ? p.delta[((size_t)b * p.H + h) * p.Lq + r0 + row] : 0.f;
    }
  };
  stage_tile(Qs[0], p.q, 0, p.Lq, p.B, p.ldq, b, h, tid);
  stage_tile(Ds[0], p.dout, 0, p.Lq, p.B, p.E, b, h, tid - 128);
  stage_stats(0, 0);
  for (int it = 0; it < ntiles; ++it) {
    __syncthreads();
    const int cur = it & 1;
    if (it + 1 < ntiles) {
      stage_tile(Qs[cur ^ 1], p.q, (it + 1) * 32, p.Lq, p.B, p.ldq, b, h, tid);
      stage_tile(Ds[cur ^ 1], p.dout, (it + 1) * 32, p.Lq, p.B, p.E, b, h, tid - 128);
      stage_stats(cur ^ 1, (it + 1) * 32);
    }
    bf16x8 pf, dsf;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      // S tile = Q (A: lane (t = query 16a + t, g)) x K^T (B: lane (t = key, g)) -> lane (t = key, g): queries 16a+4g+r
      const bf16x8 qa = frag_rows(Qs[cur], a, t, g), da = frag_rows(Ds[cur], a, t, g);
      const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * a + 4 * g + r;
        const float pe = kdead ? 0.f : __expf(s[r] * p.scale - Ls[cur][ql]);
        const float keep = mha_keep(p, rk, b, h, it * 32 + ql, mykey);
        pf[a * 4 + r] = (__bf16)(pe * keep);
        dsf[a * 4 + r] = (__bf16)(pe * (dp[r] * keep - Dl[cur][ql]) * p.scale);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 dc = frag_cols(Ds[cur], j, t, g), qc = frag_cols(Qs[cur], j, t, g);
      dv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, dc, dv[j], 0, 0, 0);
      dk[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, qc, dk[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = key0 + 4 * g + r;
    if (kk < p.Lk) {
      const size_t off = ((size_t)kk * p.B + b) * p.E + h * MHA_D, offk = ((size_t)kk * p.B + b) * p.ldk + h * MHA_D;
      p.dk[offk + t] = (__bf16)dk[0][r];
      p.dk[offk + 16 + t] = (__bf16)dk[1][r];
      p.dv[off + t] = (__bf16)dv[0][r];
      p.dv[off + 16 + t] = (__bf16)dv[1][r];
    }
  }
}

// ------------------------------------------------------------------ backward, version 2
// The forward v2 structure for the two backward kernels: nwv waves x 16 rows per block sized to one round of blocks, the
// streamed operand pair in 128-row LDS chunks (double-buffered, one barrier per chunk, unconditional prefetch loads), one
// basic block per chunk.  Per score: one fma + exp2 (P recomputed from the saved log-sum-exp in the exp2 domain), one
// multiply, the bf16 packs; "- delta" enters the dP product as its accumulator input, the key-padding mask the score
// product's (dQ) or is applied to the finished rows (dK / dV: a padded key's rows are computed and discarded); the
// softmax scale multiplies the accumulators once at the end.
// dQ: a wave owns 16 queries (as the forward), streams K / V
template <bool DROP>
__global__ __launch_bounds__(768) void mha_bwd_dq2_kernel(const MhaK p) {
  const unsigned long long rk = mha_rng_key(p);
  __shared__ __attribute__((aligned(16))) char Ks[2 * 8192], Vs[2 * 8192];
  __shared__ __attribute__((aligned(16))) float Mb[2][MHA2_KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x;
  const int t = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.x * (nth >> 2) + wave * 16;
  const int myq = q0 + t;
  bf16x8 qf = {}, dof = {};
  float neg_l = -INFINITY, dl = 0.f;   // -lse * log2(e) of query t (-inf: a fully masked or out-of-range row weighs nothing)
  if (myq < p.Lq) {
    const size_t rb = (size_t)myq * p.B + b;
    qf = *(const bf16x8*)(p.q + rb * p.ldq + h * MHA_D + g * 8);
    dof = *(const bf16x8*)(p.dout + rb * p.E + h * MHA_D + g * 8);
    const float lse = p.lse[((size_t)b * p.H + h) * p.Lq + myq];
    neg_l = lse == -INFINITY ? -INFINITY : -lse * 1.44269504088896341f;
    dl = p.delta[((size_t)b * p.H + h) * p.Lq + myq];
  }
  const float sc2 = p.scale * 1.44269504088896341f;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const f32x4 ndl = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-dl, -dl, -dl, -dl};
  const int nchunks = (p.Lk + MHA2_KC - 1) / MHA2_KC;
  u32x4 kreg[2], vreg[2];
  unsigned char mreg = 0;
  const uint8_t* const mrow = p.mask ? p.mask + (size_t)b * p.Lk : (const uint8_t*)p.k;
  auto load_chunk = [&](int c) {
    const int k0 = c * MHA2_KC;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + u * nth;
      idx = idx < 511 ? idx : 511;
      int row = k0 + (idx >> 2);
      row = row < p.Lk ? row : p.Lk - 1;
      const size_t rb = (size_t)row * p.B + b;
      const int tail = h * MHA_D + (idx & 3) * 8;
      kreg[u] = *(const u32x4*)(p.k + rb * p.ldk + tail);
      vreg[u] = *(const u32x4*)(p.v + rb * p.E + tail);
    }
    const int key = k0 + (tid & (MHA2_KC - 1));
    mreg = mrow[key < p.Lk ? key : p.Lk - 1];
  };
  auto store_chunk = [&](int c) {
    const int buf = c & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + u * nth;
      idx = idx < 511 ? idx : 511;
      const int row = idx >> 2;
      const int o = buf * 8192 + (row >> 5) * 2048 + tile_off(row & 31, idx & 3);
      *(u32x4*)(Ks + o) = kreg[u];
      *(u32x4*)(Vs + o) = vreg[u];
    }
    const int key = c * MHA2_KC + (tid & (MHA2_KC - 1));
    Mb[buf][tid & (MHA2_KC - 1)] = (key >= p.Lk || (p.mask && mreg)) ? -INFINITY : 0.f;
  };
  load_chunk(0);
  store_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    const int cur = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    const int k0 = c * MHA2_KC;
    // S^T and dP^T - delta: lane (t = query, g): [a][r] = key k0 + 16 a + 4 g + r
    f32x4 s[8], dp[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const bf16x8 kf = frag_rows(Ks + cur * 8192 + (a >> 1) * 2048, a & 1, t, g);
      const bf16x8 vf = frag_rows(Vs + cur * 8192 + (a >> 1) * 2048, a & 1, t, g);
      const f32x4 bias = *(const f32x4*)(&Mb[cur][16 * a + 4 * g]);
      s[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, bias, 0, 0, 0);
      dp[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof, ndl, 0, 0, 0);
    }
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
      bf16x8 dsf;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int a = 2 * t2 + (e >> 2), r = e & 3;
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[a][r], sc2, neg_l));   // (masked key: s = -inf -> 0)
        float ds;
        if constexpr (DROP) ds = pe * (dp[a][r] * mha_keep(p, rk, b, h, myq, k0 + 16 * a + 4 * g + r) - dl);
        else ds = pe * dp[a][r];
        dsf[e] = (__bf16)ds;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 kc = frag_cols(Ks + cur * 8192 + t2 * 2048, j, t, g);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, kc, acc[j], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) store_chunk(c + 1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = q0 + 4 * g + r;
    if (qq < p.Lq) {
      __bf16* op = p.dq + ((size_t)qq * p.B + b) * p.ldq + h * MHA_D;
      op[t] = (__bf16)(acc[0][r] * p.scale);
      op[16 + t] = (__bf16)(acc[1][r] * p.scale);
    }
  }
}

// dK, dV: a wave owns 16 keys, streams Q / dO with the queries' -lse * log2(e) and -delta
template <bool DROP>
__global__ __launch_bounds__(768) void mha_bwd_dkv2_kernel(const MhaK p) {
  const unsigned long long rk = mha_rng_key(p);
  __shared__ __attribute__((aligned(16))) char Qs[2 * 8192], Ds[2 * 8192];
  __shared__ __attribute__((aligned(16))) float Nl[2][MHA2_KC], Nd[2][MHA2_KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x;
  const int t = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int key0 = blockIdx.x * (nth >> 2) + wave * 16;
  const int mykey = key0 + t;
  bf16x8 kf = {}, vf = {};
  {
    const int row = mykey < p.Lk ? mykey : p.Lk - 1;   // (rows past the last key and padded keys are discarded at the end)
    const size_t rb = (size_t)row * p.B + b;
    kf = *(const bf16x8*)(p.k + rb * p.ldk + h * MHA_D + g * 8);
    vf = *(const bf16x8*)(p.v + rb * p.E + h * MHA_D + g * 8);
  }
  const float sc2 = p.scale * 1.44269504088896341f;
  f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int nchunks = (p.Lq + MHA2_KC - 1) / MHA2_KC;
  u32x4 qreg[2], dreg[2];
  float lreg = 0.f, ereg = 0.f;
  const float* const lrow = p.lse + ((size_t)b * p.H + h) * p.Lq;
  const float* const drow = p.delta + ((size_t)b * p.H + h) * p.Lq;
  auto load_chunk = [&](int c) {
    const int r0 = c * MHA2_KC;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + u * nth;
      idx = idx < 511 ? idx : 511;
      int row = r0 + (idx >> 2);
      row = row < p.Lq ? row : p.Lq - 1;
      const size_t rb = (size_t)row * p.B + b;
      const int tail = h * MHA_D + (idx & 3) * 8;
      qreg[u] = *(const u32x4*)(p.q + rb * p.ldq + tail);
      dreg[u] = *(const u32x4*)(p.dout + rb * p.E + tail);
    }
    const int qq = r0 + (tid & (MHA2_KC - 1));
    lreg = lrow[qq < p.Lq ? qq : p.Lq - 1];
    ereg = drow[qq < p.Lq ? qq : p.Lq - 1];
  };
  auto store_chunk = [&](int c) {
    const int buf = c & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int idx = tid + u * nth;
      idx = idx < 511 ? idx : 511;
      const int row = idx >> 2;
      const int o = buf * 8192 + (row >> 5) * 2048 + tile_off(row & 31, idx & 3);
      *(u32x4*)(Qs + o) = qreg[u];
      *(u32x4*)(Ds + o) = dreg[u];
    }
    const int qq = c * MHA2_KC + (tid & (MHA2_KC - 1));
    const bool live = qq < p.Lq && lreg != -INFINITY;   // (a fully masked query row contributes nothing)
    Nl[buf][tid & (MHA2_KC - 1)] = live ? -lreg * 1.44269504088896341f : -INFINITY;
    Nd[buf][tid & (MHA2_KC - 1)] = qq < p.Lq ? -ereg : 0.f;
  };
  load_chunk(0);
  store_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    const int cur = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    const int r0 = c * MHA2_KC;
    // S and dP - delta: lane (t = key, g): [a][r] = query r0 + 16 a + 4 g + r.  All eight tile pairs of the chunk up front
    // (one 32-query tile at a time with dropout: its random numbers need the registers)
    constexpr int TG = DROP ? 1 : 4;
#pragma unroll
    for (int tg = 0; tg < 4; tg += TG) {
      f32x4 s[2 * TG], dp[2 * TG];
#pragma unroll
      for (int aa = 0; aa < 2 * TG; ++aa) {
        const int a = 2 * tg + aa;
        const bf16x8 qa = frag_rows(Qs + cur * 8192 + (a >> 1) * 2048, a & 1, t, g);
        const bf16x8 da = frag_rows(Ds + cur * 8192 + (a >> 1) * 2048, a & 1, t, g);
        const f32x4 nd = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(&Nd[cur][16 * a + 4 * g]);
        s[aa] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dp[aa] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf, nd, 0, 0, 0);
      }
#pragma unroll
      for (int tt = 0; tt < TG; ++tt) {
        const int t2 = tg + tt;
        bf16x8 pf, dsf;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int a = 2 * t2 + hh, aa = 2 * tt + hh;
          const f32x4 nl = *(const f32x4*)(&Nl[cur][16 * a + 4 * g]);
          f32x4 nd = {0.f, 0.f, 0.f, 0.f};
          if constexpr (DROP) nd = *(const f32x4*)(&Nd[cur][16 * a + 4 * g]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[aa][r], sc2, nl[r]));
            if constexpr (DROP) {
              const float keep = mha_keep(p, rk, b, h, r0 + 16 * a + 4 * g + r, mykey);
              pf[hh * 4 + r] = (__bf16)(pe * keep);
              dsf[hh * 4 + r] = (__bf16)(pe * (dp[aa][r] * keep + nd[r]));
            } else {
              pf[hh * 4 + r] = (__bf16)pe;
              dsf[hh * 4 + r] = (__bf16)(pe * dp[aa][r]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 dc = frag_cols(Ds + cur * 8192 + t2 * 2048, j, t, g), qc = frag_cols(Qs + cur * 8192 + t2 * 2048, j, t, g);
          dv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, dc, dv[j], 0, 0, 0);
          dk[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, qc, dk[j], 0, 0, 0);
        }
      }
    }
    if (c + 1 < nchunks) store_chunk(c + 1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = key0 + 4 * g + r;
    if (kk < p.Lk) {
      const bool dead = p.mask && p.mask[(size_t)b * p.Lk + kk];   // a padded key: its rows (possibly inf / NaN) are discarded
      const size_t off = ((size_t)kk * p.B + b) * p.E + h * MHA_D, offk = ((size_t)kk * p.B + b) * p.ldk + h * MHA_D;
      p.dk[offk + t] = (__bf16)(dead ? 0.f : dk[0][r] * p.scale);
      p.dk[offk + 16 + t] = (__bf16)(dead ? 0.f : dk[1][r] * p.scale);
      p.dv[off + t] = (__bf16)(dead ? 0.f : dv[0][r]);
      p.dv[off + 16 + t] = (__bf16)(dead ? 0.f : dv[1][r]);
    }
  }
}

static int mha_check(const void* q, const void* k, const void* v, int B, int H, int Lq, int Lk, int E) {
  MI_REQUIRE(q && k && v, "mha: null");
  MI_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && E == H * MHA_D, "mha: E %d must be H*%d (H %d)", E, MHA_D, H);
  MI_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0, "mha: alignment");
  MI_REQUIRE((long long)B * H < 65536, "mha: B*H too large");
  MI_REQUIRE((long long)Lq * Lk < (1LL << 32), "mha: Lq * Lk must stay below 2^32 (32-bit score offsets inside a head)");
  return MI_OK;
}

// waves (16 rows each) per block of the v2 kernels: one round of blocks on the device, 4 .. 12 waves
static int mha2_waves(int L, int BH) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  const int gq = mi_cdiv(L, 16);
  const long groups = (long)gq * BH;
  int nwv = (int)((groups + ncu - 1) / ncu);
  nwv = nwv < 4 ? 4 : (nwv > 12 ? 12 : nwv);
  if (nwv > gq) nwv = gq < 4 ? 4 : gq;
  return nwv;
}

extern const unsigned long long* g_mi_seed_off;   // runtime.hip
static int mha_set_dropout(MhaK* p, float drop_p, unsigned long long seed) {
  MI_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha: dropout p %f", drop_p);
  p->seed = seed;
  p->seed_off = g_mi_seed_off;
  p->drop_thr = drop_p > 0.f ? (unsigned)((double)drop_p * 4294967296.0) : 0u;
  if (drop_p > 0.f && p->drop_thr == 0u) p->drop_thr = 1u;
  p->drop_scale = 1.f / (1.f - drop_p);
  return MI_OK;
}

extern "C" int mi_mha_fwd(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, void* o,
                          float* lse, int B, int H, int Lq, int Lk, int E, float scale, mi_stream_t st) {
  return mi_mha_fwd_dropout(q, k, v, key_padding_mask, o, lse, B, H, Lq, Lk, E, scale, 0.f, 0ull, st);
}

extern "C" int mi_mha_fwd_dropout(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, void* o,
                                  float* lse, int B, int H, int Lq, int Lk, int E, float scale, float drop_p,
                                  uint64_t seed, mi_stream_t st) {
  return mi_mha_fwd_dropout_o32(q, k, v, key_padding_mask, o, nullptr, lse, B, H, Lq, Lk, E, scale, drop_p, seed, st);
}

extern "C" int mi_mha_fwd_dropout_o32(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, void* o,
                                      float* o_f32, float* lse, int B, int H, int Lq, int Lk, int E, float scale,
                                      float drop_p, uint64_t seed, mi_stream_t st) {
  return mi_mha_fwd_dropout_ld(q, E, k, E, v, key_padding_mask, o, o_f32, lse, B, H, Lq, Lk, E, scale, drop_p, seed, st);
}

extern "C" int mi_mha_fwd_dropout_ld(const void* q, int ldq, const void* k, int ldk, const void* v,
                                     const uint8_t* key_padding_mask, void* o, float* o_f32, float* lse, int B, int H, int Lq,
                                     int Lk, int E, float scale, float drop_p, uint64_t seed, mi_stream_t st) {
  int rc = mha_check(q, k, v, B, H, Lq, Lk, E);
  if (rc) return rc;
  MI_REQUIRE(ldq >= E && ldk >= E && ldq % 8 == 0 && ldk % 8 == 0, "mha_fwd: row strides ldq %d / ldk %d (>= E, multiples of 8)", ldq, ldk);
  MI_REQUIRE(o, "mha_fwd: null output");
  MhaK p;
  memset(&p, 0, sizeof(p));
  rc = mha_set_dropout(&p, drop_p, seed);
  if (rc) return rc;
  p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.mask = key_padding_mask;
  p.o = (__bf16*)o; p.o32 = o_f32; p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.E = E; p.scale = scale;
  p.ldq = ldq; p.ldk = ldk;
  MI_REQUIRE(!o_f32 || ((uintptr_t)o_f32 & 15) == 0, "mha_fwd: o_f32 alignment");
  static const int v2 = getenv("MI_MHA_V2") ? atoi(getenv("MI_MHA_V2")) : 1;
  if (!v2) {
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(mi_cdiv(Lq, 64), B * H), dim3(256), 0, (hipStream_t)st, p);
    MI_CHECK_LAUNCH("mha_fwd");
    return MI_OK;
  }
  const int nwv = mha2_waves(Lq, B * H);
  const dim3 grid(mi_cdiv(mi_cdiv(Lq, 16), nwv), B * H), blk(nwv * 64);
  if (p.drop_thr) hipLaunchKernelGGL(mha_fwd2_kernel<true>, grid, blk, 0, (hipStream_t)st, p);
  else hipLaunchKernelGGL(mha_fwd2_kernel<false>, grid, blk, 0, (hipStream_t)st, p);
  MI_CHECK_LAUNCH("mha_fwd2");
  return MI_OK;
}

extern "C" int mi_mha_bwd(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask, const void* o,
                          const float* lse, const void* dout, float* delta_ws, void* dq, void* dk, void* dv, int B,
                          int H, int Lq, int Lk, int E, float scale, mi_stream_t st) {
  return mi_mha_bwd_dropout(q, k, v, key_padding_mask, o, lse, dout, delta_ws, dq, dk, dv, B, H, Lq, Lk, E, scale, 0.f,
                            0ull, st);
}

// the keep mask itself (uint8 [B][H][Lq][Lk]) - for tests of the dropout path against a reference that applies the
// same mask
__global__ __launch_bounds__(256) void mha_mask_kernel(const MhaK p, uint8_t* out, long long n) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= n) return;
  out[i] = mi_rng32(p.seed, (unsigned long long)i) >= p.drop_thr ? 1 : 0;
}
extern "C" int mi_mha_dropout_mask(uint8_t* out, int B, int H, int Lq, int Lk, float drop_p, uint64_t seed,
                                   mi_stream_t st) {
  MI_REQUIRE(out && B > 0 && H > 0 && Lq > 0 && Lk > 0, "mha_dropout_mask: args");
  MhaK p;
  memset(&p, 0, sizeof(p));
  int rc = mha_set_dropout(&p, drop_p, seed);
  if (rc) return rc;
  const long long n = (long long)B * H * Lq * Lk;
  hipLaunchKernelGGL(mha_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, p, out, n);
  MI_CHECK_LAUNCH("mha_dropout_mask");
  return MI_OK;
}

extern "C" int mi_mha_bwd_dropout(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask,
                                  const void* o, const float* lse, const void* dout, float* delta_ws, void* dq, void* dk,
                                  void* dv, int B, int H, int Lq, int Lk, int E, float scale, float drop_p, uint64_t seed,
                                  mi_stream_t st) {
  return mi_mha_bwd_dropout_o32(q, k, v, key_padding_mask, o, nullptr, lse, dout, delta_ws, dq, dk, dv, B, H, Lq, Lk, E, scale,
                                drop_p, seed, st);
}

extern "C" int mi_mha_bwd_dropout_o32(const void* q, const void* k, const void* v, const uint8_t* key_padding_mask,
                                      const void* o, const float* o_f32, const float* lse, const void* dout, float* delta_ws,
                                      void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int E, float scale,
                                      float drop_p, uint64_t seed, mi_stream_t st) {
  return mi_mha_bwd_dropout_ld(q, E, k, E, v, key_padding_mask, o, o_f32, lse, dout, delta_ws, dq, dk, dv, B, H, Lq, Lk, E, scale,
                               drop_p, seed, st);
}

extern "C" int mi_mha_bwd_dropout_ld(const void* q, int ldq, const void* k, int ldk, const void* v,
                                     const uint8_t* key_padding_mask, const void* o, const float* o_f32, const float* lse,
                                     const void* dout, float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk,
                                     int E, float scale, float drop_p, uint64_t seed, mi_stream_t st) {
  int rc = mha_check(q, k, v, B, H, Lq, Lk, E);
  if (rc) return rc;
  MI_REQUIRE(ldq >= E && ldk >= E && ldq % 8 == 0 && ldk % 8 == 0, "mha_bwd: row strides ldq %d / ldk %d (>= E, multiples of 8)", ldq, ldk);
  MI_REQUIRE(((uintptr_t)dq % 16) == 0 && ((uintptr_t)dk % 16) == 0, "mha_bwd: dq / dk alignment");
  MI_REQUIRE(o && lse && dout && delta_ws && dq && dk && dv, "mha_bwd: null");
  MhaK p;
  memset(&p, 0, sizeof(p));
  rc = mha_set_dropout(&p, drop_p, seed);
  if (rc) return rc;
  p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.mask = key_padding_mask;
  p.o = (__bf16*)o; p.o32 = (float*)o_f32; p.lse = (float*)lse; p.dout = (const __bf16*)dout; p.delta = delta_ws; p.dq = (__bf16*)dq;
  p.dk = (__bf16*)dk; p.dv = (__bf16*)dv; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.E = E; p.scale = scale;
  p.ldq = ldq; p.ldk = ldk;
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(mha_delta_kernel, dim3(mi_cdiv(B * H * Lq * 4, 256)), dim3(256), 0, s, p, delta_ws);
  MI_CHECK_LAUNCH("mha_delta");
  static const int v2 = getenv("MI_MHA_V2") ? atoi(getenv("MI_MHA_V2")) : 1;
  if (!v2) {
    hipLaunchKernelGGL(mha_bwd_dq_kernel, dim3(mi_cdiv(Lq, 64), B * H), dim3(256), 0, s, p);
    MI_CHECK_LAUNCH("mha_bwd_dq");
    hipLaunchKernelGGL(mha_bwd_dkv_kernel, dim3(mi_cdiv(Lk, 64), B * H), dim3(256), 0, s, p);
    MI_CHECK_LAUNCH("mha_bwd_dkv");
    return MI_OK;
  }
  const int nq = mha2_waves(Lq, B * H), nk = mha2_waves(Lk, B * H);
  const dim3 gq(mi_cdiv(mi_cdiv(Lq, 16), nq), B * H), gk(mi_cdiv(mi_cdiv(Lk, 16), nk), B * H);
  if (p.drop_thr) hipLaunchKernelGGL(mha_bwd_dq2_kernel<true>, gq, dim3(nq * 64), 0, s, p);
  else hipLaunchKernelGGL(mha_bwd_dq2_kernel<false>, gq, dim3(nq * 64), 0, s, p);
  MI_CHECK_LAUNCH("mha_bwd_dq2");
  if (p.drop_thr) hipLaunchKernelGGL(mha_bwd_dkv2_kernel<true>, gk, dim3(nk * 64), 0, s, p);
  else hipLaunchKernelGGL(mha_bwd_dkv2_kernel<false>, gk, dim3(nk * 64), 0, s, p);
  MI_CHECK_LAUNCH("mha_bwd_dkv2");
  return MI_OK;
}
