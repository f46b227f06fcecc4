// The batch-building half of the DETR / SparseInst steps on the device: what the reference does per image with a handful of
// torch calls each (normalise, zero-pad into ImageList.from_tensors' batch tensor; pad, resize and stack the ground-truth
// masks) as ONE launch per batch.  The eager prefix of a captured SparseInst step was 78 launches of ~5 us (0.4 ms of a
// 12.2 ms step, tools/host_step_probe.py); with these two it is ~12.
//
//   mi_normalize_pad_batch  yolov7/modeling/meta_arch/detr.py:273-278 (`self.normalizer(x["image"].to(self.device))`,
//                           `ImageList.from_tensors(images)`) and meta_arch/sparseinst.py:95-98 (`..., 32)`):
//                           dst[b][c][y][x] = (img_b[c][y][x] - mean[c]) / std[c] inside the image, 0 in the pad.
//   mi_mask_targets_batch   yolov7/utils/misc.py:148-170 (nested_masks_from_list: zero-pad every image's masks to the batch
//                           shape) + yolov7/modeling/loss/sparseinst_loss.py:149-151 / 326-328 (F.interpolate(..., size=
//                           prediction size, mode="bilinear", align_corners=False)) into the fixed-capacity layout of
//                           modeling/sparseinst.py PackedMaskTargets: fp32 rows [B * cap][P], their bf16 transpose [B][P][cap]
//                           (the matcher's operand) and the labels [B][cap].
//
// The per-image records travel as KERNEL ARGUMENTS (by value, <= MI_FEED_MAX_IMAGES images per launch): no job table in
// device memory, no copy to wait for, nothing for the host to keep alive.
// Arithmetic: the expressions of the torch kernels they replace, in their order, without fma contraction (Makefile:
// -ffp-contract=off) - (x - mean) / std with an IEEE division; upsample_bilinear2d's
// h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d) with src = scale * (dst + 0.5) - 0.5 clamped at 0.
#include "common.h"
#include <cstring>

struct FeedImgK {
  mi_image_job j[MI_FEED_MAX_IMAGES];
  float* dst;
  int B, Hp, Wp, b0;      // b0: batch index of j[0]
  float mean[3], std[3];
};

template <class T>
__device__ __forceinline__ float feed_ld(const void* p, int64_t i) {
  return (float)((const T*)p)[i];
}

// one thread = 4 consecutive x of one (image, channel, row): a 16-byte store; grid.y = image, grid.x over 3 * Hp * Wp / 4
__global__ __launch_bounds__(256) void normalize_pad_kernel(const FeedImgK p) {
  const int b = blockIdx.y;
  const mi_image_job jb = p.j[b];
  const int W4 = (p.Wp + 3) / 4;
  const bool vec = (p.Wp & 3) == 0;      // whole 16-byte stores (rows start 16-byte aligned); else element stores
  const int total = 3 * p.Hp * W4;
  float* const out = p.dst + (int64_t)(p.b0 + b) * 3 * p.Hp * p.Wp;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int x4 = idx % W4;
    const int r = idx / W4;
    const int y = r % p.Hp, c = r / p.Hp;
    float4 o = {0.f, 0.f, 0.f, 0.f};
    if (y < jb.h) {
      const int64_t base = ((int64_t)c * jb.h + y) * jb.w;
      const float m = p.mean[c], s = p.std[c];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = x4 * 4 + e;
        v[e] = 0.f;
        if (x < jb.w) {
          const float f = jb.dtype == 0 ? feed_ld<float>(jb.src, base + x) : feed_ld<uint8_t>(jb.src, base + x);
          v[e] = (f - m) / s;
        }
      }
      o = {v[0], v[1], v[2], v[3]};
    }
    float* const q = out + ((int64_t)c * p.Hp + y) * p.Wp + x4 * 4;
    if (vec) {
      *(float4*)q = o;
    } else {
      const float ov[4] = {o.x, o.y, o.z, o.w};
      for (int e = 0; e < 4 && x4 * 4 + e < p.Wp; ++e) q[e] = ov[e];
    }
  }
}

extern "C" int mi_normalize_pad_batch(const mi_image_job* jobs, int B, float* dst, int Hp, int Wp, const float* mean3,
                                      const float* std3, mi_stream_t st) {
  MI_REQUIRE(jobs && dst && mean3 && std3 && B >= 1 && Hp >= 1 && Wp >= 1, "normalize_pad_batch: args");
  MI_REQUIRE((Wp % 4 != 0 || ((uintptr_t)dst & 15) == 0) && (int64_t)3 * Hp * (Wp + 3) < (1LL << 31), "normalize_pad_batch: dst alignment / size");
  for (int b = 0; b < B; ++b)
    MI_REQUIRE(jobs[b].src && jobs[b].h >= 1 && jobs[b].w >= 1 && jobs[b].h <= Hp && jobs[b].w <= Wp &&
                   (jobs[b].dtype == 0 || jobs[b].dtype == 1),
               "normalize_pad_batch: image %d (%d x %d, dtype %d) does not fit %d x %d", b, jobs[b].h, jobs[b].w, jobs[b].dtype, Hp, Wp);
  for (int b0 = 0; b0 < B; b0 += MI_FEED_MAX_IMAGES) {
    FeedImgK k;
    memset(&k, 0, sizeof(k));
    const int nb = B - b0 < MI_FEED_MAX_IMAGES ? B - b0 : MI_FEED_MAX_IMAGES;
    for (int b = 0; b < nb; ++b) k.j[b] = jobs[b0 + b];
    k.dst = dst; k.B = nb; k.Hp = Hp; k.Wp = Wp; k.b0 = b0;
    for (int c = 0; c < 3; ++c) { k.mean[c] = mean3[c]; k.std[c] = std3[c]; }
    const int total = 3 * Hp * ((Wp + 3) / 4);
    int gx = (total + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(normalize_pad_kernel, dim3(gx, nb), dim3(256), 0, (hipStream_t)st, k);
    MI_CHECK_LAUNCH("normalize_pad_batch");
  }
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ packed mask targets
struct FeedMaskK {
  mi_mask_job j[MI_FEED_MAX_IMAGES];
  float* tgt;          // [B * cap][P]
  __bf16* tgtT;        // [B][P][cap] or NULL
  int64_t* labels;     // [B][cap] or NULL
  float* t2ws;         // [B][cap][nchunk] or NULL: sum of squares of the block's 64 pixels of every row
  int nchunk, pad_;
  int B, cap, Hi, Wi, Ho, Wo, b0;     // (Hi, Wi): the padded batch shape the masks are zero-extended to before the resize
  float sh, sw;
};

// upsample_bilinear2d's source index (align_corners = False): scale * (dst + 0.5) - 0.5, clamped at 0
__device__ __forceinline__ void feed_src(int o, float scale, int n, int* i0, int* ip, float* l1) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  const int a = (int)s;
  *i0 = a;
  *ip = a < n - 1 ? 1 : 0;
  *l1 = s - (float)a;
}

#define FEED_TP 64      // pixels per block
// block = (image b, FEED_TP consecutive output pixels); its 256 threads walk the cap rows four at a time: thread (pixel =
// t % 64, row lane = t / 64).  The fp32 rows leave as 256-byte runs per row; the bf16 values meet in LDS as the [pixel][cap]
// tile, which is one contiguous piece of tgtT and leaves as 16-byte stores.
__global__ __launch_bounds__(256) void mask_targets_kernel(const FeedMaskK p) {
  extern __shared__ __attribute__((aligned(16))) __bf16 tile[];      // [FEED_TP][cap]
  const int b = blockIdx.y;
  const mi_mask_job jb = p.j[b];
  const int P = p.Ho * p.Wo;
  const int p0 = blockIdx.x * FEED_TP;
  const int px = threadIdx.x % FEED_TP, rl = threadIdx.x / FEED_TP;
  const int pix = p0 + px;
  const bool live = pix < P;
  int y0 = 0, yp = 0, x0 = 0, xp = 0;
  float ly = 0.f, lx = 0.f;
  if (live) {
    feed_src(pix / p.Wo, p.sh, p.Hi, &y0, &yp, &ly);
    feed_src(pix % p.Wo, p.sw, p.Wi, &x0, &xp, &lx);
  }
  const float h1 = ly, h0 = 1.f - ly, w1 = lx, w0 = 1.f - lx;
  const int ya = y0, yb = y0 + yp, xa = x0, xb = x0 + xp;
  float* const trow = p.tgt + (int64_t)(p.b0 + b) * p.cap * P;
  for (int j = rl; j < p.cap; j += 256 / FEED_TP) {
    float v = 0.f;
    if (live && j < jb.M) {
      // the mask zero-extended to (Hi, Wi): taps outside (h, w) read 0
      const int64_t base = (int64_t)j * jb.h * jb.w;
      auto tap = [&](int y, int x) -> float {
        if (y >= jb.h || x >= jb.w) return 0.f;
        const int64_t i = base + (int64_t)y * jb.w + x;
        return jb.dtype == 0 ? feed_ld<float>(jb.masks, i) : (feed_ld<uint8_t>(jb.masks, i) != 0.f ? 1.f : 0.f);
      };
      const float a = tap(ya, xa), bb = tap(ya, xb), c = tap(yb, xa), d = tap(yb, xb);
      v = h0 * (w0 * a + w1 * bb) + h1 * (w0 * c + w1 * d);
    }
    if (live) trow[(int64_t)j * P + pix] = v;
    tile[px * p.cap + j] = (__bf16)v;
    if (p.t2ws) {       // (a wave = the 64 pixels of one row: a fixed-order butterfly; pixels past the map contribute 0)
      float q = v * v;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
      if (px == 0) p.t2ws[((int64_t)(p.b0 + b) * p.cap + j) * p.nchunk + blockIdx.x] = q;
    }
  }
  if (p.labels && blockIdx.x == 0) {
    for (int j = threadIdx.x; j < p.cap; j += 256)
      p.labels[(int64_t)(p.b0 + b) * p.cap + j] = (j < jb.M && jb.labels) ? jb.labels[j] : 0;
  }
  if (!p.tgtT) return;
  __syncthreads();
  const int npx = P - p0 < FEED_TP ? P - p0 : FEED_TP;
  const int n16 = npx * p.cap / 8;                 // cap % 8 == 0: whole 16-byte pieces
  uint4* const dst = (uint4*)(p.tgtT + ((int64_t)(p.b0 + b) * P + p0) * p.cap);
  for (int i = threadIdx.x; i < n16; i += 256) dst[i] = ((const uint4*)tile)[i];
}

__global__ __launch_bounds__(256) void mask_targets_t2_kernel(const float* __restrict__ ws, float* t2, int rows, int nchunk) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += ws[(int64_t)r * nchunk + c];
  t2[r] = s;
}

extern "C" int mi_mask_targets_batch(const mi_mask_job* jobs, int B, int cap, int Hi, int Wi, int Ho, int Wo, float* tgt,
                                     void* tgtT_bf16, int64_t* labels, float* t2, float* t2_ws, mi_stream_t st) {
  MI_REQUIRE(!t2 || t2_ws, "mask_targets_batch: t2 needs its workspace");
  MI_REQUIRE(jobs && tgt && B >= 1 && cap >= 8 && cap % 8 == 0 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1, "mask_targets_batch: args");
  MI_REQUIRE((size_t)FEED_TP * cap * 2 <= 64 * 1024, "mask_targets_batch: capacity %d exceeds the LDS tile", cap);
  MI_REQUIRE(!tgtT_bf16 || ((uintptr_t)tgtT_bf16 & 15) == 0, "mask_targets_batch: tgtT alignment");
  for (int b = 0; b < B; ++b)
    MI_REQUIRE(jobs[b].M >= 0 && jobs[b].M <= cap && (jobs[b].M == 0 || (jobs[b].masks && jobs[b].h >= 1 && jobs[b].w >= 1 &&
                                                                         jobs[b].h <= Hi && jobs[b].w <= Wi)) &&
                   (jobs[b].dtype == 0 || jobs[b].dtype == 1),
               "mask_targets_batch: image %d (%d masks of %d x %d, dtype %d; capacity %d, batch shape %d x %d)", b, jobs[b].M,
               jobs[b].h, jobs[b].w, jobs[b].dtype, cap, Hi, Wi);
  const int P = Ho * Wo;
  for (int b0 = 0; b0 < B; b0 += MI_FEED_MAX_IMAGES) {
    FeedMaskK k;
    memset(&k, 0, sizeof(k));
    const int nb = B - b0 < MI_FEED_MAX_IMAGES ? B - b0 : MI_FEED_MAX_IMAGES;
    for (int b = 0; b < nb; ++b) k.j[b] = jobs[b0 + b];
    k.tgt = tgt; k.tgtT = (__bf16*)tgtT_bf16; k.labels = labels;
    k.B = nb; k.cap = cap; k.Hi = Hi; k.Wi = Wi; k.Ho = Ho; k.Wo = Wo; k.b0 = b0;
    k.t2ws = t2 ? t2_ws : nullptr; k.nchunk = (P + FEED_TP - 1) / FEED_TP;
    k.sh = (float)Hi / (float)Ho; k.sw = (float)Wi / (float)Wo;        // area_pixel_compute_scale, no scale_factor given
    hipLaunchKernelGGL(mask_targets_kernel, dim3((P + FEED_TP - 1) / FEED_TP, nb), dim3(256), (size_t)FEED_TP * cap * 2,
                       (hipStream_t)st, k);
    MI_CHECK_LAUNCH("mask_targets_batch");
  }
  if (t2) {
    hipLaunchKernelGGL(mask_targets_t2_kernel, dim3((B * cap + 255) / 256), dim3(256), 0, (hipStream_t)st, (const float*)t2_ws, t2,
                       B * cap, (P + FEED_TP - 1) / FEED_TP);
    MI_CHECK_LAUNCH("mask_targets_batch (t2)");
  }
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ padding masks
// MaskedBackbone.mask_out_padding (yolov7/modeling/meta_arch/detr.py:385-403): per feature level a [B][H][W] mask that is 0
// on the ceil(h / stride) x ceil(w / stride) corner the image covers and 1 on the padding - here from the DEVICE copy of the
// image sizes (int64 [B][2] = (h, w)), all levels in one launch (the torch spelling was nine calls per level).
struct FeedPadK {
  uint8_t* out[MI_FEED_MAX_LEVELS];
  int H[MI_FEED_MAX_LEVELS], W[MI_FEED_MAX_LEVELS], stride[MI_FEED_MAX_LEVELS];
  const int64_t* sizes;
  int B, nlev;
};
__global__ __launch_bounds__(256) void padding_masks_kernel(const FeedPadK p) {
  const int lv = blockIdx.y;
  const int H = p.H[lv], W = p.W[lv], st = p.stride[lv];
  const int total = p.B * H * W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int x = i % W, r = i / W;
    const int y = r % H, b = r / H;
    const int64_t hv = (p.sizes[b * 2] + (st - 1)) / st, wv = (p.sizes[b * 2 + 1] + (st - 1)) / st;      // ceil(size / stride)
    p.out[lv][i] = (y >= hv || x >= wv) ? 1 : 0;
  }
}
extern "C" int mi_padding_masks(const int64_t* sizes_dev, int B, int nlev, void* const* out, const int* H, const int* W,
                                const int* stride, mi_stream_t st) {
  MI_REQUIRE(sizes_dev && out && H && W && stride && B >= 1 && nlev >= 1 && nlev <= MI_FEED_MAX_LEVELS, "padding_masks: args");
  FeedPadK k;
  memset(&k, 0, sizeof(k));
  int mx = 0;
  for (int l = 0; l < nlev; ++l) {
    MI_REQUIRE(out[l] && H[l] >= 1 && W[l] >= 1 && stride[l] >= 1, "padding_masks: level %d", l);
    k.out[l] = (uint8_t*)out[l]; k.H[l] = H[l]; k.W[l] = W[l]; k.stride[l] = stride[l];
    if (B * H[l] * W[l] > mx) mx = B * H[l] * W[l];
  }
  k.sizes = sizes_dev; k.B = B; k.nlev = nlev;
  int gx = (mx + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(padding_masks_kernel, dim3(gx, nlev), dim3(256), 0, (hipStream_t)st, k);
  MI_CHECK_LAUNCH("padding_masks");
  return MI_OK;
}
