// IoU-family box regression losses with gradient: CIoU / DIoU / GIoU / SIoU / plain IoU.
// replaces IOUlossV6.__call__ (yolov7/utils/boxes.py:666-752; used by the YOLOv6 head, head/yolov6_head.py:346,512)
// and its autograd backward: loss[n] = 1 - iou_variant(pred[n], target[n]), eps 1e-7, CIoU's alpha under no_grad.
// The gradient with respect to the prediction is carried along the forward expression as forward-mode dual numbers
// (value + 4 partials), so forward and backward are ONE pass over the boxes and cannot drift apart.
#include "common.h"
#include "dual4.h"
#include "iou_v6.h"

__global__ __launch_bounds__(256) void iou_loss_v6_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                          int n, int type, int xyxy, float eps, const float* dloss,
                                                          float* loss, float* dpred) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const D4 iou = iou_v6_dual(pred + (size_t)i * 4, tgt + (size_t)i * 4, type, xyxy, eps);
  if (loss) loss[i] = 1.0f - iou.v;
  if (dpred) {
    const float gl = dloss ? dloss[i] : 1.f;
    for (int c = 0; c < 4; ++c) dpred[(size_t)i * 4 + c] = -iou.d[c] * gl;
  }
}

extern "C" int mi_iou_loss_v6(const float* pred, const float* target, int n, int iou_type, int box_xyxy, float eps,
                              const float* dloss, float* loss, float* dpred, mi_stream_t st) {
  MI_REQUIRE(pred && target && (loss || dpred) && n >= 0, "iou_loss_v6: args");
  MI_REQUIRE(iou_type >= IOU_PLAIN && iou_type <= IOU_SIOU, "iou_loss_v6: type %d", iou_type);
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(iou_loss_v6_kernel, dim3(mi_cdiv(n, 256)), dim3(256), 0, (hipStream_t)st, pred, target, n, iou_type,
                     box_xyxy, eps, dloss, loss, dpred);
  MI_CHECK_LAUNCH("iou_loss_v6");
  return MI_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// IOUloss of the YOLOX head as a standalone op (yolov7/utils/boxes.py:125-168): boxes (cx, cy, w, h);
// loss_type 0 "iou": 1 - iou^2, 1 "giou": 1 - clamp(giou, -1, 1).  torch.max / torch.min of two tensors split the
// gradient evenly at a tie (ATen maximum / minimum backward), which these two helpers reproduce.
__device__ __forceinline__ D4 dmax_tie(D4 a, D4 b) {
  if (a.v > b.v) return a;
  if (a.v < b.v) return b;
  return (a + b) * 0.5f;
}
__device__ __forceinline__ D4 dmin_tie(D4 a, D4 b) {
  if (a.v < b.v) return a;
  if (a.v > b.v) return b;
  return (a + b) * 0.5f;
}
__global__ __launch_bounds__(256) void yolox_iou_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                             int n, int type, const float* dloss, float* loss,
                                                             float* dpred) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* pb = pred + (size_t)i * 4;
  const float* tb = tgt + (size_t)i * 4;
  const D4 x = dvar(pb[0], 0), y = dvar(pb[1], 1), w = dvar(pb[2], 2), h = dvar(pb[3], 3);
  const D4 px1 = x - w * 0.5f, px2 = x + w * 0.5f, py1 = y - h * 0.5f, py2 = y + h * 0.5f;
  const D4 gx1 = dconst(tb[0] - tb[2] / 2), gx2 = dconst(tb[0] + tb[2] / 2);
  const D4 gy1 = dconst(tb[1] - tb[3] / 2), gy2 = dconst(tb[1] + tb[3] / 2);
  const D4 tlx = dmax_tie(px1, gx1), tly = dmax_tie(py1, gy1), brx = dmin_tie(px2, gx2), bry = dmin_tie(py2, gy2);
  const float en = (tlx.v < brx.v ? 1.f : 0.f) * (tly.v < bry.v ? 1.f : 0.f);
  const D4 area_i = (brx - tlx) * (bry - tly) * en;
  const D4 area_p = w * h;
  const float area_g = tb[2] * tb[3];
  const D4 iou = area_i / (area_p + area_g - area_i + 1e-16f);
  D4 l;
  if (type == 0) {
    l = dconst(1.f) - dsqr(iou);
  } else {
    const D4 cx1 = dmin_tie(px1, gx1), cy1 = dmin_tie(py1, gy1), cx2 = dmax_tie(px2, gx2), cy2 = dmax_tie(py2, gy2);
    const D4 area_c = (cx2 - cx1) * (cy2 - cy1);
    const D4 den = area_c.v >= 1e-16f ? area_c : dconst(1e-16f);   // .clamp(1e-16): gradient passes on [min, inf)
    D4 giou = iou - (area_c - area_i) / den;
    if (giou.v < -1.f) giou = dconst(-1.f);                         // .clamp(-1, 1): zero gradient outside
    else if (giou.v > 1.f) giou = dconst(1.f);
    l = dconst(1.f) - giou;
  }
  if (loss) loss[i] = l.v;
  if (dpred) {
    const float gl = dloss ? dloss[i] : 1.f;
    for (int c = 0; c < 4; ++c) dpred[(size_t)i * 4 + c] = l.d[c] * gl;
  }
}
extern "C" int mi_yolox_iou_loss(const float* pred, const float* target, int n, int loss_type, const float* dloss,
                                 float* loss, float* dpred, mi_stream_t st) {
  MI_REQUIRE(pred && target && (loss || dpred) && n >= 0, "yolox_iou_loss: args");
  MI_REQUIRE(loss_type == 0 || loss_type == 1, "yolox_iou_loss: loss_type %d (0 iou, 1 giou)", loss_type);
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(yolox_iou_loss_kernel, dim3(mi_cdiv(n, 256)), dim3(256), 0, (hipStream_t)st, pred, target, n,
                     loss_type, dloss, loss, dpred);
  MI_CHECK_LAUNCH("yolox_iou_loss");
  return MI_OK;
}

// pairwise IoU matrix: bboxes_iou (utils/boxes.py:57-81) and pairwise_bbox_iou (utils/boxes.py:755-779, the YOLOv6
// head's SimOTA cost, yolov6_head.py).  out[i][j] = inter / (area1[i] + area2[j] - inter), inter zeroed unless lt < rb
// in both axes; no epsilon (as the reference: 0/0 -> nan for two empty boxes).
__global__ __launch_bounds__(256) void pairwise_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                           int N, int M, int xyxy, float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= M) return;
  const float* a = b1 + (size_t)i * 4;
  const float* b = b2 + (size_t)j * 4;
  float ax1, ay1, ax2, ay2, bx1, by1, bx2, by2, area1, area2;
  if (xyxy) {
    ax1 = a[0]; ay1 = a[1]; ax2 = a[2]; ay2 = a[3];
    bx1 = b[0]; by1 = b[1]; bx2 = b[2]; by2 = b[3];
    area1 = (a[2] - a[0]) * (a[3] - a[1]);
    area2 = (b[2] - b[0]) * (b[3] - b[1]);
  } else {
    ax1 = a[0] - a[2] / 2; ax2 = a[0] + a[2] / 2; ay1 = a[1] - a[3] / 2; ay2 = a[1] + a[3] / 2;
    bx1 = b[0] - b[2] / 2; bx2 = b[0] + b[2] / 2; by1 = b[1] - b[3] / 2; by2 = b[1] + b[3] / 2;
    area1 = a[2] * a[3];
    area2 = b[2] * b[3];
  }
  const float ltx = fmaxf(ax1, bx1), lty = fmaxf(ay1, by1), rbx = fminf(ax2, bx2), rby = fminf(ay2, by2);
  const float valid = (ltx < rbx ? 1.f : 0.f) * (lty < rby ? 1.f : 0.f);
  const float inter = ((rbx - ltx) * (rby - lty)) * valid;
  out[(size_t)i * M + j] = inter / (area1 + area2 - inter);
}
extern "C" int mi_pairwise_bbox_iou(const float* box1, const float* box2, int N, int M, int box_xyxy, float* out,
                                    mi_stream_t st) {
  MI_REQUIRE(N >= 0 && M >= 0 && N <= 65535, "pairwise_bbox_iou: sizes (N <= 65535)");
  if (N == 0 || M == 0) return MI_OK;   // an empty matrix: nothing to do (the pointers of empty tensors are NULL)
  MI_REQUIRE(box1 && box2 && out, "pairwise_bbox_iou: null");
  hipLaunchKernelGGL(pairwise_iou_kernel, dim3(mi_cdiv(M, 256), N), dim3(256), 0, (hipStream_t)st, box1, box2, N, M,
                     box_xyxy, out);
  MI_CHECK_LAUNCH("pairwise_bbox_iou");
  return MI_OK;
}
