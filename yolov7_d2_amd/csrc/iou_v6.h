// IOUlossV6's IoU variants as forward-mode dual numbers (value + d/d(pred x, y, w, h)): shared by the standalone loss
// kernel (iou_loss.hip) and the YOLOv6 form of the SimOTA loss kernels (yolox_loss.hip).
// yolov7/utils/boxes.py:666-752: eps 1e-7, CIoU's alpha under no_grad.
#pragma once
#include "dual4.h"

enum { IOU_PLAIN = 0, IOU_GIOU = 1, IOU_DIOU = 2, IOU_CIOU = 3, IOU_SIOU = 4 };

// pb: prediction, tb: target; (cx,cy,w,h) or (x1,y1,x2,y2) when xyxy.  Returns the IoU VARIANT (the loss is 1 - it).
__device__ __forceinline__ D4 iou_v6_dual(const float* pb, const float* tb, int type, int xyxy, float eps) {
  D4 b1x1, b1y1, b1x2, b1y2;
  float b2x1, b2y1, b2x2, b2y2;
  if (xyxy) {
    b1x1 = dvar(pb[0], 0); b1y1 = dvar(pb[1], 1); b1x2 = dvar(pb[2], 2); b1y2 = dvar(pb[3], 3);
    b2x1 = tb[0]; b2y1 = tb[1]; b2x2 = tb[2]; b2y2 = tb[3];
  } else {
    const D4 x = dvar(pb[0], 0), y = dvar(pb[1], 1), w = dvar(pb[2], 2), h = dvar(pb[3], 3);
    b1x1 = x - w * 0.5f; b1x2 = x + w * 0.5f; b1y1 = y - h * 0.5f; b1y2 = y + h * 0.5f;
    b2x1 = tb[0] - tb[2] / 2; b2x2 = tb[0] + tb[2] / 2; b2y1 = tb[1] - tb[3] / 2; b2y2 = tb[1] + tb[3] / 2;
  }
  const D4 c2x1 = dconst(b2x1), c2y1 = dconst(b2y1), c2x2 = dconst(b2x2), c2y2 = dconst(b2y2);
  const D4 inter = dclamp0(dmin(b1x2, c2x2) - dmax(b1x1, c2x1)) * dclamp0(dmin(b1y2, c2y2) - dmax(b1y1, c2y1));
  const D4 w1 = b1x2 - b1x1, h1 = b1y2 - b1y1 + eps;
  const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + eps;
  const D4 uni = w1 * h1 + (w2 * h2) - inter + eps;
  D4 iou = inter / uni;
  const D4 cw = dmax(b1x2, c2x2) - dmin(b1x1, c2x1), ch = dmax(b1y2, c2y2) - dmin(b1y1, c2y1);
  if (type == IOU_GIOU) {
    const D4 carea = cw * ch + eps;
    iou = iou - (carea - uni) / carea;
  } else if (type == IOU_DIOU || type == IOU_CIOU) {
    const D4 c2 = dsqr(cw) + dsqr(ch) + eps;
    const D4 rho2 = (dsqr(dconst(b2x1 + b2x2) - b1x1 - b1x2) + dsqr(dconst(b2y1 + b2y2) - b1y1 - b1y2)) * 0.25f;
    if (type == IOU_DIOU) {
      iou = iou - rho2 / c2;
    } else {
      const float k = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
      const D4 v = dsqr(dconst(atanf(w2 / h2)) - datan(w1 / h1)) * k;
      const float alpha = v.v / (v.v - iou.v + (1.f + eps));  // torch.no_grad(): a constant for the gradient
      iou = iou - (rho2 / c2 + v * alpha);
    }
  } else if (type == IOU_SIOU) {
    const D4 scw = (dconst(b2x1 + b2x2) - b1x1 - b1x2) * 0.5f, sch = (dconst(b2y1 + b2y2) - b1y1 - b1y2) * 0.5f;
    const D4 sigma = dsqrt(dsqr(scw) + dsqr(sch));
    const D4 sa1 = dabs(scw) / sigma, sa2 = dabs(sch) / sigma;
    const D4 sa = sa1.v > 0.70710678118654752f ? sa2 : sa1;
    // cos(2 asin(s) - pi/2) = sin(2 asin s) = 2 s sqrt(1 - s^2);  d/ds = cos(2 asin s) * 2 / sqrt(1 - s^2)
    const float as = asinf(sa.v);
    const float om = sqrtf(fmaxf(1.f - sa.v * sa.v, 1e-30f));
    const D4 angle = dchain(sa, cosf(as * 2.f - 1.57079632679489661923f), cosf(2.f * as) * 2.f / om);
    const D4 rx = dsqr(scw / cw), ry = dsqr(sch / ch);
    const D4 gamma = angle - 2.f;
    const D4 dist = dconst(2.f) - dexp(gamma * rx) - dexp(gamma * ry);
    const D4 cw2 = dconst(w2), chh2 = dconst(h2);
    const D4 ow = dabs(w1 - cw2) / dmax(w1, cw2), oh = dabs(h1 - chh2) / dmax(h1, chh2);
    const D4 shape = dpow4(dconst(1.f) - dexp(ow * -1.f)) + dpow4(dconst(1.f) - dexp(oh * -1.f));
    iou = iou - (dist + shape) * 0.5f;
  }
  return iou;
}
