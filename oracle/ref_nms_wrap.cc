// TEST INFRASTRUCTURE (oracle/): C entry points around the reference's OWN greedy NMS, compiled from where it lies.
//
// The only native code of the reference on this path is deploy/trt_cc/demo_yolox.cc:53-135 (struct Object,
// intersection_area, qsort_descent_inplace, nms_sorted_bboxes).  The file as a whole needs TensorRT / CUDA / OpenCV, so
// oracle/Makefile cuts exactly that span out of /root/reference into oracle/_ref/demo_yolox_nms.inc (a build output,
// git-ignored, never committed) and this wrapper #includes it over a stand-in for the one OpenCV type it uses.
// cv::Rect_<float> is RESTATED from OpenCV's published behaviour (x, y, width, height; area() = width * height;
// operator& = intersection, an empty rectangle when the operands do not overlap) - OpenCV is an un-vendored, un-pinned
// dependency of the reference.  The NMS loop itself (sort order, IoU expression, `>` threshold test) is the
// reference's code, not a restatement: it pins oracle/yolox_oracle.py::nms and mi_batched_nms on tie-free inputs.
#include <algorithm>
#include <vector>

namespace cv {
template <typename T>
struct Rect_ {
  T x, y, width, height;
  Rect_() : x(0), y(0), width(0), height(0) {}
  Rect_(T x_, T y_, T w_, T h_) : x(x_), y(y_), width(w_), height(h_) {}
  T area() const { return width * height; }
  bool empty() const { return width <= 0 || height <= 0; }
};
// modules/core/include/opencv2/core/types.hpp: Rect_ operator &
template <typename T>
static inline Rect_<T> operator&(const Rect_<T>& a, const Rect_<T>& b) {
  if (a.empty() || b.empty()) return Rect_<T>();
  const T x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
  const T w = std::min(a.x + a.width, b.x + b.width) - x1, h = std::min(a.y + a.height, b.y + b.height) - y1;
  if (w <= 0 || h <= 0) return Rect_<T>();
  return Rect_<T>(x1, y1, w, h);
}
}  // namespace cv

#include "_ref/demo_yolox_nms.inc"   // struct Object ... nms_sorted_bboxes, verbatim from the reference tree

// boxes: n x (x, y, w, h); returns the number of kept boxes; kept[] = their indices into the INPUT arrays in the order
// the reference picks them (descending score)
extern "C" int ref_nms_sorted_bboxes(const float* boxes, const float* scores, int n, float thr, int* kept) {
  std::vector<Object> objs(n);
  for (int i = 0; i < n; ++i) {
    objs[i].rect = cv::Rect_<float>(boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]);
    objs[i].label = i;   // carries the input index through the reference's sort
    objs[i].prob = scores[i];
  }
  qsort_descent_inplace(objs);
  std::vector<int> picked;
  nms_sorted_bboxes(objs, picked, thr);
  for (size_t k = 0; k < picked.size(); ++k) kept[k] = objs[picked[k]].label;
  return (int)picked.size();
}
