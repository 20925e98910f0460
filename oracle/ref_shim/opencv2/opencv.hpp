/*
 * oracle/ref_shim/opencv2/opencv.hpp -- TEST-ONLY stand-in for OpenCV (never shipped).
 *
 * OpenCV is not installed in this image, so the reference's VideoFrameTransform.cpp cannot be
 * built as its authors build it.  This header provides exactly the slice of the cv:: API that
 * file touches, so that the reference's OWN projection, filter-configuration and frame
 * orchestration code compiles unmodified from /root/reference into oracle/_ref/ and can be
 * used to pin the oracle restatement (see oracle/Makefile, oracle/ref_probe.cpp).
 *
 *   cv::Mat               a ref-counted 2-D array with ROI views that remember their parent
 *                         (what cv::Mat::locateROI reports), enough for at<T>(), zeros(),
 *                         setTo(), operator()(Rect) and operator/=(double)
 *   cv::remap / cv::sepFilter2D
 *                         bound to the oracle's restatement of OpenCV's arithmetic
 *                         (t360_oracle_cv.c) -- so oracle/_ref pins the ORCHESTRATION around
 *                         those calls, not OpenCV's arithmetic itself (PARITY UNPINNED there)
 *   cv::resize            INTER_AREA shrink, bound to the oracle's restatement as well
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <string>

/* The two oracle entry points the shim forwards to (declared here rather than through
 * t360_oracle.h, whose copy of the ABI enums would collide with the reference's own header). */
extern "C" {
void t360o_remap_rows(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                      size_t dstep, const float* map, int interp, int borderType, int row0, int row1);
int t360o_sepfilter_roi(const uint8_t* parent, int pw, int ph, size_t pstep, uint8_t* dparent,
                        size_t dstep, int left, int top, int width, int height, const float* kx,
                        int kx_len, const float* ky, int ky_len);
int t360o_resize_area(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                      size_t dstep);
}

#define CV_8U 0
#define CV_32F 5
#define CV_32FC2 13

namespace cv {

enum BorderTypes {
  BORDER_CONSTANT = 0,
  BORDER_REPLICATE = 1,
  BORDER_REFLECT = 2,
  BORDER_WRAP = 3,
  BORDER_REFLECT_101 = 4,
  BORDER_TRANSPARENT = 5
};
enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3, INTER_LANCZOS4 = 4 };

class Exception : public std::exception {
 public:
  explicit Exception(const std::string& m) : msg(m) {}
  const char* what() const noexcept override { return msg.c_str(); }
  std::string msg;
};

struct Point {
  Point() : x(0), y(0) {}
  Point(int x_, int y_) : x(x_), y(y_) {}
  int x, y;
};
struct Point2f {
  Point2f() : x(0), y(0) {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
  float x, y;
};
struct Size {
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
  int width, height;
};
struct Rect {
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
  int x, y, width, height;
};
struct Scalar {
  Scalar() : v(0) {}
  Scalar(double v_) : v(v_) {}
  double v;
};

inline size_t elemSizeOf(int type) {
  switch (type) {
    case CV_8U: return 1;
    case CV_32F: return 4;
    case CV_32FC2: return 8;
    default: throw Exception("shim: unsupported Mat type");
  }
}

class Mat {
 public:
  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  Mat(Size s, int t) { create(s.height, s.width, t); }
  Mat(Size s, int t, const Scalar& v) {
    create(s.height, s.width, t);
    setTo(v);
  }
  /* user-owned buffer, row stride in bytes */
  Mat(int r, int c, int t, void* d, size_t stepBytes)
      : rows(r), cols(c), step(stepBytes), data((uint8_t*)d), type_(t), wholeRows(r), wholeCols(c),
        datastart((uint8_t*)d) {}

  static Mat zeros(int r, int c, int t) {
    Mat m(r, c, t);
    std::memset(m.data, 0, m.step * (size_t)r);
    return m;
  }
  static Mat zeros(Size s, int t) { return zeros(s.height, s.width, t); }

  template <typename T>
  T& at(int i, int j) {
    return *(T*)(data + (size_t)i * step + (size_t)j * sizeof(T));
  }
  template <typename T>
  const T& at(int i, int j) const {
    return *(const T*)(data + (size_t)i * step + (size_t)j * sizeof(T));
  }

  /* ROI view; OpenCV asserts the rectangle lies inside the matrix */
  Mat operator()(const Rect& r) const {
    if (!(0 <= r.x && 0 <= r.width && r.x + r.width <= cols && 0 <= r.y && 0 <= r.height &&
          r.y + r.height <= rows))
      throw Exception("(-215:Assertion failed) 0 <= roi.x && 0 <= roi.width && roi.x + roi.width <= m.cols && 0 <= roi.y && 0 <= roi.height && roi.y + roi.height <= m.rows");
    Mat m(*this);
    m.data = data + (size_t)r.y * step + (size_t)r.x * elemSizeOf(type_);
    m.rows = r.height;
    m.cols = r.width;
    return m;
  }

  /* cv::Mat::operator/=(double) is convertTo(*this, -1, 1./s): for CV_32F data the scale is
   * applied as a float multiply [OpenCV mat.inl.hpp / convert_scale, from memory] */
  Mat& operator/=(double s) {
    if (type_ != CV_32F) throw Exception("shim: operator/= only for CV_32F");
    const float a = (float)(1.0 / s);
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < cols; j++) at<float>(i, j) = at<float>(i, j) * a;
    return *this;
  }

  Mat& setTo(const Scalar& s) {
    if (type_ != CV_8U) throw Exception("shim: setTo only for CV_8U");
    for (int i = 0; i < rows; i++) std::memset(data + (size_t)i * step, (int)s.v, (size_t)cols);
    return *this;
  }

  Size size() const { return Size(cols, rows); }
  int type() const { return type_; }
  bool empty() const { return data == nullptr; }
  /* what cv::Mat::locateROI reports */
  void locateROI(Size& whole, Point& ofs) const {
    size_t delta = (size_t)(data - datastart);
    ofs.y = step ? (int)(delta / step) : 0;
    ofs.x = step ? (int)((delta - (size_t)ofs.y * step) / elemSizeOf(type_)) : 0;
    whole = Size(wholeCols, wholeRows);
  }

  int rows = 0, cols = 0;
  size_t step = 0;
  uint8_t* data = nullptr;

 private:
  void create(int r, int c, int t) {
    rows = wholeRows = r;
    cols = wholeCols = c;
    type_ = t;
    step = (size_t)c * elemSizeOf(t);
    owner.reset(new uint8_t[step * (size_t)(r > 0 ? r : 0) + 1], std::default_delete<uint8_t[]>());
    data = datastart = owner.get();
  }
  int type_ = CV_8U;
  int wholeRows = 0, wholeCols = 0;
  uint8_t* datastart = nullptr;
  std::shared_ptr<uint8_t> owner;

  friend void remap(const Mat&, Mat&, const Mat&, const Mat&, int, int);
  friend void sepFilter2D(const Mat&, Mat&, int, const Mat&, const Mat&, Point, double, int);
};

/* cv::remap(src, dst, map1(CV_32FC2), map2(empty), interpolation, borderMode) */
inline void remap(const Mat& src, Mat& dst, const Mat& map1, const Mat& map2, int interpolation,
                  int borderMode) {
  (void)map2;
  if (src.type() != CV_8U || dst.type() != CV_8U || map1.type() != CV_32FC2)
    throw Exception("shim: remap expects CV_8U images and a CV_32FC2 map");
  if (dst.rows != map1.rows || dst.cols != map1.cols) throw Exception("shim: dst/map size mismatch");
  if (map1.step != (size_t)map1.cols * 8) throw Exception("shim: map must be continuous");
  if (interpolation == INTER_AREA) interpolation = INTER_LINEAR; /* imgwarp.cpp: cv::remap */
  if (!(interpolation == 0 || interpolation == 1 || interpolation == 2 || interpolation == 4))
    throw Exception("Unknown interpolation method");
  t360o_remap_rows(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step,
                   (const float*)map1.data, interpolation, borderMode, 0, dst.rows);
}

/* cv::sepFilter2D without BORDER_ISOLATED: pixels outside the ROI come from the parent */
inline void sepFilter2D(const Mat& src, Mat& dst, int ddepth, const Mat& kernelX, const Mat& kernelY,
                        Point anchor, double delta, int borderType) {
  if (ddepth != -1 || anchor.x != -1 || anchor.y != -1 || delta != 0 || borderType != BORDER_REPLICATE)
    throw Exception("shim: sepFilter2D argument combination not used by the reference");
  if (src.type() != CV_8U || dst.type() != CV_8U || kernelX.type() != CV_32F || kernelY.type() != CV_32F)
    throw Exception("shim: sepFilter2D types");
  Size swhole, dwhole;
  Point sofs, dofs;
  src.locateROI(swhole, sofs);
  dst.locateROI(dwhole, dofs);
  if (sofs.x != dofs.x || sofs.y != dofs.y) throw Exception("shim: src/dst ROI offsets differ");
  t360o_sepfilter_roi(src.datastart, swhole.width, swhole.height, src.step, dst.datastart, dst.step,
                      sofs.x, sofs.y, src.cols, src.rows, (const float*)kernelX.data,
                      kernelX.rows * kernelX.cols, (const float*)kernelY.data,
                      kernelY.rows * kernelY.cols);
}

/* the reference calls resize(scaled, outputMat, Size(outputWidth, outputHeight), 0, 0, INTER_AREA)
 * with a preallocated CV_8U destination of that size (VideoFrameTransform.cpp:770-776) */
inline void resize(const Mat& src, Mat& dst, Size dsize, double fx, double fy, int interpolation) {
  if (interpolation != INTER_AREA || fx != 0 || fy != 0) throw Exception("shim: only resize(..., 0, 0, INTER_AREA)");
  if (src.type() != CV_8U || dst.type() != CV_8U) throw Exception("shim: resize only for CV_8U");
  if (dst.cols != dsize.width || dst.rows != dsize.height) throw Exception("shim: resize needs a preallocated dst");
  if (!t360o_resize_area(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step))
    throw Exception("shim: cv::resize(INTER_AREA) failed");
}

}  // namespace cv
