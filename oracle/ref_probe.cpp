/*
 * oracle/ref_probe.cpp -- TEST-ONLY accessors into the REFERENCE's own class.
 *
 * Compiled together with /root/reference/Transform360/Library/VideoFrameTransform.cpp (from
 * where it lies; never copied) against oracle/ref_shim into oracle/_ref/libt360ref.so by
 * oracle/Makefile.  It lets the tests read the reference's warp maps, segment rectangles and
 * 1-D kernels (private members of VideoFrameTransform, VideoFrameTransform.h:147-159) and run
 * the reference's frame path, so the oracle restatement can be pinned to the reference.
 * Only built where /root/reference exists; the GPU box uses the prebuilt .so.
 */
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <opencv2/opencv.hpp>

#define private public
#include "VideoFrameTransform.h"
#undef private

extern "C" {

void* t360ref_new(const FrameTransformContext* ctx) {
  FrameTransformContext copy = *ctx;
  return new VideoFrameTransform(&copy);
}

void t360ref_delete(void* h) { delete (VideoFrameTransform*)h; }

int t360ref_generateMapForPlane(void* h, int inW, int inH, int outW, int outH, int idx) {
  return ((VideoFrameTransform*)h)->generateMapForPlane(inW, inH, outW, outH, idx);
}

int t360ref_transformFramePlane(void* h, uint8_t* in, uint8_t* out, int inW, int inH, int inStride,
                                int outW, int outH, int outStride, int idx, int imagePlane) {
  return ((VideoFrameTransform*)h)
      ->transformFramePlane(in, out, inW, inH, inStride, outW, outH, outStride, idx, imagePlane);
}

int t360ref_transform_pos(void* h, float x, float y, float* ox, float* oy, int idx, float ipw) {
  return ((VideoFrameTransform*)h)->transformPos(x, y, ox, oy, idx, ipw);
}

int t360ref_map_size(void* h, int idx, int* w, int* hgt) {
  VideoFrameTransform* t = (VideoFrameTransform*)h;
  auto it = t->warpMats_.find(idx);
  if (it == t->warpMats_.end()) return 0;
  *w = it->second.cols;
  *hgt = it->second.rows;
  return 1;
}

int t360ref_copy_map(void* h, int idx, float* dst) {
  VideoFrameTransform* t = (VideoFrameTransform*)h;
  auto it = t->warpMats_.find(idx);
  if (it == t->warpMats_.end()) return 0;
  const cv::Mat& m = it->second;
  for (int i = 0; i < m.rows; i++)
    std::memcpy(dst + (size_t)i * m.cols * 2, m.data + (size_t)i * m.step, (size_t)m.cols * 8);
  return 1;
}

int t360ref_num_segments(void* h, int idx) {
  VideoFrameTransform* t = (VideoFrameTransform*)h;
  auto it = t->segmentFilteringConfigs_.find(idx);
  return it == t->segmentFilteringConfigs_.end() ? 0 : (int)it->second.size();
}

/* rect4 = left, top, width, height; lens2 = taps of kX, kY */
int t360ref_segment(void* h, int idx, int i, int* rect4, int* lens2) {
  VideoFrameTransform* t = (VideoFrameTransform*)h;
  const SegmentFilteringConfig& c = t->segmentFilteringConfigs_[idx][i];
  rect4[0] = c.left;
  rect4[1] = c.top;
  rect4[2] = c.width;
  rect4[3] = c.height;
  lens2[0] = t->filterKernelsX_[idx][i].cols * t->filterKernelsX_[idx][i].rows;
  lens2[1] = t->filterKernelsY_[idx][i].cols * t->filterKernelsY_[idx][i].rows;
  return 1;
}

int t360ref_copy_kernels(void* h, int idx, int i, float* kx, float* ky) {
  VideoFrameTransform* t = (VideoFrameTransform*)h;
  const cv::Mat& mx = t->filterKernelsX_[idx][i];
  const cv::Mat& my = t->filterKernelsY_[idx][i];
  std::memcpy(kx, mx.data, sizeof(float) * (size_t)(mx.cols * mx.rows));
  std::memcpy(ky, my.data, sizeof(float) * (size_t)(my.cols * my.rows));
  return 1;
}

/* the reference's filterPlane alone (private), for low-pass orchestration checks */
int t360ref_filterPlane(void* h, uint8_t* in, int inW, int inH, int inStride, uint8_t* dst,
                        int dstStride, int idx) {
  VideoFrameTransform* t = (VideoFrameTransform*)h;
  cv::Mat inputMat(inH, inW, CV_8U, in, (size_t)inStride);
  cv::Mat blurred = t->filterPlane(inputMat, idx, idx);
  for (int y = 0; y < inH; y++)
    std::memcpy(dst + (size_t)y * dstStride, blurred.data + (size_t)y * blurred.step, (size_t)inW);
  return 1;
}

}  // extern "C"
