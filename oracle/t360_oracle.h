/*
 * oracle/t360_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the reference's equirect->cubemap remap path, used only
 * as the checker for the HIP implementation:
 *   - tests/                       (parity tests)
 *   - __graft_entry__.smoke()      (one small parity check)
 *   - bench.py's cpu_baseline leg  (timed on the host cores, kind = "port")
 * Nothing under transform360_amd/ may include, link or call anything in oracle/.
 *
 * What is restated, and from where:
 *   t360_oracle_map.c    projection + low-pass configuration of the reference itself
 *                        (reference Transform360/Library/VideoFrameTransform.cpp:53-170,
 *                         210-576, 796-1316); PINNED: compared entry-by-entry with the
 *                        reference's own code compiled from /root/reference
 *                        (oracle/_ref, see oracle/Makefile) and with the golden hashes of
 *                        SURVEY.md Appendix B (tests/golden/).
 *   t360_oracle_cv.c     the arithmetic of the OpenCV calls the reference makes per frame:
 *                        cv::remap (VideoFrameTransform.cpp:748-754), cv::sepFilter2D
 *                        (:189-197).  OpenCV is an UN-VENDORED, UN-VERSIONED dependency of
 *                        the reference (CMakeLists.txt:11, "libopencv-dev") and is not
 *                        installed here; the reference has no tests or golden images.
 *                        This part restates OpenCV 4.x's published fixed-point algorithms
 *                        (imgwarp.cpp remap/initInterTab2D, filter.simd.hpp) and is
 *                        ** PARITY UNPINNED ** at the OpenCV boundary.
 *   t360_oracle_frame.c  the per-frame orchestration (VideoFrameTransform.cpp:173-204,
 *                        579-794, 1319-1351) with the reference's threading structure
 *                        (row stripes for remap, one task per segment for the low-pass).
 *                        PINNED for orchestration: oracle/_ref runs the reference's own
 *                        frame path with the cv:: calls bound to t360_oracle_cv.c.
 */
#ifndef T360_ORACLE_H
#define T360_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "Transform360/VideoFrameTransformHelper.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- OpenCV constants restated (imgproc.hpp / core/base.hpp) ---- */
enum {
  T360O_BORDER_CONSTANT = 0,
  T360O_BORDER_REPLICATE = 1,
  T360O_BORDER_REFLECT = 2,
  T360O_BORDER_WRAP = 3,
  T360O_BORDER_REFLECT_101 = 4,
  T360O_BORDER_TRANSPARENT = 5
};

/* ---- projection (t360_oracle_map.c) ---- */

/* One call of the reference's transformPos (VideoFrameTransform.cpp:893-1316).
 * Returns 1 on success (the reference's "true"), 0 for an invalid layout. */
int t360o_transform_pos(const FrameTransformContext* ctx, float x, float y,
                        float* outX, float* outY, float inputPixelWidth);

/* generateMapForPlane's map loop (VideoFrameTransform.cpp:524-554).  `map` receives
 * scaledH*scaledW interleaved (x,y) float pairs, row-major.  Returns 1/0. */
int t360o_scaled_size(const FrameTransformContext* ctx, int outW, int outH, int* scaledW, int* scaledH);
int t360o_generate_map(const FrameTransformContext* ctx, int inW, int inH, int outW, int outH,
                       float* map);

/* ---- low-pass configuration (t360_oracle_map.c) ---- */
typedef struct T360OSegment {
  int left, top, width, height; /* SegmentFilteringConfig, VideoFrameTransform.h:25-38 */
  int kx_len, ky_len;           /* taps */
  float* kx;                    /* 1-D kernels, owned */
  float* ky;
} T360OSegment;

typedef struct T360OFilterConfig {
  int count;
  int capacity;
  T360OSegment* seg;
} T360OFilterConfig;

/* calculateKernel (VideoFrameTransform.cpp:78-94).  Returns malloc'd taps, *len set. */
float* t360o_calculate_kernel(float sigma, int* len);
/* calcualteFilteringConfig (VideoFrameTransform.cpp:367-501): appends to cfg. */
int t360o_filter_config(const FrameTransformContext* ctx, int inW, int inH, int outW, int outH,
                         T360OFilterConfig* cfg);
void t360o_filter_config_free(T360OFilterConfig* cfg);
double t360o_effective_ratio(double angularDist, double offset); /* :168-170 */

/* ---- OpenCV arithmetic (t360_oracle_cv.c) ---- */
int t360o_border_interpolate(int p, int len, int borderType);
/* 1024-entry 2-D Q15 coefficient table for interp 1/2/4 (ksize 2/4/8); ksize*ksize shorts each. */
const int16_t* t360o_inter_tab(int interp, int* ksize);
/* cv::remap for 8-bit single channel, CV_32FC2 map, no map2.  rows [row0,row1) only. */
void t360o_remap_rows(const uint8_t* src, int sw, int sh, size_t sstep,
                      uint8_t* dst, int dw, int dh, size_t dstep,
                      const float* map, int interp, int borderType, int row0, int row1);
/* cv::sepFilter2D(parent(roi), dst(roi), -1, kx, ky, (-1,-1), 0, BORDER_REPLICATE) on 8-bit data.
 * Returns the path taken: 1 = fixed-point, 0 = float; -1 = ROI outside the parent (nothing done,
 * like the cv::Exception the reference swallows at VideoFrameTransform.cpp:198-203). */
int t360o_sepfilter_roi(const uint8_t* parent, int pw, int ph, size_t pstep,
                        uint8_t* dparent, size_t dstep,
                        int left, int top, int width, int height,
                        const float* kx, int kx_len, const float* ky, int ky_len);
int t360o_kernel_type(const float* k, int len); /* cv::getKernelType with the default anchor */
/* tie-rounding variant of the two places where a SIMD build of OpenCV and its scalar code differ (t360_oracle_cv.c) */
void t360o_set_cv_variant(int column_simd_lanes, int area_tail_lanes);

/* cv::resize(..., INTER_AREA) for CV_8UC1, shrinking only (VideoFrameTransform.cpp:770-776);
 * returns 0 for requests it does not restate (enlargement). */
int t360o_resize_area(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                      size_t dstep);

/* ---- frame path with the reference's call protocol (t360_oracle_frame.c) ---- */
typedef struct T360Oracle T360Oracle;

T360Oracle* t360o_new(const FrameTransformContext* ctx);
void t360o_delete(T360Oracle* o);
/* threads <= 0: std::thread::hardware_concurrency() equivalent; 1: fully serial. */
void t360o_set_threads(T360Oracle* o, int threads);
int t360o_generateMapForPlane(T360Oracle* o, int inW, int inH, int outW, int outH, int mapIdx);
int t360o_transformFramePlane(T360Oracle* o, const uint8_t* in, uint8_t* out,
                              int inW, int inH, int inStride, int outW, int outH, int outStride,
                              int mapIdx, int imagePlaneIdx);
/* accessors for parity tests */
const float* t360o_map(const T360Oracle* o, int mapIdx, int* w, int* h);
const T360OFilterConfig* t360o_segments(const T360Oracle* o, int mapIdx);
/* low-pass only (filterPlane, VideoFrameTransform.cpp:621-704): dst gets the blurred plane. */
int t360o_filterPlane(T360Oracle* o, const uint8_t* in, int inW, int inH, int inStride,
                      uint8_t* dst, int dstStride, int mapIdx);

/* FNV-1a 64 used by the golden vectors. */
uint64_t t360o_fnv1a64(const void* data, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
