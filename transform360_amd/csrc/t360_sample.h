// t360_sample.h -- one output sample gathered straight from global memory (device code shared by
// the general gather kernel and the direct tiles of the LDS-tiled kernel).
//
// Arithmetic: OpenCV's fixed-point remap, SURVEY.md Appendix A.3/A.4 (see t360_remap.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "t360_internal.h"

namespace t360 {

__device__ __forceinline__ int wrap_coord(int p, int len) {
  // cv::borderInterpolate(BORDER_WRAP): floored modulo
  if ((unsigned)p < (unsigned)len) return p;
  int m = p % len;
  return m < 0 ? m + len : m;
}

__device__ __forceinline__ int reflect101(int p, int len) {
  // cv::borderInterpolate(BORDER_REFLECT_101)
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p - 1 + 1;
    else
      p = len - 1 - (p - len) - 1;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Returns -1 when BORDER_TRANSPARENT says "leave the destination alone".
template <int KS, bool TRANSPARENT>
__device__ __forceinline__ int sample(const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                      const int16_t* __restrict__ wtab, LutEntry e) {
  if (KS == 1) {
    int sx = e.ix, sy = e.iy;
    if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) return src[(size_t)sy * sstride + sx];
    if (TRANSPARENT) return -1;
    return src[(size_t)wrap_coord(sy, sh) * sstride + wrap_coord(sx, sw)];
  } else {
    constexpr int H = KS / 2 - 1;
    const int cx = e.ix, cy = e.iy;
    const int sx = cx - H, sy = cy - H;
    const int16_t* __restrict__ w = wtab + (size_t)e.frac * (KS * KS);
    int sum = 0;
    if ((unsigned)sx < (unsigned)max(sw - (KS - 1), 0) && (unsigned)sy < (unsigned)max(sh - (KS - 1), 0)) {
      const uint8_t* __restrict__ S = src + (size_t)sy * sstride + sx;
#pragma unroll
      for (int r = 0; r < KS; r++) {
#pragma unroll
        for (int c = 0; c < KS; c++) sum += (int)S[c] * (int)w[r * KS + c];
        S += sstride;
      }
    } else {
      if (TRANSPARENT) {
        if (KS == 2) return -1;  // remapBilinear skips every outlier (single channel)
        if ((unsigned)cx >= (unsigned)sw || (unsigned)cy >= (unsigned)sh) return -1;
      }
      int xi[KS];
#pragma unroll
      for (int c = 0; c < KS; c++) xi[c] = TRANSPARENT ? reflect101(sx + c, sw) : wrap_coord(sx + c, sw);
#pragma unroll
      for (int r = 0; r < KS; r++) {
        const int yr = TRANSPARENT ? reflect101(sy + r, sh) : wrap_coord(sy + r, sh);
        const uint8_t* __restrict__ S = src + (size_t)yr * sstride;
#pragma unroll
        for (int c = 0; c < KS; c++) sum += (int)S[xi[c]] * (int)w[r * KS + c];
      }
    }
    return sat_u8((sum + (1 << (kCoefBits - 1))) >> kCoefBits);
  }
}

}  // namespace t360
