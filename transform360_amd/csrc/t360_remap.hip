// t360_remap.hip -- the per-frame gather (cv::remap as the reference calls it,
// reference VideoFrameTransform.cpp:748-754: CV_8U plane, CV_32FC2 map, BORDER_WRAP, or
// BORDER_TRANSPARENT for the barrel layouts).
//
// Arithmetic restated from OpenCV's fixed-point remap:
//   NEAREST : dst = src[wrap(iy)][wrap(ix)]
//   others  : sum = SUM_{r,c<k} src[wrap(iy-h+r)][wrap(ix-h+c)] * itab[frac][r*k+c]   (int32)
//             dst = saturate_u8((sum + 16384) >> 15),  k = 2/4/8, h = k/2-1
// The (ix, iy, frac) triples come from the LUT built once by t360_mapgen.hip.
//
// Kernel in this file: remap_gather_kernel -- direct gather from global memory, four output
// pixels per lane, any interpolation / border mode / plane shape.  It is the general path and
// the fallback of the LDS-tiled kernels (pole tiles whose source footprint does not fit LDS).
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"

namespace t360 {

namespace {

__device__ __forceinline__ int wrap_coord(int p, int len) {
  // cv::borderInterpolate(BORDER_WRAP): the result of a floored modulo
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

// One output sample.  Returns -1 when BORDER_TRANSPARENT says "leave the destination alone".
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

template <int KS, bool TRANSPARENT>
__global__ __launch_bounds__(256) void remap_gather_kernel(GatherArgs a) {
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y;
  const int f = blockIdx.z;
  if (x0 >= a.dw) return;
  const uint8_t* __restrict__ src = a.src + (size_t)f * a.src_frame_bytes;
  uint8_t* __restrict__ drow = a.dst + (size_t)f * a.dst_frame_bytes + (size_t)y * a.dstride;
  const LutEntry* __restrict__ L = a.lut + (size_t)y * a.dw + x0;

  int v[4];
  const int n = min(4, a.dw - x0);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = -1;
    if (k < n) v[k] = sample<KS, TRANSPARENT>(src, a.sw, a.sh, a.sstride, a.wtab, L[k]);
  }
  uint8_t* d = drow + x0;
  if (!TRANSPARENT && n == 4 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(d) =
        (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < n && v[k] >= 0) d[k] = (uint8_t)v[k];
  }
}

template <int KS>
hipError_t launch_gather_ks(const GatherArgs& a, int nframes, hipStream_t stream) {
  dim3 block(64, 1, 1);
  dim3 grid((a.dw + 255) / 256, a.dh, nframes);
  if (a.border == kBorderTransparent)
    hipLaunchKernelGGL((remap_gather_kernel<KS, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((remap_gather_kernel<KS, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_remap_gather(const GatherArgs& a, int nframes, hipStream_t stream) {
  if (a.dw <= 0 || a.dh <= 0 || nframes <= 0) return hipSuccess;
  switch (a.interp) {
    case NEAREST: return launch_gather_ks<1>(a, nframes, stream);
    case LINEAR: return launch_gather_ks<2>(a, nframes, stream);
    case CUBIC: return launch_gather_ks<4>(a, nframes, stream);
    case LANCZOS4: return launch_gather_ks<8>(a, nframes, stream);
    default: return hipErrorInvalidValue;
  }
}

// ---- plane fill (barrel chroma pre-fill with 128, reference VideoFrameTransform.cpp:743-747) ----
__global__ __launch_bounds__(256) void fill_plane_kernel(uint8_t* dst, int64_t frame_bytes, int w, int h,
                                                         int stride, int value) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x < w && y < h) dst[(size_t)blockIdx.z * frame_bytes + (size_t)y * stride + x] = (uint8_t)value;
}

hipError_t launch_fill_plane(uint8_t* dst, int64_t frame_bytes, int w, int h, int stride, int value,
                             int nframes, hipStream_t stream) {
  if (w <= 0 || h <= 0 || nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_plane_kernel, dim3((w + 255) / 256, h, nframes), dim3(256), 0, stream, dst,
                     frame_bytes, w, h, stride, value);
  return hipGetLastError();
}

}  // namespace t360
