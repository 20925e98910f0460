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
#include "t360_sample.h"

namespace t360 {

namespace {

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
