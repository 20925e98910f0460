// t360_resize.hip -- cv::resize(..., INTER_AREA) for 8-bit planes, shrinking: the decimation step
// of the reference's supersample antialiasing (transformPlane's needResize branch,
// VideoFrameTransform.cpp:759-776: remap into a warp-map-sized image, then resize to the output).
//
// Arithmetic of OpenCV 4.x resize.cpp (DESIGN.md section 2 lists what is pinned):
//   integer factors (source = factor x destination exactly), "ResizeAreaFast":
//       2 x 2      (a + b + c + d + 2) >> 2
//       otherwise  sat_u8(rint((float)int_sum * (1.f / area)))
//   fractional factors, "ResizeArea" with DecimateAlpha tables built on the host in double:
//       buf = SUM_k S[si_k] * alpha_k   (float, table order, mul then add)
//       sum = beta_0 * buf_0 + beta_1 * buf_1 + ...   (float, row order)
//       out = sat_u8(rint(sum))
//   either factor ENLARGING (scale < 1): OpenCV emulates INTER_AREA with its 8-bit bilinear kernels and area-mode
//       coefficients (11-bit fixed point, host tables):
//       D_r = S_r[sx] * a0 + S_r[sx + 1] * a1  (S_r[sx] * 2048 from column xmax on),  rows sy, sy + 1 clamped,
//       out = (((b0 * (D_0 >> 4)) >> 16) + ((b1 * (D_1 >> 4)) >> 16) + 2) >> 2
// One lane per output pixel; the work is a few source bytes per output byte, HBM/L2 streaming.
#include <hip/hip_runtime.h>

#include "t360_kernels.h"

#pragma clang fp contract(off)

namespace t360 {

namespace {

__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void resize_area_kernel(ResizeArgs a) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y;
  if (dx >= a.dw) return;
  const uint8_t* __restrict__ src = a.src + (size_t)blockIdx.z * a.src_frame_bytes;
  uint8_t* __restrict__ dst = a.dst + (size_t)blockIdx.z * a.dst_frame_bytes;
  int out;
  if (a.linear) {
    // x tables: x_si = sx, xofs[2 dx], [2 dx + 1] = a0, a1; y tables: y_si = sy, yofs[2 dy], [2 dy + 1] = b0, b1
    const int sx = a.x_si[dx], a0 = a.xofs[2 * dx], a1 = a.xofs[2 * dx + 1];
    const int sy = a.y_si[dy], b0 = a.yofs[2 * dy], b1 = a.yofs[2 * dy + 1];
    const int r0 = min(max(sy, 0), a.sh - 1), r1 = min(max(sy + 1, 0), a.sh - 1);
    const uint8_t* __restrict__ S0 = src + (size_t)r0 * a.sstride;
    const uint8_t* __restrict__ S1 = src + (size_t)r1 * a.sstride;
    int d0, d1;
    if (dx < a.xmax) {
      d0 = S0[sx] * a0 + S0[sx + 1] * a1;
      d1 = S1[sx] * a0 + S1[sx + 1] * a1;
    } else {
      d0 = S0[sx] * 2048;
      d1 = S1[sx] * 2048;
    }
    out = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
  } else if (a.iscale_x > 0) {
    int sum = 0;
    const uint8_t* __restrict__ S = src + (size_t)(dy * a.iscale_y) * a.sstride + (size_t)dx * a.iscale_x;
    for (int sy = 0; sy < a.iscale_y; sy++, S += a.sstride)
      for (int sx = 0; sx < a.iscale_x; sx++) sum += S[sx];
    if (a.iscale_x == 2 && a.iscale_y == 2)
      out = (sum + 2) >> 2;
    else
      out = sat_u8(__float2int_rn(__fmul_rn((float)sum, a.inv_area)));
  } else {
    const int k0 = a.xofs[dx], k1 = a.xofs[dx + 1];
    const int j0 = a.yofs[dy], j1 = a.yofs[dy + 1];
    float sum = 0.f;
    for (int j = j0; j < j1; j++) {
      const uint8_t* __restrict__ S = src + (size_t)a.y_si[j] * a.sstride;
      float buf = 0.f;
      for (int k = k0; k < k1; k++) buf = __fadd_rn(buf, __fmul_rn((float)S[a.x_si[k]], a.x_alpha[k]));
      sum = __fadd_rn(sum, __fmul_rn(a.y_alpha[j], buf));  // first term: 0 + v == v exactly
    }
    out = sat_u8(__float2int_rn(sum));
  }
  dst[(size_t)dy * a.dstride + dx] = (uint8_t)out;
}

}  // namespace

hipError_t launch_resize_area(const ResizeArgs& a, int nframes, hipStream_t stream) {
  if (a.dw <= 0 || a.dh <= 0 || nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(resize_area_kernel, dim3((a.dw + 255) / 256, a.dh, nframes), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace t360
