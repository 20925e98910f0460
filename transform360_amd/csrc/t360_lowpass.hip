// t360_lowpass.hip -- segmented separable low-pass filter (the reference's filterPlane ->
// runFiltering -> filterSegment -> cv::sepFilter2D chain, VideoFrameTransform.cpp:173-204,
// 579-704).
//
// Semantics restated: for an output pixel (x, y) of segment S
//     out = colpass_S( rowpass_S( plane[clampY(y+dy)][clampX(x+dx)] ) )
// i.e. pixels outside the segment are the REAL neighbours in the whole plane (the segment is an
// ROI of the parent image and BORDER_ISOLATED is not set), replicated only beyond the plane's
// own edges, and halo rows are row-filtered with S's own horizontal kernel.
//   fixed-point path (both kernels SMOOTH|SYMMETRICAL): taps * 256 rounded to int,
//       r = SUM kx_q8 * src (int32), c = SUM ky_q8 * r, out = sat_u8((c + 32768) >> 16)
//   float path: r = SUM kx * src (ascending), c = ky[0]*r0 + SUM ky[k]*(r+k + r-k) for a
//       symmetric vertical kernel (ascending otherwise), out = sat_u8(rint(c))
//
// One workgroup per (frame, tile); a tile is a <= tile_w x tile_h rectangle inside ONE segment.
// Pass 1 writes the row-filtered values of the tile's rows plus the vertical halo to LDS,
// pass 2 reads them column-wise.  Source bytes come straight from global memory (L2-resident:
// neighbouring tiles overlap by the kernel radius).
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"

namespace t360 {

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void lowpass_kernel(LowpassArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_rows[];  // [rows][tile_w] int32 or float bits

  const LowpassTile t = a.tiles[blockIdx.x];
  const SegmentDev s = a.segs[t.seg];
  const uint8_t* __restrict__ src = a.src + (size_t)blockIdx.y * a.src_frame_bytes;
  uint8_t* __restrict__ dst = a.dst + (size_t)blockIdx.y * a.dst_frame_bytes;
  const int rx = s.kx_len >> 1, ry = s.ky_len >> 1;
  const int rows = t.h + 2 * ry;
  const int tw = t.w;
  const int pitch = a.tile_w;

  if (s.fixed_point) {
    const int* __restrict__ kx = a.taps_q8 + s.kx_off;
    const int* __restrict__ ky = a.taps_q8 + s.ky_off;
    for (int idx = threadIdx.x; idx < rows * tw; idx += blockDim.x) {
      const int r = idx / tw, x = idx - r * tw;
      const uint8_t* __restrict__ S = src + (size_t)clampi(t.y0 - ry + r, 0, a.h - 1) * a.sstride;
      const int xs = t.x0 + x - rx;
      int acc = 0;
      if (xs >= 0 && xs + s.kx_len <= a.w) {
        for (int k = 0; k < s.kx_len; k++) acc += kx[k] * (int)S[xs + k];
      } else {
        for (int k = 0; k < s.kx_len; k++) acc += kx[k] * (int)S[clampi(xs + k, 0, a.w - 1)];
      }
      lds_rows[r * pitch + x] = acc;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < t.h * tw; idx += blockDim.x) {
      const int y = idx / tw, x = idx - y * tw;
      int acc = 0;
      for (int k = 0; k < s.ky_len; k++) acc += ky[k] * lds_rows[(y + k) * pitch + x];
      dst[(size_t)(t.y0 + y) * a.dstride + t.x0 + x] = (uint8_t)sat_u8((acc + (1 << 15)) >> 16);
    }
  } else {
    float* frow = reinterpret_cast<float*>(lds_rows);
    const float* __restrict__ kx = a.taps_f32 + s.kx_off;
    const float* __restrict__ ky = a.taps_f32 + s.ky_off;
    for (int idx = threadIdx.x; idx < rows * tw; idx += blockDim.x) {
      const int r = idx / tw, x = idx - r * tw;
      const uint8_t* __restrict__ S = src + (size_t)clampi(t.y0 - ry + r, 0, a.h - 1) * a.sstride;
      const int xs = t.x0 + x - rx;
      float acc = __fmul_rn(kx[0], (float)S[clampi(xs, 0, a.w - 1)]);
      for (int k = 1; k < s.kx_len; k++)
        acc = __fadd_rn(acc, __fmul_rn(kx[k], (float)S[clampi(xs + k, 0, a.w - 1)]));
      frow[r * pitch + x] = acc;
    }
    __syncthreads();
    // symmetric vertical kernel? (device-side check mirrors cv::getKernelType's palindrome test)
    bool symmetric = (s.ky_len & 1) != 0;
    for (int k = 0; k < ry && symmetric; k++) symmetric = ky[k] == ky[s.ky_len - 1 - k];
    for (int idx = threadIdx.x; idx < t.h * tw; idx += blockDim.x) {
      const int y = idx / tw, x = idx - y * tw;
      float acc;
      if (symmetric) {
        acc = __fmul_rn(ky[ry], frow[(y + ry) * pitch + x]);
        for (int k = 1; k <= ry; k++)
          acc = __fadd_rn(acc, __fmul_rn(ky[ry + k], __fadd_rn(frow[(y + ry + k) * pitch + x],
                                                               frow[(y + ry - k) * pitch + x])));
      } else {
        acc = __fmul_rn(ky[0], frow[y * pitch + x]);
        for (int k = 1; k < s.ky_len; k++) acc = __fadd_rn(acc, __fmul_rn(ky[k], frow[(y + k) * pitch + x]));
      }
      dst[(size_t)(t.y0 + y) * a.dstride + t.x0 + x] = (uint8_t)sat_u8(__float2int_rn(acc));
    }
  }
}

}  // namespace

hipError_t launch_lowpass(const LowpassArgs& a, int nframes, hipStream_t stream) {
  if (a.ntiles <= 0 || nframes <= 0) return hipSuccess;
  const size_t lds = (size_t)a.max_rows * (size_t)a.tile_w * sizeof(int);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lowpass_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(lowpass_kernel, dim3(a.ntiles, nframes, 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace t360
