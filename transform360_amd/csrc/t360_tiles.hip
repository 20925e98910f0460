// t360_tiles.hip -- init-time planning kernels for the LDS-tiled gather.
//
// The tiled gather (t360_remap_tiled.hip) stages, per output tile, the bounding box of all
// source pixels the tile's stencils touch.  These kernels derive that geometry from the sample
// LUT once per map:
//   tile_scan_kernel  one workgroup per 32x32 output macro tile: bounding boxes (in source
//                     pixel-centre coordinates, before the stencil halo) of the macro tile and of
//                     its four 16x16 quadrants, each in two variants -- raw x, and x shifted by
//                     -W for the right half of the plane, so that a tile straddling the +-180
//                     degree seam (the BACK face) gets a narrow box.
//   tile_lut_kernel   one workgroup per planned tile: rewrites the absolute LUT entries into
//                     box-relative packed words, in the exact lane order the gather consumes.
// The host (t360_plan.cpp) turns the scanned boxes into the tile work list in between.
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"

namespace t360 {

namespace {

__device__ __forceinline__ void box_init(int* b) {
  b[0] = b[2] = b[4] = 1 << 30;     // min x, min x shifted, min y
  b[1] = b[3] = b[5] = -(1 << 30);  // max x, max x shifted, max y
}

}  // namespace

// One workgroup per 128x32 output macro region.  out[region][box 0..23][6]:
//   box 0..3   the four 128x8 strips (top to bottom)
//   box 4..7   the four 32x32 tiles (left to right)
//   box 8..23  the sixteen 16x16 quadrants: 8 + 4*tile + quadrant (quadrant row-major in its tile)
__global__ __launch_bounds__(256) void tile_scan_kernel(const LutEntry* __restrict__ lut, int dw, int dh, int sw,
                                                        int regions_x, int* __restrict__ out) {
  __shared__ int box[kScanBoxes][6];
  const int tid = threadIdx.x;
  if (tid < kScanBoxes) box_init(box[tid]);
  __syncthreads();
  const int rx = blockIdx.x % regions_x, ry = blockIdx.x / regions_x;
  // lane = (column within the region, band of 16 rows); 16 pixels per lane
  const int lx = tid & 127, band = tid >> 7;
  const int ox = rx * 128 + lx;
  if (ox < dw) {
    for (int k = 0; k < 16; k++) {
      const int ly = band * 16 + k;
      const int oy = ry * 32 + ly;
      if (oy >= dh) break;
      const LutEntry e = lut[(size_t)oy * dw + ox];
      const int x = e.ix, y = e.iy;
      const int xs = x >= (sw >> 1) ? x - sw : x;
      const int tile = lx >> 5;
      const int ids[3] = {ly >> 3, 4 + tile, 8 + 4 * tile + ((ly >> 4) << 1) + ((lx & 31) >> 4)};
      for (int b = 0; b < 3; b++) {
        int* B = box[ids[b]];
        atomicMin(&B[0], x);
        atomicMax(&B[1], x);
        atomicMin(&B[2], xs);
        atomicMax(&B[3], xs);
        atomicMin(&B[4], y);
        atomicMax(&B[5], y);
      }
    }
  }
  __syncthreads();
  if (tid < kScanBoxes * 6) out[(size_t)blockIdx.x * (kScanBoxes * 6) + tid] = box[tid / 6][tid % 6];
}

hipError_t launch_tile_scan(const LutEntry* lut, int dw, int dh, int sw, int* out, hipStream_t stream) {
  const int regions_x = (dw + 127) / 128, regions_y = (dh + 31) / 32;
  hipLaunchKernelGGL(tile_scan_kernel, dim3(regions_x * regions_y), dim3(256), 0, stream, lut, dw, dh, sw, regions_x, out);
  return hipGetLastError();
}

// Box-relative LUT word:  bits 0..9 column of the stencil's left tap inside the staged box,
// bits 10..17 row of its top tap, bits 18..27 sub-pixel phase (frac), bit 31 = pixel outside
// the plane (partial tile).
__global__ __launch_bounds__(256) void tile_lut_kernel(const LutEntry* __restrict__ lut, int dw, int dh, int sw,
                                                       const TileDesc* __restrict__ tiles, int halo,
                                                       uint32_t* __restrict__ tlut) {
  const TileDesc t = tiles[blockIdx.x];
  if (t.kind == kTileDirect16) return;  // the direct path reads the absolute LUT
  const int tid = threadIdx.x;
  const int npx = t.kind == kTileStaged16 ? 1 : 4;
  for (int p = 0; p < npx; p++) {
    int ox, oy;
    if (t.kind == kTileStaged32) {
      // lane = column (tid & 31) of a band of 4 rows: the 32 lanes of a half-wave read ONE
      // source-row neighbourhood per instruction (conflict-free ds_read2_b32, see the gather)
      ox = t.ox + (tid & 31);
      oy = t.oy + (tid >> 5) * 4 + p;
    } else if (t.kind == kTileStrip128) {
      ox = t.ox + (tid & 127);
      oy = t.oy + (tid >> 7) * 4 + p;
    } else if (t.kind == kTileWide64) {
      ox = t.ox + (tid & 63);
      oy = t.oy + (tid >> 6) * 4 + p;
    } else {
      ox = t.ox + (tid & 15);
      oy = t.oy + (tid >> 4);
    }
    uint32_t word = 0x80000000u;
    if (ox < dw && oy < dh) {
      const LutEntry e = lut[(size_t)oy * dw + ox];
      int x = e.ix;
      if (t.flags & kTileSeamShift) x = x >= (sw >> 1) ? x - sw : x;
      const int rx = x - halo - t.x0;
      const int ry = (int)e.iy - halo - t.y0;
      word = (uint32_t)rx | ((uint32_t)ry << 10) | ((uint32_t)e.frac << 18);
    }
    tlut[(size_t)t.tlut + (size_t)tid * npx + p] = word;
  }
}

hipError_t launch_tile_lut(const LutEntry* lut, int dw, int dh, int sw, const TileDesc* tiles, int ntiles,
                           int halo, uint32_t* tlut, hipStream_t stream) {
  if (ntiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(tile_lut_kernel, dim3(ntiles), dim3(256), 0, stream, lut, dw, dh, sw, tiles, halo, tlut);
  return hipGetLastError();
}

}  // namespace t360
