// t360_kernels.h -- launch interfaces of the HIP kernels (internal to libTransform360).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t360_internal.h"

namespace t360 {

// ---- projection (t360_mapgen.hip) ----
hipError_t launch_mapgen(const MapGenParams& P, float2* map, LutEntry* lut, hipStream_t stream);
hipError_t launch_fill_noise(uint8_t* dst, int64_t nbytes, uint64_t seed, hipStream_t stream);

// ---- gather (t360_remap.hip) ----
// One plane of `nframes` frames: frame f lives at src + f*src_frame_bytes / dst + f*dst_frame_bytes.
struct GatherArgs {
  const uint8_t* src;
  int64_t src_frame_bytes;
  int sw, sh, sstride;
  uint8_t* dst;
  int64_t dst_frame_bytes;
  int dw, dh, dstride;
  const LutEntry* lut;   // dw*dh entries
  const int16_t* wtab;   // 1024 * k*k Q15 weights (unused for NEAREST)
  int interp;            // InterpolationAlg
  int border;            // kBorderWrap or kBorderTransparent
};
hipError_t launch_remap_gather(const GatherArgs& a, int nframes, hipStream_t stream);
hipError_t launch_fill_plane(uint8_t* dst, int64_t frame_bytes, int w, int h, int stride, int value,
                             int nframes, hipStream_t stream);

// ---- tile planning (t360_tiles.hip) ----
// out: kScanBoxes*6 ints per 128x32 macro region ({minx, maxx, minx_shifted, maxx_shifted, miny, maxy} per box)
constexpr int kScanBoxes = 24;
hipError_t launch_tile_scan(const LutEntry* lut, int dw, int dh, int sw, int* out, hipStream_t stream);
hipError_t launch_tile_lut(const LutEntry* lut, int dw, int dh, int sw, const TileDesc* tiles, int ntiles,
                           int halo, uint32_t* tlut, hipStream_t stream);

// ---- LDS-tiled bicubic gather (t360_remap_tiled.hip) ----
struct TiledPlane {
  const uint8_t* src;       // plane base of frame 0
  int64_t src_frame_bytes;  // distance between consecutive frames of this plane
  uint8_t* dst;
  int64_t dst_frame_bytes;
  int sw, sh, sstride;
  int dw, dh, dstride;
  const TileDesc* tiles;
  const uint32_t* tlut;     // box-relative LUT words
  const LutEntry* lut;      // absolute LUT (direct tiles)
  int ntiles;
  int dst_dword_ok;         // plane base, stride and frame distance are 4-byte aligned: dword stores
  int src_vec_ok;           // ... 16-byte aligned and sw % 16 == 0: every staged chunk is one dwordx4
  int ndirect;              // kTileDirect16 descriptors stored behind the ntiles staged ones
};
struct TiledArgs {
  const int16_t* wtab;      // OpenCV Q15 table (direct tiles)
  const uint32_t* wpack;    // pack_dwords(ks) per phase (staged tiles)
  int ks;                   // taps per axis of the interpolation: 1, 2, 4 (bicubic) or 8
  int nframes;
  int frames_per_block;
  int nplanes;
  int total_tiles;
  int ring_bytes;           // LDS ring of the DMA-staged kernel
  int loader_waves;         // DMA loader waves per workgroup (1..4) next to the 4 consumer waves
  int debug;                // experiments (T360_DEBUG): bit2 no steady-state DMA, bit3 no gather
  int variant;              // DMA kernel build: bit0 LDS reads in groups of 2 px, bit2 no loader wave,
                            // bit3 persistent workgroups, bit4 LDS flags instead of the frame barrier
  unsigned long long* trace;  // optional: 8 timestamps (100 MHz) per workgroup (T360_TRACE)
  int* work_counters;       // persistent variant (bit3): 8 zeroed ints, one item queue per XCD
  int persist_slots;        // persistent variant: workgroups to launch (resident slots of the device)
  TiledPlane plane[4];
};
// Fused launch over all planes; every plane must have src_vec_ok (chunks go global -> LDS by DMA).
hipError_t launch_remap_tiled_cubic_dma(const TiledArgs& a, hipStream_t stream);
// Direct (unstaged) tiles of all planes in `a` (plane[k].tiles + ntiles, ndirect of them): one launch.
hipError_t launch_remap_direct_cubic(const TiledArgs& a, hipStream_t stream);
// One plane, chunks staged through registers (any alignment / width).
hipError_t launch_remap_tiled_cubic_regs(const TiledArgs& a, hipStream_t stream);

// ---- segmented separable low-pass (t360_lowpass.hip) ----
struct LowpassArgs {
  const uint8_t* src;
  int64_t src_frame_bytes;
  int sstride;
  uint8_t* dst;
  int64_t dst_frame_bytes;
  int dstride;
  int w, h;                   // plane size (replicate border at its edges only)
  const LowpassTile* tiles;   // ntiles work items
  int ntiles;
  const SegmentDev* segs;
  const int* taps_q8;         // packed integer taps (Q8)
  const float* taps_f32;      // packed float taps, same offsets
  int max_rows;               // max over tiles of (tile.h + 2*ry): sizes the LDS row buffer
  int tile_w;                 // tile width used when the list was built
  // fast Q8 path (lowpass_q8_kernel): tiles of <= 128 x 128 px inside one qualifying segment
  const LowpassTile* fast_tiles;
  int nfast;
  int fast_ky;                // vertical taps of every fast tile (3, 5 or 7)
  int fast_lds_bytes;         // max over fast tiles of the staged source rectangle
  const uint32_t* taps_pk;    // horizontal Q8 taps packed 4 per dword, zero padded (SegmentDev::kxp_off)
  int dst_dword_ok;           // dst base and stride are multiples of 4
};
hipError_t launch_lowpass(const LowpassArgs& a, int nframes, hipStream_t stream);

// ---- INTER_AREA shrink of the supersampled plane (t360_resize.hip) ----
struct ResizeArgs {
  const uint8_t* src;
  int64_t src_frame_bytes;
  int sstride, sw, sh;
  uint8_t* dst;
  int64_t dst_frame_bytes;
  int dstride, dw, dh;
  int iscale_x, iscale_y;  // > 0: integer factors (ResizeAreaFast); 0: table driven (ResizeArea)
  float inv_area;          // 1.f / (iscale_x * iscale_y)
  const int* xofs;         // [dw + 1] ranges into x_si / x_alpha
  const int* x_si;
  const float* x_alpha;
  const int* yofs;         // [dh + 1]
  const int* y_si;
  const float* y_alpha;
};
hipError_t launch_resize_area(const ResizeArgs& a, int nframes, hipStream_t stream);

}  // namespace t360
