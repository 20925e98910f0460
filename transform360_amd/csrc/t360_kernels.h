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

// ---- LDS-tiled gather (t360_remap_tiled.hip; work list planned by t360_plan.cpp) ----
struct TiledPlane {
  const uint8_t* src;       // plane base of frame 0
  int64_t src_frame_bytes;  // distance between consecutive frames of this plane
  uint8_t* dst;
  int64_t dst_frame_bytes;
  int sw, sh, sstride;
  int dw, dh, dstride;
  const TileDesc* tiles;    // ntiles staged tiles, then ndirect direct tiles
  const uint32_t* tlut;     // pixel words of the staged tiles
  const uint32_t* chunks;   // chunk tables + row tables of the staged tiles
  const LutEntry* lut;      // absolute LUT (direct tiles)
  int ntiles;
  int ndirect;              // direct tiles: those of the upper half of the plane first
  int ndirect_top;
  int dst_dword_ok;         // plane base, stride and frame distance are 4-byte aligned: dword stores
  int scatter;              // this plane's chunk tables carry 128 block origins per tile (scatter plan, t360_plan.h)
};
struct TiledArgs {
  const int16_t* wtab;      // OpenCV Q15 table (direct tiles)
  const uint32_t* wpack;    // pack_dwords(ks) per phase (staged tiles)
  int ks;                   // taps per axis of the interpolation: 1, 2, 4 (bicubic) or 8
  int nframes;
  int frames_per_block;     // frames one workgroup walks with its tile
  int nplanes;
  int total_tiles;          // staged tiles of all planes
  int total_direct;         // direct tiles of all planes
  int groups;               // frame groups = ceil(nframes / frames_per_block): work items per tile
  int tail_percent;         // the last tail_percent % of every XCD's tiles use runs of tail_frames frames instead
  int tail_frames, tail_groups;
  int direct_blocks;        // work items reserved for direct tiles: total_direct * groups, rounded up to a multiple of 8
  int max_pieces;           // the plan's staging budget per tile and copy (1 KiB pieces)
  int ring_kb;              // LDS per workgroup in KiB and waves per workgroup (4 or 8): select the kernel instantiation
  int waves;
#ifdef T360_INSTRUMENT
  int lds_pad;              // instrumented build only: extra dynamic LDS per workgroup (occupancy experiments)
  int k_lo, k_hi;           // instrumented build only (INCOMPLETE OUTPUT): only work items k_lo <= k < k_hi of every XCD run (k_hi 0 = all)
  int debug;                // instrumented build only (WRONG PIXELS): bit0 no gather, bit1 no steady-state DMA,
                            // bit2 skip direct tiles, bit3 skip 16x16 tiles, bit4 skip 4-px tiles, bit5 no copy B,
                            // bit6 every frame reads frame 0's source (L2 hits), bit7 no frame barrier, bit8 no output stores;
                            // (right pixels) bit9 s_setprio 3 around the DMA issue, bit10 around the deferred store and the DMA issue;
                            // remap_fused_kernel (WRONG PIXELS): bit0 no gather, bit1 no steady-state DMA, bit11 no filter pass,
                            // bit12 no blurred writes, bit13 no second barrier
  unsigned long long* trace;  // instrumented build only: 8 timestamps (100 MHz) per workgroup (T360_TRACE)
  unsigned long long* phases; // instrumented build only: 2 x 8 cycle sums per workgroup (T360_PHASES)
#endif
  TiledPlane plane[4];
};
// One launch for all planes of a batch (<= 4): staged tiles by LDS-DMA, pole tiles gathered directly.
// Every plane's source must be 16-byte friendly (base, stride, frame distance, width).
hipError_t launch_remap_tiled(const TiledArgs& a, hipStream_t stream);
// the instantiation launch_remap_tiled() picks for these parameters (reporting)
const char* remap_tiled_kernel_name(int ks, int ring_kb, int waves);

// ---- LDS-tiled gather with the low-pass FUSED in (t360_remap_tiled.hip: remap_fused_kernel; t360_internal.h "fused
// low-pass tiles") ----
// `base` describes the launch like a launch of the tiled kernel: plane[k].src is the RAW source plane, tiles / tlut / chunks
// are the FUSED work list of the plan (tables at fused_chunk_dwords(max_pieces)), ntiles its length, ndirect 0; taps[k] =
// the plane's packed fusable kernels (kFusedTapDwords dwords each).  Bilinear and bicubic, workgroups of 8 waves, 76 KiB rings.
struct FusedArgs {
  TiledArgs base;
  const uint32_t* taps[4];
};
hipError_t launch_remap_fused(const FusedArgs& a, hipStream_t stream);

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
  // wide fast path (lowpass_q8w_kernel): tiles of <= 512 x 32 px of a run of segments with identical kernels,
  // x0 a multiple of 4; listed row-major, executed in XCD-contiguous ranges
  const LowpassTile* wide_tiles;
  int nwide;
  int wide_lds_bytes;
  const uint32_t* taps_sh;    // SegmentDev::kxs_off
};
hipError_t launch_lowpass(const LowpassArgs& a, int nframes, hipStream_t stream);
// the wide fast path of up to three planes (the Y, U and V of a batch) as ONE launch; only when every plane is served by
// wide tiles alone with the same vertical tap count (lowpass_mergeable)
struct LowpassMulti {
  LowpassArgs p[3];
  int n;
};
bool lowpass_mergeable(const LowpassArgs* a, int n);
hipError_t launch_lowpass_multi(const LowpassArgs* a, int n, int nframes, hipStream_t stream);

// ---- INTER_AREA shrink of the supersampled plane (t360_resize.hip) ----
struct ResizeArgs {
  const uint8_t* src;
  int64_t src_frame_bytes;
  int sstride, sw, sh;
  uint8_t* dst;
  int64_t dst_frame_bytes;
  int dstride, dw, dh;
  int linear, xmax;        // linear: the bilinear emulation of an enlarging INTER_AREA (tables re-used, see kernel)
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
