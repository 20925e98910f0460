// t360_internal.h -- types shared by the host side and the HIP kernels of libTransform360.
//
// Data layout in HBM:
//   * frames        : 8-bit planar, caller-described (pointer + row stride per plane)
//   * warp map      : float2 per output pixel, row-major -- the reference's warpMats_[idx]
//                     (reference VideoFrameTransform.h:148), kept for introspection/parity
//   * sample LUT    : one packed 8-byte entry per output pixel derived from the warp map the way
//                     cv::remap's RemapInvoker derives (XY, A) from a CV_32FC2 map: integer
//                     source coordinate + 10-bit sub-pixel phase
//   * weight tables : OpenCV's 1024-entry Q15 2-D coefficient tables (2x2 / 4x4 / 8x8 taps)
//   * low-pass work : one record per (segment, tile) + packed per-segment 1-D kernels
#pragma once

#include <stdint.h>

#include "Transform360/VideoFrameTransformHelper.h"

namespace t360 {

constexpr int kMaxMaps = 8;          // transformMatPlaneIndex values accepted (reference uses 0,1)
constexpr int kInterTabSize = 32;    // cv::INTER_TAB_SIZE (1/32 pixel phases)
constexpr int kInterBits = 5;        // cv::INTER_BITS
constexpr int kCoefBits = 15;        // cv::INTER_REMAP_COEF_BITS

// cv::BorderTypes values used on this path
enum Border : int { kBorderReplicate = 1, kBorderWrap = 3, kBorderReflect101 = 4, kBorderTransparent = 5 };

// One entry of the sample LUT.  For NEAREST ix/iy are the rounded source pixel, frac is 0.
struct __attribute__((aligned(8))) LutEntry {
  int16_t ix;     // saturate_cast<short>(round(x*32) >> 5)   (or round(x) for NEAREST)
  int16_t iy;
  uint16_t frac;  // (sy & 31) * 32 + (sx & 31)
  uint16_t pad;
};
static_assert(sizeof(LutEntry) == 8, "LutEntry must be 8 bytes");

// Per-pixel-independent inputs of the projection kernel; everything transcendental that does
// not depend on the pixel is evaluated once on the host with the host libm, exactly where the
// reference evaluates it per pixel with the same arguments.
struct MapGenParams {
  int map_w, map_h;  // scaled output size = warp map size (VideoFrameTransform.cpp:524-526)
  int in_w, in_h;    // input plane size
  int input_layout, output_layout;
  int input_stereo, output_stereo;
  int vflip;
  int interp;        // InterpolationAlg, selects the LUT quantisation
  int offcenter;     // any |fixed_cube_offcenter_*| > 1e-9 (VideoFrameTransform.cpp:1192-1194)
  int horizontal_offset;
  float expand_coef, input_expand_coef;
  float off_x, off_y, off_z;
  float rot[9];      // float rotation coefficients in the reference's evaluation order (:1240-1244)
  float hfov, vfov, yaw_deg, pitch_deg;  // FLAT_FIXED
  float input_pixel_width;               // VideoFrameTransform.cpp:528-531
  // EQUIRECT / BARREL / BARREL_SPLIT / EAC_32 outputs: the libm values the reference takes per
  // pixel (sinf/cosf of yaw and pitch, tan of the face coordinate) depend on the column OR the row
  // only; the host evaluates them with the same libm and the kernel looks them up.
  //   EQUIRECT, BARREL : col_tab[2j] = sinf(yaw_j), [2j+1] = cosf(yaw_j); row_tab likewise for pitch
  //   BARREL_SPLIT     : col_tab[2(vFace*map_w + j)], vFace in {0,1}; row_tab as above
  //   EAC_32           : col_tab[j] = warped face x, row_tab[i] = warped face y
  const float* col_tab;
  const float* row_tab;
};

// One low-pass segment (SegmentFilteringConfig + its kernels, VideoFrameTransform.h:25-38,150-159)
struct SegmentDev {
  int left, top, width, height;
  int kx_off, kx_len;  // into the packed tap arrays (int32 taps and float taps share offsets)
  int ky_off, ky_len;
  int fixed_point;     // 1: Q8 x Q8 integer path, 0: float path
  int kxp_off, kx_groups;  // packed byte taps of the fast low-pass path (kx_groups = 0: not eligible)
};

// One unit of low-pass work: a tile of one segment (tiles never straddle segments because the
// filter kernels change at segment borders).
struct LowpassTile {
  int seg;
  int x0, y0, w, h;  // absolute plane coordinates (stereo eye offset already applied)
};

// ---- LDS-tiled gather (t360_remap_tiled.hip) ----
// Tile kinds of the work list.
enum : int {
  kTileStaged32 = 0,  // 32x32 output px, 4 px per lane, source box staged through LDS
  kTileStrip128 = 3,  // 128x8 output px, 4 px per lane: ~330-byte source row fragments, which the
                      // memory system streams at ~6 TB/s where 96-byte fragments reach ~3 TB/s
                      // (tools/ubench/fragment_bw.hip)
  kTileWide64 = 4,    // 64x16 output px, 4 px per lane (lane = column, 4 bands of 4 rows): same pixels per
                      // workgroup as a 32x32 tile with half the horizontal box borders (halo + 16-byte alignment
                      // cost ~29 bytes per ~60-byte row of a 32x32 tile, the 3 halo rows only ~10 % of its height)
  kTileStaged16 = 1,  // 16x16 output px, 1 px per lane, source box staged through LDS
  kTileDirect16 = 2,  // 16x16 output px, gathers straight from global memory (box too large)
};
enum : int {
  kTilePartial = 1,    // crosses the right/bottom plane edge: per-pixel bounds checks, byte stores
  kTileSeamShift = 2,  // box x coordinates use (x >= W/2 ? x - W : x)
};

constexpr int kStageChunk = 16;                 // bytes per staged chunk (one dwordx4 per lane)
constexpr int kStageChunksPerLane = 4;          // max chunks a lane fetches per frame
constexpr int kStageMaxBytes = 256 * kStageChunksPerLane * kStageChunk;  // 16 KiB box per tile
constexpr int kStageMaxCols = 1024;             // 10-bit box column in the tile LUT word
constexpr int kStageMaxRows = 256;              // 8-bit box row

// One unit of gather work.
struct __attribute__((aligned(16))) TileDesc {
  int16_t ox, oy;     // output origin
  int16_t kind;       // kTile*
  int16_t flags;      // kTilePartial | kTileSeamShift
  int32_t x0, y0;     // source box origin incl. stencil halo; x0 is a multiple of 16 (may be < 0)
  int16_t cpr;        // 16-byte chunks per box row (row pitch in LDS = cpr * 16 bytes)
  int16_t rows;       // box rows
  int32_t tlut;       // first word of this tile in the box-relative LUT
  int16_t cpr_src;    // chunks per row that hold source bytes; columns cpr_src..cpr-1 are LDS padding that
                      // moves consecutive rows onto different banks (their lanes re-read chunk 0)
  int16_t pad16;
  int32_t pad;
};
static_assert(sizeof(TileDesc) == 32, "TileDesc must be 32 bytes");

// Bicubic weights re-packed for v_dot4: per phase 12 dwords
//   [0..3]  high bytes (signed)  of the 4 taps of rows 0..3:  w >> 8
//   [4..7]  low bytes (unsigned) of the 4 taps of rows 0..3:  w & 255
//   [8]     rounding + bias constant: 16384 + 128*256*SUM(w >> 8)   (pixels are fed as p-128)
//   [9..11] padding
constexpr int kCubicPackDwords = 12;
// dwords per sub-pixel phase of the dot4-packed weight table of a KS x KS interpolation:
// [KS*WIN signed high-byte dwords][KS*WIN unsigned low-byte dwords][bias][padding to 16 bytes],
// WIN = 4-byte windows per stencil row (2 for Lanczos4, else 1).  Bilinear fills bytes 0-1 only.
constexpr int pack_dwords(int ks) { return ks == 2 ? 8 : ks == 4 ? kCubicPackDwords : ks == 8 ? 36 : 0; }

}  // namespace t360
