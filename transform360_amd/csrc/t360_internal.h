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
  int kxs_off, kxs_nd;     // shifted byte taps of the wide fast path: 4 variants x kWideTapStride dwords, of which
                           // the first kxs_nd are non-zero (kxs_nd = 0: not eligible); t360_lowpass.hip
};
constexpr int kWideTapStride = 12;  // dwords per shifted tap variant (zero padded)
constexpr int kWideMaxNd = 11;      // longest window, in dwords, the wide low-pass path instantiates
#ifndef T360_LP_TILE_W
#define T360_LP_TILE_W 512
#endif
constexpr int kWideTileW = T360_LP_TILE_W, kWideTileH = 32;  // 1024 / kWideTileW row groups of kWideTileW / 4 lanes

// One unit of low-pass work: a tile of one segment (tiles never straddle segments because the
// filter kernels change at segment borders).
struct LowpassTile {
  int seg;
  int x0, y0, w, h;  // absolute plane coordinates (stereo eye offset already applied)
};

// ---- LDS-tiled gather (t360_remap_tiled.hip, planned by t360_plan.cpp) ----
// Tile kinds of the work list.  A staged tile is 256 lanes x NPX pixels: lane = (column, band of 4 rows).
enum : int {
  kTileStaged32 = 0,  // 32x32 output px, 4 px per lane (8 bands)
  kTileStaged16 = 1,  // 16x16 output px, 1 px per lane (near the poles, where wider tiles do not fit the LDS)
  kTileDirect16 = 2,  // 16x16 output px gathered straight from global memory (the ~4 tiles around each pole whose
                      // source footprint spans a quadrant of longitudes)
  kTileStrip128 = 3,  // 128x8 output px, 4 px per lane (2 bands): ~330-byte source row fragments
  kTileWide64 = 4,    // 64x16 output px, 4 px per lane (4 bands)
  kTileWide128 = 5,   // 128x16 output px, 4 px per lane on 512 lanes (4 bands): workgroups of 8 waves only
  kTileWide256 = 6,   // 256x8 output px, 4 px per lane on 512 lanes (2 bands): workgroups of 8 waves only; ~490-byte
                      // source row fragments (5 lines for 4 of payload where a 128-wide tile fetches 3 for 2)
  kTileScatter = 7,   // 128 blocks of 4x4 output px anywhere in the plane, 4 px per lane on 512 lanes (lanes 4q..4q+3 of
                      // band b = block b*32 + q): the planner groups blocks by where their stencils lie in the SOURCE, so
                      // the tile's footprint is a compact, line-aligned source rectangle instead of the slanted band an
                      // output rectangle makes.  Block origins: 128 dwords (ox | oy << 16) behind the tile's row table.
};
enum : int {
  kTilePartial = 1,    // crosses the right/bottom plane edge: per-pixel bounds checks, byte stores
  kTileSeamShift = 2,  // staged x coordinates use (x >= W/2 ? x - W : x): the tile straddles the +-180 degree seam
  kTileFused = 4,      // low-pass fused tile (remap_fused_kernel): the DMA stages the RAW, dilated footprint and the
                       // workgroup filters it in LDS before it gathers; see "fused low-pass tiles" below
};

constexpr int kStageChunk = 16;   // bytes per staged chunk (one dwordx4 per DMA lane)
constexpr int kPieceChunks = 64;  // chunks per DMA instruction (one per lane): a 1 KiB "piece"
constexpr int kMaxPieces = 32;    // pieces of a tile's staged region (each wave of the workgroup moves up to 4)

// One unit of gather work.  The staged region of a tile is NOT a bounding box: the planner lists exactly the
// 16-byte source chunks the tile's stencils touch (the footprint of an output tile in the equirect source is a
// curved band, on the polar faces an annular sector) and packs the box rows back to back in LDS: row r occupies
// chunk positions row_pos[r] .. row_pos[r] + len[r] - 1.  Every pixel carries the LDS address of each of its
// stencil rows, so rows need no common pitch and HBM only delivers bytes that are used.
struct __attribute__((aligned(16))) TileDesc {
  int16_t ox, oy;     // output origin
  int16_t kind;       // kTile*
  int16_t flags;      // kTilePartial | kTileSeamShift
  int16_t pieces;     // 1 KiB DMA pieces per copy of the staged region
  int16_t rows;       // box rows (entries of the row table)
  int32_t fetched;    // distinct source chunks among the positions (statistics)
  int32_t pad[4];
};
static_assert(sizeof(TileDesc) == 32, "TileDesc must be 32 bytes");

// Pixel word of a staged tile:
//   bits 0..10   x of the stencil's left tap relative to the box origin (byte units, < 2048)
//   bits 11..18  box row of the stencil's top tap (< 256)
//   bits 19..28  sub-pixel phase
//   bit 31       pixel outside the plane (partial tile)
// LDS byte offset of the tap at stencil row k = row_base[row + k] * 16 + x, with the tile's row table
// (int16 per box row, in 16-byte chunks: LDS position of the row minus its first staged column).
//
// Per-tile tables live at FIXED strides, so a workgroup can fetch them from its tile index alone, in parallel with
// the descriptor (one memory round trip less in every workgroup's prologue):
//   pixel words   tlut   + tile * tile_words(ks, waves)    (a uint4 per lane; 16x16 tiles use its first word)
//   chunk table   chunks + tile * tile_chunk_dwords(max_pieces):  64 * max_pieces chunk entries (the first
//                 64 * pieces are meaningful), then the row table: 64 dwords = 128 int16
// pixel words per tile slot: a uint4 per lane (Lanczos4 plans hold 16x16 tiles only: one word per lane)
constexpr int tile_words(int ks, int waves) { return ks == 8 ? 256 : 256 * waves; }
constexpr int kScatterBlocks = 128;  // 4x4-px blocks of a scatter tile
constexpr int tile_chunk_dwords(int max_pieces, bool scatter = false) { return max_pieces * 64 + 64 + (scatter ? kScatterBlocks : 0); }
constexpr uint32_t kWordDead = 0x80000000u;
constexpr int kWordRowShift = 11, kWordFracShift = 19;
constexpr int kBoxMaxCols = 2048 / 16;  // chunk columns of a box
constexpr int kBoxMaxRows = 128;        // rows of a box: the row table is 64 dwords, one per lane of a wave
// Chunk table entry: source row (already wrapped) and 16-byte column (already wrapped) of one LDS position.
inline uint32_t chunk_entry(uint32_t sy, uint32_t cx) { return (sy << 12) | cx; }

// ---- fused low-pass tiles (t360_remap_tiled.hip: remap_fused_kernel; planned by t360_plan.cpp) ----
// For tiles whose source rows all lie in low-pass segments with fixed-point kernels of <= 7 horizontal and exactly 3
// vertical taps (BASELINE config 3: every row between +-66 degrees of latitude) the blurred plane never goes through HBM:
// the DMA stages the RAW footprint dilated by the kernel radius (R), the workgroup computes the blurred footprint (B) from
// it in LDS -- same arithmetic as lowpass_q8w_kernel: r = SUM kx_q8 * p, c = SUM ky_q8 * r, sat_u8((c + 32768) >> 16) --
// writes it IN PLACE into the same ring slot (after a barrier: every R byte has been read), and gathers from it.
// Per-tile tables of a fused plan, at a fixed stride (fused_chunk_dwords):
//   [R chunk table: 64 * max_pieces entries][B row table: 64 dwords][R row table: 64 dwords][run words: 512][wave info: 16]
// B row table: int16 per B row, as in an unfused plan.  R row table: int16 per R row t = 0 .. rows + 1 (source row
// y0 - 1 + t, clamped to the plane: BORDER_REPLICATE at the plane's top and bottom edge): LDS chunk position of the row
// minus its first staged chunk column, relative to the R box origin (one chunk left of the B box origin).
// Run word of lane l (one vertical run of <= kFusedMaxRun blurred dwords per lane and frame, all in ONE kernel band):
//   bits 0..8    dword column of the run inside the B box (byte x = 4 * column)
//   bits 9..15   first B row of the run
//   bits 16..19  rows in the run (1 .. kFusedMaxRun)
//   bit 20 / 21  the run is the plane's first / last dword column: the dword left / right of it is BORDER_REPLICATE, not
//                what the (wrapping) gather footprint holds there
//   bit 31       no run on this lane (never set in a plan: lanes without work of their own repeat a run of their wave --
//                the same bytes written twice -- so the filter phase needs no per-lane predicates)
// Wave info: dword w (0..7) = kernel index of wave w's runs (every wave's runs share one kernel: its taps sit in SGPRs),
// dword 8 = the longest run of the tile.
constexpr int kFusedMaxRun = 8;
constexpr int kFusedLanes = 512;
constexpr int fused_chunk_dwords(int max_pieces) { return max_pieces * 64 + 64 + 64 + kFusedLanes + 16; }
constexpr uint32_t kRunDead = 0x80000000u, kRunLeftEdge = 1u << 20, kRunRightEdge = 1u << 21;
// Packed taps of one fusable kernel (kFusedTapDwords dwords): the horizontal taps, zero-padded to 7 and centred, as the
// four byte-shifted variants of a 12-byte window that starts 4 bytes left of the output dword -- variant j (output byte
// j) holds tap k at window byte 1 + j + k: [v0.d0 v0.d1 | v1.d0 v1.d1 v1.d2 | v2.d0 v2.d1 v2.d2 | v3.d1 v3.d2] -- then
// the three vertical taps as plain integers.
constexpr int kFusedTapDwords = 16;  // 10 + 3, padded to a 64-byte s_load_dwordx16

// Weights re-packed for v_dot4, per sub-pixel phase of a KS x KS interpolation:
//   [KS*WIN signed high-byte dwords: w >> 8][KS*WIN unsigned low-byte dwords: w & 255]
// WIN = 4-byte windows per stencil row (2 for Lanczos4, else 1); bilinear fills bytes 0-1 of its window only.
// Nothing else: 16 / 32 / 128 bytes per phase, a whole number of 16-byte loads.  The bias of the signed half (pixels
// are fed as p - 128: 128 * SUM(w >> 8)) is recomputed by the kernel from the high bytes (a v_dot4 per dword, once per
// tile) -- round 4: it used to be a third load per pixel from a 48-byte entry, and the table gather is what a
// workgroup's prologue waits for (64 distinct entries per load instruction).
constexpr int pack_dwords(int ks) { return ks == 2 ? 4 : ks == 4 ? 8 : ks == 8 ? 32 : 0; }

}  // namespace t360
