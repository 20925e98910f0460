// t360_plan.h -- init-time planning of the LDS-tiled gather for one map (see t360_plan.cpp).
//
// Pure host C++ (no HIP): the planner works on a host copy of the sample LUT, so it can be unit-tested
// and simulated on a machine without a GPU (tests/plan_sim.py).
#pragma once

#include <stdint.h>

#include <vector>

#include "t360_internal.h"

namespace t360 {

// What the planner needs to know about a plane's segmented low-pass to FUSE it into the gather (t360_internal.h "fused
// low-pass tiles"; MONO inputs only).  row_kid[y] = index of the (fusable) kernel every pixel of source row y is filtered
// with, or -1 (the row's segments have different or unfusable kernels, or do not cover it); segs = the segment
// rectangles, for the list of segments some UNFUSED tile still reads blurred pixels of.
struct FuseSegment {
  int left, top, width, height;
};
struct FuseInfo {
  std::vector<int16_t> row_kid;
  std::vector<FuseSegment> segs;
};

struct PlanOptions {
  int ks = 4;            // taps per axis of the interpolation: 1, 2, 4, 8
  int waves = 4;         // waves per workgroup of the gather kernel: 4, or 8 (then 128x16 tiles of 512 lanes exist)
  int max_pieces = 12;   // largest staged region of one tile, in 1 KiB DMA pieces (<= kMaxPieces)
  int wide_pct = 200;    // 64x16 tiles replace a pair of 32x32 tiles unless they fetch more than this % of the pair
  int strip_pct = 0;     // > 0: 128x8 strips replace the region's other tiles when they fetch <= this % of them
  int wide256_pct = 0;   // > 0 (workgroups of 8 waves): four 256x8 tiles replace a 256x32 region's other tiles when they
                         // cost <= this % of them
  bool cost_lines = false;  // shapes are compared by the distinct 128-byte lines of their footprints, not by chunks
  int scatter = 0;       // > 0 (workgroups of 8 waves, bilinear / bicubic, plane sides multiples of 4): output regions whose
                         // 4x4 blocks have compact stencils are cut into SCATTER tiles -- blocks grouped by source position
                         // in strips of `scatter` 128-byte lines -- instead of output rectangles (kTileScatter)
  int band = 4;          // order 0: region rows walked column by column (execution order, see t360_plan.cpp)
  int order = 2;         // execution order of the tiles: 0 bands of region rows (below), 1 raster, 2 Z-order of 64x16 cells
  int row_pad = 0;       // > 0: up to this many padding chunks behind a staged row (LDS bank spreading)
  int row_align = 8;     // LDS chunk position of a staged row == its source chunk column + skew * row, modulo this
                         // (1: rows packed back to back)
  bool row_search = true;  // per tile, the skew with the fewest modelled bank conflicts
  bool model_dual = false;   // bank model of ds_read_b64 on two copies instead of two ds_read_b32 on one
  int model_b_shift = 0;     // extra byte offset of copy B in the bank model (tests/plan_sim.py)
  bool model_stats = false;  // fill PlanStats::lds_cycles_model (tests/plan_sim.py)
  const FuseInfo* fuse = nullptr;  // != nullptr (workgroups of 8 waves, bilinear / bicubic): tiles that can filter their own
                                   // footprint go to the fused work list (HostGatherPlan::f*)
};

struct PlanStats {
  int n_strip = 0, n_wide = 0, n_sq = 0, n_16 = 0, n_direct = 0, n_wide128 = 0, n_wide256 = 0, n_scatter = 0;
  int64_t fetched_bytes = 0;   // distinct source chunks fetched per frame x 16 (HBM/L2 -> LDS, one copy)
  int64_t lds_bytes = 0;       // LDS positions per frame x 16 (incl. holes), one copy
  int64_t direct_pixels = 0;
  int64_t line_bytes = 0;      // model_stats: bytes of the distinct 128-byte lines each tile touches, summed
  int64_t lds_cycles_model = 0;  // modelled ds_read_b64 LDS cycles (32-lane groups x stencil rows), summed over the tiles
  int pieces_hist[33] = {0};   // staged tiles per size (1 KiB pieces per copy)
  int n_fused = 0;             // fused low-pass tiles (their bytes are NOT in the sums above)
  int64_t fused_raw_bytes = 0;      // raw (dilated) chunks fused tiles stage per frame x 16
  int64_t fused_blurred_bytes = 0;  // blurred dwords they compute per frame x 4
  int64_t fused_run_slots = 0;      // lanes x longest run, summed (occupancy of the filter phase = blurred / 4 / this)
};

struct HostGatherPlan {
  std::vector<TileDesc> tiles;    // staged tiles in execution order, then the direct tiles
  int ntiles = 0, ndirect = 0, ndirect_top = 0;  // direct tiles: those of the upper half of the plane first
  std::vector<uint32_t> tlut;     // pixel words, lane order (tile_word())
  std::vector<uint32_t> chunks;   // per staged tile 64 * pieces entries: chunk_entry()
  bool scatter = false;           // the chunk tables carry 128 block origins per tile (tile_chunk_dwords(.., true))
  // fused low-pass tiles (PlanOptions::fuse), a work list of their own in execution order: descriptors, pixel words at the
  // stride of the unfused list, tables at fused_chunk_dwords(max_pieces)
  std::vector<TileDesc> ftiles;
  std::vector<uint32_t> ftlut, fchunks;
  int nftiles = 0;
  std::vector<uint8_t> seg_needed;  // per FuseInfo segment: an unfused (staged or direct) tile reads its blurred pixels
  PlanStats stats;
};

// lut: dw x dh entries (host memory); source plane sw x sh with sw % 16 == 0 (the DMA path's precondition:
// a staged chunk never straddles the +-180 degree seam).
bool plan_gather(const LutEntry* lut, int dw, int dh, int sw, int sh, const PlanOptions& opt, HostGatherPlan* out);

// Re-pack OpenCV's Q15 table of a ks x ks interpolation for v_dot4 (layout: t360_internal.h pack_dwords)
void pack_weights(const std::vector<int16_t>& q15_table, int ks, std::vector<uint32_t>* out);

}  // namespace t360
