// t360_plan.cpp -- init-time planning of the LDS-tiled gather (host side).
//
// From the scanned per-tile bounding boxes (tile_scan_kernel) build the tile work list:
//   32x32 output tile  -> staged through LDS when its source box (incl. stencil halo, x aligned
//                         down to 16 bytes) fits the per-workgroup staging budget;
//   otherwise its four 16x16 quadrants, each staged if it fits, else gathered directly
//                         (the few tiles touching a pole span a full quadrant of longitudes,
//                         SURVEY.md 7 H4).
// Tiles are emitted in output raster order; the kernel hands contiguous ranges to each XCD.
#include "t360_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace t360 {

namespace {

struct Box {
  int x0, cpr, y0, rows;
  bool seam_shift;
  bool empty;
  bool fits;
};

// b: {minx, maxx, minx_shifted, maxx_shifted, miny, maxy} of pixel-centre taps
Box make_box(const int* b, int halo_lo, int halo_hi, int max_chunks) {
  Box r{};
  r.empty = b[0] > b[1];
  if (r.empty) return r;
  const int w_raw = b[1] - b[0], w_shift = b[3] - b[2];
  r.seam_shift = w_shift < w_raw;
  const int xmin = (r.seam_shift ? b[2] : b[0]) - halo_lo;
  const int xmax = (r.seam_shift ? b[3] : b[1]) + halo_hi;
  r.x0 = xmin & ~(kStageChunk - 1);  // two's complement: floors negatives too
  r.cpr = (xmax - r.x0) / kStageChunk + 1;
  r.y0 = b[4] - halo_lo;
  r.rows = (b[5] + halo_hi) - r.y0 + 1;
  r.fits = r.cpr * kStageChunk <= kStageMaxCols && r.rows <= kStageMaxRows &&
           r.cpr * r.rows <= max_chunks;
  return r;
}

}  // namespace

bool build_gather_plan(const LutEntry* d_lut, int dw, int dh, int sw, int sh, int ksize, int max_box_bytes,
                       hipStream_t stream, GatherPlan* plan) {
  plan->valid = false;
  plan->ntiles = 0;
  if (!(ksize == 1 || ksize == 2 || ksize == 4 || ksize == 8)) return true;
  // taps of a ksize-wide stencil around the LUT's integer coordinate: -(ksize/2-1) .. +ksize/2
  // (nearest: the pixel itself)
  const int halo_lo = ksize == 1 ? 0 : ksize / 2 - 1, halo_hi = ksize == 1 ? 0 : ksize / 2;
  // Lanczos4 keeps 32 weight dwords per pixel in registers: one pixel per lane, 16x16 tiles only
  const bool only16 = ksize == 8;
  int max_chunks = max_box_bytes / kStageChunk;
  if (max_chunks > 256 * kStageChunksPerLane) max_chunks = 256 * kStageChunksPerLane;
  const int regions_x = (dw + 127) / 128, regions_y = (dh + 31) / 32;
  const size_t nregions = (size_t)regions_x * regions_y;
  const int per_region = kScanBoxes * 6;

  DeviceBuffer scan;
  if (!scan.reserve(nregions * per_region * sizeof(int))) return false;
  if (launch_tile_scan(d_lut, dw, dh, sw, scan.as<int>(), stream) != hipSuccess) return false;
  std::vector<int> boxes(nregions * per_region);
  if (hipMemcpyAsync(boxes.data(), scan.as<void>(), boxes.size() * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess)
    return false;
  if (hipStreamSynchronize(stream) != hipSuccess) return false;

  std::vector<TileDesc> tiles, direct;
  tiles.reserve(nregions * 4);
  int64_t tlut_words = 0, staged_bytes = 0;
  int n32 = 0, n16 = 0, nstrip = 0, ndirect = 0, nwide = 0;
  // 128x8 strips give ~330-byte row fragments (better for HBM) but measured ~4 % slower end to end
  // while the kernel is VALU-issue-bound; opt-in until that changes (DESIGN.md, round-1 notes)
  // 64x16 tiles replace pairs of 32x32 tiles unless they would stage more than wide_pct % of the pair's
  // bytes (T360_WIDE64=0 turns them off).  Measured on config 2: 6-7 % faster end to end at 150-1000 %
  // although more bytes go through LDS -- the ~160-byte row fragments and the halved number of
  // horizontal box borders are what the fabric and HBM see.
  const int wide_pct = getenv("T360_WIDE64") ? atoi(getenv("T360_WIDE64")) : 200;
  const bool wide64 = wide_pct > 0 && (ksize == 2 || ksize == 4);  // nearest has no halo to save, Lanczos4 is 16x16 only
  const int strips_mode = getenv("T360_STRIPS") ? atoi(getenv("T360_STRIPS")) : 0;  // 2: wherever a strip fits (experiment)
  const bool allow_strips = strips_mode > 0;
  const int pad_mode = getenv("T360_PAD") ? atoi(getenv("T360_PAD")) : 1;
  int cpr_hist[64] = {0};
  auto emit = [&](const Box& bx, int kind, int tox, int toy, int ew, int eh) {
    TileDesc t{};
    t.ox = (int16_t)tox;
    t.oy = (int16_t)toy;
    t.kind = (int16_t)kind;
    t.flags = (int16_t)((bx.seam_shift ? kTileSeamShift : 0) | ((tox + ew > dw || toy + eh > dh) ? kTilePartial : 0));
    t.x0 = bx.x0;
    t.y0 = bx.y0;
    int cpr = bx.cpr;
    if (pad_mode && kind != kTileDirect16) {
      // LDS row pitch = 16*cpr bytes = 4*cpr banks.  The lanes of one output row drift over 2-3 source
      // rows; with a pitch near a multiple of 128 bytes those rows land on the same banks (57 % of
      // the LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT).  Padding columns (never fetched
      // from HBM) put consecutive rows 8..24 banks apart.
      int want = cpr;
      if (pad_mode == 1)
        while ((want & 7) == 7 || (want & 7) == 0 || (want & 7) == 1) want++;
      else if (pad_mode == 2)
        while ((want & 7) != 4) want++;
      else
        while ((want & 7) < 3 || (want & 7) > 5) want++;
      if ((int64_t)want * bx.rows <= (int64_t)max_chunks && want * kStageChunk <= kStageMaxCols) cpr = want;
    }
    cpr_hist[cpr < 64 ? cpr : 63]++;
    t.cpr = (int16_t)cpr;
    t.cpr_src = (int16_t)bx.cpr;
    t.rows = (int16_t)bx.rows;
    t.tlut = (int32_t)tlut_words;
    if (kind == kTileDirect16) {
      ndirect++;
      direct.push_back(t);
      return;
    }
    tlut_words += kind == kTileStaged16 ? 256 : 1024;
    staged_bytes += (int64_t)cpr * kStageChunk * bx.rows;
    (kind == kTileStaged16 ? n16 : kind == kTileStaged32 ? n32 : kind == kTileWide64 ? nwide : nstrip)++;
    tiles.push_back(t);
  };
  // Emission order = execution order (each XCD gets a contiguous range).  Region rows are walked in
  // bands of `band` rows, column by column inside a band, so that vertically adjacent tiles -- whose
  // source boxes share the stencil halo and the rows the curved footprint adds -- run at the same time
  // on the same XCD and meet in its L2 (+2 % measured; T360_BAND=1 is plain raster order).
  const int band = getenv("T360_BAND") ? std::max(1, atoi(getenv("T360_BAND"))) : 4;
  for (int ry0 = 0; ry0 < regions_y; ry0 += band)
    for (int rx = 0; rx < regions_x; rx++)
      for (int ry = ry0; ry < std::min(ry0 + band, regions_y); ry++) {
      const int* b = &boxes[((size_t)ry * regions_x + rx) * per_region];
      const int ox = rx * 128, oy = ry * 32;
      // option A: four 128x8 strips (wide row fragments stream ~2x faster from HBM than the
      // ~100-byte fragments of 32x32 tiles); option B: four 32x32 tiles with 16x16 fallback.
      Box strip[4], tile[4];
      bool strips_ok = allow_strips && !only16;
      int64_t strip_bytes = 0, tile_bytes = 0;
      for (int k = 0; k < 4; k++) {
        strip[k] = make_box(b + 6 * k, halo_lo, halo_hi, max_chunks);
        tile[k] = make_box(b + 6 * (4 + k), halo_lo, halo_hi, max_chunks);
        if (only16) tile[k].fits = false;
        if (!strip[k].empty) {
          strips_ok = strips_ok && strip[k].fits;
          strip_bytes += (int64_t)strip[k].cpr * kStageChunk * strip[k].rows;
        }
        if (!tile[k].empty) {
          if (tile[k].fits) {
            tile_bytes += (int64_t)tile[k].cpr * kStageChunk * tile[k].rows;
          } else {
            for (int qd = 0; qd < 4; qd++) {
              const Box sub = make_box(b + 6 * (8 + 4 * k + qd), halo_lo, halo_hi, max_chunks);
              if (!sub.empty) tile_bytes += sub.fits ? (int64_t)sub.cpr * kStageChunk * sub.rows : (int64_t)1 << 20;
            }
          }
        }
      }
      // strips win unless they stage clearly more bytes (curved rows on the polar faces)
      if (strips_ok && (strip_bytes * 2 <= tile_bytes * 3 || strips_mode >= 2)) {
        for (int k = 0; k < 4; k++)
          if (!strip[k].empty) emit(strip[k], kTileStrip128, ox, oy + 8 * k, 128, 8);
        continue;
      }
      // option C: per half region (64x32 output px) two 64x16 tiles instead of two 32x32 tiles when they
      // stage fewer bytes; their boxes are the unions of the scanned 16x16 quadrant boxes
      bool half_done[2] = {false, false};
      if (wide64) {
        for (int h = 0; h < 2; h++) {
          const Box& ta = tile[2 * h];
          const Box& tb = tile[2 * h + 1];
          if (ta.empty || tb.empty || !ta.fits || !tb.fits) continue;
          Box w[2];
          bool ok = true;
          int64_t wide_bytes = 0;
          for (int v = 0; v < 2 && ok; v++) {
            int u[6] = {1 << 30, -(1 << 30), 1 << 30, -(1 << 30), 1 << 30, -(1 << 30)};
            for (int t2 = 0; t2 < 2; t2++)
              for (int qx = 0; qx < 2; qx++) {
                const int* q = b + 6 * (8 + 4 * (2 * h + t2) + 2 * v + qx);
                if (q[0] > q[1]) continue;  // empty quadrant (plane edge)
                u[0] = std::min(u[0], q[0]); u[1] = std::max(u[1], q[1]);
                u[2] = std::min(u[2], q[2]); u[3] = std::max(u[3], q[3]);
                u[4] = std::min(u[4], q[4]); u[5] = std::max(u[5], q[5]);
              }
            w[v] = make_box(u, halo_lo, halo_hi, max_chunks);
            ok = !w[v].empty && w[v].fits;
            if (ok) wide_bytes += (int64_t)w[v].cpr * kStageChunk * w[v].rows;
          }
          const int64_t sq_bytes = (int64_t)ta.cpr * kStageChunk * ta.rows + (int64_t)tb.cpr * kStageChunk * tb.rows;
          if (ok && wide_bytes * 100 <= sq_bytes * wide_pct) {
            emit(w[0], kTileWide64, ox + 64 * h, oy, 64, 16);
            emit(w[1], kTileWide64, ox + 64 * h, oy + 16, 64, 16);
            half_done[h] = true;
          }
        }
      }
      for (int k = 0; k < 4; k++) {
        if (tile[k].empty || half_done[k >> 1]) continue;
        if (tile[k].fits) {
          emit(tile[k], kTileStaged32, ox + 32 * k, oy, 32, 32);
          continue;
        }
        for (int qd = 0; qd < 4; qd++) {
          const Box sub = make_box(b + 6 * (8 + 4 * k + qd), halo_lo, halo_hi, max_chunks);
          if (sub.empty) continue;
          emit(sub, sub.fits ? kTileStaged16 : kTileDirect16, ox + 32 * k + (qd & 1) * 16, oy + (qd >> 1) * 16, 16, 16);
        }
      }
    }
  if (tiles.size() > 0x7fffffff || tlut_words > 0x7fffffff) return false;

  // staged tiles first, direct tiles behind them in the same buffer
  const size_t nstaged = tiles.size();
  tiles.insert(tiles.end(), direct.begin(), direct.end());
  if (!plan->tiles.reserve(tiles.size() * sizeof(TileDesc)) ||
      !plan->tlut.reserve((size_t)(tlut_words > 0 ? tlut_words : 1) * sizeof(uint32_t)))
    return false;
  if (hipMemcpyAsync(plan->tiles.as<void>(), tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice,
                     stream) != hipSuccess)
    return false;
  if (launch_tile_lut(d_lut, dw, dh, sw, plan->tiles.as<TileDesc>(), (int)nstaged, halo_lo,
                      plan->tlut.as<uint32_t>(), stream) != hipSuccess)
    return false;
  if (hipStreamSynchronize(stream) != hipSuccess) return false;
  plan->ntiles = (int)nstaged;
  plan->n32 = n32;
  plan->nstrip = nstrip;
  plan->n16 = n16;
  plan->ndirect = ndirect;
  plan->staged_bytes = staged_bytes;
  plan->valid = true;
  if (getenv("T360_VERBOSE"))
    printf("transform360: gather plan %dx%d <- %dx%d: %d staged tiles (%d strips 128x8, %d tiles 32x32, %d tiles 16x16) + "
           "%d direct, %.2f MB staged per plane (%.2fx the source plane)\n",
           dw, dh, sw, sh, plan->ntiles, nstrip, n32, n16, ndirect, staged_bytes / 1e6,
           (double)staged_bytes / ((double)sw * sh));
  if (getenv("T360_VERBOSE") && nwide) printf("transform360:   %d tiles 64x16\n", nwide);
  if (getenv("T360_VERBOSE")) {
    printf("transform360:   chunks per row (16 B each): ");
    for (int c = 0; c < 64; c++)
      if (cpr_hist[c]) printf("%d:%d ", c, cpr_hist[c]);
    printf("\n");
  }
  return true;
}

// Re-pack OpenCV's Q15 table of a ks x ks interpolation for v_dot4 (layout: t360_internal.h pack_dwords):
// window (r, w) holds taps 4w .. 4w+3 of stencil row r; bilinear's two taps sit in bytes 0-1.
void pack_weights(const std::vector<int16_t>& tab, int ks, std::vector<uint32_t>* out) {
  const int phases = kInterTabSize * kInterTabSize;
  const int stride = pack_dwords(ks);
  const int win = ks == 8 ? 2 : 1, nw = ks * win;
  out->assign((size_t)phases * stride, 0);
  for (int f = 0; f < phases; f++) {
    const int16_t* w = &tab[(size_t)f * ks * ks];
    uint32_t* o = &(*out)[(size_t)f * stride];
    int sum_hi = 0;
    for (int r = 0; r < ks; r++)
      for (int q = 0; q < win; q++) {
        uint32_t hi = 0, lo = 0;
        for (int c = 0; c < 4; c++) {
          const int tap = 4 * q + c;
          if (tap >= ks) break;
          const int v = w[r * ks + tap];
          const int h = v >> 8;   // arithmetic shift: floor, in [-128, 127]
          const int l = v & 255;  // v == h * 256 + l
          sum_hi += h;
          hi |= (uint32_t)(h & 255) << (8 * c);
          lo |= (uint32_t)l << (8 * c);
        }
        o[r * win + q] = hi;
        o[nw + r * win + q] = lo;
      }
    o[2 * nw] = (uint32_t)((1 << (kCoefBits - 1)) + 128 * 256 * sum_hi);
  }
}

void pack_cubic_weights(const std::vector<int16_t>& tab, std::vector<uint32_t>* out) { pack_weights(tab, 4, out); }

}  // namespace t360
