// tests/plan_sim/plan_sim.cpp -- the product's gather planner (transform360_amd/csrc/t360_plan.cpp) built
// for the host, so that tile shapes, fetched bytes and modelled LDS bank conflicts can be compared offline
// (tests/plan_sim/plan_sim.py feeds it the oracle's LUT).  Development tool, not part of the library.
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <vector>

#include "t360_filtercfg.h"
#include "t360_plan.h"

extern "C" int t360_plan_sim(const t360::LutEntry* lut, int dw, int dh, int sw, int sh, int ks, int max_pieces,
                             int wide_pct, int strip_pct, int row_pad, int unused, long long* stats /*[80]*/) {
  t360::PlanOptions o;
  o.ks = ks;
  o.max_pieces = max_pieces;
  o.wide_pct = wide_pct;
  o.strip_pct = strip_pct;
  o.row_pad = row_pad & 255;
  o.row_align = ((row_pad >> 8) & 255) ? ((row_pad >> 8) & 255) : 1;
  o.row_search = (row_pad >> 16) != 0;
  o.model_b_shift = unused;
  o.model_stats = true;
  o.waves = (row_pad >> 24) ? 8 : 4;
  t360::HostGatherPlan plan;
  if (!t360::plan_gather(lut, dw, dh, sw, sh, o, &plan)) return 0;
  const t360::PlanStats& s = plan.stats;
  stats[0] = s.n_strip + s.n_wide128 * 2 + s.n_wide256 * 2; stats[1] = s.n_wide; stats[2] = s.n_sq; stats[3] = s.n_16; stats[4] = s.n_direct;
  stats[5] = s.fetched_bytes; stats[6] = s.lds_bytes; stats[7] = s.direct_pixels; stats[8] = s.line_bytes;
  stats[9] = (long long)plan.chunks.size() * 4; stats[10] = (long long)plan.tlut.size() * 4;
  int maxp = 0;
  for (int i = 0; i < plan.ntiles; i++) maxp = plan.tiles[i].pieces > maxp ? plan.tiles[i].pieces : maxp;
  stats[11] = maxp;
  for (int i = 0; i < 33; i++) stats[12 + i] = s.pieces_hist[i];
  // FNV-1a over the plan's three tables: two builds of the planner that agree here emit the same plan
  auto fnv = [](const void* p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return (long long)h;
  };
  stats[50] = fnv(plan.tiles.data(), plan.tiles.size() * sizeof(plan.tiles[0]));
  stats[51] = fnv(plan.tlut.data(), plan.tlut.size() * 4);
  stats[52] = fnv(plan.chunks.data(), plan.chunks.size() * 4);
  return 1;
}

// the staged tiles of the plan in execution order: out[4*i] = kind, pieces, ox, oy (returns the tile count, <= cap)
extern "C" int t360_plan_tiles(const t360::LutEntry* lut, int dw, int dh, int sw, int sh, int ks, int max_pieces, int waves,
                               int* out, int cap) {
  t360::PlanOptions o;
  o.ks = ks;
  o.max_pieces = max_pieces;
  o.waves = waves;
  t360::HostGatherPlan plan;
  if (!t360::plan_gather(lut, dw, dh, sw, sh, o, &plan)) return -1;
  int n = 0;
  for (int i = 0; i < plan.ntiles && n < cap; i++, n++) {
    out[4 * n] = plan.tiles[i].kind;
    out[4 * n + 1] = plan.tiles[i].pieces;
    out[4 * n + 2] = plan.tiles[i].ox;
    out[4 * n + 3] = plan.tiles[i].oy;
  }
  return n;
}

// CPU emulation of the gather THROUGH the plan: stage every tile's chunk table into a fake LDS, look every pixel's
// stencil rows up the way the kernel does (pixel word -> row table -> LDS address) and compare the bytes with the
// source sampled directly from the LUT (BORDER_WRAP on both axes).  Also: every output pixel is covered exactly once.
// Returns the number of violations (0 = the plan is sound); prints the first few.
extern "C" long long t360_plan_verify(const t360::LutEntry* lut, int dw, int dh, int sw, int sh, int ks, int max_pieces, int waves,
                                      const unsigned char* src /* sw x sh, stride sw */) {
  using namespace t360;
  PlanOptions o;
  o.ks = ks;
  o.max_pieces = max_pieces;
  o.waves = waves & 0xff;
  o.wide256_pct = (waves >> 8) & 0xfff;   // bits 8..19 of `waves`: PlanOptions::wide256_pct
  o.cost_lines = ((waves >> 20) & 1) != 0;
  o.scatter = (waves >> 24) & 0xf;        // bits 24..27: PlanOptions::scatter (source strip width in lines)
  waves &= 0xff;
  HostGatherPlan plan;
  if (!plan_gather(lut, dw, dh, sw, sh, o, &plan)) return -1;
  auto wrapi = [](int v, int n) { v %= n; return v < 0 ? v + n : v; };
  const int lo = ks == 1 ? 0 : ks / 2 - 1;
  const int mp = max_pieces < kMaxPieces ? max_pieces : kMaxPieces;
  const size_t cstride = (size_t)tile_chunk_dwords(mp, plan.scatter);
  const size_t wstride = (size_t)tile_words(ks, waves);
  const int per_lane = ks == 8 ? 1 : 4;
  std::vector<unsigned char> cover((size_t)dw * dh, 0);
  std::vector<unsigned char> lds;
  long long bad = 0;
  auto complain = [&](const char* what, int tile, int px, int py) {
    if (bad++ < 8) printf("plan_verify: %s (tile %d, output pixel %d,%d)\n", what, tile, px, py);
  };
  for (int ti = 0; ti < plan.ntiles; ti++) {
    const TileDesc& t = plan.tiles[(size_t)ti];
    const uint32_t* tc = &plan.chunks[(size_t)ti * cstride];
    const uint32_t* words = &plan.tlut[(size_t)ti * wstride];
    if (t.pieces <= 0 || t.pieces > max_pieces) complain("piece count outside the budget", ti, t.ox, t.oy);
    lds.assign((size_t)t.pieces * 1024, 0);
    for (int pos = 0; pos < t.pieces * kPieceChunks; pos++) {
      const uint32_t e = tc[pos];
      const int sy = (int)(e >> 12), cx = (int)(e & 4095u);
      if (sy >= sh || (cx + 1) * kStageChunk > sw) {
        complain("chunk entry outside the source plane", ti, t.ox, t.oy);
        continue;
      }
      memcpy(&lds[(size_t)pos * kStageChunk], src + (size_t)sy * sw + (size_t)cx * kStageChunk, kStageChunk);
    }
    auto row_base = [&](int r) { return (int)(int16_t)(tc[(size_t)mp * kPieceChunks + (size_t)(r >> 1)] >> (16 * (r & 1))); };
    const uint32_t* origins = tc + (size_t)mp * kPieceChunks + 64;   // scatter tiles: ox | oy << 16 per 4x4 block
    int w = 0, h = 0, lanes = 256, npx = 4;
    switch (t.kind) {
      case kTileStaged32: w = 32; h = 32; break;
      case kTileStaged16: w = 16; h = 16; npx = 1; break;
      case kTileStrip128: w = 128; h = 8; break;
      case kTileWide64: w = 64; h = 16; break;
      case kTileWide128: w = 128; h = 16; lanes = 512; break;
      case kTileWide256: w = 256; h = 8; lanes = 512; break;
      case kTileScatter: w = 128; h = 16; lanes = 512; break;
      default: complain("unknown tile kind", ti, t.ox, t.oy); continue;
    }
    if (t.kind == kTileScatter || (npx == 4 && !(t.flags & kTilePartial))) {
      // tiles that take the kernel's dword store: after the 4x4 byte transpose a lane stores pixels that four lanes
      // computed, so every quad of lanes must be entirely live or (scatter tiles only) entirely dead
      for (int q = 0; q < lanes / 4; q++) {
        int live = 0;
        for (int l = 0; l < 4; l++)
          for (int p = 0; p < 4; p++) live += (words[(size_t)(4 * q + l) * per_lane + p] >> 31) == 0;
        if (live != 16 && !(live == 0 && t.kind == kTileScatter)) complain("partly live quad in a dword-stored tile", ti, t.ox, t.oy);
      }
    }
    for (int tid = 0; tid < lanes; tid++)
      for (int p = 0; p < npx; p++) {
        int px, py;
        bool scatter_dead = false;
        if (t.kind == kTileScatter) {
          // lanes 4q..4q+3 of band b hold block b*32 + q: column tid & 3, rows 0..3 (the kernel's out_pos())
          const int q = (tid >> 7) * 32 + ((tid & 127) >> 2);
          const uint32_t o = origins[q];
          px = (int)(o & 0xffffu) + (tid & 3);
          py = (int)(o >> 16) + p;
          // a block that is not there has dead words and (by convention) origin 0: tell it from a live block at (0, 0)
          scatter_dead = (words[(size_t)tid * per_lane + p] >> 31) != 0;
        } else if (npx == 4) {
          px = t.ox + tid % w;
          py = t.oy + (tid / w) * 4 + p;
        } else {
          px = t.ox + (tid & 15);
          py = t.oy + (tid >> 4);
        }
        const uint32_t word = words[(size_t)tid * per_lane + p];
        if (scatter_dead) continue;
        const bool inside = t.kind == kTileScatter ? (px < dw && py < dh) : (px < dw && py < dh && py < t.oy + h);
        if (!inside) {
          if (!(word >> 31)) complain("live pixel word outside the plane", ti, px, py);
          continue;
        }
        if (word >> 31) {
          complain("dead pixel word inside the plane", ti, px, py);
          continue;
        }
        cover[(size_t)py * dw + px]++;
        const LutEntry& e = lut[(size_t)py * dw + px];
        const int x = (int)(word & 2047u), row = (int)((word >> kWordRowShift) & 255u), frac = (int)((word >> kWordFracShift) & 1023u);
        if (ks != 1 && frac != (int)e.frac) complain("phase differs from the LUT", ti, px, py);
        for (int k = 0; k < ks; k++) {
          const int off = row_base(row + k) * kStageChunk + x;
          for (int c = 0; c < ks; c++) {
            const unsigned char want = src[(size_t)wrapi((int)e.iy - lo + k, sh) * sw + (size_t)wrapi((int)e.ix - lo + c, sw)];
            if (off + c < 0 || (size_t)(off + c) >= lds.size() || lds[(size_t)(off + c)] != want) {
              complain("staged byte differs from the source tap", ti, px, py);
              k = ks;
              break;
            }
          }
        }
      }
  }
  for (int i = 0; i < plan.ndirect; i++) {
    const TileDesc& t = plan.tiles[(size_t)plan.ntiles + (size_t)i];
    for (int y = t.oy; y < t.oy + 16 && y < dh; y++)
      for (int x = t.ox; x < t.ox + 16 && x < dw; x++) cover[(size_t)y * dw + x]++;
  }
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++)
      if (cover[(size_t)y * dw + x] != 1) complain("output pixel not covered exactly once", -1, x, y);
  return bad;
}

// pack_weights() of the planner for a Q15 table of phases x ks x ks shorts; out must hold phases * pack_dwords(ks) dwords
extern "C" int t360_pack_weights(const short* tab, int ks, unsigned* out) {
  std::vector<int16_t> t(tab, tab + (size_t)t360::kInterTabSize * t360::kInterTabSize * ks * ks);
  std::vector<uint32_t> v;
  t360::pack_weights(t, ks, &v);
  memcpy(out, v.data(), v.size() * sizeof(uint32_t));
  return (int)v.size();
}

// ---- fused low-pass tiles (round 6) ------------------------------------------------------------------------------------
// CPU emulation of remap_fused_kernel THROUGH the plan: every fused tile's R chunk table is staged from the RAW source into a
// fake LDS slot, every lane's run is filtered with the kernel's integer arithmetic (aligned dwords, the byte-shifted tap
// variants, r = SUM kx*p, c = SUM ky*r, (c + 32768) >> 16 saturated) out of the R rows the R row table names, the blurred
// dwords are written IN PLACE at the B positions, and every pixel's stencil bytes found there are compared with `blurred`
// (the whole plane filtered by the oracle).  The UNFUSED tiles of the same plan are staged from a copy of `blurred` in which
// every segment the plan does NOT list as needed is inverted: a tap that lands there is an error.  Also: every output
// pixel is covered exactly once over both work lists.  Returns the number of violations; stats[0..3] = fused tiles, unfused
// staged tiles, direct tiles, needed segments.
extern "C" long long t360_plan_verify_fused(const t360::LutEntry* lut, int dw, int dh, int sw, int sh, int ks, int max_pieces,
                                            const unsigned char* src, const unsigned char* blurred, const int* seg_rects,
                                            int nsegs, const short* row_kid, const unsigned* taps, long long* stats) {
  using namespace t360;
  FuseInfo fi;
  fi.row_kid.assign(row_kid, row_kid + sh);
  for (int i = 0; i < nsegs; i++) fi.segs.push_back({seg_rects[4 * i], seg_rects[4 * i + 1], seg_rects[4 * i + 2], seg_rects[4 * i + 3]});
  PlanOptions o;
  o.ks = ks;
  o.max_pieces = max_pieces;
  o.waves = 8;
  o.fuse = &fi;
  HostGatherPlan plan;
  if (!plan_gather(lut, dw, dh, sw, sh, o, &plan)) return -1;
  auto wrapi = [](int v, int n) { v %= n; return v < 0 ? v + n : v; };
  const int lo = ks == 1 ? 0 : ks / 2 - 1;
  const int mp = max_pieces < kMaxPieces ? max_pieces : kMaxPieces;
  const size_t wstride = (size_t)tile_words(ks, 8);
  long long bad = 0;
  auto complain = [&](const char* what, int tile, int px, int py) {
    if (bad++ < 8) printf("plan_verify_fused: %s (tile %d, output pixel %d,%d)\n", what, tile, px, py);
  };
  std::vector<unsigned char> cover((size_t)dw * dh, 0);
  std::vector<unsigned char> lds;
  auto shape_of = [&](const TileDesc& t, int* w, int* h, int* lanes, int* npx) {
    *lanes = 256; *npx = 4;
    switch (t.kind) {
      case kTileStaged32: *w = 32; *h = 32; return true;
      case kTileStaged16: *w = 16; *h = 16; *npx = 1; return true;
      case kTileStrip128: *w = 128; *h = 8; return true;
      case kTileWide64: *w = 64; *h = 16; return true;
      case kTileWide128: *w = 128; *h = 16; *lanes = 512; return true;
      default: return false;
    }
  };
  // pixels of a tile against `want_plane`, their stencil bytes looked up in `lds` through row table `rowtab`
  auto check_pixels = [&](const TileDesc& t, int ti, const uint32_t* words, const uint32_t* rowtab, const unsigned char* want_plane) {
    int w, h, lanes, npx;
    if (!shape_of(t, &w, &h, &lanes, &npx)) { complain("unknown tile kind", ti, t.ox, t.oy); return; }
    auto row_base = [&](int r) { return (int)(int16_t)(rowtab[(size_t)(r >> 1)] >> (16 * (r & 1))); };
    for (int tid = 0; tid < lanes; tid++)
      for (int p = 0; p < npx; p++) {
        const int px = npx == 4 ? t.ox + tid % w : t.ox + (tid & 15), py = npx == 4 ? t.oy + (tid / w) * 4 + p : t.oy + (tid >> 4);
        const uint32_t word = words[(size_t)tid * 4 + p];
        if (!(px < dw && py < dh && py < t.oy + h)) {
          if (!(word >> 31)) complain("live pixel word outside the plane", ti, px, py);
          continue;
        }
        if (word >> 31) { complain("dead pixel word inside the plane", ti, px, py); continue; }
        cover[(size_t)py * dw + px]++;
        const LutEntry& e = lut[(size_t)py * dw + px];
        const int x = (int)(word & 2047u), row = (int)((word >> kWordRowShift) & 255u);
        for (int k = 0; k < ks; k++) {
          const int off = row_base(row + k) * kStageChunk + x;
          for (int c = 0; c < ks; c++) {
            const unsigned char want = want_plane[(size_t)wrapi((int)e.iy - lo + k, sh) * sw + (size_t)wrapi((int)e.ix - lo + c, sw)];
            if (off + c < 0 || (size_t)(off + c) >= lds.size() || lds[(size_t)(off + c)] != want) {
              complain("byte under a stencil tap differs from the blurred plane", ti, px, py);
              k = ks;
              break;
            }
          }
        }
      }
  };
  // ---- fused tiles ----
  const size_t fcs = (size_t)fused_chunk_dwords(mp);
  for (int ti = 0; ti < plan.nftiles; ti++) {
    const TileDesc& t = plan.ftiles[(size_t)ti];
    const uint32_t* tc = &plan.fchunks[(size_t)ti * fcs];
    const uint32_t* btab = tc + (size_t)mp * kPieceChunks;
    const uint32_t* rtab = btab + 64;
    const uint32_t* runs = rtab + 64;
    const uint32_t* winfo = runs + kFusedLanes;
    if (!(t.flags & kTileFused) || t.pieces <= 0 || t.pieces > mp) complain("fused tile descriptor", ti, t.ox, t.oy);
    lds.assign((size_t)t.pieces * 1024, 0);
    for (int pos = 0; pos < t.pieces * kPieceChunks; pos++) {
      const uint32_t e = tc[pos];
      const int sy = (int)(e >> 12), cx = (int)(e & 4095u);
      if (sy >= sh || (cx + 1) * kStageChunk > sw) { complain("chunk entry outside the source plane", ti, t.ox, t.oy); continue; }
      memcpy(&lds[(size_t)pos * kStageChunk], src + (size_t)sy * sw + (size_t)cx * kStageChunk, kStageChunk);
    }
    auto rbase = [&](int tr) { return (int)(int16_t)(rtab[(size_t)(tr >> 1)] >> (16 * (tr & 1))); };
    auto bbase = [&](int r) { return (int)(int16_t)(btab[(size_t)(r >> 1)] >> (16 * (r & 1))); };
    const int ni = (int)winfo[8];
    std::vector<std::pair<int, uint32_t>> writes;
    for (int lane = 0; lane < kFusedLanes; lane++) {
      const uint32_t rw = runs[lane];
      if (rw >> 31) { complain("idle lane in a fused tile", ti, t.ox, t.oy); continue; }
      const int dcol = (int)(rw & 511u), r0 = (int)((rw >> 9) & 127u), len = (int)((rw >> 16) & 15u);
      if (len < 1 || len > ni || r0 + len > t.rows) { complain("run word", ti, t.ox, t.oy); continue; }
      const uint32_t* tp = taps + (size_t)winfo[lane / 64] * kFusedTapDwords;
      const uint32_t V[4][3] = {{tp[0], tp[1], 0}, {tp[2], tp[3], tp[4]}, {tp[5], tp[6], tp[7]}, {0, tp[8], tp[9]}};
      uint32_t res[3][4] = {{0}};
      for (int i = 0; i < len + 2; i++) {
        const int a = rbase(r0 + i) * kStageChunk + kStageChunk + 4 * dcol - 4;
        uint32_t d[3] = {0, 0, 0};
        if (a < 0 || (size_t)(a + 12) > lds.size()) { complain("filter read outside the slot", ti, t.ox, t.oy); break; }
        memcpy(d, &lds[(size_t)a], 12);
        if (rw & kRunLeftEdge) d[0] = (d[1] & 255u) * 0x01010101u;
        if (rw & kRunRightEdge) d[2] = (d[1] >> 24) * 0x01010101u;
        for (int j = 0; j < 4; j++) {
          uint32_t acc = 0;
          for (int q = 0; q < 3; q++)
            for (int b = 0; b < 4; b++) acc += ((d[q] >> (8 * b)) & 255u) * ((V[j][q] >> (8 * b)) & 255u);
          res[i % 3][j] = acc;
        }
        if (i >= 2) {
          uint32_t out = 0;
          for (int j = 0; j < 4; j++) {
            uint32_t c = (1u << 15) + tp[10] * res[(i - 2) % 3][j] + tp[11] * res[(i - 1) % 3][j] + tp[12] * res[i % 3][j];
            c >>= 16;
            out |= (c > 255u ? 255u : c) << (8 * j);
          }
          writes.push_back({bbase(r0 + i - 2) * kStageChunk + 4 * dcol, out});
        }
      }
    }
    for (const auto& wv : writes) {
      if (wv.first < 0 || (size_t)(wv.first + 4) > lds.size()) { complain("blurred dword outside the slot", ti, t.ox, t.oy); continue; }
      memcpy(&lds[(size_t)wv.first], &wv.second, 4);
    }
    check_pixels(t, ti, &plan.ftlut[(size_t)ti * wstride], btab, blurred);
  }
  // ---- unfused tiles: staged from the blurred plane with every segment nobody asked for inverted ----
  std::vector<unsigned char> partial(blurred, blurred + (size_t)sw * sh);
  int needed = 0;
  for (int i = 0; i < nsegs; i++) {
    const bool need = (size_t)i < plan.seg_needed.size() && plan.seg_needed[(size_t)i];
    needed += need;
    if (need) continue;
    for (int y = seg_rects[4 * i + 1]; y < seg_rects[4 * i + 1] + seg_rects[4 * i + 3]; y++)
      for (int x = seg_rects[4 * i]; x < seg_rects[4 * i] + seg_rects[4 * i + 2]; x++) partial[(size_t)y * sw + x] ^= 0xff;
  }
  const size_t cstride = (size_t)tile_chunk_dwords(mp, false);
  for (int ti = 0; ti < plan.ntiles; ti++) {
    const TileDesc& t = plan.tiles[(size_t)ti];
    const uint32_t* tc = &plan.chunks[(size_t)ti * cstride];
    lds.assign((size_t)t.pieces * 1024, 0);
    for (int pos = 0; pos < t.pieces * kPieceChunks; pos++) {
      const uint32_t e = tc[pos];
      const int sy = (int)(e >> 12), cx = (int)(e & 4095u);
      if (sy >= sh || (cx + 1) * kStageChunk > sw) { complain("chunk entry outside the source plane", ti, t.ox, t.oy); continue; }
      memcpy(&lds[(size_t)pos * kStageChunk], partial.data() + (size_t)sy * sw + (size_t)cx * kStageChunk, kStageChunk);
    }
    check_pixels(t, 100000 + ti, &plan.tlut[(size_t)ti * wstride], tc + (size_t)mp * kPieceChunks, blurred);
  }
  for (int i = 0; i < plan.ndirect; i++) {
    const TileDesc& t = plan.tiles[(size_t)plan.ntiles + (size_t)i];
    for (int y = t.oy; y < t.oy + 16 && y < dh; y++)
      for (int x = t.ox; x < t.ox + 16 && x < dw; x++) {
        cover[(size_t)y * dw + x]++;
        const LutEntry& e = lut[(size_t)y * dw + x];
        for (int k = 0; k < ks; k++)
          for (int c = 0; c < ks; c++) {
            const size_t at = (size_t)wrapi((int)e.iy - lo + k, sh) * sw + (size_t)wrapi((int)e.ix - lo + c, sw);
            if (partial[at] != blurred[at]) { complain("direct tile reads a segment that is not listed as needed", -2, x, y); k = ks; break; }
          }
      }
  }
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++)
      if (cover[(size_t)y * dw + x] != 1) complain("output pixel not covered exactly once", -1, x, y);
  if (stats) {
    stats[0] = plan.nftiles; stats[1] = plan.ntiles; stats[2] = plan.ndirect; stats[3] = needed;
    stats[4] = plan.stats.fused_raw_bytes; stats[5] = plan.stats.fused_blurred_bytes; stats[6] = plan.stats.fused_run_slots;
    stats[7] = plan.stats.fetched_bytes;
  }
  return bad;
}

// build_fuse_info() of the library for a context: returns the number of kernels (0: nothing fusable); row_kid[h], rects[4 * cap],
// taps[kFusedTapDwords * cap_k]
extern "C" int t360_host_fuse_info(const FrameTransformContext* ctx, int inW, int inH, int outW, int outH, short* row_kid,
                                   int* rects, int cap, unsigned* taps, int cap_k, int* nsegs) {
  t360::FilterConfig cfg;
  if (!t360::build_filter_config(*ctx, inW, inH, outW, outH, &cfg)) return -1;
  t360::FuseInfo fi;
  std::vector<uint32_t> packed;
  const bool any = t360::build_fuse_info(*ctx, cfg, inW, inH, &fi, &packed);
  *nsegs = (int)fi.segs.size();
  if ((int)fi.segs.size() > cap || (int)(packed.size() / t360::kFusedTapDwords) > cap_k) return -2;
  for (size_t i = 0; i < fi.segs.size(); i++) {
    rects[4 * i] = fi.segs[i].left; rects[4 * i + 1] = fi.segs[i].top; rects[4 * i + 2] = fi.segs[i].width; rects[4 * i + 3] = fi.segs[i].height;
  }
  for (int y = 0; y < inH; y++) row_kid[y] = fi.row_kid[(size_t)y];
  memcpy(taps, packed.data(), packed.size() * sizeof(uint32_t));
  return any ? (int)(packed.size() / t360::kFusedTapDwords) : 0;
}

// ---- request model (round 6) -----------------------------------------------------------------
// TCP -> TCC read requests of the staging per frame: every DMA instruction (one 1 KiB piece: 64 lanes x 16 bytes) asks its
// CU's L1 for the distinct 128-byte lines under its 64 chunks -- the L1 does not keep lines between the instructions of a
// streaming kernel, so that is the number of requests it sends on to the L2 -- next to the tile-level line count (distinct
// lines per TILE: what the L2 is asked for if a workgroup's own pieces never repeat a line) and the staged bytes.
// out[kind * 4 + 0..3] = tiles, pieces, requests (distinct lines per instruction, summed), distinct lines per tile (summed);
// kind 0..7 as kTile*, row 8 = all staged tiles.  Returns the tile count.
extern "C" int t360_plan_requests(const t360::LutEntry* lut, int dw, int dh, int sw, int sh, int ks, int max_pieces, int waves,
                                  int cost_lines, long long* out /*[9 * 4]*/) {
  using namespace t360;
  PlanOptions o;
  o.ks = ks;
  o.max_pieces = max_pieces;
  o.waves = waves & 0xff;
  o.wide256_pct = (waves >> 8) & 0xfff;   // the packing of t360_plan_verify
  o.scatter = (waves >> 24) & 0xf;
  o.cost_lines = cost_lines != 0 || ((waves >> 20) & 1) != 0;
  HostGatherPlan plan;
  if (!plan_gather(lut, dw, dh, sw, sh, o, &plan)) return -1;
  const int mp = max_pieces < kMaxPieces ? max_pieces : kMaxPieces;
  const size_t cstride = (size_t)tile_chunk_dwords(mp, plan.scatter);
  for (int i = 0; i < 36; i++) out[i] = 0;
  std::vector<uint32_t> seen;
  for (int ti = 0; ti < plan.ntiles; ti++) {
    const TileDesc& t = plan.tiles[(size_t)ti];
    const uint32_t* tc = &plan.chunks[(size_t)ti * cstride];
    long long req = 0;
    std::vector<uint32_t> tile_lines;
    for (int p = 0; p < t.pieces; p++) {
      seen.clear();
      for (int l = 0; l < kPieceChunks; l++) {
        const uint32_t e = tc[p * kPieceChunks + l];
        const uint32_t line = ((e >> 12) << 12) | ((e & 4095u) >> 3);
        bool dup = false;
        for (uint32_t s : seen) dup = dup || s == line;
        if (!dup) seen.push_back(line);
      }
      req += (long long)seen.size();
      tile_lines.insert(tile_lines.end(), seen.begin(), seen.end());
    }
    std::sort(tile_lines.begin(), tile_lines.end());
    const long long distinct = (long long)(std::unique(tile_lines.begin(), tile_lines.end()) - tile_lines.begin());
    for (int row : {(int)t.kind, 8}) {
      out[row * 4 + 0] += 1;
      out[row * 4 + 1] += t.pieces;
      out[row * 4 + 2] += req;
      out[row * 4 + 3] += distinct;
    }
  }
  return plan.ntiles;
}
