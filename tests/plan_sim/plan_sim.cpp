// tests/plan_sim/plan_sim.cpp -- the product's gather planner (transform360_amd/csrc/t360_plan.cpp) built
// for the host, so that tile shapes, fetched bytes and modelled LDS bank conflicts can be compared offline
// (tests/plan_sim/plan_sim.py feeds it the oracle's LUT).  Development tool, not part of the library.
#include <cstdio>
#include <cstring>

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
