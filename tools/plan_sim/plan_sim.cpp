// tools/plan_sim/plan_sim.cpp -- the product's gather planner (transform360_amd/csrc/t360_plan.cpp) built
// for the host, so that tile shapes, fetched bytes and modelled LDS bank conflicts can be compared offline
// (tools/plan_sim/plan_sim.py feeds it the oracle's LUT).  Development tool, not part of the library.
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
  stats[0] = s.n_strip + s.n_wide128 * 2; stats[1] = s.n_wide; stats[2] = s.n_sq; stats[3] = s.n_16; stats[4] = s.n_direct;
  stats[5] = s.fetched_bytes; stats[6] = s.lds_bytes; stats[7] = s.direct_pixels; stats[8] = s.line_bytes;
  stats[9] = (long long)plan.chunks.size() * 4; stats[10] = (long long)plan.tlut.size() * 4;
  int maxp = 0;
  for (int i = 0; i < plan.ntiles; i++) maxp = plan.tiles[i].pieces > maxp ? plan.tiles[i].pieces : maxp;
  stats[11] = maxp;
  for (int i = 0; i < 33; i++) stats[12 + i] = s.pieces_hist[i];
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
