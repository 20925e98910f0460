// tests/plan_sim/l2sim.cpp -- offline model of the fabric reads of the tiled gather (development tool).
//
// Replays the launch the kernel would make for a yuv420p batch: every XCD owns a contiguous range of the
// execution-ordered tile list (Y tiles, then U, then V), `slots` workgroups are resident per XCD, each walks `nframes`
// frames with its tile, one frame per tick, and takes the XCD's next tile when it is done.  Every 16-byte chunk of a
// tile's footprint is a request to the XCD's L2 (128-byte lines, set-associative LRU); a miss is one 128-byte fabric
// read.  Output lines are allocated in the L2 as the stores would.  Prints lines fetched vs the source size.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>

#include "t360_plan.h"

namespace {
struct L2 {
  int sets, ways;
  std::vector<uint64_t> tag;
  std::vector<uint32_t> age;
  uint32_t clock = 0;
  long long miss = 0, hit = 0;
  L2(int bytes, int ways_) : ways(ways_) {
    sets = bytes / 128 / ways;
    tag.assign((size_t)sets * ways, ~0ull);
    age.assign((size_t)sets * ways, 0);
  }
  bool touch(uint64_t line, bool count) {
    // hash the set index a little: rows 30 lines apart must not pile onto few sets
    const uint64_t h = line ^ (line >> 11) ^ (line >> 22);
    const size_t s = (size_t)(h % (uint64_t)sets) * ways;
    clock++;
    int victim = 0;
    uint32_t oldest = 0xffffffffu;
    for (int w = 0; w < ways; w++) {
      if (tag[s + w] == line) {
        age[s + w] = clock;
        if (count) hit++;
        return true;
      }
      if (age[s + w] < oldest) oldest = age[s + w], victim = w;
    }
    tag[s + victim] = line;
    age[s + victim] = clock;
    if (count) miss++;
    return false;
  }
};

struct SimTile {
  int plane;                     // 0 Y, 1 U, 2 V
  std::vector<uint32_t> chunks;  // distinct (row << 12 | col16)
  int ox, oy, w, h;
};
}  // namespace


// planner options under study come from the environment of the simulator (never of the library)
static void sim_options(t360::PlanOptions* o) {
  if (const char* v = getenv("T360_SIM_WIDE256")) o->wide256_pct = atoi(v);
  if (const char* v = getenv("T360_SIM_COST_LINES")) o->cost_lines = atoi(v) != 0;
  if (const char* v = getenv("T360_SIM_WIDE")) o->wide_pct = atoi(v);
  if (const char* v = getenv("T360_SIM_SCATTER")) o->scatter = atoi(v);
  if (const char* v = getenv("T360_SIM_BAND")) o->band = atoi(v);
}

static bool build_tiles(const t360::LutEntry* lut_y, int dwy, int dhy, int swy, int shy, const t360::LutEntry* lut_c, int dwc, int dhc,
                        int swc, int shc, int ks, int max_pieces, int waves, int order, std::vector<SimTile>* tiles, int* ndirect) {
  using namespace t360;
  HostGatherPlan plan[2];
  for (int k = 0; k < 2; k++) {
    PlanOptions o;
    o.ks = ks;
    o.max_pieces = max_pieces;
    o.waves = waves;
    o.order = order;
    sim_options(&o);
    if (!plan_gather(k ? lut_c : lut_y, k ? dwc : dwy, k ? dhc : dhy, k ? swc : swy, k ? shc : shy, o, &plan[k])) return false;
  }
  const size_t cstride = (size_t)tile_chunk_dwords(max_pieces < kMaxPieces ? max_pieces : kMaxPieces, plan[0].scatter);
  *ndirect = plan[0].ndirect + 2 * plan[1].ndirect;
  for (int pl = 0; pl < 3; pl++) {
    const HostGatherPlan& p = plan[pl ? 1 : 0];
    for (int ti = 0; ti < p.ntiles; ti++) {
      SimTile t;
      t.plane = pl;
      const TileDesc& d = p.tiles[(size_t)ti];
      t.ox = d.ox; t.oy = d.oy;
      t.w = d.kind == kTileWide256 ? 256 : d.kind == kTileWide128 || d.kind == kTileStrip128 || d.kind == kTileScatter ? 128 : d.kind == kTileWide64 ? 64 : d.kind == kTileStaged32 ? 32 : 16;
      t.h = d.kind == kTileStrip128 || d.kind == kTileWide256 ? 8 : d.kind == kTileStaged32 ? 32 : 16;
      const uint32_t* tc = &p.chunks[(size_t)ti * cstride];
      uint32_t prev = ~0u;
      for (int pos = 0; pos < d.pieces * kPieceChunks; pos++)
        if (tc[pos] != prev) t.chunks.push_back(tc[pos]), prev = tc[pos];
      tiles->push_back(std::move(t));
    }
  }
  return true;
}

// Replay of a measured launch: item i = (tile, f0, f1, xcd, t0, t1) -- frame f of the item is requested at
// t0 + (t1 - t0) * (f - f0) / (f1 - f0).  Returns fabric read lines.
extern "C" long long t360_l2replay(const t360::LutEntry* lut_y, int dwy, int dhy, int swy, int shy, const t360::LutEntry* lut_c, int dwc,
                                   int dhc, int swc, int shc, int ks, int max_pieces, int waves, int order, int l2_bytes, int ways,
                                   int nitems, const int* item_tile, const int* item_f0, const int* item_f1, const int* item_xcd,
                                   const double* item_t0, const double* item_t1, long long* out) {
  std::vector<SimTile> tiles;
  int ndirect = 0;
  if (!build_tiles(lut_y, dwy, dhy, swy, shy, lut_c, dwc, dhc, swc, shc, ks, max_pieces, waves, order, &tiles, &ndirect)) return -1;
  const long long ybytes = (long long)swy * shy, cbytes = (long long)swc * shc, frame_in = ybytes + 2 * cbytes;
  const long long oy_bytes = (long long)dwy * dhy, oc_bytes = (long long)dwc * dhc, frame_out = oy_bytes + 2 * oc_bytes;
  const uint64_t out_base = (uint64_t)1 << 40;
  long long misses = 0, hits = 0;
  for (int xcd = 0; xcd < 8; xcd++) {
    struct Ev { double t; int item, f; };
    std::vector<Ev> ev;
    for (int i = 0; i < nitems; i++) {
      if (item_xcd[i] != xcd || item_tile[i] < 0 || item_tile[i] >= (int)tiles.size()) continue;
      const int nf = item_f1[i] - item_f0[i];
      // T360_SIM_ROTATE=period_us: the item starts at the frame a wall clock of that period is on and wraps around
      // inside its run (every workgroup resident at time t is then near frame t / period: lockstep without waiting)
      static const double rot_period = getenv("T360_SIM_ROTATE") ? atof(getenv("T360_SIM_ROTATE")) : 0.0;
      const int phi = rot_period > 0 ? (int)(item_t0[i] / rot_period) % nf : 0;
      // T360_SIM_FOLLOW=period_us: at EVERY step the item takes the frame the wall clock is on, or the next one after it
      // that it has not done yet (cyclically): it re-synchronises with everything else that is resident at every step,
      // and pays for running slower or faster than the clock with frames done alone at the end
      static const double follow_period = getenv("T360_SIM_FOLLOW") ? atof(getenv("T360_SIM_FOLLOW")) : 0.0;
      if (follow_period > 0 && nf <= 64) {
        unsigned long long done = 0;
        for (int f = 0; f < nf; f++) {
          const double t = item_t0[i] + (item_t1[i] - item_t0[i]) * f / nf;
          int fr = (int)((long long)(t / follow_period) % nf);
          while ((done >> fr) & 1ull) fr = fr + 1 == nf ? 0 : fr + 1;
          done |= 1ull << fr;
          ev.push_back({t, i, item_f0[i] + fr});
        }
        continue;
      }
      for (int f = 0; f < nf; f++) ev.push_back({item_t0[i] + (item_t1[i] - item_t0[i]) * f / nf, i, item_f0[i] + (phi + f) % nf});
    }
    std::sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.t < b.t; });
    L2 l2(l2_bytes, ways);
    for (const Ev& e : ev) {
      const SimTile& t = tiles[(size_t)item_tile[e.item]];
      const long long pbase = (long long)e.f * frame_in + (t.plane == 0 ? 0 : t.plane == 1 ? ybytes : ybytes + cbytes);
      const int stride = t.plane ? swc : swy;
      for (uint32_t c : t.chunks) l2.touch((uint64_t)(pbase + (long long)(c >> 12) * stride + (long long)(c & 4095u) * 16) >> 7, true);
      const long long obase = (long long)e.f * frame_out + (t.plane == 0 ? 0 : t.plane == 1 ? oy_bytes : oy_bytes + oc_bytes);
      const int ostride = t.plane ? dwc : dwy;
      for (int y = 0; y < t.h; y++)
        for (int x = 0; x < t.w; x += 128)
          l2.touch((out_base + (uint64_t)(obase + (long long)(t.oy + y) * ostride + t.ox + x)) >> 7, false);
    }
    misses += l2.miss;
    hits += l2.hit;
  }
  out[0] = misses;
  out[1] = hits;
  out[2] = (long long)tiles.size();
  out[3] = ndirect;
  return misses;
}

// plans: luma + chroma.  Returns fabric read lines; out[0..7] = stats
extern "C" long long t360_l2sim(const t360::LutEntry* lut_y, int dwy, int dhy, int swy, int shy, const t360::LutEntry* lut_c, int dwc,
                                int dhc, int swc, int shc, int ks, int max_pieces, int waves, int order, int nframes, int slots,
                                int l2_bytes, int ways, int jitter, int fpb, int lead, int spread, int window, long long* out) {
  using namespace t360;
  HostGatherPlan plan[2];
  for (int k = 0; k < 2; k++) {
    PlanOptions o;
    o.ks = ks;
    o.max_pieces = max_pieces;
    o.waves = waves;
    o.order = order;
    sim_options(&o);
    if (!plan_gather(k ? lut_c : lut_y, k ? dwc : dwy, k ? dhc : dhy, k ? swc : swy, k ? shc : shy, o, &plan[k])) return -1;
  }
  const size_t cstride = (size_t)tile_chunk_dwords(max_pieces < kMaxPieces ? max_pieces : kMaxPieces, plan[0].scatter);
  std::vector<SimTile> tiles;
  long long staged_chunks = 0;
  for (int pl = 0; pl < 3; pl++) {
    const HostGatherPlan& p = plan[pl ? 1 : 0];
    for (int ti = 0; ti < p.ntiles; ti++) {
      SimTile t;
      t.plane = pl;
      const TileDesc& d = p.tiles[(size_t)ti];
      t.ox = d.ox; t.oy = d.oy;
      t.w = d.kind == kTileWide256 ? 256 : d.kind == kTileWide128 || d.kind == kTileStrip128 || d.kind == kTileScatter ? 128 : d.kind == kTileWide64 ? 64 : d.kind == kTileStaged32 ? 32 : 16;
      t.h = d.kind == kTileStrip128 || d.kind == kTileWide256 ? 8 : d.kind == kTileStaged32 ? 32 : 16;
      const uint32_t* tc = &p.chunks[(size_t)ti * cstride];
      uint32_t prev = ~0u;
      for (int pos = 0; pos < d.pieces * kPieceChunks; pos++)
        if (tc[pos] != prev) t.chunks.push_back(tc[pos]), prev = tc[pos];
      staged_chunks += (long long)t.chunks.size();
      tiles.push_back(std::move(t));
    }
  }
  // T360_SIM_MERGE=x|y: model workgroups of twice the tile (256x16 / 128x32): the two tiles run in lock step
  double per_frame_scale = 1.0;
  if (const char* mg = getenv("T360_SIM_MERGE")) {
    const bool mx = mg[0] == 'x';
    std::vector<SimTile> merged;
    std::vector<char> used(tiles.size(), 0);
    for (size_t i = 0; i < tiles.size(); i++) {
      if (used[i]) continue;
      used[i] = 1;
      SimTile t = tiles[i];
      for (size_t j = 0; j < tiles.size(); j++) {
        const SimTile& b = tiles[j];
        if (used[j] || b.plane != t.plane || b.w != t.w || b.h != t.h) continue;
        const bool adj = mx ? (b.oy == t.oy && (b.ox == t.ox + t.w || t.ox == b.ox + b.w) && (std::min(b.ox, t.ox) / t.w) % 2 == 0)
                            : (b.ox == t.ox && (b.oy == t.oy + t.h || t.oy == b.oy + b.h) && (std::min(b.oy, t.oy) / t.h) % 2 == 0);
        if (!adj) continue;
        used[j] = 1;
        t.chunks.insert(t.chunks.end(), b.chunks.begin(), b.chunks.end());
        std::sort(t.chunks.begin(), t.chunks.end());
        t.chunks.erase(std::unique(t.chunks.begin(), t.chunks.end()), t.chunks.end());
        if (mx) t.ox = std::min(t.ox, b.ox), t.w *= 2; else t.oy = std::min(t.oy, b.oy), t.h *= 2;
        break;
      }
      merged.push_back(std::move(t));
    }
    tiles.swap(merged);
    staged_chunks = 0;
    for (const SimTile& t : tiles) staged_chunks += (long long)t.chunks.size();
    per_frame_scale = 0.5;  // twice the waves work on it
  }
  const int total = (int)tiles.size();
  // T360_SIM_NT=1: lines (of one frame) that exactly one tile touches are loaded past the L2 (`nt`: fetched, not kept)
  const bool sim_nt = getenv("T360_SIM_NT") && atoi(getenv("T360_SIM_NT")) != 0;
  const int sim_block = getenv("T360_SIM_BLOCK") ? atoi(getenv("T360_SIM_BLOCK")) : 0;
  std::vector<uint8_t> line_users;  // per 128-byte line of one input frame: tiles touching it (saturating)
  if (sim_nt) {
    const long long yb = (long long)swy * shy, cb = (long long)swc * shc;
    line_users.assign((size_t)((yb + 2 * cb) >> 7) + 2, 0);
    for (const SimTile& t : tiles) {
      const long long pbase = t.plane == 0 ? 0 : t.plane == 1 ? yb : yb + cb;
      const int stride = t.plane ? swc : swy;
      long long prev = -1;
      std::vector<long long> ls;
      for (uint32_t e : t.chunks) ls.push_back((pbase + (long long)(e >> 12) * stride + (long long)(e & 4095u) * 16) >> 7);
      std::sort(ls.begin(), ls.end());
      for (long long l : ls)
        if (l != prev) { prev = l; if (line_users[(size_t)l] < 255) line_users[(size_t)l]++; }
    }
    long long single = 0, multi = 0;
    for (uint8_t u : line_users) { single += u == 1; multi += u > 1; }
    fprintf(stderr, "lines of a frame: %lld touched by one tile, %lld by several\n", single, multi);
  }
  std::vector<std::vector<int>> neigh((size_t)total);
  for (int i = 0; i < total; i++)
    for (int j = 0; j < total; j++) {
      if (i == j || tiles[(size_t)i].plane != tiles[(size_t)j].plane) continue;
      const SimTile &a = tiles[(size_t)i], &b = tiles[(size_t)j];
      const bool xo = a.ox < b.ox + b.w && b.ox < a.ox + a.w, yo = a.oy < b.oy + b.h && b.oy < a.oy + a.h;
      const bool xt = a.ox == b.ox + b.w || b.ox == a.ox + a.w, yt = a.oy == b.oy + b.h || b.oy == a.oy + a.h;
      if ((xo && yt) || (yo && xt)) neigh[(size_t)i].push_back(j);
    }
  const long long ybytes = (long long)swy * shy, cbytes = (long long)swc * shc;
  const long long frame_in = ybytes + 2 * cbytes;
  const long long oy_bytes = (long long)dwy * dhy, oc_bytes = (long long)dwc * dhc;
  const long long frame_out = oy_bytes + 2 * oc_bytes;
  const uint64_t out_base = (uint64_t)1 << 40;
  long long misses = 0, hits = 0, wr_lines = 0;
  double tend = 0, waited = 0;
  std::mt19937 rng(12345);
  for (int xcd = 0; xcd < 8; xcd++) {
    const int q = total >> 3, rem = total & 7;
    const int len = q + (xcd < rem ? 1 : 0), start = xcd * q + (xcd < rem ? xcd : rem);
    L2 l2(l2_bytes, ways);
    struct Slot { int tile = -1, f = 0, f1 = 0, wait = 0, rot = 0, done = 0; long long myF = -1; unsigned long long mask = 0; };
    // lead <= -2 (round 5, follow-the-frontier): the XCD's 'newest frame started' word (frame = counter mod run length);
    // lead == -4: the frontier is the AVERAGE progress instead (steps taken on the XCD / resident workgroups)
    long long frontier = 0, votes = 0;
    int front = 0;
    const double startup = 5.0;
    std::vector<Slot> slot((size_t)slots);
    // work items: (tile, f0, f1) in order
    struct Item { int tile, f0, f1; };
    std::vector<Item> items;
    if (fpb < 0) {  // frame-group major: the XCD sweeps all its tiles for frames 0..|fpb|-1, then the next group
      for (int f0 = 0; f0 < nframes; f0 += -fpb)
        for (int t = 0; t < len; t++) items.push_back({start + t, f0, std::min(nframes, f0 - fpb)});
    } else {
      for (int t = 0; t < len; t++)
        for (int f0 = 0; f0 < nframes; f0 += fpb) items.push_back({start + t, f0, std::min(nframes, f0 + fpb)});
    }
    size_t next = 0;
    // event driven: every slot has the time of its next frame; a frame of a tile of p pieces takes a + b*p (us), with
    // `jitter` % of uniform noise; `lead` > 0: a workgroup may not run more than `lead` frames ahead of the slowest
    // resident workgroup of its XCD (it polls every 0.2 us)
    std::vector<double> tnext((size_t)slots, 0.0);
    std::vector<double> speed((size_t)slots, 1.0);  // per work item: this much slower or faster for its whole life
    const double ta = 0.69, tb = 0.0234;
    // neighbour-relative limit (lead >= 100: lead - 100 frames): a workgroup may not run more than that many frames
    // ahead of the slowest RESIDENT workgroup whose tile touches its own
    std::vector<int> where((size_t)total, -1);  // tile -> slot it is resident in
    std::uniform_real_distribution<double> uni(-1.0, 1.0);
    for (;;) {
      int si = -1;
      double best = 1e30;
      for (int i = 0; i < slots; i++) {
        Slot& s = slot[(size_t)i];
        if ((s.tile < 0 || s.f >= s.f1) && next >= items.size()) { s.tile = -1; continue; }
        if (tnext[(size_t)i] < best) best = tnext[(size_t)i], si = i;
      }
      if (si < 0) break;
      Slot& s = slot[(size_t)si];
      if (s.tile < 0 || s.f >= s.f1) {
        if (s.tile >= 0) where[(size_t)s.tile] = -1;
        s.tile = items[next].tile; s.f = items[next].f0; s.f1 = items[next].f1; next++;
        where[(size_t)s.tile] = si;
        speed[(size_t)si] = 1.0 + 0.01 * spread * uni(rng);
        tnext[(size_t)si] += startup;  // start-up
        s.rot = 0;
        if (lead == -1) {  // start at the frame the XCD's front is on (+ what it will advance during my start-up), wrap around
          s.rot = (front + (int)(startup / 1.0)) % (s.f1 - s.f);
        }
        s.done = 0;
        s.myF = -1; s.mask = 0;
        continue;
      }
      if (lead >= 100) {
        int slowest = 1 << 30;
        for (int nb : neigh[(size_t)s.tile]) {
          const int sj = where[(size_t)nb];
          // a neighbour hopelessly far behind (another generation) is not waited for: nothing to share with it
          if (sj >= 0 && slot[(size_t)sj].tile == nb && slot[(size_t)sj].f < slot[(size_t)sj].f1 && s.f - slot[(size_t)sj].f <= window)
            slowest = std::min(slowest, slot[(size_t)sj].f);
        }
        if (s.f > slowest + (lead - 100)) { tnext[(size_t)si] += 0.2; waited += 0.2; continue; }
      } else if (lead > 0) {
        int slowest = 1 << 30;
        for (int i = 0; i < slots; i++)
          if (slot[(size_t)i].tile >= 0 && slot[(size_t)i].f < slot[(size_t)i].f1)
            slowest = std::min(slowest, slot[(size_t)i].f);
        if (s.f > slowest + lead) { tnext[(size_t)si] += 0.2; continue; }
      }
      const SimTile& t = tiles[(size_t)s.tile];
      const int nfr = s.f1 - (s.f - s.done);
      int fr = (s.f - s.done) + (s.rot + s.done) % nfr;
      if (lead == -1) front = fr;
      if (lead <= -2) {
        // follow the frontier without waiting: propose my last frontier value + 1, take the maximum anyone has proposed
        // (one L2-local atomic max per step); if I have done that frame already, fill a hole (-2: the next undone frame
        // after it, cyclically; -3: my lowest undone one).  -4: the frontier is the XCD's average progress.
        long long cur;
        if (lead == -4) {
          votes++;
          cur = votes / slots;
        } else {
          cur = std::max(frontier, s.myF + 1);
          frontier = cur;
        }
        s.myF = cur;
        int r = (int)(cur % nfr);
        if ((s.mask >> r) & 1ull) {
          if (lead != -3) { while ((s.mask >> r) & 1ull) r = r + 1 == nfr ? 0 : r + 1; }
          else { r = 0; while ((s.mask >> r) & 1ull) r++; }
        }
        s.mask |= 1ull << r;
        fr = (s.f - s.done) + r;
      }
      const long long pbase = (long long)fr * frame_in + (t.plane == 0 ? 0 : t.plane == 1 ? ybytes : ybytes + cbytes);
      const int stride = t.plane ? swc : swy;
      long long prev_nt = -1;
      const long long fbase = (long long)fr * frame_in;
      for (uint32_t e : t.chunks) {
        long long a = pbase + (long long)(e >> 12) * stride + (long long)(e & 4095u) * 16;
        if (sim_block > 0) {
          // T360_SIM_BLOCK=h: the plane stored in blocks of (128 / h) x h pixels (one 128-byte line each) instead of rows
          const long long row = e >> 12, colb = (long long)(e & 4095u) * 16, bw = 128 / sim_block;
          a = pbase + ((row / sim_block) * (stride / bw) + colb / bw) * 128 + (row % sim_block) * bw + colb % bw;
        }
        if (sim_nt && line_users[(size_t)((a - fbase) >> 7)] == 1) {
          if ((a >> 7) != prev_nt) l2.miss++, prev_nt = a >> 7; else l2.hit++;
          continue;
        }
        l2.touch((uint64_t)a >> 7, true);
      }
      const long long obase = (long long)fr * frame_out + (t.plane == 0 ? 0 : t.plane == 1 ? oy_bytes : oy_bytes + oc_bytes);
      const int ostride = t.plane ? dwc : dwy;
      for (int y = 0; y < t.h; y++)
        for (int x = 0; x < t.w; x += 128) {
          l2.touch((out_base + (uint64_t)(obase + (long long)(t.oy + y) * ostride + t.ox + x)) >> 7, false);
          wr_lines++;
        }
      s.f++;
      s.done++;
      const double pieces = per_frame_scale * (double)t.chunks.size() / 64.0;
      tnext[(size_t)si] += (ta + tb * pieces) * speed[(size_t)si] * (1.0 + 0.01 * jitter * uni(rng));
      tend = std::max(tend, tnext[(size_t)si]);
    }
    misses += l2.miss;
    hits += l2.hit;
  }
  out[0] = misses;
  out[1] = hits;
  out[2] = staged_chunks;
  out[3] = total;
  out[4] = frame_in;
  out[5] = wr_lines;
  out[6] = (long long)(tend * 1000);
  out[7] = (long long)waited;
  return misses;
}
