// t360_plan.cpp -- init-time planning of the LDS-tiled gather (host side, no HIP).
//
// The output plane is cut into tiles of 256 lanes x NPX pixels (128x8, 64x16 or 32x32 px with 4 px per lane;
// 16x16 px with 1 px per lane near the poles) and, for workgroups of 8 waves, 128x16 px tiles of 512 lanes.  For every tile the planner derives from the sample LUT
//   * the FOOTPRINT: the set of 16-byte source chunks the tile's stencils touch.  In the equirect source the
//     footprint of a cube-face tile is a curved band (an annular sector on the polar faces); its bounding box
//     would stage up to 2.6x the bytes that are used, so the footprint is kept exact, row by row;
//   * an LDS placement: the box rows are packed back to back (row r holds the chunks first[r] .. last[r] of its
//     source row at LDS chunk positions pos[r] ..), so LDS and DMA lanes carry almost no holes either;
//   * the chunk table the waves walk when they stage a frame: per LDS position the source (row, 16-byte column), with the
//     +-180 degree seam and BORDER_WRAP across the poles already resolved; positions nobody needs repeat
//     their predecessor's source chunk (an L1 hit, no HBM traffic);
//   * the pixel words in the lane order of the gather (box row and x of the stencil's top-left tap, phase) and the
//     row table that turns a box row into an LDS address (rows have no common pitch).
// Tiles whose footprint exceeds the staging budget even at 16x16 (the ~4 tiles around each pole)
// are listed as direct tiles.  Tiles are emitted in execution order; the kernel hands contiguous ranges of the
// list to each XCD so that neighbouring tiles -- whose footprints share the stencil halo -- share an L2.
#include "t360_plan.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
#include <utility>

namespace t360 {

namespace {

struct TileShape {
  int kind, w, h, npx, lanes;
};
constexpr TileShape kStrip{kTileStrip128, 128, 8, 4, 256};
constexpr TileShape kWide{kTileWide64, 64, 16, 4, 256};
constexpr TileShape kSquare{kTileStaged32, 32, 32, 4, 256};
constexpr TileShape kSmall{kTileStaged16, 16, 16, 1, 256};
constexpr TileShape kWide128{kTileWide128, 128, 16, 4, 512};
constexpr TileShape kWide256{kTileWide256, 256, 8, 4, 512};
constexpr TileShape kScatter{kTileScatter, 128, 16, 4, 512};  // w, h: the lane grid (128 columns x 4 bands), not a rectangle

inline int floor_div16(int v) { return v >> 4; }  // arithmetic shift: floors negatives too
inline int wrap(int v, int n) {
  v %= n;
  return v < 0 ? v + n : v;
}

// Footprint and LDS placement of one candidate tile.
struct Foot {
  bool empty = true;      // no pixel inside the plane
  bool feasible = false;  // fits the staging budget
  bool seam = false;
  int ox = 0, oy = 0;
  TileShape shape{};
  int y0 = 0, rows = 0;   // box rows: source rows y0 .. y0 + rows - 1 (before wrapping)
  int c0 = 0, ncols = 0;  // box chunk columns c0 .. c0 + ncols - 1 (seam-shifted coordinates, may be negative)
  std::vector<uint8_t> mask;  // rows x ncols: chunk is part of the footprint
  std::vector<int> first, last, pos;  // per box row: marked chunk columns first..last (-1: none), LDS chunk position
  int npos = 0;           // LDS chunk positions
  int skew = 0;           // LDS bank skew per box row, in chunks (see place())
  bool scanned = false;   // first / last are those of the current mask
  int pieces = 0;
  int fetched = 0;        // marked chunks
  int lines = 0;          // distinct 128-byte source lines among them (what the fabric delivers if nothing is shared)
  std::vector<uint32_t> blocks;  // kTileScatter: origins (ox | oy << 16) of its 4x4-px blocks, block q on lanes 4q..4q+3 of
                                 // band q / 32
  uint32_t order_key = 0;        // kTileScatter: position in the execution order (Z-order over source strips and rows)
};

class Planner {
 public:
  Planner(const LutEntry* lut, int dw, int dh, int sw, int sh, const PlanOptions& opt)
      : lut_(lut), dw_(dw), dh_(dh), sw_(sw), sh_(sh), opt_(opt) {
    lo_ = opt.ks == 1 ? 0 : opt.ks / 2 - 1;  // taps of a ks-wide stencil: -(ks/2-1) .. +ks/2 around the LUT's
    hi_ = opt.ks == 1 ? 0 : opt.ks / 2;      // integer coordinate (nearest: the pixel itself)
    max_pos_ = std::min(opt.max_pieces, kMaxPieces) * kPieceChunks;
    if (fuse_on()) {
      // segments that intersect a source row, by row (the unfused tiles' "which blurred segments do I read" lookup)
      row_segs_.assign((size_t)sh_, {});
      for (size_t i = 0; i < opt_.fuse->segs.size(); i++) {
        const FuseSegment& g = opt_.fuse->segs[i];
        for (int y = std::max(0, g.top); y < std::min(sh_, g.top + g.height); y++) row_segs_[(size_t)y].push_back((int)i);
      }
    }
  }
  bool fuse_on() const {
    return opt_.fuse != nullptr && opt_.waves == 8 && (opt_.ks == 2 || opt_.ks == 4) && (int)opt_.fuse->row_kid.size() == sh_ &&
           opt_.scatter <= 0;
  }

  // the output pixel (if any) that lane `tid` holds as its pixel `p`
  bool pixel_of_lane(const Foot& f, int tid, int p, int* px, int* py) const {
    const TileShape& s = f.shape;
    if (s.kind == kTileScatter) {
      const size_t q = (size_t)(tid >> 7) * 32 + (size_t)((tid & 127) >> 2);
      if (q >= f.blocks.size()) return false;
      *px = (int)(f.blocks[q] & 0xffffu) + (tid & 3);
      *py = (int)(f.blocks[q] >> 16) + p;
    } else if (s.npx == 4) {
      *px = f.ox + tid % s.w;
      *py = f.oy + (tid / s.w) * 4 + p;
    } else {
      *px = f.ox + (tid & 15);
      *py = f.oy + (tid >> 4);
    }
    return *px < dw_ && *py < dh_;
  }
  // every output pixel of the tile, once
  template <class Fn>
  void for_pixels(const Foot& f, Fn&& fn) const {
    if (f.shape.kind == kTileScatter) {
      for (uint32_t b : f.blocks)
        for (int dy = 0; dy < 4; dy++)
          for (int dx = 0; dx < 4; dx++) fn((int)(b & 0xffffu) + dx, (int)(b >> 16) + dy);
    } else {
      const int x1 = std::min(f.ox + f.shape.w, dw_), y1 = std::min(f.oy + f.shape.h, dh_);
      for (int y = f.oy; y < y1; y++)
        for (int x = f.ox; x < x1; x++) fn(x, y);
    }
  }

  void footprint(int ox, int oy, const TileShape& shape, Foot* f) const {
    f->empty = true;
    f->feasible = false;
    f->ox = ox;
    f->oy = oy;
    f->shape = shape;
    f->blocks.clear();
    if (ox >= dw_ || oy >= dh_) return;
    measure(f);
  }
  // a scatter tile: `blocks` = origins of <= 128 4x4-px blocks (all inside the plane)
  void footprint_blocks(const std::vector<uint32_t>& blocks, Foot* f) const {
    f->empty = true;
    f->feasible = false;
    f->shape = kScatter;
    f->blocks = blocks;
    f->ox = f->oy = 0;
    if (blocks.empty()) return;
    f->ox = (int)(blocks[0] & 0xffffu);
    f->oy = (int)(blocks[0] >> 16);
    measure(f);
  }

  // footprint (exact chunk mask), lines and LDS placement of the tile's pixel set
  void measure(Foot* f) const {
    int minx = 1 << 30, maxx = -(1 << 30), mins = 1 << 30, maxs = -(1 << 30), miny = 1 << 30, maxy = -(1 << 30);
    for_pixels(*f, [&](int x, int y) {
      const LutEntry& e = lut_[(size_t)y * dw_ + x];
      const int sx = e.ix, sy = e.iy;
      const int ss = sx >= (sw_ >> 1) ? sx - sw_ : sx;
      minx = std::min(minx, sx); maxx = std::max(maxx, sx);
      mins = std::min(mins, ss); maxs = std::max(maxs, ss);
      miny = std::min(miny, sy); maxy = std::max(maxy, sy);
    });
    f->empty = false;
    f->seam = (maxs - mins) < (maxx - minx);
    const int xa = (f->seam ? mins : minx) - lo_, xb = (f->seam ? maxs : maxx) + hi_;
    f->c0 = floor_div16(xa);
    f->ncols = floor_div16(xb) - f->c0 + 1;
    f->y0 = miny - lo_;
    f->rows = maxy + hi_ - f->y0 + 1;
    // a footprint wider than the plane or taller than what 16 pieces hold cannot be staged
    if (f->ncols * kStageChunk > sw_ || f->ncols > kBoxMaxCols || f->rows > kBoxMaxRows || f->rows > max_pos_) return;
    f->mask.assign((size_t)f->rows * f->ncols, 0);
    f->scanned = false;
    int fetched = 0;
    for_pixels(*f, [&](int x, int y) {
      const LutEntry& e = lut_[(size_t)y * dw_ + x];
      int sx = e.ix;
      if (f->seam && sx >= (sw_ >> 1)) sx -= sw_;
      const int ca = floor_div16(sx - lo_) - f->c0, cb = floor_div16(sx + hi_) - f->c0;
      const int r0 = e.iy - lo_ - f->y0;
      for (int r = r0; r < r0 + opt_.ks; r++) {
        uint8_t* m = &f->mask[(size_t)r * f->ncols];
        for (int c = ca; c <= cb; c++) {
          fetched += m[c] == 0;
          m[c] = 1;
        }
      }
    });
    f->fetched = fetched;
    {
      int lines = 0;
      for (int r = 0; r < f->rows; r++) {
        const uint8_t* m = &f->mask[(size_t)r * f->ncols];
        int prev_line = -(1 << 30);
        for (int c = 0; c < f->ncols; c++)
          if (m[c]) {
            const int line = (f->c0 + c) >> 3;  // 8 chunks of 16 bytes
            if (line != prev_line) lines++, prev_line = line;
          }
      }
      f->lines = lines;
    }
    if (fetched > max_pos_) return;
    f->skew = 0;
    place(f);
  }

  // the skew whose modelled bank conflicts are fewest (the staircase of source rows under a row of output pixels
  // climbs or descends depending on where the tile sits on its cube face); run on the tiles that are emitted
  void choose_skew(Foot* f) const {
    if (!(opt_.row_search && opt_.row_align > 1 && f->feasible && opt_.ks != 1)) return;
    // Which box row and which byte of it every sampled read of the bank model touches does not depend on the placement:
    // looked up ONCE per tile (the LUT accesses and the seam arithmetic were most of the search, and the search more than
    // half of a Lanczos4 plan); a placement then only contributes the rows' LDS positions.  Same reads in the same order
    // as lds_cycles(), so the same skew wins.
    const TileShape& s = f->shape;
    const int npx = s.npx;
    std::vector<int32_t> tap_row, tap_x;  // per sampled read: 32 lanes (row -1: lane outside the plane)
    for (int g = 0; g < s.lanes / 32; g++)
      for (int p = 0; p < npx; p += (npx > 1 ? npx - 1 : 1))
        for (int k = 0; k < opt_.ks; k += (opt_.ks > 1 ? opt_.ks - 1 : 1))
          for (int l = 0; l < 32; l++) {
            const int tid = g * 32 + l;
            int px, py;
            if (!pixel_of_lane(*f, tid, p, &px, &py)) {
              tap_row.push_back(-1);
              tap_x.push_back(0);
              continue;
            }
            const LutEntry& e = lut_[(size_t)py * dw_ + px];
            int sx = e.ix;
            if (f->seam && sx >= (sw_ >> 1)) sx -= sw_;
            tap_row.push_back(e.iy - lo_ - f->y0 + k);
            tap_x.push_back(sx - lo_ - f->c0 * kStageChunk);
          }
    const bool dual = opt_.model_dual;
    const int nbanks = dual ? 64 : 32;
    const int bbase = (1 << 20) + 4 + opt_.model_b_shift;
    auto cycles = [&]() {
      int total = 0;
      for (size_t rd = 0; rd < tap_row.size(); rd += 32)
        // (two ds_read_b32 per window: the second reads every lane's NEXT dword, i.e. the same pattern one bank further
        // -- the same worst case, counted twice instead of modelled twice)
        for (int acc = 0; acc < 1; acc++) {
          int cnt[64];
          int addr_of[64][4];
          memset(cnt, 0, sizeof(cnt));
          int worst = 1;
          for (int l = 0; l < 32; l++) {
            const int r = tap_row[rd + (size_t)l];
            if (r < 0) continue;
            const int off = row_base(*f, r) * kStageChunk + tap_x[rd + (size_t)l];
            const int a = dual ? (off & ~3) + ((off & 4) ? bbase : 0) : (off & ~3) + 4 * acc;
            for (int j = 0; j < (dual ? 2 : 1); j++) {
              const int d = (a >> 2) + j, b = d & (nbanks - 1);
              bool seen = false;
              for (int i = 0; i < cnt[b] && i < 4; i++) seen = seen || addr_of[b][i] == d;
              if (!seen) {
                if (cnt[b] < 4) addr_of[b][cnt[b]] = d;
                cnt[b]++;
                worst = std::max(worst, cnt[b]);
              }
            }
          }
          total += dual ? worst : 2 * worst;
        }
      return total;
    };
    int best = cycles(), best_skew = 0;
    for (int sk = 1; sk < opt_.row_align; sk++) {
      f->skew = sk;
      place(f);
      if (!f->feasible) continue;
      const int c = cycles();
      if (c < best) best = c, best_skew = sk;
    }
    f->skew = best_skew;
    place(f);
  }

  // LDS placement of the box rows.  Every pixel looks its stencil rows up in the row table, so rows may sit anywhere
  // and in any order.  row_align > 1: the LDS chunk position of a row is congruent to its source chunk column
  // (+ skew * row) modulo row_align.  With 8 (128 bytes = one sweep of the 32 banks a ds_read_b32 sees) and skew 0
  // the bank of a staged byte depends on its source x only; the lanes of a read are neighbouring output columns
  // whose taps span < 128 source bytes but drift over ~10 source rows, and the skew (chosen per tile by the bank
  // model, footprint()) keeps the staircase's steps off each other's banks.  Rows are then chained greedily so that
  // each starts where the previous one ends or a few chunks later: the padding is LDS and DMA lanes, never HBM traffic.
  void place(Foot* f) const {
    f->pos.assign((size_t)f->rows, 0);
    f->feasible = false;
    if (!f->scanned) {  // first / last marked chunk of every box row: a property of the mask, not of the placement
      f->first.assign((size_t)f->rows, -1);
      f->last.assign((size_t)f->rows, -1);
      for (int r = 0; r < f->rows; r++) {
        const uint8_t* m = &f->mask[(size_t)r * f->ncols];
        for (int c = 0; c < f->ncols; c++)
          if (m[c]) {
            if (f->first[(size_t)r] < 0) f->first[(size_t)r] = c;
            f->last[(size_t)r] = c;
          }
      }
      f->scanned = true;
    }
    std::vector<int> todo;
    for (int r = 0; r < f->rows; r++)
      if (f->first[(size_t)r] >= 0) todo.push_back(r);
    const int A = std::max(1, opt_.row_align);
    int at = 0;
    while (!todo.empty()) {
      size_t pick = 0;
      int gap = 0;
      if (A > 1) {
        int best = A;
        for (size_t i = 0; i < todo.size() && best > 0; i++) {
          const int r = todo[i];
          const int g = wrap(f->c0 + f->first[(size_t)r] + f->skew * r - at, A);
          if (g < best) best = g, pick = i;
        }
        gap = best;
      }
      const int r = todo[pick];
      todo.erase(todo.begin() + (long)pick);
      at += gap;
      f->pos[(size_t)r] = at;
      at += f->last[(size_t)r] - f->first[(size_t)r] + 1 + row_pad(*f, r);
    }
    f->npos = at;
    f->pieces = (at + kPieceChunks - 1) / kPieceChunks;
    if (at <= 0 || at > max_pos_) return;
    f->feasible = true;
  }

  // optional padding chunks behind a row (LDS bank spreading, PlanOptions::row_pad)
  int row_pad(const Foot& f, int r) const {
    if (opt_.row_pad <= 0) return 0;
    const int len = f.last[(size_t)r] - f.first[(size_t)r] + 1;
    // make (len + pad) odd multiples that avoid 0 mod 8: rows then start on different 128-byte bank halves
    int pad = 0;
    while (pad < opt_.row_pad && ((len + pad) & 7) == 0) pad++;
    return pad;
  }

  // LDS chunk-granular address term of box row r: byte offset of source x (seam-shifted, minus the box origin)
  // in row r is row_base(r) * 16 + (x - c0 * 16)
  int row_base(const Foot& f, int r) const { return f.pos[(size_t)r] - f.first[(size_t)r]; }

  // LDS byte offset (inside one copy) of the tap at stencil row k of the pixel with LUT entry e
  int tap_offset(const Foot& f, const LutEntry& e, int k) const {
    int sx = e.ix;
    if (f.seam && sx >= (sw_ >> 1)) sx -= sw_;
    const int r = e.iy - lo_ - f.y0 + k;
    return row_base(f, r) * kStageChunk + (sx - lo_ - f.c0 * kStageChunk);
  }

  // Modelled LDS cycles of the tile's stencil-row reads (MI355X_MICROARCH.md "LDS": a wave64 access is served in two
  // 32-lane groups; cycles of a group = most distinct dword addresses on one bank; checked on these patterns by
  // tools/ubench/lds_patterns.hip).  model_dual = false: two ds_read_b32 per window (dwords k and k+1 separately,
  // bank = dword % 32); true: one ds_read_b64 on the A or B copy (bank = dword % 64).  Summed over the stencil rows.
  int lds_cycles(const Foot& f) const {
    const bool dual = opt_.model_dual;
    const int nbanks = dual ? 64 : 32;
    const int bbase = (1 << 20) + 4 + opt_.model_b_shift;  // copy B: a multiple of 1 KiB plus 4 (+ bank stagger)
    const TileShape& s = f.shape;
    int total = 0;
    const int npx = s.npx;
    for (int g = 0; g < s.lanes / 32; g++) {  // groups of 32 lanes
      // a sample of the reads is enough to rank placements: first / last pixel of the lane, first / last stencil row
      for (int p = 0; p < npx; p += (npx > 1 ? npx - 1 : 1))
        for (int k = 0; k < opt_.ks; k += (opt_.ks > 1 ? opt_.ks - 1 : 1))
          for (int acc = 0; acc < (dual ? 1 : 2); acc++) {
            int cnt[64];
            int addr_of[64][4];
            memset(cnt, 0, sizeof(cnt));
            int worst = 1;
            for (int l = 0; l < 32; l++) {
              const int tid = g * 32 + l;
              int px, py;
              if (!pixel_of_lane(f, tid, p, &px, &py)) continue;
              const int off = tap_offset(f, lut_[(size_t)py * dw_ + px], k);
              const int a = dual ? (off & ~3) + ((off & 4) ? bbase : 0) : (off & ~3) + 4 * acc;
              for (int j = 0; j < (dual ? 2 : 1); j++) {
                const int d = (a >> 2) + j, b = d & (nbanks - 1);
                bool seen = false;
                for (int i = 0; i < cnt[b] && i < 4; i++) seen = seen || addr_of[b][i] == d;
                if (!seen) {
                  if (cnt[b] < 4) addr_of[b][cnt[b]] = d;
                  cnt[b]++;
                  worst = std::max(worst, cnt[b]);
                }
              }
            }
            total += worst;
          }
    }
    return total;
  }

  // pixel words of tile f in the lane order of the gather, at w[0 .. tile_words)
  void pixel_words(const Foot& f, uint32_t* w) const {
    const TileShape& s = f.shape;
    const int per_lane = opt_.ks == 8 ? 1 : 4;
    for (int tid = 0; tid < s.lanes; tid++)
      for (int p = 0; p < s.npx; p++) {
        int px, py;
        if (!pixel_of_lane(f, tid, p, &px, &py)) continue;
        const LutEntry& e = lut_[(size_t)py * dw_ + px];
        int sx = e.ix;
        if (f.seam && sx >= (sw_ >> 1)) sx -= sw_;
        const uint32_t xrel = (uint32_t)(sx - lo_ - f.c0 * kStageChunk);
        const uint32_t r0 = (uint32_t)(e.iy - lo_ - f.y0);
        w[tid * per_lane + p] = xrel | (r0 << kWordRowShift) | ((uint32_t)e.frac << kWordFracShift);
      }
  }

  // ---- fused low-pass tiles (t360_internal.h) ---------------------------------------------------------------------
  // Marks the segments whose BLURRED pixels the (unfused) tile reads: its staged rows, or -- a direct tile -- its stencils.
  void mark_needed(const Foot& f, bool direct, HostGatherPlan* out) const {
    if (out->seg_needed.size() != opt_.fuse->segs.size()) out->seg_needed.assign(opt_.fuse->segs.size(), 0);
    auto mark = [&](int y, int xa, int xb) {  // source row y (wrapped), pixel columns xa..xb inside the plane
      for (int si : row_segs_[(size_t)y]) {
        const FuseSegment& g = opt_.fuse->segs[(size_t)si];
        if (g.left <= xb && g.left + g.width > xa) out->seg_needed[(size_t)si] = 1;
      }
    };
    auto mark_wrapped = [&](int y, int xa, int xb) {  // xa..xb in unwrapped coordinates, at most one plane wide
      y = wrap(y, sh_);
      if (xb - xa + 1 >= sw_) return mark(y, 0, sw_ - 1);
      const int a = wrap(xa, sw_), b = wrap(xb, sw_);
      if (a <= b) return mark(y, a, b);
      mark(y, a, sw_ - 1);
      mark(y, 0, b);
    };
    if (direct) {
      for_pixels(f, [&](int x, int y) {
        const LutEntry& e = lut_[(size_t)y * dw_ + x];
        for (int r = e.iy - lo_; r <= e.iy + hi_; r++) mark_wrapped(r, e.ix - lo_, e.ix + hi_);
      });
      return;
    }
    for (int r = 0; r < f.rows; r++)
      if (f.first[(size_t)r] >= 0)
        mark_wrapped(f.y0 + r, (f.c0 + f.first[(size_t)r]) * kStageChunk, (f.c0 + f.last[(size_t)r]) * kStageChunk + kStageChunk - 1);
  }

  // The tile as a FUSED tile, if it can be one: appended to out->ftiles / ftlut / fchunks.  f is placed (B layout).
  bool emit_fused(Foot& f, HostGatherPlan* out) const {
    const TileShape& s = f.shape;
    const std::vector<int16_t>& row_kid = opt_.fuse->row_kid;
    if (s.npx != 4 || s.kind == kTileScatter || f.y0 < 0 || f.y0 + f.rows > sh_ || f.rows + 2 > kBoxMaxRows) return false;
    std::vector<int> kid((size_t)f.rows, -1);
    for (int r = 0; r < f.rows; r++) {
      if (f.first[(size_t)r] < 0) continue;
      // (the filter replicates the plane's left / right edge where the gather wraps: the runs of the plane's first and last
      // dword column say so in their run words, and the kernel replaces the neighbour dword it would read across the edge)
      kid[(size_t)r] = row_kid[(size_t)(f.y0 + r)];
      if (kid[(size_t)r] < 0) return false;
    }
    // R rows t = 0 .. rows + 1 <-> source row y0 - 1 + t; chunk columns relative to the R box origin (c0 - 1)
    const int rrows = f.rows + 2;
    std::vector<int> rfirst((size_t)rrows, -1), rlast((size_t)rrows, -1), alias((size_t)rrows);
    for (int t = 0; t < rrows; t++) {
      const int ys = f.y0 - 1 + t, yc = std::min(std::max(ys, 0), sh_ - 1);
      alias[(size_t)t] = t + (yc - ys);  // BORDER_REPLICATE above / below the plane: the clamped row's staged bytes
    }
    for (int r = 0; r < f.rows; r++) {
      if (f.first[(size_t)r] < 0) continue;
      for (int t = r; t <= r + 2; t++) {
        const int a = alias[(size_t)t];
        const int lo = f.first[(size_t)r], hi = f.last[(size_t)r] + 2;  // [first - 1, last + 1] + 1
        rfirst[(size_t)a] = rfirst[(size_t)a] < 0 ? lo : std::min(rfirst[(size_t)a], lo);
        rlast[(size_t)a] = std::max(rlast[(size_t)a], hi);
      }
    }
    // R placement: the LDS chunk position of a row is congruent to its source chunk column modulo 8 (128 bytes = one sweep
    // of the 32 banks a ds_read_b32 sees), so the bank of a staged byte depends on its source x alone: the lanes of a filter
    // read are neighbouring dword columns (run order below) and never meet on a bank, whatever rows their runs are in.
    // Rows are chained greedily by the smallest gap, as in place(); a gap is LDS and DMA lanes, never HBM traffic.
    std::vector<int> rpos((size_t)rrows, 0);
    int rat = 0;
    {
      std::vector<int> todo;
      for (int t = 0; t < rrows; t++)
        if (alias[(size_t)t] == t && rfirst[(size_t)t] >= 0) todo.push_back(t);
      while (!todo.empty()) {
        size_t pick = 0;
        int best = 8;
        for (size_t i = 0; i < todo.size() && best > 0; i++) {
          const int g = wrap(f.c0 - 1 + rfirst[(size_t)todo[i]] - rat, 8);
          if (g < best) best = g, pick = i;
        }
        const int t = todo[pick];
        todo.erase(todo.begin() + (long)pick);
        rat += best;
        rpos[(size_t)t] = rat;
        rat += rlast[(size_t)t] - rfirst[(size_t)t] + 1;
      }
    }
    const int slot_pos = std::max(rat, f.npos);
    if (rat <= 0 || slot_pos > max_pos_) return false;
    // runs: per B chunk column the maximal vertical spans of rows that hold it, cut where the kernel changes
    struct Span { int c, r0, len, kid; };
    std::vector<Span> spans;
    for (int c = 0; c < f.ncols; c++) {
      int r = 0;
      while (r < f.rows) {
        auto in = [&](int q) { return q < f.rows && f.first[(size_t)q] >= 0 && f.first[(size_t)q] <= c && c <= f.last[(size_t)q]; };
        if (!in(r)) { r++; continue; }
        int e = r + 1;
        while (in(e) && kid[(size_t)e] == kid[(size_t)r]) e++;
        spans.push_back({c, r, e - r, kid[(size_t)r]});
        r = e;
      }
    }
    std::stable_sort(spans.begin(), spans.end(), [](const Span& a, const Span& b) {
      return a.kid != b.kid ? a.kid < b.kid : a.c != b.c ? a.c < b.c : a.r0 < b.r0;
    });
    int ni = 0;
    for (int cand = 1; cand <= kFusedMaxRun && ni == 0; cand++) {
      int lanes = 0, in_kid = 0, prev = -1;
      for (const Span& sp : spans) {
        if (sp.kid != prev) lanes += (in_kid + 63) / 64 * 64, in_kid = 0, prev = sp.kid;
        in_kid += 4 * ((sp.len + cand - 1) / cand);
      }
      lanes += (in_kid + 63) / 64 * 64;
      if (lanes <= kFusedLanes) ni = cand;
    }
    if (ni == 0) return false;

    TileDesc t{};
    t.ox = (int16_t)f.ox;
    t.oy = (int16_t)f.oy;
    t.kind = (int16_t)s.kind;
    const bool partial = f.ox + s.w > dw_ || f.oy + s.h > dh_;
    t.flags = (int16_t)(kTileFused | (partial ? kTilePartial : 0));
    t.pieces = (int16_t)((slot_pos + kPieceChunks - 1) / kPieceChunks);
    t.rows = (int16_t)f.rows;
    t.fetched = rat;
    t.pad[1] = ni;
    t.pad[2] = f.npos;
    const int mp = std::min(opt_.max_pieces, kMaxPieces);
    const size_t cstride = (size_t)fused_chunk_dwords(mp), base = out->fchunks.size();
    out->fchunks.resize(base + cstride, 0);
    uint32_t* ct = &out->fchunks[base];
    // R chunk table; the gaps between rows and the positions behind the last one repeat their predecessor's chunk
    std::fill(ct, ct + (size_t)mp * kPieceChunks, 0xffffffffu);
    for (int tr = 0; tr < rrows; tr++)
      if (alias[(size_t)tr] == tr && rfirst[(size_t)tr] >= 0) {
        const int sy = f.y0 - 1 + tr;
        for (int c = rfirst[(size_t)tr]; c <= rlast[(size_t)tr]; c++) {
          const int cx = wrap((f.c0 - 1 + c) * kStageChunk, sw_) / kStageChunk;
          ct[rpos[(size_t)tr] + c - rfirst[(size_t)tr]] = chunk_entry((uint32_t)sy, (uint32_t)cx);
        }
      }
    {
      uint32_t prev = 0xffffffffu;
      for (int i = 0; i < mp * kPieceChunks && prev == 0xffffffffu; i++) prev = ct[i];
      for (int i = 0; i < mp * kPieceChunks; i++) {
        if (ct[i] == 0xffffffffu)
          ct[i] = prev;
        else
          prev = ct[i];
      }
    }
    uint32_t* btab = ct + (size_t)mp * kPieceChunks;
    for (int r = 0; r < f.rows; r++) {
      const uint32_t v = (uint32_t)(uint16_t)(int16_t)(f.first[(size_t)r] < 0 ? 0 : row_base(f, r));
      btab[r / 2] |= v << (16 * (r & 1));
    }
    uint32_t* rtab = btab + 64;
    for (int tr = 0; tr < rrows; tr++) {
      const int a = alias[(size_t)tr];
      const uint32_t v = (uint32_t)(uint16_t)(int16_t)(rfirst[(size_t)a] < 0 ? 0 : rpos[(size_t)a] - rfirst[(size_t)a]);
      rtab[tr / 2] |= v << (16 * (tr & 1));
    }
    uint32_t* runs = rtab + 64;
    uint32_t* winfo = runs + kFusedLanes;
    std::fill(runs, runs + kFusedLanes, kRunDead);
    {
      // lane order: kernel, then the run's index inside its column span, then the dword column -- the 32 lanes of an LDS
      // access are then neighbouring dword columns (distinct banks under the x-only placement above), in similar rows
      struct Run { int kid, k, d, a, len; };
      std::vector<Run> rl;
      int items = 0;
      for (const Span& sp : spans) {
        const int n = (sp.len + ni - 1) / ni;
        for (int k = 0; k < n; k++) {
          const int a = sp.r0 + (int)((int64_t)sp.len * k / n), b = sp.r0 + (int)((int64_t)sp.len * (k + 1) / n);
          for (int j = 0; j < 4; j++) rl.push_back({sp.kid, k, sp.c * 4 + j, a, b - a});
          items += 4 * (b - a);
        }
      }
      std::stable_sort(rl.begin(), rl.end(), [](const Run& x, const Run& y) {
        return x.kid != y.kid ? x.kid < y.kid : x.k != y.k ? x.k < y.k : x.d < y.d;
      });
      int lane = 0, prev = -1;
      for (const Run& r : rl) {
        if (r.kid != prev) {
          lane = (lane + 63) / 64 * 64;
          prev = r.kid;
        }
        const int xabs = wrap(f.c0 * kStageChunk + 4 * r.d, sw_);
        runs[lane] = (uint32_t)r.d | ((uint32_t)r.a << 9) | ((uint32_t)r.len << 16) | (xabs == 0 ? kRunLeftEdge : 0u) |
                     (xabs == sw_ - 4 ? kRunRightEdge : 0u);
        winfo[lane / 64] = (uint32_t)r.kid;
        lane++;
      }
      // lanes without a run of their own repeat the last run of their wave (a wave without any: the tile's last run and its
      // kernel): identical dwords written twice, and no lane of the filter phase is ever predicated off
      for (int w0 = 0; w0 < kFusedLanes; w0 += 64) {
        uint32_t rep = kRunDead;
        for (int l = w0; l < w0 + 64; l++)
          if (runs[l] != kRunDead) rep = runs[l];
        if (rep == kRunDead) {
          rep = runs[lane - 1];
          winfo[w0 / 64] = winfo[(lane - 1) / 64];
        }
        for (int l = w0; l < w0 + 64; l++)
          if (runs[l] == kRunDead) runs[l] = rep;
      }
      winfo[8] = (uint32_t)ni;
      out->stats.fused_blurred_bytes += (int64_t)items * 4;
      out->stats.fused_run_slots += (int64_t)kFusedLanes * ni;
    }
    const size_t wstride = (size_t)tile_words(opt_.ks, opt_.waves), wb = out->ftlut.size();
    out->ftlut.resize(wb + wstride, kWordDead);
    pixel_words(f, &out->ftlut[wb]);
    out->ftiles.push_back(t);
    out->stats.n_fused++;
    out->stats.fused_raw_bytes += (int64_t)rat * kStageChunk;
    return true;
  }

  void emit(Foot& f, HostGatherPlan* out, std::vector<TileDesc>* direct) const {
    choose_skew(&f);
    if (fuse_on()) {
      if (f.feasible && emit_fused(f, out)) return;
      mark_needed(f, !f.feasible, out);
    }
    TileDesc t{};
    t.ox = (int16_t)f.ox;
    t.oy = (int16_t)f.oy;
    const TileShape& s = f.shape;
    const bool partial = s.kind == kTileScatter ? (int)f.blocks.size() < kScatterBlocks : (f.ox + s.w > dw_ || f.oy + s.h > dh_);
    t.flags = (int16_t)((f.seam ? kTileSeamShift : 0) | (partial ? kTilePartial : 0));
    PlanStats& st = out->stats;
    if (!f.feasible) {
      t.kind = kTileDirect16;
      direct->push_back(t);
      st.n_direct++;
      st.direct_pixels += (int64_t)(std::min(f.ox + s.w, dw_) - f.ox) * (std::min(f.oy + s.h, dh_) - f.oy);
      return;
    }
    t.kind = (int16_t)s.kind;
    t.pad[0] = (int32_t)f.order_key;
    t.pieces = (int16_t)f.pieces;
    t.rows = (int16_t)f.rows;
    t.fetched = f.fetched;
    // chunk table at a fixed stride: position -> source chunk; holes repeat the previous valid entry
    const int mp = std::min(opt_.max_pieces, kMaxPieces);
    const int cstride = tile_chunk_dwords(mp, scatter_on());
    const size_t base = out->chunks.size();
    out->chunks.resize(base + (size_t)cstride, 0);
    const size_t used = (size_t)f.pieces * kPieceChunks;
    std::fill(out->chunks.begin() + (long)base, out->chunks.begin() + (long)(base + used), 0xffffffffu);
    for (int r = 0; r < f.rows; r++) {
      if (f.first[(size_t)r] < 0) continue;
      const uint8_t* m = &f.mask[(size_t)r * f.ncols];
      const int sy = wrap(f.y0 + r, sh_);
      for (int c = f.first[(size_t)r]; c <= f.last[(size_t)r]; c++)
        if (m[c]) {
          const int cx = wrap((f.c0 + c) * kStageChunk, sw_) / kStageChunk;
          out->chunks[base + (size_t)(f.pos[(size_t)r] + c - f.first[(size_t)r])] = chunk_entry((uint32_t)sy, (uint32_t)cx);
        }
    }
    uint32_t prev = 0xffffffffu;
    for (size_t i = base; i < base + used && prev == 0xffffffffu; i++) prev = out->chunks[i];
    for (size_t i = base; i < base + used; i++) {
      if (out->chunks[i] == 0xffffffffu)
        out->chunks[i] = prev;
      else
        prev = out->chunks[i];
    }
    for (size_t i = base + used; i < base + (size_t)mp * kPieceChunks; i++) out->chunks[i] = prev;  // never staged, but in bounds
    // row table: 64 dwords behind the chunk entries
    {
      const size_t rb = base + (size_t)mp * kPieceChunks;
      for (int r = 0; r < f.rows; r++) {
        const uint32_t v = (uint32_t)(uint16_t)(int16_t)(f.first[(size_t)r] < 0 ? 0 : row_base(f, r));
        out->chunks[rb + (size_t)r / 2] |= v << (16 * (r & 1));
      }
    }
    // scatter plans: the origins of the tile's 4x4 blocks behind the row table (rectangular tiles of such a plan: unused)
    if (s.kind == kTileScatter)
      for (size_t q = 0; q < f.blocks.size(); q++) out->chunks[base + (size_t)mp * kPieceChunks + 64 + q] = f.blocks[q];
    // pixel words at a fixed stride, lane order of the gather; 16x16 tiles of a mixed plan use the first word of a uint4
    const int wstride = tile_words(opt_.ks, opt_.waves);
    const size_t wb = out->tlut.size();
    out->tlut.resize(wb + (size_t)wstride, kWordDead);
    pixel_words(f, &out->tlut[wb]);
    out->tiles.push_back(t);
    st.fetched_bytes += (int64_t)f.fetched * kStageChunk;
    if (opt_.model_stats) {
      // 128-byte lines the tile touches (what HBM delivers if no other workgroup's fetch of the line is still in L2)
      int64_t lines = 0;
      for (int r = 0; r < f.rows; r++) {
        if (f.first[(size_t)r] < 0) continue;
        const uint8_t* m = &f.mask[(size_t)r * f.ncols];
        int prev_line = -(1 << 30);
        for (int c = f.first[(size_t)r]; c <= f.last[(size_t)r]; c++)
          if (m[c]) {
            const int line = (f.c0 + c) >> 3;  // 8 chunks of 16 bytes
            if (line != prev_line) lines++, prev_line = line;
          }
      }
      st.line_bytes += lines * 128;
    }
    st.lds_bytes += (int64_t)f.npos * kStageChunk;
    st.pieces_hist[f.pieces < 32 ? f.pieces : 32]++;
    if (opt_.model_stats && opt_.ks != 1) st.lds_cycles_model += lds_cycles(f);
    (s.kind == kTileStrip128 ? st.n_strip : s.kind == kTileWide64 ? st.n_wide : s.kind == kTileStaged32 ? st.n_sq
     : s.kind == kTileWide128 ? st.n_wide128 : s.kind == kTileWide256 ? st.n_wide256 : s.kind == kTileScatter ? st.n_scatter : st.n_16)++;
  }

  // what a tile costs when shapes are compared: the distinct 128-byte lines of its footprint (cost_lines: what the
  // fabric delivers when no neighbour's fetch of a shared line is still in the L2) or its 16-byte chunks (what is staged)
  int64_t cost_of(const Foot& f) const { return opt_.cost_lines ? (int64_t)f.lines * 8 : (int64_t)f.fetched; }

  // the tiles of one 128x32 output region at (ox, oy), appended to `pick` in execution order; returns their cost
  // (infeasible 16x16 tiles -- the direct tiles around the poles -- count as their pixels' stencils read one by one)
  int64_t choose_region(int ox, int oy, std::vector<Foot>* pick) const {
    const bool only16 = opt_.ks == 8;  // Lanczos4 keeps 32 weight dwords per pixel in registers: one pixel per lane
    // (nearest has no stencil halo to share: by staged chunks a wider tile never wins -- by the 128-byte LINES under its
    // longer row fragments it does, so nearest maps take wide tiles only when shapes are compared by lines)
    const bool wide_ok = !only16 && opt_.wide_pct > 0 && (opt_.ks != 1 || opt_.cost_lines);
    const bool strip_ok = !only16 && opt_.strip_pct > 0;
    Foot strip[4], wide[2], sq[4], small;
    // squares first (always evaluated: they are the fallback), then the wider shapes
    auto cost_sq = [&](int k, bool* all_staged) -> int64_t {
      if (sq[k].empty) return 0;
      if (sq[k].feasible) return cost_of(sq[k]);
      *all_staged = false;
      return (int64_t)1 << 40;
    };
    for (int k = 0; k < 4; k++) {
      if (only16)
        sq[k].empty = ox + 32 * k >= dw_ || oy >= dh_, sq[k].feasible = false;
      else
        footprint(ox + 32 * k, oy, kSquare, &sq[k]);
    }
    if (opt_.waves == 8 && wide_ok) {
      // 128x16 tiles (workgroups of 8 waves): half the tile borders per pixel -- fewer halo bytes, and half as many row
      // fragments that end inside a 128-byte line some other workgroup fetches again
      Foot big[2];
      footprint(ox, oy, kWide128, &big[0]);
      footprint(ox, oy + 16, kWide128, &big[1]);
      bool sq_ok = true;
      int64_t cq = 0;
      for (int k = 0; k < 4; k++) cq += cost_sq(k, &sq_ok);
      const bool bf = !big[0].empty && big[0].feasible && (big[1].empty || big[1].feasible);
      const int64_t cb = (big[0].empty ? 0 : cost_of(big[0])) + (big[1].empty ? 0 : cost_of(big[1]));
      if (bf && (!sq_ok || cb * 100 <= cq * opt_.wide_pct)) {
        pick->push_back(big[0]);
        if (!big[1].empty) pick->push_back(big[1]);
        return cb;
      }
    }
    if (strip_ok) {
      bool ok = true, sq_ok = true;
      int64_t cs = 0, cq = 0;
      for (int k = 0; k < 4; k++) {
        footprint(ox, oy + 8 * k, kStrip, &strip[k]);
        if (!strip[k].empty) {
          ok = ok && strip[k].feasible;
          cs += cost_of(strip[k]);
        }
        cq += cost_sq(k, &sq_ok);
      }
      if (ok && (!sq_ok || cs * 100 <= cq * opt_.strip_pct)) {
        for (int k = 0; k < 4; k++)
          if (!strip[k].empty) pick->push_back(strip[k]);
        return cs;
      }
    }
    int64_t total = 0;
    for (int h = 0; h < 2; h++) {
      if (wide_ok && !sq[2 * h].empty && !sq[2 * h + 1].empty) {
        footprint(ox + 64 * h, oy, kWide, &wide[0]);
        footprint(ox + 64 * h, oy + 16, kWide, &wide[1]);
        const bool wf = (wide[0].empty || wide[0].feasible) && (wide[1].empty || wide[1].feasible) && !wide[0].empty;
        bool sq_ok = true;
        const int64_t cq = cost_sq(2 * h, &sq_ok) + cost_sq(2 * h + 1, &sq_ok);
        const int64_t cw = (wide[0].empty ? 0 : cost_of(wide[0])) + (wide[1].empty ? 0 : cost_of(wide[1]));
        if (wf && (!sq_ok || cw * 100 <= cq * opt_.wide_pct)) {
          if (!wide[0].empty) pick->push_back(wide[0]);
          if (!wide[1].empty) pick->push_back(wide[1]);
          total += cw;
          continue;
        }
      }
      for (int k = 2 * h; k < 2 * h + 2; k++) {
        if (sq[k].empty) continue;
        if (sq[k].feasible) {
          pick->push_back(sq[k]);
          total += cost_of(sq[k]);
          continue;
        }
        for (int qd = 0; qd < 4; qd++) {
          footprint(ox + 32 * k + (qd & 1) * 16, oy + (qd >> 1) * 16, kSmall, &small);
          if (small.empty) continue;
          pick->push_back(small);  // staged 16x16 or direct
          total += small.feasible ? cost_of(small) : (int64_t)256 * opt_.ks * 8;
        }
      }
    }
    return total;
  }

  // One planning region: 128x32 output px, or -- with PlanOptions::wide256_pct > 0 and workgroups of 8 waves -- 256x32,
  // where four 256x8 tiles replace the two 128x32 halves' tiles when they cost no more than wide256_pct % of them.
  // A 256-px-wide tile fetches ~490-byte row fragments: 5 lines for 4 of payload, where a 128-px-wide one fetches 3
  // for 2 -- and most of the ragged line ends of neighbouring tiles are fetched twice in practice, because
  // neighbouring workgroups drift apart in frame number.
  void plan_region(int rx, int ry, HostGatherPlan* out, std::vector<TileDesc>* direct) const {
    std::vector<Foot> pick;
    const int rw = region_w();
    const int ox = rx * rw, oy = ry * 32;
    if (rw == 256) {
      std::vector<Foot> halves;
      int64_t ch = choose_region(ox, oy, &halves);
      if (ox + 128 < dw_) ch += choose_region(ox + 128, oy, &halves);
      Foot w256[4];
      bool ok = true;
      int64_t cw = 0;
      for (int k = 0; k < 4 && ok; k++) {
        footprint(ox, oy + 8 * k, kWide256, &w256[k]);
        if (w256[k].empty) continue;
        ok = w256[k].feasible;
        cw += cost_of(w256[k]);
      }
      if (ok && !w256[0].empty && cw * 100 <= ch * opt_.wide256_pct) {
        for (int k = 0; k < 4; k++)
          if (!w256[k].empty) pick.push_back(w256[k]);
      } else {
        pick.swap(halves);
      }
    } else {
      choose_region(ox, oy, &pick);
    }
    for (Foot& f : pick) emit(f, out, direct);
  }
  bool scatter_on() const {
    return opt_.scatter > 0 && opt_.wide256_pct <= 0 && opt_.waves == 8 && (opt_.ks == 2 || opt_.ks == 4) && (dw_ & 3) == 0 && (dh_ & 3) == 0;
  }
  int region_w() const { return (opt_.waves == 8 && opt_.wide256_pct > 0 && opt_.ks != 8 && opt_.ks != 1) ? 256 : 128; }


  // ---- scatter tiles ----------------------------------------------------------------------------------------------
  // A 128x16 output rectangle on a cube face maps to a SLANTED band of the equirect source: every source row holds a
  // short fragment of it, and the 128-byte lines under the fragments' ends are fetched by the neighbouring tiles too --
  // again and again once neighbours drift apart in frame number (2.28x the source plane in distinct lines per tile,
  // summed over the tiles of BASELINE config 2's luma map).  A scatter tile is cut the other way round:
  // the plane's 4x4-pixel blocks are sorted by where their stencils lie in the SOURCE (strips of opt_.scatter lines,
  // top to bottom) and every 128 consecutive blocks form a tile, whose footprint is then a compact source rectangle
  // (1.79x unshared; with the same drift an LRU model of the L2 reads 1.24x the source instead of 1.45x).
  struct PoolBlock {
    uint32_t origin;  // ox | oy << 16
    int cx, cy;       // centre of the block's stencil box in the source (cx wrapped into the plane)
  };
  struct Group {
    std::vector<uint32_t> blocks;
    uint32_t key;
  };
  // stencil box of the 4x4 block at (bx4, by4) [pixels]; false: too wide / tall to share a tile with its source neighbours
  bool block_box(int ox, int oy, PoolBlock* b) const {
    int minx = 1 << 30, maxx = -(1 << 30), mins = 1 << 30, maxs = -(1 << 30), miny = 1 << 30, maxy = -(1 << 30);
    for (int y = oy; y < oy + 4; y++)
      for (int x = ox; x < ox + 4; x++) {
        const LutEntry& e = lut_[(size_t)y * dw_ + x];
        const int sx = e.ix, ss = sx >= (sw_ >> 1) ? sx - sw_ : sx;
        minx = std::min(minx, sx); maxx = std::max(maxx, sx);
        mins = std::min(mins, ss); maxs = std::max(maxs, ss);
        miny = std::min(miny, (int)e.iy); maxy = std::max(maxy, (int)e.iy);
      }
    const bool seam = (maxs - mins) < (maxx - minx);
    const int xa = seam ? mins : minx, xb = seam ? maxs : maxx;
    b->origin = (uint32_t)ox | ((uint32_t)oy << 16);
    b->cx = wrap((xa + xb) / 2, sw_);
    b->cy = (miny + maxy) / 2;
    return xb - xa + opt_.ks <= 64 && maxy - miny + opt_.ks <= 40 && miny - lo_ >= 0 && maxy + hi_ < sh_;
  }
  // one group of the pool -> one scatter tile, or two halves of it when its footprint does not fit the staging budget
  bool plan_group(const std::vector<uint32_t>& blocks, uint32_t key, HostGatherPlan* out, std::vector<TileDesc>* direct) const {
    Foot f;
    footprint_blocks(blocks, &f);
    if (f.feasible) {
      f.order_key = key;
      emit(f, out, direct);
      return true;
    }
    if (blocks.size() < 2) return false;
    const std::vector<uint32_t> a(blocks.begin(), blocks.begin() + (long)(blocks.size() / 2)), b(blocks.begin() + (long)(blocks.size() / 2), blocks.end());
    return plan_group(a, key, out, direct) && plan_group(b, key, out, direct);
  }

  // A scatter group whose footprint cannot be staged even as single blocks (block_box() bounds a block's size, not its
  // feasibility under the piece budget and the row alignment) sends the 128x32 output regions of its blocks back to the
  // rectangular path and the plan is made again, instead of failing the whole plane into the general gather (ADVICE
  // round 4).  Terminates: every retry grows `rect`, and with every region in it no scatter group exists.
  bool run(HostGatherPlan* out) const {
    std::set<uint32_t> rect;
    for (;;) {
      std::vector<uint32_t> back;
      *out = HostGatherPlan();
      if (run_once(out, rect, &back)) return true;
      if (back.empty()) return false;
      const size_t before = rect.size();
      rect.insert(back.begin(), back.end());
      if (rect.size() == before) return false;
    }
  }

  static uint32_t region_key(int rx, int ry) { return (uint32_t)rx | ((uint32_t)ry << 16); }

  bool run_once(HostGatherPlan* out, const std::set<uint32_t>& rect, std::vector<uint32_t>* back) const {
    const int regions_x = (dw_ + region_w() - 1) / region_w(), regions_y = (dh_ + 31) / 32;
    const int band = std::max(1, opt_.band);
    // Emission order = execution order.  raster = false: region rows are walked in bands, column by column inside
    // a band, so that vertically adjacent tiles -- whose footprints share the stencil halo and the rows a curved
    // footprint adds -- run at the same time on the same XCD.
    std::vector<std::pair<int, int>> regions;
    std::vector<Group> groups;
    {
      std::vector<PoolBlock> pool;
      for (int ry0 = 0; ry0 < regions_y; ry0 += band)
        for (int rx = 0; rx < regions_x; rx++)
          for (int ry = ry0; ry < std::min(ry0 + band, regions_y); ry++) {
            bool regular = scatter_on() && !rect.count(region_key(rx, ry));
            const size_t mark = pool.size();
            for (int oy = ry * 32; regular && oy < std::min(ry * 32 + 32, dh_); oy += 4)
              for (int ox = rx * 128; regular && ox < std::min(rx * 128 + 128, dw_); ox += 4) {
                PoolBlock b;
                regular = block_box(ox, oy, &b);
                pool.push_back(b);
              }
            if (regular) continue;         // its blocks are in the pool
            pool.resize(mark);
            regions.push_back({rx, ry});   // planned as output rectangles (the polar caps, mostly)
          }
      if (!pool.empty()) {
        // strips of opt_.scatter 128-byte lines, top to bottom; a tile = 128 consecutive blocks of a strip
        const int strip_w = 128 * std::max(1, opt_.scatter);
        std::stable_sort(pool.begin(), pool.end(), [&](const PoolBlock& a, const PoolBlock& b) {
          const int sa = a.cx / strip_w, sb = b.cx / strip_w;
          return sa != sb ? sa < sb : a.cy < b.cy;
        });
        auto morton = [](unsigned x, unsigned y) {
          uint32_t m = 0;
          for (int b = 0; b < 12; b++) m |= (((x >> b) & 1u) << (2 * b)) | (((y >> b) & 1u) << (2 * b + 1));
          return m;
        };
        size_t i = 0;
        while (i < pool.size()) {
          const int strip = pool[i].cx / strip_w;
          size_t j = i;
          while (j < pool.size() && j - i < (size_t)kScatterBlocks && pool[j].cx / strip_w == strip) j++;
          Group g;
          for (size_t k = i; k < j; k++) g.blocks.push_back(pool[k].origin);
          // WHICH blocks share a tile is decided in the source; their ORDER inside the tile follows the output raster
          // (oy, then ox): neighbouring blocks of an output row then sit on neighbouring lane quads, so a wave's stores
          // are runs of up to 64 contiguous bytes per row again (scattered 4-byte stores made the kernel 4x slower: every
          // store instruction touched 64 lines) and the 32 lanes of an LDS read are neighbouring columns, as the bank
          // placement assumes
          std::sort(g.blocks.begin(), g.blocks.end(), [](uint32_t a, uint32_t b) {
            return (a >> 16) != (b >> 16) ? (a >> 16) < (b >> 16) : (a & 0xffffu) < (b & 0xffffu);
          });
          g.key = morton((unsigned)strip, (unsigned)std::max(0, pool[(i + j) / 2].cy) >> 4);
          groups.push_back(std::move(g));
          i = j;
        }
      }
    }
    // regions and groups are independent: plan contiguous slices of the job list on a few host threads, splice in order
    const size_t n = regions.size() + groups.size();
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)32, (n + 7) / 8}));
    std::vector<HostGatherPlan> part(nthreads);
    std::vector<std::vector<TileDesc>> part_direct(nthreads);
    // no exception may leave a worker (std::terminate) or this function (the C ABI returns 0/1): a slice that runs out
    // of memory marks the plan as failed, a thread that cannot be started has its slice planned here
    std::atomic<bool> failed{false};
    std::mutex back_mu;
    auto work = [&](size_t ti) {
      try {
        const size_t lo = n * ti / nthreads, hi = n * (ti + 1) / nthreads;
        for (size_t i = lo; i < hi && !failed.load(std::memory_order_relaxed); i++) {
          if (i < regions.size())
            plan_region(regions[i].first, regions[i].second, &part[ti], &part_direct[ti]);
          else if (!plan_group(groups[i - regions.size()].blocks, groups[i - regions.size()].key, &part[ti], &part_direct[ti])) {
            std::lock_guard<std::mutex> lk(back_mu);
            for (uint32_t o : groups[i - regions.size()].blocks) back->push_back(region_key((int)(o & 0xffffu) / 128, (int)(o >> 16) / 32));
            failed.store(true);
          }
        }
      } catch (...) {
        failed.store(true);
      }
    };
    {
      std::vector<std::thread> th;
      std::vector<size_t> here;
      for (size_t ti = 1; ti < nthreads; ti++) {
        try {
          th.emplace_back(work, ti);
        } catch (...) {
          here.push_back(ti);
        }
      }
      work(0);
      for (size_t ti : here) work(ti);
      for (auto& t : th) t.join();
    }
    if (failed.load()) return false;
    // Assemble the plan: the execution order is decided on the descriptors alone, then every tile's tables are copied
    // ONCE from the slice that planned it to their final place, on the same threads (a Lanczos4 plan of an 8K map is
    // 330 MB of tables: appending the slices and re-ordering the result moved them three times on one thread).
    out->stats = PlanStats();
    std::vector<TileDesc> direct;
    std::vector<std::pair<uint32_t, uint32_t>> where;  // concatenated tile -> (slice, tile inside the slice)
    for (size_t ti = 0; ti < nthreads; ti++) {
      const HostGatherPlan& p = part[ti];
      for (size_t k = 0; k < p.tiles.size(); k++) where.push_back({(uint32_t)ti, (uint32_t)k});
      direct.insert(direct.end(), part_direct[ti].begin(), part_direct[ti].end());
      PlanStats& a = out->stats;
      const PlanStats& b = p.stats;
      a.n_strip += b.n_strip; a.n_wide += b.n_wide; a.n_sq += b.n_sq; a.n_16 += b.n_16; a.n_direct += b.n_direct;
      a.n_wide128 += b.n_wide128; a.n_wide256 += b.n_wide256; a.n_scatter += b.n_scatter;
      a.fetched_bytes += b.fetched_bytes; a.lds_bytes += b.lds_bytes; a.direct_pixels += b.direct_pixels;
      a.lds_cycles_model += b.lds_cycles_model;
      a.line_bytes += b.line_bytes;
      for (int i = 0; i < 33; i++) a.pieces_hist[i] += b.pieces_hist[i];
      a.n_fused += b.n_fused; a.fused_raw_bytes += b.fused_raw_bytes; a.fused_blurred_bytes += b.fused_blurred_bytes;
      a.fused_run_slots += b.fused_run_slots;
    }
    if (fuse_on()) {
      // the fused work list, in the same execution order as the unfused one (Z-order of 64x16 cells), and the segments the
      // unfused tiles still need
      out->seg_needed.assign(opt_.fuse->segs.size(), 0);
      std::vector<std::pair<uint32_t, uint32_t>> fw;
      for (size_t ti = 0; ti < nthreads; ti++) {
        for (size_t k = 0; k < part[ti].ftiles.size(); k++) fw.push_back({(uint32_t)ti, (uint32_t)k});
        for (size_t i = 0; i < part[ti].seg_needed.size(); i++) out->seg_needed[i] |= part[ti].seg_needed[i];
      }
      auto fmorton = [](unsigned x, unsigned y) {
        uint64_t m = 0;
        for (int b = 0; b < 16; b++) m |= ((uint64_t)((x >> b) & 1) << (2 * b)) | ((uint64_t)((y >> b) & 1) << (2 * b + 1));
        return m;
      };
      std::vector<uint64_t> fkey(fw.size());
      std::vector<size_t> fidx(fw.size());
      for (size_t i = 0; i < fw.size(); i++) {
        const TileDesc& t = part[fw[i].first].ftiles[fw[i].second];
        fkey[i] = ((opt_.order == 2 ? fmorton((unsigned)t.ox >> 6, (unsigned)t.oy >> 4) : 0) << 32) | ((uint64_t)(uint16_t)t.oy << 16) | (uint16_t)t.ox;
        fidx[i] = i;
      }
      if (opt_.order != 0) std::stable_sort(fidx.begin(), fidx.end(), [&](size_t a, size_t b) { return fkey[a] < fkey[b]; });
      const size_t fws = (size_t)tile_words(opt_.ks, opt_.waves), fcs = (size_t)fused_chunk_dwords(std::min(opt_.max_pieces, kMaxPieces));
      out->ftiles.assign(fw.size(), TileDesc{});
      out->ftlut.assign(fw.size() * fws, 0u);
      out->fchunks.assign(fw.size() * fcs, 0u);
      for (size_t i = 0; i < fw.size(); i++) {
        const HostGatherPlan& p = part[fw[fidx[i]].first];
        const size_t k = fw[fidx[i]].second;
        out->ftiles[i] = p.ftiles[k];
        memcpy(&out->ftlut[i * fws], &p.ftlut[k * fws], fws * sizeof(uint32_t));
        memcpy(&out->fchunks[i * fcs], &p.fchunks[k * fcs], fcs * sizeof(uint32_t));
      }
      out->nftiles = (int)fw.size();
    }
    const size_t nt = where.size();
    std::vector<size_t> idx(nt);
    for (size_t i = 0; i < nt; i++) idx[i] = i;
    if (opt_.order != 0) {
      // order 1: raster; order 2: Z-order over 64x16 cells (neighbours in BOTH directions are a few list positions
      // apart: horizontally and vertically adjacent tiles -- whose footprints share source lines -- start back to back
      // on the same XCD and walk the frames in step, so the shared lines come from HBM once and from that XCD's L2
      // the second time)
      auto morton = [](unsigned x, unsigned y) {
        uint64_t m = 0;
        for (int b = 0; b < 16; b++) m |= ((uint64_t)((x >> b) & 1) << (2 * b)) | ((uint64_t)((y >> b) & 1) << (2 * b + 1));
        return m;
      };
      const bool z = opt_.order == 2;
      auto tile_at = [&](size_t i) -> const TileDesc& { return part[where[i].first].tiles[where[i].second]; };
      std::vector<uint64_t> key(nt);
      for (size_t i = 0; i < nt; i++) {
        const TileDesc& t = tile_at(i);
        key[i] = t.kind == kTileScatter
                     ? (uint64_t)(uint32_t)t.pad[0]
                     : ((uint64_t)1 << 63) | ((z ? morton((unsigned)t.ox >> 6, (unsigned)t.oy >> 4) : 0) << 32) | ((uint64_t)(uint16_t)t.oy << 16) | (uint16_t)t.ox;
      }
      std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    }
    const size_t ws = (size_t)tile_words(opt_.ks, opt_.waves), cs = (size_t)tile_chunk_dwords(std::min(opt_.max_pieces, kMaxPieces), scatter_on());
    out->scatter = scatter_on();
    out->tiles.assign(nt, TileDesc{});
    out->tlut.assign(nt * ws, 0u);
    out->chunks.assign(nt * cs, 0u);
    auto copy_slice = [&](size_t ti) {
      try {
        for (size_t i = nt * ti / nthreads; i < nt * (ti + 1) / nthreads; i++) {
          const HostGatherPlan& p = part[where[idx[i]].first];
          const size_t k = where[idx[i]].second;
          out->tiles[i] = p.tiles[k];
          memcpy(&out->tlut[i * ws], &p.tlut[k * ws], ws * sizeof(uint32_t));
          memcpy(&out->chunks[i * cs], &p.chunks[k * cs], cs * sizeof(uint32_t));
        }
      } catch (...) {
        failed.store(true);
      }
    };
    {
      std::vector<std::thread> th;
      std::vector<size_t> here;
      for (size_t ti = 1; ti < nthreads; ti++) {
        try {
          th.emplace_back(copy_slice, ti);
        } catch (...) {
          here.push_back(ti);
        }
      }
      copy_slice(0);
      for (size_t ti : here) copy_slice(ti);
      for (auto& t : th) t.join();
    }
    if (failed.load()) return false;
    out->ntiles = (int)out->tiles.size();
    // direct tiles: upper half of the plane first (each pole's tiles go to one XCD, t360_remap_tiled.hip)
    std::stable_sort(direct.begin(), direct.end(), [&](const TileDesc& x, const TileDesc& y) {
      return (x.oy >= dh_ / 2) < (y.oy >= dh_ / 2);
    });
    out->ndirect = (int)direct.size();
    out->ndirect_top = 0;
    for (const TileDesc& t : direct) out->ndirect_top += t.oy < dh_ / 2 ? 1 : 0;
    out->tiles.insert(out->tiles.end(), direct.begin(), direct.end());
    if (out->tlut.empty()) out->tlut.push_back(kWordDead);
    if (out->chunks.empty()) out->chunks.push_back(0);
    return out->tlut.size() < 0x7fffffffu && out->chunks.size() < 0x7fffffffu;
  }

 private:
  const LutEntry* lut_;
  int dw_, dh_, sw_, sh_;
  PlanOptions opt_;
  int lo_, hi_, max_pos_;
  std::vector<std::vector<int>> row_segs_;  // fuse_on(): segments intersecting each source row
};

}  // namespace

bool plan_gather(const LutEntry* lut, int dw, int dh, int sw, int sh, const PlanOptions& opt, HostGatherPlan* out) {
  if (!(opt.ks == 1 || opt.ks == 2 || opt.ks == 4 || opt.ks == 8) || dw <= 0 || dh <= 0 || sw <= 0 || sh <= 0 ||
      (sw % kStageChunk) != 0 || dw > 32767 || dh > 32767)
    return false;
  try {
    Planner p(lut, dw, dh, sw, sh, opt);
    return p.run(out);
  } catch (...) {  // std::bad_alloc of the splice / the tables: not plannable, the general gather serves the map
    return false;
  }
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
    for (int r = 0; r < ks; r++)
      for (int q = 0; q < win; q++) {
        uint32_t hi = 0, lo = 0;
        for (int c = 0; c < 4; c++) {
          const int tap = 4 * q + c;
          if (tap >= ks) break;
          const int v = w[r * ks + tap];
          const int h = v >> 8;   // arithmetic shift: floor, in [-128, 127]
          const int l = v & 255;  // v == h * 256 + l
          hi |= (uint32_t)(h & 255) << (8 * c);
          lo |= (uint32_t)l << (8 * c);
        }
        o[r * win + q] = hi;
        o[nw + r * win + q] = lo;
      }
    // (pixels enter the signed part as p - 128:  SUM p*w = 256 * (SUM (p-128)*wh + 128 * SUM wh) + SUM p*wl; the kernel
    // derives 128 * SUM wh from the high bytes)
  }
}

}  // namespace t360
