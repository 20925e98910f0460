// t360_remap_tiled.hip -- LDS-tiled gather over a batch of frames (the hot kernels).
//
// Same arithmetic as t360_remap.hip (cv::remap, BORDER_WRAP, Q15 weights, (sum + 16384) >> 15,
// SURVEY.md Appendix A.3/A.4); written around the bicubic 4x4 stencil and instantiated for nearest,
// bilinear and Lanczos4 as well.  Organised for MI355X:
//
//   * one workgroup (4 consumer waves + 1 loader wave) owns one OUTPUT tile (64x16 or 32x32 px,
//     4 px per lane, lane = column; 16x16 px, 1 px per lane near the poles; the tile shape is the
//     plan's choice, t360_plan.cpp) and walks `frames_per_block` frames of the batch with it.  Everything
//     that depends only on geometry -- the lane's LDS read addresses, its 16 Q15 weights per
//     pixel, the addresses of the source chunks it stages -- is computed ONCE per tile and kept
//     in registers for all frames: per frame a lane only moves bytes and issues dot products.
//   * per frame the tile's source bounding box (planned at init, t360_tiles.hip) is staged
//     through LDS in 16-byte chunks (equirect rows are contiguous in HBM).  In the main kernel
//     the chunks go global -> LDS by DMA (global_load_lds_dwordx4, no VGPR round trip) into a
//     RING of K slots, K-1 frames ahead of the frame being computed; completion is tracked
//     with counted s_waitcnt vmcnt(N) and ONE workgroup barrier per frame, so HBM latency
//     (~1-2 us) is covered by K-1 frames of work instead of being paid once per frame.
//     Taps that wrap across the +-180 degree seam or the poles are resolved while staging
//     (a chunk's source address is wrapped), so the gather itself never wraps.
//   * the 4x4 stencil of one output pixel costs 4 ds_read2_b32 + 4 v_alignbit (unaligned 4-byte
//     row windows) and 8 v_dot4: weights are split into a signed high byte and an unsigned low
//     byte (w = 256*wh + wl) and pixels enter the high part as p-128,
//         SUM p*w = 256*SUM (p-128)*wh + SUM p*wl + 128*256*SUM wh,
//     all exact in int32, so results are bit-identical to the integer formulation.
//   * all planes of the frame (Y, U, V) are tiles of ONE launch; workgroups are numbered so that
//     every XCD gets a contiguous range of the raster-ordered tile list (shared halo -> shared L2).
//   * no MFMA: this is a gather, not a contraction.
#include <hip/hip_runtime.h>

#include <atomic>

#include "t360_internal.h"
#include "t360_kernels.h"
#include "t360_sample.h"

namespace t360 {

namespace {

constexpr int kRingMaxSlots = 8;

__device__ __forceinline__ uint32_t bias128(uint32_t px4) { return px4 ^ 0x80808080u; }

// ---- per-pixel geometry shared by both staging variants --------------------------------------
// KS = taps per axis: 1 nearest, 2 bilinear, 4 bicubic, 8 Lanczos4.  A stencil row is read as
// WIN 4-byte windows (bilinear uses the first two bytes of its window; the packed weights of the
// other two are zero).
template <int KS>
struct Stencil {
  static constexpr int ROWS = KS;
  static constexpr int WIN = KS == 8 ? 2 : 1;
  static constexpr int NW = KS == 1 ? 0 : ROWS * WIN;  // weight dwords per half (hi / lo)
  static constexpr int PACK = pack_dwords(KS);         // dwords per phase in the packed table
};

template <int NPX, int KS>
struct PixelSetup {
  static constexpr int NWA = Stencil<KS>::NW > 0 ? Stencil<KS>::NW : 1;
  int off[NPX];           // byte offset of the stencil's top-left tap inside the staged box
  uint32_t wh[NPX][NWA];  // signed high bytes of the weights, one dword per 4-byte window
  uint32_t wl[NPX][NWA];  // unsigned low bytes
  int bias[NPX];          // 16384 + 128*256*SUM(wh)
  bool live[NPX];         // pixel inside the plane (partial tiles)
};

template <int NPX, int KS>
__device__ __forceinline__ void load_pixels(const TiledPlane& pl, const uint32_t* __restrict__ wpack,
                                            const TileDesc& t, int pitch, PixelSetup<NPX, KS>& s, int debug = 0) {
  constexpr int NW = Stencil<KS>::NW;
  const int tid = threadIdx.x;
  uint32_t words[4];
  if (NPX == 4) {
    const uint4 v = reinterpret_cast<const uint4*>(pl.tlut + t.tlut)[tid];
    words[0] = v.x; words[1] = v.y; words[2] = v.z; words[3] = v.w;
  } else {
    words[0] = pl.tlut[t.tlut + tid];
  }
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    const uint32_t e = words[p];
    s.live[p] = (e >> 31) == 0;
    const int rx = e & 1023, ry = (e >> 10) & 255, frac = (debug & 128) ? 0 : (e >> 18) & 1023;
    s.off[p] = s.live[p] ? ry * pitch + rx : 0;
    s.bias[p] = 0;
    if (NW > 0) {
      // [NW high dwords][NW low dwords], 16-byte aligned: 2*NW/4 vector loads
      const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(wpack + (size_t)frac * Stencil<KS>::PACK);
      uint32_t w[2 * (NW > 0 ? NW : 2)];
#pragma unroll
      for (int k = 0; k < (2 * NW) / 4; k++) {
        const uint4 v = wp[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
      }
      // rounding + bias term 16384 + 128*256*SUM(wh): SUM of the signed high bytes by dot4 with ones
      int sh = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        s.wh[p][k] = w[k];
        s.wl[p][k] = w[NW + k];
        sh = __builtin_amdgcn_sdot4((int)w[k], 0x01010101, sh, false);
      }
      s.bias[p] = (1 << (kCoefBits - 1)) + 128 * 256 * sh;
    }
  }
}

// Make hipcc wait for its own (counted) loads HERE: every loaded value passes through an empty
// asm, so the compiler-inserted s_waitcnt lands before it and not in front of the first use
// inside the frame loop, where it would also drain the DMA ring.
template <int NPX, int KS>
__device__ __forceinline__ void pin_pixels(PixelSetup<NPX, KS>& s) {
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    asm volatile("" : "+v"(s.off[p]), "+v"(s.bias[p]));
#pragma unroll
    for (int r = 0; r < Stencil<KS>::NW; r++) asm volatile("" : "+v"(s.wh[p][r]), "+v"(s.wl[p][r]));
  }
}

// one frame of one tile: gather from the staged box at `box`, write the output pixels.
// GROUP = pixels whose LDS reads are in flight together: 4 -> one LDS round trip per frame and
// 32 VGPRs of read data (bicubic); 2 -> two round trips, 16 VGPRs (fits 6 waves per SIMD).
// (Unaligned ds_read_b32 windows were measured 2.8x SLOWER than aligned ds_read2_b32 + v_alignbit
// on gfx950, so a window is always assembled from two aligned dwords.)
template <int NPX, int KS, int GROUP>
__device__ __forceinline__ void gather_store(const PixelSetup<NPX, KS>& s, const uint8_t* __restrict__ box, int pitch,
                                             uint8_t* __restrict__ d, int dstride, bool dword_store) {
  constexpr int G = GROUP < NPX ? GROUP : NPX;
  constexpr int ROWS = Stencil<KS>::ROWS, WIN = Stencil<KS>::WIN;
  int v[NPX];
  if (KS == 1) {
    // nearest: the byte itself (cv::remap INTER_NEAREST)
#pragma unroll
    for (int p = 0; p < NPX; p++) v[p] = box[s.off[p]];
  } else {
#pragma unroll
    for (int p0 = 0; p0 < NPX; p0 += G) {
      // Phase 1: the group's LDS reads in flight at once; phase 2: the dot products.
      uint32_t win[G][ROWS][WIN + 1];
#pragma unroll
      for (int p = 0; p < G; p++) {
        const int a4 = s.off[p0 + p] & ~3;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(box + a4 + r * pitch);
#pragma unroll
          for (int k = 0; k < WIN + 1; k++) win[p][r][k] = q[k];
        }
      }
      // keep hipcc from sinking the reads next to their uses (it would serialise the round trips)
#pragma unroll
      for (int p = 0; p < G; p++)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          if (WIN == 1) {
            uint64_t pr = (uint64_t)win[p][r][0] | ((uint64_t)win[p][r][1] << 32);
            asm volatile("" : "+v"(pr));
            win[p][r][0] = (uint32_t)pr;
            win[p][r][1] = (uint32_t)(pr >> 32);
          } else {
#pragma unroll
            for (int k = 0; k < WIN + 1; k++) asm volatile("" : "+v"(win[p][r][k]));
          }
        }
#pragma unroll
      for (int p = 0; p < G; p++) {
        const uint32_t sh = (uint32_t)(s.off[p0 + p] & 3) * 8u;
        int hi = 0;
        uint32_t lo = (uint32_t)s.bias[p0 + p];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
#pragma unroll
          for (int w = 0; w < WIN; w++) {
            // 4 consecutive source bytes out of the aligned dword pair
            const uint32_t px4 = __builtin_amdgcn_alignbit(win[p][r][w + 1], win[p][r][w], sh);
            hi = __builtin_amdgcn_sdot4((int)bias128(px4), (int)s.wh[p0 + p][r * WIN + w], hi, false);
            lo = __builtin_amdgcn_udot4(px4, s.wl[p0 + p][r * WIN + w], lo, false);
          }
        }
        const int sum = (hi << 8) + (int)lo;  // = SUM p*w + 16384
        v[p0 + p] = sat_u8(sum >> kCoefBits);
      }
    }
  }
  if (NPX == 4) {
    // v[k] is the pixel of column x = lane & 31 in row 4*(lane >> 5) + k of the tile.
    if (dword_store) {
      // 4x4 byte transpose inside each quad of lanes (DPP quad broadcasts + v_perm), so that
      // lane i of a quad owns row i, columns 4j..4j+3 -> one coalesced dword store per lane
      const uint32_t b = (uint32_t)v[0] | ((uint32_t)v[1 % NPX] << 8) | ((uint32_t)v[2 % NPX] << 16) |
                         ((uint32_t)v[3 % NPX] << 24);
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x00, 0xf, 0xf, false);  // quad_perm 0,0,0,0
      const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x55, 0xf, 0xf, false);  // 1,1,1,1
      const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xaa, 0xf, 0xf, false);  // 2,2,2,2
      const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xff, 0xf, 0xf, false);  // 3,3,3,3
      const uint32_t i = threadIdx.x & 3;
      const uint32_t sel_lo = 0x0c0c0000u | ((4u + i) << 8) | i;           // [b0.byte_i, b1.byte_i, 0, 0]
      const uint32_t sel_hi = 0x00000c0cu | ((4u + i) << 24) | (i << 16);  // [0, 0, b2.byte_i, b3.byte_i]
      const uint32_t w = __builtin_amdgcn_perm(b1, b0, sel_lo) | __builtin_amdgcn_perm(b3, b2, sel_hi);
      *reinterpret_cast<uint32_t*>(d) = w;
    } else {
      // partial tiles only: keep the row stride opaque here, or hipcc turns d + p*dstride into three
      // more 64-bit induction variables that the (hot) dword path then updates every frame
      int rs = dstride;
      asm volatile("" : "+s"(rs));
#pragma unroll
      for (int p = 0; p < NPX; p++)
        if (s.live[p]) d[(size_t)p * rs] = (uint8_t)v[p];
    }
  } else {
    if (s.live[0]) d[0] = (uint8_t)v[0];
  }
}

// Where a lane's output goes.  32x32 tiles: with dword stores lane (x, band) writes row
// 4*band + (x & 3), columns (x & ~3)..+3 after the quad transpose; with byte stores it writes its
// own column x, rows 4*band + 0..3.
template <int NPX>
__device__ __forceinline__ size_t out_pos(const TiledPlane& pl, const TileDesc& t, bool dword_store) {
  const int tid = threadIdx.x;
  int ox, oy;
  if (NPX == 4) {
    // 32x32 tile: 32 columns x 8 bands of 4 rows; 64x16: 64 columns x 4 bands; 128x8 strip: 128 columns x 2 bands
    const int logw = t.kind == kTileStrip128 ? 7 : (t.kind == kTileWide64 ? 6 : 5);
    const int x = tid & ((1 << logw) - 1), band = tid >> logw;
    if (dword_store) {
      ox = t.ox + (x & ~3);
      oy = t.oy + band * 4 + (x & 3);
    } else {
      ox = t.ox + x;
      oy = t.oy + band * 4;
    }
  } else {
    ox = t.ox + (tid & 15);
    oy = t.oy + (tid >> 4);
  }
  return (size_t)oy * pl.dstride + ox;
}

// XCD-aware order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch");
// give every XCD one contiguous range of the raster-ordered tile list so neighbouring tiles --
// whose source boxes overlap by the stencil halo -- share an L2.  Bijective for any n.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int xcd = b & 7, k = b >> 3;
  const int q = n >> 3, rem = n & 7;
  return xcd * q + (xcd < rem ? xcd : rem) + k;
}

// ============================ variant 1: DMA ring (main path) ================================
// Workgroup = 5 waves: wave 4 is the LOADER, waves 0-3 are CONSUMERS (persistent loader/consumer
// split, cdna_hip_programming.md 5.6).  The loader's instruction stream is only address setup,
// global_load_lds_dwordx4 and counted vmcnt waits; the consumers' frame loop is only
// barrier -> ds_read2 -> dot4 -> store.  They meet at ONE s_barrier per frame:
//     loader  : wait until frame i has landed | BARRIER i | refill the slot frame i-1 used
//     consumer:                                 BARRIER i | gather frame i from its slot, store
// The loader's vmcnt stream holds nothing but its own in-order DMA loads, so the count is exact.

constexpr int kLoaderWave = 4;

// optional per-workgroup phase timestamps (debug builds of the schedule, T360_TRACE)
__device__ __forceinline__ void trace_mark(const TiledArgs& a, int slot) {
  if (a.trace && (threadIdx.x & 63) == 0) {
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    a.trace[wg * 8 + slot] = wall_clock64();
  }
}
constexpr int kMaxLoaderInstr = kStageChunksPerLane * 4;  // 16 x (64 lanes x 16 B) = 16 KiB box

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction only takes an immediate):
// a computed jump into a table of 8-byte entries {s_waitcnt vmcnt(k); s_branch out}.  Two taken
// branches per call; a switch() compiles to a ~12-branch decision tree, which is measurable in a
// loop whose whole body is a few hundred nanoseconds.
__device__ __forceinline__ void wait_vmcnt(int n) {
  n = n < 0 ? 0 : (n > 63 ? 63 : n);
  // s_getpc yields the address of the s_add below; the table starts 12 bytes further
  const uint32_t skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(12u + 8u * (uint32_t)n));
#define T360_W(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch 1f\n\t"
  asm volatile(
      "s_getpc_b64 vcc\n\t"
      "s_add_u32 vcc_lo, vcc_lo, %0\n\t"
      "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
      "s_setpc_b64 vcc\n\t"
      T360_W(0) T360_W(1) T360_W(2) T360_W(3) T360_W(4) T360_W(5) T360_W(6) T360_W(7)
      T360_W(8) T360_W(9) T360_W(10) T360_W(11) T360_W(12) T360_W(13) T360_W(14) T360_W(15)
      T360_W(16) T360_W(17) T360_W(18) T360_W(19) T360_W(20) T360_W(21) T360_W(22) T360_W(23)
      T360_W(24) T360_W(25) T360_W(26) T360_W(27) T360_W(28) T360_W(29) T360_W(30) T360_W(31)
      T360_W(32) T360_W(33) T360_W(34) T360_W(35) T360_W(36) T360_W(37) T360_W(38) T360_W(39)
      T360_W(40) T360_W(41) T360_W(42) T360_W(43) T360_W(44) T360_W(45) T360_W(46) T360_W(47)
      T360_W(48) T360_W(49) T360_W(50) T360_W(51) T360_W(52) T360_W(53) T360_W(54) T360_W(55)
      T360_W(56) T360_W(57) T360_W(58) T360_W(59) T360_W(60) T360_W(61) T360_W(62) T360_W(63)
      "1:"
      :
      : "s"(skip)
      : "memory", "vcc", "scc");
#undef T360_W
}

// One frame of one tile, global -> LDS by DMA: nj x (64 lanes x 16 bytes), nj wave-uniform in
// 1..16.  SGPR-base + 32-bit VGPR-offset addressing, so per frame only the scalar base changes;
// M0 (the LDS destination) is written and stepped next to the instruction that uses it.  hipcc
// does not count these loads (cdna_hip_programming.md 5.7): completion is ours to track with
// wait_vmcnt().
// The 16 {load; step M0; nop} triples are 16 bytes each and laid out back to back; a computed jump
// enters the chain at triple 16-nj (Duff's device), so no per-frame decision tree.  Triple k moves
// PIECE 15-k: `off[k]` must hold the source offset of piece 15-k, M0 starts at the last piece's
// destination and walks down.
__device__ __forceinline__ void dma_frame_n(int nj, const uint8_t* frame_base, uint32_t lds_dst, uint32_t lds_step,
                                            const int (&off)[16]) {
  const uint32_t skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(12u + 16u * (uint32_t)(16 - nj)));
  const uint32_t m0_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_dst + (uint32_t)(nj - 1) * lds_step));
#define T360_DMA(k) "global_load_lds_dwordx4 %" #k ", %16\n\ts_sub_u32 m0, m0, %17\n\ts_nop 0\n\t"
  asm volatile(
      "s_mov_b32 m0, %18\n\t"
      "s_getpc_b64 vcc\n\t"
      "s_add_u32 vcc_lo, vcc_lo, %19\n\t"
      "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
      "s_setpc_b64 vcc\n\t"
      T360_DMA(0) T360_DMA(1) T360_DMA(2) T360_DMA(3) T360_DMA(4) T360_DMA(5) T360_DMA(6) T360_DMA(7)
      T360_DMA(8) T360_DMA(9) T360_DMA(10) T360_DMA(11) T360_DMA(12) T360_DMA(13) T360_DMA(14) T360_DMA(15)
      :
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]),
        "v"(off[8]), "v"(off[9]), "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "v"(off[14]), "v"(off[15]),
        "s"(frame_base), "s"(lds_step), "s"(m0_start), "s"(skip)
      : "memory", "vcc", "scc");
#undef T360_DMA
}

__device__ __forceinline__ void frame_barrier() {
  // a bare s_barrier: __syncthreads() would add fences whose waits could drain the DMA ring
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ring geometry of one tile (identical in loader and consumers)
struct RingGeom {
  int pitch, nch, slot_bytes, K;
};
__device__ __forceinline__ RingGeom ring_geom(const TileDesc& t, int ring_bytes) {
  RingGeom g;
  g.pitch = (int)t.cpr * kStageChunk;
  g.nch = (int)t.cpr * (int)t.rows;
  const int nj = (g.nch + 63) >> 6;                      // DMA instructions per frame (1 KiB each)
  g.slot_bytes = nj * 1024 + 64;                         // whole pieces (+ the hi dword of the last window)
  int K = ring_bytes / g.slot_bytes;                     // the plan guarantees >= 2
  K = K > kRingMaxSlots ? kRingMaxSlots : K;
  while (K > 2 && (K - 2) * nj > 63) K--;                // vmcnt is a 6-bit counter
  g.K = K;
  return g;
}

// ---- flag synchronisation (VARIANT bit 4) -------------------------------------------------------
// One s_barrier per frame makes every frame cost the SLOWEST of the four consumer waves (measured:
// wave 0 spends 0.56 us gathering and 0.25 us waiting for its siblings, per frame).  With flags in
// LDS the waves only meet through the ring: the loader publishes `ready` = frames landed, each
// consumer wave publishes `done[w]` = frames it has finished reading, the loader refills a slot once
// min(done) has passed it.  A fast wave may run up to K-1 frames ahead of a slow one.
// ctrl[0] = ready, ctrl[4..7] = done[0..3]; LDS operations of one wave execute in order, so a flag
// written after a wave's reads (or after the loader's vmcnt wait) is ordered behind them.
__device__ __forceinline__ volatile uint32_t* ring_ctrl(const uint8_t* lds, int ring_bytes) {
  return reinterpret_cast<volatile uint32_t*>(const_cast<uint8_t*>(lds) + ring_bytes);
}
__device__ __forceinline__ int poll_at_least(volatile uint32_t* flag, int want, int seen) {
  while (seen < want) {
    seen = __builtin_amdgcn_readfirstlane((int)*flag);
    if (seen < want) __builtin_amdgcn_s_sleep(1);
  }
  return seen;
}
__device__ __forceinline__ int poll_min4_at_least(volatile uint32_t* flags, int want, int seen) {
  while (seen < want) {
    const uint32_t a0 = flags[0], a1 = flags[1], a2 = flags[2], a3 = flags[3];
    seen = __builtin_amdgcn_readfirstlane((int)min(min(a0, a1), min(a2, a3)));
    if (seen < want) __builtin_amdgcn_s_sleep(1);
  }
  return seen;
}

// `which` of `nloaders` loader waves: it owns the 1 KiB pieces j = which, which + nloaders, ...
template <bool FLAGS>
__device__ __forceinline__ void loader_wave(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t,
                                            uint32_t lds_base, int f0, int f1, int which, int nloaders,
                                            volatile uint32_t* ctrl = nullptr) {
  const int lane = threadIdx.x & 63;
  const RingGeom g = ring_geom(t, a.ring_bytes);
  const int nj_all = (g.nch + 63) >> 6;
  const int nj = (nj_all - which + nloaders - 1) / nloaders;  // pieces of this loader (may be 0)
  // chunk q = lane + 64*j lives at LDS byte 16*q of the slot; its source offset inside the plane
  // is fixed per tile.  Lanes past the end of the box re-read chunk 0 into the slot's padding:
  // every DMA instruction runs with all 64 lanes (no exec juggling in the issue path).
  const uint32_t inv = (65536u + (uint32_t)t.cpr - 1u) / (uint32_t)t.cpr;  // exact q / cpr for q < 1024
  int goff[kMaxLoaderInstr];  // goff[k] = source offset of this loader's piece 15-k (dma_frame_n's order)
#pragma unroll
  for (int k = 0; k < kMaxLoaderInstr; k++) {
    const int j = kMaxLoaderInstr - 1 - k;
    goff[k] = 0;
    if (j < nj) {  // wave-uniform
      int q = lane + 64 * (which + j * nloaders);
      q = q < g.nch ? q : 0;
      int r = (int)(((uint32_t)q * inv) >> 16), cc = q - r * (int)t.cpr;
      if (cc >= (int)t.cpr_src) r = cc = 0;  // padding column
      const int sy = wrap_coord(t.y0 + r, pl.sh);
      int sx = t.x0 + cc * kStageChunk;  // multiple of 16; the plane width is a multiple of 16 here
      if (sx < 0)
        sx += pl.sw;
      else if (sx >= pl.sw)
        sx -= pl.sw;
      goff[k] = sy * pl.sstride + sx;
    }
  }
  auto issue = [&](int f, int slot) {
    if (nj > 0) dma_frame_n(nj, pl.src + (size_t)f * pl.src_frame_bytes,
                lds_base + (uint32_t)(slot * g.slot_bytes + which * 1024), (uint32_t)(nloaders * 1024), goff);
  };
  const int nf = f1 - f0;
  const int K = g.K;
  if (which == 0) trace_mark(a, 1);
  if (!(a.debug & 256))
    for (int j = 0; j < K - 1 && j < nf; j++) issue(f0 + j, j);
  if (which == 0) trace_mark(a, 2);
  int fill = (K - 1) % K;
  int min_done = 0;
  unsigned long long acc_wait = 0, acc_bar = 0, acc_issue = 0;
  const bool tracing = a.trace != nullptr && (a.debug & 16);
  for (int i = 0; i < nf; i++) {
    // loads younger than frame i's: frames i+1 .. min(i+K-2, nf-1), nj instructions each
    unsigned long long c0 = tracing ? wall_clock64() : 0;
    wait_vmcnt((a.debug & 4) ? 0 : min(K - 2, nf - 1 - i) * nj);
    unsigned long long c1 = tracing ? wall_clock64() : 0;
    if (i == 0 && which == 0 && !tracing) trace_mark(a, 3);
    if (FLAGS) {
      if (lane == 0) ctrl[0] = (uint32_t)(i + 1);  // frame i has landed
      // the slot frame i-1 used is refilled next: every consumer wave must have left it
      if (i + K - 1 < nf) min_done = poll_min4_at_least(ctrl + 4, i, min_done);
    } else {
      frame_barrier();  // frame i is visible to the consumers; they have left frame i-1's slot
    }
    unsigned long long c2 = tracing ? wall_clock64() : 0;
    if (i + K - 1 < nf && !(a.debug & 4)) issue(f0 + i + K - 1, fill);
    fill = fill + 1 == K ? 0 : fill + 1;
    if (tracing) {
      acc_wait += c1 - c0;
      acc_bar += c2 - c1;
      acc_issue += wall_clock64() - c2;
    }
  }
  if (tracing && which == 0 && (threadIdx.x & 63) == 0) {
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    a.trace[wg * 8 + 1] = acc_wait;   // loader: total time in counted vmcnt waits
    a.trace[wg * 8 + 2] = acc_bar;    // loader: total time at the frame barrier (waiting for consumers)
    a.trace[wg * 8 + 3] = acc_issue;  // loader: total time issuing DMA
    a.trace[wg * 8 + 4] = (unsigned long long)(nj * 1000 + K);
  }
  if (which == 0 && !(a.debug & (32 | 16))) trace_mark(a, 6);
}

// CPR > 0: the tile's LDS row pitch (16 * CPR bytes) is a compile-time constant, so the four stencil
// rows of a pixel are immediate offsets of ONE address (12 v_add fewer per 4 pixels); 0 = run time.
template <int NPX, int KS, int GROUP, bool FLAGS = false, int CPR = 0>
__device__ __forceinline__ void consumer_waves(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t,
                                               const uint8_t* __restrict__ lds, int f0, int f1) {
  RingGeom g = ring_geom(t, a.ring_bytes);
  if (CPR > 0) g.pitch = CPR * kStageChunk;
  PixelSetup<NPX, KS> px;
  load_pixels<NPX, KS>(pl, a.wpack, t, g.pitch, px, a.debug);
  if (a.trace && !(a.debug & 16)) {
    pin_pixels<NPX, KS>(px);  // make the compiler wait for the loads before the timestamp
    if (threadIdx.x < 64) trace_mark(a, 4);
  }
  const bool dword_store = NPX == 4 && !(t.flags & kTilePartial) && pl.dst_dword_ok;
  uint8_t* __restrict__ d = pl.dst + (size_t)f0 * pl.dst_frame_bytes + out_pos<NPX>(pl, t, dword_store);
  const uint8_t* __restrict__ box = lds;
  const uint8_t* const ring_end = lds + g.K * g.slot_bytes;
  const int nf = f1 - f0;
  unsigned long long acc_bar = 0, acc_work = 0;
  const bool tracing = a.trace != nullptr && (a.debug & 16);
  volatile uint32_t* const ctrl = ring_ctrl(lds, a.ring_bytes);
  const int my_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int ready = 0;
  if (FLAGS) pin_pixels<NPX, KS>(px);  // hipcc's wait for its own loads: before the loop, not inside it
  for (int i = 0; i < nf; i++) {
    unsigned long long c0 = tracing ? wall_clock64() : 0;
    if (FLAGS)
      ready = poll_at_least(ctrl, i + 1, ready);  // frame i is complete in LDS
    else
      frame_barrier();
    unsigned long long c1 = tracing ? wall_clock64() : 0;
    if (!(a.debug & 8)) gather_store<NPX, KS, GROUP>(px, box, g.pitch, d, pl.dstride, dword_store);
    if (FLAGS) {
      asm volatile("" ::: "memory");
      if ((threadIdx.x & 63) == 0) ctrl[4 + my_wave] = (uint32_t)(i + 1);  // behind this wave's reads of frame i
    }
    if (tracing) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc_bar += c1 - c0;
      acc_work += wall_clock64() - c1;
    }
    d += pl.dst_frame_bytes;
    box += g.slot_bytes;
    if (box == ring_end) box = lds;
    if (i == 0 && threadIdx.x < 64 && !tracing) trace_mark(a, 5);
  }
  if (threadIdx.x < 64) trace_mark(a, 7);
  if (tracing && threadIdx.x == 0) {
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    a.trace[wg * 8 + 5] = acc_bar;   // consumer wave 0: total time at the frame barrier
    a.trace[wg * 8 + 6] = acc_work;  // consumer wave 0: total gather + store issue time
  }
}

// ---- variant without a loader wave: every consumer wave also moves a quarter of the box -------
// Workgroup = 4 waves (one per SIMD, so any number of workgroups spreads evenly over the SIMDs;
// the 5-wave loader/consumer workgroup leaves the 4th slot of a CU empty most of the time).
// Wave w owns the 1 KiB pieces w, w+4, w+8, w+12 of every frame.  Per frame:
//     wait until MY pieces of frame i have landed | BARRIER i | refill the slot frame i-1 used
//     with my pieces of frame i+K-1 | gather frame i, store
// The wave's vmcnt stream now also holds its output stores, which may complete out of order with
// the loads.  The counted wait stays SAFE: loads complete in order among themselves, so
// "at most D operations outstanding", D = my loads younger than frame i's, implies frame i's
// pieces are done whatever the stores do; outstanding stores only make the wait conservative.
__device__ __forceinline__ void dma_frame_4(int nj, const uint8_t* frame_base, uint32_t lds_dst, uint32_t lds_step,
                                            const int (&off)[4]) {
  const uint32_t skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(12u + 16u * (uint32_t)(4 - nj)));
  const uint32_t m0_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_dst + (uint32_t)(nj - 1) * lds_step));
#define T360_DMA(k) "global_load_lds_dwordx4 %" #k ", %4\n\ts_sub_u32 m0, m0, %5\n\ts_nop 0\n\t"
  asm volatile(
      "s_mov_b32 m0, %6\n\t"
      "s_getpc_b64 vcc\n\t"
      "s_add_u32 vcc_lo, vcc_lo, %7\n\t"
      "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
      "s_setpc_b64 vcc\n\t"
      T360_DMA(0) T360_DMA(1) T360_DMA(2) T360_DMA(3)
      :
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(frame_base), "s"(lds_step), "s"(m0_start), "s"(skip)
      : "memory", "vcc", "scc");
#undef T360_DMA
}

template <int NPX, int GROUP>
__device__ __forceinline__ void self_loading_waves(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t,
                                                   const uint8_t* __restrict__ lds, int f0, int f1) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const RingGeom g = ring_geom(t, a.ring_bytes);
  const int nj_all = (g.nch + 63) >> 6;
  const int nj = (nj_all - wave + 3) >> 2;  // my pieces: wave, wave + 4, ... (0..4 of them)
  const uint32_t inv = (65536u + (uint32_t)t.cpr - 1u) / (uint32_t)t.cpr;  // exact q / cpr for q < 1024
  int goff[4];  // goff[k] = source offset of my piece 3-k (dma_frame_4 walks its chain backwards)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int j = 3 - k;
    goff[k] = 0;
    if (j < nj) {  // wave-uniform
      int q = lane + 64 * (wave + 4 * j);
      q = q < g.nch ? q : 0;  // lanes past the end of the box re-read chunk 0 into the slot's padding
      int r = (int)(((uint32_t)q * inv) >> 16), cc = q - r * (int)t.cpr;
      if (cc >= (int)t.cpr_src) r = cc = 0;  // padding column
      const int sy = wrap_coord(t.y0 + r, pl.sh);
      int sx = t.x0 + cc * kStageChunk;
      if (sx < 0)
        sx += pl.sw;
      else if (sx >= pl.sw)
        sx -= pl.sw;
      goff[k] = sy * pl.sstride + sx;
    }
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
  auto issue = [&](int f, int slot) {
    if (nj > 0)
      dma_frame_4(nj, pl.src + (size_t)f * pl.src_frame_bytes, lds_base + (uint32_t)(slot * g.slot_bytes + wave * 1024),
                  4096u, goff);
  };
  const int nf = f1 - f0;
  const int K = g.K;
  if (!(a.debug & 256))
    for (int j = 0; j < K - 1 && j < nf; j++) issue(f0 + j, j);
  if (lane == 0 && wave == 0) trace_mark(a, 2);

  PixelSetup<NPX, 4> px;
  load_pixels<NPX, 4>(pl, a.wpack, t, g.pitch, px, a.debug);
  pin_pixels<NPX, 4>(px);  // hipcc's wait for its own loads lands here (and drains the prologue DMA with it)
  if (lane == 0 && wave == 0) trace_mark(a, 4);

  const bool dword_store = NPX == 4 && !(t.flags & kTilePartial) && pl.dst_dword_ok;
  uint8_t* __restrict__ d = pl.dst + (size_t)f0 * pl.dst_frame_bytes + out_pos<NPX>(pl, t, dword_store);
  const uint8_t* __restrict__ box = lds;
  const uint8_t* const ring_end = lds + K * g.slot_bytes;
  int fill = (K - 1) % K;
  for (int i = 0; i < nf; i++) {
    // my loads younger than frame i's: frames i+1 .. min(i+K-2, nf-1), nj instructions each
    if (nj > 0 && !(a.debug & 512)) wait_vmcnt((a.debug & 4) ? 0 : min(K - 2, nf - 1 - i) * nj);
    if (!(a.debug & 1024)) frame_barrier();  // frame i is complete in LDS; everyone has left frame i-1's slot
    if (i + K - 1 < nf && !(a.debug & 4)) issue(f0 + i + K - 1, fill);
    fill = fill + 1 == K ? 0 : fill + 1;
    if (!(a.debug & 8)) gather_store<NPX, 4, GROUP>(px, box, g.pitch, d, pl.dstride, dword_store);
    d += pl.dst_frame_bytes;
    box += g.slot_bytes;
    if (box == ring_end) box = lds;
    if (i == 0 && lane == 0 && wave == 0) trace_mark(a, 5);
  }
  if (lane == 0 && wave == 0) trace_mark(a, 7);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void remap_tiled_cubic_self_kernel(TiledArgs a) {
  constexpr int GROUP = (VARIANT & 1) ? 2 : 4;
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  int b = xcd_contiguous(blockIdx.x, a.total_tiles);
  TiledPlane pl = a.plane[0];
  if (a.nplanes > 1 && b >= pl.ntiles) {
    b -= pl.ntiles;
    pl = a.plane[1];
    if (a.nplanes > 2 && b >= pl.ntiles) {
      b -= pl.ntiles;
      pl = a.plane[2];
      if (a.nplanes > 3 && b >= pl.ntiles) {
        b -= pl.ntiles;
        pl = a.plane[3];
      }
    }
  }
  const TileDesc t = pl.tiles[b];
  const int f0 = blockIdx.y * a.frames_per_block;
  const int f1 = min(f0 + a.frames_per_block, a.nframes);
  if (threadIdx.x == 0) trace_mark(a, 0);
  if (a.trace && (a.debug & 32) && threadIdx.x == 0) {  // where did this workgroup run?
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
  if (t.kind == kTileStaged16)
    self_loading_waves<1, GROUP>(a, pl, t, lds, f0, f1);
  else
    self_loading_waves<4, GROUP>(a, pl, t, lds, f0, f1);
}

// VARIANT bit 0: LDS reads in groups of 2 pixels instead of 4 (78 instead of 92 VGPRs: 4 workgroups per CU);
//         bit 4: LDS flags instead of the per-frame barrier; bit 5: instrumented (debug / trace) build
//         KS: taps per axis of the interpolation (1, 2, 4, 8)
template <int VARIANT, int KS>
__global__ __launch_bounds__(512, 1) void remap_tiled_dma_kernel(TiledArgs a) {
  constexpr int GROUP = (VARIANT & 1) ? 2 : 4;
  // VARIANT bit 5: the instrumented build (T360_DEBUG / T360_TRACE).  In the production build the
  // switches are compile-time zero, so none of their branches is left in the frame loops.
  if (!(VARIANT & 32)) {
    a.debug = 0;
    a.trace = nullptr;
  }
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  int b = xcd_contiguous(blockIdx.x, a.total_tiles);
  // pick the plane with scalar selects: indexing a.plane[] with a run-time index would make
  // hipcc copy the whole argument block to scratch
  TiledPlane pl = a.plane[0];
  if (a.nplanes > 1 && b >= pl.ntiles) {
    b -= pl.ntiles;
    pl = a.plane[1];
    if (a.nplanes > 2 && b >= pl.ntiles) {
      b -= pl.ntiles;
      pl = a.plane[2];
      if (a.nplanes > 3 && b >= pl.ntiles) {
        b -= pl.ntiles;
        pl = a.plane[3];
      }
    }
  }
  const TileDesc t = pl.tiles[b];
  const int f0 = blockIdx.y * a.frames_per_block;
  const int f1 = min(f0 + a.frames_per_block, a.nframes);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave == 0) trace_mark(a, 0);
  if (a.trace && (a.debug & 32) && threadIdx.x == 0) {  // where did this workgroup run?
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
  constexpr bool FLAGS = (VARIANT & 16) != 0;
  if (FLAGS) {  // flag block behind the ring: ready = 0, done[] = 0
    if (threadIdx.x < 8) ring_ctrl(lds, a.ring_bytes)[threadIdx.x] = 0u;
    __syncthreads();
  }
  if (wave >= kLoaderWave) {
    loader_wave<FLAGS>(a, pl, t, (uint32_t)(uintptr_t)lds, f0, f1, wave - kLoaderWave, (int)(blockDim.x >> 6) - kLoaderWave,
                       ring_ctrl(lds, a.ring_bytes));
  } else if (KS == 8 || t.kind == kTileStaged16) {
    // Lanczos4 is planned as 16x16 tiles only (32 weight VGPRs per pixel)
    if (KS == 4 && !FLAGS && t.cpr == 4)
      consumer_waves<1, KS, GROUP, FLAGS, (KS == 4 ? 4 : 0)>(a, pl, t, lds, f0, f1);
    else
      consumer_waves<1, KS, GROUP, FLAGS>(a, pl, t, lds, f0, f1);
  } else {
    // 32x32 tiles and 128x8 strips; the common pitches of the bicubic workloads are specialised
    constexpr int N4 = KS == 8 ? 1 : 4;
    constexpr bool SPEC = KS == 4 && !FLAGS;
    switch (SPEC ? (int)t.cpr : 0) {
      case 4: consumer_waves<N4, KS, GROUP, FLAGS, SPEC ? 4 : 0>(a, pl, t, lds, f0, f1); break;
      case 5: consumer_waves<N4, KS, GROUP, FLAGS, SPEC ? 5 : 0>(a, pl, t, lds, f0, f1); break;
      case 6: consumer_waves<N4, KS, GROUP, FLAGS, SPEC ? 6 : 0>(a, pl, t, lds, f0, f1); break;
      case 10: consumer_waves<N4, KS, GROUP, FLAGS, SPEC ? 10 : 0>(a, pl, t, lds, f0, f1); break;
      case 11: consumer_waves<N4, KS, GROUP, FLAGS, SPEC ? 11 : 0>(a, pl, t, lds, f0, f1); break;
      case 12: consumer_waves<N4, KS, GROUP, FLAGS, SPEC ? 12 : 0>(a, pl, t, lds, f0, f1); break;
      default: consumer_waves<N4, KS, GROUP, FLAGS>(a, pl, t, lds, f0, f1); break;
    }
  }
}

// ---- persistent variant: one workgroup per resident slot, work pulled from per-XCD queues ------
// The grid-per-tile launch pays a workgroup turnover (~2-3 us between a workgroup's end and its
// successor's first instruction, measured with T360_TRACE) for every 16 frames of one tile, and
// its tail is a whole wave of workgroups.  Here a workgroup stays resident and pulls
// (frame group, tile) items: XCD x owns the tile range xcd_contiguous() gives it (neighbouring
// tiles share their source halo in that XCD's L2) and walks it frame group by frame group; when
// its own queue is empty it takes items from the other XCDs' queues, so the tail shrinks to one item.
template <int VARIANT>
__global__ __launch_bounds__(512, 1) void remap_tiled_cubic_persist_kernel(TiledArgs a) {
  constexpr int GROUP = (VARIANT & 1) ? 2 : 4;
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  __shared__ int next_item[2];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ngroups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int my_xcd = (int)(xcc & 7);
  const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
  int victim = 0;  // how many queues (starting with my own) are already known to be empty
  for (;;) {
    if (threadIdx.x == 0) {
      int got = -1, got_xcd = 0;
      while (victim < 8) {
        const int x = (my_xcd + victim) & 7;
        const int len = q + (x < rem ? 1 : 0);
        const int li = atomicAdd(a.work_counters + x, 1);
        if (li < len * ngroups) {
          got = li;
          got_xcd = x;
          break;
        }
        victim++;
      }
      next_item[0] = got;
      next_item[1] = got_xcd;
    }
    __syncthreads();  // item boundary: everyone has also left the ring
    const int li = __builtin_amdgcn_readfirstlane(next_item[0]);
    const int x = __builtin_amdgcn_readfirstlane(next_item[1]);
    __syncthreads();
    if (li < 0) break;
    const int len = q + (x < rem ? 1 : 0);
    const int start = x * q + (x < rem ? x : rem);
    const int g = li / len;
    int b = start + (li - g * len);
    TiledPlane pl = a.plane[0];
    if (a.nplanes > 1 && b >= pl.ntiles) {
      b -= pl.ntiles;
      pl = a.plane[1];
      if (a.nplanes > 2 && b >= pl.ntiles) {
        b -= pl.ntiles;
        pl = a.plane[2];
        if (a.nplanes > 3 && b >= pl.ntiles) {
          b -= pl.ntiles;
          pl = a.plane[3];
        }
      }
    }
    const TileDesc t = pl.tiles[b];
    const int f0 = g * a.frames_per_block;
    const int f1 = min(f0 + a.frames_per_block, a.nframes);
    if (wave >= kLoaderWave) {
      loader_wave<false>(a, pl, t, (uint32_t)(uintptr_t)lds, f0, f1, wave - kLoaderWave, (int)(blockDim.x >> 6) - kLoaderWave);
    } else if (t.kind == kTileStaged16) {
      consumer_waves<1, 4, GROUP>(a, pl, t, lds, f0, f1);
    } else {
      consumer_waves<4, 4, GROUP>(a, pl, t, lds, f0, f1);
    }
  }
}

// ===================== variant 2: chunks staged through registers ============================
// For planes whose base / stride / width are not 16-byte friendly: chunks that are not one
// aligned dwordx4 are assembled byte by byte with BORDER_WRAP.

__device__ __noinline__ uint4 fetch_wrapped(const uint8_t* __restrict__ row, int sx, int sw) {
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t acc = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) acc |= (uint32_t)row[wrap_coord(sx + k * 4 + b, sw)] << (8 * b);
    w[k] = acc;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int NPX>
__device__ __forceinline__ void staged_tile_regs(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t,
                                                 uint8_t* __restrict__ lds, int f0, int f1) {
  const int tid = threadIdx.x;
  const int pitch = (int)t.cpr * kStageChunk;
  PixelSetup<NPX, 4> px;
  load_pixels<NPX, 4>(pl, a.wpack, t, pitch, px);

  const int nch = (int)t.cpr * (int)t.rows;
  int goff[kStageChunksPerLane];  // fast chunk: byte offset inside the plane; slow: (row << 16) | col chunk
  int loff[kStageChunksPerLane];  // LDS byte offset, -1 = lane has no such chunk
  bool fast[kStageChunksPerLane];
#pragma unroll
  for (int c = 0; c < kStageChunksPerLane; c++) {
    const int q = tid + c * 256;
    loff[c] = -1;
    goff[c] = 0;
    fast[c] = false;
    if (q < nch) {
      int r = q / (int)t.cpr, cc = q - r * (int)t.cpr;
      loff[c] = q * kStageChunk;
      if (cc >= (int)t.cpr_src) r = cc = 0;  // padding column
      const int sy = wrap_coord(t.y0 + r, pl.sh);
      int sx = t.x0 + cc * kStageChunk;
      if (sx + kStageChunk <= 0)
        sx += pl.sw;
      else if (sx >= pl.sw)
        sx -= pl.sw;
      fast[c] = pl.src_vec_ok && sx >= 0 && sx + kStageChunk <= pl.sw && (sx & 15) == 0;
      goff[c] = fast[c] ? sy * pl.sstride + sx : ((r << 16) | cc);
    }
  }
  uint4 stage[kStageChunksPerLane];
  auto fetch = [&](int f) {
    const uint8_t* __restrict__ base = pl.src + (size_t)f * pl.src_frame_bytes;
#pragma unroll
    for (int c = 0; c < kStageChunksPerLane; c++) {
      if (loff[c] < 0) continue;
      if (fast[c]) {
        stage[c] = *reinterpret_cast<const uint4*>(base + goff[c]);
      } else {
        const int r = goff[c] >> 16, cc = goff[c] & 0xffff;
        const int sy = wrap_coord(t.y0 + r, pl.sh);
        stage[c] = fetch_wrapped(base + (size_t)sy * pl.sstride, t.x0 + cc * kStageChunk, pl.sw);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int c = 0; c < kStageChunksPerLane; c++)
      if (loff[c] >= 0) *reinterpret_cast<uint4*>(lds + loff[c]) = stage[c];
  };

  const bool dword_store = NPX == 4 && !(t.flags & kTilePartial) && pl.dst_dword_ok;
  const size_t dpos = out_pos<NPX>(pl, t, dword_store);
  fetch(f0);
  commit();
  __syncthreads();
  for (int f = f0; f < f1; f++) {
    if (f + 1 < f1) fetch(f + 1);  // in flight while this frame is computed
    gather_store<NPX, 4, 4>(px, lds, pitch, pl.dst + (size_t)f * pl.dst_frame_bytes + dpos, pl.dstride, dword_store);
    __syncthreads();  // everyone is done reading this frame's box
    if (f + 1 < f1) {
      commit();
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void remap_tiled_cubic_regs_kernel(TiledArgs a) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const TiledPlane pl = a.plane[0];
  const TileDesc t = pl.tiles[xcd_contiguous(blockIdx.x, pl.ntiles)];
  const int f0 = blockIdx.y * a.frames_per_block;
  const int f1 = min(f0 + a.frames_per_block, a.nframes);
  if (t.kind == kTileStaged16)
    staged_tile_regs<1>(a, pl, t, lds, f0, f1);
  else
    staged_tile_regs<4>(a, pl, t, lds, f0, f1);
}

// ===================== tiles too large to stage: direct gather ================================
// The few 16x16 tiles around each pole whose source box exceeds the staging budget (they span a
// full quadrant of longitudes): one workgroup per (tile, frame), one pixel per
// lane, 16 independent byte loads in flight per lane.
template <int KS>
__global__ __launch_bounds__(256) void remap_direct_kernel(TiledArgs a) {
  int b = blockIdx.x;
  TiledPlane pl = a.plane[0];
  if (a.nplanes > 1 && b >= pl.ndirect) {
    b -= pl.ndirect;
    pl = a.plane[1];
    if (a.nplanes > 2 && b >= pl.ndirect) {
      b -= pl.ndirect;
      pl = a.plane[2];
      if (a.nplanes > 3 && b >= pl.ndirect) {
        b -= pl.ndirect;
        pl = a.plane[3];
      }
    }
  }
  const TileDesc t = pl.tiles[pl.ntiles + b];
  const int tid = threadIdx.x;
  const int ox = t.ox + (tid & 15), oy = t.oy + (tid >> 4);
  if (ox >= pl.dw || oy >= pl.dh) return;
  const int f = blockIdx.y;
  const LutEntry e = pl.lut[(size_t)oy * pl.dw + ox];
  const int v = sample<KS, false>(pl.src + (size_t)f * pl.src_frame_bytes, pl.sw, pl.sh, pl.sstride, a.wtab, e);
  pl.dst[(size_t)f * pl.dst_frame_bytes + (size_t)oy * pl.dstride + ox] = (uint8_t)v;
}

}  // namespace

hipError_t launch_remap_direct_cubic(const TiledArgs& a, hipStream_t stream) {
  int total = 0;
  for (int k = 0; k < a.nplanes; k++) total += a.plane[k].ndirect;
  if (total <= 0 || a.nframes <= 0) return hipSuccess;
  const dim3 grid(total, a.nframes, 1);
  switch (a.ks) {
    case 1: hipLaunchKernelGGL(remap_direct_kernel<1>, grid, dim3(256), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(remap_direct_kernel<2>, grid, dim3(256), 0, stream, a); break;
    case 8: hipLaunchKernelGGL(remap_direct_kernel<8>, grid, dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL(remap_direct_kernel<4>, grid, dim3(256), 0, stream, a); break;
  }
  return hipGetLastError();
}

template <int VARIANT, int KS>
static hipError_t launch_dma_variant(const TiledArgs& a, int groups, int nload, hipStream_t stream) {
  static std::atomic<int> configured_lds{0};  // handles on several host threads may launch concurrently
  if (a.ring_bytes > 64 * 1024 && configured_lds < a.ring_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_dma_kernel<VARIANT, KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.ring_bytes);
    if (e != hipSuccess) return e;
    configured_lds = a.ring_bytes;
  }
  hipLaunchKernelGGL((remap_tiled_dma_kernel<VARIANT, KS>), dim3(a.total_tiles, groups, 1), dim3(256 + 64 * nload),
                     (size_t)a.ring_bytes + ((VARIANT & 16) ? 64 : 0), stream, a);
  return hipGetLastError();
}

hipError_t launch_remap_tiled_cubic_dma(const TiledArgs& a, hipStream_t stream) {
  if (a.total_tiles <= 0 || a.nframes <= 0) return hipSuccess;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  const int nload = a.loader_waves < 1 ? 1 : (a.loader_waves > 4 ? 4 : a.loader_waves);
  if ((a.variant & 4) && a.ks == 4) {
    static std::atomic<int> configured_self{0};
    if (a.ring_bytes > 64 * 1024 && configured_self < a.ring_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_cubic_self_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, a.ring_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_cubic_self_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, a.ring_bytes);
      if (e != hipSuccess) return e;
      configured_self = a.ring_bytes;
    }
    if (a.variant & 1)
      hipLaunchKernelGGL(remap_tiled_cubic_self_kernel<1>, dim3(a.total_tiles, groups, 1), dim3(256), (size_t)a.ring_bytes,
                         stream, a);
    else
      hipLaunchKernelGGL(remap_tiled_cubic_self_kernel<0>, dim3(a.total_tiles, groups, 1), dim3(256), (size_t)a.ring_bytes,
                         stream, a);
    return hipGetLastError();
  }
  if ((a.variant & 8) && a.ks == 4) {
    // persistent workgroups: a.work_counters (8 ints) must be zero at launch
    if (!a.work_counters || a.persist_slots <= 0) return hipErrorInvalidValue;
    static std::atomic<int> configured_persist{0};
    if (a.ring_bytes > 60 * 1024 && configured_persist < a.ring_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_cubic_persist_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, a.ring_bytes);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_cubic_persist_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, a.ring_bytes);
      if (e != hipSuccess) return e;
      configured_persist = a.ring_bytes;
    }
    const int items = a.total_tiles * groups;
    const int grid = items < a.persist_slots ? items : a.persist_slots;
    TiledArgs p = a;
    p.trace = nullptr;  // trace slots are indexed by blockIdx
    if (a.variant & 1)
      hipLaunchKernelGGL(remap_tiled_cubic_persist_kernel<1>, dim3(grid), dim3(256 + 64 * nload), (size_t)a.ring_bytes, stream, p);
    else
      hipLaunchKernelGGL(remap_tiled_cubic_persist_kernel<0>, dim3(grid), dim3(256 + 64 * nload), (size_t)a.ring_bytes, stream, p);
    return hipGetLastError();
  }
  const bool instrumented = a.debug != 0 || a.trace != nullptr;
  if (a.ks == 1) return instrumented ? launch_dma_variant<33, 1>(a, groups, nload, stream) : launch_dma_variant<1, 1>(a, groups, nload, stream);
  if (a.ks == 2) return instrumented ? launch_dma_variant<33, 2>(a, groups, nload, stream) : launch_dma_variant<1, 2>(a, groups, nload, stream);
  if (a.ks == 8) return instrumented ? launch_dma_variant<33, 8>(a, groups, nload, stream) : launch_dma_variant<1, 8>(a, groups, nload, stream);
  if ((a.variant & 16) && nload == 1) return launch_dma_variant<17, 4>(a, groups, nload, stream);
  switch ((a.variant & 1) | (instrumented ? 32 : 0)) {
    case 0: return launch_dma_variant<0, 4>(a, groups, nload, stream);
    case 1: return launch_dma_variant<1, 4>(a, groups, nload, stream);
    case 32: return launch_dma_variant<32, 4>(a, groups, nload, stream);
    default: return launch_dma_variant<33, 4>(a, groups, nload, stream);
  }
}

hipError_t launch_remap_tiled_cubic_regs(const TiledArgs& a, hipStream_t stream) {
  if (a.plane[0].ntiles <= 0 || a.nframes <= 0) return hipSuccess;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  const size_t lds = (size_t)kStageMaxBytes + 64;
  hipLaunchKernelGGL(remap_tiled_cubic_regs_kernel, dim3(a.plane[0].ntiles, groups, 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace t360
