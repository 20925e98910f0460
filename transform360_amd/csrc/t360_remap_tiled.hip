// t360_remap_tiled.hip -- LDS-tiled gather over a batch of frames (the hot kernel).
//
// Same arithmetic as t360_remap.hip (cv::remap, BORDER_WRAP, Q15 weights, (sum + 16384) >> 15,
// SURVEY.md Appendix A.3/A.4); written around the bicubic 4x4 stencil and instantiated for nearest,
// bilinear and Lanczos4 as well.  Organised for MI355X:
//
//   * one workgroup (4 consumer waves + 1 loader wave) owns one OUTPUT tile (64x16, 32x32 or 128x8 px with
//     4 px per lane, lane = column; 16x16 px with 1 px per lane near the poles; the plan's choice,
//     t360_plan.cpp) and walks `frames_per_block` frames of the batch with it.  Everything that depends only
//     on geometry -- the LDS address of each stencil row of each pixel, the 16 Q15 weights per pixel, the
//     source addresses of the chunks it stages -- is set up ONCE per tile and kept in registers for all
//     frames: per frame a lane only moves bytes and issues dot products.
//   * per frame the tile's source FOOTPRINT (exactly the 16-byte chunks its stencils touch, packed row after
//     row; planned at init) goes global -> LDS by DMA (global_load_lds_dwordx4, no VGPR round trip) into a
//     ring of K slots, K-1 frames ahead of the frame being computed; completion is tracked with counted
//     s_waitcnt vmcnt(N) and ONE workgroup barrier per frame.  The +-180 degree seam and BORDER_WRAP across
//     the poles are resolved in the chunk addresses, so the gather itself never wraps.
//   * every chunk is written to LDS TWICE: copy A at its natural position, copy B four bytes further (LDS-DMA
//     accepts any dword-aligned destination).  A 4-byte stencil-row window at byte offset o then lies inside
//     ONE 8-byte aligned qword of copy A (o % 8 < 4) or copy B (o % 8 >= 4) and is fetched with ds_read_b64,
//     which costs half the LDS cycles of the two aligned dwords (ds_read2_b32) a single copy needs
//     (MI355X_MICROARCH.md "LDS"; tools/ubench/lds_patterns.hip).  The second write is an L1 hit.
//   * the ring slots have a compile-time size and the frame loop is unrolled over them, so the slot base is
//     an immediate of the ds_read: no per-frame address arithmetic at all.
//   * the 4x4 stencil of one output pixel costs 4 ds_read_b64 + 4 v_alignbit and 8 v_dot4: weights are split
//     into a signed high byte and an unsigned low byte (w = 256*wh + wl) and pixels enter the high part as
//     p-128,   SUM p*w = 256 * (SUM (p-128)*wh + 128 * SUM wh) + SUM p*wl,
//     all exact in int32, so results are bit-identical to the integer formulation.
//   * all planes of the frame (Y, U, V) are tiles of ONE launch, including the few tiles around the poles
//     that are gathered straight from global memory; workgroups are numbered so that every XCD gets a
//     contiguous range of the execution-ordered tile list (shared halo -> shared L2).
//   * no MFMA: this is a gather, not a contraction.
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"
#include "t360_sample.h"

namespace t360 {

namespace {

constexpr int kLoaderWave = 4;

// ---- ring geometry (compile time) -------------------------------------------------------------
// PMAX = largest staged region of a tile in 1 KiB pieces (the plan's max_pieces); DUAL = two copies.
template <int PMAX, bool DUAL>
struct Ring {
  static constexpr int kCopy = PMAX * 1024;
  static constexpr int kCopyB = kCopy + 4;  // copy B: the same bytes, 4 further (8-byte aligned odd dwords)
  static constexpr int kSlot = (DUAL ? 2 * kCopy : kCopy) + 64;
};

// ---- per-pixel geometry -----------------------------------------------------------------------
// KS = taps per axis: 1 nearest, 2 bilinear, 4 bicubic, 8 Lanczos4.  A stencil row is read as WIN 4-byte
// windows (bilinear uses the first two bytes of its window; the packed weights of the other two are zero).
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
  uint32_t addr[NPX][KS];  // LDS byte address (inside a slot) of the aligned qword holding each stencil row's window
  uint32_t sh[NPX];        // bit shift of the window inside that qword (0, 8, 16, 24)
  uint32_t wh[NPX][NWA];   // signed high bytes of the weights, one dword per 4-byte window
  uint32_t wl[NPX][NWA];   // unsigned low bytes
  int hb[NPX];             // 128 * SUM(wh): initial value of the high accumulator
  bool live[NPX];          // pixel inside the plane (partial tiles)
};

template <int NPX, int KS, int PMAX>
__device__ __forceinline__ void load_pixels(const TiledPlane& pl, const uint32_t* __restrict__ wpack, const TileDesc& t,
                                            PixelSetup<NPX, KS>& s) {
  constexpr int NW = Stencil<KS>::NW;
  const int tid = threadIdx.x;
  uint32_t words[4];
  if (NPX == 4) {
    const uint4 v = reinterpret_cast<const uint4*>(pl.tlut + t.tlut)[tid];
    words[0] = v.x; words[1] = v.y; words[2] = v.z; words[3] = v.w;
  } else {
    words[0] = pl.tlut[t.tlut + tid];
  }
  // row table: int16 per box row behind the tile's chunk table
  const uint16_t* __restrict__ rowtab = reinterpret_cast<const uint16_t*>(pl.chunks + t.chunks + (int)t.pieces * kPieceChunks);
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    const uint32_t e = words[p];
    s.live[p] = (e >> 31) == 0;
    const int x = e & 2047, row = s.live[p] ? (int)((e >> kWordRowShift) & 255) : 0;
    const int frac = (e >> kWordFracShift) & 1023;
    s.sh[p] = (uint32_t)(x & 3) * 8u;
#pragma unroll
    for (int k = 0; k < KS; k++) {
      const int off = s.live[p] ? (int)(int16_t)rowtab[row + k] * kStageChunk + x : 0;
      // KS == 1 reads the byte itself from copy A; otherwise the aligned qword of copy A or B that holds the window
      s.addr[p][k] = KS == 1 ? (uint32_t)off : (uint32_t)((off & ~3) + ((off & 4) ? Ring<PMAX, true>::kCopyB : 0));
    }
    s.hb[p] = 0;
    if (NW > 0) {
      // [NW high dwords][NW low dwords][128 * SUM(wh)], 16-byte aligned: 2*NW/4 vector loads + 1
      const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(wpack + (size_t)frac * Stencil<KS>::PACK);
      uint32_t w[2 * (NW > 0 ? NW : 2)];
#pragma unroll
      for (int k = 0; k < (2 * NW) / 4; k++) {
        const uint4 v = wp[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < NW; k++) {
        s.wh[p][k] = w[k];
        s.wl[p][k] = w[NW + k];
      }
      s.hb[p] = (int)wpack[(size_t)frac * Stencil<KS>::PACK + 2 * NW];
    }
  }
}

__device__ __forceinline__ uint32_t bias128(uint32_t px4) { return px4 ^ 0x80808080u; }

// SUM over the windows of one pixel: returns 256 * (hb + SUM (p-128).wh) + (16384 + SUM p.wl) = SUM p*w + 16384.
// One asm statement per pixel, for two reasons: hipcc only selects the accumulate-in-place v_dot4c for the signed
// product, which costs a v_mov per pixel to seed the accumulator (the three-address v_dot4_i32_i8 does not); and a
// dot result may be read by a different instruction only 3 wait states later (cdna hazard; same-opcode accumulation
// through src2 needs none) -- inside the statement that distance is explicit.
__device__ __forceinline__ int pixel_dots(const uint32_t (&p)[4], const uint32_t* wh, const uint32_t* wl, int hb) {
  int hi, sum;
  uint32_t lo, x;
  asm("v_xor_b32 %3, 0x80808080, %4\n\t"
      "v_dot4_u32_u8 %2, %4, %12, %17\n\t"
      "v_dot4_i32_i8 %1, %3, %8, %16\n\t"
      "v_xor_b32 %3, 0x80808080, %5\n\t"
      "v_dot4_u32_u8 %2, %5, %13, %2\n\t"
      "v_dot4_i32_i8 %1, %3, %9, %1\n\t"
      "v_xor_b32 %3, 0x80808080, %6\n\t"
      "v_dot4_u32_u8 %2, %6, %14, %2\n\t"
      "v_dot4_i32_i8 %1, %3, %10, %1\n\t"
      "v_xor_b32 %3, 0x80808080, %7\n\t"
      "v_dot4_u32_u8 %2, %7, %15, %2\n\t"
      "v_dot4_i32_i8 %1, %3, %11, %1\n\t"
      "s_nop 2\n\t"
      "v_lshl_add_u32 %0, %1, 8, %2"
      : "=v"(sum), "=&v"(hi), "=&v"(lo), "=&v"(x)
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(wh[0]), "v"(wh[1]), "v"(wh[2]), "v"(wh[3]), "v"(wl[0]), "v"(wl[1]),
        "v"(wl[2]), "v"(wl[3]), "v"(hb), "s"(1 << (kCoefBits - 1)));
  return sum;
}
__device__ __forceinline__ int pixel_dots(const uint32_t (&p)[2], const uint32_t* wh, const uint32_t* wl, int hb) {
  int hi, sum;
  uint32_t lo, x;
  asm("v_xor_b32 %3, 0x80808080, %4\n\t"
      "v_dot4_u32_u8 %2, %4, %8, %11\n\t"
      "v_dot4_i32_i8 %1, %3, %6, %10\n\t"
      "v_xor_b32 %3, 0x80808080, %5\n\t"
      "v_dot4_u32_u8 %2, %5, %9, %2\n\t"
      "v_dot4_i32_i8 %1, %3, %7, %1\n\t"
      "s_nop 2\n\t"
      "v_lshl_add_u32 %0, %1, 8, %2"
      : "=v"(sum), "=&v"(hi), "=&v"(lo), "=&v"(x)
      : "v"(p[0]), "v"(p[1]), "v"(wh[0]), "v"(wh[1]), "v"(wl[0]), "v"(wl[1]), "v"(hb), "s"(1 << (kCoefBits - 1)));
  return sum;
}
// any number of windows (Lanczos4: 16): the compiler's own dot products
template <int N>
__device__ __forceinline__ int pixel_dots(const uint32_t (&p)[N], const uint32_t* wh, const uint32_t* wl, int hb) {
  int hi = hb;
  uint32_t lo = 1u << (kCoefBits - 1);
#pragma unroll
  for (int k = 0; k < N; k++) {
    hi = __builtin_amdgcn_sdot4((int)bias128(p[k]), (int)wh[k], hi, false);
    lo = __builtin_amdgcn_udot4(p[k], wl[k], lo, false);
  }
  return (hi << 8) + (int)lo;
}

// dword / byte store at (wave-uniform base) + (32-bit lane offset): the SGPR-base form, so the per-frame
// advance of the base is scalar arithmetic
__device__ __forceinline__ void store_dword(uint8_t* base, uint32_t off, uint32_t v) {
  asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void store_byte(uint8_t* base, uint32_t off, uint32_t v) {
  asm volatile("global_store_byte %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}

// saturate two int16 lanes of a dword to two uint8 (bytes 0 and 1 of the result)
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t two_i16) {
  uint32_t r;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(two_i16));
  return r;
}

// one frame of one tile: gather from the ring slot at byte SLOT of the LDS, write the output pixels.
// GROUP = pixels whose LDS reads are in flight together.
template <int NPX, int KS, int GROUP, int SLOT>
__device__ __forceinline__ void gather_store(const PixelSetup<NPX, KS>& s, const uint8_t* __restrict__ lds,
                                             uint8_t* __restrict__ dbase, uint32_t doff, int dstride, bool dword_store) {
  constexpr int G = GROUP < NPX ? GROUP : NPX;
  constexpr int ROWS = Stencil<KS>::ROWS, WIN = Stencil<KS>::WIN;
  int v[NPX];
  if (KS == 1) {
    // nearest: the byte itself (cv::remap INTER_NEAREST)
#pragma unroll
    for (int p = 0; p < NPX; p++) v[p] = lds[s.addr[p][0] + SLOT];
  } else {
    int t15[NPX];
#pragma unroll
    for (int p0 = 0; p0 < NPX; p0 += G) {
      // Phase 1: the group's LDS reads in flight at once; phase 2: the dot products.
      uint64_t win[G][ROWS];
      uint32_t ext[G][ROWS];  // Lanczos4: third dword of the 8-byte window
#pragma unroll
      for (int p = 0; p < G; p++)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          win[p][r] = *reinterpret_cast<const uint64_t*>(lds + s.addr[p0 + p][r] + SLOT);
          if (WIN == 2) ext[p][r] = *reinterpret_cast<const uint32_t*>(lds + s.addr[p0 + p][r] + SLOT + 8);
        }
      // keep hipcc from sinking the reads next to their uses (it would serialise the round trips)
#pragma unroll
      for (int p = 0; p < G; p++)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          asm volatile("" : "+v"(win[p][r]));
          if (WIN == 2) asm volatile("" : "+v"(ext[p][r]));
        }
#pragma unroll
      for (int p = 0; p < G; p++) {
        const uint32_t sh = s.sh[p0 + p];
        uint32_t px4[ROWS * WIN];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const uint32_t d0 = (uint32_t)win[p][r], d1 = (uint32_t)(win[p][r] >> 32);
          px4[r * WIN] = __builtin_amdgcn_alignbit(d1, d0, sh);  // 4 consecutive source bytes out of the aligned qword
          if (WIN == 2) px4[r * WIN + 1] = __builtin_amdgcn_alignbit(ext[p][r], d1, sh);
        }
        t15[p0 + p] = pixel_dots(px4, s.wh[p0 + p], s.wl[p0 + p], s.hb[p0 + p]) >> kCoefBits;  // fits int16
      }
    }
    if (NPX == 4) {
      // saturate and pack: two int16 per dword -> v_sat_pk_u8_i16
      const uint32_t u01 = sat_pk_u8(__builtin_amdgcn_perm((uint32_t)t15[1 % NPX], (uint32_t)t15[0], 0x05040100u));
      const uint32_t u23 = sat_pk_u8(__builtin_amdgcn_perm((uint32_t)t15[3 % NPX], (uint32_t)t15[2 % NPX], 0x05040100u));
      v[0] = (int)(u01 | (u23 << 16));  // all four pixels, byte k = row k of this lane's column
    } else {
      v[0] = sat_u8(t15[0]);
    }
  }
  if (NPX == 4) {
    uint32_t b;
    if (KS == 1)
      b = (uint32_t)v[0] | ((uint32_t)v[1 % NPX] << 8) | ((uint32_t)v[2 % NPX] << 16) | ((uint32_t)v[3 % NPX] << 24);
    else
      b = (uint32_t)v[0];
    // byte k of b is the pixel of column x = lane % W in row 4*band + k of the tile.
    if (dword_store) {
      // 4x4 byte transpose inside each quad of lanes (DPP quad broadcasts + v_perm), so that
      // lane i of a quad owns row i, columns 4j..4j+3 -> one coalesced dword store per lane
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x00, 0xf, 0xf, true);  // quad_perm 0,0,0,0
      const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x55, 0xf, 0xf, true);  // 1,1,1,1
      const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xaa, 0xf, 0xf, true);  // 2,2,2,2
      const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xff, 0xf, 0xf, true);  // 3,3,3,3
      const uint32_t i = threadIdx.x & 3;
      const uint32_t sel_lo = 0x0c0c0000u | ((4u + i) << 8) | i;           // [b0.byte_i, b1.byte_i, 0, 0]
      const uint32_t sel_hi = 0x00000c0cu | ((4u + i) << 24) | (i << 16);  // [0, 0, b2.byte_i, b3.byte_i]
      const uint32_t w = __builtin_amdgcn_perm(b1, b0, sel_lo) | __builtin_amdgcn_perm(b3, b2, sel_hi);
      store_dword(dbase, doff, w);
    } else {
      // partial tiles / unaligned destinations only
#pragma unroll
      for (int p = 0; p < NPX; p++)
        if (s.live[p]) store_byte(dbase, doff + (uint32_t)(p * dstride), b >> (8 * p));
    }
  } else {
    if (s.live[0]) store_byte(dbase, doff, (uint32_t)v[0]);
  }
}

// Where a lane's output goes (offset inside the frame's plane).  W-wide tiles of 4 px per lane: with dword
// stores lane (x, band) writes row 4*band + (x & 3), columns (x & ~3)..+3 after the quad transpose; with byte
// stores it writes its own column x, rows 4*band + 0..3.
template <int NPX>
__device__ __forceinline__ uint32_t out_pos(const TiledPlane& pl, const TileDesc& t, bool dword_store) {
  const int tid = threadIdx.x;
  int ox, oy;
  if (NPX == 4) {
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
  return (uint32_t)oy * (uint32_t)pl.dstride + (uint32_t)ox;
}

// XCD-aware order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch");
// give every XCD one contiguous range of the execution-ordered tile list so neighbouring tiles --
// whose footprints overlap by the stencil halo -- share an L2.  Bijective for any n.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int xcd = b & 7, k = b >> 3;
  const int q = n >> 3, rem = n & 7;
  return xcd * q + (xcd < rem ? xcd : rem) + k;
}

// ============================ loader / consumer split ==========================================
// Workgroup = 5 waves: wave 4 is the LOADER, waves 0-3 are CONSUMERS (cdna_hip_programming.md 5.6).  The
// loader's instruction stream is only address setup, global_load_lds_dwordx4 and counted vmcnt waits; the
// consumers' frame loop is only barrier -> ds_read -> dot4 -> store.  They meet at ONE s_barrier per frame:
//     loader  : wait until frame i has landed | BARRIER i | refill the slot frame i-1 used
//     consumer:                                 BARRIER i | gather frame i from its slot, store
// The loader's vmcnt stream holds nothing but its own in-order DMA loads, so the count is exact.

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction only takes an immediate):
// a computed jump into a table of 8-byte entries {s_waitcnt vmcnt(k); s_branch out}.
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

// One copy of one frame of one tile, global -> LDS by DMA: nj x (64 lanes x 16 bytes), nj wave-uniform in
// 1..16.  SGPR-base + 32-bit VGPR-offset addressing, so per frame only the scalar base changes; M0 (the LDS
// destination) is written and stepped next to the instruction that uses it.  hipcc does not count these loads
// (cdna_hip_programming.md 5.7): completion is ours to track with wait_vmcnt().
// The 16 {load; step M0 by an SGPR; nop} triples are 16 bytes each (8 + 4 + 4) and laid out back to back; a computed jump enters the
// chain at triple 16-nj (Duff's device), so no per-frame decision tree.  Triple k moves PIECE 15-k: `off[k]`
// must hold the source offset of piece 15-k, M0 starts at the last piece's destination and walks down.
__device__ __forceinline__ void dma_frame_n(int nj, const uint8_t* frame_base, uint32_t lds_dst, const int (&off)[16]) {
  const uint32_t skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(12u + 16u * (uint32_t)(16 - nj)));
  const uint32_t m0_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_dst + (uint32_t)(nj - 1) * 1024u));
#define T360_DMA(k) "global_load_lds_dwordx4 %" #k ", %16\n\ts_sub_u32 m0, m0, %19\n\ts_nop 0\n\t"
  asm volatile(
      "s_mov_b32 m0, %17\n\t"
      "s_getpc_b64 vcc\n\t"
      "s_add_u32 vcc_lo, vcc_lo, %18\n\t"
      "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
      "s_setpc_b64 vcc\n\t"
      T360_DMA(0) T360_DMA(1) T360_DMA(2) T360_DMA(3) T360_DMA(4) T360_DMA(5) T360_DMA(6) T360_DMA(7)
      T360_DMA(8) T360_DMA(9) T360_DMA(10) T360_DMA(11) T360_DMA(12) T360_DMA(13) T360_DMA(14) T360_DMA(15)
      :
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]),
        "v"(off[8]), "v"(off[9]), "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "v"(off[14]), "v"(off[15]),
        "s"(frame_base), "s"(m0_start), "s"(skip), "s"(1024u)
      : "memory", "vcc", "scc");
#undef T360_DMA
}

__device__ __forceinline__ void frame_barrier() {
  // a bare s_barrier: __syncthreads() would add fences whose waits could drain the DMA ring
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int PMAX, int K, bool DUAL>
__device__ __forceinline__ void loader_wave(const TiledPlane& pl, const TileDesc& t, uint32_t lds_base, int f0, int f1) {
  using R = Ring<PMAX, DUAL>;
  const int lane = threadIdx.x & 63;
  const int nj = (int)t.pieces;  // 1..PMAX
  // chunk q = lane + 64*j lives at LDS byte 16*q of each copy; its source (row, 16-byte column) is the plan's
  // (holes repeat a neighbour's chunk): every DMA instruction runs with all 64 lanes.
  int goff[kMaxPieces];  // goff[k] = source offset of piece 15-k (dma_frame_n's order)
#pragma unroll
  for (int k = 0; k < kMaxPieces; k++) {
    const int j = kMaxPieces - 1 - k;
    goff[k] = 0;
    if (j < nj && j < PMAX) {  // wave-uniform
      const uint32_t e = pl.chunks[t.chunks + j * kPieceChunks + lane];
      goff[k] = (int)(e >> 12) * pl.sstride + (int)(e & 4095u) * kStageChunk;
    }
  }
  auto issue = [&](int f, int slot) {
    const uint8_t* base = pl.src + (size_t)f * pl.src_frame_bytes;
    const uint32_t dst = lds_base + (uint32_t)(slot * R::kSlot);
    dma_frame_n(nj, base, dst, goff);
    if (DUAL) dma_frame_n(nj, base, dst + (uint32_t)R::kCopyB, goff);
  };
  const int per_frame = DUAL ? 2 * nj : nj;  // DMA instructions per frame
  const int nf = f1 - f0;
  for (int j = 0; j < K - 1 && j < nf; j++) issue(f0 + j, j);
  int fill = (K - 1) % K;
  for (int i = 0; i < nf; i++) {
    // loads younger than frame i's: frames i+1 .. min(i+K-2, nf-1)
    if (K == 2)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      wait_vmcnt(min(K - 2, nf - 1 - i) * per_frame);
    frame_barrier();  // frame i is visible to the consumers; they have left frame i-1's slot
    if (i + K - 1 < nf) issue(f0 + i + K - 1, fill);
    fill = fill + 1 == K ? 0 : fill + 1;
  }
}

template <int NPX, int KS, int GROUP, int PMAX, int K>
__device__ __forceinline__ void consumer_waves(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t,
                                               const uint8_t* __restrict__ lds, int f0, int f1) {
  using R = Ring<PMAX, KS != 1>;
  PixelSetup<NPX, KS> px;
  load_pixels<NPX, KS, PMAX>(pl, a.wpack, t, px);
  const bool dword_store = NPX == 4 && !(t.flags & kTilePartial) && pl.dst_dword_ok;
  const uint32_t doff = out_pos<NPX>(pl, t, dword_store);
  uint8_t* __restrict__ d = pl.dst + (size_t)f0 * pl.dst_frame_bytes;  // wave-uniform: the store uses SGPR base + VGPR offset
  const int nf = f1 - f0;
  for (int i = 0; i < nf; i += K) {
#define T360_STEP(S)                                                                             \
    if constexpr (S < K) if (i + S < nf) {                                                                 \
      frame_barrier();                                                                           \
      gather_store<NPX, KS, GROUP, S * R::kSlot>(px, lds, d, doff, pl.dstride, dword_store);     \
      d += pl.dst_frame_bytes;                                                                   \
    }
    T360_STEP(0) T360_STEP(1) T360_STEP(2) T360_STEP(3)
#undef T360_STEP
  }
}

// ---- tiles too large to stage: direct gather ---------------------------------------------------
// The few 16x16 tiles around each pole whose source footprint exceeds the staging budget (they span a
// quadrant of longitudes): one pixel per lane, KS*KS independent byte loads in flight.
template <int KS>
__device__ __forceinline__ void direct_tile(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t, int f0, int f1) {
  const int tid = threadIdx.x;
  if (tid >= 256) return;
  const int ox = t.ox + (tid & 15), oy = t.oy + (tid >> 4);
  if (ox >= pl.dw || oy >= pl.dh) return;
  const LutEntry e = pl.lut[(size_t)oy * pl.dw + ox];
  for (int f = f0; f < f1; f++) {
    const int v = sample<KS, false>(pl.src + (size_t)f * pl.src_frame_bytes, pl.sw, pl.sh, pl.sstride, a.wtab, e);
    pl.dst[(size_t)f * pl.dst_frame_bytes + (size_t)oy * pl.dstride + ox] = (uint8_t)v;
  }
}

// Grid: x = padded direct tiles of all planes (started first: they are the slowest per pixel), then the
// staged tiles of all planes; y = frame groups.
template <int KS, int PMAX, int K>
__global__ __launch_bounds__(320) void remap_tiled_kernel(TiledArgs a) {
  constexpr int GROUP = 2;  // LDS reads in groups of 2 px: fewer VGPRs, one more workgroup per CU
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const int f0 = blockIdx.y * a.frames_per_block;
  const int f1 = min(f0 + a.frames_per_block, a.nframes);
  int b = blockIdx.x;
  // pick the plane with scalar selects: indexing a.plane[] with a run-time index would make
  // hipcc copy the whole argument block to scratch
  TiledPlane pl = a.plane[0];
  if (b < a.direct_blocks) {
    if (b >= a.total_direct) return;
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
    direct_tile<KS>(a, pl, pl.tiles[pl.ntiles + b], f0, f1);
    return;
  }
  b = xcd_contiguous(b - a.direct_blocks, a.total_tiles);  // direct_blocks is a multiple of 8: XCD = blockIdx.x % 8 still
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
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= kLoaderWave) {
    loader_wave<PMAX, K, KS != 1>(pl, t, (uint32_t)(uintptr_t)lds, f0, f1);
  } else if (KS == 8 || t.kind == kTileStaged16) {
    consumer_waves<1, KS, GROUP, PMAX, K>(a, pl, t, lds, f0, f1);
  } else {
    consumer_waves<(KS == 8 ? 1 : 4), KS, GROUP, PMAX, K>(a, pl, t, lds, f0, f1);
  }
}

template <int KS, int PMAX, int K>
hipError_t launch_one(const TiledArgs& a, int groups, hipStream_t stream) {
  constexpr int lds_bytes = K * Ring<PMAX, KS != 1>::kSlot;
  if (lds_bytes > 64 * 1024) {
    // per device and cheap: not cached (handles may live on several devices of one process)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_kernel<KS, PMAX, K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((remap_tiled_kernel<KS, PMAX, K>), dim3(a.direct_blocks + a.total_tiles, groups, 1), dim3(320),
                     (size_t)lds_bytes, stream, a);
  return hipGetLastError();
}

template <int KS>
hipError_t launch_ks(const TiledArgs& a, int groups, hipStream_t stream) {
  if (a.max_pieces == 8 && a.ring_slots == 2) return launch_one<KS, 8, 2>(a, groups, stream);
  if (a.max_pieces == 8 && a.ring_slots == 3) return launch_one<KS, 8, 3>(a, groups, stream);
  if (a.max_pieces == 6 && a.ring_slots == 3) return launch_one<KS, 6, 3>(a, groups, stream);
  if (a.max_pieces == 12 && a.ring_slots == 2) return launch_one<KS, 12, 2>(a, groups, stream);
  return hipErrorInvalidValue;
}

}  // namespace

const char* remap_tiled_kernel_name(int ks, int max_pieces, int ring_slots) {
  static thread_local char buf[64];
  snprintf(buf, sizeof(buf), "remap_tiled_kernel<%d, %d, %d>", ks, max_pieces, ring_slots);
  return buf;
}

hipError_t launch_remap_tiled(const TiledArgs& a, hipStream_t stream) {
  if (a.total_tiles + a.total_direct <= 0 || a.nframes <= 0) return hipSuccess;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  switch (a.ks) {
    case 1: return launch_ks<1>(a, groups, stream);
    case 2: return launch_ks<2>(a, groups, stream);
    case 4: return launch_ks<4>(a, groups, stream);
    case 8: return launch_ks<8>(a, groups, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace t360
