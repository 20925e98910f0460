// t360_remap_tiled.hip -- LDS-tiled gather over a batch of frames (the hot kernel).
//
// Same arithmetic as t360_remap.hip (cv::remap, BORDER_WRAP, Q15 weights, (sum + 16384) >> 15,
// SURVEY.md Appendix A.3/A.4); written around the bicubic 4x4 stencil and instantiated for nearest,
// bilinear and Lanczos4 as well.  Organised for MI355X:
//
//   * one workgroup (8 waves; 4 for Lanczos4) owns one OUTPUT tile (128x16 px on 512 lanes, 4 px per lane, lane =
//     column; 64x16 / 32x32 px on 256 lanes where that does not fit the staging budget; 16x16 px with 1 px per lane
//     near the poles; the plan's choice, t360_plan.cpp) and walks `frames_per_block` frames of the batch with it.
//     Everything that depends only on geometry -- the LDS address of each stencil row of each pixel, the 16 Q15
//     weights per pixel, the source addresses of the chunks it stages -- is set up ONCE per tile and kept in
//     registers for all frames: per frame a lane only moves bytes and issues dot products.
//   * per frame the tile's source FOOTPRINT (exactly the 16-byte chunks its stencils touch, packed row after
//     row with a bank-aware placement; planned at init) goes global -> LDS by DMA (global_load_lds_dwordx4, no VGPR
//     round trip) into a ring of K <= 3 slots, K-1 frames ahead of the frame being computed.  Every wave moves its
//     share of the chunks and gathers its share of the pixels; completion is tracked with counted
//     s_waitcnt vmcnt(N) and ONE workgroup barrier per frame.  The +-180 degree seam and BORDER_WRAP across
//     the poles are resolved in the chunk addresses, so the gather itself never wraps.
//   * a 4-byte stencil-row window is read as the two aligned dwords that hold it (+ v_alignbit).  T360_DUAL=1 keeps
//     the variant that writes every chunk twice (copy B four bytes further) and reads ONE ds_read_b64 per window:
//     half the LDS cycles, but twice the LDS per frame in flight, and the kernel waits for memory, not for LDS.
//     (gfx950 serves unaligned ds_read_b32 correctly but at 26 cycles: tools/ubench/lds_unaligned.hip.)
//   * ring slots come in a few compile-time sizes (the tile picks the smallest that holds it) and the frame loop
//     is unrolled over the slots, so a slot is a constant of the unrolled code: the 16-bit immediate offset of the
//     two ds_read_b32 that fetch a window (inline asm + one s_waitcnt lgkmcnt(0) per group of reads: ds_read2_b32 has
//     8-bit offsets, and hipcc would add the slot base to every address register in every frame).
//   * the 4x4 stencil of one output pixel costs 8 LDS dwords + 4 v_alignbit + 4 v_xor and 8 v_dot4: weights are
//     split into a signed high byte and an unsigned low byte (w = 256*wh + wl) and pixels enter the high part as
//     p-128,   SUM p*w = 256 * (SUM (p-128)*wh + 128 * SUM wh) + SUM p*wl,
//     all exact in int32, so results are bit-identical to the integer formulation.
//   * all planes of the frame (Y, U, V) are tiles of ONE launch, including the few tiles around the poles
//     that are gathered straight from global memory (dealt to all eight XCDs); staged workgroups are numbered so
//     that every XCD gets a contiguous range of the execution-ordered (Z-order) tile list: shared halo -> shared L2.
//   * no MFMA: this is a gather, not a contraction.
//   * what bounds it (measured): staging alone 0.245 ms, gathering alone 0.205 ms, both 0.25 ms per 64
//     frames of BASELINE config 2 -- memory throughput (1.31 GB per launch at 5.4 TB/s, 1.53x the algorithmic bytes:
//     neighbouring tiles drift apart in frame number and re-fetch the lines they share) and the per-frame dependency chain
//     of a wave (LDS reads -> dot products -> barrier; ~94 VALU per wave and frame, VALUs 36 % and LDS 41 % busy) are
//     within 20 % of each other.
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"
#include "t360_sample.h"

#include <type_traits>

namespace t360 {

namespace {


// ---- ring geometry (compile time) -------------------------------------------------------------
// A workgroup owns RINGKB KiB of LDS.  A tile of p pieces (1 KiB of staged source each) uses the smallest slot
// class P >= p; its ring then holds K = min(4, ring / slot) frames.  Slot sizes are compile-time constants so
// that the slot base is an immediate of the consumer's ds_read (the frame loop is unrolled over the K slots):
// most tiles stage 4-8 KiB and keep 4 frames in the ring, the few large ones near the poles 2-3.
// frames in the ring of a workgroup, at most (per ring size, for A/B builds).  3 = two frames' DMA in flight.  Round 5
// (profiles/r05_experiments/README.md calls 17, 18): with 4 the DMA-only time of a 64-frame launch does not move (0.2237 -> 0.2239 ms:
// the memory side is not bound by the bytes a workgroup has in flight) and the whole launch is 1.5 % FASTER in the instrumented
// build but 1.5 % SLOWER in the shipped configuration (0.2386 -> 0.2421, three interleaved rounds at +-0.1 %: the frame loop is
// unrolled over the slots); 8-frame steps lose with 4 in both.
#ifndef T360_MAX_SLOTS
#define T360_MAX_SLOTS 3
#endif
#ifndef T360_MAX_SLOTS_SMALL
#define T360_MAX_SLOTS_SMALL 3
#endif
constexpr int max_slots_of(int ringkb) { return ringkb >= 64 ? T360_MAX_SLOTS : T360_MAX_SLOTS_SMALL; }
// T360_DUAL: 1 = every chunk is staged twice (copy B four bytes further) so that each stencil-row window is ONE aligned
// ds_read_b64; 0 = one copy, two aligned ds_read_b32 per window: twice the LDS read cycles, but half the LDS per frame in
// flight.  The gather is bound by bytes in flight (HBM latency x bandwidth), not by LDS cycles: 0 measured faster.
#ifndef T360_DUAL
#define T360_DUAL 0
#endif
constexpr bool dual_copy(int ks) { return T360_DUAL != 0 && ks != 1; }
#ifndef T360_ASMREAD
#define T360_ASMREAD 1
#endif
// T360_B64M: 1 = a stencil-row window is ONE ds_read_b64 at the 4-byte-aligned address below it (addresses that are
// 4 mod 8 included; tools/ubench/lds_b64_align4.hip), 0 = two ds_read_b32
#ifndef T360_B64M
#define T360_B64M 0
#endif
// T360_DMA_FIRST: 1 = the first frames' DMA is issued before the weight gather of the prologue (round 4: 0.2498 ->
// 0.2427 ms per 64-frame launch on the same box; profiles/r04_experiments/README.md call 1)
#ifndef T360_DMA_FIRST
#define T360_DMA_FIRST 1
#endif
template <int P, bool DUAL>
struct Slot {
  // copy B: the same bytes 4 further (odd dwords become 8-byte aligned) and half a bank row (32 dwords) apart,
  // so that the A and B qwords of neighbouring lanes do not meet on the same banks
  static constexpr int kCopyB = P * 1024 + 4 + 128;
  static constexpr int kSlot = DUAL ? 2 * P * 1024 + 128 + 64 : P * 1024 + 64;
};
template <int RINGKB, int P, bool DUAL>
struct Cls {
  static constexpr int kFit = RINGKB * 1024 / Slot<P, DUAL>::kSlot;
  static constexpr int K = kFit > max_slots_of(RINGKB) ? max_slots_of(RINGKB) : kFit;  // < 2: the class does not fit this ring
};
// smallest class that holds `pieces`
#define T360_FOR_CLASS(pieces, F)    \
  if ((pieces) <= 8) { F(8) }        \
  else if ((pieces) <= 12) { F(12) } \
  else if ((pieces) <= 16) { F(16) } \
  else if ((pieces) <= 24) { F(24) } \
  else { F(32) }

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

// What a workgroup fetches from its tile index alone, before (and in parallel with) the tile descriptor.
struct TileFetch {
  uint32_t words[4];  // pixel words of this lane (16x16 tiles: words[0])
  uint32_t chunk[4];  // chunk entries of this lane in pieces wave, wave + WAVES, wave + 2 WAVES, wave + 3 WAVES
  uint32_t rowdw;     // dword `lane` of the row table = row_base of box rows 2*lane and 2*lane + 1
  uint32_t origin;    // scatter plans: ox | oy << 16 of this lane's 4x4 block (kTileScatter)
};

template <int KS, int WAVES>
__device__ __forceinline__ TileFetch fetch_tile(const TiledPlane& pl, int tile, int max_pieces, bool scatter) {
  TileFetch f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (KS == 8) {
    f.words[0] = tid < 256 ? pl.tlut[(size_t)tile * tile_words(KS, WAVES) + tid] : kWordDead;
    f.words[1] = f.words[2] = f.words[3] = kWordDead;
  } else {
    const uint4 v = reinterpret_cast<const uint4*>(pl.tlut + (size_t)tile * tile_words(KS, WAVES))[tid];
    f.words[0] = v.x; f.words[1] = v.y; f.words[2] = v.z; f.words[3] = v.w;
  }
  const uint32_t* __restrict__ tc = pl.chunks + (size_t)tile * tile_chunk_dwords(max_pieces, scatter);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int piece = wave + WAVES * j;
    f.chunk[j] = piece < max_pieces ? tc[piece * kPieceChunks + lane] : 0u;
  }
  f.rowdw = tc[max_pieces * kPieceChunks + lane];
  // (lanes 4q..4q+3 of band b hold block b*32 + q; only 8-wave workgroups run scatter plans)
  f.origin = scatter && WAVES == 8 ? tc[max_pieces * kPieceChunks + 64 + (tid >> 7) * 32 + ((tid & 127) >> 2)] : 0u;
  return f;
}

// row_base of box row r out of the wave-distributed row table (lane r/2 holds rows r & ~1 and r | 1)
__device__ __forceinline__ int row_base_of(uint32_t rowdw, int r) {
  const int v = __builtin_amdgcn_ds_bpermute((r >> 1) << 2, (int)rowdw);
  return (r & 1) ? (v >> 16) : (int)(int16_t)v;
}

template <int NPX, int KS, int P>
__device__ __forceinline__ void load_pixels(const TileFetch& tf, const uint32_t* __restrict__ wpack, PixelSetup<NPX, KS>& s) {
  constexpr int NW = Stencil<KS>::NW;
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    const uint32_t e = tf.words[p];
    s.live[p] = (e >> 31) == 0;
    const int x = e & 2047, row = s.live[p] ? (int)((e >> kWordRowShift) & 255) : 0;
    const int frac = (e >> kWordFracShift) & 1023;
    s.sh[p] = (uint32_t)(x & 3) * 8u;
#pragma unroll
    for (int k = 0; k < KS; k++) {
      const int off = row_base_of(tf.rowdw, (row + k) & (kBoxMaxRows - 1)) * kStageChunk + x;
      // KS == 1 reads the byte itself from copy A; otherwise the aligned qword of copy A or B that holds the window
      s.addr[p][k] = !s.live[p]    ? 0u
                     : KS == 1      ? (uint32_t)off
                     : dual_copy(KS) ? (uint32_t)((off & ~3) + ((off & 4) ? Slot<P, true>::kCopyB : 0))
                                     : (uint32_t)(off & ~3);
    }
    s.hb[p] = 0;
    if (NW > 0) {
      // [NW high dwords][NW low dwords], 16-byte aligned: 2*NW/4 vector loads
      const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(wpack + (size_t)frac * Stencil<KS>::PACK);
      uint32_t w[2 * (NW > 0 ? NW : 2)];
#pragma unroll
      for (int k = 0; k < (2 * NW) / 4; k++) {
        const uint4 v = wp[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
      }
      // 128 * SUM(signed high bytes): the bias of the signed half of the products (pixels enter it as p - 128)
      int sum_hi = 0;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        s.wh[p][k] = w[k];
        s.wl[p][k] = w[NW + k];
        sum_hi = __builtin_amdgcn_sdot4((int)w[k], 0x01010101, sum_hi, false);
      }
      s.hb[p] = 128 * sum_hi;
    }
  }
}

// Make hipcc wait for its own (counted) loads HERE: every loaded value passes through an empty asm, so the
// compiler-inserted s_waitcnt lands before it and not in front of the first use inside the frame loop, where it
// would also drain the DMA ring.
template <int NPX, int KS>
__device__ __forceinline__ void pin_pixels(PixelSetup<NPX, KS>& s) {
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    asm volatile("" : "+v"(s.hb[p]), "+v"(s.sh[p]));
#pragma unroll
    for (int r = 0; r < KS; r++) asm volatile("" : "+v"(s.addr[p][r]));
#pragma unroll
    for (int r = 0; r < Stencil<KS>::NW; r++) asm volatile("" : "+v"(s.wh[p][r]), "+v"(s.wl[p][r]));
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
__device__ __forceinline__ uint8_t* uniform_ptr(uint8_t* p) {  // the value IS wave-uniform; make hipcc see it
  const uint64_t v = (uint64_t)(uintptr_t)p;
  return reinterpret_cast<uint8_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
#ifdef T360_INSTRUMENT
#define T360_UNIFORM(p) uniform_ptr(p)  // the divergent trace branches of this build hide the uniformity from hipcc
#else
#define T360_UNIFORM(p) (p)
#endif
// NT: a streaming store -- output lines are written once and never read, and every line they do not claim in the L2
// is a source line a neighbouring workgroup may still find there (-1 % on the bicubic kernel; the nearest-neighbour
// kernel, which shares nothing, measured 2 % slower with it; `nt` LOADS make the staging a third slower: they give up
// exactly the lines the neighbours share.  tools/experiments_r03/gpu_call27.sh, gpu_call28.sh)
#ifndef T360_STORE_NT
#define T360_STORE_NT 1
#endif
template <bool NT>
__device__ __forceinline__ void store_dword(uint8_t* base, uint32_t off, uint32_t v) {
  if (NT)
    asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(off), "v"(v), "s"(base) : "memory");
  else
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

// one frame of one tile: gather from the ring slot at byte SLOT of the LDS; returns what the lane stores (the
// store itself is deferred by one frame, see tile_waves()).  GROUP = pixels whose LDS reads are in flight together.
template <int NPX, int KS, int GROUP, int SLOT>
__device__ __forceinline__ uint32_t gather(const PixelSetup<NPX, KS>& s, const uint8_t* __restrict__ lds, bool dword_store) {
  constexpr int G = GROUP < NPX ? GROUP : NPX;
  constexpr int ROWS = Stencil<KS>::ROWS, WIN = Stencil<KS>::WIN;
  int v[NPX];
  if (KS == 1) {
    // nearest: the byte itself (cv::remap INTER_NEAREST)
#pragma unroll
    for (int p = 0; p < NPX; p++) v[p] = lds[s.addr[p][0] + SLOT];
  } else if (KS == 8) {
    // Lanczos4: 8 rows x 8 taps, every window consumed as soon as it is assembled: the 64 weights already take 32
    // registers per pixel.  (Two halves of 12 dwords with one wait each: 3.63 ms per 64 frames of BASELINE config 4;
    // counted waits per row: 3.54; all 24 dwords up front: 3.50.)
    static_assert(KS != 8 || NPX == 1, "Lanczos4 tiles hold one pixel per lane");
    const uint32_t sh = s.sh[0];
    int hi = s.hb[0];
    uint32_t lo = 1u << (kCoefBits - 1);
    {
      // all 24 dwords of the stencil up front, rows consumed as they arrive (the LGKM counter has 4 bits: rows 0-2 wait for
      // "at most 15 outstanding", which 24 - 9 = 15 reads behind row 2 make exact for row 2 and generous for rows 0, 1)
      uint32_t d[8][3];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const uint32_t la = (uint32_t)(uintptr_t)lds + s.addr[0][r];
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[r][k]) : "v"(la), "n"(SLOT + 4 * k));
        }
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (r == 0) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
        if (r == 3) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        if (r == 4) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
        if (r == 5) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        if (r == 6) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        if (r == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 3; k++) asm volatile("" : "+v"(d[r][k]));
#pragma unroll
        for (int w = 0; w < 2; w++) {
          const uint32_t px4 = __builtin_amdgcn_alignbit(d[r][w + 1], d[r][w], sh);
          hi = __builtin_amdgcn_sdot4((int)bias128(px4), (int)s.wh[0][r * 2 + w], hi, false);
          lo = __builtin_amdgcn_udot4(px4, s.wl[0][r * 2 + w], lo, false);
        }
      }
    }
    v[0] = sat_u8(((hi << 8) + (int)lo) >> kCoefBits);
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
          if (dual_copy(KS)) {
            win[p][r] = *reinterpret_cast<const uint64_t*>(lds + s.addr[p0 + p][r] + SLOT);
          } else if (T360_ASMREAD && WIN == 1) {
            // the ring slot as the IMMEDIATE offset of two ds_read_b32 (16 bits; ds_read2_b32 has 8-bit offsets, so for
            // slots 1.. hipcc adds the slot base to every address register first: 16 VALU per 4 pixels and frame)
            const uint32_t la = (uint32_t)(uintptr_t)lds + s.addr[p0 + p][r];  // loop invariant: hoisted out of the frame loop
#if T360_B64M
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(win[p][r]) : "v"(la), "n"(SLOT));
#else
            uint32_t d0, d1;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d0) : "v"(la), "n"(SLOT));
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d1) : "v"(la), "n"(SLOT + 4));
            win[p][r] = (uint64_t)d0 | ((uint64_t)d1 << 32);
#endif
          } else {
            // two aligned dwords (the slot base does not fit ds_read2_b32's 8-bit offsets: two ds_read_b32)
            const uint32_t d0 = *reinterpret_cast<const uint32_t*>(lds + s.addr[p0 + p][r] + SLOT);
            const uint32_t d1 = *reinterpret_cast<const uint32_t*>(lds + s.addr[p0 + p][r] + SLOT + 4);
            win[p][r] = (uint64_t)d0 | ((uint64_t)d1 << 32);
          }
          if (WIN == 2) ext[p][r] = *reinterpret_cast<const uint32_t*>(lds + s.addr[p0 + p][r] + SLOT + 8);
        }
      if (T360_ASMREAD && WIN == 1 && !dual_copy(KS)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // hipcc does not count asm loads
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
    if (!dword_store) return b;  // partial tiles / unaligned destinations: four byte stores
    // 4x4 byte transpose inside each quad of lanes (DPP quad broadcasts + v_perm), so that
    // lane i of a quad owns row i, columns 4j..4j+3 -> one coalesced dword store per lane
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x00, 0xf, 0xf, true);  // quad_perm 0,0,0,0
    const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x55, 0xf, 0xf, true);  // 1,1,1,1
    const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xaa, 0xf, 0xf, true);  // 2,2,2,2
    const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xff, 0xf, 0xf, true);  // 3,3,3,3
    const uint32_t i = threadIdx.x & 3;
    const uint32_t sel_lo = 0x0c0c0000u | ((4u + i) << 8) | i;           // [b0.byte_i, b1.byte_i, 0, 0]
    const uint32_t sel_hi = 0x00000c0cu | ((4u + i) << 24) | (i << 16);  // [0, 0, b2.byte_i, b3.byte_i]
    return __builtin_amdgcn_perm(b1, b0, sel_lo) | __builtin_amdgcn_perm(b3, b2, sel_hi);
  }
  return (uint32_t)v[0];
}

// the deferred store of one frame's value
template <int NPX, int KS>
__device__ __forceinline__ void emit(const PixelSetup<NPX, KS>& s, uint32_t val, uint8_t* __restrict__ dbase, uint32_t doff,
                                     int dstride, bool dword_store, bool all_live) {
  dbase = T360_UNIFORM(dbase);
  if (NPX == 4) {
    if (dword_store) {
      // After the quad transpose a lane stores pixels that came from four lanes.  Rectangular tiles take the dword path
      // only when they are not partial (all_live, wave-uniform: every quad is whole); a scatter tile's missing blocks
      // have dead pixel words on all four lanes of the quad and store nothing (tests/plan_sim checks that every quad of
      // a scatter plan is entirely live or entirely dead).
      if (all_live || s.live[0]) store_dword<(KS != 1) && T360_STORE_NT>(dbase, doff, val);
    } else {
#pragma unroll
      for (int p = 0; p < NPX; p++)
        if (s.live[p]) store_byte(dbase, doff + (uint32_t)(p * dstride), val >> (8 * p));
    }
  } else {
    if (s.live[0]) store_byte(dbase, doff, val);
  }
}

// Where a lane's output goes (offset inside the frame's plane).  W-wide tiles of 4 px per lane: with dword
// stores lane (x, band) writes row 4*band + (x & 3), columns (x & ~3)..+3 after the quad transpose; with byte
// stores it writes its own column x, rows 4*band + 0..3.
template <int NPX>
__device__ __forceinline__ uint32_t out_pos(const TiledPlane& pl, const TileDesc& t, bool dword_store, uint32_t origin) {
  const int tid = threadIdx.x;
  int ox, oy;
  if (NPX == 4 && t.kind == kTileScatter) {
    // this lane's 4x4 block sits at `origin`; the lane is its column tid & 3 (byte stores: rows 0..3 of that column) or,
    // after the quad transpose, its row tid & 3 (one dword)
    ox = (int)(origin & 0xffffu) + (dword_store ? 0 : (tid & 3));
    oy = (int)(origin >> 16) + (dword_store ? (tid & 3) : 0);
  } else if (NPX == 4) {
    const int logw = t.kind == kTileWide256 ? 8 : (t.kind == kTileStrip128 || t.kind == kTileWide128 || t.kind == kTileScatter) ? 7 : (t.kind == kTileWide64 ? 6 : 5);
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

// ============================ staging ==========================================================
// Every wave of the workgroup moves a quarter of the tile's staged bytes (pieces w, w+4, ...) and gathers a
// quarter of its pixels; the waves meet at ONE s_barrier per frame.  A workgroup of exactly 4 waves puts one
// wave on each SIMD of the CU wherever the dispatcher starts, so workgroups pack the CU without fragmentation
// (a 4+1 loader/consumer split measured 2.2 resident workgroups per CU where 4 fit: tools/ubench/residency_bench).

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

// A wave's share of one copy of one frame of one tile, global -> LDS by DMA: nj x (64 lanes x 16 bytes), nj
// wave-uniform in 1..4; the wave's pieces are lds_step bytes apart in LDS (wave w of W moves pieces w, w+W, w+2W, ...).
// SGPR-base + 32-bit VGPR-offset addressing, so per frame only the scalar base changes; M0 (the LDS destination)
// is written and stepped next to the instruction that uses it.  hipcc does not count these loads
// (cdna_hip_programming.md 5.7): completion is ours to track with wait_vmcnt().
// The 4 {load; step M0 by an SGPR; nop} triples are 16 bytes each (8 + 4 + 4) and laid out back to back; a computed
// jump enters the chain at triple 4-nj (Duff's device), so no per-frame decision tree.  Triple k moves the
// wave's piece 3-k: `off[k]` must hold its source offset; M0 starts at the last piece's destination and walks down.
__device__ __forceinline__ void dma_frame_4(int nj, const uint8_t* frame_base, uint32_t lds_dst, uint32_t lds_step,
                                            const int (&off)[4]) {
  const uint32_t skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)(12u + 16u * (uint32_t)(4 - nj)));
  const uint32_t m0_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_dst + (uint32_t)(nj - 1) * lds_step));
#ifndef T360_DMA_POLICY
#define T360_DMA_POLICY ""  // cache policy bits of the staging loads ("" | " sc1" | " sc0 sc1" | " nt"): A/B builds only
#endif
#define T360_DMA(k) "global_load_lds_dwordx4 %" #k ", %4" T360_DMA_POLICY "\n\ts_sub_u32 m0, m0, %7\n\ts_nop 0\n\t"
  asm volatile(
      "s_mov_b32 m0, %5\n\t"
      "s_getpc_b64 vcc\n\t"
      "s_add_u32 vcc_lo, vcc_lo, %6\n\t"
      "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
      "s_setpc_b64 vcc\n\t"
      T360_DMA(0) T360_DMA(1) T360_DMA(2) T360_DMA(3)
      :
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(frame_base), "s"(m0_start), "s"(skip), "s"(lds_step)
      : "memory", "vcc", "scc");
#undef T360_DMA
}

__device__ __forceinline__ void frame_barrier() {
  // a bare s_barrier: __syncthreads() would add fences whose waits could drain the DMA ring
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef T360_INSTRUMENT
#define T360_DBG(a, bit) (((a).debug >> (bit)) & 1)
// per-workgroup phase timestamps of the instrumented build (T360_TRACE=file; tools/trace_stats.py)
__device__ __forceinline__ void trace_mark(const TiledArgs& a, int slot) {
  if (a.trace && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    a.trace[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
  }
}
#define T360_MARK(a, slot) trace_mark(a, slot)
#else
#define T360_DBG(a, bit) 0
#define T360_MARK(a, slot)
#endif

// The four waves of a workgroup, one tile, frames f0..f1-1.  P = slot class, K = frames in the ring.
// Timeline of a wave at frame i (slot i % K):
//     wait until MY pieces of frame i have landed | BARRIER i | store frame i-1's pixels | refill the slot frame i-1
//     used with my pieces of frame i+K-1 | gather frame i
// The store is deferred past the barrier so that the wave's vmcnt stream, which now holds its DMA loads AND its
// stores, has no store younger than the loads it is about to wait for: loads complete in order among themselves, so
// "at most D operations outstanding", D = my loads younger than frame i's, implies frame i's pieces are done whatever
// the (older) stores do.
template <int NPX, int KS, int GROUP, int P, int K, int WAVES>
__device__ __forceinline__ void tile_waves(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t, const TileFetch& tf,
                                           const uint8_t* __restrict__ lds, int f0, int f1) {
  constexpr bool DUAL = dual_copy(KS);
  using R = Slot<P, DUAL>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  (void)lane;
  const int mine = ((int)t.pieces - wave + WAVES - 1) / WAVES;  // my pieces: wave, wave + WAVES, ... (0..4 of them)
  // does this wave hold pixels?  (a 256-lane tile in a workgroup of 8 waves: waves 4..7 only move bytes)
  const bool has_px = WAVES == 4 || t.kind == kTileWide128 || t.kind == kTileWide256 || t.kind == kTileScatter || wave < 4;
  // chunk q = lane + 64*piece lives at LDS byte 16*q of each copy; its source (row, 16-byte column) is the plan's
  // (holes repeat a neighbour's chunk): every DMA instruction runs with all 64 lanes.
  int goff[4];  // goff[k] = source offset of my piece 3-k (dma_frame_4's order)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int j = 3 - k;
    const uint32_t e = tf.chunk[j];
    goff[k] = (j < mine && WAVES * j < P) ? (int)(e >> 12) * pl.sstride + (int)(e & 4095u) * kStageChunk : 0;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds + (uint32_t)wave * 1024u;
  auto issue = [&](int f, int slot_bytes) {
    if (mine <= 0) return;
    const uint8_t* base = T360_UNIFORM(const_cast<uint8_t*>(pl.src + (size_t)(T360_DBG(a, 6) ? 0 : f) * pl.src_frame_bytes));
    dma_frame_4(mine, base, lds_base + (uint32_t)slot_bytes, WAVES * 1024u, goff);
    if (DUAL && !T360_DBG(a, 5)) dma_frame_4(mine, base, lds_base + (uint32_t)(slot_bytes + R::kCopyB), WAVES * 1024u, goff);
  };
  const int per_frame = DUAL ? 2 * mine : mine;  // my DMA instructions per frame
  const int nf = f1 - f0;
  T360_MARK(a, 1);  // tile tables here
  // weights first (they depend on the pixel words only), then the prologue DMA, then wait for both
  PixelSetup<NPX, KS> px;
#if T360_DMA_FIRST
  // the first frames' DMA goes out BEFORE the weight gather: the vector memory pipe is in order, and the 12 table loads
  // per lane of load_pixels() keep its front end busy for ~3 us (64 distinct lines per instruction) -- queued behind
  // them, the DMA requests left only then and their ~2 us of HBM latency came on top
#pragma unroll
  for (int j = 0; j < K - 1; j++)
    if (j < nf) issue(f0 + j, j * R::kSlot);
  load_pixels<NPX, KS, P>(tf, a.wpack, px);
#else
  load_pixels<NPX, KS, P>(tf, a.wpack, px);
#pragma unroll
  for (int j = 0; j < K - 1; j++)
    if (j < nf) issue(f0 + j, j * R::kSlot);
#endif
  pin_pixels<NPX, KS>(px);
  T360_MARK(a, 2);  // pixel setup here (and, with tracing on, the prologue DMA landed)
  const bool dword_store = NPX == 4 && (!(t.flags & kTilePartial) || t.kind == kTileScatter) && pl.dst_dword_ok;
  const bool all_live = t.kind != kTileScatter;  // (of the dword path: partial rectangular tiles store bytes)
  const uint32_t doff = out_pos<NPX>(pl, t, dword_store, tf.origin);
  uint8_t* __restrict__ d = uniform_ptr(pl.dst + (size_t)f0 * pl.dst_frame_bytes);  // the store uses SGPR base + VGPR offset
  uint32_t pending = 0;
#ifdef T360_INSTRUMENT
  // phase profile (T360_PHASES=file): shader-clock cycles this wave spent in each part of the frame loop, summed over
  // the frames: [0] waiting for its DMA pieces, [1] at the barrier, [2] store + DMA issue, [3] gather, [4] the rest
  unsigned long long ph_acc[5] = {0, 0, 0, 0, 0}, ph_last = a.phases ? __builtin_readcyclecounter() : 0;
#define T360_PHASE(k)                                          \
  if (a.phases) {                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ph_acc[k] += now_ - ph_last;                               \
    ph_last = now_;                                            \
  }
#else
#define T360_PHASE(k)
#endif
  for (int i = 0; i < nf; i += K) {
#define T360_STEP(S)                                                                                       \
    if constexpr (S < K) if (i + S < nf) {                                                                 \
      T360_PHASE(4);                                                                                       \
      wait_vmcnt(min(K - 2, nf - 1 - (i + S)) * per_frame); /* my loads younger than this frame's */       \
      T360_PHASE(0);                                                                                       \
      if (!T360_DBG(a, 7)) frame_barrier(); /* the frame is complete in LDS; everyone has left the previous frame's slot */ \
      T360_PHASE(1);                                                                                       \
      if (T360_DBG(a, 10)) asm volatile("s_setprio 3");                                                    \
      if (i + S > 0) {                                                                                     \
        if (has_px && !T360_DBG(a, 8)) emit<NPX, KS>(px, pending, d, doff, pl.dstride, dword_store, all_live);       \
        d += pl.dst_frame_bytes;                                                                           \
      }                                                                                                    \
      if (T360_DBG(a, 9)) asm volatile("s_setprio 3");                                                     \
      if (i + S + K - 1 < nf && !T360_DBG(a, 1)) issue(f0 + i + S + K - 1, ((S + K - 1) % K) * R::kSlot);  \
      if (T360_DBG(a, 9) || T360_DBG(a, 10)) asm volatile("s_setprio 0");                                  \
      T360_PHASE(2);                                                                                       \
      if (has_px && !T360_DBG(a, 0)) pending = gather<NPX, KS, GROUP, S * R::kSlot>(px, lds, dword_store); \
      asm volatile("" : "+v"(pending));                                                                    \
      T360_PHASE(3);                                                                                       \
      if (i + S == 0) T360_MARK(a, 3);                                                                     \
      if (i + S == 1) T360_MARK(a, 4);                                                                     \
    }
    T360_STEP(0) T360_STEP(1) T360_STEP(2) T360_STEP(3)
#undef T360_STEP
  }
  if (nf > 0 && has_px) emit<NPX, KS>(px, pending, d, doff, pl.dstride, dword_store, all_live);
#ifdef T360_INSTRUMENT
  if (a.phases && (wave == 0 || wave == WAVES - 1) && lane == 0) {
    unsigned long long* o = a.phases + (size_t)blockIdx.x * 16 + (wave == 0 ? 0 : 8);
#pragma unroll
    for (int k = 0; k < 5; k++) o[k] = ph_acc[k];
    o[5] = (unsigned long long)nf;
    o[6] = ((unsigned long long)(unsigned)t.kind << 32) | (unsigned)t.pieces;
  }
#endif
#undef T360_PHASE
  T360_MARK(a, 5);
}

// ---- tiles too large to stage: direct gather ---------------------------------------------------
// The few 16x16 tiles around each pole whose source footprint exceeds the staging budget (they span a
// quadrant of longitudes): one pixel per lane, KS*KS independent byte loads in flight.
template <int KS>
__device__ __forceinline__ void direct_tile(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t, int f0, int f1) {
  // 16x16 pixels on lanes 0..255; in a workgroup of 8 waves lanes 256..511 take every other group of frames
  const int tid = threadIdx.x & 255, part = threadIdx.x >> 8, nparts = (int)blockDim.x >> 8;
  const int ox = t.ox + (tid & 15), oy = t.oy + (tid >> 4);
  if (ox >= pl.dw || oy >= pl.dh) return;
  const LutEntry e = pl.lut[(size_t)oy * pl.dw + ox];
  uint8_t* __restrict__ d = pl.dst + (size_t)oy * pl.dstride + ox;
  if constexpr (KS == 2 || KS == 4) {
    // Row offsets and weights are the same for every frame: set up once.  A stencil row is ONE (unaligned) dword
    // load unless it crosses the +-180 degree seam, and NF frames' loads are in flight together.  These few tiles
    // hold a workgroup slot (and its LDS) for as long as they take, so they are worth making quick.
    constexpr int H = KS / 2 - 1, KK = KS * KS;
    constexpr int NF = 4;  // frames in flight per lane
    int roff[KS], w[KK];
    const int16_t* __restrict__ wt = a.wtab + (size_t)e.frac * KK;
    const int x0 = (int)e.ix - H;
    const bool contiguous = x0 >= 0 && x0 + 4 <= pl.sw;  // 4 bytes are read per row (bilinear uses the first 2)
#pragma unroll
    for (int r = 0; r < KS; r++) {
      roff[r] = wrap_coord((int)e.iy - H + r, pl.sh) * pl.sstride;
#pragma unroll
      for (int c = 0; c < KS; c++) w[r * KS + c] = wt[r * KS + c];
    }
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; c++) xo[c] = wrap_coord(x0 + c, pl.sw);
    // the frame loop once per case (not a branch per row: the loads of an iteration must all be in flight together)
    auto frames = [&](auto contig) {
      constexpr bool CONTIG = decltype(contig)::value;
      for (int f = f0 + part * NF; f < f1; f += NF * nparts) {
        uint32_t v[NF][KS];
#pragma unroll
        for (int k = 0; k < NF; k++) {
          // frames past the end repeat the last one (their loads hit the same lines; nothing is stored for them)
          const uint8_t* __restrict__ sf = pl.src + (size_t)min(f + k, f1 - 1) * pl.src_frame_bytes;
#pragma unroll
          for (int r = 0; r < KS; r++) {
            if (CONTIG) {
              __builtin_memcpy(&v[k][r], sf + roff[r] + x0, 4);
            } else {
              v[k][r] = 0;
#pragma unroll
              for (int c = 0; c < 4; c++) v[k][r] |= (uint32_t)sf[roff[r] + xo[c]] << (8 * c);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < NF; k++) {
          int sum = 1 << (kCoefBits - 1);
#pragma unroll
          for (int r = 0; r < KS; r++)
#pragma unroll
            for (int c = 0; c < KS; c++) sum += (int)((v[k][r] >> (8 * c)) & 255u) * w[r * KS + c];
          if (f + k < f1) d[(size_t)(f + k) * pl.dst_frame_bytes] = (uint8_t)sat_u8(sum >> kCoefBits);
        }
      }
    };
    if (contiguous)
      frames(std::true_type{});
    else
      frames(std::false_type{});
  } else if constexpr (KS == 8) {
    // Lanczos4: the 64 taps one stencil row at a time (8 byte loads in flight), wrapped columns set up once per pixel.  The
    // generic sample<8>() unrolls all 64 loads and weights and made the whole kernel spill 18 VGPRs to scratch
    // (`scratch=76` in the round-5 profiles); same sum, same rounding.
    constexpr int H = KS / 2 - 1;
    const int16_t* __restrict__ wt = a.wtab + (size_t)e.frac * (KS * KS);
    int xo[KS];
#pragma unroll
    for (int c = 0; c < KS; c++) xo[c] = wrap_coord((int)e.ix - H + c, pl.sw);
    for (int f = f0 + part; f < f1; f += nparts) {
      const uint8_t* __restrict__ sf = pl.src + (size_t)f * pl.src_frame_bytes;
      int sum = 1 << (kCoefBits - 1);
#pragma unroll 1
      for (int r = 0; r < KS; r++) {
        const uint8_t* __restrict__ S = sf + (size_t)wrap_coord((int)e.iy - H + r, pl.sh) * pl.sstride;
#pragma unroll
        for (int c = 0; c < KS; c++) sum += (int)S[xo[c]] * (int)wt[r * KS + c];
      }
      d[(size_t)f * pl.dst_frame_bytes] = (uint8_t)sat_u8(sum >> kCoefBits);
    }
  } else {
    for (int f = f0 + part; f < f1; f += nparts) {
      const int v = sample<KS, false>(pl.src + (size_t)f * pl.src_frame_bytes, pl.sw, pl.sh, pl.sstride, a.wtab, e);
      d[(size_t)f * pl.dst_frame_bytes] = (uint8_t)v;
    }
  }
}

// Grid (1-D): the direct tiles' work items first (they are the slowest per pixel), then the staged tiles'.
template <int KS, int RINGKB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void remap_tiled_kernel(TiledArgs a) {
#ifndef T360_GROUP
#define T360_GROUP 4
#endif
  constexpr int GROUP = T360_GROUP;  // pixels whose LDS reads are in flight together (4: one LDS round trip per frame)
  constexpr bool DUAL = dual_copy(KS);
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  // Work items = (tile, frame group), numbered with the frame group fastest: the groups of one tile start within
  // microseconds of each other on the same XCD, so the tile's tables come from HBM once and from L2 afterwards.
  int id = blockIdx.x, b, g, f0, f1;
  // pick the plane with scalar selects: indexing a.plane[] with a run-time index would make
  // hipcc copy the whole argument block to scratch
  TiledPlane pl = a.plane[0];
  if (id < a.direct_blocks) {
    // direct tiles, one work item per (tile, frame group); consecutive ids run on different XCDs, so the pole tiles --
    // every lane of which pulls whole 128-byte lines of the polar source rows through its XCD's L2 for 4 bytes each --
    // are spread over all eight L2s (concentrated on one XCD per pole they slowed that XCD's staged tiles enough to
    // set the launch's critical path)
    int t_idx = id / a.groups;
    g = id - t_idx * a.groups;
    if (t_idx >= a.total_direct || T360_DBG(a, 2)) return;
    if (a.nplanes > 1 && t_idx >= pl.ndirect) {
      t_idx -= pl.ndirect;
      pl = a.plane[1];
      if (a.nplanes > 2 && t_idx >= pl.ndirect) {
        t_idx -= pl.ndirect;
        pl = a.plane[2];
        if (a.nplanes > 3 && t_idx >= pl.ndirect) {
          t_idx -= pl.ndirect;
          pl = a.plane[3];
        }
      }
    }
    f0 = g * a.frames_per_block;
    direct_tile<KS>(a, pl, pl.tiles[pl.ntiles + t_idx], f0, min(f0 + a.frames_per_block, a.nframes));
    return;
  }
  {
    // XCD-aware order: workgroup id runs on XCD id % 8 (MI355X_MICROARCH.md "Workgroup dispatch"; direct_blocks is a
    // multiple of 8); every XCD owns one contiguous range of the execution-ordered tile list, so neighbouring tiles --
    // whose footprints overlap by the stencil halo -- share an L2
    id -= a.direct_blocks;
    const int xcd = id & 7, k = id >> 3;
#ifdef T360_INSTRUMENT
    if (a.k_hi > 0 && (k < a.k_lo || k >= a.k_hi)) return;  // generation experiments: a slice of every XCD's list only
#endif
    const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
    const int len = q + (xcd < rem ? 1 : 0), start = xcd * q + (xcd < rem ? xcd : rem);
    // The last tiles of every XCD's range walk the batch in shorter runs (tail_frames instead of frames_per_block):
    // the workgroups that finish the launch are then short ones, and the machine drains in ~tail_frames frame times
    // instead of frames_per_block (a quarter of the grid costs a few extra start-ups, the tail shrinks 4x).
    const int len_tail = (len * a.tail_percent) / 100, len_head = len - len_tail;
    int fpb;
    if (k < len_head * a.groups) {
      const int t_local = k / a.groups;
      b = start + t_local;
      g = k - t_local * a.groups;
      fpb = a.frames_per_block;
    } else {
      const int k2 = k - len_head * a.groups;
      const int t_local = k2 / a.tail_groups;
      if (t_local >= len_tail) return;
      b = start + len_head + t_local;
      g = k2 - t_local * a.tail_groups;
      fpb = a.tail_frames;
    }
    f0 = g * fpb;
    f1 = min(f0 + fpb, a.nframes);
  }
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
  const TileFetch tf = fetch_tile<KS, WAVES>(pl, b, a.max_pieces, pl.scatter != 0);  // independent of the descriptor: all in flight together
  const TileDesc t = pl.tiles[b];
#ifdef T360_INSTRUMENT
  if (a.trace && threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const size_t wg = blockIdx.x;
    a.trace[wg * 8 + 7] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    a.trace[wg * 8 + 6] = ((unsigned long long)(unsigned)t.kind << 32) | (unsigned)t.pieces;
  }
  T360_MARK(a, 0);  // tile descriptor here
#endif
  if (T360_DBG(a, 3) && t.kind == kTileStaged16) return;
  if (T360_DBG(a, 4) && t.kind != kTileStaged16) return;
  const int pieces = (int)t.pieces;
  if (KS == 8 || t.kind == kTileStaged16) {
#define T360_TILE1(P) \
    if constexpr (Cls<RINGKB, P, DUAL>::K >= 2) tile_waves<1, KS, GROUP, P, Cls<RINGKB, P, DUAL>::K, WAVES>(a, pl, t, tf, lds, f0, f1);
    T360_FOR_CLASS(pieces, T360_TILE1)
#undef T360_TILE1
  } else {
#define T360_TILE4(P)                                                        \
    if constexpr (Cls<RINGKB, P, DUAL>::K >= 2)                               \
      tile_waves<(KS == 8 ? 1 : 4), KS, GROUP, P, Cls<RINGKB, P, DUAL>::K, WAVES>(a, pl, t, tf, lds, f0, f1);
    T360_FOR_CLASS(pieces, T360_TILE4)
#undef T360_TILE4
  }
}

// largest tile (in pieces) the ring of RINGKB KiB can hold two frames of
template <int RINGKB, bool DUAL>
constexpr int max_pieces_of() {
  return Cls<RINGKB, 32, DUAL>::K >= 2   ? 32
         : Cls<RINGKB, 24, DUAL>::K >= 2 ? 24
         : Cls<RINGKB, 16, DUAL>::K >= 2 ? 16
         : Cls<RINGKB, 12, DUAL>::K >= 2 ? 12
         : Cls<RINGKB, 8, DUAL>::K >= 2  ? 8
                                         : 0;
}

template <int KS, int RINGKB, int WAVES>
hipError_t launch_one(const TiledArgs& a, int groups, hipStream_t stream) {
  constexpr int lds_bytes = RINGKB * 1024;
  if (a.max_pieces > max_pieces_of<RINGKB, dual_copy(KS)>()) return hipErrorInvalidValue;
  if (lds_bytes > 64 * 1024) {
    // per device and cheap: not cached (handles may live on several devices of one process)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_kernel<KS, RINGKB, WAVES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
  }
  const int per_xcd = (a.total_tiles + 7) / 8;
  const int tail = (per_xcd * a.tail_percent) / 100;
  const int items = (per_xcd - tail) * groups + (tail + 1) * a.tail_groups;  // >= any XCD's item count
  size_t lds_request = (size_t)lds_bytes;
#ifdef T360_INSTRUMENT
  if (a.lds_pad > 0) {  // occupancy experiments: ask for more LDS than the ring needs (fewer workgroups per CU)
    lds_request += (size_t)a.lds_pad;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_kernel<KS, RINGKB, WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_request);
  }
#endif
  hipLaunchKernelGGL((remap_tiled_kernel<KS, RINGKB, WAVES>), dim3(a.direct_blocks + 8 * items, 1, 1), dim3(64 * WAVES),
                     lds_request, stream, a);
  return hipGetLastError();
}

template <int KS>
hipError_t launch_ks(const TiledArgs& a, int groups, hipStream_t stream) {
  if (a.waves == 8) {
    if (a.ring_kb == 76) return launch_one<KS, 76, 8>(a, groups, stream);  // 2 workgroups of 8 waves per CU
#ifdef T360_INSTRUMENT
    if (a.ring_kb == 50) return launch_one<KS, 50, 8>(a, groups, stream);  // 3 workgroups of 8 waves per CU
#endif
    return hipErrorInvalidValue;
  }
  if (a.ring_kb == 38) return launch_one<KS, 38, 4>(a, groups, stream);  // 4 workgroups of 4 waves per CU
#ifdef T360_INSTRUMENT
  if (a.ring_kb == 26) return launch_one<KS, 26, 4>(a, groups, stream);  // 6 workgroups per CU
  if (a.ring_kb == 31) return launch_one<KS, 31, 4>(a, groups, stream);  // 5 workgroups per CU
  if (a.ring_kb == 50) return launch_one<KS, 50, 4>(a, groups, stream);  // 3 workgroups per CU
  if (a.ring_kb == 76) return launch_one<KS, 76, 4>(a, groups, stream);  // 2 workgroups per CU
#endif
  return hipErrorInvalidValue;
}


// ============================ fused low-pass tiles ==============================================
// remap_fused_kernel: the tiled gather with the segmented low-pass of its own footprint done in LDS (t360_internal.h
// "fused low-pass tiles"; reference filterPlane -> filterSegment -> cv::sepFilter2D feeding cv::remap on ONE plane,
// VideoFrameTransform.cpp:727-733 + :748-754, :173-204).  What changes against tile_waves():
//   * the DMA stages the RAW footprint R, dilated by the kernel radius (one 16-byte chunk left and right, one row above and
//     below, clamped at the plane's top and bottom: BORDER_REPLICATE);
//   * every lane owns one vertical RUN of <= kFusedMaxRun blurred dwords (4 px each) of one dword column: per R row of the
//     run it reads the three aligned dwords around its column, takes the row pass of its 4 pixels as v_dot4_u32_u8 against
//     the byte-shifted tap variants (SGPRs: a wave's runs share one kernel), keeps the last three row results in registers
//     and takes the column pass as v_mad_u32_u24 -- the arithmetic of lowpass_q8w_rows(), bit for bit;
//   * the blurred dwords are written IN PLACE into the same ring slot at the B positions (the layout the gather's pixel
//     words address) after a second barrier: every R byte of the slot has been read by then;
//   * the gather runs one frame behind the filter, so the frame loop is
//       wait DMA(i) | BARRIER | gather B(i-1) | filter R(i) into registers | BARRIER | store(i-1), write B(i), DMA(i+K-1).
#ifndef T360_FUSED_GROUP
#define T360_FUSED_GROUP 2
#endif
struct FusedFetch {
  uint32_t rowdw_r;  // dword `lane` of the R row table
  uint32_t run;      // this lane's run word
  uint32_t kid;      // kernel index of this wave's runs
  uint32_t ni;       // longest run of the tile
};

template <int KS, int P, int K>
__device__ __forceinline__ void fused_waves(const TiledArgs& a, const TiledPlane& pl, const uint32_t* __restrict__ taps,
                                            const TileDesc& t, const TileFetch& tf, const FusedFetch& ff,
                                            const uint8_t* __restrict__ lds, int f0, int f1) {
  // (GROUP = 2: two pixels' stencil reads in flight at a time instead of all four -- 16 registers the filter phase's
  // addresses and row window need; the frame's LDS round trips are dominated by the filter's row steps anyway)
  constexpr int WAVES = 8, NPX = 4, GROUP = T360_FUSED_GROUP;
  using R = Slot<P, false>;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mine = ((int)t.pieces - wave + WAVES - 1) / WAVES;
  const bool has_px = t.kind == kTileWide128 || wave < 4;
  int goff[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int j = 3 - k;
    const uint32_t e = tf.chunk[j];
    goff[k] = (j < mine && WAVES * j < P) ? (int)(e >> 12) * pl.sstride + (int)(e & 4095u) * kStageChunk : 0;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds + (uint32_t)wave * 1024u;
  auto issue = [&](int f, int slot_bytes) {
    if (mine <= 0) return;
    const uint8_t* base = pl.src + (size_t)f * pl.src_frame_bytes;
    dma_frame_4(mine, base, lds_base + (uint32_t)slot_bytes, WAVES * 1024u, goff);
  };
  const int per_frame = mine;
  const int nf = f1 - f0;
#pragma unroll
  for (int j = 0; j < K - 1; j++)
    if (j < nf) issue(f0 + j, j * R::kSlot);
  PixelSetup<NPX, KS> px;
  load_pixels<NPX, KS, P>(tf, a.wpack, px);
  // ---- the lane's run: LDS addresses (inside a slot) of its R rows and of the B dwords it writes, two per register ----
  const uint32_t rw = ff.run;
  const int run_len = (int)((rw >> 16) & 15u);  // >= 1: a plan has no idle lanes (t360_internal.h)
  const int dcol = (int)(rw & 511u), r0 = (int)((rw >> 9) & 127u);
  const int ni = __builtin_amdgcn_readfirstlane((int)ff.ni);
  constexpr int NR = kFusedMaxRun + 2;
  uint32_t raddr2[NR / 2], baddr2[kFusedMaxRun / 2];
#pragma unroll
  for (int i = 0; i < NR; i += 2) {
    // R row t of the tile is source row y0 - 1 + t; the R box starts one chunk left of the B box: byte 4 * dcol of the B box is
    // byte 16 + 4 * dcol of the R box, and the 12-byte window starts 4 bytes left of it
    const int a0 = row_base_of(ff.rowdw_r, (r0 + i) & (kBoxMaxRows - 1)) * kStageChunk + 12 + 4 * dcol;
    const int a1 = row_base_of(ff.rowdw_r, (r0 + i + 1) & (kBoxMaxRows - 1)) * kStageChunk + 12 + 4 * dcol;
    raddr2[i / 2] = i < ni + 2 ? ((uint32_t)a0 & 0xffffu) | ((uint32_t)a1 << 16) : 0u;
  }
#pragma unroll
  for (int j = 0; j < kFusedMaxRun; j += 2) {
    // slots of the run beyond its length aim at its LAST dword: the dwords are written in descending order, so whatever a
    // shorter run computes out of rows that are not its own is overwritten by the real thing (one lane's LDS stores are
    // executed in order) -- no per-lane predicate anywhere in the filter phase
    const int b0 = row_base_of(tf.rowdw, (r0 + min(j, run_len - 1)) & (kBoxMaxRows - 1)) * kStageChunk + 4 * dcol;
    const int b1 = row_base_of(tf.rowdw, (r0 + min(j + 1, run_len - 1)) & (kBoxMaxRows - 1)) * kStageChunk + 4 * dcol;
    baddr2[j / 2] = j < ni ? ((uint32_t)b0 & 0xffffu) | ((uint32_t)b1 << 16) : 0u;
  }
  // the wave's kernel: 10 horizontal tap dwords + 3 vertical taps, scalar
  uint32_t tp[13];
  {
    const uint32_t* __restrict__ q = taps + (size_t)__builtin_amdgcn_readfirstlane((int)ff.kid) * kFusedTapDwords;
#pragma unroll
    for (int k = 0; k < 13; k++) tp[k] = __builtin_amdgcn_readfirstlane((int)q[k]);
  }
  pin_pixels<NPX, KS>(px);
#pragma unroll
  for (int i = 0; i < NR / 2; i++) asm volatile("" : "+v"(raddr2[i]));
#pragma unroll
  for (int j = 0; j < kFusedMaxRun / 2; j++) asm volatile("" : "+v"(baddr2[j]));
  const bool edge_l = (rw & kRunLeftEdge) != 0, edge_r = (rw & kRunRightEdge) != 0;
  const bool any_edge = __builtin_amdgcn_readfirstlane((int)(__ballot(edge_l || edge_r) != 0ull)) != 0;

  const bool dword_store = !(t.flags & kTilePartial) && pl.dst_dword_ok;
  const uint32_t doff = out_pos<NPX>(pl, t, dword_store, 0u);
  uint8_t* __restrict__ d = uniform_ptr(pl.dst + (size_t)f0 * pl.dst_frame_bytes);
  uint32_t pending = 0;
  uint32_t half = 1u << 15;
  asm volatile("" : "+v"(half));

  // one frame's filter pass out of the slot at byte SLOT: blurred dwords of the run into bl[]
  uint32_t bl[kFusedMaxRun] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto filter = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    uint32_t res[3][4];
    // (plain loads: hipcc counts them itself, so it may keep the next row in flight while this one is consumed.  With asm
    // loads it could not be kept from COPYING a destination register before the s_waitcnt that covers it -- the row steps
    // are separate basic blocks, and the copies appear where they merge)
#pragma unroll
    for (int i = 0; i < NR; i++) {
      if (i < ni + 2) {  // wave-uniform
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(
            lds + SLOT + ((i & 1) ? (raddr2[i / 2] >> 16) : (raddr2[i / 2] & 0xffffu)));
        uint32_t D0 = q[0], D1 = q[1], D2 = q[2];
        if (any_edge) {
          if (edge_l) D0 = (D1 & 0xffu) * 0x01010101u;
          if (edge_r) D2 = (D1 >> 24) * 0x01010101u;
        }
        uint32_t* r = res[i % 3];
        r[0] = __builtin_amdgcn_udot4(D1, tp[1], __builtin_amdgcn_udot4(D0, tp[0], 0u, false), false);
        r[1] = __builtin_amdgcn_udot4(D2, tp[4], __builtin_amdgcn_udot4(D1, tp[3], __builtin_amdgcn_udot4(D0, tp[2], 0u, false), false), false);
        r[2] = __builtin_amdgcn_udot4(D2, tp[7], __builtin_amdgcn_udot4(D1, tp[6], __builtin_amdgcn_udot4(D0, tp[5], 0u, false), false), false);
        r[3] = __builtin_amdgcn_udot4(D2, tp[9], __builtin_amdgcn_udot4(D1, tp[8], 0u, false), false);
        if (i >= 2) {
          uint32_t c[4];
#pragma unroll
          for (int p = 0; p < 4; p++) {
            c[p] = half;
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
              const uint32_t kyk = tp[10 + k], rv = res[(i - 2 + k) % 3][p], cv = c[p];
              uint32_t nv;
              asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(nv) : "s"(kyk), "v"(rv), "v"(cv));
              c[p] = nv;
            }
          }
          const uint32_t lo = sat_pk_u8(__builtin_amdgcn_perm(c[1], c[0], 0x07060302u));
          const uint32_t hi = sat_pk_u8(__builtin_amdgcn_perm(c[3], c[2], 0x07060302u));
          bl[i - 2] = lo | (hi << 16);
        }
      }
    }
  };
  auto write_blurred = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int j = kFusedMaxRun - 1; j >= 0; j--)
      if (j < ni) {  // wave-uniform
        *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + SLOT + ((j & 1) ? (baddr2[j / 2] >> 16) : (baddr2[j / 2] & 0xffffu))) = bl[j];
      }
  };

  for (int i = 0; i <= nf; i += K) {
#define T360_FSTEP(S)                                                                                          \
    if constexpr (S < K) if (i + S <= nf) {                                                                    \
      constexpr int SP = (S + K - 1) % K; /* the slot of the previous frame */                                 \
      const int fr = i + S;               /* frame filtered in this step; the gather handles frame fr - 1 */   \
      if (fr < nf) wait_vmcnt(min(K - 2, nf - 1 - fr) * per_frame);                                            \
      frame_barrier(); /* R(fr) is complete in slot S, B(fr - 1) in slot SP */                                 \
      if (fr > 0 && has_px && !T360_DBG(a, 0)) pending = gather<NPX, KS, GROUP, SP * R::kSlot>(px, lds, dword_store);             \
      asm volatile("" : "+v"(pending));                                                                        \
      if (fr < nf && !T360_DBG(a, 11)) filter(std::integral_constant<int, S * R::kSlot>{});                    \
      if (!T360_DBG(a, 13)) frame_barrier(); /* every read of slot S (R) and of slot SP (B) is done */         \
      if (fr > 0) {                                                                                            \
        if (has_px) emit<NPX, KS>(px, pending, d, doff, pl.dstride, dword_store, true);                        \
        d += pl.dst_frame_bytes;                                                                               \
      }                                                                                                        \
      if (fr < nf && !T360_DBG(a, 12)) write_blurred(std::integral_constant<int, S * R::kSlot>{});             \
      if (fr + K - 1 < nf && !T360_DBG(a, 1)) issue(f0 + fr + K - 1, SP * R::kSlot);                           \
    }
    T360_FSTEP(0) T360_FSTEP(1) T360_FSTEP(2)
#undef T360_FSTEP
  }
}

template <int KS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void remap_fused_kernel(FusedArgs fa) {
  constexpr int RINGKB = 76, WAVES = 8;
  const TiledArgs& a = fa.base;
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  // the XCD-aware work item order of remap_tiled_kernel (no direct tiles in this list)
  int id = blockIdx.x, b, g, f0, f1;
  TiledPlane pl = a.plane[0];
  const uint32_t* taps = fa.taps[0];
  {
    const int xcd = id & 7, k = id >> 3;
    const int q = a.total_tiles >> 3, rem = a.total_tiles & 7;
    const int len = q + (xcd < rem ? 1 : 0), start = xcd * q + (xcd < rem ? xcd : rem);
    const int len_tail = (len * a.tail_percent) / 100, len_head = len - len_tail;
    int fpb;
    if (k < len_head * a.groups) {
      const int t_local = k / a.groups;
      b = start + t_local;
      g = k - t_local * a.groups;
      fpb = a.frames_per_block;
    } else {
      const int k2 = k - len_head * a.groups;
      const int t_local = k2 / a.tail_groups;
      if (t_local >= len_tail) return;
      b = start + len_head + t_local;
      g = k2 - t_local * a.tail_groups;
      fpb = a.tail_frames;
    }
    f0 = g * fpb;
    f1 = min(f0 + fpb, a.nframes);
  }
  if (a.nplanes > 1 && b >= pl.ntiles) {
    b -= pl.ntiles;
    pl = a.plane[1];
    taps = fa.taps[1];
    if (a.nplanes > 2 && b >= pl.ntiles) {
      b -= pl.ntiles;
      pl = a.plane[2];
      taps = fa.taps[2];
      if (a.nplanes > 3 && b >= pl.ntiles) {
        b -= pl.ntiles;
        pl = a.plane[3];
        taps = fa.taps[3];
      }
    }
  }
  // per-tile tables: pixel words, R chunk entries, B row table (as in an unfused plan), then the R row table, the run words
  // and the wave info
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TileFetch tf;
  FusedFetch ff;
  {
    const uint4 v = reinterpret_cast<const uint4*>(pl.tlut + (size_t)b * tile_words(KS, WAVES))[tid];
    tf.words[0] = v.x; tf.words[1] = v.y; tf.words[2] = v.z; tf.words[3] = v.w;
    const uint32_t* __restrict__ tc = pl.chunks + (size_t)b * fused_chunk_dwords(a.max_pieces);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int piece = wave + WAVES * j;
      tf.chunk[j] = piece < a.max_pieces ? tc[piece * kPieceChunks + lane] : 0u;
    }
    const uint32_t* __restrict__ tb = tc + a.max_pieces * kPieceChunks;
    tf.rowdw = tb[lane];
    tf.origin = 0u;
    ff.rowdw_r = tb[64 + lane];
    ff.run = tb[128 + tid];
    ff.kid = tb[128 + kFusedLanes + wave];
    ff.ni = tb[128 + kFusedLanes + 8];
  }
  const TileDesc t = pl.tiles[b];
  const int pieces = (int)t.pieces;
#define T360_FTILE(P) \
  if constexpr (Cls<RINGKB, P, false>::K >= 2) fused_waves<KS, P, Cls<RINGKB, P, false>::K>(a, pl, taps, t, tf, ff, lds, f0, f1);
  T360_FOR_CLASS(pieces, T360_FTILE)
#undef T360_FTILE
}

template <int KS>
hipError_t launch_fused_ks(const FusedArgs& fa, hipStream_t stream) {
  constexpr int lds_bytes = 76 * 1024;
  const TiledArgs& a = fa.base;
  if (a.max_pieces > max_pieces_of<76, false>()) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_fused_kernel<KS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  const int per_xcd = (a.total_tiles + 7) / 8;
  const int tail = (per_xcd * a.tail_percent) / 100;
  const int items = (per_xcd - tail) * groups + (tail + 1) * a.tail_groups;
  hipLaunchKernelGGL((remap_fused_kernel<KS>), dim3(8 * items, 1, 1), dim3(512), (size_t)lds_bytes, stream, fa);
  return hipGetLastError();
}

}  // namespace

const char* remap_tiled_kernel_name(int ks, int ring_kb, int waves) {
  static thread_local char buf[64];
  snprintf(buf, sizeof(buf), "remap_tiled_kernel<%d, %d, %d>", ks, ring_kb, waves);
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

hipError_t launch_remap_fused(const FusedArgs& fa, hipStream_t stream) {
  const TiledArgs& a = fa.base;
  if (a.total_tiles <= 0 || a.nframes <= 0) return hipSuccess;
  if (a.waves != 8 || a.ring_kb != 76 || a.total_direct != 0) return hipErrorInvalidValue;
  switch (a.ks) {
    case 2: return launch_fused_ks<2>(fa, stream);
    case 4: return launch_fused_ks<4>(fa, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace t360
