// t360_lowpass.hip -- segmented separable low-pass filter (the reference's filterPlane ->
// runFiltering -> filterSegment -> cv::sepFilter2D chain, VideoFrameTransform.cpp:173-204,
// 579-704).
//
// Semantics restated: for an output pixel (x, y) of segment S
//     out = colpass_S( rowpass_S( plane[clampY(y+dy)][clampX(x+dx)] ) )
// i.e. pixels outside the segment are the REAL neighbours in the whole plane (the segment is an
// ROI of the parent image and BORDER_ISOLATED is not set), replicated only beyond the plane's
// own edges, and halo rows are row-filtered with S's own horizontal kernel.
//   fixed-point path (both kernels SMOOTH|SYMMETRICAL): taps * 256 rounded to int,
//       r = SUM kx_q8 * src (int32), c = SUM ky_q8 * r, out = sat_u8((c + 32768) >> 16)
//   float path: r = SUM kx * src (ascending), c = ky[0]*r0 + SUM ky[k]*(r+k + r-k) for a
//       symmetric vertical kernel (ascending otherwise), out = sat_u8(rint(c))
//
// One workgroup per (frame, tile); a tile is a <= tile_w x tile_h rectangle inside ONE segment.
// Two kernels:
//   lowpass_q8_kernel   the fast path (fixed-point segments, byte-sized horizontal taps, 3/5/7 vertical
//                       taps): registers only, v_dot4 row pass, see below;
//   lowpass_kernel      everything else (float path, long kernels, odd alignments): pass 1 writes the
//                       row-filtered values of the tile's rows plus the vertical halo to LDS, pass 2 reads
//                       them column-wise; source bytes come straight from global memory.
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"

namespace t360 {

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void lowpass_kernel(LowpassArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_rows[];  // [rows][tile_w] int32 or float bits

  const LowpassTile t = a.tiles[blockIdx.x];
  const SegmentDev s = a.segs[t.seg];
  const uint8_t* __restrict__ src = a.src + (size_t)blockIdx.y * a.src_frame_bytes;
  uint8_t* __restrict__ dst = a.dst + (size_t)blockIdx.y * a.dst_frame_bytes;
  const int rx = s.kx_len >> 1, ry = s.ky_len >> 1;
  const int rows = t.h + 2 * ry;
  const int tw = t.w;
  const int pitch = a.tile_w;

  if (s.fixed_point) {
    const int* __restrict__ kx = a.taps_q8 + s.kx_off;
    const int* __restrict__ ky = a.taps_q8 + s.ky_off;
    for (int idx = threadIdx.x; idx < rows * tw; idx += blockDim.x) {
      const int r = idx / tw, x = idx - r * tw;
      const uint8_t* __restrict__ S = src + (size_t)clampi(t.y0 - ry + r, 0, a.h - 1) * a.sstride;
      const int xs = t.x0 + x - rx;
      int acc = 0;
      if (xs >= 0 && xs + s.kx_len <= a.w) {
        for (int k = 0; k < s.kx_len; k++) acc += kx[k] * (int)S[xs + k];
      } else {
        for (int k = 0; k < s.kx_len; k++) acc += kx[k] * (int)S[clampi(xs + k, 0, a.w - 1)];
      }
      lds_rows[r * pitch + x] = acc;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < t.h * tw; idx += blockDim.x) {
      const int y = idx / tw, x = idx - y * tw;
      int acc = 0;
      for (int k = 0; k < s.ky_len; k++) acc += ky[k] * lds_rows[(y + k) * pitch + x];
      dst[(size_t)(t.y0 + y) * a.dstride + t.x0 + x] = (uint8_t)sat_u8((acc + (1 << 15)) >> 16);
    }
  } else {
    float* frow = reinterpret_cast<float*>(lds_rows);
    const float* __restrict__ kx = a.taps_f32 + s.kx_off;
    const float* __restrict__ ky = a.taps_f32 + s.ky_off;
    for (int idx = threadIdx.x; idx < rows * tw; idx += blockDim.x) {
      const int r = idx / tw, x = idx - r * tw;
      const uint8_t* __restrict__ S = src + (size_t)clampi(t.y0 - ry + r, 0, a.h - 1) * a.sstride;
      const int xs = t.x0 + x - rx;
      float acc = __fmul_rn(kx[0], (float)S[clampi(xs, 0, a.w - 1)]);
      for (int k = 1; k < s.kx_len; k++)
        acc = __fadd_rn(acc, __fmul_rn(kx[k], (float)S[clampi(xs + k, 0, a.w - 1)]));
      frow[r * pitch + x] = acc;
    }
    __syncthreads();
    // symmetric vertical kernel? (device-side check mirrors cv::getKernelType's palindrome test)
    bool symmetric = (s.ky_len & 1) != 0;
    for (int k = 0; k < ry && symmetric; k++) symmetric = ky[k] == ky[s.ky_len - 1 - k];
    for (int idx = threadIdx.x; idx < t.h * tw; idx += blockDim.x) {
      const int y = idx / tw, x = idx - y * tw;
      float acc;
      if (symmetric) {
        acc = __fmul_rn(ky[ry], frow[(y + ry) * pitch + x]);
        for (int k = 1; k <= ry; k++)
          acc = __fadd_rn(acc, __fmul_rn(ky[ry + k], __fadd_rn(frow[(y + ry + k) * pitch + x],
                                                               frow[(y + ry - k) * pitch + x])));
      } else {
        acc = __fmul_rn(ky[0], frow[y * pitch + x]);
        for (int k = 1; k < s.ky_len; k++) acc = __fadd_rn(acc, __fmul_rn(ky[k], frow[(y + k) * pitch + x]));
      }
      dst[(size_t)(t.y0 + y) * a.dstride + t.x0 + x] = (uint8_t)sat_u8(__float2int_rn(acc));
    }
  }
}


// ============================ fast path: Q8 x Q8 fixed point ===================================
// For segments on the fixed-point path whose horizontal taps all fit a byte (every Gaussian of
// >= 3 taps does, <= 64 taps) and whose vertical kernel has KY in {3, 5, 7} taps; plane width,
// bases and strides multiples of 4.  One workgroup per tile of <= 128 x 128 px of ONE segment:
//   1. the tile's source rectangle (+ kernel radius) goes to LDS as bytes, ONE coalesced dword
//      load per source dword (the first version let every lane load its own window from global
//      memory: 6x redundant, and the vector-memory path, not HBM, bound the kernel).  The
//      replicate border is resolved here: the plane is a whole number of dwords wide, so a dword
//      is either inside the row or entirely outside it (splat of the row's first / last byte);
//      rows above / below the plane are clamped.
//   2. a row group of 16 or 32 lanes owns <= 64 / 128 consecutive pixels of a run of rows, 4 px
//      per lane.  Per source row and per 16 taps a lane reads the <= 6 LDS dwords covering its 4
//      pixels' windows, realigns them (v_alignbit) and takes the row pass as v_dot4_u32_u8 against
//      the packed taps held in SGPRs: 2 VALU per 4 taps per pixel.
//   3. the last KY row-pass results stay in registers (sliding window); the column pass is KY
//      v_mad_u32_u24 per pixel; 4 pixels leave as one dword store.
// Arithmetic is identical to the generic kernel above: r = SUM kx_q8*src, c = SUM ky_q8*r,
// out = sat_u8((c + 32768) >> 16), all exact in int32 (r < 2^17, taps <= 256: 24-bit multiplies).

// NG = tap groups (dwords of 4 packed taps) the instantiation holds; EXACT: G == NG, no guards.
template <int KY, int NG, bool EXACT>
__device__ __forceinline__ void lowpass_q8_rows(const LowpassArgs& a, const LowpassTile& t,
                                                const uint32_t* __restrict__ box, int pitch, int m, int nd, int G,
                                                const uint32_t* __restrict__ kxp, const uint32_t (&kyv)[KY],
                                                uint8_t* __restrict__ dst) {
  constexpr int ry = KY >> 1;
  uint32_t kx4[NG];
#pragma unroll
  for (int j = 0; j < NG; j++) kx4[j] = (EXACT || j < G) ? kxp[j] : 0u;  // scalar loads, once

  // lanes per row group: 16 for tiles up to 64 px wide, else 32
  const int lshift = t.w <= 64 ? 4 : 5;
  const int lir = threadIdx.x & ((1 << lshift) - 1), grp = threadIdx.x >> lshift;
  const int ngroups = 256 >> lshift;
  const int rpg = (t.h + ngroups - 1) / ngroups;
  const int r0 = grp * rpg;  // first output row of this group, relative to the tile
  const int r1 = min(r0 + rpg, t.h);
  if (r0 >= r1 || 4 * lir >= t.w) return;
  const int px0 = t.x0 + 4 * lir;
  const int npx = min(4, t.w - 4 * lir);
  const bool dword_out = npx == 4 && a.dst_dword_ok && (px0 & 3) == 0;
  const uint32_t mshift = (uint32_t)m * 8u;

  // the last KY row-pass results: slot (row index mod KY); the row loop is unrolled KY times so
  // slot numbers are compile-time constants and the window never moves between registers
  uint32_t win[KY][4];
#pragma unroll
  for (int k = 0; k < KY; k++)
#pragma unroll
    for (int p = 0; p < 4; p++) win[k][p] = 0u;

  const uint32_t* __restrict__ lane_row = box + r0 * pitch + lir;
  uint8_t* __restrict__ d = dst + (size_t)(t.y0 + r0) * a.dstride + px0;
  const int rend = r1 + 2 * ry;  // staged rows r0 .. rend-1 (staged row r = source row t.y0 - ry + r)
  for (int rb = r0; rb < rend; rb += KY) {
#pragma unroll
    for (int u = 0; u < KY; u++) {
      const int r = rb + u;
      if (r < rend) {
        uint32_t D[NG + 2];
#pragma unroll
        for (int i = 0; i < NG + 2; i++) D[i] = (i < NG + 1 || m != 0) && (EXACT || i < nd) ? lane_row[i] : 0u;
        uint32_t R[NG + 1];  // R[i] = source bytes (px0 - rx) + 4i .. +3
#pragma unroll
        for (int i = 0; i < NG + 1; i++) R[i] = __builtin_amdgcn_alignbit(D[i + 1], D[i], mshift);
        uint32_t acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < NG; j++) {
          if (EXACT || j < G) {  // wave-uniform
            acc[0] = __builtin_amdgcn_udot4(R[j], kx4[j], acc[0], false);
            acc[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbit(R[j + 1], R[j], 8u), kx4[j], acc[1], false);
            acc[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbit(R[j + 1], R[j], 16u), kx4[j], acc[2], false);
            acc[3] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbit(R[j + 1], R[j], 24u), kx4[j], acc[3], false);
          }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) win[u][p] = acc[p];
        if (r >= r0 + 2 * ry) {  // the window of output row r - 2 ry is complete: slots u+1 .. u (mod KY)
          uint32_t c[4];
#pragma unroll
          for (int p = 0; p < 4; p++) {
            c[p] = 1u << 15;
#pragma unroll
            for (int k = 0; k < KY; k++) c[p] = __umul24(kyv[k], win[(u + 1 + k) % KY][p]) + c[p];
            c[p] = min(c[p], 0x00ffffffu);  // byte 2 is now sat_u8(c >> 16)
          }
          if (dword_out) {
            const uint32_t lo = __builtin_amdgcn_perm(c[1], c[0], 0x0c0c0602u);  // [c0.b2, c1.b2, 0, 0]
            const uint32_t hi = __builtin_amdgcn_perm(c[3], c[2], 0x06020c0cu);  // [0, 0, c2.b2, c3.b2]
            *reinterpret_cast<uint32_t*>(d) = lo | hi;
          } else {
#pragma unroll
            for (int p = 0; p < 4; p++)
              if (p < npx) d[p] = (uint8_t)(c[p] >> 16);
          }
          d += a.dstride;
        }
        lane_row += pitch;
      }
    }
  }
}

template <int KY>
__global__ __launch_bounds__(256) void lowpass_q8_kernel(LowpassArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_rows[];
  uint32_t* __restrict__ box = reinterpret_cast<uint32_t*>(lds_rows);

  const LowpassTile t = a.fast_tiles[blockIdx.x];
  const SegmentDev s = a.segs[t.seg];
  const uint8_t* __restrict__ src = a.src + (size_t)blockIdx.y * a.src_frame_bytes;
  uint8_t* __restrict__ dst = a.dst + (size_t)blockIdx.y * a.dst_frame_bytes;
  const int rx = s.kx_len >> 1;
  constexpr int ry = KY >> 1;
  const int G = s.kx_groups;  // packed tap dwords, 1..16
  const uint32_t* __restrict__ kxp = a.taps_pk + s.kxp_off;
  const int* __restrict__ ky = a.taps_q8 + s.ky_off;
  uint32_t kyv[KY];
#pragma unroll
  for (int k = 0; k < KY; k++) kyv[k] = (uint32_t)ky[k];

  // ---- 1. stage the source rectangle ----
  const int m = (t.x0 - rx) & 3;            // byte offset of the first needed byte inside its dword
  const int dx0 = (t.x0 - rx - m) >> 2;     // first staged dword column (may be < 0)
  const int nd = G + 1 + (m ? 1 : 0);       // dwords one lane's window spans
  const int ndw = ((t.w + 3) >> 2) + nd - 1;  // staged dwords per row (<= 32 + 17)
  const int pitch = (ndw + 3) & ~3;           // LDS row pitch in dwords: rows start 16-byte aligned
  const int rows = t.h + 2 * ry;
  const int W4 = a.w >> 2;
  {
    // 16 lanes x 16 bytes per row, 16 rows per pass of the workgroup
    const int c4 = (threadIdx.x & 15) * 4;
    if (c4 < ndw) {
      const int di = dx0 + c4;
      const bool inside = di >= 0 && di + 4 <= W4;
      for (int r = threadIdx.x >> 4; r < rows; r += 16) {
        const uint32_t* __restrict__ q =
            reinterpret_cast<const uint32_t*>(src + (size_t)clampi(t.y0 - ry + r, 0, a.h - 1) * a.sstride);
        uint4 v;
        if (inside) {
          v = *reinterpret_cast<const uint4*>(q + di);  // dword aligned; the hardware takes it
        } else {
          uint32_t e[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int dk = di + k;
            uint32_t u = q[clampi(dk, 0, W4 - 1)];
            if (dk < 0) u = (u & 0xffu) * 0x01010101u;
            if (dk >= W4) u = (u >> 24) * 0x01010101u;
            e[k] = u;
          }
          v = make_uint4(e[0], e[1], e[2], e[3]);
        }
        *reinterpret_cast<uint4*>(box + r * pitch + c4) = v;
      }
    }
  }
  __syncthreads();

  // ---- 2./3. row pass from LDS, column pass in registers ----
  // specialised on the number of tap groups so the packed taps are loop-invariant scalars
  switch (G) {
    case 1: lowpass_q8_rows<KY, 1, true>(a, t, box, pitch, m, nd, G, kxp, kyv, dst); break;
    case 2: lowpass_q8_rows<KY, 2, true>(a, t, box, pitch, m, nd, G, kxp, kyv, dst); break;
    case 3: lowpass_q8_rows<KY, 3, true>(a, t, box, pitch, m, nd, G, kxp, kyv, dst); break;
    case 4: lowpass_q8_rows<KY, 4, true>(a, t, box, pitch, m, nd, G, kxp, kyv, dst); break;
    default:
      if (G <= 8)
        lowpass_q8_rows<KY, 8, false>(a, t, box, pitch, m, nd, G, kxp, kyv, dst);
      else if (G <= 12)
        lowpass_q8_rows<KY, 12, false>(a, t, box, pitch, m, nd, G, kxp, kyv, dst);
      else
        lowpass_q8_rows<KY, 16, false>(a, t, box, pitch, m, nd, G, kxp, kyv, dst);
  }
}

// ============================ wide fast path ====================================================
// Same arithmetic and the same register-resident column pass as lowpass_q8_kernel, but
//   * tiles are <= 512 x 32 px of a run of segments (2 row groups of 128 lanes x 4 px): a staged row is
//     ~550 bytes instead of ~170, so the 128-byte lines at its two ends are a fifth of what it fetches, not half
//     (the 128 x 128 tiles read 3x the plane through the fabric: profiles/r02_*), and workgroup ids map to
//     XCD-contiguous ranges of the row-major tile list, so vertical neighbours share an L2;
//   * the row pass never realigns pixels.  A lane's 4 output pixels px0 .. px0+3 (px0 % 4 == 0) read the ALIGNED
//     source dwords from byte (px0 - rx - m) on, m = (-rx) & 3, and output pixel j takes its dot products against
//     the packed taps shifted by m + j bytes: four tap variants (SGPRs, set up by the host) replace 7 of every
//     8 v_alignbit, ND v_dot4 per pixel and nothing else.
// two int16 -> two saturated uint8 in bytes 0 and 1 (upper half zero)
__device__ __forceinline__ uint32_t sat_pk_u8_i16(uint32_t two_i16) {
  uint32_t r;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(two_i16));
  return r;
}

// KX > 0: the exact number of horizontal taps is a compile-time constant, so the all-zero dwords of a shifted tap
// variant (which dwords they are follows from KX alone) cost nothing: 6 instead of 12 v_dot4 per 4 pixels for 3
// taps, 8 for 5, 10 for 7, 16 instead of 20 for 13.  KX == 0: any length whose window fits ND dwords.
template <int KY, int ND, int KX>
__device__ __forceinline__ void lowpass_q8w_rows(const LowpassArgs& a, const LowpassTile& t,
                                                 const uint32_t* __restrict__ box, int pitch,
                                                 const uint32_t* __restrict__ kxs, const uint32_t (&kyv)[KY],
                                                 uint8_t* __restrict__ dst) {
  constexpr int ry = KY >> 1;
  constexpr int M = KX > 0 ? (4 - (KX / 2) % 4) % 4 : 0;  // the host's m for this length
  auto used = [](int j, int i) { return KX == 0 || (i >= (M + j) / 4 && i <= (M + j + KX - 1) / 4); };
  uint32_t T[4][ND];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < ND; i++) T[j][i] = used(j, i) ? kxs[j * kWideTapStride + i] : 0u;  // scalar loads
  constexpr int kGroups = 1024 / kWideTileW;  // row groups of kWideTileW / 4 lanes
  // a row group is a whole number of waves: its row range is wave-uniform (scalar loop control)
  const int lir = threadIdx.x & (kWideTileW / 4 - 1), grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (kWideTileW / 4));
  const int rpg = (t.h + kGroups - 1) / kGroups;
  const int r0 = grp * rpg;  // first output row of this group, relative to the tile
  const int r1 = min(r0 + rpg, t.h);
  if (r0 >= r1 || 4 * lir >= t.w) return;
  const int px0 = t.x0 + 4 * lir;
  const int npx = min(4, t.w - 4 * lir);
  const bool dword_out = npx == 4 && a.dst_dword_ok;

  uint32_t win[KY][4];
#pragma unroll
  for (int k = 0; k < KY; k++)
#pragma unroll
    for (int p = 0; p < 4; p++) win[k][p] = 0u;
  uint32_t half = 1u << 15;  // FixedPtCastEx's rounding term, in a register (v_mad_u32_u24 takes no literal)
  asm volatile("" : "+v"(half));

  const uint32_t* __restrict__ lane_row = box + r0 * pitch + lir;
  uint8_t* __restrict__ d = dst + (size_t)(t.y0 + r0) * a.dstride + px0;
  const int rend = r1 + 2 * ry;  // staged rows r0 .. rend-1 (staged row r = source row t.y0 - ry + r)
  for (int rb = r0; rb < rend; rb += KY) {
#pragma unroll
    for (int u = 0; u < KY; u++) {
      const int r = rb + u;
      if (r < rend) {
        uint32_t D[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) D[i] = lane_row[i];
#pragma unroll
        for (int p = 0; p < 4; p++) {
          uint32_t acc = 0u;
#pragma unroll
          for (int i = 0; i < ND; i++)
            if (used(p, i)) acc = __builtin_amdgcn_udot4(D[i], T[p][i], acc, false);
          win[u][p] = acc;
        }
        if (r >= r0 + 2 * ry) {  // the window of output row r - 2 ry is complete: slots u+1 .. u (mod KY)
          // KY multiply-adds per pixel, seeded with the rounding constant (hipcc, left alone, multiplies twice and adds
          // three values: four instructions where three do)
          uint32_t c[4];
#pragma unroll
          for (int p = 0; p < 4; p++) {
            c[p] = half;
#pragma unroll
            for (int k = 0; k < KY; k++)
              asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c[p]) : "v"(kyv[k]), "v"(win[(u + 1 + k) % KY][p]), "v"(c[p]));
          }
          // c >> 16 is 0 .. 256: saturate two at a time (v_sat_pk_u8_i16) instead of a v_min per pixel
          const uint32_t lo = sat_pk_u8_i16(__builtin_amdgcn_perm(c[1], c[0], 0x07060302u));  // [c0 >> 16, c1 >> 16] as i16
          const uint32_t hi = sat_pk_u8_i16(__builtin_amdgcn_perm(c[3], c[2], 0x07060302u));
          const uint32_t out = lo | (hi << 16);
          if (dword_out) {
            // a streaming store: the filtered plane is read again only after ~700 MB of other traffic, and the lines it
            // does not claim in the L2 are halo rows the tile below still finds there (-1 % on BASELINE config 3)
            __builtin_nontemporal_store(out, reinterpret_cast<uint32_t*>(d));
          } else {
#pragma unroll
            for (int p = 0; p < 4; p++)
              if (p < npx) d[p] = (uint8_t)(out >> (8 * p));
          }
          d += a.dstride;
        }
        lane_row += pitch;
      }
    }
  }
}

// One tile of one frame.  The staged rectangle starts at a 16-byte aligned source column and the plane is a whole
// number of 16-byte pieces wide, so a piece is inside its row or entirely outside it: outside pieces are loaded
// from somewhere harmless and overwritten by the border fix-up (replicate: a splat of the row's first / last
// byte), which only the tiles at the plane's left and right edge run.  Pieces are dealt to the 256 threads in
// linear order, all loads of a thread in flight together.
template <int KY, int ND, int KX>
__device__ __forceinline__ void lowpass_q8w_tile(const LowpassArgs& a, const LowpassTile& t, const SegmentDev& s,
                                                 uint32_t* __restrict__ box, const uint8_t* __restrict__ src,
                                                 uint8_t* __restrict__ dst) {
  constexpr int ry = KY >> 1;
  // staging slots per thread: ceil((32 + 6) rows x ceil((3 + W/4 + 10) / 4) pieces / 256 threads)
  constexpr int NS = ((kWideTileH + 6) * ((3 + kWideTileW / 4 + kWideMaxNd - 1 + 3) / 4) + 255) / 256;
  // (the segment record is workgroup-uniform; say so, or hipcc reads the taps with vector loads into VGPRs)
  const int rx = __builtin_amdgcn_readfirstlane(s.kx_len) >> 1;
  const uint32_t* __restrict__ kxs = a.taps_sh + __builtin_amdgcn_readfirstlane(s.kxs_off);
  const int* __restrict__ ky = a.taps_q8 + __builtin_amdgcn_readfirstlane(s.ky_off);
  uint32_t kyv[KY];
#pragma unroll
  for (int k = 0; k < KY; k++) kyv[k] = (uint32_t)ky[k];

  const int m = (t.x0 - rx) & 3;            // == (-rx) & 3: t.x0 is a multiple of 4
  const int dxw = (t.x0 - rx - m) >> 2;     // first dword column any lane's window needs (may be < 0)
  const int dx0 = dxw & ~3;                 // first staged dword column: 16-byte aligned
  const int lead = dxw - dx0;
  const int ndw = lead + ((t.w + 3) >> 2) + __builtin_amdgcn_readfirstlane(s.kxs_nd) - 1;
  const int pitch = (ndw + 3) & ~3, n4 = pitch >> 2;
  const int rows = t.h + 2 * ry;
  const int W4 = a.w >> 2;
  const int last = rows * n4 - 1;
  {
    uint4 v[NS];
    int loff[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const int i = min((int)threadIdx.x + 256 * k, last);  // surplus slots repeat the last piece
      const int r = i / n4, c4 = (i - r * n4) * 4;
      loff[k] = r * pitch + c4;
      v[k] = *reinterpret_cast<const uint4*>(src + (size_t)clampi(t.y0 - ry + r, 0, a.h - 1) * a.sstride +
                                             clampi(dx0 + c4, 0, W4 - 4) * 4);
    }
#pragma unroll
    for (int k = 0; k < NS; k++) *reinterpret_cast<uint4*>(box + loff[k]) = v[k];
  }
  __syncthreads();
  const int nleft = dx0 < 0 ? -dx0 : 0;    // staged dwords left of the plane
  const int first_right = W4 - dx0;        // first staged dword right of the plane
  if (nleft > 0 || first_right < pitch) {  // workgroup-uniform
    if ((int)threadIdx.x < rows) {
      uint32_t* __restrict__ row = box + (int)threadIdx.x * pitch;
      if (nleft > 0) {
        const uint32_t e = (row[nleft] & 0xffu) * 0x01010101u;
        for (int j = 0; j < nleft; j++) row[j] = e;
      }
      if (first_right < pitch) {
        const uint32_t e = (row[first_right - 1] >> 24) * 0x01010101u;
        for (int j = first_right; j < pitch; j++) row[j] = e;
      }
    }
    __syncthreads();
  }
  lowpass_q8w_rows<KY, ND, KX>(a, t, box + lead, pitch, kxs, kyv, dst);
}

// one wide tile `ti` of plane `a`, frame blockIdx.y
template <int KY>
__device__ __forceinline__ void lowpass_q8w_item(const LowpassArgs& a, int ti, uint32_t* __restrict__ box) {
  const LowpassTile t = a.wide_tiles[ti];
  const SegmentDev s = a.segs[t.seg];
  const uint8_t* __restrict__ src = a.src + (size_t)blockIdx.y * a.src_frame_bytes;
  uint8_t* __restrict__ dst = a.dst + (size_t)blockIdx.y * a.dst_frame_bytes;
  // instantiated per tap count for the short kernels, per window length for the rest
  const int nd = __builtin_amdgcn_readfirstlane(s.kxs_nd), kx = __builtin_amdgcn_readfirstlane(s.kx_len);
  switch (kx) {
    case 3: lowpass_q8w_tile<KY, 3, 3>(a, t, s, box, src, dst); break;
    case 5: lowpass_q8w_tile<KY, 3, 5>(a, t, s, box, src, dst); break;
    case 7: lowpass_q8w_tile<KY, 3, 7>(a, t, s, box, src, dst); break;
    case 9: lowpass_q8w_tile<KY, 3, 9>(a, t, s, box, src, dst); break;
    case 11: lowpass_q8w_tile<KY, 5, 11>(a, t, s, box, src, dst); break;
    case 13: lowpass_q8w_tile<KY, 5, 13>(a, t, s, box, src, dst); break;
    default:
      if (nd <= 5)
        lowpass_q8w_tile<KY, 5, 0>(a, t, s, box, src, dst);
      else if (nd <= 8)
        lowpass_q8w_tile<KY, 8, 0>(a, t, s, box, src, dst);
      else
        lowpass_q8w_tile<KY, kWideMaxNd, 0>(a, t, s, box, src, dst);
  }
}

template <int KY>
__global__ __launch_bounds__(256) void lowpass_q8w_kernel(LowpassArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_rows[];
  // XCD x (= id % 8) walks the x-th eighth of the row-major tile list
  const int per = (a.nwide + 7) >> 3;
  const int xcd = blockIdx.x & 7, ti = xcd * per + (int)(blockIdx.x >> 3);
  if (ti >= min((xcd + 1) * per, a.nwide)) return;
  lowpass_q8w_item<KY>(a, ti, reinterpret_cast<uint32_t*>(lds_rows));
}

// The planes of a batch (Y, U, V) in ONE launch: the merged tile list is plane 0's row-major list, then plane 1's, then
// plane 2's, and XCD x walks the x-th eighth of it.  (Round 5: three launches on three streams overlap as well, but each
// ramps up and drains on its own and the step pays two event joins; reference filterPlane is called once per plane,
// VideoFrameTransform.cpp:727-733 -- the planes are independent.)
template <int KY>
__global__ __launch_bounds__(256) void lowpass_q8w_multi_kernel(LowpassMulti m) {
  extern __shared__ __attribute__((aligned(16))) int lds_rows[];
  const int total = m.p[0].nwide + (m.n > 1 ? m.p[1].nwide : 0) + (m.n > 2 ? m.p[2].nwide : 0);
  const int per = (total + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  int ti = xcd * per + (int)(blockIdx.x >> 3);
  if (ti >= min((xcd + 1) * per, total)) return;
  // pick the plane with scalar selects (indexing m.p[] with a run-time index would copy the argument block to scratch)
  LowpassArgs a = m.p[0];
  if (m.n > 1 && ti >= a.nwide) {
    ti -= a.nwide;
    a = m.p[1];
    if (m.n > 2 && ti >= a.nwide) {
      ti -= a.nwide;
      a = m.p[2];
    }
  }
  lowpass_q8w_item<KY>(a, ti, reinterpret_cast<uint32_t*>(lds_rows));
}

}  // namespace

bool lowpass_mergeable(const LowpassArgs* a, int n) {
  if (n < 2 || n > 3) return false;
  for (int k = 0; k < n; k++)
    if (a[k].nwide <= 0 || a[k].nfast > 0 || a[k].ntiles > 0 || a[k].fast_ky != a[0].fast_ky) return false;
  return a[0].fast_ky == 3 || a[0].fast_ky == 5 || a[0].fast_ky == 7;
}

hipError_t launch_lowpass_multi(const LowpassArgs* a, int n, int nframes, hipStream_t stream) {
  if (nframes <= 0) return hipSuccess;
  if (!lowpass_mergeable(a, n)) return hipErrorInvalidValue;
  LowpassMulti m;
  m.n = n;
  int total = 0, lds = 0;
  for (int k = 0; k < 3; k++) {
    m.p[k] = a[k < n ? k : 0];
    if (k < n) {
      total += a[k].nwide;
      lds = a[k].wide_lds_bytes > lds ? a[k].wide_lds_bytes : lds;
    }
  }
  const dim3 grid(8 * ((total + 7) / 8), nframes, 1);
  switch (a[0].fast_ky) {
    case 3: hipLaunchKernelGGL(lowpass_q8w_multi_kernel<3>, grid, dim3(256), (size_t)lds, stream, m); break;
    case 5: hipLaunchKernelGGL(lowpass_q8w_multi_kernel<5>, grid, dim3(256), (size_t)lds, stream, m); break;
    default: hipLaunchKernelGGL(lowpass_q8w_multi_kernel<7>, grid, dim3(256), (size_t)lds, stream, m); break;
  }
  return hipGetLastError();
}

hipError_t launch_lowpass(const LowpassArgs& a, int nframes, hipStream_t stream) {
  if (nframes <= 0) return hipSuccess;
  if (a.nwide > 0) {
    const dim3 grid(8 * ((a.nwide + 7) / 8), nframes, 1);
    switch (a.fast_ky) {
      case 3: hipLaunchKernelGGL(lowpass_q8w_kernel<3>, grid, dim3(256), (size_t)a.wide_lds_bytes, stream, a); break;
      case 5: hipLaunchKernelGGL(lowpass_q8w_kernel<5>, grid, dim3(256), (size_t)a.wide_lds_bytes, stream, a); break;
      case 7: hipLaunchKernelGGL(lowpass_q8w_kernel<7>, grid, dim3(256), (size_t)a.wide_lds_bytes, stream, a); break;
      default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.nfast > 0) {
    const dim3 grid(a.nfast, nframes, 1);
    switch (a.fast_ky) {
      // LDS: (128 + 2*ry) rows of up to 32 + 17 dwords
      case 3: hipLaunchKernelGGL(lowpass_q8_kernel<3>, grid, dim3(256), (size_t)a.fast_lds_bytes, stream, a); break;
      case 5: hipLaunchKernelGGL(lowpass_q8_kernel<5>, grid, dim3(256), (size_t)a.fast_lds_bytes, stream, a); break;
      case 7: hipLaunchKernelGGL(lowpass_q8_kernel<7>, grid, dim3(256), (size_t)a.fast_lds_bytes, stream, a); break;
      default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.ntiles <= 0) return hipSuccess;
  const size_t lds = (size_t)a.max_rows * (size_t)a.tile_w * sizeof(int);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lowpass_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(lowpass_kernel, dim3(a.ntiles, nframes, 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace t360
