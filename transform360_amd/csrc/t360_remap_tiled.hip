// t360_remap_tiled.hip -- LDS-tiled bicubic gather over a batch of frames (the hot kernels).
//
// Same arithmetic as t360_remap.hip (cv::remap INTER_CUBIC, BORDER_WRAP, Q15 weights,
// (sum + 16384) >> 15), organised for MI355X:
//
//   * one 256-lane workgroup owns one OUTPUT tile (32x32 px, 4 px per lane; 16x16 px, 1 px per
//     lane near the poles) and walks `frames_per_block` frames of the batch with it.  Everything
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

#include "t360_internal.h"
#include "t360_kernels.h"
#include "t360_sample.h"

namespace t360 {

namespace {

constexpr int kRingMaxSlots = 8;

__device__ __forceinline__ uint32_t bias128(uint32_t px4) { return px4 ^ 0x80808080u; }

// ---- per-pixel geometry shared by both staging variants --------------------------------------
template <int NPX>
struct PixelSetup {
  int off[NPX];         // byte offset of the stencil's top-left tap inside the staged box
  uint32_t wh[NPX][4];  // signed high bytes of the 4x4 weights, one dword per stencil row
  uint32_t wl[NPX][4];  // unsigned low bytes
  int bias[NPX];        // 16384 + 128*256*SUM(wh)
  bool live[NPX];       // pixel inside the plane (partial tiles)
};

template <int NPX>
__device__ __forceinline__ void load_pixels(const TiledPlane& pl, const uint32_t* __restrict__ wpack,
                                            const TileDesc& t, int pitch, PixelSetup<NPX>& s) {
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
    const int rx = e & 1023, ry = (e >> 10) & 255, frac = (e >> 18) & 1023;
    s.off[p] = s.live[p] ? ry * pitch + rx : 0;
    const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(wpack + (size_t)frac * kCubicPackDwords);
    const uint4 h = wp[0], l = wp[1], c = wp[2];
    s.wh[p][0] = h.x; s.wh[p][1] = h.y; s.wh[p][2] = h.z; s.wh[p][3] = h.w;
    s.wl[p][0] = l.x; s.wl[p][1] = l.y; s.wl[p][2] = l.z; s.wl[p][3] = l.w;
    s.bias[p] = (int)c.x;
  }
}

// Make hipcc wait for its own (counted) loads HERE: every loaded value passes through an empty
// asm, so the compiler-inserted s_waitcnt lands before it and not in front of the first use
// inside the frame loop, where it would also drain the DMA ring.
template <int NPX>
__device__ __forceinline__ void pin_pixels(PixelSetup<NPX>& s) {
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    asm volatile("" : "+v"(s.off[p]), "+v"(s.bias[p]));
#pragma unroll
    for (int r = 0; r < 4; r++) asm volatile("" : "+v"(s.wh[p][r]), "+v"(s.wl[p][r]));
  }
}

// one frame of one tile: gather from the staged box at `box`, write the output pixels
template <int NPX>
__device__ __forceinline__ void gather_store(const PixelSetup<NPX>& s, const uint8_t* __restrict__ box, int pitch,
                                             uint8_t* __restrict__ d, bool dword_store) {
  int v[NPX];
#pragma unroll
  for (int p = 0; p < NPX; p++) {
    const int a4 = s.off[p] & ~3;
    const uint32_t sh = (uint32_t)(s.off[p] & 3) * 8u;
    int hi = 0;
    uint32_t lo = (uint32_t)s.bias[p];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(box + a4 + r * pitch);
      const uint32_t px4 = __builtin_amdgcn_alignbit(q[1], q[0], sh);  // 4 consecutive source bytes
      hi = __builtin_amdgcn_sdot4((int)bias128(px4), (int)s.wh[p][r], hi, false);
      lo = __builtin_amdgcn_udot4(px4, s.wl[p][r], lo, false);
    }
    const int sum = (hi << 8) + (int)lo;  // = SUM p*w + 16384
    v[p] = sat_u8(sum >> kCoefBits);
  }
  if (NPX == 4) {
    if (dword_store) {
      *reinterpret_cast<uint32_t*>(d) =
          (uint32_t)v[0] | ((uint32_t)v[1 % NPX] << 8) | ((uint32_t)v[2 % NPX] << 16) | ((uint32_t)v[3 % NPX] << 24);
    } else {
#pragma unroll
      for (int p = 0; p < NPX; p++)
        if (s.live[p]) d[p] = (uint8_t)v[p];
    }
  } else {
    if (s.live[0]) d[0] = (uint8_t)v[0];
  }
}

template <int NPX>
__device__ __forceinline__ size_t out_pos(const TiledPlane& pl, const TileDesc& t) {
  const int tid = threadIdx.x;
  int ox, oy;
  if (NPX == 4) {
    ox = t.ox + (tid & 7) * 4;
    oy = t.oy + (tid >> 3);
  } else {
    ox = t.ox + (tid & 15);
    oy = t.oy + (tid >> 4);
  }
  return (size_t)oy * pl.dstride + ox;
}

// tiles whose source box does not fit the staging budget (the four tiles around each pole).
// Arguments by value: taking the address of the plane descriptor would force it into scratch.
__device__ __noinline__ void direct_tile(const uint8_t* __restrict__ src, int64_t src_frame_bytes, int sw, int sh,
                                         int sstride, uint8_t* __restrict__ dst, int64_t dst_frame_bytes, int dw,
                                         int dh, int dstride, const LutEntry* __restrict__ lut,
                                         const int16_t* __restrict__ wtab, int tox, int toy, int f0, int f1) {
  const int tid = threadIdx.x;
  const int ox = tox + (tid & 15), oy = toy + (tid >> 4);
  if (ox >= dw || oy >= dh) return;
  const LutEntry e = lut[(size_t)oy * dw + ox];
  for (int f = f0; f < f1; f++) {
    const int v = sample<4, false>(src + (size_t)f * src_frame_bytes, sw, sh, sstride, wtab, e);
    dst[(size_t)f * dst_frame_bytes + (size_t)oy * dstride + ox] = (uint8_t)v;
  }
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

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
#define T360_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    T360_W(0) T360_W(1) T360_W(2) T360_W(3) T360_W(4) T360_W(5) T360_W(6) T360_W(7)
    T360_W(8) T360_W(9) T360_W(10) T360_W(11) T360_W(12) T360_W(13) T360_W(14) T360_W(15)
    T360_W(16) T360_W(17) T360_W(18) T360_W(19) T360_W(20) T360_W(21) T360_W(22) T360_W(23)
    T360_W(24) T360_W(25) T360_W(26) T360_W(27) T360_W(28)
#undef T360_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// 64 lanes x 16 bytes global -> LDS; lane i lands at lds_dst + 16*i.  hipcc does not count this
// load (cdna_hip_programming.md 5.7): completion is ours to track with wait_vmcnt().
__device__ __forceinline__ void dma16(const uint8_t* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int NPX>
__device__ __forceinline__ void staged_tile_dma(const TiledArgs& a, const TiledPlane& pl, const TileDesc& t,
                                                uint8_t* __restrict__ lds, int f0, int f1) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pitch = (int)t.cpr * kStageChunk;
  const int nch = (int)t.cpr * (int)t.rows;
  const int slot_bytes = (nch * kStageChunk + 16 + 63) & ~63;  // +16: hi dword of the last row window
  int K = a.ring_bytes / slot_bytes;
  K = K > kRingMaxSlots ? kRingMaxSlots : K;  // the plan guarantees K >= 2

  PixelSetup<NPX> px;
  load_pixels<NPX>(pl, a.wpack, t, pitch, px);

  // staging assignments: lane owns chunks tid, tid+256, ... ; chunk q lives at LDS byte 16*q
  int goff[kStageChunksPerLane];
  int my_nld = 0;  // DMA instructions THIS WAVE issues per frame (wave-uniform)
#pragma unroll
  for (int c = 0; c < kStageChunksPerLane; c++) {
    const int q = tid + c * 256;
    goff[c] = -1;
    if (q < nch) {
      const int r = q / (int)t.cpr, cc = q - r * (int)t.cpr;
      const int sy = wrap_coord(t.y0 + r, pl.sh);
      int sx = t.x0 + cc * kStageChunk;  // multiple of 16; plane width is a multiple of 16 here
      if (sx < 0)
        sx += pl.sw;
      else if (sx >= pl.sw)
        sx -= pl.sw;
      goff[c] = sy * pl.sstride + sx;
    }
    if (c * 256 + wave * 64 < nch) my_nld++;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;  // LDS byte address of the ring

  auto issue = [&](int f, int slot) {
    const uint8_t* __restrict__ base = pl.src + (size_t)f * pl.src_frame_bytes;
    const uint32_t sbase = lds_base + (uint32_t)(slot * slot_bytes + wave * 64 * kStageChunk);
#pragma unroll
    for (int c = 0; c < kStageChunksPerLane; c++) {
      if (c * 256 + wave * 64 < nch) {  // wave-uniform: the instruction count per wave is exact
        if (goff[c] >= 0)
          dma16(base + goff[c], (uint32_t)__builtin_amdgcn_readfirstlane((int)(sbase + (uint32_t)(c * 256 * kStageChunk))));
      }
    }
  };

  const size_t dpos = out_pos<NPX>(pl, t);
  const bool dword_store = NPX == 4 && !(t.flags & kTilePartial) && pl.dst_dword_ok;

  // hipcc's own loads (LUT words, weights) must not sit in the queue behind the DMA
  pin_pixels<NPX>(px);

  const int nf = f1 - f0;
  for (int j = 0; j < K - 1 && j < nf; j++) issue(f0 + j, j);
  int slot = 0, fill = (K - 1) % K;
  for (int i = 0; i < nf; i++) {
    // DMA loads of this wave younger than frame i's: frames i+1 .. min(i+K-2, nf-1)
    const int younger = min(K - 2, nf - 1 - i);
    wait_vmcnt(younger * my_nld);
    // frame i's box is complete for every wave; everyone left frame i-1's slot.  A bare
    // s_barrier: __syncthreads() would add fences whose waits could drain the DMA ring.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + K - 1 < nf) issue(f0 + i + K - 1, fill);
    gather_store<NPX>(px, lds + slot * slot_bytes, pitch,
                      pl.dst + (size_t)(f0 + i) * pl.dst_frame_bytes + dpos, dword_store);
    slot = slot + 1 == K ? 0 : slot + 1;
    fill = fill + 1 == K ? 0 : fill + 1;
  }
}

__global__ __launch_bounds__(256) void remap_tiled_cubic_dma_kernel(TiledArgs a) {
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
  if (t.kind == kTileStaged32)
    staged_tile_dma<4>(a, pl, t, lds, f0, f1);
  else if (t.kind == kTileStaged16)
    staged_tile_dma<1>(a, pl, t, lds, f0, f1);
  else
    direct_tile(pl.src, pl.src_frame_bytes, pl.sw, pl.sh, pl.sstride, pl.dst, pl.dst_frame_bytes, pl.dw, pl.dh,
                pl.dstride, pl.lut, a.wtab, t.ox, t.oy, f0, f1);
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
  PixelSetup<NPX> px;
  load_pixels<NPX>(pl, a.wpack, t, pitch, px);

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
      const int r = q / (int)t.cpr, cc = q - r * (int)t.cpr;
      loff[c] = q * kStageChunk;
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

  const size_t dpos = out_pos<NPX>(pl, t);
  const bool dword_store = NPX == 4 && !(t.flags & kTilePartial) && pl.dst_dword_ok;
  fetch(f0);
  commit();
  __syncthreads();
  for (int f = f0; f < f1; f++) {
    if (f + 1 < f1) fetch(f + 1);  // in flight while this frame is computed
    gather_store<NPX>(px, lds, pitch, pl.dst + (size_t)f * pl.dst_frame_bytes + dpos, dword_store);
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
  if (t.kind == kTileStaged32)
    staged_tile_regs<4>(a, pl, t, lds, f0, f1);
  else if (t.kind == kTileStaged16)
    staged_tile_regs<1>(a, pl, t, lds, f0, f1);
  else
    direct_tile(pl.src, pl.src_frame_bytes, pl.sw, pl.sh, pl.sstride, pl.dst, pl.dst_frame_bytes, pl.dw, pl.dh,
                pl.dstride, pl.lut, a.wtab, t.ox, t.oy, f0, f1);
}

}  // namespace

hipError_t launch_remap_tiled_cubic_dma(const TiledArgs& a, hipStream_t stream) {
  if (a.total_tiles <= 0 || a.nframes <= 0) return hipSuccess;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  static int configured_lds = 0;
  if (a.ring_bytes > 64 * 1024 && configured_lds < a.ring_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(remap_tiled_cubic_dma_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, a.ring_bytes);
    if (e != hipSuccess) return e;
    configured_lds = a.ring_bytes;
  }
  hipLaunchKernelGGL(remap_tiled_cubic_dma_kernel, dim3(a.total_tiles, groups, 1), dim3(256), (size_t)a.ring_bytes,
                     stream, a);
  return hipGetLastError();
}

hipError_t launch_remap_tiled_cubic_regs(const TiledArgs& a, hipStream_t stream) {
  if (a.plane[0].ntiles <= 0 || a.nframes <= 0) return hipSuccess;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  const size_t lds = (size_t)kStageMaxBytes + 64;
  hipLaunchKernelGGL(remap_tiled_cubic_regs_kernel, dim3(a.plane[0].ntiles, groups, 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace t360
