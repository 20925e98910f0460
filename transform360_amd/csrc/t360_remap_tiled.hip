// t360_remap_tiled.hip -- LDS-tiled bicubic gather over a batch of frames (the hot kernel).
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
//     through LDS with 16-byte coalesced loads (equirect rows are contiguous in HBM); the next
//     frame's chunks are already in flight in registers while the current frame is computed.
//     Taps that wrap across the +-180 degree seam or the poles are resolved while staging, so
//     the gather itself never wraps.
//   * the 4x4 stencil of one output pixel costs 4 ds_read2_b32 + 4 v_alignbit (unaligned 4-byte
//     row windows) and 8 v_dot4: weights are split into a signed high byte and an unsigned low
//     byte (w = 256*wh + wl) and pixels enter the high part as p-128,
//         SUM p*w = 256*SUM (p-128)*wh + SUM p*wl + 128*256*SUM wh,
//     all exact in int32, so results are bit-identical to the integer formulation.
//   * no MFMA: this is a gather, not a contraction.
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_kernels.h"
#include "t360_sample.h"

namespace t360 {

namespace {

__device__ __forceinline__ uint32_t bias128(uint32_t px4) { return px4 ^ 0x80808080u; }

// 16 bytes of one box row that straddle the plane edge: byte-wise with BORDER_WRAP in x
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
__device__ __forceinline__ void staged_tile(const TiledArgs& a, const TileDesc& t, uint8_t* __restrict__ lds,
                                            int f0, int f1) {
  const int tid = threadIdx.x;
  const int P = (int)t.cpr * kStageChunk;  // LDS row pitch in bytes
  const bool partial = (t.flags & kTilePartial) != 0;

  // ---- per-pixel geometry, once per tile ----
  int off[NPX];          // byte offset of the stencil's top-left tap inside the staged box
  uint32_t wh[NPX][4];   // signed high bytes of the 4x4 weights, one dword per stencil row
  uint32_t wl[NPX][4];   // unsigned low bytes
  int bias[NPX];         // 16384 + 128*256*SUM(wh)
  bool live[NPX];
  {
    uint32_t words[NPX];
    if (NPX == 4) {
      const uint4 v = reinterpret_cast<const uint4*>(a.tlut + t.tlut)[tid];
      words[0] = v.x;
      if (NPX > 1) {
        words[1 % NPX] = v.y;
        words[2 % NPX] = v.z;
        words[3 % NPX] = v.w;
      }
    } else {
      words[0] = a.tlut[t.tlut + tid];
    }
#pragma unroll
    for (int p = 0; p < NPX; p++) {
      const uint32_t e = words[p];
      live[p] = (e >> 31) == 0;
      const int rx = e & 1023, ry = (e >> 10) & 255, frac = (e >> 18) & 1023;
      off[p] = live[p] ? ry * P + rx : 0;
      const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(a.wpack + (size_t)frac * kCubicPackDwords);
      const uint4 h = wp[0], l = wp[1], c = wp[2];
      wh[p][0] = h.x; wh[p][1] = h.y; wh[p][2] = h.z; wh[p][3] = h.w;
      wl[p][0] = l.x; wl[p][1] = l.y; wl[p][2] = l.z; wl[p][3] = l.w;
      bias[p] = (int)c.x;
    }
  }

  // ---- staging assignments, once per tile: lane owns chunks tid, tid+256, ... of the box ----
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
      loff[c] = r * P + cc * kStageChunk;
      const int sy = wrap_coord(t.y0 + r, a.sh);
      int sx = t.x0 + cc * kStageChunk;
      if (sx + kStageChunk <= 0)
        sx += a.sw;
      else if (sx >= a.sw)
        sx -= a.sw;
      fast[c] = a.src_vec_ok && sx >= 0 && sx + kStageChunk <= a.sw && (sx & 15) == 0;
      goff[c] = fast[c] ? sy * a.sstride + sx : ((r << 16) | cc);
    }
  }

  uint4 stage[kStageChunksPerLane];
  auto fetch = [&](int f) {
    const uint8_t* __restrict__ base = a.src + (size_t)f * a.src_frame_bytes;
#pragma unroll
    for (int c = 0; c < kStageChunksPerLane; c++) {
      if (loff[c] < 0) continue;
      if (fast[c]) {
        stage[c] = *reinterpret_cast<const uint4*>(base + goff[c]);
      } else {
        const int r = goff[c] >> 16, cc = goff[c] & 0xffff;
        const int sy = wrap_coord(t.y0 + r, a.sh);
        stage[c] = fetch_wrapped(base + (size_t)sy * a.sstride, t.x0 + cc * kStageChunk, a.sw);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int c = 0; c < kStageChunksPerLane; c++)
      if (loff[c] >= 0) *reinterpret_cast<uint4*>(lds + loff[c]) = stage[c];
  };

  // output addressing
  int ox, oy;
  if (NPX == 4) {
    ox = t.ox + (tid & 7) * 4;
    oy = t.oy + (tid >> 3);
  } else {
    ox = t.ox + (tid & 15);
    oy = t.oy + (tid >> 4);
  }
  const size_t dpos = (size_t)oy * a.dstride + ox;

  fetch(f0);
  commit();
  __syncthreads();
  for (int f = f0; f < f1; f++) {
    if (f + 1 < f1) fetch(f + 1);  // in flight while this frame is computed

    int v[NPX];
#pragma unroll
    for (int p = 0; p < NPX; p++) {
      const int a4 = off[p] & ~3;
      const uint32_t sh = (uint32_t)(off[p] & 3) * 8u;
      int hi = 0;
      uint32_t lo = (uint32_t)bias[p];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(lds + a4 + r * P);
        const uint32_t px4 = __builtin_amdgcn_alignbit(q[1], q[0], sh);  // 4 consecutive source bytes
        hi = __builtin_amdgcn_sdot4((int)bias128(px4), (int)wh[p][r], hi, false);
        lo = __builtin_amdgcn_udot4(px4, wl[p][r], lo, false);
      }
      const int sum = (hi << 8) + (int)lo;  // = SUM p*w + 16384
      v[p] = sat_u8(sum >> kCoefBits);
    }

    uint8_t* __restrict__ d = a.dst + (size_t)f * a.dst_frame_bytes + dpos;
    if (NPX == 4) {
      if (!partial && a.dst_dword_ok) {
        *reinterpret_cast<uint32_t*>(d) =
            (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3 % NPX] << 24);
      } else {
#pragma unroll
        for (int p = 0; p < NPX; p++)
          if (live[p]) d[p] = (uint8_t)v[p];
      }
    } else {
      if (live[0]) d[0] = (uint8_t)v[0];
    }

    __syncthreads();  // everyone is done reading this frame's box
    if (f + 1 < f1) {
      commit();
      __syncthreads();
    }
  }
}

// tiles whose source box does not fit the staging budget (the four tiles around each pole)
__device__ __noinline__ void direct_tile(const TiledArgs& a, const TileDesc& t, int f0, int f1) {
  const int tid = threadIdx.x;
  const int ox = t.ox + (tid & 15), oy = t.oy + (tid >> 4);
  if (ox >= a.dw || oy >= a.dh) return;
  const LutEntry e = a.lut[(size_t)oy * a.dw + ox];
  for (int f = f0; f < f1; f++) {
    const int v = sample<4, false>(a.src + (size_t)f * a.src_frame_bytes, a.sw, a.sh, a.sstride, a.wtab, e);
    a.dst[(size_t)f * a.dst_frame_bytes + (size_t)oy * a.dstride + ox] = (uint8_t)v;
  }
}

__global__ __launch_bounds__(256) void remap_tiled_cubic_kernel(TiledArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // XCD-aware order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md); give every XCD one
  // contiguous range of the (raster-ordered) tile list so neighbouring tiles -- whose source
  // boxes overlap by the stencil halo -- share an L2.
  const int n = a.ntiles;
  const int b = blockIdx.x;
  const int xcd = b & 7, k = b >> 3;
  const int q = n >> 3, rem = n & 7;
  const int tile_index = xcd * q + (xcd < rem ? xcd : rem) + k;
  const TileDesc t = a.tiles[tile_index];
  const int f0 = blockIdx.y * a.frames_per_block;
  const int f1 = min(f0 + a.frames_per_block, a.nframes);
  if (t.kind == kTileStaged32)
    staged_tile<4>(a, t, lds, f0, f1);
  else if (t.kind == kTileStaged16)
    staged_tile<1>(a, t, lds, f0, f1);
  else
    direct_tile(a, t, f0, f1);
}

}  // namespace

hipError_t launch_remap_tiled_cubic(const TiledArgs& a, hipStream_t stream) {
  if (a.ntiles <= 0 || a.nframes <= 0) return hipSuccess;
  const int groups = (a.nframes + a.frames_per_block - 1) / a.frames_per_block;
  const size_t lds = (size_t)kStageMaxBytes + 64;  // +slack: the hi dword of the last row window
  hipLaunchKernelGGL(remap_tiled_cubic_kernel, dim3(a.ntiles, groups, 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace t360
