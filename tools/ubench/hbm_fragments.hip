// tools/ubench/hbm_fragments.hip -- how fast does HBM deliver a plane that is read as ROW FRAGMENTS of S bytes (rows of 3840
// bytes), every byte exactly once and every 128-byte line whole, as a function of S and of which fragments are in flight
// together?  (The gather's footprints are 40-70 rows of 130-260 bytes; tools/ubench/ldsdma_pattern.hip showed that the
// staging rate depends on the pattern, not on the number of frames in flight.)
//
// The buffer is 1.97 GB of 3840-byte rows.  A workgroup (8 waves, LDS-DMA pieces of 1 KiB = 1024 / S rows x S bytes, two
// "frames" of 8 pieces in flight, one barrier per frame: the gather's staging loop and nothing else) reads a column strip of
// S bytes, 512 KiB in all.  Workgroup -> strip:
//   row-major   neighbours in id are neighbours in the row (what is in flight together covers whole rows)
//   shuffled    ids are permuted: what is in flight together is scattered over the buffer
//   frames      the gather's own shape: the buffer is 64 frames of three 3840 x 1920 planes, a workgroup is a TILE whose
//               footprint (8 KiB: 512 / (S / 16) rows of S bytes) it reads from frame 0, 1, ... 63; neighbours in id are
//               neighbours in the row of tiles
//   overlap     as frames, but the tiles' columns are S / 2 apart: every line is read by TWO tiles, neighbours in id (on
//               different XCDs: no L2 between them) -- are second reads, microseconds after the first, cheaper?
//   overlap-xcd as overlap, but the two tiles of a line run on the SAME XCD (ids 8 apart): does its L2 serve the second read?
//   overlap-far the same tiles, even columns in the first half of the grid and odd columns in the second half: the second
//               read comes half a launch (~0.7 GB of traffic) after the first
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/hbmfrag hbm_fragments.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kPitch = 3840;
constexpr int kFrames = 64;  // iterations of 8 KiB per workgroup

template <int S>
__global__ __launch_bounds__(512) void strips(const uint8_t* src, int n_wg, int shuffled, unsigned* out) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned w = blockIdx.x;
  if (shuffled) w = (unsigned)(((unsigned long long)w * 2654435761ull + 12345ull) % (unsigned)n_wg);  // a bijection: the host picks n_wg coprime to the multiplier
  constexpr int strips_per_row = kPitch / S, cpr = S / 16;                 // chunks of 16 bytes per fragment
  constexpr long rows_per_wg = ((long)kFrames * 8 * 64 + cpr - 1) / cpr;  // the strip's chunks, row-major
  const long base = ((long)(w / strips_per_row) * rows_per_wg) * kPitch + (long)(w % strips_per_row) * S;
  constexpr int rows_per_tile = (512 + cpr - 1) / cpr, tile_rows = 1920 / rows_per_tile, tiles_per_plane = strips_per_row * tile_rows;
  const long plane_bytes = (long)kPitch * 1920;
  const unsigned t = blockIdx.x % tiles_per_plane, plane = blockIdx.x / tiles_per_plane;
  const int n_tile = wave * 64 + lane;
  long tile_off = plane * plane_bytes + ((long)(t / strips_per_row) * rows_per_tile + n_tile / cpr) * kPitch + (long)(t % strips_per_row) * S + (n_tile % cpr) * 16;
  if (shuffled >= 3) {
    // columns S / 2 apart, 2 * strips_per_row - 1 of them; the grid has 3 planes x tile_rows x cols tiles
    constexpr int cols = 2 * strips_per_row - 1;
    unsigned u = blockIdx.x;
    int c, r, pln;
    if (shuffled == 3) {
      c = u % cols; r = (u / cols) % tile_rows; pln = u / (cols * tile_rows);
    } else if (shuffled == 5) {
      // workgroup u runs on XCD u % 8: a row of tiles per XCD at a time, its columns 8 ids apart
      const unsigned xcd = u & 7, k = u >> 3;
      const unsigned R = (k / cols) * 8 + xcd;  // row of tiles over all three planes
      if (R >= 3u * tile_rows) return;
      c = k % cols; r = R % tile_rows; pln = R / tile_rows;
    } else {
      const unsigned n_even = 3u * tile_rows * strips_per_row;
      if (u < n_even) { c = 2 * (u % strips_per_row); r = (u / strips_per_row) % tile_rows; pln = u / (strips_per_row * tile_rows); }
      else { u -= n_even; c = 2 * (u % (strips_per_row - 1)) + 1; r = (u / (strips_per_row - 1)) % tile_rows; pln = u / ((strips_per_row - 1) * tile_rows); }
    }
    tile_off = pln * plane_bytes + ((long)r * rows_per_tile + n_tile / cpr) * kPitch + (long)c * (S / 2) + (n_tile % cpr) * 16;
  }
  auto issue = [&](int f, int slot) {
    const int n = (f * 8 + wave) * 64 + lane;  // chunk number in the strip: one piece = 64 consecutive chunks
    const uint8_t* a = shuffled >= 2 ? src + (long)f * 3 * plane_bytes + tile_off : src + base + (long)(n / cpr) * kPitch + (n % cpr) * 16;
    const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + slot * 8192 + wave * 1024);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(a), "s"(m0) : "memory");
  };
  issue(0, 0);
  issue(1, 1);
  unsigned acc = 0;
  for (int f = 0; f < kFrames; f++) {
    if (f + 1 < kFrames) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (f + 2 < kFrames) issue(f + 2, (f + 2) % 3);
    acc += lds[(f % 3) * 8192 + threadIdx.x * 4];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int S>
void launch(int n_wg, int shuffled, const uint8_t* src, unsigned* out) {
  hipLaunchKernelGGL(strips<S>, dim3(n_wg), dim3(512), 76 * 1024, 0, src, n_wg, shuffled, out);
}

static int gcd(long a, long b) { return b ? gcd(b, a % b) : (int)a; }

int main() {
  const int sizes[] = {128, 256, 384, 640, 1280, 3840};
  const long wg_bytes = (long)kFrames * 8 * 1024;
  uint8_t* src; unsigned* out;
  const long total = 3760L * wg_bytes;  // 3760 workgroups x 512 KiB = 1.97 GB; 3760 = 30 * 125 + 10 ...
  (void)hipMalloc(&src, total + (64 << 20)); (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 5, total + (64 << 20));
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int shuffled = 0; shuffled < 6; shuffled++)
    for (int S : sizes) {
      const int strips_per_row = kPitch / S;
      int n_wg = 3760 / strips_per_row * strips_per_row;  // whole rows of strips
      while (gcd(2654435761ll % n_wg, n_wg) != 1) n_wg -= strips_per_row;
      if (shuffled == 2) {
        const int cpr = S / 16, rows_per_tile = (512 + cpr - 1) / cpr;
        n_wg = 3 * strips_per_row * (1920 / rows_per_tile);
      }
      if (shuffled >= 3) {
        if (S > 1280) continue;
        const int cpr = S / 16, rows_per_tile = (512 + cpr - 1) / cpr;
        n_wg = 3 * (2 * strips_per_row - 1) * (1920 / rows_per_tile);
        if (shuffled == 5) n_wg = (3 * (1920 / rows_per_tile) + 7) / 8 * 8 * (2 * strips_per_row - 1);  // some exit at once; bytes below count the real ones
      }
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        switch (S) {
          case 128: launch<128>(n_wg, shuffled, src, out); break;
          case 256: launch<256>(n_wg, shuffled, src, out); break;
          case 384: launch<384>(n_wg, shuffled, src, out); break;
          case 640: launch<640>(n_wg, shuffled, src, out); break;
          case 1280: launch<1280>(n_wg, shuffled, src, out); break;
          default: launch<3840>(n_wg, shuffled, src, out); break;
        }
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      double bytes = (double)n_wg * wg_bytes;
      if (shuffled == 5) {
        const int cpr = S / 16, rows_per_tile = (512 + cpr - 1) / cpr;
        bytes = (double)(3 * (2 * strips_per_row - 1) * (1920 / rows_per_tile)) * wg_bytes;
      }
      printf("fragments of %4d bytes, %-11s: %5d workgroups, %.3f ms for %.0f MB -> %.2f TB/s (%.1f GB/s per CU)\n", S,
             shuffled == 5 ? "overlap-xcd" : shuffled == 4 ? "overlap-far" : shuffled == 3 ? "overlap" : shuffled == 2 ? "frames" : shuffled ? "shuffled" : "row-major", n_wg, best, bytes / 1e6, bytes / best / 1e9, bytes / best / 1e6 / 256);
    }
  return 0;
}
