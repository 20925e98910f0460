// tools/ubench/ldsdma_pattern.hip -- what does the gather's STAGING cost as a function of the address pattern of its LDS-DMA
// pieces?  512 workgroups of 8 waves (two per CU, like remap_tiled_kernel<4, 76, 8>), each staging P pieces of 1 KiB per
// "frame" for F frames into a ring of D + 1 slots (D frames in flight, counted s_waitcnt, one barrier per frame), nothing else.
// A piece = one global_load_lds_dwordx4 of a wave: 64 lanes x 16 bytes.  Patterns of the 64 chunk addresses:
//   contig    1 KiB contiguous
//   rows256   4 rows x 256 bytes, rows 3840 bytes apart            (an un-slanted footprint)
//   rows96    ~11 rows x 96 bytes (6 chunks), unaligned start      (a slanted footprint near a face edge)
//   rows32    32 rows x 32 bytes                                   (a polar footprint)
//   rows208   ~5 rows x 208 bytes at a drifting, unaligned start   (the average footprint row of the bicubic gather; round 5)
// Source: frames of 3840 x 1920 bytes, 64 of them (HBM), every workgroup its own region; or ONE frame (L2 / MALL resident).
// Reports GB/s per CU and over the chip.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsdma ldsdma_pattern.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  // at most 12 outstanding in this benchmark: a small switch
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int PPW, int D>  // pieces per wave and frame; frames in flight (ring of D + 1 slots)
__global__ __launch_bounds__(512) void stage(const uint8_t* src, long frame_bytes, int frames, int pattern, int same_frame, unsigned* out) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // this workgroup's region of the plane: 512 regions on a 32 x 16 grid of 120 x 120-byte cells ... keep it simple: rows
  const int wg = blockIdx.x;
  const long region = (long)(wg % 30) * 128 + (long)(wg / 30) * 100 * 3840;  // column block of 128 bytes, band of 100 rows
  int off[PPW];
#pragma unroll
  for (int p = 0; p < PPW; p++) {
    const int piece = wave * PPW + p;  // 0 .. 8 * PPW - 1
    int o;
    if (pattern == 0) o = piece * 1024 + lane * 16;                                          // contiguous
    else if (pattern == 1) o = (piece * 4 + lane / 16) * 3840 + (lane % 16) * 16;           // 4 rows x 256 B
    else if (pattern == 2) o = (piece * 11 + lane / 6) * 3840 + 40 + (lane % 6) * 16 + (lane / 6) * 16;  // ~11 rows x 96 B, drifting start
    else if (pattern == 3) o = (piece * 32 + lane / 2) * 3840 + (lane % 2) * 16 + (lane / 2 % 4) * 32;        // 32 rows x 32 B
    else o = (piece * 5 + lane / 13) * 3840 + 24 + (lane % 13) * 16 + (lane / 13) * 48;      // ~5 rows x 208 B, drifting unaligned start (the gather; round 5)
    off[p] = (int)region + o;
  }
  const unsigned slot_bytes = 8 * PPW * 1024;
  auto issue = [&](int f, int slot) {
    const uint8_t* base = src + (same_frame ? 0 : (long)f * frame_bytes);
    const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + slot * slot_bytes + wave * PPW * 1024);
#pragma unroll
    for (int p = 0; p < PPW; p++)
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off[p]), "s"(base), "s"(__builtin_amdgcn_readfirstlane(m0 + p * 1024)) : "memory");
  };
  for (int f = 0; f < D && f < frames; f++) issue(f, f);
  unsigned acc = 0;
  for (int f = 0; f < frames; f++) {
    wait_vmcnt_dyn(f + D - 1 < frames ? (D - 1) * PPW : 0);  // the tail drains early: D - 1 frames of 64
    asm volatile("s_barrier" ::: "memory");
    if (f + D < frames) issue(f + D, (f + D) % (D + 1));
    acc += lds[(f % (D + 1)) * slot_bytes + threadIdx.x * 4];  // touch the slot (one LDS read per lane)
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const long frame_bytes = 3840L * 1920;
  const int F = 64;
  uint8_t* src; unsigned* out;
  (void)hipMalloc(&src, frame_bytes * F + (4 << 20)); (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 3, frame_bytes * F + (4 << 20));
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const char* names[] = {"contig", "rows256", "rows96", "rows32", "rows208"};
  // 76 KiB of LDS per workgroup whatever the ring needs: two workgroups per CU, like the gather
  const size_t lds_bytes = 76 * 1024;
  for (int same = 0; same < 2; same++)
    for (int cfg = 0; cfg < 8; cfg++)
      for (int pat = 0; pat < 5; pat++) {
        float best = 1e9f;
        static const int ppws[] = {1, 1, 1, 2, 2, 1, 2, 1}, ds[] = {2, 4, 8, 2, 3, 3, 4, 6};
        const int ppw = ppws[cfg], d = ds[cfg];
        for (int rep = 0; rep < 3; rep++) {
          (void)hipEventRecord(a);
          for (int k = 0; k < 4; k++) {  // 4 "generations" of 512 workgroups
            switch (cfg) {
              case 0: hipLaunchKernelGGL((stage<1, 2>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              case 1: hipLaunchKernelGGL((stage<1, 4>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              case 2: hipLaunchKernelGGL((stage<1, 8>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              case 3: hipLaunchKernelGGL((stage<2, 2>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              case 4: hipLaunchKernelGGL((stage<2, 3>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              case 5: hipLaunchKernelGGL((stage<1, 3>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              case 6: hipLaunchKernelGGL((stage<2, 4>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
              default: hipLaunchKernelGGL((stage<1, 6>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, pat, same, out); break;
            }
          }
          (void)hipEventRecord(b); (void)hipEventSynchronize(b);
          float ms; (void)hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        const double bytes = 4.0 * 512 * F * 8 * ppw * 1024;
        printf("%-8s %d KiB per frame, %d frames in flight, %s: %.3f ms for %.0f MB -> %.2f TB/s, %.1f GB/s per CU, %.2f us per frame\n", names[pat],
               8 * ppw, d, same ? "one frame (cached)" : "64 frames (HBM)  ", best, bytes / 1e6, bytes / best / 1e9, bytes / best / 1e6 / 256,
               best * 1e3 / 4 / F);
      }
  return 0;
}
