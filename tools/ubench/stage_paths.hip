// tools/ubench/stage_paths.hip -- is the rate at which a CU stages ragged row fragments from HBM a limit of the LDS-DMA path
// (global_load_lds_dwordx4) or of the CU's vector memory path as a whole?  The staging loop of the gather (512 workgroups of 8
// waves, two per CU, PPW pieces of 1 KiB per wave and frame, D = 2 frames in flight, one barrier per frame), with the pieces moved
//   mode 0: all by LDS-DMA                                  (what remap_tiled_kernel does)
//   mode 1: all through VGPRs (global_load_dwordx4, then ds_write_b128)
//   mode 2: waves 0-3 by LDS-DMA, waves 4-7 through VGPRs   (same bytes; if the DMA path had a cap of its own, this would beat 0)
// Address pattern of a piece: ~5 rows x 208 bytes at a drifting, unaligned start (the gather's footprint rows are 130-260 bytes).
// Reports GB/s per CU from HBM (64 frames) and from one cached frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int PPW, int MODE>
__global__ __launch_bounds__(512) void stage(const uint8_t* src, long frame_bytes, int frames, int same_frame, unsigned* out) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = blockIdx.x;
  const long region = (long)(wg % 16) * 224 + (long)(wg / 16) * 48 * 3840;
  int off[PPW];
#pragma unroll
  for (int p = 0; p < PPW; p++) {
    const int piece = wave * PPW + p;
    off[p] = (int)region + (piece * 5 + lane / 13) * 3840 + 24 + (lane % 13) * 16 + (lane / 13) * 48;
  }
  const bool dma = MODE == 0 || (MODE == 2 && wave < 4);
  const unsigned slot_bytes = 8 * PPW * 1024;
  const unsigned my = (unsigned)(uintptr_t)lds + wave * PPW * 1024;
  unsigned acc = 0;
  if (dma) {
    auto issue = [&](int f, int slot) {
      const uint8_t* base = src + (same_frame ? 0 : (long)f * frame_bytes);
#pragma unroll
      for (int p = 0; p < PPW; p++)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off[p]), "s"(base), "s"(__builtin_amdgcn_readfirstlane(my + slot * slot_bytes + p * 1024)) : "memory");
    };
    issue(0, 0);
    if (frames > 1) issue(1, 1);
    for (int f = 0; f < frames; f++) {
      if (f + 1 < frames) { if (PPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (f + 2 < frames) issue(f + 2, (f + 2) % 3);
      acc += lds[(f % 3) * slot_bytes + threadIdx.x * 4];
    }
  } else {
    // two register sets (frames f and f + 1 in flight), the frame loop unrolled by two
    uint4 ra[PPW], rb[PPW];
    auto load = [&](uint4 (&r)[PPW], int f) {
      const uint8_t* base = src + (same_frame ? 0 : (long)f * frame_bytes);
#pragma unroll
      for (int p = 0; p < PPW; p++) r[p] = *reinterpret_cast<const uint4*>(base + off[p]);
    };
    auto store = [&](const uint4 (&r)[PPW], int slot) {
#pragma unroll
      for (int p = 0; p < PPW; p++) *reinterpret_cast<uint4*>(lds + wave * PPW * 1024 + slot * slot_bytes + p * 1024 + lane * 16) = r[p];
    };
    load(ra, 0);
    if (frames > 1) load(rb, 1);
    for (int f = 0; f < frames; f += 2) {
      store(ra, f % 3);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (f + 2 < frames) load(ra, f + 2);
      acc += lds[(f % 3) * slot_bytes + threadIdx.x * 4];
      if (f + 1 < frames) {
        store(rb, (f + 1) % 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (f + 3 < frames) load(rb, f + 3);
        acc += lds[((f + 1) % 3) * slot_bytes + threadIdx.x * 4];
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const long frame_bytes = 3840L * 1920;
  const int F = 64;
  uint8_t* src; unsigned* out;
  (void)hipMalloc(&src, frame_bytes * F + (8 << 20)); (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 3, frame_bytes * F + (8 << 20));
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const size_t lds_bytes = 76 * 1024;
  for (int same = 0; same < 2; same++)
    for (int ppw = 1; ppw <= 2; ppw++)
      for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
          (void)hipEventRecord(a);
          for (int k = 0; k < 4; k++) {
            if (ppw == 1) {
              if (mode == 0) hipLaunchKernelGGL((stage<1, 0>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, same, out);
              else if (mode == 1) hipLaunchKernelGGL((stage<1, 1>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, same, out);
              else hipLaunchKernelGGL((stage<1, 2>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, same, out);
            } else {
              if (mode == 0) hipLaunchKernelGGL((stage<2, 0>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, same, out);
              else if (mode == 1) hipLaunchKernelGGL((stage<2, 1>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, same, out);
              else hipLaunchKernelGGL((stage<2, 2>), dim3(512), dim3(512), lds_bytes, 0, src, frame_bytes, F, same, out);
            }
          }
          (void)hipEventRecord(b); (void)hipEventSynchronize(b);
          float ms; (void)hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        const double bytes = 4.0 * 512 * F * 8 * ppw * 1024;
        static const char* names[] = {"all LDS-DMA", "all through VGPRs", "4 waves DMA + 4 waves VGPRs"};
        printf("%d KiB per frame, %-28s %s: %.3f ms -> %.2f TB/s, %.1f GB/s per CU, %.2f us per frame\n", 8 * ppw, names[mode],
               same ? "one frame (cached)" : "64 frames (HBM)  ", best, bytes / best / 1e9, bytes / best / 1e6 / 256, best * 1e3 / 4 / F);
      }
  return 0;
}
