// tools/ubench/smem_stream.hip -- round 6, verdict item 1(b): is the SCALAR memory path (s_load_dwordx16 -> SGPRs ->
// v_writelane x16 -> one 16-lane ds_write_b32) a second staging channel beside the vector path?  Two questions:
//   1. how many GB/s per CU does it stream from HBM on its own (all 16 waves of a CU, 1 or 2 loads of 64 bytes in flight per
//      wave and "frame", double-buffered: 32 or 64 of a wave's ~100 SGPRs), discarding the data or moving it into LDS;
//   2. is it ADDITIVE beside the gather's LDS-DMA stream (the rows208 pattern of ldsdma_pattern.hip: ~5 rows x 208 bytes per
//      1 KiB instruction, 8 KiB per workgroup and frame, two frames in flight, one barrier per frame)?
// Geometry of the gather: 512 workgroups of 8 waves, two per CU, 76 KiB of LDS each, 64 "frames" per workgroup.
// Modes: 0 DMA only | 1 SMEM only, discard | 2 SMEM only, into LDS | 3 DMA + SMEM discard | 4 DMA + SMEM into LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o smem_stream.bin smem_stream.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u32x16 sload16(const uint8_t* p) {
  u32x16 v;
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// 64 bytes out of SGPRs into LDS at `dst` (wave-uniform): 16 v_writelane + one ds_write_b32 of 16 lanes
__device__ __forceinline__ void to_lds(const u32x16& s, uint8_t* lds, unsigned dst, int lane) {
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s[k]), "n"(k));
  if (lane < 16) *reinterpret_cast<uint32_t*>(lds + dst + lane * 4) = v;
}

template <int MODE, int NS>  // NS = scalar loads of 64 bytes per wave and frame (1 or 2)
__global__ __launch_bounds__(512) void stage(const uint8_t* src, const uint8_t* src2, long frame_bytes, long sframe_bytes, int frames, unsigned* out) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  constexpr bool DMA = MODE == 0 || MODE >= 3, SM = MODE >= 1, TOLDS = MODE == 2 || MODE == 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = blockIdx.x;
  // vector side: the rows208 pattern of ldsdma_pattern.hip, one 1 KiB piece per wave and frame
  const long region = (long)(wg % 16) * 224 + (long)(wg / 16) * 48 * 3840;
  const int off = (int)region + (wave * 5 + lane / 13) * 3840 + 24 + (lane % 13) * 16 + (lane / 13) * 48;
  // scalar side: a second buffer, every wave its own 256-byte stripe of a 1 MiB slot, frames sframe_bytes apart; every launch of
  // the measurement reads its own slot (src2 is advanced by the host), so nothing comes from the Infinity Cache
  const uint8_t* sp = src2 + ((long)wg * 8 + wave) * 256;
  const unsigned slot_bytes = 8 * 1024 + 8 * NS * 64;
  const unsigned my = (unsigned)(uintptr_t)lds + wave * 1024;
  auto issue = [&](int f, int slot) {
    const uint8_t* base = src + (long)f * frame_bytes;
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(__builtin_amdgcn_readfirstlane(my + slot * slot_bytes)) : "memory");
  };
  unsigned acc = 0, sacc = 0;
  u32x16 a0, a1, b0, b1;
  if (DMA) {
    issue(0, 0);
    if (frames > 1) issue(1, 1);
  }
  if (SM) {
    a0 = sload16(sp);
    if (NS > 1) a1 = sload16(sp + 64);
  }
  for (int f = 0; f < frames; f += 2) {
    // ---- even frame: consume set a, refill set b ----
#define HALF(F, CUR0, CUR1, NXT0, NXT1)                                                                        \
    if ((F) < frames) {                                                                                        \
      if (SM) {                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        asm volatile("" : "+s"(CUR0));                                                                         \
        if (NS > 1) asm volatile("" : "+s"(CUR1));                                                             \
        if ((F) + 1 < frames) {                                                                                \
          NXT0 = sload16(sp + (long)((F) + 1) * sframe_bytes);                                                 \
          if (NS > 1) NXT1 = sload16(sp + (long)((F) + 1) * sframe_bytes + 64);                                 \
        }                                                                                                      \
      }                                                                                                        \
      if (DMA) {                                                                                               \
        if ((F) + 1 < frames) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
        asm volatile("s_barrier" ::: "memory");                                                                \
        if ((F) + 2 < frames) issue((F) + 2, ((F) + 2) % 3);                                                   \
        acc += lds[((F) % 3) * slot_bytes + threadIdx.x * 4];                                                  \
      }                                                                                                        \
      if (SM) {                                                                                                \
        if (TOLDS) {                                                                                           \
          to_lds(CUR0, lds, ((F) % 3) * slot_bytes + 8192 + wave * NS * 64, lane);                             \
          if (NS > 1) to_lds(CUR1, lds, ((F) % 3) * slot_bytes + 8192 + wave * NS * 64 + 64, lane);            \
        } else {                                                                                               \
          sacc ^= CUR0[0] ^ CUR0[15];                                                                          \
          if (NS > 1) sacc ^= CUR1[0] ^ CUR1[15];                                                              \
        }                                                                                                      \
      }                                                                                                        \
    }
    HALF(f, a0, a1, b0, b1)
    HALF(f + 1, b0, b1, a0, a1)
#undef HALF
  }
  if (TOLDS) acc += lds[8192 + threadIdx.x];
  if (acc + sacc == 0x12345678u) out[0] = acc;
}

template <int MODE, int NS>
float run(const uint8_t* src, const uint8_t* src2, long fb, int F, unsigned* out) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const size_t lds_bytes = 76 * 1024;
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(a);
    for (int k = 0; k < 4; k++) hipLaunchKernelGGL((stage<MODE, NS>), dim3(512), dim3(512), lds_bytes, 0, src, src2 + (long)(rep * 4 + k) * (1 << 20), fb, 12L << 20, F, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const long fb = 3840L * 1920;
  const int F = 64;
  uint8_t *src, *src2; unsigned* out;
  (void)hipMalloc(&src, fb * F + (4 << 20)); (void)hipMalloc(&src2, (12L << 20) * F); (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 3, fb * F + (4 << 20)); (void)hipMemset(src2, 5, (12L << 20) * F);
  const double dma_bytes = 4.0 * 512 * F * 8 * 1024;
  auto rep = [&](const char* name, float ms, bool dma, int ns) {
    const double sb = 4.0 * 512 * F * 8 * ns * 64;
    printf("%-34s %.3f ms", name, ms);
    if (dma) printf("  DMA %.1f GB/s per CU", dma_bytes / ms / 1e6 / 256);
    if (ns) printf("  SMEM %.2f GB/s per CU (%d x 64 B per wave and frame)", sb / ms / 1e6 / 256, ns);
    printf("\n");
  };
  rep("0 DMA only", run<0, 1>(src, src2, fb, F, out), true, 0);
  rep("1 SMEM only, discard, 1 load", run<1, 1>(src, src2, fb, F, out), false, 1);
  rep("1 SMEM only, discard, 2 loads", run<1, 2>(src, src2, fb, F, out), false, 2);
  rep("2 SMEM only, into LDS, 1 load", run<2, 1>(src, src2, fb, F, out), false, 1);
  rep("2 SMEM only, into LDS, 2 loads", run<2, 2>(src, src2, fb, F, out), false, 2);
  rep("3 DMA + SMEM discard, 1 load", run<3, 1>(src, src2, fb, F, out), true, 1);
  rep("3 DMA + SMEM discard, 2 loads", run<3, 2>(src, src2, fb, F, out), true, 2);
  rep("4 DMA + SMEM into LDS, 1 load", run<4, 1>(src, src2, fb, F, out), true, 1);
  rep("4 DMA + SMEM into LDS, 2 loads", run<4, 2>(src, src2, fb, F, out), true, 2);
  rep("0 DMA only (again)", run<0, 1>(src, src2, fb, F, out), true, 0);
  return 0;
}
