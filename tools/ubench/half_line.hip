// micro-benchmark: does any cache policy make the L2 fetch HALF a 128-byte line from the fabric?
// Every kernel touches the first 64 bytes of every 128-byte line of a 1 GiB buffer (512 MiB asked for); `full`
// reads whole lines.  If a variant's time is ~half of `full`'s, its misses cost 64-byte fabric reads.  Under
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- ./half_line.bin
// the request sizes can be read per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LD(policy)                                                                                    \
  asm volatile("global_load_dwordx4 %0, %1, off " policy "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory")

// P: 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc1 nt, 7 = whole lines (plain)
template <int P>
__global__ __launch_bounds__(256) void touch(const uint8_t* __restrict__ src, size_t lines, unsigned* out) {
  unsigned acc = 0;
  const size_t nthreads = (size_t)gridDim.x * 256, tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (P == 7) {
    for (size_t i = tid; i < lines * 8; i += nthreads) {
      const uint8_t* p = src + i * 16;
      uint4 v;
      LD("");
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  } else {
    for (size_t i = tid; i < lines * 4; i += nthreads) {
      const uint8_t* p = src + (i >> 2) * 128 + (i & 3) * 16;
      uint4 v;
      if (P == 0) LD("");
      if (P == 1) LD("nt");
      if (P == 2) LD("sc0");
      if (P == 3) LD("sc1");
      if (P == 4) LD("sc0 sc1");
      if (P == 5) LD("sc0 sc1 nt");
      if (P == 6) LD("sc1 nt");
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// the same through LDS-DMA (what the gather's staging uses)
template <int P>
__global__ __launch_bounds__(256) void touch_dma(const uint8_t* __restrict__ src, size_t lines, unsigned* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  const size_t nthreads = (size_t)gridDim.x * 256, tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)lds + (threadIdx.x >> 6) * 1024u));
  for (size_t i = tid; i < lines * 4; i += nthreads) {
    const uint8_t* p = src + (i >> 2) * 128 + (i & 3) * 16;
    if (P == 0) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(p), "s"(la) : "memory");
    if (P == 1) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(p), "s"(la) : "memory");
    if (P == 4) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off sc0 sc1" : : "v"(p), "s"(la) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (lds[threadIdx.x] == 0x5a && tid == 0x7fffffff) out[0] = 1;
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  const size_t bytes = (size_t)1 << 30, lines = bytes / 128;
  uint8_t* src;
  unsigned* out;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(src, 1, bytes);
  const int grid = 256 * 16;
  const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "whole lines"};
#define RUN(P) { const float ms = time_ms([&] { hipLaunchKernelGGL(touch<P>, dim3(grid), dim3(256), 0, 0, src, lines, out); }); \
                 printf("load    %-12s %.3f ms  (%.2f TB/s of lines x 128 B)\n", names[P], ms, bytes / ms / 1e9); }
  RUN(7) RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
#define RUND(P) { const float ms = time_ms([&] { hipLaunchKernelGGL(touch_dma<P>, dim3(grid), dim3(256), 0, 0, src, lines, out); }); \
                 printf("lds-dma %-12s %.3f ms  (%.2f TB/s of lines x 128 B)\n", names[P], ms, bytes / ms / 1e9); }
  RUND(0) RUND(1) RUND(4)
  return 0;
}
