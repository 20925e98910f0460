// micro-benchmark: how the XCD's L2 treats the access patterns of the tiled gather.  Run under
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum -- ./l2_probe.bin
// and read the fabric reads per kernel (tools/ubench/l2_probe_report.py).
//   halves<M>  : every 128-byte line is read as two 64-byte halves.  M=0 one instruction per 8 whole lines; M=1 the
//                two halves by consecutive instructions of one wave; M=2 second halves ~microseconds later (after the
//                wave has asked for the first halves of 16 KiB); M=3 second halves by ANOTHER wave of the workgroup.
//                Fabric reads = lines (merged / hit) or 2 x lines (each half fetched on its own)?
//   reread<D>  : 64 workgroups on ONE XCD (blockIdx % 8 == 0) stream a region of S bytes in total twice; the second
//                pass hits if the L2 kept S bytes: effective capacity for streamed lines.  D=1: through LDS-DMA.
//   lag<D>     : workgroup pairs on one XCD read the same lines, the second `lag` microseconds later, while the
//                other workgroups of the XCD stream fresh lines at full rate: how long does a line survive?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }  // 100 MHz

template <int M>
__global__ __launch_bounds__(256) void halves(const uint8_t* __restrict__ src, unsigned* out) {
  // workgroup = 64 KiB = 512 lines; wave w owns lines w*128 .. w*128+127
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t* base = src + (size_t)blockIdx.x * 65536;
  unsigned acc = 0;
  auto ld = [&](size_t off) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + off);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  };
  if (M == 0) {
    for (int i = 0; i < 16; i++) ld((size_t)wave * 16384 + (size_t)i * 1024 + (size_t)lane * 16);
  } else if (M == 1) {
    for (int i = 0; i < 8; i++) {  // 16 lines per pair of instructions
      const size_t l0 = (size_t)wave * 16384 + (size_t)i * 2048 + (size_t)(lane >> 2) * 128 + (size_t)(lane & 3) * 16;
      ld(l0);
      ld(l0 + 64);
    }
  } else if (M == 2) {
    for (int i = 0; i < 8; i++) ld((size_t)wave * 16384 + (size_t)i * 2048 + (size_t)(lane >> 2) * 128 + (size_t)(lane & 3) * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 8; i++) ld((size_t)wave * 16384 + (size_t)i * 2048 + (size_t)(lane >> 2) * 128 + (size_t)(lane & 3) * 16 + 64);
  } else {
    const int w2 = wave ^ 1;  // the other wave's second halves, issued at about the same time
    for (int i = 0; i < 8; i++) {
      ld((size_t)wave * 16384 + (size_t)i * 2048 + (size_t)(lane >> 2) * 128 + (size_t)(lane & 3) * 16);
      ld((size_t)w2 * 16384 + (size_t)i * 2048 + (size_t)(lane >> 2) * 128 + (size_t)(lane & 3) * 16 + 64);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__device__ __forceinline__ void dma16(const uint8_t* gbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}

template <int DMA>
__global__ __launch_bounds__(256) void reread(const uint8_t* __restrict__ src, size_t bytes_per_wg, int passes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  if (blockIdx.x & 7) return;  // XCD 0 only
  const int wg = blockIdx.x >> 3;
  const uint8_t* base = src + (size_t)wg * bytes_per_wg;
  unsigned acc = 0;
  for (int p = 0; p < passes; p++) {
    for (size_t off = (size_t)threadIdx.x * 16; off < bytes_per_wg; off += 4096) {
      if (DMA) {
        const uint32_t la = (uint32_t)(uintptr_t)lds + (threadIdx.x >> 6) * 1024u;
        dma16(base, (uint32_t)off, (uint32_t)__builtin_amdgcn_readfirstlane((int)la));
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(base + off);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (DMA) acc += lds[threadIdx.x];
  if (acc == 0x12345678u) out[0] = acc;
}

// pairs (leader, follower) of workgroups on XCD 0 read the same region of `region` bytes in steps of 16 KiB every
// `step_ticks`; the follower starts `lag_ticks` later.  The other (noise) workgroups of the XCD stream private lines.
template <int DMA>
__global__ __launch_bounds__(256) void lag(const uint8_t* __restrict__ src, const uint8_t* __restrict__ noise, size_t region,
                                           int lag_ticks, int step_ticks, int npairs, unsigned long long* t0p, unsigned* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  if (blockIdx.x & 7) return;
  const int wg = blockIdx.x >> 3;
  // a common time origin: first workgroup to arrive publishes it
  __shared__ unsigned long long t0s;
  if (threadIdx.x == 0) {
    unsigned long long t = now() + 2000;  // 20 us from now
    unsigned long long old = atomicCAS(t0p, 0ull, t);
    t0s = old ? old : t;
  }
  __syncthreads();
  const unsigned long long t0 = t0s;
  unsigned acc = 0;
  const bool paired = wg < 2 * npairs;
  const bool follower = paired && (wg & 1);
  const uint8_t* base = paired ? src + (size_t)(wg >> 1) * region : noise + (size_t)(wg - 2 * npairs) * region;
  const int steps = (int)(region / 16384);
  for (int s = 0; s < steps; s++) {
    const unsigned long long due = t0 + (unsigned long long)s * step_ticks + (follower ? lag_ticks : 0);
    while (now() < due) __builtin_amdgcn_s_sleep(2);
    for (int k = 0; k < 4; k++) {
      const size_t off = (size_t)s * 16384 + (size_t)k * 4096 + (size_t)threadIdx.x * 16;
      if (DMA) {
        const uint32_t la = (uint32_t)(uintptr_t)lds + (threadIdx.x >> 6) * 1024u;
        dma16(base, (uint32_t)off, (uint32_t)__builtin_amdgcn_readfirstlane((int)la));
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(base + off);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (DMA) acc += lds[threadIdx.x];
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t N = (size_t)1 << 30;
  uint8_t* src; unsigned* out; unsigned long long* t0;
  if (hipMalloc(&src, N) != hipSuccess || hipMalloc(&out, 64) != hipSuccess || hipMalloc(&t0, 8) != hipSuccess) return 1;
  (void)hipMemset(src, 1, N);
  (void)hipDeviceSynchronize();
  // ---- halves: 256 MiB per kernel (4096 workgroups x 64 KiB), a different region each: 2 097 152 lines
  hipLaunchKernelGGL(halves<0>, dim3(4096), dim3(256), 0, 0, src, out);
  hipLaunchKernelGGL(halves<1>, dim3(4096), dim3(256), 0, 0, src + ((size_t)256 << 20), out);
  hipLaunchKernelGGL(halves<2>, dim3(4096), dim3(256), 0, 0, src + ((size_t)512 << 20), out);
  hipLaunchKernelGGL(halves<3>, dim3(4096), dim3(256), 0, 0, src + ((size_t)768 << 20), out);
  (void)hipDeviceSynchronize();
  // ---- reread: S = 0.5 .. 8 MiB on one XCD, 64 workgroups, 2 passes (+ a 1-pass reference)
  const double S[] = {0.5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 5, 6, 8};
  size_t at = 0;
  for (int dma = 0; dma < 2; dma++)
    for (double s : S) {
      const size_t per_wg = ((size_t)(s * 1048576.0) / 64) & ~(size_t)4095;
      for (int passes = 1; passes <= 2; passes++) {
        if (dma) hipLaunchKernelGGL(reread<1>, dim3(512), dim3(256), 0, 0, src + at, per_wg, passes, out);
        else hipLaunchKernelGGL(reread<0>, dim3(512), dim3(256), 0, 0, src + at, per_wg, passes, out);
        at += (size_t)16 << 20;
        (void)hipDeviceSynchronize();
        printf("reread dma=%d S=%.1f MiB passes=%d lines=%zu\n", dma, s, passes, per_wg * 64 / 128);
      }
    }
  // ---- lag: 16 pairs + 32 noise workgroups on one XCD, 1 MiB regions, 16 KiB per 1 us step (=> XCD streams
  // 64 x 16 KiB = 1 MiB of requests per us, like the gather), follower lag 0.5 .. 16 us
  const int lags[] = {0, 50, 100, 200, 400, 800, 1600};
  for (int dma = 0; dma < 2; dma++)
    for (int lg : lags) {
      (void)hipMemset(t0, 0, 8);
      const size_t region = (size_t)1 << 20;
      const uint8_t* s0 = src + ((size_t)(dma * 8 + (&lg - lags)) * 64 << 20) % (N - ((size_t)64 << 20));
      if (dma) hipLaunchKernelGGL(lag<1>, dim3(512), dim3(256), 0, 0, s0, s0 + ((size_t)16 << 20), region, lg, 100, 16, t0, out);
      else hipLaunchKernelGGL(lag<0>, dim3(512), dim3(256), 0, 0, s0, s0 + ((size_t)16 << 20), region, lg, 100, 16, t0, out);
      (void)hipDeviceSynchronize();
      printf("lag dma=%d lag=%.1f us lines_min=%d lines_nosharing=%d\n", dma, lg / 100.0, (16 + 32) * 8192, (32 + 32) * 8192);
    }
  printf("done: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
