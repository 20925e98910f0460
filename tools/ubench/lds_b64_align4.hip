// tools/ubench/lds_b64_align4.hip -- can the gather read a stencil-row window (4 bytes at any byte offset) as ONE
// ds_read_b64 at the 4-byte-aligned address below it, i.e. at addresses that are 4 mod 8?  (Today: two ds_read_b32;
// the T360_DUAL variant keeps a second copy of the staged bytes 4 further so that every window sits in an 8-byte-aligned
// qword.)  Checks the returned bytes and measures LDS throughput with 16 waves per CU, 16 reads per wait:
//   b32      ds_read_b32, dword stride            (128 B/clk/CU by MI355X_MICROARCH.md)
//   b32x2    two ds_read_b32 (addr, addr + 4)     (what the gather does per window today)
//   b64a     ds_read_b64, 8-byte aligned          (256 B/clk/CU)
//   b64m     ds_read_b64 at 4 mod 8
//   r2       ds_read2_b32 offset0:0 offset1:1
// and the same with the gather's address pattern (neighbouring lanes 1.9 bytes apart, rounded down to a dword).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_b64 lds_b64_align4.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc, int iters, int pattern, int check) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0x01000193u * (uint32_t)i + 12345u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t a0;
  if (pattern == 0) {
    a0 = (uint32_t)(lane * (MODE == 0 ? 4 : 8) + wave * 1024);  // conflict-free strides
    if (MODE == 3 || MODE == 4 || MODE == 1) a0 += 4;             // 4 mod 8
  } else {
    a0 = ((uint32_t)(lane * 19 / 10 + wave * 331) & (MODE == 2 ? ~7u : ~3u)) + wave * 1024;  // the gather's pattern: dword of a drifting byte address
  }
  a0 += (uint32_t)(uintptr_t)lds;
  uint32_t acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    uint32_t v[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t a = a0 + (uint32_t)j * (pattern == 0 ? 0u : 96u);
      if (MODE == 0) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[j]) : "v"(a), "n"(0));
        v[16 + j] = 0;
      } else if (MODE == 1) {
        asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(a));
        asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(v[16 + j]) : "v"(a));
      } else if (MODE == 2 || MODE == 3) {
        uint64_t q;
        asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(a));
        v[j] = (uint32_t)q;
        v[16 + j] = (uint32_t)(q >> 32);
      } else {
        uint64_t q;
        asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(q) : "v"(a));
        v[j] = (uint32_t)q;
        v[16 + j] = (uint32_t)(q >> 32);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 32; j++) {
      asm volatile("" : "+v"(v[j]));
      acc ^= v[j];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0xdeadbeefu) out[0] = acc;
  if (check && blockIdx.x == 0) {
    // one read per lane, values back to the host
    uint32_t lo, hi = 0;
    if (MODE == 0) {
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(a0) : "memory");
    } else if (MODE == 1) {
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(a0) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(hi) : "v"(a0) : "memory");
    } else if (MODE == 4) {
      uint64_t q;
      asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a0) : "memory");
      lo = (uint32_t)q; hi = (uint32_t)(q >> 32);
    } else {
      uint64_t q;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a0) : "memory");
      lo = (uint32_t)q; hi = (uint32_t)(q >> 32);
    }
    out[1 + threadIdx.x * 3] = a0 - (uint32_t)(uintptr_t)lds;
    out[2 + threadIdx.x * 3] = lo;
    out[3 + threadIdx.x * 3] = hi;
  }
}

template <int MODE>
void run(const char* name, int pattern) {
  uint32_t* out; unsigned long long* cyc;
  const int blocks = 256, iters = 2000;
  (void)hipMalloc(&out, 4 * (1 + 3 * 1024)); (void)hipMalloc(&cyc, 8 * blocks);
  (void)hipMemset(out, 0, 4 * (1 + 3 * 1024));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, cyc, 10, pattern, 0);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, pattern, 1);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<uint32_t> h(1 + 3 * 1024);
  std::vector<unsigned long long> hc(blocks);
  (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hc.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 1024; t++) {
    const uint32_t off = h[1 + t * 3];
    auto word = [](uint32_t byte_off) { return 0x01000193u * (byte_off / 4) + 12345u; };
    const uint32_t want_lo = word(off), want_hi = MODE == 0 ? 0 : word(off + 4);
    if (h[2 + t * 3] != want_lo || h[3 + t * 3] != want_hi) bad++;
  }
  double c = 0; for (auto v : hc) c += (double)v; c /= blocks;
  const double windows = (double)iters * 16 * 16;  // per CU: 16 waves x 16 windows per iteration
  printf("%-6s pattern %d: %s  %.1f cycles per wave-window per CU (%.2f us kernel), window = %d bytes per lane\n", name, pattern,
         bad ? "WRONG BYTES" : "bytes ok   ", c / windows, ms * 1e3, MODE == 0 ? 4 : 8);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int pattern = 0; pattern < 2; pattern++) {
    run<0>("b32", pattern);
    run<1>("b32x2", pattern);
    run<2>("b64a", pattern);
    run<3>("b64m", pattern);
    run<4>("r2", pattern);
  }
  return 0;
}
