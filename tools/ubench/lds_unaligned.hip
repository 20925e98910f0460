// tools/ubench/lds_unaligned.hip -- does gfx950 serve ds_read_b32 / ds_read_b64 at byte addresses that are not
// multiples of 4, does it return the right bytes, and what does it cost?  (Would remove the v_alignbit per stencil
// window of the gather.)  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void k(uint32_t* out, long long* cyc, int off, int stride, int iters, int width) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t a0 = (uint32_t)(uintptr_t)lds + threadIdx.x * stride + off;
  uint32_t acc = 0;
  uint32_t a = a0;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    uint32_t v0, v1 = 0;
    if (width == 4)
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v0) : "v"(a) : "memory");
    else {
      uint64_t v;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      v0 = (uint32_t)v; v1 = (uint32_t)(v >> 32);
    }
    acc += v0 ^ (v1 * 3);
    a = a0 + ((acc & 1) ? 0 : 0);  // keep the address loop-carried for the compiler, unchanged in value
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // correctness of one read
  uint32_t v0, v1 = 0;
  if (width == 4)
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v0) : "v"(a0) : "memory");
  else {
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a0) : "memory");
    v0 = (uint32_t)v; v1 = (uint32_t)(v >> 32);
  }
  out[threadIdx.x * 2] = v0;
  out[threadIdx.x * 2 + 1] = v1;
  out[128 + 0] += acc & 0;
}

// throughput: 8 independent reads per iteration, 16 waves on one CU
__global__ void kt(uint32_t* out, long long* cyc, int off, int stride, int iters, int width) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * stride + off + (threadIdx.x >> 6) * 64;
  uint32_t acc = 0;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (width == 4) {
      uint32_t v[8];
      asm volatile(
          "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
          "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
          : "v"(a)
          : "memory");
      for (int j = 0; j < 8; j++) acc += v[j];
    } else {
      uint64_t v[8];
      asm volatile(
          "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:256\n\tds_read_b64 %2, %8 offset:512\n\tds_read_b64 %3, %8 offset:768\n\t"
          "ds_read_b64 %4, %8 offset:1024\n\tds_read_b64 %5, %8 offset:1280\n\tds_read_b64 %6, %8 offset:1536\n\tds_read_b64 %7, %8 offset:1792\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
          : "v"(a)
          : "memory");
      for (int j = 0; j < 8; j++) acc += (uint32_t)v[j] ^ (uint32_t)(v[j] >> 32);
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = acc;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint32_t* d; long long* c;
  hipMalloc(&d, 4096); hipMalloc(&c, 64); hipMemset(d, 0, 4096);
  const int iters = 2000;
  for (int width : {4, 8})
    for (int stride : {4, 8, 20})
      for (int off = 0; off < 8; off++) {
        if (width == 4 && off >= 4) continue;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, off, stride, iters, width);
        uint32_t h[130]; long long hc;
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; t++) {
          const int base = t * stride + off;
          for (int b = 0; b < width; b++) {
            const uint8_t want = (uint8_t)((base + b) * 7 + 3);
            const uint8_t got = (uint8_t)(h[t * 2 + b / 4] >> (8 * (b % 4)));
            bad += want != got;
          }
        }
        printf("ds_read_b%d stride %2d off %d : %s  %.1f clock64 ticks per read (incl. round trip)\n", width * 8, stride, off,
               bad ? "WRONG BYTES" : "ok", (double)hc / iters);
      }
  for (int width : {4, 8})
    for (int stride : {4, 8, 20})
      for (int off : {0, 1, 2, 4}) {
        hipLaunchKernelGGL(kt, dim3(1), dim3(1024), 0, 0, d, c, off, stride, iters, width);
        long long hc;
        hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("THROUGHPUT ds_read_b%d stride %2d off %d : %.2f cycles per wave-instruction (16 waves x 8 reads per iteration on one CU)\n",
               width * 8, stride, off, (double)hc / iters / (16 * 8));
      }
  return 0;
}
