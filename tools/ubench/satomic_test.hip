// are scalar atomics (s_atomic_add glc) coherent across the XCDs?  Every workgroup takes tickets from ONE counter;
// all tickets must be distinct and dense.  Also: latency of one blocking scalar / vector atomic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void tickets(unsigned* ctr, unsigned* out, unsigned long long* lat, int per_wg, int vec) {
  for (int i = 0; i < per_wg; i++) {
    unsigned v = 1;
    const unsigned long long t0 = wall_clock64();
    if (vec) {
      if (threadIdx.x == 0) v = atomicAdd(ctr, 1u);
      v = __builtin_amdgcn_readfirstlane(v);
    } else {
      asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
      out[blockIdx.x * per_wg + i] = v;
      if (i == per_wg - 1) lat[blockIdx.x] = t1 - t0;
    }
  }
}
int main() {
  const int wgs = 2048, per = 16;
  unsigned *ctr, *out; unsigned long long* lat;
  (void)hipMalloc(&ctr, 4); (void)hipMalloc(&out, wgs * per * 4); (void)hipMalloc(&lat, wgs * 8);
  for (int vec = 0; vec < 2; vec++) {
    (void)hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(tickets, dim3(wgs), dim3(64), 0, 0, ctr, out, lat, per, vec);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * per); std::vector<unsigned long long> l(wgs);
    (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(l.data(), lat, l.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += h[i] != i;
    std::sort(l.begin(), l.end());
    printf("%s atomics: %zu tickets, %zu out of place (0 = coherent across XCDs); latency median %.2f us, p90 %.2f us (%s)\n",
           vec ? "vector" : "scalar", h.size(), bad, l[l.size() / 2] / 100.0, l[l.size() * 9 / 10] / 100.0, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
