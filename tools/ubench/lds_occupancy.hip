// how many workgroups fit a CU for a given dynamic LDS size? (runtime's answer + a measured one: workgroups that
// overlap in time on CU 0)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k512(unsigned long long* out) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);  // 20 us
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t0; out[blockIdx.x * 2 + 1] = wall_clock64(); }
}
__global__ __launch_bounds__(256) void k256(unsigned long long* out) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t0; out[blockIdx.x * 2 + 1] = wall_clock64(); }
}
int main() {
  unsigned long long* out; (void)hipMalloc(&out, 1 << 20);
  for (int kb : {76, 78, 79, 80}) {
    int n = 0;
    (void)hipFuncSetAttribute((const void*)k512, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k512, 512, kb * 1024);
    // measured: 2 x 256 workgroups; if all are resident at once the launch takes ~20 us, else ~40
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k512, dim3(512), dim3(512), kb * 1024, 0, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("512 threads, %d KiB: runtime says %d per CU; 512 workgroups took %.1f us (%s)\n", kb, n, ms * 1e3, hipGetErrorString(hipGetLastError()));
  }
  for (int kb : {38, 39, 40}) {
    int n = 0;
    (void)hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k256, 256, kb * 1024);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k256, dim3(1024), dim3(256), kb * 1024, 0, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("256 threads, %d KiB: runtime says %d per CU; 1024 workgroups took %.1f us (%s)\n", kb, n, ms * 1e3, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
