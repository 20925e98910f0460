// micro-benchmark: how many workgroups of a given shape are co-resident per CU (MI355X)?
// Each workgroup spins for T microseconds; concurrency = blocks * T / kernel_time.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NV>
__global__ void spin(unsigned long long ticks, float* out, int flag) {
  extern __shared__ int lds[];
  float v[NV];
  for (int i = 0; i < NV; i++) v[i] = threadIdx.x * (i + 1.0f) + flag;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
    for (int i = 0; i < NV; i++) v[i] = v[i] * 1.0001f + 0.5f;
  }
  float s = 0;
  for (int i = 0; i < NV; i++) s += v[i];
  if (s == 12345.678f) out[0] = s + lds[0];
}
template <int NV>
void run(int threads, int lds, int blocks, double T_us = 20.0) {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(spin<NV>, dim3(256), dim3(threads), lds, 0, 100, out, 0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(spin<NV>, dim3(blocks), dim3(threads), lds, 0, (unsigned long long)(T_us * 100), out, 0);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin<NV>));
  printf("T %.0f threads %d regs %3d lds %6d blocks %5d : %.1f us -> %.2f workgroups/CU resident\n", T_us, threads, fa.numRegs, lds, blocks,
         ms * 1e3, blocks * T_us / (ms * 1e3) / 256.0);
  (void)hipFree(out);
}
int main() {
  for (double T : {5.0, 20.0, 100.0}) {
    run<8>(256, 1024, 256 * 24, T);
    run<48>(256, 1024, 256 * 24, T);
    run<96>(256, 1024, 256 * 24, T);
    run<48>(320, 26 * 1024, 256 * 24, T);
  }
  return 0;
}
