// micro-benchmark: cost of a per-iteration workgroup barrier in a resident grid (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int MODE>
__global__ void k(int iters, int* out, int flag) {
  extern __shared__ int lds[];
  int acc = threadIdx.x;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (MODE == 1) __syncthreads();
    if (MODE == 2) { asm volatile("s_nop 0" ::: "memory"); }
    if (MODE == 3) { // barrier + a little LDS work
      lds[threadIdx.x] = acc; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); acc += lds[(threadIdx.x + 1) % blockDim.x];
    }
    acc += flag;
  }
  if (acc == 0x12345678) out[0] = acc;
}

template <int MODE>
float run(int blocks, int threads, int lds, int iters) {
  int* out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, 10, out, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, iters, out, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}

int main() {
  const int iters = 2000;
  for (int threads : {256, 320}) for (int lds : {1024, 26*1024, 40*1024}) for (int blocks : {256, 1280, 5616}) {
    float t0 = run<0>(blocks, threads, lds, iters), t1 = run<1>(blocks, threads, lds, iters), t2 = run<2>(blocks, threads, lds, iters), t3 = run<3>(blocks, threads, lds, iters);
    printf("threads %d lds %5d blocks %5d : ns/iter/kernel  asm-barrier %.1f  syncthreads %.1f  nop %.1f  barrier+lds %.1f\n", threads, lds, blocks,
           t0*1e6/iters, t1*1e6/iters, t2*1e6/iters, t3*1e6/iters);
  }
  return 0;
}
