// micro-benchmark: issue rate of the integer VALU ops the gather uses (wave64, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v2s __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned* out, unsigned seed) {
  unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  const unsigned b = 0x01020304u + seed;
  for (int i = 0; i < iters; i++) {
#define STEP(x)                                                                                      \
    if (OP == 0) x = __builtin_amdgcn_udot4(x, b, x, false);                                         \
    if (OP == 1) x = (unsigned)__builtin_amdgcn_sdot4((int)x, (int)b, (int)x, false);                \
    if (OP == 2) x = (unsigned)__builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, x), __builtin_bit_cast(v2s, b), (int)x, false); \
    if (OP == 3) x = __builtin_amdgcn_alignbit(x, b, x & 31);                                        \
    if (OP == 4) x = __builtin_amdgcn_perm(x, b, x);                                                 \
    if (OP == 5) x = x * 0x00ffffffu + b; /* v_mad_u32_u24-ish */                                    \
    if (OP == 6) x = x ^ b;                                                                          \
    if (OP == 7) { float f = __builtin_bit_cast(float, x); f = f * 1.0001f + 0.5f; x = __builtin_bit_cast(unsigned, f); }
    STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
  }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u) out[0] = a0;
}
template <int OP>
void run(const char* name) {
  unsigned* out; (void)hipMalloc(&out, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 4000, blocks = 256 * 8;  // 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, 10, out, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, out, 1u);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)blocks * 4 * iters * 8;  // wave-instructions
  // per SIMD: winstr / 1024 instructions in ms -> cycles per instruction at 2.4 GHz upper bound
  printf("%-14s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / (winstr / 1024), ms * 1e6 / (winstr / 1024) * 2.4);
  (void)hipFree(out);
}
int main() {
  run<7>("v_fma_f32"); run<6>("v_xor_b32"); run<0>("v_dot4_u32_u8"); run<1>("v_dot4_i32_i8"); run<2>("v_dot2_i32_i16");
  run<3>("v_alignbit"); run<4>("v_perm_b32"); run<5>("v_mad_u32_u24");
  return 0;
}
