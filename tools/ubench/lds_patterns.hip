// micro-benchmark: (1) LDS-DMA (global_load_lds_dwordx4) into LDS bases that are 4 / 8 / 12 bytes off a
// 16-byte boundary: correct? as fast?  (2) LDS read cost of the gather's access patterns for
// ds_read2_b32 (two aligned dwords) vs ds_read_b64 (one aligned qword), measured against the bank model of
// MI355X_MICROARCH.md "LDS" (32-lane groups; b32: bank = dword % 32, b64: bank = dword % 64; cycles of a
// group = max over banks of distinct addresses on it).  The model is what tests/plan_sim uses to rank
// LDS layouts offline, so it is checked here on the real patterns.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// ---------------------------------------------------------------- part 1: misaligned LDS-DMA
// one wave: lane l loads the 16 source bytes at src + 16*perm(l) to LDS byte (lds_off + 16*l)
__global__ __launch_bounds__(64) void dma_probe(const uint8_t* src, uint8_t* out, int lds_off, int use_inst_offset) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = 0xEE;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)lds + 256;
  const int goff = 16 * ((lane * 7) & 63);
  if (use_inst_offset) {
    // +4 carried by the instruction's immediate offset (applies to the global address as well: compensate)
    const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const int g2 = goff + 1024 - 4;  // the VGPR offset is unsigned: stay positive (src is passed 1 KiB low)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:4\n\ts_waitcnt vmcnt(0)"
                 :
                 : "v"(g2), "s"(src - 1024), "s"(m0v)
                 : "memory");
  } else {
    const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)lds_off));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_waitcnt vmcnt(0)"
                 :
                 : "v"(goff), "s"(src), "s"(m0v)
                 : "memory");
  }
  __syncthreads();
  for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}

// throughput: every wave of a 256-thread workgroup streams `iters` x 1 KiB from its own source range
__global__ __launch_bounds__(256) void dma_rate(const uint8_t* src, int iters, int lds_off, size_t wg_stride, unsigned* sink) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base_v = (uint64_t)(uintptr_t)(src + (size_t)blockIdx.x * wg_stride + (size_t)wave * (wg_stride / 4));
  const uint8_t* base = reinterpret_cast<const uint8_t*>(
      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base_v >> 32)) << 32) |
      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base_v));
  const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)lds + wave * 4096 + lds_off));
  int goff = lane * 16;
  for (int i = 0; i < iters; i++) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t" : : "v"(goff), "s"(base), "s"(m0v) : "memory");
    goff += 1024;
    if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[threadIdx.x] == 0x5A && lds[threadIdx.x + 77] == 0xA5) sink[0] = 1;
}

// ---------------------------------------------------------------- part 2: read patterns
// MODE 0: ds_read2_b32 (addr, addr+4)   MODE 1: ds_read_b64 (addr)    MODE 2: ds_read_b32 (addr)
template <int MODE>
__global__ __launch_bounds__(256) void read_rate(const int* addr_tab, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(64))) uint8_t lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  // all four waves of the workgroup use the same 64-entry pattern, on two 16 KiB regions
  const uint32_t a = (uint32_t)(uintptr_t)lds + (uint32_t)addr_tab[threadIdx.x & 63] + ((threadIdx.x >> 6) & 1) * 16384u;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
    if (MODE == 0) {
      uint64_t q0, q1, q2, q3;
      asm volatile(
          "ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset1:1\n\tds_read2_b32 %2, %4 offset1:1\n\t"
          "ds_read2_b32 %3, %4 offset1:1\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
          : "v"(a)
          : "memory");
      r0 = (uint32_t)q0; r1 = (uint32_t)(q0 >> 32); r2 = (uint32_t)q1; r3 = (uint32_t)(q1 >> 32);
      r4 = (uint32_t)q2; r5 = (uint32_t)(q2 >> 32); r6 = (uint32_t)q3; r7 = (uint32_t)(q3 >> 32);
    } else if (MODE == 1) {
      uint64_t q0, q1, q2, q3;
      asm volatile(
          "ds_read_b64 %0, %4\n\tds_read_b64 %1, %4\n\tds_read_b64 %2, %4\n\tds_read_b64 %3, %4\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
          : "v"(a)
          : "memory");
      r0 = (uint32_t)q0; r1 = (uint32_t)(q0 >> 32); r2 = (uint32_t)q1; r3 = (uint32_t)(q1 >> 32);
      r4 = (uint32_t)q2; r5 = (uint32_t)(q2 >> 32); r6 = (uint32_t)q3; r7 = (uint32_t)(q3 >> 32);
    } else {
      asm volatile(
          "ds_read_b32 %0, %4\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %4\n\tds_read_b32 %3, %4\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
          : "v"(a)
          : "memory");
      r4 = r5 = r6 = r7 = 0;
    }
    acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

static double model_cycles(const std::vector<int>& addr, int mode) {
  // mode 0: two b32 accesses (a, a+4), banks % 32; mode 1: one b64 access, dwords a/4, a/4+1, banks % 64; mode 2: b32
  double cyc = 0;
  const int naccess = mode == 0 ? 2 : 1;
  for (int acc = 0; acc < naccess; acc++)
    for (int g = 0; g < 2; g++) {
      const int nb = mode == 1 ? 64 : 32;
      std::vector<std::set<int>> banks(nb);
      for (int l = 32 * g; l < 32 * g + 32; l++) {
        const int d = addr[l] / 4 + acc;
        banks[d % nb].insert(d);
        if (mode == 1) banks[(d + 1) % nb].insert(d + 1);
      }
      size_t m = 1;
      for (auto& b : banks) m = std::max(m, b.size());
      cyc += (double)m;
    }
  return cyc;
}

template <int MODE>
static float time_pattern(const std::vector<int>& addr, int* d_tab, unsigned* sink, int iters) {
  CK(hipMemcpy(d_tab, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU -> 4 waves per SIMD
  hipLaunchKernelGGL(read_rate<MODE>, dim3(blocks), dim3(256), 32768, 0, d_tab, 10, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(read_rate<MODE>, dim3(blocks), dim3(256), 32768, 0, d_tab, iters, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  // ---------------- part 1
  uint8_t *d_src, *d_out;
  const size_t src_bytes = (size_t)1 << 30;
  CK(hipMalloc(&d_src, src_bytes));
  CK(hipMalloc(&d_out, 4096));
  std::vector<uint8_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((i * 131) ^ (i >> 7));
  CK(hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice));
  for (int inst = 0; inst < 2; inst++)
    for (int off : {0, 4, 8, 12}) {
      if (inst && off != 4) continue;
      hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 8192, 0, d_src + 4096, d_out, off, inst);
      CK(hipDeviceSynchronize());
      std::vector<uint8_t> o(4096);
      CK(hipMemcpy(o.data(), d_out, 4096, hipMemcpyDeviceToHost));
      int bad = 0, untouched_bad = 0;
      const int eff = inst ? 4 : off;
      for (int l = 0; l < 64; l++)
        for (int b = 0; b < 16; b++)
          if (o[256 + eff + 16 * l + b] != h[4096 + 16 * ((l * 7) & 63) + b]) bad++;
      for (int i = 0; i < 256 + eff; i++) untouched_bad += o[i] != 0xEE;
      for (int i = 256 + eff + 1024; i < 4096; i++) untouched_bad += o[i] != 0xEE;
      printf("LDS-DMA dwordx4 to base+%d (%s): %d wrong bytes of 1024, %d stray writes\n", eff,
             inst ? "inst offset:4" : "M0", bad, untouched_bad);
      if (bad) {
        printf("   first lanes got:");
        for (int b = 0; b < 24; b++) printf(" %02x", o[256 + b]);
        printf("\n   wanted        :");
        for (int b = 0; b < 16; b++) printf(" %02x", h[4096 + b]);
        printf("\n");
      }
    }
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  for (int off : {0, 4, 8}) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int blocks = 1024, iters = 256;  // 1024 WG x 4 waves x 256 KiB = 1 GiB
    const size_t wg_stride = (size_t)4 * iters * 1024;
    hipLaunchKernelGGL(dma_rate, dim3(blocks), dim3(256), 32768, 0, d_src, 8, off, wg_stride, sink);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(dma_rate, dim3(blocks), dim3(256), 32768, 0, d_src, iters, off, wg_stride, sink);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      best = std::min(best, ms);
    }
    printf("LDS-DMA stream, LDS base +%d: 1 GiB in %.3f ms = %.2f TB/s\n", off, best, (double)src_bytes / best / 1e9);
  }

  // ---------------- part 2
  int* d_tab;
  CK(hipMalloc(&d_tab, 64 * sizeof(int)));
  const int iters = 20000;
  // reference: conflict-free ds_read_b32 (model: 2 cycles per instruction) fixes the clock
  std::vector<int> lin(64);
  for (int l = 0; l < 64; l++) lin[l] = 4 * l;
  const float ms_ref = time_pattern<2>(lin, d_tab, sink, iters);
  const double ms_per_cycle = ms_ref / (model_cycles(lin, 2) * 4.0 * 16.0 * iters);  // 16 waves per CU x 4 instr per iter
  printf("reference ds_read_b32 linear: %.3f ms -> %.3f ns per LDS cycle if the model's 2 cycles hold (%.2f GHz)\n", ms_ref,
         ms_per_cycle * 1e6, 1e-6 / ms_per_cycle);
  struct Pat {
    const char* name;
    std::vector<int> a;   // byte address of the 4-byte window start (any alignment)
  };
  std::vector<Pat> pats;
  auto staircase = [&](const char* name, double px_step, int lanes_per_row, int pitch, int dir) {
    Pat p;
    p.name = name;
    p.a.resize(64);
    for (int g = 0; g < 2; g++)
      for (int i = 0; i < 32; i++) {
        const int row = 20 + dir * (i / lanes_per_row) + 3 * g;
        const int x = 40 + (int)std::floor(px_step * i);
        p.a[32 * g + i] = row * pitch + x;
      }
    pats.push_back(p);
  };
  staircase("one row, 2.4 B/lane", 2.4, 64, 192, 1);
  staircase("stairs 3 lanes/row, pitch 192", 2.4, 3, 192, 1);
  staircase("stairs 3 lanes/row, pitch 176", 2.4, 3, 176, 1);
  staircase("stairs 3 lanes/row, pitch 208", 2.4, 3, 208, 1);
  staircase("stairs 3 lanes/row, pitch 256", 2.4, 3, 256, 1);
  staircase("stairs down 3 lanes/row, pitch 192", 2.4, 3, 192, -1);
  staircase("stairs 2 lanes/row, pitch 160, 1.3 B/lane", 1.3, 2, 160, 1);
  staircase("stairs 8 lanes/row, pitch 192", 2.4, 8, 192, 1);
  staircase("stairs 1 lane/row, pitch 144", 2.0, 1, 144, 1);
  for (auto& p : pats) {
    std::vector<int> a2(64), a8(64);
    for (int l = 0; l < 64; l++) {
      a2[l] = p.a[l] & ~3;  // read2_b32: aligned dword pair
      a8[l] = p.a[l] & ~7;  // read_b64: aligned qword (the B copy would shift odd dwords; same bank statistics)
    }
    const float m0 = time_pattern<0>(a2, d_tab, sink, iters);
    const float m1 = time_pattern<1>(a8, d_tab, sink, iters);
    const double c0 = m0 / (4.0 * 16.0 * iters) / ms_per_cycle, c1 = m1 / (4.0 * 16.0 * iters) / ms_per_cycle;
    printf("%-44s read2_b32: measured %.2f cyc/instr (model %.0f) | read_b64: measured %.2f (model %.0f)\n", p.name, c0,
           model_cycles(a2, 0), c1, model_cycles(a8, 1));
  }
  return 0;
}
