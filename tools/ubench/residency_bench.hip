// micro-benchmark: direct measurement of co-resident workgroups per CU on MI355X.
// Every workgroup records (XCC id, HW_ID, start, end); the host counts overlaps per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };
template <int NV>
__global__ void spin(unsigned long long ticks, Rec* rec, float* out, int flag) {
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
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    rec[blockIdx.x] = Rec{xcc & 0xf, hw, t0, (unsigned long long)wall_clock64()};
  }
}
template <int NV>
void run(int threads, int lds, int blocks, double T_us) {
  float* out; (void)hipMalloc(&out, 4);
  Rec* rec; (void)hipMalloc(&rec, sizeof(Rec) * blocks);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(spin<NV>, dim3(blocks), dim3(threads), lds, 0, (unsigned long long)(T_us * 100), rec, out, 0);
  (void)hipDeviceSynchronize();
  std::vector<Rec> h(blocks);
  (void)hipMemcpy(h.data(), rec, sizeof(Rec) * blocks, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
  for (auto& r : h) {
    unsigned key = (r.xcc << 16) | (r.hwid & 0xff00);  // cu, sh, se within the XCC
    ev[key].push_back({r.t0, +1});
    ev[key].push_back({r.t1, -1});
  }
  int gmax = 0; double avgmax = 0;
  for (auto& kv : ev) {
    auto& e = kv.second; std::sort(e.begin(), e.end());
    int cur = 0, mx = 0; for (auto& x : e) { cur += x.second; mx = std::max(mx, cur); }
    gmax = std::max(gmax, mx); avgmax += mx;
  }
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin<NV>));
  printf("threads %3d regs %3d lds %6d : distinct CUs seen %zu, max co-resident workgroups per CU: max %d, mean %.2f\n", threads,
         fa.numRegs, lds, ev.size(), gmax, avgmax / ev.size());
  (void)hipFree(out); (void)hipFree(rec);
}
int main() {
  const int B = 256 * 24;
  run<8>(256, 1024, B, 50); run<24>(256, 1024, B, 50); run<48>(256, 1024, B, 50); run<64>(256, 1024, B, 50); run<96>(256, 1024, B, 50);
  run<8>(320, 1024, B, 50); run<48>(320, 1024, B, 50); run<64>(320, 1024, B, 50);
  run<48>(320, 26 * 1024, B, 50); run<48>(256, 26 * 1024, B, 50); run<48>(256, 20 * 1024, B, 50); run<48>(256, 40 * 1024, B, 50);
  run<48>(64, 1024, B, 50); run<48>(128, 1024, B, 50);
  // round 2: the tiled gather's shape (5 waves, ~88 VGPRs) against LDS per workgroup
  for (int kb : {32, 36, 38, 39, 40, 48, 50, 52}) run<76>(320, kb * 1024, B, 50);
  for (int kb : {32, 38, 40}) run<64>(320, kb * 1024, B, 50);
  return 0;
}
