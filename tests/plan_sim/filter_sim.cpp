// tests/plan_sim/filter_sim.cpp -- the library's HOST-side low-pass configuration (transform360_amd/csrc/
// t360_filtercfg.cpp: segments, Gaussian taps, Q8 taps, the shifted tap variants of the wide path) built for the host,
// so that tests/test_filtercfg_cpu.py can compare it with the oracle without a GPU.  Development / test tool.
#include <cstring>

#include "t360_filtercfg.h"

// segments of one map: returns the count (or -1 when the configuration fails like the reference's generateMapForPlane);
// rects[4*i..] = left, top, width, height; lens[2*i..] = taps of kx, ky; taps = all kx then ky floats, segment by segment;
// q8 likewise as ints
extern "C" int t360_host_filter_config(const FrameTransformContext* ctx, int inW, int inH, int outW, int outH, int cap,
                                       int* rects, int* lens, float* taps, int* q8, int tap_cap) {
  t360::FilterConfig cfg;
  if (!t360::build_filter_config(*ctx, inW, inH, outW, outH, &cfg)) return -1;
  int n = 0, at = 0;
  for (const t360::Segment& s : cfg.segments) {
    if (n >= cap || at + (int)s.kx.size() + (int)s.ky.size() > tap_cap) return -2;
    rects[4 * n] = s.left; rects[4 * n + 1] = s.top; rects[4 * n + 2] = s.width; rects[4 * n + 3] = s.height;
    lens[2 * n] = (int)s.kx.size(); lens[2 * n + 1] = (int)s.ky.size();
    for (size_t k = 0; k < s.kx.size(); k++) { taps[at] = s.kx[k]; q8[at] = k < s.kx_q8.size() ? s.kx_q8[k] : -1; at++; }
    for (size_t k = 0; k < s.ky.size(); k++) { taps[at] = s.ky[k]; q8[at] = k < s.ky_q8.size() ? s.ky_q8[k] : -1; at++; }
    n++;
  }
  return n;
}

// out: 4 x kWideTapStride dwords; returns nd (0: not eligible)
extern "C" int t360_host_shifted_taps(const int* kx_q8, int n, unsigned* out) {
  std::vector<int> k(kx_q8, kx_q8 + n);
  std::vector<uint32_t> v;
  const int nd = t360::pack_shifted_taps(k, &v);
  if (nd) memcpy(out, v.data(), v.size() * sizeof(uint32_t));
  return nd;
}
