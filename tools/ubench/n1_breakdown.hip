// tools/ubench/n1_breakdown.hip -- where the 0.37 ms per frame of the literal host-pointer ABI go:
// the six PCIe transfers of one 4K yuv420p frame (pageable caller memory, the copies the library issues), alone and
// -- what banding inside a call could buy -- the device->host copy of one plane running BESIDE the host->device copy of
// another on a second host thread and stream (full duplex).  Pure HIP, no library: ./n1_breakdown.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t in_y = 3840 * 1920, in_c = 1920 * 960, out_y = 1536 * 1024, out_c = 768 * 512;
  char *d_in, *d_out;
  hipMalloc((void**)&d_in, in_y);
  hipMalloc((void**)&d_out, out_y);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  char* hin = (char*)aligned_alloc(4096, in_y);
  char* hout = (char*)aligned_alloc(4096, out_y);
  memset(hin, 1, in_y);
  memset(hout, 2, out_y);
  auto h2d = [&](size_t n, hipStream_t s) { hipMemcpyAsync(d_in, hin, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); };
  auto d2h = [&](size_t w, size_t h, hipStream_t s) { hipMemcpy2DAsync(hout, w, d_out, (w + 255) & ~(size_t)255, w, h, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); };
  auto d2h_flat = [&](size_t n, hipStream_t s) { hipMemcpyAsync(hout, d_out, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); };
  auto timeit = [&](auto f) { std::vector<double> v; for (int r = 0; r < 40; r++) { const double t0 = now(); f(); v.push_back(now() - t0); } return median(v) * 1e6; };
  const double a = timeit([&] { h2d(in_y, s1); }), b = timeit([&] { h2d(in_c, s1); });
  const double c = timeit([&] { d2h(1536, 1024, s1); }), d = timeit([&] { d2h(768, 512, s1); });
  const double cf = timeit([&] { d2h_flat(out_y, s1); });
  printf("H2D luma %.0f us (%.1f GB/s), chroma %.0f us; D2H pitched luma %.0f us (%.1f GB/s), chroma %.0f us; D2H flat luma %.0f us\n", a, in_y / a * 1e-3, b,
         c, out_y / c * 1e-3, d, cf);
  printf("six transfers of a frame, back to back: %.0f us\n", a + 2 * b + c + 2 * d);
  // half-plane H2D next to a half-plane D2H on a second thread (what a 2-band split of one call would overlap)
  for (int mode = 0; mode < 2; mode++) {
    const double both = timeit([&] {
      std::thread th([&] { if (mode) d2h(1536, 512, s2); else d2h_flat(out_y / 2, s2); });
      h2d(in_y / 4, s1);
      th.join();
    });
    const double h = timeit([&] { h2d(in_y / 4, s1); }), o = timeit([&] { if (mode) d2h(1536, 512, s1); else d2h_flat(out_y / 2, s1); });
    printf("quarter-plane H2D %.0f us, half-plane D2H (%s) %.0f us, both at once on two threads %.0f us (thread start included)\n", h, mode ? "pitched" : "flat", o, both);
  }
  // two H2D halves on two threads / streams (two SDMA engines) vs one copy
  const double split = timeit([&] {
    std::thread th([&] { hipMemcpyAsync(d_in + in_y / 2, hin + in_y / 2, in_y / 2, hipMemcpyHostToDevice, s2); hipStreamSynchronize(s2); });
    hipMemcpyAsync(d_in, hin, in_y / 2, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1);
    th.join();
  });
  printf("luma H2D as two halves on two threads: %.0f us (one copy: %.0f us)\n", split, a);
  // a kernel-free call skeleton: H2D + D2H + one sync at the end (do the async calls return before the copies finish?)
  const double chain = timeit([&] { hipMemcpyAsync(d_in, hin, in_y, hipMemcpyHostToDevice, s1); hipMemcpy2DAsync(hout, 1536, d_out, 1536, 1536, 1024, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); });
  const double issue = timeit([&] { const double t0 = now(); hipMemcpyAsync(d_in, hin, in_y, hipMemcpyHostToDevice, s1); const double t1 = now(); hipStreamSynchronize(s1); (void)t0; (void)t1; });
  std::vector<double> ret;
  for (int r = 0; r < 20; r++) { const double t0 = now(); hipMemcpyAsync(d_in, hin, in_y, hipMemcpyHostToDevice, s1); ret.push_back(now() - t0); hipStreamSynchronize(s1); }
  printf("H2D + D2H chained, one sync: %.0f us; hipMemcpyAsync(pageable H2D) returns after %.0f us of %.0f\n", chain, median(ret) * 1e6, issue);
  return 0;
}
