// tools/ubench/h2d_pageable.hip -- what a SAFE host->device copy of a caller's (pageable) plane costs: the runtime's
// own pageable path (same buffer every time / a rotation of 4 buffers / a freshly allocated buffer every time) vs
// register + copy + unregister per call vs a cached registration (which goes stale when the caller unmaps the buffer).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  char* d; hipMalloc((void**)&d, 16 << 20);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (size_t n : {(size_t)3840 * 1920, (size_t)1920 * 960, (size_t)1536 * 1024}) {
    char* h[4];
    for (auto& p : h) { p = (char*)aligned_alloc(4096, n); memset(p, 1, n); }
    double same = 1e9, rot = 1e9, fresh = 1e9, reg = 1e9, cached = 1e9;
    for (int rep = 0; rep < 30; rep++) {
      double t0 = now(); hipMemcpyAsync(d, h[0], n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); same = std::min(same, now() - t0);
    }
    for (int rep = 0; rep < 32; rep++) {
      double t0 = now(); hipMemcpyAsync(d, h[rep & 3], n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); rot = std::min(rot, now() - t0);
    }
    for (int rep = 0; rep < 10; rep++) {
      char* f = (char*)aligned_alloc(4096, n); memset(f, 2, n);
      double t0 = now(); hipMemcpyAsync(d, f, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); fresh = std::min(fresh, now() - t0);
      free(f);
    }
    for (int rep = 0; rep < 10; rep++) {
      double t0 = now(); hipHostRegister(h[1], n, hipHostRegisterDefault); hipMemcpyAsync(d, h[1], n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
      hipHostUnregister(h[1]); reg = std::min(reg, now() - t0);
    }
    hipHostRegister(h[2], n, hipHostRegisterDefault);
    for (int rep = 0; rep < 20; rep++) { double t0 = now(); hipMemcpyAsync(d, h[2], n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); cached = std::min(cached, now() - t0); }
    hipHostUnregister(h[2]);
    printf("%8zu B H2D us: pageable same buffer %.0f | rotation of 4 %.0f | fresh buffer %.0f | register+copy+unregister %.0f | cached registration %.0f\n",
           n, same * 1e6, rot * 1e6, fresh * 1e6, reg * 1e6, cached * 1e6);
    for (auto& p : h) free(p);
  }
  return 0;
}
