// tools/ubench/h2d_split.hip -- host->device copy rate of one 4K luma plane (7.4 MB) from registered host memory:
// one hipMemcpyAsync vs the same bytes split over 2 / 4 streams (separate SDMA engines), and device->host likewise.
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t n = 3840 * 1920, nout = 1536 * 1024;
  void* h = aligned_alloc(4096, n);
  memset(h, 1, n);
  hipHostRegister(h, n, hipHostRegisterDefault);
  void* hp; hipHostMalloc(&hp, n, hipHostMallocDefault);
  char* d; hipMalloc((void**)&d, n);
  hipStream_t st[8];
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int dir = 0; dir < 2; dir++)
    for (void* host : {h, hp})
      for (int parts : {1, 2, 4, 8}) {
        const size_t bytes = dir ? nout : n;
        double best = 1e9;
        for (int rep = 0; rep < 30; rep++) {
          const double t0 = now();
          const size_t step = (bytes / parts + 4095) & ~(size_t)4095;
          for (int p = 0; p < parts; p++) {
            const size_t o = p * step, len = o >= bytes ? 0 : (o + step > bytes ? bytes - o : step);
            if (!len) continue;
            if (dir) hipMemcpyAsync((char*)host + o, d + o, len, hipMemcpyDeviceToHost, st[p]);
            else hipMemcpyAsync(d + o, (char*)host + o, len, hipMemcpyHostToDevice, st[p]);
          }
          for (int p = 0; p < parts; p++) hipStreamSynchronize(st[p]);
          const double t = now() - t0;
          if (t < best) best = t;
        }
        printf("%s %s %zu bytes in %d part(s): %.1f us  %.1f GB/s\n", dir ? "D2H" : "H2D", host == h ? "registered" : "hipHostMalloc",
               bytes, parts, best * 1e6, bytes / best * 1e-9);
      }
  return 0;
}
