// micro-benchmark: HBM read bandwidth for tiled box reads of a 3840-byte-pitch plane as a function
// of the row-fragment width (the staging pattern of the LDS-tiled gather).
// Workgroup = one box of W bytes x R rows in each of F frames; boxes tile the plane without
// overlap.  Data is summed so the loads are not eliminated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void boxread(const uint8_t* src, long frame_bytes, int pitch, int W, int R, int boxes_x,
                                               int boxes_y, int frames, unsigned* out) {
  const int b = blockIdx.x;             // box index, raster order
  const int bx = b % boxes_x, by = b / boxes_x;
  const int cpr = W / 16, nch = cpr * R;
  unsigned acc = 0;
  for (int f = 0; f < frames; f++) {
    const uint8_t* base = src + (long)f * frame_bytes + (long)(by * R) * pitch + bx * W;
    for (int q = threadIdx.x; q < nch; q += 256) {
      const int r = q / cpr, c = q - r * cpr;
      const uint4 v = *reinterpret_cast<const uint4*>(base + (long)r * pitch + c * 16);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const int pitch = 3840, H = 1920, F = 32;
  const long frame_bytes = (long)pitch * H;
  uint8_t* src; unsigned* out;
  (void)hipMalloc(&src, frame_bytes * F); (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 1, frame_bytes * F);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int Ws[] = {96, 192, 384, 768, 1920, 3840};
  for (int W : Ws) for (int fpb : {8, 32}) {
    int R = 6144 / W; if (R < 1) R = 1; if (R > 64) R = 64;      // ~6 KB boxes like the real tiles
    const int boxes_x = pitch / W, boxes_y = H / R;
    const int groups = F / fpb;
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(a);
      for (int g = 0; g < groups; g++)
        hipLaunchKernelGGL(boxread, dim3(boxes_x * boxes_y), dim3(256), 0, 0, src + (long)g * fpb * frame_bytes, frame_bytes, pitch, W, R,
                           boxes_x, boxes_y, fpb, out);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double bytes = (double)boxes_x * boxes_y * W * R * F;
    printf("fragment %4d B x %2d rows, %2d frames per workgroup: %.3f ms -> %.2f TB/s (%.0f MB)\n", W, R, fpb, best, bytes / best / 1e9, bytes / 1e6);
  }
  return 0;
}
