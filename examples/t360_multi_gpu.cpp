// examples/t360_multi_gpu.cpp -- the native form of the multi-GPU path ("process model"): ONE process, one
// host thread + one handle + one stream per device, whole frames sharded across the devices, no data-path collective;
// with --gather the output frames of every step are collected on device 0 by RCCL (grouped ncclSend / ncclRecv on a
// side stream, overlapped with the next step through two output buffers).  bench.py does the same with one process per
// GPU under torch.distributed; this file is what an integrator who links -lTransform360 -lrccl would write.
//
//   make -C examples && examples/t360_multi_gpu [--devices N] [--frames F] [--steps K] [--gather]
//
// Workload: BASELINE config 2 (3840x1920 yuv420p -> 1536x1024 CUBEMAP_32, bicubic, low-pass off), F frames per device
// and step, resident in device memory.  Prints one line per device and the aggregate rate; the checksum of device d's
// output equals the one bench.py prints for rank d (same synthetic stream).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "Transform360/t360_device.h"

namespace {

struct FrameLayout {  // yuv420p, 64-byte aligned strides, planes 256-byte aligned (transform360_amd/handler.py:FrameLayout)
  int w[3], h[3], stride[3];
  int64_t off[3], frame_bytes;
  FrameLayout(int width, int height) {
    int64_t at = 0;
    for (int k = 0; k < 3; k++) {
      w[k] = k ? (width + 1) / 2 : width;
      h[k] = k ? (height + 1) / 2 : height;
      stride[k] = (w[k] + 63) / 64 * 64;
      off[k] = at;
      at += ((int64_t)stride[k] * h[k] + 255) / 256 * 256;
    }
    frame_bytes = at;
  }
};

#define CHECK_HIP(x)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));           \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)
#define CHECK_NCCL(x)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (x);                                                                 \
    if (r_ != ncclSuccess) {                                                               \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, ncclGetErrorString(r_));          \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

FrameTransformContext config2() {
  // the filter's defaults (vf_transform360.c:407-987) with BASELINE config 2's choices
  FrameTransformContext c;
  memset(&c, 0, sizeof(c));
  c.input_layout = LAYOUT_EQUIRECT;
  c.output_layout = LAYOUT_CUBEMAP_32;
  c.input_stereo_format = STEREO_FORMAT_MONO;
  c.output_stereo_format = STEREO_FORMAT_MONO;
  c.input_expand_coef = 1.01f;
  c.expand_coef = 1.01f;
  c.interpolation_alg = CUBIC;
  c.width_scale_factor = 1.0f;
  c.height_scale_factor = 1.0f;
  c.fixed_hfov = 120.0f;
  c.fixed_vfov = 110.0f;
  c.enable_low_pass_filter = 0;
  c.kernel_height_scale_factor = 1.0f;
  c.min_kernel_half_height = 1.0f;
  c.max_kernel_half_height = 10000.0f;
  c.enable_multi_threading = 1;
  c.num_vertical_segments = 5;
  c.num_horizontal_segments = 1;
  c.adjust_kernel = 1;
  c.kernel_adjust_factor = 1.0f;
  return c;
}

}  // namespace

int main(int argc, char** argv) {
  int ndev = T360_deviceCount(), F = 64, steps = 20;
  bool gather = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--devices") && i + 1 < argc) ndev = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) F = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--gather")) gather = true;
  }
  if (ndev < 1 || ndev > T360_deviceCount()) {
    fprintf(stderr, "no such number of devices (%d visible)\n", T360_deviceCount());
    return 1;
  }
  const FrameLayout lin(3840, 1920), lout(1536, 1024);
  T360PlaneDesc planes[3];
  for (int k = 0; k < 3; k++)
    planes[k] = T360PlaneDesc{lin.off[k], lout.off[k], lin.stride[k], lout.stride[k], lin.w[k], lin.h[k], lout.w[k], lout.h[k], k ? 1 : 0};

  std::vector<ncclComm_t> comms((size_t)ndev);
  gather = gather && ndev > 1;
  if (gather) {
    std::vector<int> devs((size_t)ndev);
    for (int d = 0; d < ndev; d++) devs[(size_t)d] = d;
    CHECK_NCCL(ncclCommInitAll(comms.data(), ndev, devs.data()));
  }
  std::vector<double> ms((size_t)ndev);
  std::vector<unsigned long long> sums((size_t)ndev);
  auto worker = [&](int d) {
    CHECK_HIP(hipSetDevice(d));  // a handle lives on the device that is current when it is created
    hipStream_t stream, side;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    FrameTransformContext ctx = config2();
    VideoFrameTransform* t = VideoFrameTransform_new(&ctx);
    if (!t || !VideoFrameTransform_generateMapForPlane(t, lin.w[0], lin.h[0], lout.w[0], lout.h[0], 0) ||
        !VideoFrameTransform_generateMapForPlane(t, lin.w[1], lin.h[1], lout.w[1], lout.h[1], 1) || !T360_setStream(t, stream)) {
      fprintf(stderr, "device %d: initialisation failed\n", d);
      exit(1);
    }
    uint8_t *d_in, *d_out[2], *sink = nullptr;
    CHECK_HIP(hipMalloc(&d_in, (size_t)F * lin.frame_bytes));
    for (int b = 0; b < 2; b++) CHECK_HIP(hipMalloc(&d_out[b], (size_t)F * lout.frame_bytes));
    if (gather && d == 0) CHECK_HIP(hipMalloc(&sink, (size_t)ndev * F * lout.frame_bytes));
    for (int j = 0; j < F; j++)  // frame j of device d = frame d*F + j of the synthetic stream (bench.py's seeds)
      T360_fillNoise(d_in + (size_t)j * lin.frame_bytes, lin.frame_bytes, (0x360ull ^ ((unsigned long long)(d * F + j) << 40)) & 0xffffffffffffffffull, stream);
    hipEvent_t done[2], sent[2];
    for (int b = 0; b < 2; b++) {
      CHECK_HIP(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
      CHECK_HIP(hipEventCreateWithFlags(&sent[b], hipEventDisableTiming));
    }
    auto step = [&](int k) {
      const int b = k & 1;
      if (gather && k >= 2) CHECK_HIP(hipStreamWaitEvent(stream, sent[b], 0));  // the gather that read this buffer is over
      if (!T360_transformFrames(t, d_in, lin.frame_bytes, d_out[b], lout.frame_bytes, F, planes, 3)) exit(1);
      if (gather) {
        CHECK_HIP(hipEventRecord(done[b], stream));
        CHECK_HIP(hipStreamWaitEvent(side, done[b], 0));
        CHECK_NCCL(ncclGroupStart());
        if (d != 0) CHECK_NCCL(ncclSend(d_out[b], (size_t)F * lout.frame_bytes, ncclUint8, 0, comms[(size_t)d], side));
        if (d == 0)
          for (int r = 1; r < ndev; r++)
            CHECK_NCCL(ncclRecv(sink + (size_t)r * F * lout.frame_bytes, (size_t)F * lout.frame_bytes, ncclUint8, r, comms[0], side));
        CHECK_NCCL(ncclGroupEnd());
        CHECK_HIP(hipEventRecord(sent[b], side));
      }
    };
    // warm-up in two parts.  (1) The clock ramp -- the first step plans the gather, and the clocks need a few hundred
    // milliseconds of load to come up -- runs the transform ALONE: its length is this thread's own wall clock, and a
    // loop of that kind must not post collectives (devices would post different numbers of sends and receives and
    // the side streams would never drain).  (2) A FIXED number of full steps, the same on every device, primes the
    // gather path and its double buffering.
    for (const auto w0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - w0 < std::chrono::milliseconds(400);) {
      for (int k = 0; k < 8; k++)
        if (!T360_transformFrames(t, d_in, lin.frame_bytes, d_out[k & 1], lout.frame_bytes, F, planes, 3)) exit(1);
      CHECK_HIP(hipStreamSynchronize(stream));
    }
    constexpr int kWarmSteps = 4;  // even: the timed loop starts on buffer 0 with both `sent` events recorded
    for (int k = 0; k < kWarmSteps; k++) step(k);
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipStreamSynchronize(side));
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; k++) step(k);
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipStreamSynchronize(side));
    ms[(size_t)d] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint8_t> host((size_t)F * lout.frame_bytes);
    CHECK_HIP(hipMemcpy(host.data(), d_out[(steps - 1) & 1], host.size(), hipMemcpyDeviceToHost));
    unsigned long long s = 0;
    for (uint8_t v : host) s += v;
    sums[(size_t)d] = s;
    VideoFrameTransform_delete(t);
    CHECK_HIP(hipFree(d_in));
    for (int b = 0; b < 2; b++) CHECK_HIP(hipFree(d_out[b]));
    if (sink) CHECK_HIP(hipFree(sink));
  };
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; d++) th.emplace_back(worker, d);
  for (auto& x : th) x.join();
  double worst = 0;
  for (int d = 0; d < ndev; d++) {
    printf("device %d: %.4f ms per step of %d frames, output checksum %llu\n", d, ms[(size_t)d] / steps, F, sums[(size_t)d]);
    worst = ms[(size_t)d] > worst ? ms[(size_t)d] : worst;
  }
  printf("%d device(s), %s: %.1f Mpix/s (%.0f frames/s)\n", ndev, gather ? "outputs gathered on device 0" : "compute only",
         (double)ndev * F * steps / (worst * 1e-3) * 1.572864, (double)ndev * F * steps / (worst * 1e-3));
  if (gather)
    for (ncclComm_t c : comms) ncclCommDestroy(c);
  return 0;
}
