// examples/t360_multi_gpu.cpp -- the native form of the multi-GPU path ("process model"): ONE process, one
// host thread + one handle + one stream per device, whole frames sharded across the devices, no data-path collective;
// with --gather the output frames of every step are collected on device 0 by RCCL (grouped ncclSend / ncclRecv on a
// side stream, overlapped with the next step through two output buffers).  bench.py does the same with one process per
// GPU under torch.distributed; this file is what an integrator who links -lTransform360 -lrccl would write.
//
//   make -C examples && examples/t360_multi_gpu [--devices N] [--workers W] [--frames F | --total-frames T] [--steps K]
//                                                  [--pipelined D] [--gather | --gather-local]
//
// --workers W > devices rehearses the W-GPU process on fewer GPUs (workers share devices; no gather then);
// --total-frames 64 is BASELINE configs[4] as written (64 frames sharded over the workers: strong scaling);
// --pipelined D issues the steps through T360_transformFramesPipelined on D internal streams per handle;
// --gather-local runs the gather of --gather -- the same per-step operation lists (t360_shard_plan.h gather_ops), the same two
// output buffers and events -- with device copies in place of ncclSend / ncclRecv, so that workers that SHARE a device (which
// RCCL refuses to put into one communicator) move real bytes through it; both forms end with a check of the sink's checksum.
//
// Workload: BASELINE config 2 (3840x1920 yuv420p -> 1536x1024 CUBEMAP_32, bicubic, low-pass off), F frames per device
// and step, resident in device memory.  Prints one line per device and the aggregate rate; the checksum of device d's
// output equals the one bench.py prints for rank d (same synthetic stream).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "Transform360/t360_device.h"
#include "t360_shard_plan.h"

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
  using namespace t360_example;
  const int visible = T360_deviceCount();
  int ndev = visible, workers = 0, F = 64, steps = 20, total_frames = 0, depth = 0, ring_mb = 320;
  bool gather = false, gather_local = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--devices") && i + 1 < argc) ndev = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--workers") && i + 1 < argc) workers = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) F = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--total-frames") && i + 1 < argc) total_frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pipelined") && i + 1 < argc) depth = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--ring-mb") && i + 1 < argc) ring_mb = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--gather")) gather = true;
    else if (!strcmp(argv[i], "--gather-local")) gather = gather_local = true;
    else {
      fprintf(stderr, "usage: t360_multi_gpu [--devices N] [--workers W] [--frames F | --total-frames T] [--steps K] [--pipelined D] [--ring-mb M] [--gather]\n");
      return 2;
    }
  }
  if (ndev < 1 || ndev > visible) {
    fprintf(stderr, "no such number of devices (%d visible)\n", visible);
    return 1;
  }
  // A worker = one host thread + handle + stream; worker w runs on device w % ndev.  Normally one worker per device; MORE
  // workers than devices is the rehearsal of the N-GPU process on a box with fewer GPUs (every code path of the N-worker
  // run except the RCCL gather, which needs distinct devices).
  if (workers <= 0) workers = ndev;
  std::vector<int> device_of((size_t)workers);
  for (int w = 0; w < workers; w++) device_of[(size_t)w] = w % ndev;
  if (gather && !gather_local && !gather_possible(device_of)) {
    if (workers == 1) {
      // one worker has nobody to gather from: the communicator is still initialised and the (empty) groups are posted, so
      // that the RCCL side of the driver can be exercised on a one-GPU box
      printf("note: --gather with one worker: RCCL is initialised, there is nothing to move\n");
    } else {
      printf("note: --gather needs one device per worker (RCCL refuses duplicate devices in a communicator): compute only\n");
      gather = false;
    }
  }
  // weak scaling (default): F frames per worker and step; --total-frames T (BASELINE configs[4]: T = 64): the T frames of
  // a step are sharded over the workers in contiguous blocks, worker w owns [lo_w, hi_w)
  std::vector<int> lo((size_t)workers), hi((size_t)workers);
  for (int w = 0; w < workers; w++) {
    if (total_frames > 0) shard_range(total_frames, w, workers, &lo[(size_t)w], &hi[(size_t)w]);
    else lo[(size_t)w] = w * F, hi[(size_t)w] = (w + 1) * F;
  }
  const FrameLayout lin(3840, 1920), lout(1536, 1024);
  T360PlaneDesc planes[3];
  for (int k = 0; k < 3; k++)
    planes[k] = T360PlaneDesc{lin.off[k], lout.off[k], lin.stride[k], lout.stride[k], lin.w[k], lin.h[k], lout.w[k], lout.h[k], k ? 1 : 0};
  std::vector<int64_t> out_bytes_of((size_t)workers);
  int64_t sink_bytes = 0;
  for (int w = 0; w < workers; w++) sink_bytes += out_bytes_of[(size_t)w] = (int64_t)(hi[(size_t)w] - lo[(size_t)w]) * lout.frame_bytes;

  std::vector<ncclComm_t> comms((size_t)workers);
  if (gather && !gather_local) CHECK_NCCL(ncclCommInitAll(comms.data(), workers, device_of.data()));
  // --gather-local: what a send / recv pair becomes when both ends may sit on one device.  Worker r announces (host side) that
  // the `done` event of its gathered step number q is recorded; worker 0 then makes its side stream wait for that event,
  // copies the frames into the sink and records `copied`; worker r waits for that event before it writes the buffer again.
  struct LocalLink {
    std::mutex mu;
    std::condition_variable cv;
    long posted = 0, pulled = 0;  // gathered steps whose `done` event is recorded / whose copy is issued
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    uint8_t* buf[2] = {nullptr, nullptr};
  };
  std::vector<LocalLink> links((size_t)workers);
  unsigned long long sink_sum = 0;
  bool sink_checked = false;
  std::vector<double> ms((size_t)workers);
  std::vector<unsigned long long> sums((size_t)workers);
  // the workers enter their timed loops TOGETHER (their warm-ups end at different times, and a worker timed while the
  // others are still warming up -- or already done -- would measure a GPU it has to itself)
  std::mutex gate_mu;
  std::condition_variable gate_cv;
  int gate_waiting = 0, gate_round = 0;
  auto gate = [&]() {
    std::unique_lock<std::mutex> lk(gate_mu);
    const int round = gate_round;
    if (++gate_waiting == workers) {
      gate_waiting = 0;
      gate_round++;
      gate_cv.notify_all();
    } else {
      gate_cv.wait(lk, [&] { return gate_round != round; });
    }
  };
  std::chrono::steady_clock::time_point t_start, t_end;
  auto worker = [&](int w) {
    const int nf = hi[(size_t)w] - lo[(size_t)w];
    CHECK_HIP(hipSetDevice(device_of[(size_t)w]));  // a handle lives on the device that is current when it is created
    hipStream_t stream, side;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    FrameTransformContext ctx = config2();
    VideoFrameTransform* t = VideoFrameTransform_new(&ctx);
    if (!t || !VideoFrameTransform_generateMapForPlane(t, lin.w[0], lin.h[0], lout.w[0], lout.h[0], 0) ||
        !VideoFrameTransform_generateMapForPlane(t, lin.w[1], lin.h[1], lout.w[1], lout.h[1], 1) || !T360_setStream(t, stream) ||
        (depth > 0 && !T360_setPipelineDepth(t, depth))) {
      fprintf(stderr, "worker %d: initialisation failed\n", w);
      exit(1);
    }
    const int nbuf = depth > 2 ? depth : 2;  // an output buffer is reused every `depth` pipelined calls
    // Input ring: when one step's input is smaller than the 256 MB Infinity Cache (8 frames = 88 MB), the steps rotate
    // through `groups` batches -- ring_mb of distinct input per worker -- so that no step finds its source in a cache
    // because the previous one read the same bytes (bench.py does the same; group 0 holds the stream's own frames).
    const int64_t step_in = (int64_t)(nf > 0 ? nf : 1) * lin.frame_bytes;
    const int groups = step_in < ((int64_t)ring_mb << 20) ? (int)((((int64_t)ring_mb << 20) + step_in - 1) / step_in) : 1;
    uint8_t *d_in, *sink = nullptr;
    std::vector<uint8_t*> d_out((size_t)nbuf);
    CHECK_HIP(hipMalloc(&d_in, (size_t)groups * step_in));
    for (int b = 0; b < nbuf; b++) CHECK_HIP(hipMalloc(&d_out[(size_t)b], (size_t)(nf > 0 ? nf : 1) * lout.frame_bytes));
    if (gather && w == 0) CHECK_HIP(hipMalloc(&sink, (size_t)sink_bytes));
    for (int j = 0; j < nf * groups; j++) {  // frame j < nf of worker w = frame lo_w + j of the synthetic stream (bench.py's seeds)
      const unsigned long long fr = j < nf ? (unsigned long long)(lo[(size_t)w] + j) : 1000000ull + (unsigned long long)w * 100000ull + (unsigned long long)j;
      T360_fillNoise(d_in + (size_t)j * lin.frame_bytes, lin.frame_bytes, 0x360ull ^ (fr << 40), stream);
    }
    int in_group = 0;  // the group the next step reads
    hipEvent_t done[2], sent[2];
    for (int b = 0; b < 2; b++) {
      CHECK_HIP(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
      CHECK_HIP(hipEventCreateWithFlags(&sent[b], hipEventDisableTiming));
    }
    const std::vector<P2POp> ops = gather ? gather_ops(w, out_bytes_of) : std::vector<P2POp>();
    if (gather_local) {
      LocalLink& me = links[(size_t)w];
      for (int b = 0; b < 2; b++) {
        CHECK_HIP(hipEventCreateWithFlags(&me.done[b], hipEventDisableTiming));
        CHECK_HIP(hipEventCreateWithFlags(&me.copied[b], hipEventDisableTiming));
        me.buf[b] = d_out[(size_t)b];
      }
      gate();  // every worker's events and buffers exist before anyone's first gathered step
    }
    long seq = 0;  // gathered steps of this worker so far (warm-up included): the same number on every worker
    auto transform = [&](uint8_t* out) {
      if (nf == 0) return;
      const uint8_t* in = d_in + (size_t)in_group * step_in;
      in_group = in_group + 1 == groups ? 0 : in_group + 1;
      const int ok = depth > 0 ? T360_transformFramesPipelined(t, in, lin.frame_bytes, out, lout.frame_bytes, nf, planes, 3)
                               : T360_transformFrames(t, in, lin.frame_bytes, out, lout.frame_bytes, nf, planes, 3);
      if (!ok) exit(1);
    };
    auto step = [&](int k) {
      if (!gather) {
        transform(d_out[(size_t)(k % nbuf)]);
        return;
      }
      if (gather_local) {
        const long q = seq++;
        const int b = buffer_of_step((int)(q & 1));
        LocalLink& me = links[(size_t)w];
        if (w != 0 && q >= 2) {
          // the copy that read this buffer (gathered step q - 2) is issued: wait for it on the device
          std::unique_lock<std::mutex> lk(me.mu);
          me.cv.wait(lk, [&] { return me.pulled >= q - 1; });
          lk.unlock();
          CHECK_HIP(hipStreamWaitEvent(stream, me.copied[b], 0));
        }
        transform(d_out[(size_t)b]);
        if (depth > 0 && !T360_pipelineJoin(t)) exit(1);
        CHECK_HIP(hipEventRecord(me.done[b], stream));
        {
          std::lock_guard<std::mutex> lk(me.mu);
          me.posted = q + 1;
        }
        me.cv.notify_all();
        for (const P2POp& op : ops) {
          if (op.send) continue;  // the sending side of a local pair is the announcement above
          LocalLink& peer = links[(size_t)op.peer];
          {
            std::unique_lock<std::mutex> lk(peer.mu);
            peer.cv.wait(lk, [&] { return peer.posted >= q + 1; });
          }
          CHECK_HIP(hipStreamWaitEvent(side, peer.done[b], 0));
          CHECK_HIP(hipMemcpyAsync(sink + op.offset, peer.buf[b], (size_t)op.bytes, hipMemcpyDeviceToDevice, side));
          CHECK_HIP(hipEventRecord(peer.copied[b], side));
          {
            std::lock_guard<std::mutex> lk(peer.mu);
            peer.pulled = q + 1;
          }
          peer.cv.notify_all();
        }
        return;
      }
      const int b = buffer_of_step(k);
      if (step_waits_for_gather(k)) CHECK_HIP(hipStreamWaitEvent(stream, sent[b], 0));  // the gather that read this buffer is over
      transform(d_out[(size_t)b]);
      if (depth > 0 && !T360_pipelineJoin(t)) exit(1);  // the send below is ordered on `stream`
      CHECK_HIP(hipEventRecord(done[b], stream));
      CHECK_HIP(hipStreamWaitEvent(side, done[b], 0));
      CHECK_NCCL(ncclGroupStart());
      for (const P2POp& op : ops) {
        if (op.send) CHECK_NCCL(ncclSend(d_out[(size_t)b], (size_t)op.bytes, ncclUint8, op.peer, comms[(size_t)w], side));
        else CHECK_NCCL(ncclRecv(sink + op.offset, (size_t)op.bytes, ncclUint8, op.peer, comms[(size_t)w], side));
      }
      CHECK_NCCL(ncclGroupEnd());
      CHECK_HIP(hipEventRecord(sent[b], side));
    };
    // warm-up in two parts.  (1) The clock ramp -- the first step plans the gather, and the clocks need a few hundred
    // milliseconds of load to come up -- runs the transform ALONE: its length is this thread's own wall clock, and a
    // loop of that kind must not post collectives (workers would post different numbers of sends and receives and
    // the side streams would never drain).  (2) A FIXED number of full steps, the same on every worker, primes the
    // gather path and its double buffering.
    for (const auto w0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - w0 < std::chrono::milliseconds(400);) {
      for (int k = 0; k < 8; k++) transform(d_out[(size_t)(k % nbuf)]);
      if (!T360_synchronize(t)) exit(1);
    }
    constexpr int kWarmSteps = 4;  // even: the timed loop starts on buffer 0 with both `sent` events recorded
    for (int k = 0; k < kWarmSteps; k++) step(k);
    if (!T360_synchronize(t)) exit(1);
    CHECK_HIP(hipStreamSynchronize(side));
    gate();
    const auto t0 = std::chrono::steady_clock::now();
    if (w == 0) t_start = t0;
    for (int k = 0; k < steps; k++) step(k);
    if (!T360_synchronize(t)) exit(1);  // the handle's stream and, with --pipelined, every lane
    CHECK_HIP(hipStreamSynchronize(side));
    ms[(size_t)w] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    gate();
    if (w == 0) t_end = std::chrono::steady_clock::now();
    // the checksum is of the stream's own frames (group 0): one more step, outside the timed region -- with a gather, a
    // gathered step into buffer 0 (`steps` and the warm-up are even), so that the sink holds exactly these frames afterwards
    in_group = 0;
    if (gather) {
      if (steps & 1) step(steps);  // keep the buffer parity: the checked step must use buffer 0
      step(steps + (steps & 1));
      if (!T360_synchronize(t)) exit(1);
      CHECK_HIP(hipStreamSynchronize(side));
      gate();  // every worker's last step, and worker 0's copies / receives, are complete
      CHECK_HIP(hipStreamSynchronize(side));
    } else if (nf > 0 && (!T360_transformFrames(t, d_in, lin.frame_bytes, d_out[0], lout.frame_bytes, nf, planes, 3) || !T360_synchronize(t))) {
      exit(1);
    }
    std::vector<uint8_t> host((size_t)nf * lout.frame_bytes);
    if (nf > 0) CHECK_HIP(hipMemcpy(host.data(), d_out[0], host.size(), hipMemcpyDeviceToHost));
    unsigned long long s = 0;
    for (uint8_t v : host) s += v;
    sums[(size_t)w] = s;
    if (gather && w == 0 && sink_bytes > out_bytes_of[0]) {
      // the frames of workers 1.. as they arrived in the sink (worker 0's own stay in its output buffer)
      std::vector<uint8_t> hs((size_t)(sink_bytes - out_bytes_of[0]));
      CHECK_HIP(hipMemcpy(hs.data(), sink + out_bytes_of[0], hs.size(), hipMemcpyDeviceToHost));
      for (uint8_t v : hs) sink_sum += v;
      sink_checked = true;
    }
    VideoFrameTransform_delete(t);
    CHECK_HIP(hipFree(d_in));
    for (int b = 0; b < nbuf; b++) CHECK_HIP(hipFree(d_out[(size_t)b]));
    if (sink) CHECK_HIP(hipFree(sink));
  };
  std::vector<std::thread> th;
  for (int w = 0; w < workers; w++) th.emplace_back(worker, w);
  for (auto& x : th) x.join();
  // the job's time: from the moment all workers are released to the moment the last one is done
  double worst = std::chrono::duration<double, std::milli>(t_end - t_start).count();
  int frames_per_step = 0;
  for (int w = 0; w < workers; w++) {
    printf("device %d (worker %d): %.4f ms per step of %d frames [%d, %d), output checksum %llu\n", device_of[(size_t)w], w,
           ms[(size_t)w] / steps, hi[(size_t)w] - lo[(size_t)w], lo[(size_t)w], hi[(size_t)w], sums[(size_t)w]);
    frames_per_step += hi[(size_t)w] - lo[(size_t)w];
  }
  printf("%d worker(s) on %d device(s), %s scaling, %s%s: %.1f Mpix/s (%.0f frames/s), %.4f ms per step of %d frames\n", workers, ndev,
         total_frames > 0 ? "strong" : "weak", gather ? "outputs gathered on worker 0" : "compute only",
         depth > 0 ? ", pipelined calls" : "", (double)frames_per_step * steps / (worst * 1e-3) * 1.572864,
         (double)frames_per_step * steps / (worst * 1e-3), worst / steps, frames_per_step);
  if (gather && !gather_local)
    for (ncclComm_t c : comms) ncclCommDestroy(c);
  if (sink_checked) {
    unsigned long long want = 0;
    for (int w = 1; w < workers; w++) want += sums[(size_t)w];
    printf("gather check (%s): the sink holds %llu, workers 1..%d computed %llu: %s\n", gather_local ? "device copies" : "RCCL send / recv",
           sink_sum, workers - 1, want, sink_sum == want ? "ok" : "MISMATCH");
    if (sink_sum != want) return 1;
  }
  return 0;
}
