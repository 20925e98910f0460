// t360_transform.cpp -- host side of the handle: init-time state, pointer classification,
// staging for host buffers, kernel sequencing on a HIP stream.
//
// reference call protocol:  _new -> _generateMapForPlane(idx 0, idx 1) ->
// per frame, per plane _transformFramePlane -> _delete.
#include "t360_transform.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace t360;

namespace t360 {

bool DeviceBuffer::reserve(size_t bytes) {
  if (bytes <= bytes_) return true;
  release();
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  ptr_ = p;
  bytes_ = bytes;
  return true;
}

void DeviceBuffer::release() {
  if (ptr_) (void)hipFree(ptr_);
  ptr_ = nullptr;
  bytes_ = 0;
}

// ---- OpenCV fixed-point interpolation tables (imgwarp.cpp initInterTab1D/2D), host build ----
namespace {

void coeffs_linear(float x, float* c) {
  c[0] = 1.f - x;
  c[1] = x;
}
void coeffs_cubic(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}
void coeffs_lanczos4(float x, float* c) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                 {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) c[i] = 0;
    c[3] = 1;
    return;
  }
  const double pi = 3.1415926535897932384626433832795;
  float sum = 0;
  const double y0 = -(x + 3) * pi * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
  for (int i = 0; i < 8; i++) {
    const double y = -(x + 3 - i) * pi * 0.25;
    c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += c[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) c[i] *= sum;
}

inline int16_t round_sat_s16(float v) {
  long r = std::lrintf(v);
  return (int16_t)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

}  // namespace

bool build_inter_table(int interp, std::vector<int16_t>* tab, int* ksize_out) {
  int ks;
  void (*gen)(float, float*);
  switch (interp) {
    case LINEAR: ks = 2; gen = coeffs_linear; break;
    case CUBIC: ks = 4; gen = coeffs_cubic; break;
    case LANCZOS4: ks = 8; gen = coeffs_lanczos4; break;
    default: return false;
  }
  const int N = kInterTabSize;
  std::vector<float> t1((size_t)N * ks);
  const float scale = 1.f / N;
  for (int i = 0; i < N; i++) gen(i * scale, &t1[(size_t)i * ks]);
  tab->assign((size_t)N * N * ks * ks, 0);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      int16_t* it = tab->data() + (size_t)(i * N + j) * ks * ks;
      int isum = 0;
      for (int k1 = 0; k1 < ks; k1++) {
        const float vy = t1[(size_t)i * ks + k1];
        for (int k2 = 0; k2 < ks; k2++) {
          const float v = vy * t1[(size_t)j * ks + k2];
          isum += it[k1 * ks + k2] = round_sat_s16(v * (float)(1 << kCoefBits));
        }
      }
      if (isum != (1 << kCoefBits)) {
        // push the rounding residue onto the largest (or smallest) of four "central" taps,
        // scanning exactly the index window OpenCV scans
        const int diff = isum - (1 << kCoefBits);
        const int h = ks / 2;
        int Mk1 = h, Mk2 = h, mk1 = h, mk2 = h;
        for (int k1 = h; k1 < h + 2; k1++)
          for (int k2 = h; k2 < h + 2; k2++) {
            if (it[k1 * ks + k2] < it[mk1 * ks + mk2])
              mk1 = k1, mk2 = k2;
            else if (it[k1 * ks + k2] > it[Mk1 * ks + Mk2])
              Mk1 = k1, Mk2 = k2;
          }
        if (diff < 0)
          it[Mk1 * ks + Mk2] = (int16_t)(it[Mk1 * ks + Mk2] - diff);
        else
          it[mk1 * ks + mk2] = (int16_t)(it[mk1 * ks + mk2] - diff);
      }
    }
  if (ksize_out) *ksize_out = ks;
  return true;
}

}  // namespace t360

// -------------------------------------------------------------------------------------------

namespace {

// scoped hipSetDevice
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
      if (hipSetDevice(dev) == hipSuccess) switched = true;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

enum class PtrKind { Host, Device };

PtrKind classify(const void* p) {
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'd memory is unknown to the runtime
    return PtrKind::Host;
  }
  if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return PtrKind::Device;
  return PtrKind::Host;
}

inline bool valid_interp(int a) { return a == NEAREST || a == LINEAR || a == CUBIC || a == LANCZOS4; }

constexpr int kTileW = 64;
constexpr int kTileH = 16;

}  // namespace

VideoFrameTransform::VideoFrameTransform(const FrameTransformContext* ctx) {
  memcpy(&ctx_, ctx, sizeof(ctx_));  // the caller's block is a stack local (vf_transform360.c:111-141)
  if (hipGetDevice(&device_) != hipSuccess) {
    printf("transform360: no usable HIP device (%s)\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  if (hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking) != hipSuccess) {
    printf("transform360: could not create a HIP stream (%s)\n", hipGetErrorString(hipGetLastError()));
    return;
  }
  stream_ = own_stream_;
  if (hipDeviceGetAttribute(&cus_, hipDeviceAttributeMultiprocessorCount, device_) != hipSuccess || cus_ <= 0) cus_ = 256;
  for (int k = 0; k < 3; k++)
    if (hipStreamCreateWithFlags(&lp_streams_[k], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&lp_join_[k], hipEventDisableTiming) != hipSuccess) {
      printf("transform360: cannot create the low-pass streams\n");
      return;
    }
  if (hipEventCreateWithFlags(&lp_fork_, hipEventDisableTiming) != hipSuccess) return;
#ifdef T360_INSTRUMENT
  // tuning switches of the instrumented build (tools/ab.sh); the shipped library reads no environment
  if (const char* e = getenv("T360_RING_KB")) ring_kb_ = atoi(e);
  if (const char* e = getenv("T360_WAVES")) waves_ = atoi(e) == 8 ? 8 : 4;
  if (const char* e = getenv("T360_MAX_PIECES")) max_pieces_ = atoi(e);
  if (const char* e = getenv("T360_FRAMES_PER_BLOCK")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 4096) frames_per_block_ = v;
  }
  if (const char* e = getenv("T360_TAIL_PCT")) tail_percent_ = atoi(e);
  if (const char* e = getenv("T360_SMALL_BATCH")) small_batch_ = atoi(e);
  if (const char* e = getenv("T360_TAIL_FRAMES")) tail_frames_ = atoi(e);
  if (const char* e = getenv("T360_WIDE64")) plan_wide_pct_ = atoi(e);
  if (const char* e = getenv("T360_STRIPS")) plan_strip_pct_ = atoi(e);
  if (const char* e = getenv("T360_WIDE256")) plan_wide256_pct_ = atoi(e);
  if (const char* e = getenv("T360_SCATTER")) plan_scatter_ = atoi(e);
  if (const char* e = getenv("T360_COST_LINES")) plan_cost_lines_ = atoi(e);
  if (const char* e = getenv("T360_BAND")) plan_band_ = atoi(e);
  if (const char* e = getenv("T360_ROW_PAD")) plan_row_pad_ = atoi(e);
  if (const char* e = getenv("T360_ROW_ALIGN")) plan_row_align_ = atoi(e);
  if (getenv("T360_NO_TILED")) use_tiled_ = false;
  if (getenv("T360_NO_FAST_LOWPASS")) use_fast_lowpass_ = false;
  if (getenv("T360_NO_WIDE_LOWPASS")) use_wide_lowpass_ = false;
  if (getenv("T360_NO_MERGED_LOWPASS")) merge_lowpass_ = false;
  if (getenv("T360_FUSED_LOWPASS")) fuse_lowpass_ = atoi(getenv("T360_FUSED_LOWPASS")) != 0;
  if (getenv("T360_FUSED_SIDE_STREAM")) fuse_side_stream_ = atoi(getenv("T360_FUSED_SIDE_STREAM")) != 0;
#endif
  ok_ = true;
}

VideoFrameTransform::~VideoFrameTransform() {
  if (!ok_) return;
  DeviceGuard g(device_);
  (void)hipStreamSynchronize(stream_);
  for (int k = 0; k < 3; k++) {
    if (lp_streams_[k]) {
      (void)hipStreamSynchronize(lp_streams_[k]);
      (void)hipStreamDestroy(lp_streams_[k]);
    }
    if (lp_join_[k]) (void)hipEventDestroy(lp_join_[k]);
  }
  if (lp_fork_) (void)hipEventDestroy(lp_fork_);
  for (int k = 0; k < kMaxLanes; k++) {
    if (pipe_streams_[k]) {
      (void)hipStreamSynchronize(pipe_streams_[k]);
      (void)hipStreamDestroy(pipe_streams_[k]);
    }
    if (pipe_done_[k]) (void)hipEventDestroy(pipe_done_[k]);
  }
  if (pipe_fork_) (void)hipEventDestroy(pipe_fork_);
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

bool VideoFrameTransform::check(hipError_t e, const char* what) const {
  if (e == hipSuccess) return true;
  printf("transform360: %s failed: %s\n", what, hipGetErrorString(e));
  (void)hipGetLastError();
  return false;
}

bool VideoFrameTransform::setStream(void* s) {
  DeviceGuard g(device_);
  if (!check(hipStreamSynchronize(stream_), "hipStreamSynchronize") || !drainLanes()) return false;
  stream_ = static_cast<hipStream_t>(s);  // nullptr = HIP's NULL stream
  return true;
}

bool VideoFrameTransform::useOwnStream() {
  DeviceGuard g(device_);
  if (!check(hipStreamSynchronize(stream_), "hipStreamSynchronize") || !drainLanes()) return false;
  stream_ = own_stream_;
  return true;
}

bool VideoFrameTransform::synchronize() {
  DeviceGuard g(device_);
  return check(hipStreamSynchronize(stream_), "hipStreamSynchronize") && drainLanes();
}

// ---- T360_transformFramesPipelined: a stream of independent batches on pipe_depth_ internal streams ----
bool VideoFrameTransform::ensureLanes() {
  if (!pipe_fork_ && !check(hipEventCreateWithFlags(&pipe_fork_, hipEventDisableTiming), "hipEventCreate")) return false;
  for (int k = 0; k < pipe_depth_; k++) {
    if (!pipe_streams_[k] && !check(hipStreamCreateWithFlags(&pipe_streams_[k], hipStreamNonBlocking), "hipStreamCreate")) return false;
    if (!pipe_done_[k] && !check(hipEventCreateWithFlags(&pipe_done_[k], hipEventDisableTiming), "hipEventCreate")) return false;
  }
  return true;
}

// Per-map tables that depend on a call's plane size (low-pass tile lists, INTER_AREA tables) are rebuilt in place when a
// later call names the map with another size; with pipelined calls in flight a lane may still be reading the old ones.
bool VideoFrameTransform::quiesceLanes() {
  bool any = false;
  for (int k = 0; k < kMaxLanes; k++) any = any || (pipe_streams_[k] && pipe_busy_[k]);
  return !any || check(hipDeviceSynchronize(), "hipDeviceSynchronize");
}

bool VideoFrameTransform::drainLanes() {
  bool ok = true;
  for (int k = 0; k < kMaxLanes; k++)
    if (pipe_streams_[k] && pipe_busy_[k]) {
      ok = check(hipStreamSynchronize(pipe_streams_[k]), "hipStreamSynchronize") && ok;
      pipe_busy_[k] = false;
    }
  return ok;
}

bool VideoFrameTransform::setPipelineDepth(int depth) {
  if (!ok_) return false;
  if (depth < 1 || depth > kMaxLanes) {
    printf("transform360: T360_setPipelineDepth: depth must be 1..%d\n", kMaxLanes);
    return false;
  }
  DeviceGuard g(device_);
  // the lane of call k is k mod depth: changing the modulus in mid-sequence would break "an output buffer may be reused
  // every depth calls", so the sequence ends here
  if (!drainLanes()) return false;
  pipe_depth_ = depth;
  pipe_next_ = 0;
  return true;
}

bool VideoFrameTransform::pipelineJoin() {
  if (!ok_) return false;
  DeviceGuard g(device_);
  for (int k = 0; k < kMaxLanes; k++)
    if (pipe_streams_[k] && pipe_busy_[k]) {
      if (!check(hipEventRecord(pipe_done_[k], pipe_streams_[k]), "hipEventRecord") ||
          !check(hipStreamWaitEvent(stream_, pipe_done_[k], 0), "hipStreamWaitEvent"))
        return false;
      // pipe_busy_[k] stays set: the join is a DEVICE-side wait, the lane may still be executing, and quiesceLanes() /
      // drainLanes() use the flag to decide whether anything can still read the per-map tables they are about to rewrite
      // (ADVICE round 5).  Only a host-side wait (drainLanes, synchronize) clears it; a second join re-records the event.
    }
  pipe_next_ = 0;
  return true;
}

bool VideoFrameTransform::transformFramesPipelined(const uint8_t* d_in, int64_t in_frame_bytes, uint8_t* d_out,
                                                   int64_t out_frame_bytes, int n_frames, const T360PlaneDesc* planes,
                                                   int n_planes) {
  if (!ok_) return false;
  DeviceGuard g(device_);
  if (!ensureLanes()) return false;
  const int lane = pipe_next_;
  hipStream_t ls = pipe_streams_[lane];
  // ordering IN: the lane waits (on the device) for what is queued on the handle's stream right now.  An idle stream has
  // nothing to wait for, and the record + wait pair is not free: the lane's launch then sits behind a barrier packet that
  // resolves ~9 us later (8-frame steps on ONE lane: 0.0370 ms plain, 0.0456 with the pair in front of every launch).
  const hipError_t busy = hipStreamQuery(stream_);
  if (busy == hipErrorNotReady) {
    if (!check(hipEventRecord(pipe_fork_, stream_), "hipEventRecord") ||
        !check(hipStreamWaitEvent(ls, pipe_fork_, 0), "hipStreamWaitEvent"))
      return false;
  } else if (busy != hipSuccess) {
    return check(busy, "hipStreamQuery");
  }
  // the call runs as a plain one whose stream and scratch set are the lane's; restored on every way out (the vectors of
  // runPlanes may throw, and the C entry point turns that into a 0)
  struct Restore {
    VideoFrameTransform* t;
    hipStream_t saved;
    ~Restore() {
      t->stream_ = saved;
      t->scratch_ = 0;
    }
  } restore{this, stream_};
  stream_ = ls;
  scratch_ = 1 + lane;
  pipe_busy_[lane] = true;
  const bool ok = transformFrames(d_in, in_frame_bytes, d_out, out_frame_bytes, n_frames, planes, n_planes);
  pipe_next_ = (lane + 1) % pipe_depth_;
  return ok;
}

bool VideoFrameTransform::ensureWeights() {
  if (weights_ready_) return true;
  const int interp = (int)ctx_.interpolation_alg;
  if (interp == NEAREST || !valid_interp(interp)) {
    // NEAREST needs no table; an unknown code never reaches the gather (runPlane prints and
    // returns true like the reference's default: branch)
    weights_ready_ = true;
    return true;
  }
  std::vector<int16_t> tab;
  int ks = 0;
  if (!build_inter_table(interp, &tab, &ks)) return false;
  if (!weights_.reserve(tab.size() * sizeof(int16_t))) return check(hipErrorOutOfMemory, "hipMalloc(weights)");
  if (!check(hipMemcpy(weights_.as<void>(), tab.data(), tab.size() * sizeof(int16_t), hipMemcpyHostToDevice),
             "hipMemcpy(weights)"))
    return false;
  {
    std::vector<uint32_t> pack;
    pack_weights(tab, ks, &pack);
    if (!weights_pack_.reserve(pack.size() * sizeof(uint32_t))) return check(hipErrorOutOfMemory, "hipMalloc(weights)");
    if (!check(hipMemcpy(weights_pack_.as<void>(), pack.data(), pack.size() * sizeof(uint32_t), hipMemcpyHostToDevice),
               "hipMemcpy(weights)"))
      return false;
  }
  weights_ready_ = true;
  return true;
}

// reference VideoFrameTransform::generateMapForPlane (VideoFrameTransform.cpp:504-576)
bool VideoFrameTransform::generateMapForPlane(int inputWidth, int inputHeight, int outputWidth,
                                              int outputHeight, int idx) {
  if (!ok_) return false;
  if (idx < 0 || idx >= kMaxMaps) {
    printf("Could not generate map for plane %d. Error: plane index out of range\n", idx);
    return false;
  }
  // the reference asserts these only in debug builds (:511-520); the GPU path needs them true
  if (inputWidth <= 0 || inputHeight <= 0 || outputWidth <= 0 || outputHeight <= 0 ||
      !(ctx_.width_scale_factor > 0) || !(ctx_.height_scale_factor > 0)) {
    printf("Could not generate map for plane %d. Error: invalid plane size\n", idx);
    return false;
  }
  if (inputWidth > 32767 || inputHeight > 32767) {
    // cv::remap keeps source coordinates as shorts; larger planes are outside its contract
    printf("Could not generate map for plane %d. Error: input plane larger than 32767\n", idx);
    return false;
  }
  const int olay = (int)ctx_.output_layout;
  if (olay < 0 || olay >= LAYOUT_N) {
    // reference transformPos default: branch (:1080-1083) -> generateMapForPlane fails (:539-543)
    printf("Invalid layout type.\nFailed to find the mapping coordinate for point (0, 0)\n");
    return false;
  }
  if (ctx_.enable_low_pass_filter &&
      !(ctx_.num_vertical_segments >= 1 && ctx_.num_horizontal_segments >= 1 &&
        ctx_.kernel_height_scale_factor > 0)) {
    printf("Could not generate map for plane %d. Error: invalid low-pass segment counts\n", idx);
    return false;
  }
  DeviceGuard g(device_);
  if (!quiesceLanes()) return false;  // a pipelined call on another lane may still be using this index's map and tables
  PlaneState& p = planes_[idx];

  MapGenParams P;
  memset(&P, 0, sizeof(P));
  const double mw = (double)ctx_.width_scale_factor * outputWidth + 0.5, mh = (double)ctx_.height_scale_factor * outputHeight + 0.5;
  if (!(mw >= 1.0 && mh >= 1.0 && mw < 32768.0 && mh < 32768.0)) {  // also false for NaN / infinite factors
    printf("Could not generate map for plane %d. Error: warp map size outside 1 .. 32767\n", idx);
    return false;
  }
  P.map_w = (int)(ctx_.width_scale_factor * outputWidth + 0.5);    // :524
  P.map_h = (int)(ctx_.height_scale_factor * outputHeight + 0.5);  // :525-526
  if (P.map_w <= 0 || P.map_h <= 0) return false;
  if ((int64_t)P.map_w * P.map_h > ((int64_t)1 << 28)) {
    // 16K x 16K: the map, its sample LUT and the host copy the planner works on are 6 GB together at this size; the
    // reference would try (and die of) an allocation of any size
    printf("Could not generate map for plane %d. Error: warp map larger than 2^28 entries\n", idx);
    return false;
  }
  // every argument check has passed (a call refused by them leaves the map of this index as it was): from here on the
  // state is rewritten, and a failure below must not leave the previous map half replaced
  p.valid = false;
  P.in_w = inputWidth;
  P.in_h = inputHeight;
  P.input_layout = (int)ctx_.input_layout;
  P.output_layout = olay;
  P.input_stereo = (int)ctx_.input_stereo_format;
  P.output_stereo = (int)ctx_.output_stereo_format;
  P.vflip = ctx_.vflip;
  P.interp = (int)ctx_.interpolation_alg;
  P.expand_coef = ctx_.expand_coef;
  P.input_expand_coef = ctx_.input_expand_coef;
  P.off_x = ctx_.fixed_cube_offcenter_x;
  P.off_y = ctx_.fixed_cube_offcenter_y;
  P.off_z = ctx_.fixed_cube_offcenter_z;
  P.offcenter = std::fabs(P.off_x) > 1e-9 || std::fabs(P.off_y) > 1e-9 || std::fabs(P.off_z) > 1e-9;  // :1192-1194
  P.horizontal_offset = ctx_.is_horizontal_offset;
  P.hfov = ctx_.fixed_hfov;
  P.vfov = ctx_.fixed_vfov;
  P.yaw_deg = ctx_.fixed_yaw;
  P.pitch_deg = ctx_.fixed_pitch;
  P.input_pixel_width = 1.0f / inputWidth;  // :528-531
  if (ctx_.input_stereo_format == STEREO_FORMAT_LR) P.input_pixel_width *= 2;
  {
    // :1233-1244 -- double sin/cos of (deg * M_PI / 180.0f) narrowed to float, then the float
    // coefficient expressions in the reference's association order
    const float s1 = (float)std::sin(ctx_.fixed_yaw * M_PI / 180.0f);
    const float s2 = (float)std::sin(ctx_.fixed_pitch * M_PI / 180.0f);
    const float s3 = (float)std::sin(ctx_.fixed_roll * M_PI / 180.0f);
    const float c1 = (float)std::cos(ctx_.fixed_yaw * M_PI / 180.0f);
    const float c2 = (float)std::cos(ctx_.fixed_pitch * M_PI / 180.0f);
    const float c3 = (float)std::cos(ctx_.fixed_roll * M_PI / 180.0f);
    P.rot[0] = c1 * c3 + s1 * s2 * s3;
    P.rot[1] = c3 * s1 * s2 - c1 * s3;
    P.rot[2] = c2 * s1;
    P.rot[3] = c2 * s3;
    P.rot[4] = c2 * c3;
    P.rot[5] = -s2;
    P.rot[6] = c1 * s2 * s3 - c3 * s1;
    P.rot[7] = c1 * c3 * s2 + s1 * s3;
    P.rot[8] = c1 * c2;
  }

  // Column / row tables of the layouts whose per-pixel libm calls depend on one coordinate only
  // (t360_internal.h, MapGenParams::col_tab).  The front end below is transformPos :903-938 for
  // one coordinate; every expression keeps the reference's types (float * M_PI is double).
  if (olay == LAYOUT_EQUIRECT || olay == LAYOUT_BARREL || olay == LAYOUT_BARREL_SPLIT || olay == LAYOUT_EAC_32) {
    const bool stereo_in = ctx_.input_stereo_format != STEREO_FORMAT_MONO;
    auto front_x = [&](int j) {
      float x = ((float)j + 0.5f) / (float)P.map_w;
      if (stereo_in && ctx_.output_stereo_format == STEREO_FORMAT_LR) x = x > 0.5f ? (x - 0.5f) / 0.5f : x / 0.5f;
      return x;
    };
    auto front_y = [&](int i) {
      float y = ((float)i + 0.5f) / (float)P.map_h;
      if (stereo_in && ctx_.output_stereo_format == STEREO_FORMAT_TB) {
        if (y > 0.5f) {
          y = (y - 0.5f) / 0.5f;
          if (ctx_.vflip) y = 1.0f - y;
        } else {
          y = y / 0.5f;
        }
      }
      return 1.0f - y;  // :936-938
    };
    const float e = ctx_.expand_coef;
    std::vector<float> col, row;
    if (olay == LAYOUT_EAC_32) {  // :1016-1027
      col.resize((size_t)P.map_w);
      row.resize((size_t)P.map_h);
      for (int j = 0; j < P.map_w; j++) {
        float x = front_x(j);
        const int hFace = (int)(x * 3);
        x = x * 3.0f - hFace;
        col[(size_t)j] = (float)(std::tan((x - 0.5f) * M_PI * 0.5f) * 0.5f + 0.5f);
      }
      for (int i = 0; i < P.map_h; i++) {
        float y = front_y(i);
        const int vFace = (int)(y * 2);
        y = y * 2.0f - vFace;
        row[(size_t)i] = (float)(std::tan((y - 0.5f) * M_PI * 0.5f) * 0.5f + 0.5f);
      }
    } else {
      const int halves = olay == LAYOUT_BARREL_SPLIT ? 2 : 1;
      col.resize((size_t)P.map_w * 2 * halves);
      row.resize((size_t)P.map_h * 2);
      for (int h = 0; h < halves; h++)
        for (int j = 0; j < P.map_w; j++) {
          const float x = front_x(j);
          float yaw;
          if (olay == LAYOUT_EQUIRECT)
            yaw = (float)((2.0f * x - 1.0f) * M_PI);  // :962
          else if (olay == LAYOUT_BARREL)
            yaw = (float)((2.5f * x - 1.0f) * e * M_PI);  // :967
          else
            yaw = (float)(((3.0f / 2.0f * x - 0.5f) * e - h + 1.0f) * M_PI);  // :981, h = vFace
          const size_t o = ((size_t)h * P.map_w + j) * 2;
          col[o] = sinf(yaw);  // :1094-1097: float arguments -> the float overloads
          col[o + 1] = cosf(yaw);
        }
      for (int i = 0; i < P.map_h; i++) {
        const float y = front_y(i);
        float pitch;
        if (olay == LAYOUT_EQUIRECT) {
          pitch = (float)((y - 0.5f) * M_PI);  // :963
        } else if (olay == LAYOUT_BARREL) {
          pitch = (float)((y * 0.5f - 0.25f) * e * M_PI);  // :968
        } else {
          const int vFace = (int)(y * 2);
          pitch = (float)((y - 0.25f - 0.5f * vFace) * e * M_PI);  // :982
        }
        row[(size_t)i * 2] = sinf(pitch);
        row[(size_t)i * 2 + 1] = cosf(pitch);
      }
    }
    if (!p.col_tab.reserve(col.size() * sizeof(float)) || !p.row_tab.reserve(row.size() * sizeof(float)))
      return check(hipErrorOutOfMemory, "hipMalloc(tables)");
    if (!check(hipMemcpyAsync(p.col_tab.as<void>(), col.data(), col.size() * sizeof(float), hipMemcpyHostToDevice, stream_),
               "hipMemcpy(tables)") ||
        !check(hipMemcpyAsync(p.row_tab.as<void>(), row.data(), row.size() * sizeof(float), hipMemcpyHostToDevice, stream_),
               "hipMemcpy(tables)") ||
        !check(hipStreamSynchronize(stream_), "hipStreamSynchronize"))
      return false;
    P.col_tab = p.col_tab.as<float>();
    P.row_tab = p.row_tab.as<float>();
  }

  const size_t n = (size_t)P.map_w * (size_t)P.map_h;
  if (!p.map.reserve(n * sizeof(float2)) || !p.lut.reserve(n * sizeof(LutEntry)))
    return check(hipErrorOutOfMemory, "hipMalloc(map)");
  if (!check(launch_mapgen(P, p.map.as<float2>(), p.lut.as<LutEntry>(), stream_), "mapgen launch")) return false;
  if (!ensureWeights()) return false;
  if (!buildGatherPlan(p, P, inputWidth, inputHeight)) return false;

  p.in_w = inputWidth;
  p.in_h = inputHeight;
  p.out_w = outputWidth;
  p.out_h = outputHeight;
  p.map_w = P.map_w;
  p.map_h = P.map_h;
  p.lp.tiles_w = p.lp.tiles_h = p.lp_part.tiles_w = p.lp_part.tiles_h = -1;
  p.lp.ntiles = p.lp_part.ntiles = 0;
  p.fuse_ok = false;
  p.filter.segments.clear();

  if (ctx_.enable_low_pass_filter) {
    // the reference appends on a repeated call (:237, :290-294); the duplicates only redo the
    // same work, so replacing gives identical output
    if (!build_filter_config(ctx_, inputWidth, inputHeight, P.map_w, P.map_h, &p.filter)) return false;
    std::vector<SegmentDev> segs;
    std::vector<int> q8;
    std::vector<float> f32;
    std::vector<uint32_t> pk, sh;
    p.seg_fast.clear();
    p.fast_ky = 0;
    for (const Segment& s : p.filter.segments) {
      SegmentDev d;
      // register-only Q8 kernel (t360_lowpass.hip): fixed-point segment, horizontal taps that fit
      // a byte (<= 64 of them), 3/5/7 vertical taps shared by all eligible segments of the plane
      bool fast = s.fixed_point && s.kx_q8.size() <= 64 && (s.ky_q8.size() == 3 || s.ky_q8.size() == 5 || s.ky_q8.size() == 7) &&
                  (p.fast_ky == 0 || p.fast_ky == (int)s.ky_q8.size()) && use_fast_lowpass_;
      for (int v : s.kx_q8) fast = fast && v >= 0 && v <= 255;
      d.kxp_off = (int)pk.size();
      d.kx_groups = 0;
      if (fast) {
        p.fast_ky = (int)s.ky_q8.size();
        d.kx_groups = ((int)s.kx_q8.size() + 3) / 4;
        for (int grp = 0; grp < d.kx_groups; grp++) {
          uint32_t w = 0;
          for (int b = 0; b < 4; b++) {
            const size_t k = (size_t)grp * 4 + b;
            if (k < s.kx_q8.size()) w |= (uint32_t)s.kx_q8[k] << (8 * b);
          }
          pk.push_back(w);
        }
      }
      // wide fast path: the row pass of output pixel px0 + j (px0 % 4 == 0, j = 0..3) is a dot product of the
      // ALIGNED source dwords from (px0 - rx - m) on, m = (-rx) & 3, with the taps shifted by m + j bytes -- four
      // variants of the packed taps instead of a realignment of the pixels per output pixel
      d.kxs_off = (int)sh.size();
      d.kxs_nd = fast ? pack_shifted_taps(s.kx_q8, &sh) : 0;
      p.seg_fast.push_back(fast ? (d.kxs_nd ? 2 : 1) : 0);
      d.left = s.left;
      d.top = s.top;
      d.width = s.width;
      d.height = s.height;
      d.kx_off = (int)q8.size();
      d.kx_len = (int)s.kx.size();
      q8.insert(q8.end(), s.kx_q8.begin(), s.kx_q8.end());
      f32.insert(f32.end(), s.kx.begin(), s.kx.end());
      d.ky_off = (int)q8.size();
      d.ky_len = (int)s.ky.size();
      q8.insert(q8.end(), s.ky_q8.begin(), s.ky_q8.end());
      f32.insert(f32.end(), s.ky.begin(), s.ky.end());
      d.fixed_point = s.fixed_point ? 1 : 0;
      segs.push_back(d);
    }
    if (!segs.empty()) {
      if (pk.empty()) pk.push_back(0);
      if (sh.empty()) sh.push_back(0);
      if (!p.taps_sh.reserve(sh.size() * sizeof(uint32_t))) return check(hipErrorOutOfMemory, "hipMalloc(segments)");
      if (!check(hipMemcpyAsync(p.taps_sh.as<void>(), sh.data(), sh.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                stream_), "hipMemcpy(taps)"))
        return false;
      if (!p.segs.reserve(segs.size() * sizeof(SegmentDev)) || !p.taps_q8.reserve(q8.size() * sizeof(int)) ||
          !p.taps_f32.reserve(f32.size() * sizeof(float)) || !p.taps_pk.reserve(pk.size() * sizeof(uint32_t)))
        return check(hipErrorOutOfMemory, "hipMalloc(segments)");
      if (!check(hipMemcpyAsync(p.taps_pk.as<void>(), pk.data(), pk.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                stream_), "hipMemcpy(taps)"))
        return false;
      if (!check(hipMemcpyAsync(p.segs.as<void>(), segs.data(), segs.size() * sizeof(SegmentDev),
                                hipMemcpyHostToDevice, stream_), "hipMemcpy(segments)") ||
          !check(hipMemcpyAsync(p.taps_q8.as<void>(), q8.data(), q8.size() * sizeof(int),
                                hipMemcpyHostToDevice, stream_), "hipMemcpy(taps)") ||
          !check(hipMemcpyAsync(p.taps_f32.as<void>(), f32.data(), f32.size() * sizeof(float),
                                hipMemcpyHostToDevice, stream_), "hipMemcpy(taps)"))
        return false;
    }
    // fused low-pass tiles: the kernels a gather tile can apply to its own footprint, and which source rows they cover
    std::vector<uint32_t> ftaps;
    p.fuse_ok = fuse_lowpass_ && build_fuse_info(ctx_, p.filter, inputWidth, inputHeight, &p.fuse_info, &ftaps);
    if (p.fuse_ok) {
      if (!p.fuse_taps.reserve(ftaps.size() * sizeof(uint32_t))) return check(hipErrorOutOfMemory, "hipMalloc(fused taps)");
      if (!check(hipMemcpyAsync(p.fuse_taps.as<void>(), ftaps.data(), ftaps.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                stream_), "hipMemcpy(fused taps)") ||
          !check(hipStreamSynchronize(stream_), "hipStreamSynchronize"))
        return false;
    }
  }
  // host vectors above go out of scope: finish the uploads (init-time only)
  if (!check(hipStreamSynchronize(stream_), "hipStreamSynchronize")) return false;
  if (!buildResizePlan(p, p.out_w, p.out_h)) return false;
  p.valid = true;
  return true;
}

// cv::resize(..., INTER_AREA) from the warp-map size to the output size (reference :770-776);
// which of OpenCV's code paths applies and, for fractional factors, its DecimateAlpha tables
// (resize.cpp computeResizeAreaTab, evaluated in double like OpenCV does).
bool VideoFrameTransform::buildResizePlan(PlaneState& p, int dw, int dh) {
  PlaneState::ResizePlan& r = p.resize;
  if (r.dw >= 0 && !quiesceLanes()) return false;  // tables of a previous target size may be in use on another lane
  r.dw = dw;
  r.dh = dh;
  r.needed = p.map_w != dw || p.map_h != dh;
  r.supported = false;
  r.iscale_x = r.iscale_y = 0;
  if (!r.needed) return true;
  const int sw = p.map_w, sh = p.map_h;
  if (dw <= 0 || dh <= 0) return true;
  const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
  r.supported = true;
  r.linear = false;
  if (!(scale_x >= 1 && scale_y >= 1)) {
    // an enlarging factor: OpenCV emulates INTER_AREA with its bilinear kernels (resize.cpp, ksize = 2, area_mode
    // coefficients, 11-bit fixed point); tables in OpenCV's own float / double expressions
    r.linear = true;
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    std::vector<int> xs((size_t)dw), xa((size_t)dw * 2), ys((size_t)dh), ya((size_t)dh * 2);
    auto coef = [](float c) { long v = std::lrintf(c * 2048.f); return (int)(v > 32767 ? 32767 : v); };
    r.xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
      int sx = (int)std::floor(dx * scale_x);
      float fx = (float)((dx + 1) - (sx + 1) * inv_x);
      fx = fx <= 0 ? 0.f : fx - std::floor(fx);
      if (sx + 1 >= sw) {
        r.xmax = std::min(r.xmax, dx);
        if (sx >= sw - 1) fx = 0, sx = sw - 1;
      }
      xs[(size_t)dx] = sx;
      xa[(size_t)2 * dx] = coef(1.f - fx);
      xa[(size_t)2 * dx + 1] = coef(fx);
    }
    for (int dy = 0; dy < dh; dy++) {
      const int sy = (int)std::floor(dy * scale_y);
      float fy = (float)((dy + 1) - (sy + 1) * inv_y);
      fy = fy <= 0 ? 0.f : fy - std::floor(fy);
      ys[(size_t)dy] = sy;
      ya[(size_t)2 * dy] = coef(1.f - fy);
      ya[(size_t)2 * dy + 1] = coef(fy);
    }
    auto up = [&](t360::DeviceBuffer& b, const void* src, size_t bytes) {
      if (!b.reserve(bytes ? bytes : 4)) return check(hipErrorOutOfMemory, "hipMalloc(resize tables)");
      return bytes == 0 || check(hipMemcpyAsync(b.as<void>(), src, bytes, hipMemcpyHostToDevice, stream_), "hipMemcpy(resize tables)");
    };
    if (!up(r.x_si, xs.data(), xs.size() * 4) || !up(r.xofs, xa.data(), xa.size() * 4) || !up(r.y_si, ys.data(), ys.size() * 4) ||
        !up(r.yofs, ya.data(), ya.size() * 4) || !r.x_alpha.reserve(4) || !r.y_alpha.reserve(4))
      return false;
    return check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  }
  const int ix = (int)std::lrint(scale_x), iy = (int)std::lrint(scale_y);  // saturate_cast<int>(double)
  if (std::fabs(scale_x - ix) < DBL_EPSILON && std::fabs(scale_y - iy) < DBL_EPSILON && (int64_t)ix * dw == sw &&
      (int64_t)iy * dh == sh) {
    r.iscale_x = ix;
    r.iscale_y = iy;
    return true;
  }
  auto build = [](int ssize, int dsize, double scale, std::vector<int>* ofs, std::vector<int>* si, std::vector<float>* al) {
    ofs->assign((size_t)dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
      (*ofs)[(size_t)dx] = (int)si->size();
      const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
      const double cellWidth = std::min(scale, ssize - fsx1);
      int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
      sx2 = std::min(sx2, ssize - 1);
      sx1 = std::min(sx1, sx2);
      if (sx1 - fsx1 > 1e-3) {
        si->push_back(sx1 - 1);
        al->push_back((float)((sx1 - fsx1) / cellWidth));
      }
      for (int sx = sx1; sx < sx2; sx++) {
        si->push_back(sx);
        al->push_back((float)(1.0 / cellWidth));
      }
      if (fsx2 - sx2 > 1e-3) {
        si->push_back(sx2);
        al->push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth));
      }
    }
    (*ofs)[(size_t)dsize] = (int)si->size();
  };
  std::vector<int> xofs, xsi, yofs, ysi;
  std::vector<float> xal, yal;
  build(sw, dw, scale_x, &xofs, &xsi, &xal);
  build(sh, dh, scale_y, &yofs, &ysi, &yal);
  auto up = [&](t360::DeviceBuffer& b, const void* src, size_t bytes) {
    if (!b.reserve(bytes ? bytes : 4)) return check(hipErrorOutOfMemory, "hipMalloc(resize tables)");
    return bytes == 0 || check(hipMemcpyAsync(b.as<void>(), src, bytes, hipMemcpyHostToDevice, stream_), "hipMemcpy(resize tables)");
  };
  if (!up(r.xofs, xofs.data(), xofs.size() * 4) || !up(r.x_si, xsi.data(), xsi.size() * 4) ||
      !up(r.x_alpha, xal.data(), xal.size() * 4) || !up(r.yofs, yofs.data(), yofs.size() * 4) ||
      !up(r.y_si, ysi.data(), ysi.size() * 4) || !up(r.y_alpha, yal.data(), yal.size() * 4))
    return false;
  return check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

// b directly continues a to the right with the same band and bit-identical kernels
static bool same_run(const t360::Segment& a, const t360::Segment& b) {
  return a.top == b.top && a.height == b.height && a.left + a.width == b.left && a.fixed_point == b.fixed_point &&
         a.kx_q8 == b.kx_q8 && a.ky_q8 == b.ky_q8;
}

// Tile work list of the low-pass for a plane of w x h (reference filterPlane's segment loop,
// VideoFrameTransform.cpp:630-691: every segment once per eye).
bool VideoFrameTransform::ensureTiles(PlaneState& p, PlaneState::LowpassLists& l, const std::vector<uint8_t>* needed, int w, int h,
                                      int imagePlaneIndex) {
  // `needed` (one flag per segment, MONO inputs) restricts the lists to the segments some unfused gather tile still reads;
  // a handle's needed set belongs to its long-batch plan and never changes between calls
  if (l.tiles_w == w && l.tiles_h == h) return true;
  auto wanted = [&](size_t i) { return needed == nullptr || (i < needed->size() && (*needed)[i]); };
  // the lists are about to be rewritten in place: a call still running on another pipeline lane may be reading them
  if (l.tiles_w >= 0 && !quiesceLanes()) return false;
  std::vector<LowpassTile> tiles, fast_tiles, rest_tiles, wide_tiles;
  int max_rows_rest = 0, fast_lds = 0, wide_lds = 0;
  int ox[2] = {0, 0}, oy[2] = {0, 0}, eyes = 1;
  if (ctx_.input_stereo_format == STEREO_FORMAT_LR) {
    eyes = 2;
    ox[1] = (int)(0.5 * w);
  } else if (ctx_.input_stereo_format == STEREO_FORMAT_TB) {
    eyes = 2;
    oy[1] = (int)(0.5 * h);
  }
  int64_t covered = 0;
  int max_rows = 0;
  for (int e = 0; e < eyes; e++)
    for (size_t i = 0; i < p.filter.segments.size(); i++) {
      const Segment& s = p.filter.segments[i];
      if (!wanted(i)) continue;
      const int L = s.left + ox[e], T = s.top + oy[e];
      if (L < 0 || T < 0 || s.width < 0 || s.height < 0 || L + s.width > w || T + s.height > h) {
        // cv::Mat::operator()(Rect) throws; filterSegment prints and carries on (:198-203)
        printf("Could not filter segment for the plane %d. Error: segment outside the plane\n", imagePlaneIndex);
        continue;
      }
      covered += (int64_t)s.width * s.height;
      const int ry = (int)s.ky.size() / 2;
      int th = kTileH;
      const int rows_cap = (160 * 1024) / (kTileW * (int)sizeof(int));
      if (2 * ry + 1 > rows_cap) {
        printf("Could not filter plane %d. Error: vertical kernel of %d taps exceeds the LDS budget\n",
               imagePlaneIndex, (int)s.ky.size());
        return false;
      }
      if (th + 2 * ry > rows_cap) th = rows_cap - 2 * ry;
      for (int y = 0; y < s.height; y += th)
        for (int x = 0; x < s.width; x += kTileW) {
          LowpassTile t;
          t.seg = (int)i;
          t.x0 = L + x;
          t.y0 = T + y;
          t.w = std::min(kTileW, s.width - x);
          t.h = std::min(th, s.height - y);
          tiles.push_back(t);
          max_rows = std::max(max_rows, t.h + 2 * ry);
          if (!p.seg_fast[i]) {
            rest_tiles.push_back(t);
            max_rows_rest = std::max(max_rows_rest, t.h + 2 * ry);
          }
        }
      if (p.seg_fast[i]) {
        // Horizontally adjacent segments of a band with IDENTICAL kernels are filtered as one run:
        // sepFilter2D on an ROI reads the parent's real pixels beyond the ROI, so with equal kernels
        // the result does not depend on where the ROI borders are.  Runs are cut into tiles of
        // <= 128 x 128 px: 8 row groups of 32 lanes x 4 px per workgroup.
        const size_t n = p.filter.segments.size();
        const bool continues_prev = i > 0 && wanted(i - 1) && p.seg_fast[i - 1] && same_run(p.filter.segments[i - 1], s);
        if (!continues_prev) {
          // the run = this segment and the members that follow it AND still fit the plane (a plane smaller than the one
          // the segments were computed for -- the filter's alpha-plane quirk -- keeps the members that fit, like the
          // reference, which range-checks every segment on its own)
          int run_w = s.width;
          for (size_t k = i + 1; k < n && wanted(k) && p.seg_fast[k] && same_run(p.filter.segments[k - 1], p.filter.segments[k]) &&
                                 L + run_w + p.filter.segments[k].width <= w;
               k++)
            run_w += p.filter.segments[k].width;
          if (p.seg_fast[i] == 2 && L % 4 == 0 && w % 16 == 0 && use_wide_lowpass_) {
            // wide tiles, row-major: <= 512 x 32 px (2 row groups of 128 lanes x 4 px)
            const int nd = ((int)s.kx_q8.size() + (4 - ((int)s.kx_q8.size() / 2) % 4) % 4 + 6) / 4;
            for (int y = 0; y < s.height; y += kWideTileH)
              for (int x = 0; x < run_w; x += kWideTileW) {
                LowpassTile t;
                t.seg = (int)i;
                t.x0 = L + x;
                t.y0 = T + y;
                t.w = std::min(kWideTileW, run_w - x);
                t.h = std::min(kWideTileH, s.height - y);
                wide_tiles.push_back(t);
                const int ndw = 3 + (t.w + 3) / 4 + nd - 1;  // the staged rectangle starts 16-byte aligned: <= 3 dwords early
                wide_lds = std::max(wide_lds, (t.h + 2 * ry) * ((ndw + 3) & ~3) * 4);
              }
            continue;
          }
          for (int y = 0; y < s.height; y += 128)
            for (int x = 0; x < run_w; x += 128) {
              LowpassTile t;
              t.seg = (int)i;
              t.x0 = L + x;
              t.y0 = T + y;
              t.w = std::min(128, run_w - x);
              t.h = std::min(128, s.height - y);
              fast_tiles.push_back(t);
              // staged rectangle: (h + 2 ry) rows of ceil(w/4) + groups + 1 dwords, 16-byte row pitch
              const int ndw = (t.w + 3) / 4 + ((int)s.kx_q8.size() + 3) / 4 + 1;
              fast_lds = std::max(fast_lds, (t.h + 2 * ry) * ((ndw + 3) & ~3) * 4);
            }
        }
      }
    }
  l.ntiles = (int)tiles.size();
  l.max_rows = max_rows;
  // (segments of one plane never overlap; a partial list exists only for planes their segments cover, and nobody reads
  // what it leaves out)
  l.full_cover = needed != nullptr || covered == (int64_t)w * h;
  if (l.ntiles) {
    if (!l.tiles.reserve(tiles.size() * sizeof(LowpassTile))) return check(hipErrorOutOfMemory, "hipMalloc(tiles)");
    if (!check(hipMemcpyAsync(l.tiles.as<void>(), tiles.data(), tiles.size() * sizeof(LowpassTile),
                              hipMemcpyHostToDevice, stream_), "hipMemcpy(tiles)") ||
        !check(hipStreamSynchronize(stream_), "hipStreamSynchronize"))
      return false;
  }
  l.nfast = (int)fast_tiles.size();
  l.nrest = (int)rest_tiles.size();
  l.max_rows_rest = max_rows_rest;
  l.fast_lds_bytes = fast_lds;
  l.nwide = (int)wide_tiles.size();
  l.wide_lds_bytes = wide_lds;
  if (l.nwide) {
    if (!l.tiles_wide.reserve(wide_tiles.size() * sizeof(LowpassTile))) return check(hipErrorOutOfMemory, "hipMalloc(tiles)");
    if (!check(hipMemcpyAsync(l.tiles_wide.as<void>(), wide_tiles.data(), wide_tiles.size() * sizeof(LowpassTile),
                              hipMemcpyHostToDevice, stream_), "hipMemcpy(tiles)"))
      return false;
  }
  if (l.nfast) {
    if (!l.tiles_fast.reserve(fast_tiles.size() * sizeof(LowpassTile))) return check(hipErrorOutOfMemory, "hipMalloc(tiles)");
    if (!check(hipMemcpyAsync(l.tiles_fast.as<void>(), fast_tiles.data(), fast_tiles.size() * sizeof(LowpassTile),
                              hipMemcpyHostToDevice, stream_), "hipMemcpy(tiles)"))
      return false;
  }
  if (l.nrest) {
    if (!l.tiles_rest.reserve(rest_tiles.size() * sizeof(LowpassTile))) return check(hipErrorOutOfMemory, "hipMalloc(tiles)");
    if (!check(hipMemcpyAsync(l.tiles_rest.as<void>(), rest_tiles.data(), rest_tiles.size() * sizeof(LowpassTile),
                              hipMemcpyHostToDevice, stream_), "hipMemcpy(tiles)"))
      return false;
  }
  if (!check(hipStreamSynchronize(stream_), "hipStreamSynchronize")) return false;
  l.tiles_w = w;
  l.tiles_h = h;
  return true;
}

// launch arguments of the low-pass of one plane (after ensureTiles)
void VideoFrameTransform::fillLowpassArgs(const PlaneState& p, const PlaneState::LowpassLists& l, const uint8_t* d_in, int64_t in_frame_bytes, int in_stride,
                                          uint8_t* d_out, int64_t out_frame_bytes, int out_stride, int w, int h,
                                          t360::LowpassArgs* out) const {
  LowpassArgs a;
  a.src = d_in;
  a.src_frame_bytes = in_frame_bytes;
  a.sstride = in_stride;
  a.dst = d_out;
  a.dst_frame_bytes = out_frame_bytes;
  a.dstride = out_stride;
  a.w = w;
  a.h = h;
  a.segs = p.segs.as<SegmentDev>();
  a.taps_q8 = p.taps_q8.as<int>();
  a.taps_f32 = p.taps_f32.as<float>();
  a.taps_pk = p.taps_pk.as<uint32_t>();
  a.tile_w = kTileW;
  a.dst_dword_ok = ((uintptr_t)d_out % 4 == 0 && out_stride % 4 == 0 && out_frame_bytes % 4 == 0) ? 1 : 0;
  const bool src_dword_ok = (uintptr_t)d_in % 4 == 0 && in_stride % 4 == 0 && in_frame_bytes % 4 == 0 && w % 4 == 0 && w >= 4;
  if ((l.nfast > 0 || l.nwide > 0) && src_dword_ok) {
    a.wide_tiles = l.tiles_wide.as<LowpassTile>();
    a.nwide = l.nwide;
    a.wide_lds_bytes = l.wide_lds_bytes;
    a.taps_sh = p.taps_sh.as<uint32_t>();
    a.fast_tiles = l.tiles_fast.as<LowpassTile>();
    a.nfast = l.nfast;
    a.fast_ky = p.fast_ky;
    a.fast_lds_bytes = l.fast_lds_bytes;
    a.tiles = l.tiles_rest.as<LowpassTile>();
    a.ntiles = l.nrest;
    a.max_rows = l.max_rows_rest;
  } else {
    a.wide_tiles = nullptr;
    a.nwide = 0;
    a.wide_lds_bytes = 0;
    a.taps_sh = nullptr;
    a.fast_tiles = nullptr;
    a.nfast = 0;
    a.fast_ky = 0;
    a.fast_lds_bytes = 0;
    a.tiles = l.tiles.as<LowpassTile>();
    a.ntiles = l.ntiles;
    a.max_rows = l.max_rows;
  }
  *out = a;
}

bool VideoFrameTransform::runLowpass(PlaneState& p, PlaneState::LowpassLists& l, const std::vector<uint8_t>* needed,
                                     const uint8_t* d_in, int64_t in_frame_bytes,
                                     int in_stride, uint8_t* d_out, int64_t out_frame_bytes,
                                     int out_stride, int w, int h, int n_frames, int imagePlaneIndex,
                                     hipStream_t stream) {
  if (!ensureTiles(p, l, needed, w, h, imagePlaneIndex)) return false;
  if (!l.full_cover) {
    // Mat::zeros(...) of filterPlane (:625): only visible where no segment writes
    for (int f = 0; f < n_frames; f++)
      if (!check(hipMemset2DAsync(d_out + (size_t)f * out_frame_bytes, (size_t)out_stride, 0, (size_t)w,
                                  (size_t)h, stream), "hipMemset2DAsync"))
        return false;
  }
  LowpassArgs a;
  fillLowpassArgs(p, l, d_in, in_frame_bytes, in_stride, d_out, out_frame_bytes, out_stride, w, h, &a);
  return check(launch_lowpass(a, n_frames, stream), "low-pass launch");
}

// transformPlane's needResize branch (VideoFrameTransform.cpp:759-776): gather into a warp-map-sized
// image that starts as Scalar(mapIdx ? 128 : 0), then cv::resize(INTER_AREA) into the output plane.
bool VideoFrameTransform::runPlanesScaled(const PlaneJob* jobs, int njobs, int n_frames) {
  std::vector<PlaneJob> inner(jobs, jobs + njobs);
  std::vector<size_t> offs((size_t)njobs);
  std::vector<int> strides((size_t)njobs);
  size_t total = 0;
  for (int k = 0; k < njobs; k++) {
    PlaneState& p = planes_[jobs[k].idx];
    if (jobs[k].out_w != p.resize.dw || jobs[k].out_h != p.resize.dh) {
      // the reference resizes to whatever size it is handed (:735-737): normally the size given to generateMapForPlane,
      // but the filter passes an alpha plane to map 0 with chroma dimensions (vf_transform360.c:368-397).  The tables
      // are rebuilt when the target size changes (a caller alternating two sizes on one map pays that per call).
      if (jobs[k].out_w <= 0 || jobs[k].out_h <= 0 || !buildResizePlan(p, jobs[k].out_w, jobs[k].out_h)) {
        printf("Could not transform the plane %d. Error: no resize plan for this output size\n", jobs[k].image_plane);
        return false;
      }
    }
    if (!p.resize.supported) {
      printf("Could not transform the plane %d. Error: no resize plan for this output size\n", jobs[k].image_plane);
      return false;
    }
    strides[(size_t)k] = (p.map_w + 255) & ~255;
    offs[(size_t)k] = total;
    total += (size_t)strides[(size_t)k] * p.map_h * (size_t)n_frames;
  }
  if (!scaled_[scratch_].reserve(total)) return check(hipErrorOutOfMemory, "hipMalloc(scaled)");
  for (int k = 0; k < njobs; k++) {
    const PlaneState& p = planes_[jobs[k].idx];
    PlaneJob& j = inner[(size_t)k];
    j.out = scaled_[scratch_].as<uint8_t>() + offs[(size_t)k];
    j.out_w = p.map_w;
    j.out_h = p.map_h;
    j.out_stride = strides[(size_t)k];
    j.out_frame_bytes = (int64_t)j.out_stride * p.map_h;
    if (!check(launch_fill_plane(j.out, j.out_frame_bytes, j.out_w, j.out_h, j.out_stride, j.idx ? 128 : 0, n_frames, stream_),
               "fill launch"))
      return false;
  }
  if (!runPlanes(inner.data(), njobs, n_frames)) return false;
  for (int k = 0; k < njobs; k++) {
    PlaneState& p = planes_[jobs[k].idx];
    const PlaneJob& j = inner[(size_t)k];
    ResizeArgs a;
    memset(&a, 0, sizeof(a));
    a.src = j.out;
    a.src_frame_bytes = j.out_frame_bytes;
    a.sstride = j.out_stride;
    a.sw = p.map_w;
    a.sh = p.map_h;
    a.dst = jobs[k].out;
    a.dst_frame_bytes = jobs[k].out_frame_bytes;
    a.dstride = jobs[k].out_stride;
    a.dw = jobs[k].out_w;
    a.dh = jobs[k].out_h;
    a.linear = p.resize.linear ? 1 : 0;
    a.xmax = p.resize.xmax;
    a.iscale_x = p.resize.iscale_x;
    a.iscale_y = p.resize.iscale_y;
    a.inv_area = p.resize.iscale_x > 0 ? 1.f / (float)(p.resize.iscale_x * p.resize.iscale_y) : 0.f;
    a.xofs = p.resize.xofs.as<int>();
    a.x_si = p.resize.x_si.as<int>();
    a.x_alpha = p.resize.x_alpha.as<float>();
    a.yofs = p.resize.yofs.as<int>();
    a.y_si = p.resize.y_si.as<int>();
    a.y_alpha = p.resize.y_alpha.as<float>();
    if (!check(launch_resize_area(a, n_frames, stream_), "resize launch")) return false;
  }
  return true;
}

// reference VideoFrameTransform::transformPlane (VideoFrameTransform.cpp:707-794), for a set of
// planes of a batch of frames: [low-pass each plane] -> gather.  Bicubic planes whose buffers are
// 16-byte friendly are gathered by ONE fused launch of the DMA-ring kernel.
bool VideoFrameTransform::runPlanes(const PlaneJob* jobs, int njobs, int n_frames) {
  const bool barrel = ctx_.output_layout == LAYOUT_BARREL || ctx_.output_layout == LAYOUT_BARREL_SPLIT;
  const int interp = (int)ctx_.interpolation_alg;
  if (!valid_interp(interp)) {
    // reference :780-783: message, nothing written, still "true"
    for (int k = 0; k < njobs; k++) printf("Could not find interpolation algorithm for plane %d", jobs[k].image_plane);
    return true;
  }
  {
    // One map may be named with two plane sizes in one call -- the filter itself does it: an alpha plane goes to map 0
    // with chroma dimensions (vf_transform360.c:368-397).  The per-map tables that depend on the plane size (low-pass
    // tile lists, INTER_AREA tables) exist once per map, so such jobs run one after the other: each group below holds
    // every map with ONE size; the groups' launches and table uploads are ordered on the stream.
    auto clash = [](const PlaneJob& x, const PlaneJob& y) {
      return x.idx == y.idx && (x.in_w != y.in_w || x.in_h != y.in_h || x.out_w != y.out_w || x.out_h != y.out_h);
    };
    bool any = false;
    for (int k = 1; k < njobs && !any; k++)
      for (int m = 0; m < k && !any; m++) any = clash(jobs[m], jobs[k]);
    if (any) {
      std::vector<std::vector<PlaneJob>> groups;
      for (int k = 0; k < njobs; k++) {
        size_t g = 0;
        for (; g < groups.size(); g++) {
          bool ok = true;
          for (const PlaneJob& o : groups[g]) ok = ok && !clash(o, jobs[k]);
          if (ok) break;
        }
        if (g == groups.size()) groups.emplace_back();
        groups[g].push_back(jobs[k]);
      }
      for (const std::vector<PlaneJob>& g : groups)
        if (!runPlanes(g.data(), (int)g.size(), n_frames)) return false;
      return true;
    }
  }
  {
    // planes whose output size differs from their warp map take the resize branch (:735-737, :759-776); a batch may
    // mix both kinds (rounding: a factor of 1.0006 scales 1000 -> 1001 but 500 -> 500), so split it
    std::vector<PlaneJob> scaled, plain;
    for (int k = 0; k < njobs; k++) {
      const PlaneState& p = planes_[jobs[k].idx];
      (jobs[k].out_h != p.map_h || jobs[k].out_w != p.map_w ? scaled : plain).push_back(jobs[k]);
    }
    if (!scaled.empty()) {
      if (!runPlanesScaled(scaled.data(), (int)scaled.size(), n_frames)) return false;
      if (plain.empty()) return true;
      return runPlanes(plain.data(), (int)plain.size(), n_frames);
    }
  }

  // which of the two plans: every plane of the call must have the one that is used
  // (nearest-neighbour plans hold 256-lane tiles only -- a pixel has no stencil halo to share with a wider tile -- so half
  // the waves of an 8-wave workgroup would carry no pixels: 4-wave workgroups, four to a CU, at every batch length;
  // config 1, 64 frames: 0.0652 -> 0.0621 ms on the same box, profiles/r04_experiments/README.md call 8)
  bool small = small_batch_ > 0 && (n_frames < small_batch_ || interp == NEAREST) && interp != LANCZOS4 && waves_ != 4;
  {
  // the planner's host copy of a map's sample LUT is fetched once per map even when both plans of the map end up being
  // tried below, and is gone before anything is launched (up to 2^28 entries x 8 B per map)
  std::vector<LutEntry> host_luts[kMaxMaps];
  auto ensureGatherPlan = [&](PlaneState& ps, bool sm) { return this->ensureGatherPlan(ps, sm, &host_luts[&ps - planes_]); };
  for (int k = 0; k < njobs; k++)
    if (!barrel && !ensureGatherPlan(planes_[jobs[k].idx], small)) return false;
  // (a map the 4-wave planner could not take but the 8-wave one can: every plane of the call then uses the latter)
  for (int k = 0; k < njobs && small; k++) {
    PlaneState& pk = planes_[jobs[k].idx];
    if (!pk.plan_small.valid && pk.plan_ks != 0) {
      if (!ensureGatherPlan(pk, false)) return false;
      if (pk.plan.valid) small = false;
    }
  }
  if (!small)
    for (int k = 0; k < njobs; k++)
      if (!barrel && !ensureGatherPlan(planes_[jobs[k].idx], false)) return false;
  }
  // Planes of a low-pass context whose long-batch plan fuses the filter into its tiles (nftiles > 0): the fused tiles read the
  // RAW plane (remap_fused_kernel), the plan's other tiles the blurred one, and only the segments those still need are
  // filtered.  The raw plane must be 16-byte friendly like any DMA source; if it is not, the plane takes the general
  // gather over the fully filtered plane.
  std::vector<char> lpf((size_t)njobs, 0), no_tiled((size_t)njobs, 0);
  for (int k = 0; k < njobs && ctx_.enable_low_pass_filter && !barrel; k++) {
    const PlaneJob& j = jobs[k];
    const PlaneState& pk = planes_[j.idx];
    const PlaneState::GatherPlan& gp = small ? pk.plan_small : pk.plan;
    if (!gp.valid || gp.nftiles <= 0) continue;
    const bool raw_ok = (reinterpret_cast<uintptr_t>(j.in) & 15) == 0 && (j.in_stride & 15) == 0 &&
                        (n_frames <= 1 || (j.in_frame_bytes & 15) == 0) && (j.in_w & 15) == 0 && j.in_w == pk.in_w && j.in_h == pk.in_h;
    (raw_ok ? lpf : no_tiled)[(size_t)k] = 1;
  }
  TiledArgs fused;
  auto reset_fused = [&]() {
    memset(&fused, 0, sizeof(fused));
    fused.wtab = weights_.as<int16_t>();
    fused.wpack = weights_pack_.as<uint32_t>();
    fused.nframes = n_frames;
    fused.frames_per_block = frames_per_block_ < n_frames ? frames_per_block_ : n_frames;
    fused.ks = interp == NEAREST ? 1 : interp == LINEAR ? 2 : interp == CUBIC ? 4 : 8;
    // Lanczos4 plans hold 16x16 tiles only (one pixel per lane, 32 weight dwords each): workgroups of 4 waves
    fused.max_pieces = fused.ks == 8 ? std::min(max_pieces_, 16) : small ? kSmallPlanPieces : max_pieces_;
    fused.ring_kb = (fused.ks == 8 && waves_ == 8) || small ? 38 : ring_kb_;
    fused.waves = fused.ks == 8 || small ? 4 : waves_;
#ifdef T360_INSTRUMENT
    fused.debug = getenv("T360_DEBUG") ? atoi(getenv("T360_DEBUG")) : 0;
    fused.lds_pad = getenv("T360_LDS_PAD") ? atoi(getenv("T360_LDS_PAD")) : 0;
    fused.k_lo = getenv("T360_K_LO") ? atoi(getenv("T360_K_LO")) : 0;
    fused.k_hi = getenv("T360_K_HI") ? atoi(getenv("T360_K_HI")) : 0;
#endif
  };
  // frames per work item, tail split and grid bookkeeping of a launch of the tiled (or the fused) kernel
  auto finalize_launch = [&](TiledArgs& ta) {
    // A launch with fewer work items than 1.5 x the workgroups the GPU holds at once ends on a nearly empty machine
    // (BASELINE config 1: 576 tiles for 512 slots = one full round and one of 64 workgroups, 64 frames each): every
    // tile's frames are split into more runs until the items pass that mark (config 1, 64 frames: 0.082 -> 0.064 ms with
    // two runs of 32; 16 frames on the 4-wave plan: 0.023 -> 0.020 ms).  Runs stay >= 8 frames, a workgroup's start-up
    // being worth ~5 of them; BASELINE config 2 (1 152 tiles) is not affected.
    {
      const int slots = cus_ * (ta.waves == 8 ? 2 : 4);
      const int tiles = ta.total_tiles + ta.total_direct;
      int runs = (n_frames + ta.frames_per_block - 1) / ta.frames_per_block;
      while (tiles > 0 && 2 * tiles * runs < 3 * slots && n_frames / (runs + 1) >= 8) runs++;
      ta.frames_per_block = std::min(ta.frames_per_block, (n_frames + runs - 1) / runs);
    }
    ta.groups = (n_frames + ta.frames_per_block - 1) / ta.frames_per_block;
    // the tail tiles walk the batch in at least two runs of <= tail_frames_ frames, all of EQUAL length (20 frames:
    // 10 + 10, not 16 + 4: -6 %; 8 frames: 4 + 4: -2 %)
    ta.tail_frames = std::max(1, std::min({ta.frames_per_block, tail_frames_, (n_frames + 1) / 2}));
    ta.tail_groups = (n_frames + ta.tail_frames - 1) / ta.tail_frames;
    ta.tail_frames = (n_frames + ta.tail_groups - 1) / ta.tail_groups;
    ta.tail_percent = ta.tail_groups > ta.groups ? tail_percent_ : 0;
    ta.direct_blocks = (ta.total_direct * ta.groups + 7) & ~7;  // a multiple of 8: staged ids keep their XCD
  };
  reset_fused();
  // The fused low-pass tiles of the call's planes: ONE launch of remap_fused_kernel.  It reads the raw planes only, so it does
  // not wait for the low-pass launches below: it goes to a side stream first and runs beside them and beside the tiled
  // kernel's launch for the plan's other tiles (joined at the end of the call).
  bool lpa_launched = false;
  char lpa_name[48] = "";
  {
    FusedArgs lpa;
    memset(&lpa, 0, sizeof(lpa));
    lpa.base = fused;
    const bool multi = n_frames > 1;
    for (int k = 0; k < njobs; k++) {
      if (!lpf[(size_t)k] || lpa.base.nplanes >= 4) continue;
      const PlaneJob& j = jobs[k];
      PlaneState& p = planes_[j.idx];
      const PlaneState::GatherPlan& gp = p.plan;
      TiledPlane fp;
      memset(&fp, 0, sizeof(fp));
      fp.src = j.in;  // the RAW plane: these tiles filter their own footprint
      fp.src_frame_bytes = j.in_frame_bytes;
      fp.sstride = j.in_stride;
      fp.dst = j.out;
      fp.dst_frame_bytes = j.out_frame_bytes;
      fp.sw = j.in_w;
      fp.sh = j.in_h;
      fp.dw = j.out_w;
      fp.dh = j.out_h;
      fp.dstride = j.out_stride;
      fp.tiles = gp.ftiles.as<TileDesc>();
      fp.tlut = gp.ftlut.as<uint32_t>();
      fp.chunks = gp.fchunks.as<uint32_t>();
      fp.lut = p.lut.as<LutEntry>();
      fp.ntiles = gp.nftiles;
      fp.dst_dword_ok = (reinterpret_cast<uintptr_t>(j.out) & 3) == 0 && (j.out_stride & 3) == 0 && (!multi || (j.out_frame_bytes & 3) == 0);
      lpa.taps[lpa.base.nplanes] = p.fuse_taps.as<uint32_t>();
      lpa.base.plane[lpa.base.nplanes++] = fp;
      lpa.base.total_tiles += fp.ntiles;
    }
    if (lpa.base.nplanes > 0) {
      finalize_launch(lpa.base);
      hipStream_t fs = fuse_side_stream_ ? lp_streams_[2] : stream_;
      if (fuse_side_stream_ &&
          (!check(hipEventRecord(lp_fork_, stream_), "hipEventRecord") || !check(hipStreamWaitEvent(fs, lp_fork_, 0), "hipStreamWaitEvent")))
        return false;
      if (!check(launch_remap_fused(lpa, fs), "fused remap launch")) return false;
      if (fuse_side_stream_ && !check(hipEventRecord(lp_join_[2], fs), "hipEventRecord")) return false;
      lpa_launched = true;
      snprintf(lpa_name, sizeof(lpa_name), "remap_fused_kernel<%d> + ", lpa.base.ks);
    }
  }
  // ---- stage 1: segmented low-pass into the scratch planes (filterPlane, :621-704) ----
  struct Src {
    const uint8_t* ptr;
    int64_t frame_bytes;
    int stride;
  };
  std::vector<Src> srcs((size_t)njobs);
  if (ctx_.enable_low_pass_filter) {
    size_t total = 0;
    std::vector<size_t> offs((size_t)njobs);
    for (int k = 0; k < njobs; k++) {
      const int bstride = (jobs[k].in_w + 255) & ~255;
      offs[(size_t)k] = total;
      total += (size_t)bstride * jobs[k].in_h * (size_t)n_frames;
    }
    if (!blurred_[scratch_].reserve(total)) return check(hipErrorOutOfMemory, "hipMalloc(blurred)");
    // The planes of a yuv420p batch as ONE launch when every plane is served by the wide fixed-point tiles alone (BASELINE
    // config 3 is): one ramp and one drain instead of three, no event joins.  Anything else -- float-path segments, odd
    // widths, a plane its segments do not cover, one map named with two sizes -- takes the per-plane launches below.
    bool merged = false;
    if (njobs >= 2 && njobs <= 3 && use_wide_lowpass_ && merge_lowpass_) {
      LowpassArgs la[3];
      bool ok = true;
      for (int k = 0; k < njobs && ok; k++) {
        const PlaneJob& j = jobs[k];
        PlaneState& pk = planes_[j.idx];
        for (int m = 0; m < k; m++) ok = ok && !(jobs[m].idx == j.idx && (jobs[m].in_w != j.in_w || jobs[m].in_h != j.in_h));
        if (!ok) break;
        PlaneState::LowpassLists& ll = lpf[(size_t)k] ? pk.lp_part : pk.lp;
        if (!ensureTiles(pk, ll, lpf[(size_t)k] ? &pk.plan.seg_needed : nullptr, j.in_w, j.in_h, j.image_plane)) return false;
        const int bstride = (j.in_w + 255) & ~255;
        fillLowpassArgs(pk, ll, j.in, j.in_frame_bytes, j.in_stride, blurred_[scratch_].as<uint8_t>() + offs[(size_t)k],
                        (int64_t)bstride * j.in_h, bstride, j.in_w, j.in_h, &la[k]);
        ok = ll.full_cover;
      }
      if (ok && lowpass_mergeable(la, njobs)) {
        if (!check(launch_lowpass_multi(la, njobs, n_frames, stream_), "low-pass launch")) return false;
        last_lowpass_ = "merged";
        for (int k = 0; k < njobs; k++) {
          const int bstride = (jobs[k].in_w + 255) & ~255;
          srcs[(size_t)k] = Src{blurred_[scratch_].as<uint8_t>() + offs[(size_t)k], (int64_t)bstride * jobs[k].in_h, bstride};
        }
        merged = true;
      }
    }
    if (!merged) last_lowpass_ = "per-plane";
    const bool side = njobs > 1 && njobs <= 4;  // planes 1.. on their own streams beside plane 0
    if (side && !merged && !check(hipEventRecord(lp_fork_, stream_), "hipEventRecord")) return false;
    for (int k = 0; k < njobs && !merged; k++) {
      const PlaneJob& j = jobs[k];
      const int bstride = (j.in_w + 255) & ~255;
      const int64_t plane_bytes = (int64_t)bstride * j.in_h;
      uint8_t* bl = blurred_[scratch_].as<uint8_t>() + offs[(size_t)k];
      hipStream_t st = stream_;
      if (side && k > 0) {
        st = lp_streams_[k - 1];
        if (!check(hipStreamWaitEvent(st, lp_fork_, 0), "hipStreamWaitEvent")) return false;
      }
      PlaneState& pj = planes_[j.idx];
      if (!runLowpass(pj, lpf[(size_t)k] ? pj.lp_part : pj.lp, lpf[(size_t)k] ? &pj.plan.seg_needed : nullptr, j.in, j.in_frame_bytes,
                      j.in_stride, bl, plane_bytes, bstride, j.in_w, j.in_h, n_frames, j.image_plane, st))
        return false;
      if (side && k > 0 &&
          (!check(hipEventRecord(lp_join_[k - 1], st), "hipEventRecord") ||
           !check(hipStreamWaitEvent(stream_, lp_join_[k - 1], 0), "hipStreamWaitEvent")))
        return false;
      srcs[(size_t)k] = Src{bl, plane_bytes, bstride};
    }
  } else {
    for (int k = 0; k < njobs; k++) srcs[(size_t)k] = Src{jobs[k].in, jobs[k].in_frame_bytes, jobs[k].in_stride};
  }

  // ---- stage 2: gather ----
  // Planes with a tile plan and 16-byte friendly buffers go into fused launches of the LDS-tiled kernel
  // (up to 4 planes each: Y, U and V of a yuv420p batch are ONE launch); everything else -- BARREL outputs
  // (BORDER_TRANSPARENT), odd alignments or widths -- takes the general gather.
  auto flush_fused = [&]() -> bool {
    if (fused.nplanes == 0) return true;
    finalize_launch(fused);
#ifdef T360_INSTRUMENT
    t360::DeviceBuffer trace;
    const char* trace_path = getenv("T360_TRACE");
    const size_t nwg = (size_t)fused.direct_blocks + (size_t)8 * (((fused.total_tiles + 7) / 8) + 1) * std::max(fused.groups, fused.tail_groups);
    if (trace_path && trace.reserve(nwg * 64) && hipMemsetAsync(trace.as<void>(), 0, nwg * 64, stream_) == hipSuccess)
      fused.trace = trace.as<unsigned long long>();
    t360::DeviceBuffer phases;
    const char* phases_path = getenv("T360_PHASES");
    if (phases_path && phases.reserve(nwg * 128) && hipMemsetAsync(phases.as<void>(), 0, nwg * 128, stream_) == hipSuccess)
      fused.phases = phases.as<unsigned long long>();
#endif
    const bool ok = check(launch_remap_tiled(fused, stream_), "tiled remap launch");
#ifdef T360_INSTRUMENT
    if (fused.phases) {
      std::vector<unsigned long long> host(nwg * 16);
      if (hipStreamSynchronize(stream_) == hipSuccess &&
          hipMemcpy(host.data(), phases.as<void>(), nwg * 128, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE* f = fopen(phases_path, "wb")) {
          fwrite(host.data(), 8, host.size(), f);
          fclose(f);
        }
      }
    }
    if (fused.trace) {
      std::vector<unsigned long long> host(nwg * 8);
      if (hipStreamSynchronize(stream_) == hipSuccess &&
          hipMemcpy(host.data(), trace.as<void>(), nwg * 64, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE* f = fopen(trace_path, "wb")) {
          fwrite(host.data(), 8, host.size(), f);
          fclose(f);
        }
      }
    }
#endif
    setLastKernel(remap_tiled_kernel_name(fused.ks, fused.ring_kb, fused.waves));
    reset_fused();
    return ok;
  };
  reset_fused();
  const bool multi = n_frames > 1;
  for (int k = 0; k < njobs; k++) {
    const PlaneJob& j = jobs[k];
    PlaneState& p = planes_[j.idx];
    const Src& s = srcs[(size_t)k];
    if (j.idx != 0 && barrel) {  // :743-747
      if (!check(launch_fill_plane(j.out, j.out_frame_bytes, j.out_w, j.out_h, j.out_stride, 128, n_frames, stream_),
                 "fill launch"))
        return false;
    }
    // chunks go global -> LDS by DMA: 16-byte friendly source buffers only
    const bool vec_ok = (reinterpret_cast<uintptr_t>(s.ptr) & 15) == 0 && (s.stride & 15) == 0 &&
                        (!multi || (s.frame_bytes & 15) == 0) && (j.in_w & 15) == 0;
    const PlaneState::GatherPlan& gp = small ? p.plan_small : p.plan;
    if (gp.valid && !barrel && j.in_w == p.in_w && j.in_h == p.in_h && vec_ok && !no_tiled[(size_t)k]) {
      TiledPlane tp;
      memset(&tp, 0, sizeof(tp));
      tp.src = s.ptr;
      tp.src_frame_bytes = s.frame_bytes;
      tp.dst = j.out;
      tp.dst_frame_bytes = j.out_frame_bytes;
      tp.sw = j.in_w;
      tp.sh = j.in_h;
      tp.sstride = s.stride;
      tp.dw = j.out_w;
      tp.dh = j.out_h;
      tp.dstride = j.out_stride;
      tp.tiles = gp.tiles.as<TileDesc>();
      tp.tlut = gp.tlut.as<uint32_t>();
      tp.chunks = gp.chunks.as<uint32_t>();
      tp.lut = p.lut.as<LutEntry>();
      tp.ntiles = gp.ntiles;
      tp.ndirect = gp.ndirect;
      tp.ndirect_top = gp.ndirect_top;
      tp.dst_dword_ok = (reinterpret_cast<uintptr_t>(j.out) & 3) == 0 && (j.out_stride & 3) == 0 &&
                        (!multi || (j.out_frame_bytes & 3) == 0);
      tp.scatter = gp.scatter ? 1 : 0;
      if (fused.nplanes == 4 && !flush_fused()) return false;
      fused.plane[fused.nplanes++] = tp;
      fused.total_tiles += tp.ntiles;
      fused.total_direct += tp.ndirect;
      continue;
    }
    GatherArgs a;
    a.src = s.ptr;
    a.src_frame_bytes = s.frame_bytes;
    a.sw = j.in_w;
    a.sh = j.in_h;
    a.sstride = s.stride;
    a.dst = j.out;
    a.dst_frame_bytes = j.out_frame_bytes;
    a.dw = j.out_w;
    a.dh = j.out_h;
    a.dstride = j.out_stride;
    a.lut = p.lut.as<LutEntry>();
    a.wtab = weights_.as<int16_t>();
    a.interp = interp;
    a.border = barrel ? kBorderTransparent : kBorderWrap;  // :716-719
    if (!check(launch_remap_gather(a, n_frames, stream_), "remap launch")) return false;
    setLastKernel("remap_gather_kernel");
  }
  if (!flush_fused()) return false;
  if (lpa_launched) {
    if (fuse_side_stream_ && !check(hipStreamWaitEvent(stream_, lp_join_[2], 0), "hipStreamWaitEvent")) return false;
    char name[64];
    snprintf(name, sizeof(name), "%s%s", lpa_name, last_kernel_);
    setLastKernel(name);
  }
  return true;
}

// Tile work list of the LDS-tiled gather for one map: the sample LUT is copied to the host and planned there
// (t360_plan.cpp) -- LAZILY, by the first call that needs the plan (ensureGatherPlan): batches of fewer than
// small_batch_ frames (and the single-plane calls of the reference ABI) use the plan for workgroups of 4 waves, longer
// batches the one for 8 waves, and a caller normally lives in one of the two regimes, so planning both at init would
// double the first-frame latency for nothing (the filter initialises inside its first filter_frame, vf_transform360.c:
// 346-352).  Planes the tiled kernel cannot take (BARREL outputs, widths that are not multiples of 16) simply have no
// plan and use the general gather.
bool VideoFrameTransform::buildGatherPlan(PlaneState& p, const MapGenParams& P, int in_w, int in_h) {
  p.plan.valid = p.plan.tried = false;
  p.plan_small.valid = p.plan_small.tried = false;
  p.plan_ks = 0;
  const bool barrel = P.output_layout == LAYOUT_BARREL || P.output_layout == LAYOUT_BARREL_SPLIT;
  const int ks = P.interp == NEAREST ? 1 : P.interp == LINEAR ? 2 : P.interp == CUBIC ? 4 : P.interp == LANCZOS4 ? 8 : 0;
  if (ks == 0 || barrel || !use_tiled_ || (in_w & 15) != 0) return true;
  p.plan_ks = ks;
  return true;
}

bool VideoFrameTransform::ensureGatherPlan(PlaneState& p, bool small, std::vector<LutEntry>* lut_of_this_call) {
  PlaneState::GatherPlan& g = small ? p.plan_small : p.plan;
  if (g.valid || g.tried || p.plan_ks == 0) return true;
  g.tried = true;  // not plannable stays not plannable: the general gather serves the map
  // The planner works on a host copy of the sample LUT.  It is fetched from device memory HERE and lives for the
  // transform call that needs it only (12.6 MB for the 4K luma map, up to 2^28 entries x 8 B for the largest map the
  // ABI admits): a handle never holds a host copy between calls, whichever regimes it ends up planning;
  // a call that plans both regimes of one map fetches it once.
  std::vector<LutEntry>& host_lut = *lut_of_this_call;
  if (host_lut.size() != (size_t)p.map_w * (size_t)p.map_h) {
    const size_t n = (size_t)p.map_w * (size_t)p.map_h;
    try {
      host_lut.resize(n);
    } catch (const std::exception&) {
      return true;  // no host memory for the planner's copy: the general gather serves the map
    }
    if (!check(hipMemcpyAsync(host_lut.data(), p.lut.as<void>(), n * sizeof(LutEntry), hipMemcpyDeviceToHost, stream_), "hipMemcpy(lut)") ||
        !check(hipStreamSynchronize(stream_), "hipStreamSynchronize"))
      return false;
  }
  const int ks = p.plan_ks;
  const int waves = small || ks == 8 ? 4 : waves_;
  const int max_pieces = ks == 8 ? std::min(max_pieces_, 16) : small ? kSmallPlanPieces : max_pieces_;
  PlanOptions o;
  o.ks = ks;
  o.waves = waves;
  o.max_pieces = max_pieces;
  o.wide_pct = plan_wide_pct_;
  o.strip_pct = plan_strip_pct_;
  o.wide256_pct = small ? 0 : plan_wide256_pct_;
  o.scatter = small ? 0 : plan_scatter_;
  // nearest-neighbour maps: a pixel has no stencil halo, so by staged chunks a 32x32 tile always looked as good as a 64x16
  // one -- but its 77-byte row fragments touch 1.6 lines for every 0.6 they need.  Compared by the 128-byte lines under
  // their fragments the planner takes 64x16 tiles: BASELINE config 1, 64 frames 0.0639 -> 0.0579 ms on the same box
  // (profiles/r05_experiments/README.md call 8)
  o.cost_lines = plan_cost_lines_ != 0 || ks == 1;
  o.band = plan_band_ > 0 ? plan_band_ : 4;
  o.order = plan_band_ > 0 ? 0 : plan_band_ == 0 ? 1 : 2;
  o.row_pad = plan_row_pad_;
  o.row_align = plan_row_align_;
  // long batches of a low-pass context: tiles whose source rows all have short fixed-point kernels filter their own
  // footprint in LDS (remap_fused_kernel); the plan then also says which segments its other tiles still need blurred
  if (!small && p.fuse_ok && ctx_.enable_low_pass_filter && waves == 8 && (ks == 2 || ks == 4) && o.scatter <= 0 && o.wide256_pct <= 0)
    o.fuse = &p.fuse_info;
  HostGatherPlan hp;
  if (!plan_gather(host_lut.data(), p.map_w, p.map_h, p.in_w, p.in_h, o, &hp)) return true;
  if (!g.tiles.reserve(std::max<size_t>(hp.tiles.size(), 1) * sizeof(TileDesc)) ||
      !g.tlut.reserve(hp.tlut.size() * sizeof(uint32_t)) || !g.chunks.reserve(hp.chunks.size() * sizeof(uint32_t)))
    return check(hipErrorOutOfMemory, "hipMalloc(gather plan)");
  if ((!hp.tiles.empty() &&
       !check(hipMemcpyAsync(g.tiles.as<void>(), hp.tiles.data(), hp.tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice,
                             stream_), "hipMemcpy(tiles)")) ||
      !check(hipMemcpyAsync(g.tlut.as<void>(), hp.tlut.data(), hp.tlut.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                            stream_), "hipMemcpy(tlut)") ||
      !check(hipMemcpyAsync(g.chunks.as<void>(), hp.chunks.data(), hp.chunks.size() * sizeof(uint32_t),
                            hipMemcpyHostToDevice, stream_), "hipMemcpy(chunks)") ||
      !check(hipStreamSynchronize(stream_), "hipStreamSynchronize"))
    return false;
  g.nftiles = 0;
  g.seg_needed.clear();
  if (hp.nftiles > 0) {
    if (!g.ftiles.reserve(hp.ftiles.size() * sizeof(TileDesc)) || !g.ftlut.reserve(hp.ftlut.size() * sizeof(uint32_t)) ||
        !g.fchunks.reserve(hp.fchunks.size() * sizeof(uint32_t)))
      return check(hipErrorOutOfMemory, "hipMalloc(fused gather plan)");
    if (!check(hipMemcpyAsync(g.ftiles.as<void>(), hp.ftiles.data(), hp.ftiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice,
                              stream_), "hipMemcpy(ftiles)") ||
        !check(hipMemcpyAsync(g.ftlut.as<void>(), hp.ftlut.data(), hp.ftlut.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                              stream_), "hipMemcpy(ftlut)") ||
        !check(hipMemcpyAsync(g.fchunks.as<void>(), hp.fchunks.data(), hp.fchunks.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                              stream_), "hipMemcpy(fchunks)") ||
        !check(hipStreamSynchronize(stream_), "hipStreamSynchronize"))
      return false;
    g.nftiles = hp.nftiles;
    g.seg_needed = hp.seg_needed;
  }
  g.ntiles = hp.ntiles;
  g.ndirect = hp.ndirect;
  g.ndirect_top = hp.ndirect_top;
  g.stats = hp.stats;
  g.waves = waves;
  g.max_pieces = max_pieces;
  g.scatter = hp.scatter;
  g.valid = true;
  return true;
}

// reference VideoFrameTransform::transformFramePlane (VideoFrameTransform.cpp:1319-1351)
bool VideoFrameTransform::transformFramePlane(uint8_t* inputData, uint8_t* outputData, int inputWidth,
                                              int inputHeight, int inputWidthWithPadding, int outputWidth,
                                              int outputHeight, int outputWidthWithPadding, int idx,
                                              int imagePlaneIndex) {
  if (!ok_) return false;
  if (idx < 0 || idx >= kMaxMaps || !planes_[idx].valid) {
    printf("Could not transform the plane %d. Error: no map generated for index %d\n", imagePlaneIndex, idx);
    return false;
  }
  if (!inputData || !outputData || inputWidth <= 0 || inputHeight <= 0 || outputWidth <= 0 ||
      outputHeight <= 0 || inputWidthWithPadding < inputWidth || outputWidthWithPadding < outputWidth) {
    printf("Could not transform the plane %d. Error: invalid buffer description\n", imagePlaneIndex);
    return false;
  }
  DeviceGuard g(device_);
  const PtrKind ik = classify(inputData), ok = classify(outputData);
  const size_t in_bytes = (size_t)inputWidthWithPadding * (size_t)(inputHeight - 1) + (size_t)inputWidth;
  const size_t out_bytes = (size_t)outputWidthWithPadding * (size_t)(outputHeight - 1) + (size_t)outputWidth;

  if ((ik != PtrKind::Host || ok != PtrKind::Host) && stream_ == own_stream_) {
    // device buffers usually come from work queued on the caller's streams, which this handle's
    // private non-blocking stream does not order against: the call is synchronous anyway, so wait
    // for the device first (callers that manage ordering themselves use T360_setStream)
    if (!check(hipDeviceSynchronize(), "hipDeviceSynchronize")) return false;
  }
  const uint8_t* d_in = inputData;
  int in_stride = inputWidthWithPadding;
  if (ik == PtrKind::Host) {
    // stage over PCIe (t360_hoststage.h: one contiguous copy per plane where the strides allow): rows at a 256-byte aligned pitch
    in_stride = (inputWidth + 255) & ~255;
    if (!stage_in_.reserve((size_t)std::max(in_stride, inputWidthWithPadding) * inputHeight))
      return check(hipErrorOutOfMemory, "hipMalloc(stage_in)");
    if (!stager_.to_device(inputData, inputWidth, inputHeight, inputWidthWithPadding, stage_in_.as<uint8_t>(), in_stride, stream_,
                           &in_stride))
      return check(hipErrorUnknown, "host -> device staging");
    d_in = stage_in_.as<uint8_t>();
  }
  uint8_t* d_out = outputData;
  int out_stride = outputWidthWithPadding;
  if (ok == PtrKind::Host) {
    out_stride = (outputWidth + 255) & ~255;
    if (!stage_out_.reserve((size_t)out_stride * outputHeight)) return check(hipErrorOutOfMemory, "hipMalloc(stage_out)");
    d_out = stage_out_.as<uint8_t>();
    const bool barrel = ctx_.output_layout == LAYOUT_BARREL || ctx_.output_layout == LAYOUT_BARREL_SPLIT;
    if (barrel) {
      // BORDER_TRANSPARENT leaves destination bytes untouched: start from the caller's content
      int used = out_stride;
      if (!stage_out_.reserve((size_t)std::max(out_stride, outputWidthWithPadding) * outputHeight))
        return check(hipErrorOutOfMemory, "hipMalloc(stage_out)");
      d_out = stage_out_.as<uint8_t>();
      if (!stager_.to_device(outputData, outputWidth, outputHeight, outputWidthWithPadding, d_out, out_stride, stream_, &used))
        return check(hipErrorUnknown, "host -> device staging");
      out_stride = used;
    }
  }
  (void)in_bytes;
  (void)out_bytes;
  const PlaneJob job{d_in, 0, inputWidth, inputHeight, in_stride, d_out, 0, outputWidth, outputHeight, out_stride, idx,
                     imagePlaneIndex};
  if (!runPlanes(&job, 1, 1)) return false;
  if (ok == PtrKind::Host) {
    if (!stager_.to_host(outputData, outputWidth, outputHeight, outputWidthWithPadding, d_out, out_stride, stream_))
      return check(hipErrorUnknown, "device -> host staging");
  }
  // the reference call is synchronous: the output is complete when it returns
  return check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

bool VideoFrameTransform::transformFrames(const uint8_t* d_in, int64_t in_frame_bytes, uint8_t* d_out,
                                          int64_t out_frame_bytes, int n_frames, const T360PlaneDesc* planes,
                                          int n_planes) {
  if (!ok_) return false;
  if (!d_in || !d_out || n_frames < 0 || !planes || n_planes <= 0) {
    printf("transform360: T360_transformFrames: invalid arguments\n");
    return false;
  }
  if (n_frames == 0) return true;
  DeviceGuard g(device_);
  std::vector<PlaneJob> jobs;
  for (int k = 0; k < n_planes; k++) {
    const T360PlaneDesc& d = planes[k];
    if (d.map_index < 0 || d.map_index >= kMaxMaps || !planes_[d.map_index].valid) {
      printf("Could not transform the plane %d. Error: no map generated for index %d\n", k, d.map_index);
      return false;
    }
    if (d.in_width <= 0 || d.in_height <= 0 || d.out_width <= 0 || d.out_height <= 0 ||
        d.in_stride < d.in_width || d.out_stride < d.out_width) {
      printf("Could not transform the plane %d. Error: invalid plane description\n", k);
      return false;
    }
    jobs.push_back(PlaneJob{d_in + d.in_offset, in_frame_bytes, d.in_width, d.in_height, d.in_stride,
                            d_out + d.out_offset, out_frame_bytes, d.out_width, d.out_height, d.out_stride,
                            d.map_index, k});
  }
  return runPlanes(jobs.data(), (int)jobs.size(), n_frames);
}

bool VideoFrameTransform::filterPlane(const uint8_t* d_in, uint8_t* d_out, int width, int height, int in_stride,
                                      int out_stride, int idx) {
  if (!ok_ || idx < 0 || idx >= kMaxMaps || !planes_[idx].valid) return false;
  if (!ctx_.enable_low_pass_filter) {
    printf("transform360: T360_filterPlane: the low-pass filter is disabled in this context\n");
    return false;
  }
  DeviceGuard g(device_);
  return runLowpass(planes_[idx], planes_[idx].lp, nullptr, d_in, 0, in_stride, d_out, 0, out_stride, width, height, 1, idx, stream_);
}

bool VideoFrameTransform::planStats(int idx, int64_t* st) const {
  if (idx < 0 || idx >= kMaxMaps || !planes_[idx].valid || !(planes_[idx].plan.valid || planes_[idx].plan_small.valid)) return false;
  // plans are built by the first call that needs them: the long-batch plan if it exists, else the short-batch one
  const PlaneState::GatherPlan& g = planes_[idx].plan.valid ? planes_[idx].plan : planes_[idx].plan_small;
  st[0] = g.ntiles;
  st[1] = g.ndirect;
  st[2] = g.stats.fetched_bytes;
  st[3] = g.stats.lds_bytes;
  st[4] = g.stats.direct_pixels;
  st[5] = (int64_t)(g.tiles.size() + g.tlut.size() + g.chunks.size());
  st[6] = g.stats.n_scatter;  // tiles of a scatter plan that really are scatter tiles (instrumented build only: T360_SCATTER)
  st[7] = 0;
  return true;
}

bool VideoFrameTransform::getMapSize(int idx, int* w, int* h) const {
  if (idx < 0 || idx >= kMaxMaps || !planes_[idx].valid) return false;
  *w = planes_[idx].map_w;
  *h = planes_[idx].map_h;
  return true;
}

bool VideoFrameTransform::copyMap(int idx, float* host_dst) {
  if (!ok_ || idx < 0 || idx >= kMaxMaps || !planes_[idx].valid || !host_dst) return false;
  DeviceGuard g(device_);
  const PlaneState& p = planes_[idx];
  if (!check(hipStreamSynchronize(stream_), "hipStreamSynchronize")) return false;
  return check(hipMemcpy(host_dst, p.map.as<void>(), (size_t)p.map_w * p.map_h * sizeof(float2), hipMemcpyDeviceToHost),
               "hipMemcpy(map)");
}

int VideoFrameTransform::segmentCount(int idx) const {
  if (idx < 0 || idx >= kMaxMaps || !planes_[idx].valid) return 0;
  return (int)planes_[idx].filter.segments.size();
}

bool VideoFrameTransform::getSegment(int idx, int i, int* rect4, int* lens2, int* fixed_point) const {
  if (i < 0 || i >= segmentCount(idx)) return false;
  const Segment& s = planes_[idx].filter.segments[(size_t)i];
  rect4[0] = s.left;
  rect4[1] = s.top;
  rect4[2] = s.width;
  rect4[3] = s.height;
  lens2[0] = (int)s.kx.size();
  lens2[1] = (int)s.ky.size();
  if (fixed_point) *fixed_point = s.fixed_point ? 1 : 0;
  return true;
}

bool VideoFrameTransform::copySegmentKernels(int idx, int i, float* kx, float* ky) const {
  if (i < 0 || i >= segmentCount(idx)) return false;
  const Segment& s = planes_[idx].filter.segments[(size_t)i];
  memcpy(kx, s.kx.data(), s.kx.size() * sizeof(float));
  memcpy(ky, s.ky.data(), s.ky.size() * sizeof(float));
  return true;
}
