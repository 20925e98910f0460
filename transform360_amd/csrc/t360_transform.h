// t360_transform.h -- the object behind the opaque `VideoFrameTransform*` handle.
//
// Mirrors the public surface of the reference class (reference VideoFrameTransform.h:40-160:
// ctor from FrameTransformContext, generateMapForPlane, transformFramePlane) so the C entry
// points stay one-line trampolines as in the reference (VideoFrameTransformHandler.cpp:18-64).
// State redesigned for the GPU: warpMats_ -> device float map + packed sample LUT;
// filterKernelsX_/Y_ + segmentFilteringConfigs_ -> packed device tap arrays + a tile work list.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "Transform360/t360_device.h"
#include "t360_filtercfg.h"
#include "t360_hoststage.h"
#include "t360_internal.h"
#include "t360_devbuf.h"
#include "t360_kernels.h"
#include "t360_plan.h"

namespace t360 {

// OpenCV's fixed-point 2-D interpolation table for LINEAR / CUBIC / LANCZOS4 (host build).
bool build_inter_table(int interp, std::vector<int16_t>* tab, int* ksize);

}  // namespace t360

// Global-namespace class: `typedef class VideoFrameTransform VideoFrameTransform;` in
// Transform360/VideoFrameTransformHandler.h names exactly this type.
class VideoFrameTransform {
 public:
  explicit VideoFrameTransform(const FrameTransformContext* ctx);
  ~VideoFrameTransform();
  VideoFrameTransform(const VideoFrameTransform&) = delete;
  VideoFrameTransform& operator=(const VideoFrameTransform&) = delete;

  bool ok() const { return ok_; }

  // reference VideoFrameTransform::generateMapForPlane (VideoFrameTransform.cpp:504-576)
  bool generateMapForPlane(int inputWidth, int inputHeight, int outputWidth, int outputHeight,
                           int transformMatPlaneIndex);
  // reference VideoFrameTransform::transformFramePlane (VideoFrameTransform.cpp:1319-1351);
  // buffers may be host or device memory
  bool transformFramePlane(uint8_t* inputData, uint8_t* outputData, int inputWidth, int inputHeight,
                           int inputWidthWithPadding, int outputWidth, int outputHeight,
                           int outputWidthWithPadding, int transformMatPlaneIndex, int imagePlaneIndex);

  // additive (Transform360/t360_device.h)
  bool setStream(void* hipStream);
  bool useOwnStream();
  bool synchronize();
  bool transformFrames(const uint8_t* d_in, int64_t in_frame_bytes, uint8_t* d_out,
                       int64_t out_frame_bytes, int n_frames, const T360PlaneDesc* planes, int n_planes);
  // a stream of independent batches, round-robin over pipe_depth_ internal streams (t360_device.h)
  bool transformFramesPipelined(const uint8_t* d_in, int64_t in_frame_bytes, uint8_t* d_out,
                                int64_t out_frame_bytes, int n_frames, const T360PlaneDesc* planes, int n_planes);
  bool setPipelineDepth(int depth);
  bool setFusedLowpass(bool on) {  // takes effect for maps generated afterwards (t360_device.h)
    fuse_lowpass_ = on;
    return true;
  }
  bool pipelineJoin();
  bool filterPlane(const uint8_t* d_in, uint8_t* d_out, int width, int height, int in_stride,
                   int out_stride, int map_index);
  bool getMapSize(int idx, int* w, int* h) const;
  bool copyMap(int idx, float* host_dst);
  int segmentCount(int idx) const;
  bool getSegment(int idx, int i, int* rect4, int* lens2, int* fixed_point) const;
  bool copySegmentKernels(int idx, int i, float* kx, float* ky) const;
  const char* lastKernel() const { return last_kernel_; }
  // "merged" (Y, U and V of a batch in one lowpass_q8w_multi_kernel launch), "per-plane", or "" before the first low-pass
  const char* lastLowpassPath() const { return last_lowpass_; }
  bool planStats(int idx, int64_t* stats8) const;

 private:
  struct PlaneState {
    bool valid = false;
    int in_w = 0, in_h = 0, out_w = 0, out_h = 0;  // as given to generateMapForPlane
    int map_w = 0, map_h = 0;                      // scaled output size
    t360::DeviceBuffer map;                        // float2[map_h][map_w]
    t360::DeviceBuffer lut;                        // LutEntry[map_h][map_w]
    t360::DeviceBuffer col_tab, row_tab;           // per-column / per-row libm values (MapGenParams)
    // INTER_AREA shrink map_w x map_h -> out_w x out_h (only when the scale factors are not 1)
    struct ResizePlan {
      bool needed = false, supported = false, linear = false;
      int dw = -1, dh = -1;  // the output size the tables are for
      int iscale_x = 0, iscale_y = 0, xmax = 0;
      t360::DeviceBuffer xofs, x_si, x_alpha, yofs, y_si, y_alpha;
    } resize;
    // low-pass
    t360::FilterConfig filter;
    t360::DeviceBuffer segs, taps_q8, taps_f32, taps_pk, taps_sh;
    std::vector<int> seg_fast;       // per segment: eligible for the register-only Q8 kernel
    int fast_ky = 0;                 // vertical taps shared by the eligible segments (3, 5, 7; 0 = none)
    // the low-pass work lists of one plane size: every segment (`lp`), or -- when the long-batch gather plan fuses the
    // low-pass into most of its tiles -- only the segments its UNFUSED tiles still read (`lp_part`)
    struct LowpassLists {
      int tiles_w = -1, tiles_h = -1;  // plane size the tile lists were built for
      t360::DeviceBuffer tiles;
      int ntiles = 0, max_rows = 0;    // generic tiles covering EVERY listed segment (any alignment)
      // when the buffers are dword friendly: fast tiles of the eligible segments + generic tiles of the rest
      t360::DeviceBuffer tiles_fast, tiles_rest, tiles_wide;
      int nfast = 0, nrest = 0, max_rows_rest = 0, fast_lds_bytes = 0;
      int nwide = 0, wide_lds_bytes = 0;  // seg_fast[i] == 2: the segment's runs go to the wide fast path
      bool full_cover = false;
    } lp, lp_part;
    // fused low-pass tiles (t360_internal.h): what the planner needs, and the packed kernels on the device
    t360::FuseInfo fuse_info;
    t360::DeviceBuffer fuse_taps;
    bool fuse_ok = false;
    // LDS-tiled gather: work list planned on the host at init (t360_plan.cpp)
    struct GatherPlan {
      bool valid = false, tried = false;             // tried: planning was attempted (valid or not plannable)
      int ntiles = 0, ndirect = 0, ndirect_top = 0;  // staged tiles first in `tiles`, the direct tiles behind them
      int waves = 0, max_pieces = 0;                 // what it was planned for
      bool scatter = false;                          // chunk tables with block origins (scatter plan)
      t360::DeviceBuffer tiles, tlut, chunks;
      // the fused low-pass work list of the plan (nftiles > 0: the tiles above read the BLURRED plane, these the raw one) and
      // the segments the tiles above still need filtered
      int nftiles = 0;
      t360::DeviceBuffer ftiles, ftlut, fchunks;
      std::vector<uint8_t> seg_needed;
      t360::PlanStats stats;
    } plan, plan_small;  // plan_small: workgroups of 4 waves, for batches shorter than small_batch_ frames
    int plan_ks = 0;                       // taps per axis the plans are for; 0: the tiled kernel cannot take this map
  };

  bool check(hipError_t e, const char* what) const;
  bool ensureWeights();
  bool ensureTiles(PlaneState& p, PlaneState::LowpassLists& l, const std::vector<uint8_t>* needed, int w, int h, int imagePlaneIndex);
  bool buildGatherPlan(PlaneState& p, const t360::MapGenParams& P, int in_w, int in_h);
  bool ensureGatherPlan(PlaneState& p, bool small, std::vector<t360::LutEntry>* lut_of_this_call);
  // all-device core: a set of planes of n frames
  struct PlaneJob {
    const uint8_t* in;
    int64_t in_frame_bytes;
    int in_w, in_h, in_stride;
    uint8_t* out;
    int64_t out_frame_bytes;
    int out_w, out_h, out_stride;
    int idx;          // transformMatPlaneIndex
    int image_plane;  // imagePlaneIndex (messages only, as in the reference)
  };
  bool runPlanes(const PlaneJob* jobs, int njobs, int n_frames);
  bool runPlanesScaled(const PlaneJob* jobs, int njobs, int n_frames);
  bool buildResizePlan(PlaneState& p, int dw, int dh);
  void fillLowpassArgs(const PlaneState& p, const PlaneState::LowpassLists& l, const uint8_t* d_in, int64_t in_frame_bytes, int in_stride, uint8_t* d_out,
                       int64_t out_frame_bytes, int out_stride, int w, int h, t360::LowpassArgs* out) const;
  bool runLowpass(PlaneState& p, PlaneState::LowpassLists& l, const std::vector<uint8_t>* needed, const uint8_t* d_in, int64_t in_frame_bytes, int in_stride,
                  uint8_t* d_out, int64_t out_frame_bytes, int out_stride, int w, int h, int n_frames,
                  int imagePlaneIndex, hipStream_t stream);

  FrameTransformContext ctx_;
  bool ok_ = false;
  int device_ = 0;
  hipStream_t own_stream_ = nullptr;
  hipStream_t stream_ = nullptr;
  // the low-pass launches of planes 1.. of a batch run beside plane 0's (independent planes, small grids)
  hipStream_t lp_streams_[3] = {nullptr, nullptr, nullptr};
  hipEvent_t lp_fork_ = nullptr, lp_join_[3] = {nullptr, nullptr, nullptr};
  PlaneState planes_[t360::kMaxMaps];
  t360::DeviceBuffer weights_;  // Q15 table of ctx_.interpolation_alg
  t360::DeviceBuffer weights_pack_;  // bicubic weights re-packed for v_dot4 (tiled kernel)
  bool weights_ready_ = false;
  // LDS-tiled gather: workgroups of 8 waves (128x16 px tiles: half the tile borders of 64x16 -- fewer halo bytes and
  // fewer row fragments ending inside 128-byte lines another workgroup fetches again), 76 KiB of LDS each (2 per CU:
  // 16 waves, the same as 4 x 4, but half as many tiles' working sets contend for the XCD's L2), up to 24 KiB staged
  // per tile and frame; the ring keeps 3 frames of small tiles, 2 of the largest
  int max_pieces_ = 24;
  int ring_kb_ = 76;
  int waves_ = 8;
  int frames_per_block_ = 64;  // frames one workgroup of the tiled gather walks with one tile (fewer, longer-lived workgroups:
                               // their start-up -- tables, weights, first DMA -- is ~5 us against ~1 us per frame)
  int tail_percent_ = 12, tail_frames_ = 16;  // the last eighth of every XCD's tiles walks the batch in equal runs of <= 16
                                               // frames, at least two (64 frames: 4 x 16; 20: 10 + 10; 8: 4 + 4)
                                               // (short workgroups drain the launch; each pays the ~5 us start-up again,
                                               // so more than ~15 % costs more than it saves: measured 5 .. 35 %)
  int cus_ = 256;              // compute units of the device (how many workgroups a launch needs to fill it)
  int small_batch_ = 24;       // batches of fewer frames use the 4-wave plan (0: never); measured crossover 24 - 28
  static constexpr int kSmallPlanPieces = 12;
  int plan_wide_pct_ = 200, plan_strip_pct_ = 0, plan_band_ = -1, plan_row_pad_ = 0, plan_row_align_ = 8;  // PlanOptions
  int plan_wide256_pct_ = 0, plan_cost_lines_ = 0, plan_scatter_ = 0;
  bool use_tiled_ = true;
  const char* last_lowpass_ = "";
  char last_kernel_[64] = "";  // gather kernel of the most recent launch (reporting); the buffer lives as long as the handle
  void setLastKernel(const char* name) { snprintf(last_kernel_, sizeof(last_kernel_), "%s", name); }
  bool use_fast_lowpass_ = true;
  bool merge_lowpass_ = true;     // Y, U and V of a batch in one launch where the wide tiles serve all of them (T360_NO_MERGED_LOWPASS)
  bool use_wide_lowpass_ = true;  // ... and its wide-tile variant (instrumented build: T360_NO_WIDE_LOWPASS)
  bool fuse_side_stream_ = false; // the fused launch runs beside the low-pass and tiled launches of the call (instrumented build: T360_FUSED_SAME_STREAM)
  bool fuse_lowpass_ = false;     // long batches: tiles that can filter their own footprint do (T360_setFusedLowpass; off by
                                  // default: measured slower)
  // scratch planes: [0] for the calls on the handle's stream, [1 + lane] for the pipelined calls of that lane (calls on
  // different lanes overlap on the device and must not share intermediates)
  static constexpr int kMaxLanes = 4;
  t360::DeviceBuffer blurred_[1 + kMaxLanes];  // low-pass output, n_frames planes
  t360::DeviceBuffer scaled_[1 + kMaxLanes];   // supersampled (warp-map sized) planes before the INTER_AREA shrink
  int scratch_ = 0;                            // which set the running call uses
  // T360_transformFramesPipelined: lanes are created by the first pipelined call
  int pipe_depth_ = 2, pipe_next_ = 0;
  hipStream_t pipe_streams_[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t pipe_done_[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t pipe_fork_ = nullptr;
  bool pipe_busy_[kMaxLanes] = {false, false, false, false};  // work issued on the lane since the last join
  bool ensureLanes();
  bool drainLanes();  // host-side wait for every lane
  bool quiesceLanes();  // before per-map tables are rewritten in place: nothing may still be reading them
  t360::DeviceBuffer stage_in_, stage_out_;  // host-pointer path: device side
  t360::HostStager stager_;                  // ... the copies (contiguous where the strides allow; nothing is pinned or cached)
};
