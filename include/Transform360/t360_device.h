/*
 * transform360-mi355x: ADDITIVE entry points (not present in the reference).
 *
 * The four reference symbols (VideoFrameTransformHandler.h) are synchronous and take one
 * plane of one frame; through host pointers they are PCIe-bound.  A caller
 * that keeps frames resident in HBM -- a hardware decoder feeding an encoder, the benchmark,
 * the multi-GPU frame-sharding driver -- uses the calls below instead: same handle, same maps
 * (VideoFrameTransform_generateMapForPlane must have been called for every map index used),
 * same arithmetic, but whole batches of frames per launch on a caller-chosen HIP stream.
 *
 * All functions return 1 on success and 0 on failure (message on stdout), like the
 * reference's C entry points (reference VideoFrameTransformHandler.cpp:26-64).
 * Plain C ABI: pointers, integers, no HIP or C++ types in the signatures; a hipStream_t is
 * passed as void*.
 */
#ifndef TRANSFORM360_T360_DEVICE_H
#define TRANSFORM360_T360_DEVICE_H

#include <stdint.h>

#include "VideoFrameTransformHandler.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One image plane inside a frame buffer (what one call of
 * VideoFrameTransform_transformFramePlane describes, reference vf_transform360.c:368-397). */
typedef struct T360PlaneDesc {
  int64_t in_offset;  /* byte offset of the plane inside one INPUT frame  */
  int64_t out_offset; /* byte offset of the plane inside one OUTPUT frame */
  int in_stride;      /* bytes per input row  (>= in_width)  */
  int out_stride;     /* bytes per output row (>= out_width) */
  int in_width, in_height;
  int out_width, out_height;
  int map_index;      /* transformMatPlaneIndex: 0 = luma-shaped map, 1 = chroma-shaped map */
} T360PlaneDesc;

/* Library version string, e.g. "transform360-mi355x 0.1 (gfx950)". */
const char* T360_version(void);

/* Number of visible HIP devices (0 if the runtime cannot be initialised). */
int T360_deviceCount(void);

/* Use `hip_stream` (a hipStream_t) for all device work of this handle.  NULL means HIP's NULL
 * (legacy default) stream, as everywhere in the HIP API.  A new handle starts on a private
 * non-blocking stream; T360_useOwnStream returns to it.  Device-pointer calls of the reference
 * ABI still synchronise before returning; the batch calls below do not. */
int T360_setStream(VideoFrameTransform* transform, void* hip_stream);
int T360_useOwnStream(VideoFrameTransform* transform);

/* Block until everything queued on the handle's stream has finished. */
int T360_synchronize(VideoFrameTransform* transform);

/* Transform `n_frames` frames that live in device memory:
 *   frame k input  = d_in  + k * in_frame_bytes,  frame k output = d_out + k * out_frame_bytes,
 * each holding `n_planes` planes laid out as `planes[]` says.  Asynchronous on the handle's
 * stream.  Equivalent to n_frames * n_planes calls of VideoFrameTransform_transformFramePlane. */
int T360_transformFrames(VideoFrameTransform* transform,
                         const uint8_t* d_in, int64_t in_frame_bytes,
                         uint8_t* d_out, int64_t out_frame_bytes,
                         int n_frames, const T360PlaneDesc* planes, int n_planes);

/* The same work as T360_transformFrames for a STREAM of batches: consecutive pipelined calls are taken to be
 * independent of each other (they read and write different buffers, as the batches of a frame stream do) and are issued
 * round-robin on `depth` internal HIP streams ("lanes") of the handle, so that the workgroups of call k+1 start while
 * the last ones of call k drain (a launch of the tiled gather ends on a partly empty GPU, and a short batch -- 8 frames
 * per GPU when 64 are sharded over 8 GPUs, SURVEY.md 8e -- spends a third of its life filling and draining).
 *   - ordering IN: every pipelined call starts after everything queued on the handle's stream at the time of the call;
 *   - ordering OUT: nothing waits for a lane by itself.  T360_pipelineJoin makes the handle's stream wait (on the device,
 *     the host does not block) for every pipelined call issued so far; T360_synchronize blocks the host until the
 *     handle's stream AND all lanes are idle;
 *   - calls k and k + depth run on the same lane, in order: an output buffer may be reused every `depth` calls.
 * depth: 1..4, default 2 (set with T360_setPipelineDepth before the first pipelined call or after a join).
 * Not for use while the handle's stream is being captured into a HIP graph: a pipelined call asks the stream whether it is
 * idle (hipStreamQuery) and, when tables are rebuilt, waits for the device -- both end a capture.  T360_transformFrames has
 * no such restriction.
 * Same arithmetic, same kernels, same return convention as T360_transformFrames. */
int T360_transformFramesPipelined(VideoFrameTransform* transform,
                                  const uint8_t* d_in, int64_t in_frame_bytes,
                                  uint8_t* d_out, int64_t out_frame_bytes,
                                  int n_frames, const T360PlaneDesc* planes, int n_planes);
/* n_calls pipelined calls issued back to back from native code: call k reads d_in[k] and writes d_out[k] (same frame
 * sizes, frame count and planes for all).  Exactly equivalent to the loop
 *     for (k = 0; k < n_calls; k++) T360_transformFramesPipelined(t, d_in[k], ..., d_out[k], ...);
 * for callers whose own loop is slow next to a 35-us step (an interpreter: bench.py).  Stops at the first failure. */
int T360_transformFramesPipelinedMany(VideoFrameTransform* transform, int n_calls,
                                      const uint8_t* const* d_in, int64_t in_frame_bytes,
                                      uint8_t* const* d_out, int64_t out_frame_bytes,
                                      int n_frames, const T360PlaneDesc* planes, int n_planes);
int T360_setPipelineDepth(VideoFrameTransform* transform, int depth);
int T360_pipelineJoin(VideoFrameTransform* transform);

/* Low-pass contexts, batches of >= 24 frames, bilinear / bicubic, MONO input: on = 1 makes the gather tiles whose source rows
 * all have fixed-point kernels of <= 7 horizontal and 3 vertical taps filter their own footprint in LDS (ONE pass over the raw
 * plane: reference filterPlane feeding remap, VideoFrameTransform.cpp:727-733 + :748-754, without the blurred plane's round
 * trip through HBM); the other tiles and the segments they read keep the two-pass path.  Same bytes either way (bit-exact).
 * OFF by default: on MI355X the filter is bound by integer VALU issue, not by HBM, and the fused kernel issues more of it
 * (BASELINE config 3 0.68 ms fused against 0.55 ms two-pass).  Call before VideoFrameTransform_generateMapForPlane. */
int T360_setFusedLowpass(VideoFrameTransform* transform, int on);

/* Low-pass stage only (reference filterPlane, VideoFrameTransform.cpp:621-704) on one
 * device-resident plane; asynchronous.  For parity tests of the segmented filter. */
int T360_filterPlane(VideoFrameTransform* transform, const uint8_t* d_in, uint8_t* d_out,
                     int width, int height, int in_stride, int out_stride, int map_index);

/* ---- introspection of init-time state (parity tests against the oracle) ---- */

/* Size of the warp map of `map_index` (the reference's warpMats_[idx], VideoFrameTransform.h:148). */
int T360_getMapSize(VideoFrameTransform* transform, int map_index, int* width, int* height);
/* Copy the float (x,y) pairs of the warp map to host memory: 2*width*height floats. */
int T360_copyMap(VideoFrameTransform* transform, int map_index, float* host_dst);
/* Low-pass segments of `map_index` (segmentFilteringConfigs_/filterKernelsX_/Y_, :150-159). */
int T360_getSegmentCount(VideoFrameTransform* transform, int map_index);
/* rect4 = left, top, width, height; lens2 = taps of kX, kY; *fixed_point = 1 if the segment
 * runs the 8-bit fixed-point filter path, 0 for the float path. */
int T360_getSegment(VideoFrameTransform* transform, int map_index, int i, int* rect4, int* lens2,
                    int* fixed_point);
int T360_copySegmentKernels(VideoFrameTransform* transform, int map_index, int i, float* kx, float* ky);

/* ---- what ran (benchmark reporting) ---- */

/* Name of the gather kernel the most recent transform call of this handle launched, e.g.
 * "remap_tiled_kernel<4, 76, 8>" (taps per axis, KiB of LDS per workgroup, waves per workgroup; batches of fewer than 24
 * frames and single-plane calls run "<4, 38, 4>") or "remap_gather_kernel"; "" before the first call.  The pointer
 * stays valid as long as the handle; the characters behind it change with the next transform call. */
const char* T360_lastKernel(VideoFrameTransform* transform);
/* How the most recent call's low-pass stage was launched: "merged" (the planes of a batch in one launch), "per-plane", or ""
 * (no low-pass yet).  For tests that pin a path. */
const char* T360_lastLowpassPath(VideoFrameTransform* transform);
/* Gather plan of `map_index` (the one long batches use): stats8 = staged tiles, direct (unstaged) tiles, source bytes fetched per frame
 * by the staged tiles, bytes of LDS filled per frame (one copy), pixels in direct tiles, bytes of the tile
 * tables on the device, scatter tiles among the staged ones (0 in the shipped configuration), 0.  Returns 0 when the plane has no tile plan (it then uses the general gather). */
int T360_getPlanStats(VideoFrameTransform* transform, int map_index, int64_t* stats8);
/* 0 for the shipped library.  Non-zero for a library compiled with -DT360_INSTRUMENT, which reads tuning
 * switches from the environment (development only; bench.py refuses to report numbers from it). */
int T360_buildFlags(void);

/* ---- synthetic stream generator (benchmark / tests) ---- */

/* Fill device memory with counter-based noise: byte i = top 8 bits of
 * splitmix64(seed + i).  Asynchronous on `hip_stream` (NULL = default stream).
 * The same function evaluated on the host gives identical bytes. */
int T360_fillNoise(uint8_t* d_dst, int64_t nbytes, uint64_t seed, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* TRANSFORM360_T360_DEVICE_H */
