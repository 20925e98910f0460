// t360_capi.cpp -- the exported C ABI of libTransform360.
//
// The first four functions are the reference's own entry points
// (reference Transform360/Library/VideoFrameTransformHandler.h:24-47, bodies
// VideoFrameTransformHandler.cpp:18-64) with the same names, argument meaning and error
// convention (1 = ok, 0 = failure, NULL from _new, _delete(NULL) is a no-op), so the
// unmodified vf_transform360.c links against this library.  No exception leaves this file.
// The T360_* functions are additive (include/Transform360/t360_device.h).
#include <cstdio>
#include <exception>
#include <new>

#include "Transform360/t360_device.h"
#include "t360_transform.h"

#define T360_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

template <typename F>
int guarded(const char* what, F&& f) {
  try {
    return f() ? 1 : 0;
  } catch (const std::exception& ex) {
    printf("%s. Error: %s\n", what, ex.what());
  } catch (...) {
    printf("%s. Error: unknown exception\n", what);
  }
  return 0;
}

}  // namespace

T360_EXPORT VideoFrameTransform* VideoFrameTransform_new(FrameTransformContext* ctx) {
  if (!ctx) return nullptr;
  try {
    VideoFrameTransform* t = new (std::nothrow) VideoFrameTransform(ctx);
    if (t && !t->ok()) {
      delete t;
      return nullptr;
    }
    return t;
  } catch (...) {
    return nullptr;
  }
}

T360_EXPORT void VideoFrameTransform_delete(VideoFrameTransform* transform) {
  try {
    delete transform;
  } catch (...) {
  }
}

T360_EXPORT int VideoFrameTransform_generateMapForPlane(VideoFrameTransform* transform, int inputWidth,
                                                        int inputHeight, int outputWidth, int outputHeight,
                                                        int transformMatPlaneIndex) {
  if (!transform) return 0;
  return guarded("Could not generate map", [&] {
    return transform->generateMapForPlane(inputWidth, inputHeight, outputWidth, outputHeight,
                                          transformMatPlaneIndex);
  });
}

T360_EXPORT int VideoFrameTransform_transformFramePlane(VideoFrameTransform* transform, uint8_t* inputData,
                                                        uint8_t* outputData, int inputWidth, int inputHeight,
                                                        int inputWidthWithPadding, int outputWidth,
                                                        int outputHeight, int outputWidthWithPadding,
                                                        int transformMatPlaneIndex, int imagePlaneIndex) {
  if (!transform) return 0;
  return guarded("Could not transform the plane", [&] {
    return transform->transformFramePlane(inputData, outputData, inputWidth, inputHeight, inputWidthWithPadding,
                                          outputWidth, outputHeight, outputWidthWithPadding,
                                          transformMatPlaneIndex, imagePlaneIndex);
  });
}

// ---------------------------------------------------------------------------------------------

T360_EXPORT const char* T360_version(void) { return "transform360-mi355x 0.1 (gfx950)"; }

T360_EXPORT int T360_deviceCount(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

T360_EXPORT int T360_setStream(VideoFrameTransform* t, void* s) {
  if (!t) return 0;
  return guarded("T360_setStream", [&] { return t->setStream(s); });
}

T360_EXPORT int T360_useOwnStream(VideoFrameTransform* t) {
  if (!t) return 0;
  return guarded("T360_useOwnStream", [&] { return t->useOwnStream(); });
}

T360_EXPORT int T360_synchronize(VideoFrameTransform* t) {
  if (!t) return 0;
  return guarded("T360_synchronize", [&] { return t->synchronize(); });
}

T360_EXPORT int T360_transformFrames(VideoFrameTransform* t, const uint8_t* d_in, int64_t in_frame_bytes,
                                     uint8_t* d_out, int64_t out_frame_bytes, int n_frames,
                                     const T360PlaneDesc* planes, int n_planes) {
  if (!t) return 0;
  return guarded("T360_transformFrames", [&] {
    return t->transformFrames(d_in, in_frame_bytes, d_out, out_frame_bytes, n_frames, planes, n_planes);
  });
}

T360_EXPORT int T360_transformFramesPipelined(VideoFrameTransform* t, const uint8_t* d_in, int64_t in_frame_bytes,
                                              uint8_t* d_out, int64_t out_frame_bytes, int n_frames,
                                              const T360PlaneDesc* planes, int n_planes) {
  if (!t) return 0;
  return guarded("T360_transformFramesPipelined", [&] {
    return t->transformFramesPipelined(d_in, in_frame_bytes, d_out, out_frame_bytes, n_frames, planes, n_planes);
  });
}

T360_EXPORT int T360_transformFramesPipelinedMany(VideoFrameTransform* t, int n_calls, const uint8_t* const* d_in,
                                                  int64_t in_frame_bytes, uint8_t* const* d_out, int64_t out_frame_bytes,
                                                  int n_frames, const T360PlaneDesc* planes, int n_planes) {
  if (!t || n_calls < 0 || (n_calls > 0 && (!d_in || !d_out))) return 0;
  return guarded("T360_transformFramesPipelinedMany", [&] {
    for (int k = 0; k < n_calls; k++)
      if (!t->transformFramesPipelined(d_in[k], in_frame_bytes, d_out[k], out_frame_bytes, n_frames, planes, n_planes)) return false;
    return true;
  });
}

T360_EXPORT int T360_setFusedLowpass(VideoFrameTransform* t, int on) {
  if (!t) return 0;
  return guarded("T360_setFusedLowpass", [&] { return t->setFusedLowpass(on != 0); });
}

T360_EXPORT int T360_setPipelineDepth(VideoFrameTransform* t, int depth) {
  if (!t) return 0;
  return guarded("T360_setPipelineDepth", [&] { return t->setPipelineDepth(depth); });
}

T360_EXPORT int T360_pipelineJoin(VideoFrameTransform* t) {
  if (!t) return 0;
  return guarded("T360_pipelineJoin", [&] { return t->pipelineJoin(); });
}

T360_EXPORT int T360_filterPlane(VideoFrameTransform* t, const uint8_t* d_in, uint8_t* d_out, int width,
                                 int height, int in_stride, int out_stride, int map_index) {
  if (!t) return 0;
  return guarded("T360_filterPlane",
                 [&] { return t->filterPlane(d_in, d_out, width, height, in_stride, out_stride, map_index); });
}

T360_EXPORT int T360_getMapSize(VideoFrameTransform* t, int map_index, int* width, int* height) {
  if (!t || !width || !height) return 0;
  return t->getMapSize(map_index, width, height) ? 1 : 0;
}

T360_EXPORT int T360_copyMap(VideoFrameTransform* t, int map_index, float* host_dst) {
  if (!t) return 0;
  return guarded("T360_copyMap", [&] { return t->copyMap(map_index, host_dst); });
}

T360_EXPORT int T360_getSegmentCount(VideoFrameTransform* t, int map_index) {
  return t ? t->segmentCount(map_index) : 0;
}

T360_EXPORT int T360_getSegment(VideoFrameTransform* t, int map_index, int i, int* rect4, int* lens2,
                                int* fixed_point) {
  if (!t || !rect4 || !lens2) return 0;
  return t->getSegment(map_index, i, rect4, lens2, fixed_point) ? 1 : 0;
}

T360_EXPORT int T360_copySegmentKernels(VideoFrameTransform* t, int map_index, int i, float* kx, float* ky) {
  if (!t || !kx || !ky) return 0;
  return t->copySegmentKernels(map_index, i, kx, ky) ? 1 : 0;
}

T360_EXPORT const char* T360_lastKernel(VideoFrameTransform* t) { return t ? t->lastKernel() : ""; }
T360_EXPORT const char* T360_lastLowpassPath(VideoFrameTransform* t) { return t ? t->lastLowpassPath() : ""; }

T360_EXPORT int T360_getPlanStats(VideoFrameTransform* t, int map_index, int64_t* stats8) {
  if (!t || !stats8) return 0;
  return t->planStats(map_index, stats8) ? 1 : 0;
}

T360_EXPORT int T360_buildFlags(void) {
#ifdef T360_INSTRUMENT
  return 1;
#else
  return 0;
#endif
}

T360_EXPORT int T360_fillNoise(uint8_t* d_dst, int64_t nbytes, uint64_t seed, void* hip_stream) {
  if (!d_dst || nbytes < 0) return 0;
  hipError_t e = t360::launch_fill_noise(d_dst, nbytes, seed, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) {
    printf("transform360: T360_fillNoise failed: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return 0;
  }
  return 1;
}
