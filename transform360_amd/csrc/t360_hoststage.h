// t360_hoststage.h -- host-pointer staging of the reference ABI.
//
// VideoFrameTransform_transformFramePlane is called by ffmpeg with plain malloc'd planes and must be complete when it
// returns (reference VideoFrameTransform.cpp:1319-1351, vf_transform360.c:368-397).  What matters for the rate is the
// SHAPE of the copy: one contiguous hipMemcpyAsync of the plane's whole span (line padding travels with it) moves at
// the PCIe rate straight from the caller's pageable memory on this stack (53 GB/s, tools/ubench/h2d_pageable.hip),
// a pitched 2-D copy does not.  The caller's memory is NOT registered with the runtime: a cached hipHostRegister goes
// stale when the caller frees a buffer and a later one takes its address (the driver unmaps the range from the GPU on
// munmap and does not map the new pages again), and the next copy through it is a GPU memory fault -- found by
// tests/soak/host_soak.py; round 2 shipped such a cache for a few hours and gained nothing over the contiguous copy.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace t360 {

class HostStager {
 public:
  // caller rows (host_stride apart, `width` meaningful bytes) -> device rows, asynchronous on `stream`.  The device
  // buffer must hold max(host_stride, dev_stride) * height bytes; *stride_used is dev_stride, or host_stride when the
  // plane went over as ONE contiguous copy.
  bool to_device(const uint8_t* host, int width, int height, int host_stride, uint8_t* dev, int dev_stride, hipStream_t stream,
                 int* stride_used);
  // device rows -> caller rows, asynchronous on `stream` (the caller synchronises the stream)
  bool to_host(uint8_t* host, int width, int height, int host_stride, const uint8_t* dev, int dev_stride, hipStream_t stream);
};

}  // namespace t360
