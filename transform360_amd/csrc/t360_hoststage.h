// t360_hoststage.h -- host-pointer staging of the reference ABI.
//
// VideoFrameTransform_transformFramePlane is called by ffmpeg with plain malloc'd planes and must be complete when it
// returns (reference VideoFrameTransform.cpp:1319-1351, vf_transform360.c:368-397).  Pageable memory is copied by the
// HIP runtime through its own bounce buffers (~30 GB/s on the test host); a buffer that is PINNED moves at the PCIe
// rate.  ffmpeg recycles its frame buffers (buffer pools), so a plane seen for the second time at the same address
// and size is registered with the runtime (hipHostRegister: a userptr mapping of the VIRTUAL range -- if the caller
// frees and reallocates it the driver re-resolves the pages) and stays registered in a small LRU cache.  The first
// sighting, and anything that fails to register, takes the runtime's pageable path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace t360 {

class HostStager {
 public:
  HostStager() = default;
  ~HostStager();
  HostStager(const HostStager&) = delete;
  HostStager& operator=(const HostStager&) = delete;

  // caller rows (host_stride apart, `width` meaningful bytes) -> device rows, asynchronous on `stream`.  The device
  // buffer must hold max(host_stride, dev_stride) * height bytes; *stride_used is dev_stride, or host_stride when a
  // pinned plane went over as ONE contiguous copy (the DMA engine's fastest path; line padding travels with it).
  bool to_device(const uint8_t* host, int width, int height, int host_stride, uint8_t* dev, int dev_stride, hipStream_t stream,
                 int* stride_used);
  // device rows -> caller rows, asynchronous on `stream` (the caller synchronises the stream)
  bool to_host(uint8_t* host, int width, int height, int host_stride, const uint8_t* dev, int dev_stride, hipStream_t stream);

 private:
  struct Range {
    const void* base;
    size_t bytes;
    bool pinned;      // registered with the runtime
    uint64_t last;    // LRU stamp
  };
  static constexpr size_t kMaxRanges = 24;
  bool touch(const void* base, size_t bytes);  // true: the range is pinned
  std::vector<Range> ranges_;
  uint64_t clock_ = 0;
};

}  // namespace t360
