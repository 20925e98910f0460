// t360_hoststage.cpp -- see t360_hoststage.h
#include "t360_hoststage.h"

#include <algorithm>

namespace t360 {

HostStager::~HostStager() {
  for (const Range& r : ranges_)
    if (r.pinned) {
      (void)hipHostUnregister(const_cast<void*>(r.base));
      (void)hipGetLastError();  // the caller may have freed the range meanwhile
    }
}

// Bookkeeping of one plane buffer: registered on its second sighting.
bool HostStager::touch(const void* base, size_t bytes) {
  clock_++;
  const uintptr_t lo = (uintptr_t)base, hi = lo + bytes;
  for (size_t i = 0; i < ranges_.size();) {
    Range& r = ranges_[i];
    const uintptr_t rlo = (uintptr_t)r.base, rhi = rlo + r.bytes;
    if (r.base == base && r.bytes == bytes) {
      r.last = clock_;
      if (!r.pinned) {
        if (hipHostRegister(const_cast<void*>(base), bytes, hipHostRegisterDefault) == hipSuccess)
          r.pinned = true;
        else
          (void)hipGetLastError();  // e.g. already pinned by the application: the copy below is fast anyway
      }
      return r.pinned;
    }
    if (rlo < hi && lo < rhi) {  // overlaps a different range: the caller's allocation changed
      if (r.pinned) {
        (void)hipHostUnregister(const_cast<void*>(r.base));
        (void)hipGetLastError();
      }
      ranges_.erase(ranges_.begin() + (long)i);
      continue;
    }
    i++;
  }
  if (ranges_.size() >= kMaxRanges) {
    auto victim = std::min_element(ranges_.begin(), ranges_.end(), [](const Range& a, const Range& b) { return a.last < b.last; });
    if (victim->pinned) {
      (void)hipHostUnregister(const_cast<void*>(victim->base));
      (void)hipGetLastError();
    }
    ranges_.erase(victim);
  }
  ranges_.push_back(Range{base, bytes, false, clock_});
  return false;
}

bool HostStager::to_device(const uint8_t* host, int width, int height, int host_stride, uint8_t* dev, int dev_stride,
                           hipStream_t stream, int* stride_used) {
  const size_t span = (size_t)host_stride * (size_t)(height - 1) + (size_t)width;
  const bool pinned = touch(host, span);
  hipError_t e;
  if (pinned && (host_stride & 15) == 0 && ((uintptr_t)host & 15) == 0 && host_stride <= width + width / 4) {
    *stride_used = host_stride;
    e = hipMemcpyAsync(dev, host, span, hipMemcpyHostToDevice, stream);
  } else {
    *stride_used = dev_stride;
    e = hipMemcpy2DAsync(dev, (size_t)dev_stride, host, (size_t)host_stride, (size_t)width, (size_t)height,
                         hipMemcpyHostToDevice, stream);
  }
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess;
}

bool HostStager::to_host(uint8_t* host, int width, int height, int host_stride, const uint8_t* dev, int dev_stride,
                         hipStream_t stream) {
  touch(host, (size_t)host_stride * (size_t)(height - 1) + (size_t)width);
  const hipError_t e = hipMemcpy2DAsync(host, (size_t)host_stride, dev, (size_t)dev_stride, (size_t)width, (size_t)height,
                                        hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess;
}

}  // namespace t360
