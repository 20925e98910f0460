// t360_hoststage.cpp -- see t360_hoststage.h
#include "t360_hoststage.h"

namespace t360 {

bool HostStager::to_device(const uint8_t* host, int width, int height, int host_stride, uint8_t* dev, int dev_stride,
                           hipStream_t stream, int* stride_used) {
  const size_t span = (size_t)host_stride * (size_t)(height - 1) + (size_t)width;
  hipError_t e;
  // contiguous when the caller's rows are 16-byte friendly (the gather stages 16-byte chunks) and the padding is small
  if ((host_stride & 15) == 0 && ((uintptr_t)host & 15) == 0 && host_stride >= width && host_stride <= width + width / 4) {
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
  // only `width` bytes of a caller row may be written (the rest is the caller's padding): a 2-D copy.  (One contiguous
  // hipMemcpyAsync for unpadded rows was measured: 3-4 % FEWER frames per second through the literal ABI on the same
  // box, 2 547 vs 2 644 -- into pageable memory the runtime's pitched path is the faster one.)
  const hipError_t e = hipMemcpy2DAsync(host, (size_t)host_stride, dev, (size_t)dev_stride, (size_t)width, (size_t)height,
                                        hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess;
}

}  // namespace t360
