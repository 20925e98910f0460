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
  // only `width` bytes of a caller row may be written (the rest is the caller's padding): one contiguous copy when the
  // rows have no padding on either side (what ffmpeg hands over for widths that are multiples of its alignment), else 2-D
  hipError_t e;
  if (host_stride == width && dev_stride == width)
    e = hipMemcpyAsync(host, dev, (size_t)width * (size_t)height, hipMemcpyDeviceToHost, stream);
  else
    e = hipMemcpy2DAsync(host, (size_t)host_stride, dev, (size_t)dev_stride, (size_t)width, (size_t)height,
                         hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess;
}

}  // namespace t360
