// t360_devbuf.h -- RAII device allocation used by the host side of libTransform360.
#pragma once

#include <stddef.h>

namespace t360 {

// hipMalloc'd buffer that only ever grows
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  bool reserve(size_t bytes);  // contents are NOT preserved on growth
  void release();
  template <typename T>
  T* as() const { return static_cast<T*>(ptr_); }
  size_t size() const { return bytes_; }

 private:
  void* ptr_ = nullptr;
  size_t bytes_ = 0;
};

}  // namespace t360
