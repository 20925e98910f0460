// examples/t360_shard_plan.h -- the host-side bookkeeping of the native multi-GPU driver (t360_multi_gpu.cpp), kept free
// of HIP and RCCL so that it can be compiled and checked on a machine without GPUs (tests/c/shard_plan_test.cpp):
// which frames a worker owns, which point-to-point operations a worker posts per step for the output gather, and which of
// its two output buffers a step uses.  SURVEY.md 8(e): whole frames are sharded, rank 0 collects outputs with grouped
// send / recv over xGMI; nothing here is on the data path of the transform itself.
#pragma once
#include <cstdint>
#include <vector>

namespace t360_example {

// contiguous block [lo, hi) of `n_frames` owned by worker `w` of `n` (the first n_frames % n workers get one more);
// the same rule as transform360_amd/sharding.py:shard_range
inline void shard_range(int n_frames, int w, int n, int* lo, int* hi) {
  const int base = n_frames / n, extra = n_frames % n;
  *lo = w * base + (w < extra ? w : extra);
  *hi = *lo + base + (w < extra ? 1 : 0);
}

struct P2POp {
  bool send;       // true: ncclSend to `peer`; false: ncclRecv from `peer`
  int peer;
  int64_t offset;  // recv: byte offset inside worker 0's sink; send: 0 (the worker's own output buffer)
  int64_t bytes;
};

// operations worker `w` posts (inside one ncclGroupStart / ncclGroupEnd) to gather every worker's `bytes_of[w]` output
// bytes of one step on worker 0; worker 0's own frames stay where they are
inline std::vector<P2POp> gather_ops(int w, const std::vector<int64_t>& bytes_of) {
  std::vector<P2POp> ops;
  const int n = (int)bytes_of.size();
  if (w != 0) {
    if (bytes_of[(size_t)w] > 0) ops.push_back(P2POp{true, 0, 0, bytes_of[(size_t)w]});
    return ops;
  }
  int64_t at = bytes_of[0];
  for (int r = 1; r < n; r++) {
    if (bytes_of[(size_t)r] > 0) ops.push_back(P2POp{false, r, at, bytes_of[(size_t)r]});
    at += bytes_of[(size_t)r];
  }
  return ops;
}

// double buffering of the outputs: step k writes buffer k & 1; before it may do so the gather of step k - 2 (the last
// reader of that buffer) must be over
inline int buffer_of_step(int k) { return k & 1; }
inline bool step_waits_for_gather(int k) { return k >= 2; }

// two workers may share a communicator only when they sit on different physical devices (RCCL refuses duplicates)
inline bool gather_possible(const std::vector<int>& device_of_worker) {
  for (size_t a = 0; a < device_of_worker.size(); a++)
    for (size_t b = a + 1; b < device_of_worker.size(); b++)
      if (device_of_worker[a] == device_of_worker[b]) return false;
  return device_of_worker.size() > 1;
}

}  // namespace t360_example
