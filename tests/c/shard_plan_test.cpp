// tests/c/shard_plan_test.cpp -- CPU check of examples/t360_shard_plan.h (no HIP, no RCCL): every frame owned exactly once,
// every send met by exactly one receive of the same size, sink regions disjoint and gap-free, buffer alternation.
// Built and run by tests/test_host_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../examples/t360_shard_plan.h"

using namespace t360_example;

#define REQUIRE(c)                                                    \
  do {                                                                \
    if (!(c)) {                                                       \
      printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);           \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main() {
  for (int n = 1; n <= 8; n++)
    for (int frames : {1, 7, 8, 63, 64, 65}) {
      std::vector<int> owner((size_t)frames, -1);
      std::vector<int64_t> bytes_of((size_t)n);
      const int64_t frame_bytes = 2359296;
      for (int w = 0; w < n; w++) {
        int lo, hi;
        shard_range(frames, w, n, &lo, &hi);
        REQUIRE(lo <= hi && hi <= frames);
        REQUIRE(hi - lo == frames / n || hi - lo == frames / n + 1);
        for (int f = lo; f < hi; f++) {
          REQUIRE(owner[(size_t)f] == -1);
          owner[(size_t)f] = w;
        }
        bytes_of[(size_t)w] = (int64_t)(hi - lo) * frame_bytes;
      }
      for (int f = 0; f < frames; f++) REQUIRE(owner[(size_t)f] >= 0);
      // match sends and receives
      std::vector<P2POp> recvs = gather_ops(0, bytes_of);
      std::vector<char> met(recvs.size(), 0);
      int64_t at = bytes_of[0];
      for (const P2POp& r : recvs) {
        REQUIRE(!r.send && r.peer >= 1 && r.peer < n && r.bytes == bytes_of[(size_t)r.peer]);
        REQUIRE(r.offset == at);  // gap-free, in worker order: the sink is the stream in frame order
        at += r.bytes;
      }
      int sends = 0;
      for (int w = 1; w < n; w++) {
        const std::vector<P2POp> ops = gather_ops(w, bytes_of);
        REQUIRE(ops.size() == (bytes_of[(size_t)w] > 0 ? 1u : 0u));
        for (const P2POp& s : ops) {
          REQUIRE(s.send && s.peer == 0 && s.bytes == bytes_of[(size_t)w]);
          bool found = false;
          for (size_t i = 0; i < recvs.size(); i++)
            if (!met[i] && recvs[i].peer == w && recvs[i].bytes == s.bytes) {
              met[i] = 1;
              found = true;
              break;
            }
          REQUIRE(found);
          sends++;
        }
      }
      REQUIRE(sends == (int)recvs.size());
      int64_t total = 0;
      for (int64_t b : bytes_of) total += b;
      REQUIRE(at == total);
    }
  for (int k = 0; k < 10; k++) {
    REQUIRE(buffer_of_step(k) == (k & 1));
    REQUIRE(buffer_of_step(k) != buffer_of_step(k + 1));
    REQUIRE(step_waits_for_gather(k) == (k >= 2));
  }
  REQUIRE(!gather_possible({0}));
  REQUIRE(!gather_possible({0, 0}));
  REQUIRE(gather_possible({0, 1}));
  REQUIRE(!gather_possible({0, 1, 0}));
  printf("shard plan ok\n");
  return 0;
}
