// t360_plan.h -- tile work list of the LDS-tiled gather for one map (see t360_plan.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "t360_internal.h"
#include "t360_kernels.h"
#include "t360_devbuf.h"

namespace t360 {

struct GatherPlan {
  bool valid = false;
  int ntiles = 0;            // staged tiles (first in `tiles`); the ndirect direct tiles follow them
  int n32 = 0, n16 = 0, nstrip = 0, ndirect = 0;
  int64_t staged_bytes = 0;  // sum of staged box bytes over the plane (L2 -> LDS traffic per frame)
  DeviceBuffer tiles;        // TileDesc[ntiles]
  DeviceBuffer tlut;         // box-relative LUT words
};

// d_lut: absolute LUT of the map (dw x dh entries), source plane sw x sh, ksize = taps per axis.
// max_box_bytes: largest staged box (the DMA ring must hold two of them).
bool build_gather_plan(const LutEntry* d_lut, int dw, int dh, int sw, int sh, int ksize, int max_box_bytes,
                       hipStream_t stream, GatherPlan* plan);

void pack_cubic_weights(const std::vector<int16_t>& q15_table, std::vector<uint32_t>* out);
void pack_weights(const std::vector<int16_t>& q15_table, int ks, std::vector<uint32_t>* out);

}  // namespace t360
