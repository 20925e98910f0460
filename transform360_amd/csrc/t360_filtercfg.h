// t360_filtercfg.h -- host-side low-pass configuration (see t360_filtercfg.cpp).
#pragma once

#include <algorithm>
#include <stdint.h>

#include <vector>

#include "t360_internal.h"

namespace t360 {

// One segment: rectangle (SegmentFilteringConfig, reference VideoFrameTransform.h:25-38) plus
// its two 1-D kernels (filterKernelsX_/Y_, :150-155) and their Q8 integer form.
struct Segment {
  int left = 0, top = 0, width = 0, height = 0;
  std::vector<float> kx, ky;
  std::vector<int> kx_q8, ky_q8;
  bool fixed_point = false;
};

struct FilterConfig {
  std::vector<Segment> segments;
};

// Shifted tap variants of the wide low-pass path (t360_lowpass.hip): the row pass of output pixel px0 + j (px0 % 4 == 0,
// j = 0..3) is a dot product of the ALIGNED source dwords from byte (px0 - rx - m) on, rx = taps / 2, m = (-rx) & 3,
// with the taps shifted by m + j bytes.  Appends 4 x kWideTapStride dwords (variant j at [j * kWideTapStride]) to `out`
// and returns the number of dwords a variant can be non-zero in, or 0 (nothing appended) when that exceeds kWideMaxNd.
int pack_shifted_taps(const std::vector<int>& kx_q8, std::vector<uint32_t>* out);

// Fused low-pass tiles (t360_internal.h, t360_plan.h FuseInfo): the plane's fusable kernels -- fixed-point segments with
// <= 7 horizontal and exactly 3 vertical byte-sized taps -- deduplicated and packed (kFusedTapDwords dwords each, layout in
// t360_internal.h), and per source row of a w x h plane the kernel EVERY pixel of the row is filtered with (-1: the
// row's segments disagree, are not fusable, or do not cover it).  False (nothing to fuse) for stereo inputs, planes whose
// segments do not fit, and configurations without a fusable row.
struct FuseInfo;
void pack_fused_taps(const std::vector<int>& kx_q8, const std::vector<int>& ky_q8, uint32_t* out /*[kFusedTapDwords]*/);
bool build_fuse_info(const FrameTransformContext& ctx, const FilterConfig& cfg, int w, int h, FuseInfo* info,
                     std::vector<uint32_t>* packed_taps);

// Reference calcualteFilteringConfig (VideoFrameTransform.cpp:367-501) for one plane shape.
// inputWidth/Height: plane size; outputWidth/Height: the SCALED output size (:560-565).
bool build_filter_config(const FrameTransformContext& ctx, int inputWidth, int inputHeight,
                         int outputWidth, int outputHeight, FilterConfig* cfg);

}  // namespace t360
