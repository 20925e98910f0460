// t360_filtercfg.cpp -- init-time low-pass configuration (host side, tiny).
//
// Builds, for one plane shape, the list of segment rectangles and their separable 1-D
// Gaussian kernels exactly as the reference does on the host
//   calcualteFilteringConfig            reference VideoFrameTransform.cpp:367-501
//   generateKernelsAndFilteringConfigs  :318-364
//   generateKernelAndFilteringConfig    :210-297
//   calculateKernel                     :78-94
//   getEffectiveRatio & friends         :126-170
// and classifies every segment the way cv::sepFilter2D's createSeparableLinearFilter would
// (8-bit fixed-point path iff both kernels are SMOOTH|SYMMETRICAL),
// producing the packed tap arrays the HIP low-pass kernel consumes.
//
// Attribution: the float expressions of Builder::band and effective_ratio below follow, operation by operation, the
// functions listed above of facebook/transform360 (Copyright (c) 2015-present, Facebook, Inc., BSD license, see that
// project's LICENSE file): segment kernels have to come out identical to the float bit, and there is one way to get
// there.  Data structures, classification and packing are this library's own.
//
// The reference is C++ with `using namespace std`; calls with float arguments bind to the float
// overloads (cos(angle) is cosf) and mixed expressions promote to double -- both are spelled
// out below.  Built with -ffp-contract=off.
#include "t360_filtercfg.h"
#include "t360_plan.h"

#include <cfloat>
#include <cmath>
#include <cstdio>

namespace t360 {

namespace {

constexpr double kEps = 1e-9;                 // reference kEpsilon (:33)
const double kSphereArea = 4 * M_PI;          // :34
const double kFov = 0.5333 * M_PI;            // :35

// calculateKernel (:78-94).  `kernel /= sum` on a CV_32F cv::Mat is
// convertTo(kernel, -1, 1./sum): every tap is multiplied by float(1.0 / sum).
// A negative sigma (a band whose angle lies beyond 90 degrees: planes a few rows high cut into more bands than that)
// asks the reference for a cv::Mat of negative width: Mat::zeros throws, generateMapForPlane catches and returns
// false (:78-81, :571-576).  Same outcome here.
struct NegativeKernel {};
std::vector<float> gaussian_taps(float sigma) {
  int half = (int)(sigma * 2);
  if (half < 0) throw NegativeKernel{};
  std::vector<float> k((size_t)half * 2 + 1);
  float sum = 0;
  const float comp = std::fabs(sigma) < kEps ? 0.0f : (float)(0.5 / (sigma * sigma));
  for (int u = -half; u <= half; ++u) {
    const float v = expf(-((float)(u * u) * comp));
    k[(size_t)(u + half)] = v;
    sum += v;
  }
  const float inv = (float)(1.0 / (double)sum);
  for (float& v : k) v = v * inv;
  return k;
}

double angular_distance(double yaw1, double pitch1, double yaw2, double pitch2) {  // :126-130
  return std::acos(std::sin(pitch1) * std::sin(pitch2) +
                   std::cos(pitch1) * std::cos(pitch2) * std::cos(yaw1 - yaw2));
}
double sampling_arc(double offset, double renderArc) {  // :132-134
  return M_PI - 2 * std::atan2(std::cos(0.5 * renderArc) - offset, std::sin(0.5 * renderArc));
}
double spherical_area(double angle) { return (1 - std::cos(0.5 * angle)) * 2 * M_PI; }  // :136-138

double effective_ratio(double angularDist, double offset) {  // :140-170 with fov = kFov
  const double fov = kFov;
  double major;
  if (angularDist - kEps > fov / 2) {
    if (angularDist + fov / 2 > M_PI) {
      const double edge1 = sampling_arc(offset, (2 * M_PI - angularDist - fov / 2) * 2) / 2;
      const double edge2 = sampling_arc(offset, (angularDist - fov / 2) * 2) / 2;
      major = (2 * M_PI - edge1 - edge2) / fov;
    } else {
      major = (sampling_arc(offset, 2 * angularDist + fov) - sampling_arc(offset, 2 * angularDist - fov)) / 2 / fov;
    }
  } else {
    major = (sampling_arc(offset, 2 * angularDist + fov) + sampling_arc(offset, fov - 2 * angularDist)) / 2 / fov;
  }
  const double coVertex = angular_distance(angularDist, 0.5 * fov, 0.0, 0.0);
  const double minor = sampling_arc(offset, coVertex * 2) / (coVertex * 2);
  const double r = major * minor * spherical_area(fov) / kSphereArea;
  return r > 1.0 ? 1.0 : r;
}

// cv::getKernelType for a 1-D kernel anchored at its centre
enum { kSymmetrical = 1, kAsymmetrical = 2, kSmooth = 4, kInteger = 8 };
int kernel_type(const std::vector<float>& k) {
  const int n = (int)k.size();
  int type = kSmooth + kInteger;
  if (n & 1) type |= kSymmetrical + kAsymmetrical;
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double a = k[(size_t)i], b = k[(size_t)(n - i - 1)];
    if (a != b) type &= ~kSymmetrical;
    if (a != -b) type &= ~kAsymmetrical;
    if (a < 0) type &= ~kSmooth;
    if (a != (double)std::lrint(a)) type &= ~kInteger;
    sum += a;
  }
  if (std::fabs(sum - 1) > FLT_EPSILON * (std::fabs(sum) + 1)) type &= ~kSmooth;
  return type;
}

struct Builder {
  const FrameTransformContext& c;
  FilterConfig& out;
  int inW, inH;

  // one vertical band [top, bottom] split into horizontal tiles (:210-297)
  void band(int top, int bottom, float angle, float sigmaY, const std::vector<float>& kernelY) {
    const double a = 0.5 * inW;
    const double b = sigmaY / (cosf(angle) + kEps);
    const float sigmaX = (float)(b < a ? b : a);
    const std::vector<float> kernelX = gaussian_taps(sigmaX);

    const int nH = c.adjust_kernel ? c.num_horizontal_segments : 1;
    const int segW = (int)std::ceil(1.0 * inW / nH);
    const double baseRatio = effective_ratio(0.0, 0.0);

    for (int i = 0; i < nH && i * segW < inW; ++i) {
      Segment s;
      s.left = i * segW;
      s.top = top;
      s.width = std::min(segW, inW - i * segW);
      s.height = bottom - top + 1;
      if (c.adjust_kernel) {
        const float avgYaw = (float)(2 * M_PI * ((i * segW + 0.5 * std::min(segW, inW - i * segW)) - 0.5 * inW) / inW);
        const float avgPitch = (float)(0.5 * M_PI * (inH - top - bottom) / inH);
        float yaw = (float)(c.fixed_yaw * M_PI / 180.0f);
        float pitch = (float)(c.fixed_pitch * M_PI / 180.0f);
        float offset = std::fabs(c.fixed_cube_offcenter_z);
        if (std::fabs(yaw) < kEps && std::fabs(pitch) < kEps &&
            (std::fabs(c.fixed_cube_offcenter_x) > kEps || std::fabs(c.fixed_cube_offcenter_y) > kEps ||
             c.fixed_cube_offcenter_z > kEps)) {
          offset = sqrtf(c.fixed_cube_offcenter_x * c.fixed_cube_offcenter_x +
                         c.fixed_cube_offcenter_y * c.fixed_cube_offcenter_y +
                         c.fixed_cube_offcenter_z * c.fixed_cube_offcenter_z);
          yaw = atan2f(-c.fixed_cube_offcenter_x / offset, -c.fixed_cube_offcenter_z / offset);
          pitch = asinf(-c.fixed_cube_offcenter_y / offset);
        }
        const double dist = angular_distance(yaw, pitch, avgYaw, avgPitch);
        const double ratio = effective_ratio(dist, offset);
        const double scale = c.kernel_adjust_factor * baseRatio / ratio;
        s.kx = gaussian_taps((float)(scale * sigmaX));
        s.ky = gaussian_taps((float)(scale * sigmaY));
      } else {
        s.kx = kernelX;
        s.ky = kernelY;
      }
      out.segments.push_back(std::move(s));
    }
  }

  // bands above `startBottom` going up, then below `startTop` going down (:318-364)
  void halves(int startTop, int startBottom, float sigmaY, const std::vector<float>& kernelY, int bandH) {
    for (int bottom = startBottom; bottom >= 0; bottom -= bandH) {
      const int top = std::max(bottom - bandH + 1, 0);
      const float angle = (float)(0.5 * M_PI * (inH - top - bottom) / inH);
      band(top, bottom, angle, sigmaY, kernelY);
    }
    for (int top = startTop; top < inH; top += bandH) {
      const int bottom = std::min(top + bandH - 1, inH - 1);
      const float angle = (float)(0.5 * M_PI * (top + bottom - inH) / inH);
      band(top, bottom, angle, sigmaY, kernelY);
    }
  }
};

}  // namespace

int pack_shifted_taps(const std::vector<int>& kx_q8, std::vector<uint32_t>* out) {
  const int kx = (int)kx_q8.size(), rx = kx / 2, m = (4 - rx % 4) % 4;
  const int nd = (kx + m + 3 + 3) / 4;
  if (kx <= 0 || nd > kWideMaxNd) return 0;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < kWideTapStride; i++) {
      uint32_t w = 0;
      for (int b = 0; b < 4; b++) {
        const int k = 4 * i + b - (m + j);
        if (k >= 0 && k < kx) w |= (uint32_t)kx_q8[(size_t)k] << (8 * b);
      }
      out->push_back(w);
    }
  return nd;
}

void pack_fused_taps(const std::vector<int>& kx_q8, const std::vector<int>& ky_q8, uint32_t* out) {
  // horizontal taps zero-padded to 7 and centred; variant j (output byte j of the dword) holds tap k at byte 1 + j + k of
  // the 12-byte window that starts 4 bytes left of the output dword
  int k7[7] = {0, 0, 0, 0, 0, 0, 0};
  const int n = (int)kx_q8.size(), pad = (7 - n) / 2;
  for (int k = 0; k < n; k++) k7[pad + k] = kx_q8[(size_t)k];
  uint32_t v[4][3];
  for (int j = 0; j < 4; j++)
    for (int d = 0; d < 3; d++) {
      uint32_t w = 0;
      for (int b = 0; b < 4; b++) {
        const int k = 4 * d + b - (1 + j);
        if (k >= 0 && k < 7) w |= (uint32_t)k7[k] << (8 * b);
      }
      v[j][d] = w;
    }
  // the dwords that can be non-zero: v0: 0,1  v1: 0,1,2  v2: 0,1,2  v3: 1,2
  const uint32_t packed[10] = {v[0][0], v[0][1], v[1][0], v[1][1], v[1][2], v[2][0], v[2][1], v[2][2], v[3][1], v[3][2]};
  for (int i = 0; i < 10; i++) out[i] = packed[i];
  for (int i = 0; i < 3; i++) out[10 + i] = (uint32_t)ky_q8[(size_t)i];
  for (int i = 13; i < kFusedTapDwords; i++) out[i] = 0;
}

bool build_fuse_info(const FrameTransformContext& ctx, const FilterConfig& cfg, int w, int h, FuseInfo* info,
                     std::vector<uint32_t>* packed_taps) {
  info->row_kid.assign((size_t)std::max(h, 0), -1);
  info->segs.clear();
  packed_taps->clear();
  if (ctx.input_stereo_format == STEREO_FORMAT_LR || ctx.input_stereo_format == STEREO_FORMAT_TB || w <= 0 || h <= 0) return false;
  std::vector<std::pair<std::vector<int>, std::vector<int>>> kernels;
  std::vector<int> seg_kid;
  for (const Segment& s : cfg.segments) {
    // (a segment outside the plane makes the reference print a message and skip it: nothing is fused then)
    if (s.left < 0 || s.top < 0 || s.width < 0 || s.height < 0 || s.left + s.width > w || s.top + s.height > h) return false;
    info->segs.push_back({s.left, s.top, s.width, s.height});
    bool ok = s.fixed_point && s.kx_q8.size() <= 7 && (s.kx_q8.size() & 1) == 1 && s.ky_q8.size() == 3;
    for (int v : s.kx_q8) ok = ok && v >= 0 && v <= 255;
    for (int v : s.ky_q8) ok = ok && v >= 0 && v <= 256;
    int kid = -1;
    if (ok) {
      for (size_t k = 0; k < kernels.size() && kid < 0; k++)
        if (kernels[k].first == s.kx_q8 && kernels[k].second == s.ky_q8) kid = (int)k;
      if (kid < 0 && kernels.size() < 32767) {
        kid = (int)kernels.size();
        kernels.push_back({s.kx_q8, s.ky_q8});
      }
    }
    seg_kid.push_back(kid);
  }
  // a row is fusable when the segments on it cover it exactly and share one fusable kernel
  std::vector<int64_t> covered((size_t)h, 0);
  std::vector<int> kid_of((size_t)h, -2);  // -2: no segment seen yet
  for (size_t i = 0; i < cfg.segments.size(); i++) {
    const Segment& s = cfg.segments[i];
    for (int y = s.top; y < s.top + s.height; y++) {
      covered[(size_t)y] += s.width;
      kid_of[(size_t)y] = kid_of[(size_t)y] == -2 ? seg_kid[i] : (kid_of[(size_t)y] == seg_kid[i] ? seg_kid[i] : -1);
    }
  }
  bool any = false;
  // (segments of one plane never overlap.)  A plane its segments do not cover keeps filterPlane's zeros where nothing is
  // written (:625): the partial low-pass lists of a fused plan do not reproduce that, so nothing is fused then
  for (int y = 0; y < h; y++)
    if (covered[(size_t)y] != w) return false;
  for (int y = 0; y < h; y++)
    if (kid_of[(size_t)y] >= 0) {
      info->row_kid[(size_t)y] = (int16_t)kid_of[(size_t)y];
      any = true;
    }
  packed_taps->assign(std::max<size_t>(kernels.size(), 1) * kFusedTapDwords, 0);
  for (size_t k = 0; k < kernels.size(); k++) pack_fused_taps(kernels[k].first, kernels[k].second, &(*packed_taps)[k * kFusedTapDwords]);
  return any;
}

static bool build_filter_config_or_throw(const FrameTransformContext& c, int inputWidth, int inputHeight, int outputWidth,
                                         int outputHeight, FilterConfig* cfg);

bool build_filter_config(const FrameTransformContext& c, int inputWidth, int inputHeight,
                         int outputWidth, int outputHeight, FilterConfig* cfg) {
  try {
    return build_filter_config_or_throw(c, inputWidth, inputHeight, outputWidth, outputHeight, cfg);
  } catch (const NegativeKernel&) {
    printf("Could not generate the low-pass configuration. Error: kernel of negative length\n");
    return false;
  }
}

static bool build_filter_config_or_throw(const FrameTransformContext& c, int inputWidth, int inputHeight, int outputWidth,
                                         int outputHeight, FilterConfig* cfg) {
  cfg->segments.clear();
  // one eye only; the frame path applies it to both (:373-401)
  if (c.input_stereo_format == STEREO_FORMAT_LR) inputWidth = (int)(inputWidth * 0.5);
  if (c.input_stereo_format == STEREO_FORMAT_TB) inputHeight = (int)(inputHeight * 0.5);
  if (c.output_stereo_format == STEREO_FORMAT_LR) outputWidth = (int)(outputWidth * 0.5);
  if (c.output_stereo_format == STEREO_FORMAT_TB) outputHeight = (int)(outputHeight * 0.5);

  float hFov, vFov;  // :404-446
  switch (c.output_layout) {
    case LAYOUT_CUBEMAP_32: hFov = 270.0f; vFov = 180.0f; break;
    case LAYOUT_CUBEMAP_23_OFFCENTER: hFov = 180.0f; vFov = 270.0f; break;
    case LAYOUT_FLAT_FIXED: hFov = c.fixed_hfov; vFov = c.fixed_vfov; break;
    case LAYOUT_EQUIRECT: hFov = 360.0f; vFov = 180.0f; break;
    case LAYOUT_BARREL:
    case LAYOUT_BARREL_SPLIT: hFov = 450.0f; vFov = 90.0f; break;
    case LAYOUT_EAC_32: hFov = 270.0f; vFov = 180.0f; break;
    default:
      printf("Invalid layout type.\n");
      return false;
  }

  // :448-454, all in float
  const float density = c.kernel_height_scale_factor *
      std::min(inputWidth / 360.0f, inputHeight / 180.0f) / std::max(outputWidth / hFov, outputHeight / vFov);
  const float sigmaY = 0.5f * std::min(c.max_kernel_half_height, std::max(c.min_kernel_half_height, density));
  const std::vector<float> kernelY = gaussian_taps(sigmaY);
  const int bandH = (int)std::ceil(1.0 * inputHeight / c.num_vertical_segments);  // :460
  // bandH == 0 only for an eye of height 0 (a TB frame one row high): the reference's loops then simply do not run
  // (:318-364); with any real height ceil() gives >= 1
  if (bandH <= 0 && inputHeight > 0) return false;

  Builder b{c, *cfg, inputWidth, inputHeight};
  if (c.num_vertical_segments % 2 == 0) {  // :462-473
    b.halves((int)(0.5 * inputHeight), (int)(0.5 * inputHeight - 1), sigmaY, kernelY, bandH);
  } else {  // :474-500
    const int top = (int)(0.5 * (inputHeight - bandH));
    const int bottom = top + bandH - 1;
    b.band(top, bottom, 0.0f, sigmaY, kernelY);
    b.halves(bottom + 1, top - 1, sigmaY, kernelY, bandH);
  }

  // classification + integer taps (createSeparableLinearFilter)
  for (Segment& s : cfg->segments) {
    const int smoothSym = kSmooth + kSymmetrical;
    s.fixed_point = kernel_type(s.kx) == smoothSym && kernel_type(s.ky) == smoothSym;
    s.kx_q8.resize(s.kx.size());
    s.ky_q8.resize(s.ky.size());
    // Mat::convertTo(CV_32S, 256): saturate_cast<int>(double(v) * 256) -> round half to even
    for (size_t i = 0; i < s.kx.size(); i++) s.kx_q8[i] = (int)std::lrint((double)s.kx[i] * 256.0);
    for (size_t i = 0; i < s.ky.size(); i++) s.ky_q8[i] = (int)std::lrint((double)s.ky[i] * 256.0);
  }
  return true;
}

}  // namespace t360
