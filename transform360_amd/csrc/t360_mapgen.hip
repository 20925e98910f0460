// t360_mapgen.hip -- projection kernel: one thread per output pixel evaluates the reference's
// transformPos / transformInputPos chain and writes
//   (a) the float warp map   (reference VideoFrameTransform.cpp:534-554, warpMats_[idx])
//   (b) the packed sample LUT the gather kernels read (what cv::remap's RemapInvoker derives
//       from a CV_32FC2 map on every call; here it is derived once).
//
// Bit-exactness contract: every float operation below is a single correctly
// rounded IEEE binary32 (or binary64 where the reference promotes) operation in the reference's
// evaluation order; this file is compiled with -ffp-contract=off and uses t360_libm.h for
// atan2f/asinf.  Pixel-independent transcendental values (rotation sines/cosines) arrive
// precomputed from the host in MapGenParams.
//
// All output layouts of the public header: CUBEMAP_32, CUBEMAP_23_OFFCENTER, FLAT_FIXED, EQUIRECT,
// BARREL, BARREL_SPLIT, EAC_32; inputs EQUIRECT and CUBEMAP_32; all stereo packings; rotation and
// off-centre projection.  The sinf/cosf/tan values of the EQUIRECT / BARREL* / EAC_32 outputs
// depend on the pixel's column or row only and come from host-evaluated tables
// (MapGenParams::col_tab / row_tab), so they are the host libm's values by construction.
// Attribution: the per-pixel float expressions follow, in the reference's evaluation order, transformPos /
// transformInputPos of facebook/transform360 (VideoFrameTransform.cpp:863-1316; Copyright (c) 2015-present, Facebook,
// Inc., BSD license, see that project's LICENSE file) -- bit-identical warp maps leave no freedom in the arithmetic.
// The organisation (face tables instead of switches, host-tabulated transcendentals, one lane per pixel) is this
// library's own.
#include <hip/hip_runtime.h>

#include "t360_internal.h"
#include "t360_libm.h"

#pragma clang fp contract(off)

namespace t360 {

namespace {

// cube transform parameters (reference VideoFrameTransform.cpp:38-49), indexed below
__device__ __constant__ float kP[6][3] = {
    {-0.5f, -0.5f, -0.5f},  // P0
    {0.5f, -0.5f, -0.5f},   // P1
    {0.5f, 0.5f, -0.5f},    // P3
    {-0.5f, -0.5f, 0.5f},   // P4
    {0.5f, -0.5f, 0.5f},    // P5
    {-0.5f, 0.5f, 0.5f},    // P6
};
enum { iP0 = 0, iP1, iP3, iP4, iP5, iP6 };
__device__ __constant__ float kAxis[5][3] = {
    {1.0f, 0.0f, 0.0f},   // PX
    {0.0f, 1.0f, 0.0f},   // PY
    {0.0f, 0.0f, 1.0f},   // PZ
    {-1.0f, 0.0f, 0.0f},  // NX
    {0.0f, 0.0f, -1.0f},  // NZ
};
enum { aPX = 0, aPY, aPZ, aNX, aNZ };

// face -> (p, vx, vy) for CUBEMAP_32 (:1153-1185) and CUBEMAP_23_OFFCENTER (:1120-1151)
__device__ __constant__ unsigned char kFace32[6][3] = {
    {iP5, aNZ, aPY}, {iP0, aPZ, aPY}, {iP6, aPX, aNZ}, {iP0, aPX, aPZ}, {iP4, aPX, aPY}, {iP1, aNX, aPY}};
__device__ __constant__ unsigned char kFace23[6][3] = {
    {iP4, aPY, aNZ}, {iP3, aNX, aPZ}, {iP5, aPY, aNX}, {iP1, aNX, aPY}, {iP1, aPY, aPZ}, {iP5, aNX, aNZ}};

__device__ inline float sphere_offset_dist(float x, float y, float z, float ox, float oy, float oz) {
  // reference intersectSphereOffset, VideoFrameTransform.cpp:53-75
  float loc = x * -ox + y * -oy + z * -oz;
  float odot = ox * ox + oy * oy + oz * oz;
  float root = (float)((double)(loc * loc - odot) + 1.0);
  if (root <= 0.0f) return 0.0f;
  root = t360m::sqrt_rn(root);
  if (root < loc) return 0.0f;
  return root - loc;
}

// cubemap INPUT lookup, reference transformCubeFacePos (:796-861)
__device__ inline void cube_face_pos(float e, float tx, float ty, float tz, float* outX, float* outY) {
  using t360m::div_rn;
  float x, y;
  if (tz <= -0.5f) {
    x = div_rn(tx, tz);
    y = div_rn(ty, tz);
    if (x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f) {
      *outX = div_rn(5.0f + div_rn(x, e), 6.0f);
      *outY = div_rn(3.0f + div_rn(y, e), 4.0f);
      return;
    }
  }
  if (tz >= 0.5f) {
    x = div_rn(tx, tz);
    y = div_rn(ty, tz);
    if (x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f) {
      *outX = div_rn(3.0f + div_rn(x, e), 6.0f);
      *outY = div_rn(3.0f - div_rn(y, e), 4.0f);
      return;
    }
  }
  if (tx <= -0.5f) {
    x = div_rn(tz, tx);
    y = div_rn(ty, tx);
    if (x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f) {
      *outX = div_rn(3.0f - div_rn(x, e), 6.0f);
      *outY = div_rn(1.0f + div_rn(y, e), 4.0f);
      return;
    }
  }
  if (tx >= 0.5f) {
    x = div_rn(tz, tx);
    y = div_rn(ty, tx);
    if (x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f) {
      *outX = div_rn(1.0f - div_rn(x, e), 6.0f);
      *outY = div_rn(1.0f - div_rn(y, e), 4.0f);
      return;
    }
  }
  if (ty <= -0.5f) {
    x = div_rn(tx, ty);
    y = div_rn(tz, ty);
    if (x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f) {
      *outX = div_rn(1.0f - div_rn(x, e), 6.0f);
      *outY = div_rn(3.0f + div_rn(y, e), 4.0f);
      return;
    }
  }
  if (ty >= 0.5f) {
    x = div_rn(tx, ty);
    y = div_rn(tz, ty);
    if (x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f) {
      *outX = div_rn(5.0f + div_rn(x, e), 6.0f);
      *outY = div_rn(1.0f + div_rn(y, e), 4.0f);
      return;
    }
  }
  *outX = -1.0f;
  *outY = 0.0f;
}

// reference normalize_equirectangular (:101-123)
__device__ inline void normalize_equirect(float x, float y, float* xo, float* yo) {
  if (y >= 1.0f) {
    y = 2.0f - y;
    x += 0.5f;
  } else if (y < 0.0f) {
    y = -y;
    x += 0.5f;
  }
  if (x >= 1.0f) {
    int ipart = (int)x;
    x -= (float)ipart;
  } else if (x < 0.0f) {
    int ipart = (int)(-x);
    x += (float)(ipart + 1);
  }
  *xo = x;
  *yo = y;
}

__device__ inline int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

}  // namespace

__global__ __launch_bounds__(256) void mapgen_kernel(MapGenParams P, float2* __restrict__ map,
                                                     LutEntry* __restrict__ lut) {
  using t360m::div_rn;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= P.map_w || i >= P.map_h) return;

  // generateMapForPlane (:537-538)
  float y = div_rn((float)i + 0.5f, (float)P.map_h);
  float x = div_rn((float)j + 0.5f, (float)P.map_w);

  // ---- transformPos (:893-1316) ----
  int isRight = 0;
  if (P.input_stereo != STEREO_FORMAT_MONO) {  // :903-931
    if (P.output_stereo == STEREO_FORMAT_LR) {
      if (x > 0.5f) {
        x = div_rn(x - 0.5f, 0.5f);
        isRight = 1;
      } else {
        x = div_rn(x, 0.5f);
      }
    } else if (P.output_stereo == STEREO_FORMAT_TB) {
      if (y > 0.5f) {
        y = div_rn(y - 0.5f, 0.5f);
        if (P.vflip) y = 1.0f - y;
        isRight = 1;
      } else {
        y = div_rn(y, 0.5f);
      }
    }
  }

  float outX, outY;
  bool hasMapping = true;
  if (P.output_layout == LAYOUT_FLAT_FIXED) {  // :1265-1271 (y is NOT flipped, :936)
    outX = div_rn((x - 0.5f) * P.hfov + P.yaw_deg, 360.0f) + 0.5f;
    outY = div_rn((y - 0.5f) * P.vfov - P.pitch_deg, 180.0f) + 0.5f;
    normalize_equirect(outX, outY, &outX, &outY);
  } else {
    y = 1.0f - y;  // :936-938
    int face = 0, vFace, hFace;
    const unsigned char(*ftab)[3] = kFace32;
    bool spherical = false;  // q comes from (yaw, pitch) tables instead of a cube face
    int col_idx = j;
    switch (P.output_layout) {
      case LAYOUT_CUBEMAP_32:  // :943-950
        vFace = (int)(y * 2.0f);
        hFace = (int)(x * 3.0f);
        x = x * 3.0f - (float)hFace;
        y = y * 2.0f - (float)vFace;
        face = hFace + (1 - vFace) * 3;
        break;
      case LAYOUT_CUBEMAP_23_OFFCENTER:  // :951-958
        vFace = (int)(y * 3.0f);
        hFace = (int)(x * 2.0f);
        x = x * 2.0f - (float)hFace;
        y = y * 3.0f - (float)vFace;
        face = hFace + (2 - vFace) * 2;
        ftab = kFace23;
        break;
      case LAYOUT_EQUIRECT:  // :961-964, sinf/cosf at :1090-1097
        spherical = true;
        break;
      case LAYOUT_BARREL:  // :965-977
        if (x <= 0.8f) {
          spherical = true;
        } else {
          vFace = (int)(y * 2.0f);
          face = vFace == 1 ? 2 /*TOP*/ : 3 /*BOTTOM*/;
          x = x * 5.0f - 4.0f;
          y = y * 2.0f - (float)vFace;
        }
        break;
      case LAYOUT_BARREL_SPLIT:  // :978-1015
        if (3.0f * x <= 2.0f) {
          vFace = (int)(y * 2.0f);
          spherical = true;
          col_idx = (vFace != 0 ? P.map_w : 0) + j;  // yaw depends on the half (vFace in {0, 1})
        } else {
          const int halfVFace = (int)(y * 4.0f);
          face = (halfVFace == 1 || halfVFace == 3) ? 2 /*TOP*/ : 3 /*BOTTOM*/;
          x = x * 3.0f - 2.0f;
          switch (halfVFace) {
            case 0:
              y = y * 2.0f;
              x = 1.0f - x;
              y = (0.5f - y) * P.expand_coef;
              break;
            case 1:
              y = y * 2.0f;
              x = 1.0f - x;
              y = 1.0f - P.expand_coef * (y - 0.5f);
              break;
            case 2:
              y = y * 2.0f - 0.5f;
              y = 1.0f - P.expand_coef * (1.0f - y);
              break;
            case 3:
              y = y * 2.0f - 1.5f;
              y = y * P.expand_coef;
              break;
            default:
              break;
          }
        }
        break;
      default:  // LAYOUT_EAC_32 :1016-1027: the tan() warp of both face coordinates is tabulated
        vFace = (int)(y * 2.0f);
        hFace = (int)(x * 3.0f);
        x = P.col_tab[j];
        y = P.row_tab[i];
        face = hFace + (1 - vFace) * 3;
        break;
    }
    // A face outside the enum DOES occur: x == 1.0f exactly for the centre column of an LR output of odd (scaled)
    // width, in the band where hFace + 3 = 6.  The reference then computes with uninitialised vectors (:939, its face
    // switches have no default): whatever the build left on the stack -- there is nothing to match.  The oracle and
    // this kernel use (P0, PX, PY) for such a pixel.
    const bool face_ok = face >= 0 && face <= 5;
    face = face < 0 ? 0 : (face > 5 ? 5 : face);

    float qx, qy, qz;
    if (spherical) {  // :1090-1100
      const float sin_yaw = P.col_tab[2 * col_idx], cos_yaw = P.col_tab[2 * col_idx + 1];
      const float sin_pitch = P.row_tab[2 * i], cos_pitch = P.row_tab[2 * i + 1];
      qx = sin_yaw * cos_pitch;
      qy = sin_pitch;
      qz = cos_yaw * cos_pitch;
    } else {
      if (P.output_layout == LAYOUT_BARREL || P.output_layout == LAYOUT_BARREL_SPLIT) {  // :1102-1112
        const float radius = (x - 0.5f) * (x - 0.5f) + (y - 0.5f) * (y - 0.5f);
        if (radius > 0.25f * P.expand_coef * P.expand_coef) hasMapping = false;
      }
      x = (x - 0.5f) * P.expand_coef + 0.5f;  // :1115-1116
      y = (y - 0.5f) * P.expand_coef + 0.5f;

      const float* p = kP[face_ok ? ftab[face][0] : iP0];
      const float* vx = kAxis[face_ok ? ftab[face][1] : aPX];
      const float* vy = kAxis[face_ok ? ftab[face][2] : aPY];
      qx = p[0] + vx[0] * x + vy[0] * y;  // :1187-1189
      qy = p[1] + vx[1] * x + vy[1] * y;
      qz = p[2] + vx[2] * x + vy[2] * y;
    }

    if (P.offcenter) {  // :1192-1230
      float d = t360m::sqrt_rn(qx * qx + qy * qy + qz * qz);
      qx = div_rn(qx, d);
      qy = div_rn(qy, d);
      qz = div_rn(qz, d);
      if (P.horizontal_offset) {
        d = t360m::sqrt_rn(qx * qx + qz * qz);
        qx = div_rn(qx, d);
        qy = div_rn(qy, d);
        qz = div_rn(qz, d);
        float dist = sphere_offset_dist(qx, 0.0f, qz, P.off_x, 0.0f, P.off_z);
        if (dist > 0.0f) {
          qx = qx * dist - P.off_x;
          qz = qz * dist - P.off_z;
        }
      } else {
        float dist = sphere_offset_dist(qx, qy, qz, P.off_x, P.off_y, P.off_z);
        if (dist > 0.0f) {
          qx = qx * dist - P.off_x;
          qy = qy * dist - P.off_y;
          qz = qz * dist - P.off_z;
        }
      }
    }

    // rotation (:1240-1246); rot[] holds the parenthesised float coefficients
    float tx = qx * P.rot[0] - qy * P.rot[1] + qz * P.rot[2];
    float ty = qx * P.rot[3] - qy * P.rot[4] + qz * P.rot[5];
    float tz = qx * P.rot[6] - qy * P.rot[7] + qz * P.rot[8];
    ty = -ty;

    // transformInputPos (:863-891)
    float d = t360m::sqrt_rn(tx * tx + ty * ty + tz * tz);
    if (P.input_layout == LAYOUT_CUBEMAP_32) {
      cube_face_pos(P.input_expand_coef, div_rn(tx, d), div_rn(ty, d), div_rn(tz, d), &outX, &outY);
    } else {
      // -atan2f(-tx/d, tz/d) / (M_PI * 2.0f) + 0.5f : division and sum in double (:880)
      float a = t360m::atan2_f32(div_rn(-tx, d), div_rn(tz, d));
      outX = (float)((double)(-a) / 6.283185307179586476925286766559 + 0.5);
      if (P.output_layout == LAYOUT_BARREL || P.output_layout == LAYOUT_BARREL_SPLIT) {  // :881-886
        const float hi = 1.0f - P.input_pixel_width * 0.5f;
        const float lo = P.input_pixel_width * 0.5f;
        outX = (hi < outX) ? hi : outX;  // std::min
        outX = (outX < lo) ? lo : outX;  // std::max
      }
      float s = t360m::asin_f32(div_rn(-ty, d));
      outY = (float)((double)s / 3.14159265358979323846 + 0.5);  // :887
    }
  }

  // stereo re-pack (:1278-1300); pixels without a mapping (outside the barrel caps) get (-1, 0)
  if (!hasMapping) {
    outX = -1.0f;
    outY = 0.0f;
  } else if (P.input_stereo == STEREO_FORMAT_TB) {
    outY = isRight ? (outY * 0.5f + 0.5f) : (outY * 0.5f);
  } else if (P.input_stereo == STEREO_FORMAT_LR) {
    outX = isRight ? (outX * 0.5f + 0.5f) : (outX * 0.5f);
  }

  // generateMapForPlane (:544-545)
  const float mx = outX * (float)P.in_w - 0.5f;
  const float my = outY * (float)P.in_h - 0.5f;
  const size_t o = (size_t)i * P.map_w + j;
  map[o] = make_float2(mx, my);

  // cv::remap RemapInvoker quantisation
  LutEntry e;
  if (P.interp == NEAREST) {
    e.ix = (int16_t)sat_s16(__float2int_rn(mx));
    e.iy = (int16_t)sat_s16(__float2int_rn(my));
    e.frac = 0;
  } else {
    const int sx = __float2int_rn(mx * 32.0f);
    const int sy = __float2int_rn(my * 32.0f);
    e.ix = (int16_t)sat_s16(sx >> kInterBits);
    e.iy = (int16_t)sat_s16(sy >> kInterBits);
    e.frac = (uint16_t)((sy & 31) * 32 + (sx & 31));
  }
  e.pad = 0;
  lut[o] = e;
}

// host-callable launcher (C++ linkage inside the library)
hipError_t launch_mapgen(const MapGenParams& P, float2* map, LutEntry* lut, hipStream_t stream) {
  dim3 block(256, 1, 1);
  dim3 grid((P.map_w + 255) / 256, P.map_h, 1);
  hipLaunchKernelGGL(mapgen_kernel, grid, block, 0, stream, P, map, lut);
  return hipGetLastError();
}

// ---- synthetic noise generator (bench / tests): byte i = splitmix64(seed + i) >> 56 ----
__device__ __host__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void fill_noise_kernel(uint8_t* __restrict__ dst, int64_t n, uint64_t seed) {
  // 16 bytes per thread per step, grid-stride
  const int64_t nvec = n >> 4;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
       v += (int64_t)gridDim.x * blockDim.x) {
    uint32_t w[4];
    for (int k = 0; k < 4; k++) {
      uint32_t acc = 0;
      for (int b = 0; b < 4; b++) {
        const uint64_t idx = (uint64_t)(v * 16 + k * 4 + b);
        acc |= (uint32_t)(splitmix64(seed + idx) >> 56) << (8 * b);
      }
      w[k] = acc;
    }
    reinterpret_cast<uint4*>(dst)[v] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // tail
  const int64_t tail0 = nvec << 4;
  const int64_t t = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = (uint8_t)(splitmix64(seed + (uint64_t)t) >> 56);
}

hipError_t launch_fill_noise(uint8_t* dst, int64_t n, uint64_t seed, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) != 0) return hipErrorInvalidValue;  // uint4 stores
  int blocks = (int)((((n >> 4) + 255) / 256) < 4096 ? (((n >> 4) + 255) / 256) : 4096);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_noise_kernel, dim3(blocks), dim3(256), 0, stream, dst, n, seed);
  return hipGetLastError();
}

}  // namespace t360
