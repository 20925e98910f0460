// t360_libm.h -- bit-reproducible single-precision atan / atan2 / asin for the projection kernel.
//
// Why this exists: the reference computes its warp map on the host with glibc's atan2f / asinf
// (reference VideoFrameTransform.cpp:880, 887).  Nearest-neighbour picks and the 1/32-pixel
// interpolation phase depend on the LAST BIT of those results, so the GPU map
// generator cannot use the ROCm device math library (different polynomials, different
// rounding).  The functions below evaluate the classic Sun fdlibm single-precision algorithms
// -- the ones glibc 2.35 ships for these three entry points on x86-64 (no FMA/ifunc variants
// exist for them) -- with every operation a correctly rounded IEEE-754 binary32 operation in
// the published order.  Compiled with -ffp-contract=off for both host (gcc, used by the CPU
// unit test that compares against the running libm over the full float range) and device
// (hipcc; fp32 add/mul/div/sqrt are correctly rounded on gfx950, denormals preserved).
//
// Constants are given by their IEEE bit patterns.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define T360_HD __host__ __device__ inline
#else
#define T360_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace t360m {

T360_HD float bits2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}
T360_HD uint32_t f2bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
#endif
}
T360_HD float fabs_(float x) { return bits2f(f2bits(x) & 0x7fffffffu); }

// correctly rounded square root / division, spelled so the device compiler cannot pick an
// approximate expansion
// NOTE: HIP's __fsqrt_rn() maps to the approximate native sqrt unless OCML_BASIC_ROUNDED_OPERATIONS
// is defined (measured: 4 % of map coordinates off by one ulp).  The plain builtin / operator are
// lowered by hipcc to the IEEE sequences (v_sqrt_f32 + fma correction, v_div_scale/fmas/fixup)
// because -fhip-fp32-correctly-rounded-divide-sqrt is the default.
T360_HD float sqrt_rn(float x) { return __builtin_sqrtf(x); }
T360_HD float div_rn(float a, float b) { return a / b; }

// ---- atanf: argument reduction to |x| < 7/16 around 0.5, 1, 1.5, inf; odd/even split polynomial
T360_HD float atan_f32(float x) {
  const float atanhi0 = bits2f(0x3eed6338u), atanhi1 = bits2f(0x3f490fdau),
              atanhi2 = bits2f(0x3f7b985eu), atanhi3 = bits2f(0x3fc90fdau);
  const float atanlo0 = bits2f(0x31ac3769u), atanlo1 = bits2f(0x33222168u),
              atanlo2 = bits2f(0x33140fb4u), atanlo3 = bits2f(0x33a22168u);
  const float aT0 = bits2f(0x3eaaaaabu), aT1 = bits2f(0xbe4ccccdu), aT2 = bits2f(0x3e124925u),
              aT3 = bits2f(0xbde38e38u), aT4 = bits2f(0x3dba2e6eu), aT5 = bits2f(0xbd9d8795u),
              aT6 = bits2f(0x3d886b35u), aT7 = bits2f(0xbd6ef16bu), aT8 = bits2f(0x3d4bda59u),
              aT9 = bits2f(0xbd15a221u), aT10 = bits2f(0x3c8569d7u);
  const float one = 1.0f;
  const int32_t hx = (int32_t)f2bits(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) { /* |x| >= 2^25 */
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi3 + atanlo3 : -atanhi3 - atanlo3;
  }
  if (ix < 0x3ee00000) {                 /* |x| < 0.4375 */
    if (ix < 0x31000000) return x;       /* |x| < 2^-29 */
    id = -1;
  } else {
    x = fabs_(x);
    if (ix < 0x3f980000) {   /* |x| < 1.1875 */
      if (ix < 0x3f300000) { /* 7/16 <= |x| < 11/16 */
        id = 0;
        x = div_rn(2.0f * x - one, 2.0f + x);
      } else { /* 11/16 <= |x| < 19/16 */
        id = 1;
        x = div_rn(x - one, x + one);
      }
    } else {
      if (ix < 0x401c0000) { /* |x| < 2.4375 */
        id = 2;
        x = div_rn(x - 1.5f, one + 1.5f * x);
      } else {
        id = 3;
        x = div_rn(-1.0f, x);
      }
    }
  }
  float z = x * x;
  float w = z * z;
  float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  float hi = id == 0 ? atanhi0 : id == 1 ? atanhi1 : id == 2 ? atanhi2 : atanhi3;
  float lo = id == 0 ? atanlo0 : id == 1 ? atanlo1 : id == 2 ? atanlo2 : atanlo3;
  z = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -z : z;
}

// ---- atan2f(y, x)
T360_HD float atan2_f32(float y, float x) {
  const float tiny = 1.0e-30f;
  const float pi_o_4 = bits2f(0x3f490fdbu), pi_o_2 = bits2f(0x3fc90fdbu), pi = bits2f(0x40490fdbu),
              pi_lo = bits2f(0xb3bbbd2eu);
  const int32_t hx = (int32_t)f2bits(x), hy = (int32_t)f2bits(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y; /* NaN */
  if (hx == 0x3f800000) return atan_f32(y);             /* x == 1 */
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);    /* 2*sign(x) + sign(y) */
  if (iy == 0) {
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo; /* |y/x| > 2^60 */
  else if (hx < 0 && k < -60)
    z = 0.0f; /* |y|/x < -2^60 */
  else
    z = atan_f32(fabs_(div_rn(y, x)));
  switch (m) {
    case 0: return z;
    case 1: return bits2f(f2bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// ---- asinf: glibc's float asin (Moshier's degree-4 polynomial variant)
T360_HD float asin_f32(float x) {
  const float one = 1.0f;
  const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f,
              pio4_hi = 0.785398185253143310546875f;
  const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f,
              p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
  const int32_t hx = (int32_t)f2bits(x);
  const int32_t ix = hx & 0x7fffffff;
  float t, w, p, q, c, r, s;
  if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo; /* |x| == 1 */
  if (ix > 0x3f800000) return (x - x) / (x - x);          /* NaN */
  if (ix < 0x3f000000) {                                  /* |x| < 0.5 */
    if (ix < 0x32000000) return x;                        /* |x| < 2^-27 */
    t = x * x;
    w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    return x + x * w;
  }
  w = one - fabs_(x);
  t = w * 0.5f;
  p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
  s = sqrt_rn(t);
  if (ix >= 0x3F79999A) { /* |x| > 0.975 */
    t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
  } else {
    w = bits2f(f2bits(s) & 0xfffff000u);
    c = div_rn(t - w * w, s + w);
    r = p;
    p = 2.0f * s * r - (pio2_lo - 2.0f * c);
    q = pio4_hi - 2.0f * w;
    t = pio4_hi - (p - q);
  }
  return hx > 0 ? t : -t;
}

}  // namespace t360m
