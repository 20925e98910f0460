/*
 * oracle/t360_oracle_map.c -- CPU ORACLE (test infrastructure only; see t360_oracle.h)
 *
 * Restates, operation for operation, the INIT-TIME math of the reference:
 *   projection   VideoFrameTransform.cpp:53-75, 101-123, 796-1316, 524-554
 *   low-pass cfg VideoFrameTransform.cpp:78-94, 126-170, 210-501
 *
 * The reference is C++ with `using namespace std`: a call such as sin(yaw) with a float
 * argument resolves to the float overload (sinf) while sin(fixed_yaw * M_PI / 180.0f)
 * has a double argument.  Every libm call below spells out the overload the reference
 * gets, and every mixed float/double expression keeps the reference's promotions, because
 * nearest/bicubic parity depends on the last bit of these coordinates.
 * Build WITHOUT -ffast-math / -march flags that enable FMA contraction (oracle/Makefile).
 *
 * Pinned against: the reference's own code compiled from /root/reference (oracle/_ref)
 * and the golden hashes in tests/golden/maps.json.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "t360_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* VideoFrameTransform.cpp:30-35 */
static const float kCubemapSideDistance = 0.5f;
static const float kXHalf = 0.5f;
static const float kYHalf = 0.5f;
static const double kEpsilon = 1e-9;
#define K_SPHERE_AREA (4 * M_PI)
#define K_FOV (0.5333 * M_PI)

/* VideoFrameTransform.cpp:38-49: face origins and in-face axes */
static const float P0[3] = {-0.5f, -0.5f, -0.5f};
static const float P1[3] = {0.5f, -0.5f, -0.5f};
static const float P3[3] = {0.5f, 0.5f, -0.5f};
static const float P4[3] = {-0.5f, -0.5f, 0.5f};
static const float P5[3] = {0.5f, -0.5f, 0.5f};
static const float P6[3] = {-0.5f, 0.5f, 0.5f};
static const float PX[3] = {1.0f, 0.0f, 0.0f};
static const float PY[3] = {0.0f, 1.0f, 0.0f};
static const float PZ[3] = {0.0f, 0.0f, 1.0f};
static const float NX[3] = {-1.0f, 0.0f, 0.0f};
static const float NZ[3] = {0.0f, 0.0f, -1.0f};

/* VideoFrameTransform.cpp:53-75 */
static float intersect_sphere_offset(float x, float y, float z, float ox, float oy, float oz) {
  float loc = x * -ox + y * -oy + z * -oz;
  float odot = ox * ox + oy * oy + oz * oz;
  /* "loc * loc - odot + 1.0": the float difference is promoted, 1.0 added in double,
   * and the sum rounded back to float on assignment */
  float root = (float)((double)(loc * loc - odot) + 1.0);
  if (root <= 0.0f) return 0.0f;
  root = sqrtf(root);
  if (root < loc) return 0.0f;
  return root - loc;
}

/* VideoFrameTransform.cpp:101-123 */
static void normalize_equirect(float x, float y, float* xout, float* yout) {
  if (y >= 1.0f) {
    y = 2.0f - y;
    x += 0.5f;
  } else if (y < 0.0f) {
    y = -y;
    x += 0.5f;
  }
  if (x >= 1.0f) {
    int ipart = (int)x;
    x -= ipart;
  } else if (x < 0.0f) {
    int ipart = (int)(-x);
    x += (ipart + 1);
  }
  *xout = x;
  *yout = y;
}

/* VideoFrameTransform.cpp:796-861 (cubemap INPUT) */
static void cube_face_pos(const FrameTransformContext* c, float tx, float ty, float tz,
                          float* outX, float* outY) {
  float x, y;
  const float e = c->input_expand_coef;
  if (tz <= -kCubemapSideDistance) {
    x = tx / tz;
    y = ty / tz;
    if (x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0) {
      *outX = (5.0f + x / e) / 6.0f;
      *outY = (3.0f + y / e) / 4.0f;
      return;
    }
  }
  if (tz >= kCubemapSideDistance) {
    x = tx / tz;
    y = ty / tz;
    if (x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0) {
      *outX = (3.0f + x / e) / 6.0f;
      *outY = (3.0f - y / e) / 4.0f;
      return;
    }
  }
  if (tx <= -kCubemapSideDistance) {
    x = tz / tx;
    y = ty / tx;
    if (x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0) {
      *outX = (3.0f - x / e) / 6.0f;
      *outY = (1.0f + y / e) / 4.0f;
      return;
    }
  }
  if (tx >= kCubemapSideDistance) {
    x = tz / tx;
    y = ty / tx;
    if (x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0) {
      *outX = (1.0f - x / e) / 6.0f;
      *outY = (1.0f - y / e) / 4.0f;
      return;
    }
  }
  if (ty <= -kCubemapSideDistance) {
    x = tx / ty;
    y = tz / ty;
    if (x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0) {
      *outX = (1.0f - x / e) / 6.0f;
      *outY = (3.0f + y / e) / 4.0f;
      return;
    }
  }
  if (ty >= kCubemapSideDistance) {
    x = tx / ty;
    y = tz / ty;
    if (x >= -1.0 && x <= 1.0 && y >= -1.0 && y <= 1.0) {
      *outX = (5.0f + x / e) / 6.0f;
      *outY = (1.0f + y / e) / 4.0f;
      return;
    }
  }
  *outX = -1.0f;
  *outY = 0.0f;
}

/* VideoFrameTransform.cpp:863-891 */
static void input_pos(const FrameTransformContext* c, float tx, float ty, float tz,
                      float inputPixelWidth, float* outX, float* outY) {
  if (c->input_layout == LAYOUT_CUBEMAP_32) {
    float d = sqrtf(tx * tx + ty * ty + tz * tz);
    cube_face_pos(c, tx / d, ty / d, tz / d, outX, outY);
    return;
  }
  /* anything else is read as equirect */
  float d = sqrtf(tx * tx + ty * ty + tz * tz);
  /* -atan2f(..) / (M_PI * 2.0f) + 0.5f : division and addition in double (:880) */
  *outX = (float)((double)(-atan2f(-tx / d, tz / d)) / (M_PI * 2.0f) + 0.5f);
  if (c->output_layout == LAYOUT_BARREL || c->output_layout == LAYOUT_BARREL_SPLIT) {
    float hi = 1.0f - inputPixelWidth * 0.5f;
    float lo = inputPixelWidth * 0.5f;
    *outX = (hi < *outX) ? hi : *outX; /* std::min(a,b) = (b < a) ? b : a */
    *outX = (*outX < lo) ? lo : *outX; /* std::max(a,b) = (a < b) ? b : a */
  }
  *outY = (float)((double)asinf(-ty / d) / M_PI + 0.5f); /* :887 */
}

/* VideoFrameTransform.cpp:893-1316 */
int t360o_transform_pos(const FrameTransformContext* c, float x, float y, float* outX,
                        float* outY, float inputPixelWidth) {
  int isRight = 0;

  if (c->input_stereo_format != STEREO_FORMAT_MONO) { /* :903-931 */
    switch (c->output_stereo_format) {
      case STEREO_FORMAT_LR:
        if (x > kXHalf) {
          x = (x - kXHalf) / kXHalf;
          isRight = 1;
        } else {
          x = x / kXHalf;
        }
        break;
      case STEREO_FORMAT_TB:
        if (y > kYHalf) {
          y = (y - kYHalf) / kYHalf;
          if (c->vflip) y = 1.0f - y;
          isRight = 1;
        } else {
          y = y / kYHalf;
        }
        break;
      default:
        break;
    }
  }

  float qx = 0, qy = 0, qz = 0, tx, ty, tz, d;
  float yaw = 0, pitch = 0;
  int hasMapping = 1;
  if (c->output_layout != LAYOUT_FLAT_FIXED) y = 1.0f - y; /* :936-938 */
  /* The reference declares p, vx, vy without initialisers (:939) and its face switches have no default (:1120-1185):
   * for a face value outside the enum -- x == 1.0f exactly: the centre column of an LR output of odd scaled width, in
   * the band where hFace + 3 = 6 -- it computes with whatever the stack holds; oracle/_ref built here produces
   * values that match no rule (checked).  There is nothing to restate: this oracle uses (P0, PX, PY), the HIP map
   * generator does the same, and tests/test_oracle_fuzz.py excludes those entries from the comparison with _ref. */
  const float *vx = PX, *vy = PY, *p = P0;
  int face = 0, vFace, hFace;

  switch (c->output_layout) { /* :942-1083 */
    case LAYOUT_CUBEMAP_32:
      vFace = (int)(y * 2);
      hFace = (int)(x * 3);
      x = x * 3.0f - hFace;
      y = y * 2.0f - vFace;
      face = hFace + (1 - vFace) * 3;
      break;
    case LAYOUT_CUBEMAP_23_OFFCENTER:
      vFace = (int)(y * 3);
      hFace = (int)(x * 2);
      x = x * 2.0f - hFace;
      y = y * 3.0f - vFace;
      face = hFace + (2 - vFace) * 2;
      break;
    case LAYOUT_FLAT_FIXED:
      break;
    case LAYOUT_EQUIRECT:
      yaw = (float)((2.0f * x - 1.0f) * M_PI);
      pitch = (float)((y - 0.5f) * M_PI);
      break;
    case LAYOUT_BARREL:
      if (x <= 0.8f) {
        yaw = (float)((2.5f * x - 1.0f) * c->expand_coef * M_PI);
        pitch = (float)((y * 0.5f - 0.25f) * c->expand_coef * M_PI);
        face = -1;
      } else {
        vFace = (int)(y * 2);
        face = (vFace == 1) ? TOP : BOTTOM;
        x = x * 5.0f - 4.0f;
        y = y * 2.0f - vFace;
      }
      break;
    case LAYOUT_BARREL_SPLIT:
      if (3.0f * x <= 2.0f) {
        vFace = (int)(y * 2);
        yaw = (float)(((3.0f / 2.0f * x - 0.5f) * c->expand_coef - vFace + 1.0f) * M_PI);
        pitch = (float)((y - 0.25f - 0.5f * vFace) * c->expand_coef * M_PI);
        face = -1;
      } else {
        int halfVFace = (int)(y * 4);
        face = (halfVFace == 1 || halfVFace == 3) ? TOP : BOTTOM;
        x = x * 3.0f - 2.0f;
        switch (halfVFace) {
          case 0:
            y = y * 2.0f;
            x = 1.0f - x;
            y = (0.5f - y) * c->expand_coef;
            break;
          case 1:
            y = y * 2.0f;
            x = 1.0f - x;
            y = 1.0f - c->expand_coef * (y - 0.5f);
            break;
          case 2:
            y = y * 2.0f - 0.5f;
            y = 1.0f - c->expand_coef * (1.0f - y);
            break;
          case 3:
            y = y * 2.0f - 1.5f;
            y = y * c->expand_coef;
            break;
        }
      }
      break;
    case LAYOUT_EAC_32:
      vFace = (int)(y * 2);
      hFace = (int)(x * 3);
      x = x * 3.0f - hFace;
      y = y * 2.0f - vFace;
      /* tan() of a double argument; result * 0.5f + 0.5f stays double until the store */
      x = (float)(tan((x - 0.5f) * M_PI * 0.5f) * 0.5f + 0.5f);
      y = (float)(tan((y - 0.5f) * M_PI * 0.5f) * 0.5f + 0.5f);
      face = hFace + (1 - vFace) * 3;
      break;
    default: /* LAYOUT_N and anything unknown */
      printf("Invalid layout type.\n");
      return 0;
  }

  switch (c->output_layout) { /* :1085-1276 */
    case LAYOUT_CUBEMAP_32:
    case LAYOUT_CUBEMAP_23_OFFCENTER:
    case LAYOUT_EQUIRECT:
    case LAYOUT_BARREL:
    case LAYOUT_BARREL_SPLIT:
    case LAYOUT_EAC_32: {
      if (c->output_layout == LAYOUT_EQUIRECT ||
          (c->output_layout == LAYOUT_BARREL && face < 0) ||
          (c->output_layout == LAYOUT_BARREL_SPLIT && face < 0)) {
        /* float arguments -> std::sin(float) etc. */
        float sin_yaw = sinf(yaw);
        float sin_pitch = sinf(pitch);
        float cos_yaw = cosf(yaw);
        float cos_pitch = cosf(pitch);
        qx = sin_yaw * cos_pitch;
        qy = sin_pitch;
        qz = cos_yaw * cos_pitch;
      } else {
        if (c->output_layout == LAYOUT_BARREL || c->output_layout == LAYOUT_BARREL_SPLIT) {
          float radius = (x - 0.5f) * (x - 0.5f) + (y - 0.5f) * (y - 0.5f);
          if (radius > 0.25f * c->expand_coef * c->expand_coef) {
            hasMapping = 0;
            break;
          }
        }
        x = (x - 0.5f) * c->expand_coef + 0.5f;
        y = (y - 0.5f) * c->expand_coef + 0.5f;

        if (c->output_layout == LAYOUT_CUBEMAP_23_OFFCENTER) { /* :1119-1151 */
          switch (face) {
            case RIGHT: p = P4; vx = PY; vy = NZ; break;
            case LEFT: p = P3; vx = NX; vy = PZ; break;
            case TOP: p = P5; vx = PY; vy = NX; break;
            case BOTTOM: p = P1; vx = NX; vy = PY; break;
            case FRONT: p = P1; vx = PY; vy = PZ; break;
            case BACK: p = P5; vx = NX; vy = NZ; break;
          }
        } else { /* :1152-1185 */
          switch (face) {
            case RIGHT: p = P5; vx = NZ; vy = PY; break;
            case LEFT: p = P0; vx = PZ; vy = PY; break;
            case TOP: p = P6; vx = PX; vy = NZ; break;
            case BOTTOM: p = P0; vx = PX; vy = PZ; break;
            case FRONT: p = P4; vx = PX; vy = PY; break;
            case BACK: p = P1; vx = NX; vy = PY; break;
          }
        }
        qx = p[0] + vx[0] * x + vy[0] * y;
        qy = p[1] + vx[1] * x + vy[1] * y;
        qz = p[2] + vx[2] * x + vy[2] * y;
      }

      /* :1192-1230 off-centre projection */
      if (fabsf(c->fixed_cube_offcenter_x) > kEpsilon ||
          fabsf(c->fixed_cube_offcenter_y) > kEpsilon ||
          fabsf(c->fixed_cube_offcenter_z) > kEpsilon) {
        float dist;
        d = sqrtf(qx * qx + qy * qy + qz * qz);
        qx = qx / d;
        qy = qy / d;
        qz = qz / d;
        if (c->is_horizontal_offset) {
          d = sqrtf(qx * qx + qz * qz);
          qx = qx / d;
          qy = qy / d;
          qz = qz / d;
          dist = intersect_sphere_offset(qx, 0, qz, c->fixed_cube_offcenter_x, 0,
                                         c->fixed_cube_offcenter_z);
          if (dist > 0.0f) {
            qx = qx * dist - c->fixed_cube_offcenter_x;
            qz = qz * dist - c->fixed_cube_offcenter_z;
          }
        } else {
          dist = intersect_sphere_offset(qx, qy, qz, c->fixed_cube_offcenter_x,
                                         c->fixed_cube_offcenter_y, c->fixed_cube_offcenter_z);
          if (dist > 0.0f) {
            qx = qx * dist - c->fixed_cube_offcenter_x;
            qy = qy * dist - c->fixed_cube_offcenter_y;
            qz = qz * dist - c->fixed_cube_offcenter_z;
          }
        }
      }

      /* :1233-1238 rotation: double sin/cos of (deg * M_PI / 180.0f), narrowed to float */
      float s1 = (float)sin(c->fixed_yaw * M_PI / 180.0f);
      float s2 = (float)sin(c->fixed_pitch * M_PI / 180.0f);
      float s3 = (float)sin(c->fixed_roll * M_PI / 180.0f);
      float c1 = (float)cos(c->fixed_yaw * M_PI / 180.0f);
      float c2 = (float)cos(c->fixed_pitch * M_PI / 180.0f);
      float c3 = (float)cos(c->fixed_roll * M_PI / 180.0f);

      tx = qx * (c1 * c3 + s1 * s2 * s3) - qy * (c3 * s1 * s2 - c1 * s3) + qz * (c2 * s1);
      ty = qx * (c2 * s3) - qy * (c2 * c3) + qz * (-s2);
      tz = qx * (c1 * s2 * s3 - c3 * s1) - qy * (c1 * c3 * s2 + s1 * s3) + qz * (c1 * c2);
      ty = -ty;

      input_pos(c, tx, ty, tz, inputPixelWidth, outX, outY);
      break;
    }
    case LAYOUT_FLAT_FIXED: /* :1265-1271 */
      *outX = ((x - 0.5f) * c->fixed_hfov + c->fixed_yaw) / 360.0f + 0.5f;
      *outY = ((y - 0.5f) * c->fixed_vfov - c->fixed_pitch) / 180.0f + 0.5f;
      normalize_equirect(*outX, *outY, outX, outY);
      break;
    default:
      printf("Invalid layout type.\n");
      return 0;
  }

  if (hasMapping) { /* :1278-1300 */
    switch (c->input_stereo_format) {
      case STEREO_FORMAT_TB:
        *outY = isRight ? (*outY * kYHalf + kYHalf) : (*outY * kYHalf);
        break;
      case STEREO_FORMAT_LR:
        *outX = isRight ? (*outX * kXHalf + kXHalf) : (*outX * kXHalf);
        break;
      default:
        break;
    }
  } else {
    *outX = -1;
    *outY = 0;
  }
  return 1;
}

/* VideoFrameTransform.cpp:524-526 */
int t360o_scaled_size(const FrameTransformContext* c, int outW, int outH, int* scaledW,
                      int* scaledH) {
  *scaledW = (int)(c->width_scale_factor * outW + 0.5);
  *scaledH = (int)(c->height_scale_factor * outH + 0.5);
  return *scaledW > 0 && *scaledH > 0;
}

/* VideoFrameTransform.cpp:524-554 */
int t360o_generate_map(const FrameTransformContext* c, int inW, int inH, int outW, int outH,
                       float* map) {
  int sw, sh;
  if (!t360o_scaled_size(c, outW, outH, &sw, &sh)) return 0;
  float inputPixelWidth = 1.0f / inW;
  if (c->input_stereo_format == STEREO_FORMAT_LR) inputPixelWidth *= 2;
  for (int i = 0; i < sh; ++i) {
    for (int j = 0; j < sw; ++j) {
      float outX, outY;
      float y = (i + 0.5f) / sh;
      float x = (j + 0.5f) / sw;
      if (!t360o_transform_pos(c, x, y, &outX, &outY, inputPixelWidth)) {
        printf("Failed to find the mapping coordinate for point (%d, %d)\n", i, j);
        return 0;
      }
      map[((size_t)i * sw + j) * 2 + 0] = outX * inW - 0.5f;
      map[((size_t)i * sw + j) * 2 + 1] = outY * inH - 0.5f;
    }
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * low-pass configuration
 * ------------------------------------------------------------------------------------------ */

/* VideoFrameTransform.cpp:78-94.  `kernel /= sum` is cv::Mat::operator/=(double), which OpenCV
 * implements as convertTo(kernel, -1, 1./sum): a float multiply by (float)(1.0/sum)
 * [OpenCV mat.inl.hpp + convert_scale 32f->32f, restated from memory; see DESIGN.md]. */
/* A negative sigma (the band's angle beyond 90 degrees: planes a few rows high cut into more bands than that) makes the
 * reference ask for a Mat of negative width: cv::Mat::zeros throws, generateMapForPlane catches and returns false
 * (:78-81, :571-576).  Recorded here, reported by t360o_filter_config(). */
static __thread int t_kernel_error;
float* t360o_calculate_kernel(float sigma, int* len) {
  int boxHalfLength = (int)(sigma * 2);
  if (boxHalfLength < 0) {
    t_kernel_error = 1;
    boxHalfLength = 0;
  }
  int n = boxHalfLength * 2 + 1;
  float* k = (float*)calloc((size_t)n, sizeof(float));
  float sum = 0;
  float sigmaComponent = fabsf(sigma) < kEpsilon ? 0 : (float)(0.5 / (sigma * sigma));
  for (int u = -boxHalfLength; u <= boxHalfLength; ++u) {
    float value = expf(-(u * u * sigmaComponent));
    k[u + boxHalfLength] = value;
    sum += value;
  }
  float scale = (float)(1.0 / (double)sum);
  for (int i = 0; i < n; ++i) k[i] = k[i] * scale;
  *len = n;
  return k;
}

/* :126-130 */
static double angular_distance(double yaw1, double pitch1, double yaw2, double pitch2) {
  return acos(sin(pitch1) * sin(pitch2) + cos(pitch1) * cos(pitch2) * cos(yaw1 - yaw2));
}
/* :132-134 */
static double sampling_arc(double offset, double renderArc) {
  return M_PI - 2 * atan2(cos(0.5 * renderArc) - offset, sin(0.5 * renderArc));
}
/* :136-138 */
static double spherical_area(double angle) { return (1 - cos(0.5 * angle)) * 2 * M_PI; }

/* :140-166 */
static double effective_ratio_fov(double angularDist, double offset, double fov) {
  double majorAxisScaling;
  if (angularDist - kEpsilon > fov / 2) {
    if (angularDist + fov / 2 > M_PI) {
      double edge1 = sampling_arc(offset, (2 * M_PI - angularDist - fov / 2) * 2) / 2;
      double edge2 = sampling_arc(offset, (angularDist - fov / 2) * 2) / 2;
      majorAxisScaling = (2 * M_PI - edge1 - edge2) / fov;
    } else {
      majorAxisScaling = (sampling_arc(offset, 2 * angularDist + fov) -
                          sampling_arc(offset, 2 * angularDist - fov)) /
          2 / fov;
    }
  } else {
    majorAxisScaling = (sampling_arc(offset, 2 * angularDist + fov) +
                        sampling_arc(offset, fov - 2 * angularDist)) /
        2 / fov;
  }
  double distToCoVertex = angular_distance(angularDist, 0.5 * fov, 0.0, 0.0);
  double minorAxisScaling = sampling_arc(offset, distToCoVertex * 2) / (distToCoVertex * 2);
  double r = majorAxisScaling * minorAxisScaling * spherical_area(fov) / K_SPHERE_AREA;
  return (1.0 < r) ? 1.0 : r; /* min(r, 1.0) */
}
double t360o_effective_ratio(double angularDist, double offset) {
  return effective_ratio_fov(angularDist, offset, K_FOV);
}

static void cfg_push(T360OFilterConfig* cfg, int left, int top, int width, int height, float* kx,
                     int kxl, float* ky, int kyl) {
  if (cfg->count == cfg->capacity) {
    cfg->capacity = cfg->capacity ? cfg->capacity * 2 : 64;
    cfg->seg = (T360OSegment*)realloc(cfg->seg, (size_t)cfg->capacity * sizeof(T360OSegment));
  }
  T360OSegment* s = &cfg->seg[cfg->count++];
  s->left = left;
  s->top = top;
  s->width = width;
  s->height = height;
  s->kx = kx;
  s->kx_len = kxl;
  s->ky = ky;
  s->ky_len = kyl;
}

static float* dup_kernel(const float* k, int n) {
  float* r = (float*)malloc((size_t)n * sizeof(float));
  memcpy(r, k, (size_t)n * sizeof(float));
  return r;
}

static int imin(int a, int b) { return b < a ? b : a; }
static int imax(int a, int b) { return a < b ? b : a; }

/* :210-297 */
static void kernel_and_config(const FrameTransformContext* c, T360OFilterConfig* cfg, int top,
                              int bottom, float angle, float sigmaY, const float* kernelY,
                              int kernelYLen, int inputWidth, int inputHeight) {
  /* min(0.5 * inputWidth, sigmaY / (cos(angle) + kEpsilon)); cos(float) is cosf */
  double a = 0.5 * inputWidth;
  double b = sigmaY / (cosf(angle) + kEpsilon);
  float sigmaX = (float)((b < a) ? b : a);

  int kxl;
  float* kernelX = t360o_calculate_kernel(sigmaX, &kxl);

  int numHorizontalSegments = c->adjust_kernel ? c->num_horizontal_segments : 1;
  int segmentWidth = (int)ceil(1.0 * inputWidth / numHorizontalSegments);
  double baseEffectiveRatio = t360o_effective_ratio(0.0, 0.0);

  for (int i = 0; i < numHorizontalSegments && i * segmentWidth < inputWidth; ++i) {
    int w = imin(segmentWidth, inputWidth - i * segmentWidth);
    if (c->adjust_kernel) {
      float avgYaw = (float)(2 * M_PI *
                             ((i * segmentWidth + 0.5 * imin(segmentWidth, inputWidth - i * segmentWidth)) -
                              0.5 * inputWidth) /
                             inputWidth);
      float avgPitch = (float)(0.5 * M_PI * (inputHeight - top - bottom) / inputHeight);
      float yaw = (float)(c->fixed_yaw * M_PI / 180.0f);
      float pitch = (float)(c->fixed_pitch * M_PI / 180.0f);
      float offset = fabsf(c->fixed_cube_offcenter_z);
      if (fabsf(yaw) < kEpsilon && fabsf(pitch) < kEpsilon &&
          (fabsf(c->fixed_cube_offcenter_x) > kEpsilon ||
           fabsf(c->fixed_cube_offcenter_y) > kEpsilon || c->fixed_cube_offcenter_z > kEpsilon)) {
        offset = sqrtf(c->fixed_cube_offcenter_x * c->fixed_cube_offcenter_x +
                       c->fixed_cube_offcenter_y * c->fixed_cube_offcenter_y +
                       c->fixed_cube_offcenter_z * c->fixed_cube_offcenter_z);
        yaw = atan2f(-c->fixed_cube_offcenter_x / offset, -c->fixed_cube_offcenter_z / offset);
        pitch = asinf(-c->fixed_cube_offcenter_y / offset);
      }
      double dist = angular_distance(yaw, pitch, avgYaw, avgPitch);
      double effectiveRatio = t360o_effective_ratio(dist, offset);
      double kernelScalingFactor = c->kernel_adjust_factor * baseEffectiveRatio / effectiveRatio;
      int axl, ayl;
      float* ax = t360o_calculate_kernel((float)(kernelScalingFactor * sigmaX), &axl);
      float* ay = t360o_calculate_kernel((float)(kernelScalingFactor * sigmaY), &ayl);
      cfg_push(cfg, i * segmentWidth, top, w, bottom - top + 1, ax, axl, ay, ayl);
    } else {
      cfg_push(cfg, i * segmentWidth, top, w, bottom - top + 1, dup_kernel(kernelX, kxl), kxl,
               dup_kernel(kernelY, kernelYLen), kernelYLen);
    }
  }
  free(kernelX);
}

/* :318-364 */
static void kernels_and_configs(const FrameTransformContext* c, T360OFilterConfig* cfg,
                                int startTop, int startBottom, float sigmaY, const float* kernelY,
                                int kernelYLen, int baseSegmentHeight, int inputWidth,
                                int inputHeight) {
  for (int bottom = startBottom; bottom >= 0; bottom -= baseSegmentHeight) {
    int top = imax(bottom - baseSegmentHeight + 1, 0);
    float angle = (float)(0.5 * M_PI * (inputHeight - top - bottom) / inputHeight);
    kernel_and_config(c, cfg, top, bottom, angle, sigmaY, kernelY, kernelYLen, inputWidth,
                      inputHeight);
  }
  for (int top = startTop; top < inputHeight; top += baseSegmentHeight) {
    int bottom = imin(top + baseSegmentHeight - 1, inputHeight - 1);
    float angle = (float)(0.5 * M_PI * (top + bottom - inputHeight) / inputHeight);
    kernel_and_config(c, cfg, top, bottom, angle, sigmaY, kernelY, kernelYLen, inputWidth,
                      inputHeight);
  }
}

static float fminf_std(float a, float b) { return (b < a) ? b : a; } /* std::min */
static float fmaxf_std(float a, float b) { return (a < b) ? b : a; } /* std::max */

/* :367-501 */
int t360o_filter_config(const FrameTransformContext* c, int inputWidth, int inputHeight,
                         int outputWidth, int outputHeight, T360OFilterConfig* cfg) {
  t_kernel_error = 0;
  switch (c->input_stereo_format) {
    case STEREO_FORMAT_LR: inputWidth = (int)(inputWidth * 0.5); break;
    case STEREO_FORMAT_TB: inputHeight = (int)(inputHeight * 0.5); break;
    default: break;
  }
  switch (c->output_stereo_format) {
    case STEREO_FORMAT_LR: outputWidth = (int)(outputWidth * 0.5); break;
    case STEREO_FORMAT_TB: outputHeight = (int)(outputHeight * 0.5); break;
    default: break;
  }
  float hFov, vFov;
  switch (c->output_layout) {
    case LAYOUT_CUBEMAP_32: hFov = 270.0f; vFov = 180.0f; break;
    case LAYOUT_CUBEMAP_23_OFFCENTER: hFov = 180.0f; vFov = 270.0f; break;
    case LAYOUT_FLAT_FIXED: hFov = c->fixed_hfov; vFov = c->fixed_vfov; break;
    case LAYOUT_EQUIRECT: hFov = 360.0f; vFov = 180.0f; break;
    case LAYOUT_BARREL:
    case LAYOUT_BARREL_SPLIT: hFov = 450.0f; vFov = 90.0f; break;
    case LAYOUT_EAC_32: hFov = 270.0f; vFov = 180.0f; break;
    default: printf("Invalid layout type.\n"); return 0;
  }
  float sigmaY = 0.5f *
      fminf_std(c->max_kernel_half_height,
                fmaxf_std(c->min_kernel_half_height,
                          c->kernel_height_scale_factor *
                              fminf_std(inputWidth / 360.0f, inputHeight / 180.0f) /
                              fmaxf_std(outputWidth / hFov, outputHeight / vFov)));
  int kyl;
  float* kernelY = t360o_calculate_kernel(sigmaY, &kyl);
  int baseSegmentHeight = (int)ceil(1.0 * inputHeight / c->num_vertical_segments);

  if (c->num_vertical_segments % 2 == 0) {
    kernels_and_configs(c, cfg, (int)(0.5 * inputHeight), (int)(0.5 * inputHeight - 1), sigmaY,
                        kernelY, kyl, baseSegmentHeight, inputWidth, inputHeight);
  } else {
    int top = (int)(0.5 * (inputHeight - baseSegmentHeight));
    int bottom = top + baseSegmentHeight - 1;
    kernel_and_config(c, cfg, top, bottom, 0, sigmaY, kernelY, kyl, inputWidth, inputHeight);
    kernels_and_configs(c, cfg, bottom + 1, top - 1, sigmaY, kernelY, kyl, baseSegmentHeight,
                        inputWidth, inputHeight);
  }
  free(kernelY);
  return t_kernel_error ? 0 : 1;
}

void t360o_filter_config_free(T360OFilterConfig* cfg) {
  for (int i = 0; i < cfg->count; ++i) {
    free(cfg->seg[i].kx);
    free(cfg->seg[i].ky);
  }
  free(cfg->seg);
  cfg->seg = NULL;
  cfg->count = cfg->capacity = 0;
}

uint64_t t360o_fnv1a64(const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}
