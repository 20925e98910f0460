/*
 * transform360-mi355x: configuration surface of the equirect->cubemap remap path.
 *
 * This header declares the plain-C configuration block and enumerations that the
 * ffmpeg filter (vf_transform360.c) fills in and hands to the library.  It is the
 * drop-in replacement for the reference header of the same name and must stay
 * layout-compatible with it:
 *
 *   reference: Transform360/Library/VideoFrameTransformHelper.h:18-90
 *     - TransformFaceType   (:18-25)   face order used by the CUBEMAP_32 atlas
 *     - Layout              (:27-39)   values WITHOUT the FACEBOOK_LAYOUT build flag
 *     - StereoFormat        (:41-47)
 *     - InterpolationAlg    (:49-54)   numeric values are cv::INTER_* (0,1,2,4)
 *     - FrameTransformContext (:56-90) 28 four-byte fields, 112 bytes, copied by value
 *       at VideoFrameTransform_new (reference VideoFrameTransform.cpp:206-208)
 *
 * Attribution: the declarations below (function names and signatures / enumerator names and values,
 * the field names, order and types of FrameTransformContext) reproduce the public interface of
 * facebook/transform360, Copyright (c) 2015-present, Facebook, Inc., released under the BSD license
 * in that project's LICENSE file.  They are repeated here only because binary compatibility with
 * that interface is the purpose of this library; everything behind them is an independent
 * implementation.
 *
 * Nothing here depends on HIP, OpenCV or C++.
 */
#ifndef TRANSFORM360_VIDEOFRAMETRANSFORMHELPER_H
#define TRANSFORM360_VIDEOFRAMETRANSFORMHELPER_H

#include <stdint.h>

/* Cube faces in atlas order: CUBEMAP_32 lays out  RIGHT LEFT TOP / BOTTOM FRONT BACK. */
typedef enum TransformFaceType {
  RIGHT = 0,
  LEFT = 1,
  TOP = 2,
  BOTTOM = 3,
  FRONT = 4,
  BACK = 5
} TransformFaceType;

/* Frame layouts.  The non-public LAYOUT_FB value of the reference is never present. */
typedef enum Layout {
  LAYOUT_CUBEMAP_32 = 0,           /* 3x2 face atlas (the path this library accelerates) */
  LAYOUT_CUBEMAP_23_OFFCENTER = 1, /* 2x3 face atlas */
  LAYOUT_FLAT_FIXED = 2,
  LAYOUT_EQUIRECT = 3,
  LAYOUT_BARREL = 4,
  LAYOUT_BARREL_SPLIT = 5,
  LAYOUT_EAC_32 = 6,
  LAYOUT_N = 7
} Layout;

typedef enum StereoFormat {
  STEREO_FORMAT_TB = 0,   /* top/bottom eyes */
  STEREO_FORMAT_LR = 1,   /* left/right eyes */
  STEREO_FORMAT_MONO = 2,
  STEREO_FORMAT_GUESS = 3,
  STEREO_FORMAT_N = 4
} StereoFormat;

/* Resampling kernels; the numbers are the OpenCV INTER_* codes the reference forwards. */
typedef enum InterpolationAlg {
  NEAREST = 0,
  LINEAR = 1,
  CUBIC = 2,
  LANCZOS4 = 4
} InterpolationAlg;

typedef struct FrameTransformContext {
  Layout input_layout;
  Layout output_layout;
  StereoFormat input_stereo_format;
  StereoFormat output_stereo_format;
  int vflip;
  float input_expand_coef;          /* expansion coefficient of a cubemap input */
  float expand_coef;                /* expansion coefficient of the output faces */
  InterpolationAlg interpolation_alg;
  float width_scale_factor;         /* supersampling factors (antialiasing) */
  float height_scale_factor;
  float fixed_yaw;                  /* degrees */
  float fixed_pitch;                /* degrees */
  float fixed_roll;                 /* degrees */
  float fixed_hfov;                 /* degrees */
  float fixed_vfov;                 /* degrees */
  float fixed_cube_offcenter_x;
  float fixed_cube_offcenter_y;
  float fixed_cube_offcenter_z;
  int is_horizontal_offset;
  int enable_low_pass_filter;
  float kernel_height_scale_factor;
  float min_kernel_half_height;
  float max_kernel_half_height;
  int enable_multi_threading;       /* reference: one std::thread per segment; here: ignored */
  int num_vertical_segments;
  int num_horizontal_segments;
  int adjust_kernel;
  float kernel_adjust_factor;
} FrameTransformContext;

#if defined(__cplusplus)
static_assert(sizeof(FrameTransformContext) == 112, "FrameTransformContext must stay 112 bytes");
#elif defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
_Static_assert(sizeof(FrameTransformContext) == 112, "FrameTransformContext must stay 112 bytes");
#endif

#endif /* TRANSFORM360_VIDEOFRAMETRANSFORMHELPER_H */
