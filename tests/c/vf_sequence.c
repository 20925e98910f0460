/*
 * tests/c/vf_sequence.c -- stands in for ffmpeg: replays, from plain C, the exact sequence of
 * library calls the transform360 filter makes (reference Transform360/vf_transform360.c):
 *   generate_map  (:99-165)  stack-local FrameTransformContext -> VideoFrameTransform_new ->
 *                            VideoFrameTransform_generateMapForPlane for plane 0 (luma dims) and
 *                            plane 1 (chroma dims via FF_CEIL_RSHIFT, :87-97)
 *   filter_frame  (:338-402) per frame, per plane: VideoFrameTransform_transformFramePlane with
 *                            AVFrame-style padded linesizes, map index (plane==1||plane==2)?1:0
 *   uninit        (:328-336) VideoFrameTransform_delete
 * Frames are host memory (malloc), as ffmpeg's are.  Prints one FNV-1a-64 per output plane.
 *
 *   usage: vf_sequence in_w in_h cube_edge interp lowpass nframes
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Transform360/VideoFrameTransformHandler.h"

#define CEIL_RSHIFT(a, b) (-((-(a)) >> (b)))

static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static uint64_t fnv1a(const uint8_t* p, int w, int h, int stride) {
  uint64_t hsh = 1469598103934665603ull;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      hsh ^= p[(size_t)y * stride + x];
      hsh *= 1099511628211ull;
    }
  return hsh;
}

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const int in_w = atoi(argv[1]), in_h = atoi(argv[2]), edge = atoi(argv[3]);
  const int interp = atoi(argv[4]), lowpass = atoi(argv[5]), nframes = atoi(argv[6]);
  const int out_w = (edge - edge % 16) * 3, out_h = (edge - edge % 16) * 2; /* config_output :213-218 */

  /* the filter's option defaults (:407-987), copied field by field like generate_map */
  FrameTransformContext ctx = {
      .input_layout = LAYOUT_EQUIRECT, .output_layout = LAYOUT_CUBEMAP_32,
      .input_stereo_format = STEREO_FORMAT_MONO, .output_stereo_format = STEREO_FORMAT_MONO,
      .vflip = 0, .input_expand_coef = 1.01f, .expand_coef = 1.01f,
      .interpolation_alg = (InterpolationAlg)interp, .width_scale_factor = 1.0f, .height_scale_factor = 1.0f,
      .fixed_yaw = 0, .fixed_pitch = 0, .fixed_roll = 0, .fixed_hfov = 120.0f, .fixed_vfov = 110.0f,
      .fixed_cube_offcenter_x = 0, .fixed_cube_offcenter_y = 0, .fixed_cube_offcenter_z = 0,
      .is_horizontal_offset = 0, .enable_low_pass_filter = lowpass, .kernel_height_scale_factor = 1.0f,
      .min_kernel_half_height = 1.0f, .max_kernel_half_height = 10000.0f, .enable_multi_threading = 1,
      .num_vertical_segments = 15, .num_horizontal_segments = 32, .adjust_kernel = 1, .kernel_adjust_factor = 1.0f};

  VideoFrameTransform* t = VideoFrameTransform_new(&ctx);
  if (!t) {
    printf("ENOMEM\n");
    return 1;
  }
  memset(&ctx, 0xEE, sizeof ctx); /* the library must have copied the block (:141, VFT.cpp:206-208) */

  const int log2_chroma = 1; /* yuv420p */
  for (int plane = 0; plane < 2; plane++) {
    int iw = in_w, ih = in_h, ow = out_w, oh = out_h;
    if (plane == 1) {
      iw = CEIL_RSHIFT(iw, log2_chroma); ih = CEIL_RSHIFT(ih, log2_chroma);
      ow = CEIL_RSHIFT(ow, log2_chroma); oh = CEIL_RSHIFT(oh, log2_chroma);
    }
    if (!VideoFrameTransform_generateMapForPlane(t, iw, ih, ow, oh, plane)) {
      printf("EINVAL map %d\n", plane);
      return 1;
    }
  }

  for (int f = 0; f < nframes; f++) {
    for (int plane = 0; plane < 3; plane++) {
      int iw = in_w, ih = in_h, ow = out_w, oh = out_h;
      if (plane >= 1) {
        iw = CEIL_RSHIFT(iw, log2_chroma); ih = CEIL_RSHIFT(ih, log2_chroma);
        ow = CEIL_RSHIFT(ow, log2_chroma); oh = CEIL_RSHIFT(oh, log2_chroma);
      }
      const int in_ls = (iw + 63) / 64 * 64 + 64, out_ls = (ow + 31) / 32 * 32 + 32; /* padded linesizes */
      uint8_t* in = malloc((size_t)in_ls * ih);
      uint8_t* out = malloc((size_t)out_ls * oh);
      const uint64_t seed = 0x360ull ^ ((uint64_t)f << 40) ^ ((uint64_t)plane << 36);
      for (size_t i = 0; i < (size_t)in_ls * ih; i++) in[i] = (uint8_t)(splitmix64(seed + i) >> 56);
      memset(out, 0x5A, (size_t)out_ls * oh);
      const int idx = (plane == 1 || plane == 2) ? 1 : 0;
      if (!VideoFrameTransform_transformFramePlane(t, in, out, iw, ih, in_ls, ow, oh, out_ls, idx, plane)) {
        printf("EINVAL frame %d plane %d\n", f, plane);
        return 1;
      }
      int pad_ok = 1;
      for (int y = 0; y < oh && pad_ok; y++)
        for (int x = ow; x < out_ls; x++)
          if (out[(size_t)y * out_ls + x] != 0x5A) pad_ok = 0;
      printf("frame %d plane %d %dx%d hash %016llx pad %s\n", f, plane, ow, oh,
             (unsigned long long)fnv1a(out, ow, oh, out_ls), pad_ok ? "intact" : "CLOBBERED");
      free(in);
      free(out);
    }
  }
  VideoFrameTransform_delete(t);
  VideoFrameTransform_delete(NULL);
  return 0;
}
