"""Named parity cases shared by the golden-vector generator and the tests.

BASELINE.json configs (SURVEY.md 8 plane-shape table) plus the edge cases the reference's
code paths distinguish: every interpolation kernel, the low-pass on/off, TB / LR stereo,
rotation + off-centre projection, the second cube atlas, FLAT_FIXED, a cubemap input, odd
sizes (ragged chroma), padded strides.
"""
from transform360_amd.abi import (CUBIC, LANCZOS4, LAYOUT_BARREL, LAYOUT_BARREL_SPLIT, LAYOUT_CUBEMAP_23_OFFCENTER,
                                  LAYOUT_CUBEMAP_32, LAYOUT_EAC_32, LAYOUT_EQUIRECT, LAYOUT_FLAT_FIXED, LINEAR,
                                  NEAREST, STEREO_FORMAT_LR, STEREO_FORMAT_TB, filter_defaults)

TB = dict(input_stereo_format=STEREO_FORMAT_TB, output_stereo_format=STEREO_FORMAT_TB)
LR = dict(input_stereo_format=STEREO_FORMAT_LR, output_stereo_format=STEREO_FORMAT_LR)

# name -> (ctx overrides, (inW, inH, outW, outH))   -- map cases
MAP_CASES = {
    # BASELINE configs, luma and chroma shapes
    "cfg1_luma": (dict(interpolation_alg=NEAREST, enable_low_pass_filter=0), (1920, 960, 768, 512)),
    "cfg1_chroma": (dict(interpolation_alg=NEAREST, enable_low_pass_filter=0), (960, 480, 384, 256)),
    "cfg2_luma": (dict(enable_low_pass_filter=0), (3840, 1920, 1536, 1024)),
    "cfg2_chroma": (dict(enable_low_pass_filter=0), (1920, 960, 768, 512)),
    "cfg4_luma": (dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0, **TB), (7680, 3840, 3072, 4096)),
    "cfg4_chroma": (dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0, **TB), (3840, 1920, 1536, 2048)),
    "rotated": (dict(enable_low_pass_filter=0, fixed_yaw=30, fixed_pitch=-15, fixed_roll=5,
                     fixed_cube_offcenter_z=-0.3), (1920, 960, 768, 512)),
    # further code paths of transformPos
    "lr_vflip": (dict(enable_low_pass_filter=0, vflip=1, **LR), (2048, 512, 768, 256)),
    "tb_vflip": (dict(enable_low_pass_filter=0, vflip=1, **TB), (1024, 1024, 384, 512)),
    "tb_to_lr": (dict(enable_low_pass_filter=0, input_stereo_format=STEREO_FORMAT_TB,
                      output_stereo_format=STEREO_FORMAT_LR), (1024, 1024, 768, 256)),
    "cube23_offcenter_h": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_CUBEMAP_23_OFFCENTER,
                                fixed_cube_offcenter_x=0.1, fixed_cube_offcenter_z=-0.5,
                                is_horizontal_offset=1), (1024, 512, 256, 384)),
    "offcenter_xyz": (dict(enable_low_pass_filter=0, fixed_cube_offcenter_x=-0.2, fixed_cube_offcenter_y=0.15,
                           fixed_cube_offcenter_z=0.4, fixed_yaw=-100), (1024, 512, 384, 256)),
    "flat_fixed": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_FLAT_FIXED, fixed_yaw=200,
                        fixed_pitch=80), (1024, 512, 320, 200)),
    "cubemap_input": (dict(enable_low_pass_filter=0, input_layout=LAYOUT_CUBEMAP_32, fixed_yaw=45,
                           fixed_pitch=20), (768, 512, 384, 256)),
    "odd_sizes": (dict(enable_low_pass_filter=0, interpolation_alg=LINEAR), (1001, 499, 336, 224)),
}

# the remaining output layouts: per-column / per-row libm tables on the HIP path
LAYOUT_MAP_CASES = {
    "eac32": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_EAC_32), (1024, 512, 384, 256)),
    "equirect_out": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_EQUIRECT, fixed_yaw=77), (1024, 512, 512, 256)),
    "barrel": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_BARREL), (1024, 512, 640, 256)),
    "barrel_split": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_BARREL_SPLIT), (1024, 512, 384, 256)),
}

# low-pass configuration cases: name -> (overrides, dims)
LOWPASS_CASES = {
    "cfg3_luma": (dict(num_vertical_segments=15, num_horizontal_segments=32), (3840, 1920, 1536, 1024)),
    "cfg3_chroma": (dict(num_vertical_segments=15, num_horizontal_segments=32), (1920, 960, 768, 512)),
    "defaults": (dict(), (1920, 960, 768, 512)),
    "even_bands_noadjust": (dict(num_vertical_segments=6, adjust_kernel=0), (1024, 512, 384, 256)),
    "offcenter_adjust": (dict(fixed_cube_offcenter_y=0.2, fixed_cube_offcenter_z=0.4, num_horizontal_segments=7,
                              num_vertical_segments=9, kernel_adjust_factor=1.7), (1024, 512, 384, 256)),
    "tb": (dict(num_vertical_segments=4, **TB), (1024, 1024, 384, 512)),
    "lr_odd": (dict(num_vertical_segments=7, num_horizontal_segments=3, **LR), (2047, 511, 768, 256)),
    "heavy_blur": (dict(kernel_height_scale_factor=6.0, num_vertical_segments=9, num_horizontal_segments=4),
                   (1024, 512, 192, 128)),
}

# whole frame-path cases at sizes the oracle finishes in milliseconds:
# name -> (overrides, dims, stride pad in, stride pad out)
FRAME_CASES = {
    "nearest": (dict(interpolation_alg=NEAREST, enable_low_pass_filter=0), (1920, 960, 768, 512), 0, 0),
    "linear": (dict(interpolation_alg=LINEAR, enable_low_pass_filter=0), (1920, 960, 768, 512), 32, 16),
    "cubic": (dict(interpolation_alg=CUBIC, enable_low_pass_filter=0), (1920, 960, 768, 512), 32, 16),
    "lanczos4": (dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0), (1920, 960, 768, 512), 0, 64),
    "cubic_lpf_32x15": (dict(num_vertical_segments=15, num_horizontal_segments=32), (1920, 960, 768, 512), 64, 0),
    "cubic_lpf_defaults": (dict(), (960, 480, 384, 256), 0, 0),
    "lanczos_tb": (dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0, **TB), (960, 960, 384, 512), 0, 0),
    "cubic_tb_lpf": (dict(num_vertical_segments=4, **TB), (1024, 1024, 384, 512), 8, 8),
    "cubic_lr_lpf_odd": (dict(num_vertical_segments=7, num_horizontal_segments=3, **LR), (2047, 511, 768, 256), 1, 3),
    "rotated_offcenter": (dict(enable_low_pass_filter=0, fixed_yaw=30, fixed_pitch=-15, fixed_roll=5,
                               fixed_cube_offcenter_z=-0.3), (1024, 512, 384, 256), 0, 0),
    "cube23": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_CUBEMAP_23_OFFCENTER), (1024, 512, 256, 384), 0, 0),
    "flat_fixed": (dict(output_layout=LAYOUT_FLAT_FIXED, fixed_yaw=200, fixed_pitch=80, interpolation_alg=LINEAR),
                   (1024, 512, 320, 200), 0, 0),
    "cubemap_input_nearest": (dict(enable_low_pass_filter=0, input_layout=LAYOUT_CUBEMAP_32, interpolation_alg=NEAREST,
                                   fixed_yaw=45), (768, 512, 384, 256), 0, 0),
    "heavy_blur": (dict(kernel_height_scale_factor=6.0, num_vertical_segments=9, num_horizontal_segments=4),
                   (1024, 512, 192, 128), 0, 0),
    "tiny": (dict(enable_low_pass_filter=0), (64, 32, 48, 32), 3, 5),
}

# supersample + cv::resize(INTER_AREA): 2x2 (shift rounding), other integer
# factors (float scale), fractional factors (DecimateAlpha tables)
SUPERSAMPLE_FRAME_CASES = {
    "supersample_2x2": (dict(enable_low_pass_filter=0, width_scale_factor=2.0, height_scale_factor=2.0),
                        (1024, 512, 384, 256), 0, 0),
    "supersample_3x2_linear": (dict(enable_low_pass_filter=0, width_scale_factor=3.0, height_scale_factor=2.0,
                                    interpolation_alg=LINEAR), (1024, 512, 384, 256), 32, 16),
    "supersample_1p5": (dict(enable_low_pass_filter=0, width_scale_factor=1.5, height_scale_factor=1.5),
                        (1024, 512, 384, 256), 0, 0),
    "supersample_1p3x2p7_barrel": (dict(enable_low_pass_filter=0, width_scale_factor=1.3, height_scale_factor=2.7,
                                        output_layout=LAYOUT_BARREL), (1024, 512, 640, 256), 0, 0),
    # factors below 1 (the filter accepts 0..10, vf_transform360.c:888-905): cv::resize(INTER_AREA) ENLARGES, which
    # OpenCV emulates with its bilinear kernels
    "subsample_0p5": (dict(enable_low_pass_filter=0, width_scale_factor=0.5, height_scale_factor=0.5),
                      (1024, 512, 384, 256), 0, 0),
    "subsample_0p7x0p4_linear": (dict(enable_low_pass_filter=0, width_scale_factor=0.7, height_scale_factor=0.4,
                                      interpolation_alg=LINEAR), (1024, 512, 384, 256), 16, 32),
    "mixed_1p5x0p6": (dict(enable_low_pass_filter=0, width_scale_factor=1.5, height_scale_factor=0.6),
                      (1024, 512, 384, 256), 0, 0),
}

LAYOUT_FRAME_CASES = {
    "barrel_cubic": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_BARREL), (1024, 512, 640, 256), 0, 0),
    "barrel_split_linear": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_BARREL_SPLIT, interpolation_alg=LINEAR),
                            (1024, 512, 384, 256), 0, 0),
    "eac_lanczos": (dict(enable_low_pass_filter=0, output_layout=LAYOUT_EAC_32, interpolation_alg=LANCZOS4),
                    (1024, 512, 384, 256), 0, 0),
}


def make_ctx(overrides):
    return filter_defaults(**overrides)


def case_input(name, in_w, in_h, pad):
    """Deterministic noise input of a frame case (the same on every machine)."""
    import zlib

    import numpy as np

    from transform360_amd.handler import noise_bytes
    seed = zlib.crc32(name.encode()) | (1 << 33)
    buf = noise_bytes(in_h * (in_w + pad), seed).reshape(in_h, in_w + pad)
    return np.ascontiguousarray(buf)[:, :in_w]


# ---- OpenCV pin (tests/golden/make_opencv_fixtures.py, run on any machine that has cv2) ----
# Small frame cases whose outputs come from REAL OpenCV: every interpolation x both border modes of cv::remap, the
# fixed-point and the float path of cv::sepFilter2D (segments with real neighbours), cv::resize(INTER_AREA) shrinking
# by 2x2 / 3x2 / 1.5 and enlarging (factors below 1).  MONO only: the stereo orchestration is the reference's own code
# and is pinned by oracle/_ref; this file pins the arithmetic underneath.  name -> (overrides, dims)
def _ocv(interp, **kw):
    return dict(interpolation_alg=interp, enable_low_pass_filter=0, **kw)


OPENCV_CASES = {
    "ocv_nearest_wrap": (_ocv(NEAREST), (256, 128, 96, 64)),
    "ocv_linear_wrap": (_ocv(LINEAR), (256, 128, 96, 64)),
    "ocv_cubic_wrap": (_ocv(CUBIC), (256, 128, 96, 64)),
    "ocv_lanczos4_wrap": (_ocv(LANCZOS4), (256, 128, 96, 64)),
    "ocv_cubic_wrap_rotated": (_ocv(CUBIC, fixed_yaw=171, fixed_pitch=80, fixed_roll=13), (320, 160, 192, 128)),
    "ocv_lanczos4_wrap_rotated": (_ocv(LANCZOS4, fixed_yaw=-179, fixed_pitch=-85), (320, 160, 192, 128)),
    "ocv_nearest_transparent": (_ocv(NEAREST, output_layout=LAYOUT_BARREL), (256, 128, 160, 64)),
    "ocv_linear_transparent": (_ocv(LINEAR, output_layout=LAYOUT_BARREL), (256, 128, 160, 64)),
    "ocv_cubic_transparent": (_ocv(CUBIC, output_layout=LAYOUT_BARREL), (256, 128, 160, 64)),
    "ocv_lanczos4_transparent": (_ocv(LANCZOS4, output_layout=LAYOUT_BARREL_SPLIT), (256, 128, 96, 64)),
    # low-pass: integer (Q8 x Q8) path of cv::sepFilter2D -- small symmetric kernels -- and the float path (long ones)
    "ocv_lpf_fixed_point": (dict(num_vertical_segments=5, num_horizontal_segments=4), (512, 256, 192, 128)),
    "ocv_lpf_32x15": (dict(num_vertical_segments=15, num_horizontal_segments=32), (960, 480, 192, 128)),
    "ocv_lpf_float_heavy": (dict(kernel_height_scale_factor=6.0, num_vertical_segments=9, num_horizontal_segments=4),
                            (512, 256, 96, 64)),
    "ocv_lpf_noadjust_even": (dict(num_vertical_segments=6, adjust_kernel=0, interpolation_alg=LINEAR), (512, 256, 192, 128)),
    # supersample + INTER_AREA
    "ocv_area_2x2": (_ocv(CUBIC, width_scale_factor=2.0, height_scale_factor=2.0), (512, 256, 96, 64)),
    "ocv_area_3x2": (_ocv(LINEAR, width_scale_factor=3.0, height_scale_factor=2.0), (512, 256, 96, 64)),
    "ocv_area_1p5": (_ocv(CUBIC, width_scale_factor=1.5, height_scale_factor=1.5), (512, 256, 96, 64)),
    "ocv_area_0p5": (_ocv(CUBIC, width_scale_factor=0.5, height_scale_factor=0.5), (512, 256, 96, 64)),
    "ocv_area_0p7x0p4": (_ocv(LINEAR, width_scale_factor=0.7, height_scale_factor=0.4), (512, 256, 96, 64)),
    "ocv_area_1p3x2p7_transparent": (_ocv(CUBIC, width_scale_factor=1.3, height_scale_factor=2.7, output_layout=LAYOUT_BARREL),
                                     (512, 256, 160, 64)),
}
