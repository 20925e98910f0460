"""ctypes mirror of include/Transform360/VideoFrameTransformHelper.h.

The enumerations and the 112-byte ``FrameTransformContext`` block are the configuration
surface of the reference (reference Transform360/Library/VideoFrameTransformHelper.h:18-90).
``filter_defaults`` reproduces the ffmpeg option defaults of the ``transform360`` filter
(reference Transform360/vf_transform360.c:407-987) and ``config_output`` its output-size
rules (vf_transform360.c:167-304), so tests can build the exact context the filter would.
"""
import ctypes as C

# TransformFaceType (Helper.h:18-25)
RIGHT, LEFT, TOP, BOTTOM, FRONT, BACK = range(6)
# Layout without FACEBOOK_LAYOUT (Helper.h:27-39)
(LAYOUT_CUBEMAP_32, LAYOUT_CUBEMAP_23_OFFCENTER, LAYOUT_FLAT_FIXED, LAYOUT_EQUIRECT,
 LAYOUT_BARREL, LAYOUT_BARREL_SPLIT, LAYOUT_EAC_32, LAYOUT_N) = range(8)
# StereoFormat (Helper.h:41-47)
STEREO_FORMAT_TB, STEREO_FORMAT_LR, STEREO_FORMAT_MONO, STEREO_FORMAT_GUESS, STEREO_FORMAT_N = range(5)
# InterpolationAlg (Helper.h:49-54): the values are cv::INTER_* codes
NEAREST, LINEAR, CUBIC, LANCZOS4 = 0, 1, 2, 4


class FrameTransformContext(C.Structure):
    """Helper.h:56-90 -- 28 four-byte fields, copied by value at VideoFrameTransform_new."""
    _fields_ = [
        ("input_layout", C.c_int),
        ("output_layout", C.c_int),
        ("input_stereo_format", C.c_int),
        ("output_stereo_format", C.c_int),
        ("vflip", C.c_int),
        ("input_expand_coef", C.c_float),
        ("expand_coef", C.c_float),
        ("interpolation_alg", C.c_int),
        ("width_scale_factor", C.c_float),
        ("height_scale_factor", C.c_float),
        ("fixed_yaw", C.c_float),
        ("fixed_pitch", C.c_float),
        ("fixed_roll", C.c_float),
        ("fixed_hfov", C.c_float),
        ("fixed_vfov", C.c_float),
        ("fixed_cube_offcenter_x", C.c_float),
        ("fixed_cube_offcenter_y", C.c_float),
        ("fixed_cube_offcenter_z", C.c_float),
        ("is_horizontal_offset", C.c_int),
        ("enable_low_pass_filter", C.c_int),
        ("kernel_height_scale_factor", C.c_float),
        ("min_kernel_half_height", C.c_float),
        ("max_kernel_half_height", C.c_float),
        ("enable_multi_threading", C.c_int),
        ("num_vertical_segments", C.c_int),
        ("num_horizontal_segments", C.c_int),
        ("adjust_kernel", C.c_int),
        ("kernel_adjust_factor", C.c_float),
    ]


assert C.sizeof(FrameTransformContext) == 112


def filter_defaults(**overrides):
    """The context the ffmpeg filter builds with its default options
    (vf_transform360.c:407-987 defaults, copied field by field at :111-139).
    Stereo formats default to MONO here (GUESS is resolved by config_output)."""
    ctx = FrameTransformContext(
        input_layout=LAYOUT_EQUIRECT,
        output_layout=LAYOUT_CUBEMAP_32,
        input_stereo_format=STEREO_FORMAT_MONO,
        output_stereo_format=STEREO_FORMAT_MONO,
        vflip=0,
        input_expand_coef=1.01,
        expand_coef=1.01,
        interpolation_alg=CUBIC,
        width_scale_factor=1.0,
        height_scale_factor=1.0,
        fixed_yaw=0.0,
        fixed_pitch=0.0,
        fixed_roll=0.0,
        fixed_hfov=120.0,
        fixed_vfov=110.0,
        fixed_cube_offcenter_x=0.0,
        fixed_cube_offcenter_y=0.0,
        fixed_cube_offcenter_z=0.0,
        is_horizontal_offset=0,
        enable_low_pass_filter=1,
        kernel_height_scale_factor=1.0,
        min_kernel_half_height=1.0,
        max_kernel_half_height=10000.0,
        enable_multi_threading=1,
        num_vertical_segments=5,
        num_horizontal_segments=1,
        adjust_kernel=1,
        kernel_adjust_factor=1.0,
    )
    for k, v in overrides.items():
        if not hasattr(ctx, k):
            raise AttributeError("FrameTransformContext has no field %r" % k)
        setattr(ctx, k, v)
    return ctx


def guess_stereo(in_w, in_h, input_stereo_format, output_stereo_format, output_layout):
    """GUESS resolution, vf_transform360.c:178-196."""
    if input_stereo_format == STEREO_FORMAT_GUESS:
        aspect = in_w // in_h
        input_stereo_format = (STEREO_FORMAT_TB if aspect == 1 else
                               STEREO_FORMAT_LR if aspect == 4 else STEREO_FORMAT_MONO)
    if output_stereo_format == STEREO_FORMAT_GUESS:
        if input_stereo_format == STEREO_FORMAT_MONO:
            output_stereo_format = STEREO_FORMAT_MONO
        else:
            output_stereo_format = (STEREO_FORMAT_LR if output_layout == LAYOUT_CUBEMAP_23_OFFCENTER
                                    else STEREO_FORMAT_TB)
    return input_stereo_format, output_stereo_format


def config_output(in_w, in_h, cube_edge_length, output_layout=LAYOUT_CUBEMAP_32,
                  input_stereo_format=STEREO_FORMAT_MONO, output_stereo_format=STEREO_FORMAT_MONO,
                  max_cube_edge_length=0):
    """Output frame size for a cube layout, vf_transform360.c:198-223 and :293-299.
    Returns (out_w, out_h)."""
    if max_cube_edge_length > 0:
        cube_edge_length = in_w // 8 if input_stereo_format == STEREO_FORMAT_LR else in_w // 4
        cube_edge_length = min(cube_edge_length, max_cube_edge_length)
    cube_edge_length -= cube_edge_length % 16  # macroblocks must not straddle faces (:213)
    if cube_edge_length <= 0:
        raise ValueError("w/h expression outputs are not modelled; pass a cube_edge_length")
    if output_layout == LAYOUT_CUBEMAP_32:
        out_w, out_h = cube_edge_length * 3, cube_edge_length * 2
    elif output_layout == LAYOUT_CUBEMAP_23_OFFCENTER:
        out_w, out_h = cube_edge_length * 2, cube_edge_length * 3
    else:
        raise ValueError("cube_edge_length only sizes the two cube atlases")
    if output_stereo_format == STEREO_FORMAT_TB:
        out_h *= 2
    elif output_stereo_format == STEREO_FORMAT_LR:
        out_w *= 2
    return out_w, out_h


def chroma_dims(w, h, log2_chroma_w=1, log2_chroma_h=1):
    """FF_CEIL_RSHIFT plane sizes, vf_transform360.c:87-97."""
    return -((-w) >> log2_chroma_w), -((-h) >> log2_chroma_h)
