"""Every output layout x every interpolation at BASELINE config 2's input size (3840x1920 yuv420p, short batches through
T360_transformFrames), plus rotation, off-centre cubes, stereo packings, cubemap input, the low-pass and the supersample
branch -- against per-plane oracle calls, bit-exact.  tests/test_gpu_parity.py pins these paths on small planes; at full
size every tile shape of the gather plans, the pole tiles and the seam appear for each projection."""
import pytest

from tests.test_gpu_parity import T, _batch_case  # noqa: F401  (T: the module-scoped handler fixture)
from transform360_amd.abi import (CUBIC, LANCZOS4, LAYOUT_BARREL, LAYOUT_BARREL_SPLIT, LAYOUT_CUBEMAP_23_OFFCENTER,
                                  LAYOUT_CUBEMAP_32, LAYOUT_EAC_32, LAYOUT_EQUIRECT, LAYOUT_FLAT_FIXED, LINEAR, NEAREST,
                                  STEREO_FORMAT_LR, STEREO_FORMAT_TB)

pytestmark = pytest.mark.gpu

IN = (3840, 1920)
LAYOUTS = [("CUBEMAP_32", LAYOUT_CUBEMAP_32, (1536, 1024)), ("CUBEMAP_23_OFFCENTER", LAYOUT_CUBEMAP_23_OFFCENTER, (1024, 1536)),
           ("EAC_32", LAYOUT_EAC_32, (1536, 1024)), ("EQUIRECT", LAYOUT_EQUIRECT, (2048, 1024)),
           ("FLAT_FIXED", LAYOUT_FLAT_FIXED, (1280, 720)), ("BARREL", LAYOUT_BARREL, (1920, 768)),
           ("BARREL_SPLIT", LAYOUT_BARREL_SPLIT, (1920, 768))]
INTERPS = [("nearest", NEAREST), ("bilinear", LINEAR), ("bicubic", CUBIC), ("lanczos4", LANCZOS4)]

CASES = {}
for _l, _layout, _out in LAYOUTS:
    for _i, _interp in INTERPS:
        CASES["%s-%s" % (_l, _i)] = (dict(output_layout=_layout, interpolation_alg=_interp, enable_low_pass_filter=0), IN, _out,
                                     2 if _interp == LANCZOS4 else 3)
CASES.update({
    "CUBEMAP_32-bicubic-yaw37-pitch-21-roll11": (dict(fixed_yaw=37.0, fixed_pitch=-21.0, fixed_roll=11.0, enable_low_pass_filter=0), IN, (1536, 1024), 3),
    "CUBEMAP_23_OFFCENTER-bicubic-offcentre": (dict(output_layout=LAYOUT_CUBEMAP_23_OFFCENTER, fixed_cube_offcenter_x=0.2, fixed_cube_offcenter_y=-0.1,
                                                    fixed_cube_offcenter_z=0.6, enable_low_pass_filter=0), IN, (1024, 1536), 3),
    "CUBEMAP_32-bicubic-lowpass-32x15": (dict(num_horizontal_segments=32, num_vertical_segments=15, adjust_kernel=1), IN, (1536, 1024), 3),
    "CUBEMAP_32-bicubic-lowpass-8x5-TB": (dict(num_horizontal_segments=8, num_vertical_segments=5, input_stereo_format=STEREO_FORMAT_TB,
                                               output_stereo_format=STEREO_FORMAT_TB), IN, (1536, 2048), 3),
    "CUBEMAP_32-bilinear-LR-in-LR-out": (dict(interpolation_alg=LINEAR, input_stereo_format=STEREO_FORMAT_LR, output_stereo_format=STEREO_FORMAT_LR,
                                              enable_low_pass_filter=0), IN, (3072, 1024), 3),
    "EAC_32-bicubic-vflip-expand-1.03": (dict(output_layout=LAYOUT_EAC_32, vflip=1, expand_coef=1.03, enable_low_pass_filter=0), IN, (1536, 1024), 3),
    "EQUIRECT-from-CUBEMAP_32-bicubic": (dict(input_layout=LAYOUT_CUBEMAP_32, output_layout=LAYOUT_EQUIRECT, enable_low_pass_filter=0),
                                         (3072, 2048), (2048, 1024), 3),
    "CUBEMAP_32-bicubic-supersample-2x2": (dict(width_scale_factor=2.0, height_scale_factor=2.0, enable_low_pass_filter=0), IN, (768, 512), 3),
})


@pytest.mark.parametrize("name", sorted(CASES))
def test_full_size_batch_matches_oracle(name, T, oracle_mod):
    ov, inp, out, n = CASES[name]
    _batch_case(T, oracle_mod, ov, n=n, dims=(inp[0], inp[1], out[0], out[1]), extra_pad=0, threads=32)
