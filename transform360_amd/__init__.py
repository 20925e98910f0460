"""transform360_amd -- MI355X-native equirect->cubemap remap path behind the reference's C ABI.

The package holds only what the hot path needs:
  csrc/       HIP kernels (projection, segmented low-pass, gather) + the C-ABI host library
  lib/        the built libTransform360.so (in-tree, git-ignored)
  abi.py      ctypes mirror of include/Transform360/VideoFrameTransformHelper.h
  handler.py  Python mirror of the reference's handler interface (tests, bench)
  sharding.py frame sharding of a synthetic stream across ranks (one process per GPU)

Importing the package does not load the HIP library; constructing a VideoFrameTransform does,
and fails loudly when it is missing (there is no CPU fallback).
"""
from .abi import *  # noqa: F401,F403
from .abi import FrameTransformContext, filter_defaults, config_output, chroma_dims, guess_stereo  # noqa: F401


def __getattr__(name):
    if name in ("VideoFrameTransform", "FrameLayout", "fill_noise", "noise_bytes", "frame_seed"):
        from . import handler
        return getattr(handler, name)
    raise AttributeError(name)
