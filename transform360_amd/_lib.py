"""Loader for the in-tree HIP library ``transform360_amd/lib/libTransform360.so``.

There is no CPU fallback: if the library is missing or does not export the full C ABI
declared in ``include/Transform360/*.h`` the import of the product path fails loudly.
"""
import ctypes as C
import os
import subprocess

from .abi import FrameTransformContext

_PKG = os.path.dirname(os.path.abspath(__file__))
# T360_LIB: A/B runs of two builds inside one gpurun call (tools/); never set in production
LIB_PATH = os.environ.get("T360_LIB") or os.path.join(_PKG, "lib", "libTransform360.so")
CSRC_DIR = os.path.join(_PKG, "csrc")

# every symbol include/Transform360/VideoFrameTransformHandler.h and t360_device.h declare
REFERENCE_SYMBOLS = (
    "VideoFrameTransform_new",
    "VideoFrameTransform_delete",
    "VideoFrameTransform_generateMapForPlane",
    "VideoFrameTransform_transformFramePlane",
)
ADDITIVE_SYMBOLS = (
    "T360_version", "T360_deviceCount", "T360_setStream", "T360_useOwnStream", "T360_synchronize", "T360_transformFrames",
    "T360_filterPlane", "T360_getMapSize", "T360_copyMap", "T360_getSegmentCount", "T360_getSegment",
    "T360_copySegmentKernels", "T360_fillNoise", "T360_lastKernel", "T360_lastLowpassPath", "T360_getPlanStats", "T360_buildFlags",
    "T360_transformFramesPipelined", "T360_transformFramesPipelinedMany", "T360_setPipelineDepth", "T360_pipelineJoin", "T360_setFusedLowpass",
)


class T360PlaneDesc(C.Structure):
    """include/Transform360/t360_device.h: T360PlaneDesc."""
    _fields_ = [
        ("in_offset", C.c_int64), ("out_offset", C.c_int64),
        ("in_stride", C.c_int), ("out_stride", C.c_int),
        ("in_width", C.c_int), ("in_height", C.c_int),
        ("out_width", C.c_int), ("out_height", C.c_int),
        ("map_index", C.c_int),
    ]


def build(verbose=False):
    """Compile the library for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC_DIR, "-j8", "all", "instr"], stdout=out)
    return LIB_PATH


_lib = None


def load():
    """dlopen the library and declare the prototypes.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C transform360_amd/csrc` (or "
            "__graft_entry__.build()); there is no CPU fallback for the remap path" % LIB_PATH)
    try:
        # PyTorch bundles its own libamdhip64.so.7; importing it first makes the library share
        # that runtime instance (device pointers and streams are then interchangeable).
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, the library works without it
        pass
    L = C.CDLL(LIB_PATH)
    missing = [s for s in REFERENCE_SYMBOLS + ADDITIVE_SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError("libTransform360.so does not export: %s" % ", ".join(missing))
    vp, i, u8p = C.c_void_p, C.c_int, C.c_void_p
    L.VideoFrameTransform_new.restype = vp
    L.VideoFrameTransform_new.argtypes = [C.POINTER(FrameTransformContext)]
    L.VideoFrameTransform_delete.restype = None
    L.VideoFrameTransform_delete.argtypes = [vp]
    L.VideoFrameTransform_generateMapForPlane.restype = i
    L.VideoFrameTransform_generateMapForPlane.argtypes = [vp, i, i, i, i, i]
    L.VideoFrameTransform_transformFramePlane.restype = i
    L.VideoFrameTransform_transformFramePlane.argtypes = [vp, u8p, u8p, i, i, i, i, i, i, i, i]
    L.T360_version.restype = C.c_char_p
    L.T360_deviceCount.restype = i
    L.T360_setStream.argtypes = [vp, vp]
    L.T360_synchronize.argtypes = [vp]
    L.T360_useOwnStream.argtypes = [vp]
    L.T360_transformFrames.argtypes = [vp, u8p, C.c_int64, u8p, C.c_int64, i, C.POINTER(T360PlaneDesc), i]
    L.T360_filterPlane.argtypes = [vp, u8p, u8p, i, i, i, i, i]
    L.T360_transformFramesPipelined.argtypes = [vp, u8p, C.c_int64, u8p, C.c_int64, i, C.POINTER(T360PlaneDesc), i]
    L.T360_transformFramesPipelinedMany.argtypes = [vp, i, C.POINTER(C.c_void_p), C.c_int64, C.POINTER(C.c_void_p), C.c_int64, i,
                                                    C.POINTER(T360PlaneDesc), i]
    L.T360_setPipelineDepth.argtypes = [vp, i]
    L.T360_setFusedLowpass.argtypes = [vp, i]
    L.T360_pipelineJoin.argtypes = [vp]
    L.T360_getMapSize.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
    L.T360_copyMap.argtypes = [vp, i, vp]
    L.T360_getSegmentCount.argtypes = [vp, i]
    L.T360_getSegment.argtypes = [vp, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.T360_copySegmentKernels.argtypes = [vp, i, i, vp, vp]
    L.T360_fillNoise.argtypes = [u8p, C.c_int64, C.c_uint64, vp]
    L.T360_lastKernel.argtypes = [vp]
    L.T360_getPlanStats.argtypes = [vp, i, C.POINTER(C.c_int64)]
    L.T360_buildFlags.argtypes = []
    for name in ADDITIVE_SYMBOLS[1:]:
        getattr(L, name).restype = i
    L.T360_lastKernel.restype = C.c_char_p
    L.T360_lastLowpassPath.argtypes = [vp]
    L.T360_lastLowpassPath.restype = C.c_char_p
    _lib = L
    return L
