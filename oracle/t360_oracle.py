"""ctypes bindings for the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under transform360_amd/ does.  Two libraries are bound:

  oracle/libt360oracle.so      the C restatement (oracle/t360_oracle*.c), class ``Oracle``
  oracle/_ref/libt360ref.so    the REFERENCE's own VideoFrameTransform.cpp compiled against the
                               test-only cv::Mat shim (oracle/ref_probe.cpp), class ``Ref``;
                               built only where /root/reference is mounted, the prebuilt file
                               travels to the GPU box.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from transform360_amd.abi import FrameTransformContext

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libt360oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libt360ref.so")
REFERENCE_DIR = "/root/reference/Transform360/Library"

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def build(ref=True, quiet=True):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", _HERE], stdout=out)
    if ref and os.path.isdir(REFERENCE_DIR):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=out)


class _Seg(C.Structure):
    _fields_ = [("left", C.c_int), ("top", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("kx_len", C.c_int), ("ky_len", C.c_int), ("kx", _f32p), ("ky", _f32p)]


class _Cfg(C.Structure):
    _fields_ = [("count", C.c_int), ("capacity", C.c_int), ("seg", C.POINTER(_Seg))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        L.t360o_new.restype = C.c_void_p
        L.t360o_new.argtypes = [C.POINTER(FrameTransformContext)]
        L.t360o_delete.argtypes = [C.c_void_p]
        L.t360o_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.t360o_generateMapForPlane.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.t360o_transformFramePlane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8
        L.t360o_filterPlane.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int]
        L.t360o_map.restype = _f32p
        L.t360o_map.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        L.t360o_segments.restype = C.POINTER(_Cfg)
        L.t360o_segments.argtypes = [C.c_void_p, C.c_int]
        L.t360o_fnv1a64.restype = C.c_uint64
        L.t360o_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.t360o_inter_tab.restype = C.POINTER(C.c_int16)
        L.t360o_inter_tab.argtypes = [C.c_int, _ip]
        L.t360o_transform_pos.argtypes = [C.POINTER(FrameTransformContext), C.c_float, C.c_float,
                                          _f32p, _f32p, C.c_float]
        L.t360o_kernel_type.argtypes = [_f32p, C.c_int]
        L.t360o_border_interpolate.argtypes = [C.c_int, C.c_int, C.c_int]
        L.t360o_remap_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                       C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.t360o_sepfilter_roi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                          C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.t360o_resize_area.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.t360o_set_cv_variant.argtypes = [C.c_int, C.c_int]
        L.t360o_set_cv_variant.restype = None
        _lib = L
    return _lib


def set_cv_variant(column_simd_lanes=0, area_tail_lanes=0):
    """Which tie-rounding the restatement follows where a SIMD build of OpenCV 4.x and its scalar code differ
    (t360_oracle_cv.c t360o_set_cv_variant); (0, 0) is the default the library is compared with."""
    lib().t360o_set_cv_variant(column_simd_lanes, area_tail_lanes)


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().t360o_fnv1a64(a.ctypes.data, a.nbytes))


def inter_tab(interp):
    """OpenCV's 1024-entry 2-D Q15 table for LINEAR/CUBIC/LANCZOS4 as int16 [1024, k*k]."""
    ks = C.c_int()
    p = lib().t360o_inter_tab(interp, C.byref(ks))
    n = 1024 * ks.value * ks.value
    return np.ctypeslib.as_array(p, shape=(n,)).reshape(1024, ks.value * ks.value).copy()


def quantize_map(m):
    """(sx>>5, sy>>5, frac) int32 triples of Appendix B ('q') and nearest pairs ('nn')."""
    x = m[..., 0].astype(np.float32)
    y = m[..., 1].astype(np.float32)
    sx = np.rint(x * np.float32(32)).astype(np.int32)
    sy = np.rint(y * np.float32(32)).astype(np.int32)
    q = np.stack([sx >> 5, sy >> 5, (sy & 31) * 32 + (sx & 31)], axis=-1).astype(np.int32)
    nn = np.stack([np.rint(x), np.rint(y)], axis=-1).astype(np.int32)
    return q, nn


def _plane_args(a):
    assert a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1
    return a.ctypes.data, a.shape[1], a.shape[0], a.strides[0]


class Oracle:
    """The restatement behind the reference's four-call protocol."""

    def __init__(self, ctx, threads=1):
        self._l = lib()
        self.ctx = ctx
        self._h = self._l.t360o_new(C.byref(ctx))
        if not self._h:
            raise MemoryError("t360o_new")
        self._l.t360o_set_threads(self._h, threads)

    def close(self):
        if self._h:
            self._l.t360o_delete(self._h)
            self._h = None

    __del__ = close

    def set_threads(self, n):
        self._l.t360o_set_threads(self._h, n)

    def generateMapForPlane(self, inW, inH, outW, outH, idx):
        return bool(self._l.t360o_generateMapForPlane(self._h, inW, inH, outW, outH, idx))

    def transformFramePlane(self, src, dst, idx, image_plane=0, out_w=None, out_h=None):
        sp, sw, sh, ss = _plane_args(src)
        dp, dw, dh, ds = _plane_args(dst)
        return bool(self._l.t360o_transformFramePlane(
            self._h, sp, dp, sw, sh, ss, out_w or dw, out_h or dh, ds, idx, image_plane))

    def filterPlane(self, src, idx):
        sp, sw, sh, ss = _plane_args(src)
        dst = np.empty((sh, sw), np.uint8)
        ok = self._l.t360o_filterPlane(self._h, sp, sw, sh, ss, dst.ctypes.data, sw, idx)
        return dst if ok else None

    def map(self, idx):
        w, h = C.c_int(), C.c_int()
        p = self._l.t360o_map(self._h, idx, C.byref(w), C.byref(h))
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(h.value * w.value * 2,)).reshape(h.value, w.value, 2).copy()

    def segments(self, idx):
        """List of (left, top, width, height, kx[float32], ky[float32])."""
        cfg = self._l.t360o_segments(self._h, idx).contents
        out = []
        for i in range(cfg.count):
            s = cfg.seg[i]
            kx = np.ctypeslib.as_array(s.kx, shape=(s.kx_len,)).copy()
            ky = np.ctypeslib.as_array(s.ky, shape=(s.ky_len,)).copy()
            out.append((s.left, s.top, s.width, s.height, kx, ky))
        return out


_ref = None


def ref_available():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.t360ref_new.restype = C.c_void_p
        R.t360ref_new.argtypes = [C.POINTER(FrameTransformContext)]
        R.t360ref_delete.argtypes = [C.c_void_p]
        R.t360ref_generateMapForPlane.argtypes = [C.c_void_p] + [C.c_int] * 5
        R.t360ref_transformFramePlane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8
        R.t360ref_transform_pos.argtypes = [C.c_void_p, C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.c_float]
        R.t360ref_map_size.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        R.t360ref_copy_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        R.t360ref_num_segments.argtypes = [C.c_void_p, C.c_int]
        R.t360ref_segment.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip]
        R.t360ref_copy_kernels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        R.t360ref_filterPlane.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_int]
        _ref = R
    return _ref


class Ref:
    """The reference's own class (oracle/_ref), cv:: arithmetic bound to the oracle."""

    def __init__(self, ctx):
        self._l = ref_lib()
        self._h = self._l.t360ref_new(C.byref(ctx))

    def close(self):
        if self._h:
            self._l.t360ref_delete(self._h)
            self._h = None

    __del__ = close

    def generateMapForPlane(self, inW, inH, outW, outH, idx):
        return bool(self._l.t360ref_generateMapForPlane(self._h, inW, inH, outW, outH, idx))

    def transformFramePlane(self, src, dst, idx, image_plane=0):
        sp, sw, sh, ss = _plane_args(src)
        dp, dw, dh, ds = _plane_args(dst)
        return bool(self._l.t360ref_transformFramePlane(self._h, sp, dp, sw, sh, ss, dw, dh, ds, idx, image_plane))

    def filterPlane(self, src, idx):
        sp, sw, sh, ss = _plane_args(src)
        dst = np.empty((sh, sw), np.uint8)
        self._l.t360ref_filterPlane(self._h, sp, sw, sh, ss, dst.ctypes.data, sw, idx)
        return dst

    def map(self, idx):
        w, h = C.c_int(), C.c_int()
        if not self._l.t360ref_map_size(self._h, idx, C.byref(w), C.byref(h)):
            return None
        m = np.empty((h.value, w.value, 2), np.float32)
        self._l.t360ref_copy_map(self._h, idx, m.ctypes.data)
        return m

    def segments(self, idx):
        out = []
        for i in range(self._l.t360ref_num_segments(self._h, idx)):
            rect = (C.c_int * 4)()
            lens = (C.c_int * 2)()
            self._l.t360ref_segment(self._h, idx, i, rect, lens)
            kx = np.empty(lens[0], np.float32)
            ky = np.empty(lens[1], np.float32)
            self._l.t360ref_copy_kernels(self._h, idx, i, kx.ctypes.data, ky.ctypes.data)
            out.append((rect[0], rect[1], rect[2], rect[3], kx, ky))
        return out
