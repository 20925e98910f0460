"""Python mirror of the reference's handler interface over the C ABI of libTransform360.so.

``VideoFrameTransform`` has the reference's method names and argument meaning
(reference Transform360/Library/VideoFrameTransform.h:40-72,
Transform360/Library/VideoFrameTransformHandler.h:24-47); every call goes through the exported
C symbols, so these tests exercise exactly what the ffmpeg filter would link against.
PyTorch is used only as plumbing: device memory (tensors), streams, torch.distributed.
"""
import ctypes as C

import numpy as np

from . import _lib
from .abi import FrameTransformContext, chroma_dims


def _is_tensor(x):
    return type(x).__module__.startswith("torch")


def _plane_ptr(a):
    """(pointer, width, height, row stride in bytes) of a 2-D uint8 numpy array / torch tensor."""
    if _is_tensor(a):
        import torch
        assert a.dtype == torch.uint8 and a.dim() == 2 and a.stride(1) == 1
        return a.data_ptr(), a.shape[1], a.shape[0], a.stride(0)
    assert a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1
    return a.ctypes.data, a.shape[1], a.shape[0], a.strides[0]


class FrameLayout:
    """Planes of one 8-bit planar frame inside a flat buffer (yuv420p by default): luma at a
    64-byte aligned stride, then U, then V, each plane starting 256-byte aligned."""

    def __init__(self, width, height, log2_chroma_w=1, log2_chroma_h=1, planes=3, align=64, extra_pad=0):
        cw, ch = chroma_dims(width, height, log2_chroma_w, log2_chroma_h)
        self.width, self.height = width, height
        self.dims = [(width, height)] + [(cw, ch)] * (planes - 1)
        self.strides = [((w + align - 1) // align) * align + extra_pad for (w, _) in self.dims]
        self.offsets = []
        off = 0
        for (w, h), s in zip(self.dims, self.strides):
            self.offsets.append(off)
            off += ((s * h + 255) // 256) * 256
        self.frame_bytes = off

    def payload_bytes(self):
        return sum(w * h for (w, h) in self.dims)

    def plane_view(self, frame, k):
        """2-D view (numpy or torch) of plane k inside a flat uint8 frame buffer."""
        (w, h), s, o = self.dims[k], self.strides[k], self.offsets[k]
        flat = frame[o:o + s * h]
        if _is_tensor(flat):
            return flat.view(h, s)[:, :w]
        return flat.reshape(h, s)[:, :w]


class VideoFrameTransform:
    """Handle with the reference's call protocol: construct from a FrameTransformContext,
    ``generateMapForPlane`` for map index 0 (luma shape) and 1 (chroma shape), then
    ``transformFramePlane`` per plane per frame (vf_transform360.c:141-162, 368-397)."""

    def __init__(self, ctx):
        self._l = _lib.load()
        assert isinstance(ctx, FrameTransformContext)
        self.ctx = ctx
        self._h = self._l.VideoFrameTransform_new(C.byref(ctx))
        if not self._h:
            raise RuntimeError("VideoFrameTransform_new returned NULL (no usable HIP device?)")

    def close(self):
        if getattr(self, "_h", None):
            self._l.VideoFrameTransform_delete(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- reference surface ----
    def generateMapForPlane(self, inputWidth, inputHeight, outputWidth, outputHeight, transformMatPlaneIndex):
        return bool(self._l.VideoFrameTransform_generateMapForPlane(
            self._h, inputWidth, inputHeight, outputWidth, outputHeight, transformMatPlaneIndex))

    def transformFramePlane(self, inputPlane, outputPlane, transformMatPlaneIndex, imagePlaneIndex=0):
        """inputPlane / outputPlane: 2-D uint8 numpy arrays (host pointers, staged over PCIe) or
        CUDA tensors (device pointers, used in place); row strides are taken from the views."""
        ip, iw, ih, istride = _plane_ptr(inputPlane)
        op, ow, oh, ostride = _plane_ptr(outputPlane)
        return bool(self._l.VideoFrameTransform_transformFramePlane(
            self._h, ip, op, iw, ih, istride, ow, oh, ostride, transformMatPlaneIndex, imagePlaneIndex))

    # ---- additive surface (include/Transform360/t360_device.h) ----
    def setStream(self, stream):
        """stream: torch.cuda.Stream or raw hipStream_t integer (0 / None = HIP's NULL stream,
        which is what torch's default current stream is)."""
        raw = getattr(stream, "cuda_stream", stream)
        return bool(self._l.T360_setStream(self._h, raw or None))

    def useOwnStream(self):
        return bool(self._l.T360_useOwnStream(self._h))

    def synchronize(self):
        return bool(self._l.T360_synchronize(self._h))

    def plane_descs(self, in_layout, out_layout):
        """T360PlaneDesc array for frames laid out by two FrameLayouts (plane k>0 uses map 1,
        reference vf_transform360.c:372)."""
        n = len(in_layout.dims)
        arr = (_lib.T360PlaneDesc * n)()
        for k in range(n):
            arr[k] = _lib.T360PlaneDesc(
                in_offset=in_layout.offsets[k], out_offset=out_layout.offsets[k],
                in_stride=in_layout.strides[k], out_stride=out_layout.strides[k],
                in_width=in_layout.dims[k][0], in_height=in_layout.dims[k][1],
                out_width=out_layout.dims[k][0], out_height=out_layout.dims[k][1],
                map_index=1 if k in (1, 2) else 0)
        return arr

    def transformFrames(self, d_in, in_frame_bytes, d_out, out_frame_bytes, n_frames, descs):
        """d_in / d_out: flat uint8 CUDA tensors holding n_frames frames back to back."""
        return bool(self._l.T360_transformFrames(
            self._h, d_in.data_ptr(), in_frame_bytes, d_out.data_ptr(), out_frame_bytes, n_frames,
            descs, len(descs)))

    def transformFramesPipelined(self, d_in, in_frame_bytes, d_out, out_frame_bytes, n_frames, descs):
        """T360_transformFramesPipelined: the same work for a stream of independent batches, round-robin over the handle's
        internal streams; nothing is complete before pipelineJoin() (device-side) or synchronize() (host-side)."""
        return bool(self._l.T360_transformFramesPipelined(
            self._h, d_in.data_ptr(), in_frame_bytes, d_out.data_ptr(), out_frame_bytes, n_frames,
            descs, len(descs)))

    def transformFramesPipelinedMany(self, d_ins, in_frame_bytes, d_outs, out_frame_bytes, n_frames, descs):
        """len(d_ins) pipelined calls issued by one native loop (T360_transformFramesPipelinedMany); d_ins / d_outs are lists
        of tensors or raw device addresses."""
        n = len(d_ins)
        assert n == len(d_outs)
        ptr = lambda x: x.data_ptr() if hasattr(x, "data_ptr") else int(x)
        ins = (C.c_void_p * n)(*[ptr(x) for x in d_ins])
        outs = (C.c_void_p * n)(*[ptr(x) for x in d_outs])
        return bool(self._l.T360_transformFramesPipelinedMany(self._h, n, ins, in_frame_bytes, outs, out_frame_bytes, n_frames,
                                                              descs, len(descs)))

    def setFusedLowpass(self, on):
        """before generateMapForPlane: long batches of a low-pass context filter inside the gather tiles (t360_device.h)"""
        return bool(self._l.T360_setFusedLowpass(self._h, 1 if on else 0))

    def setPipelineDepth(self, depth):
        return bool(self._l.T360_setPipelineDepth(self._h, depth))

    def pipelineJoin(self):
        return bool(self._l.T360_pipelineJoin(self._h))

    def filterPlane(self, d_in, d_out, map_index):
        ip, iw, ih, istride = _plane_ptr(d_in)
        op, _, _, ostride = _plane_ptr(d_out)
        return bool(self._l.T360_filterPlane(self._h, ip, op, iw, ih, istride, ostride, map_index))

    def lastKernel(self):
        """Name of the gather kernel the most recent transform call launched."""
        return (self._l.T360_lastKernel(self._h) or b"").decode()

    def lastLowpassPath(self):
        return (self._l.T360_lastLowpassPath(self._h) or b"").decode()

    def planStats(self, map_index):
        """dict of the gather plan of `map_index`, or None when the plane uses the general gather."""
        st = (C.c_int64 * 8)()
        if not self._l.T360_getPlanStats(self._h, map_index, st):
            return None
        return dict(staged_tiles=st[0], direct_tiles=st[1], fetched_bytes=st[2], lds_bytes=st[3],
                    direct_pixels=st[4], table_bytes=st[5])

    def map(self, map_index):
        """The float warp map of `map_index` as an [h, w, 2] float32 array (warpMats_[idx])."""
        w, h = C.c_int(), C.c_int()
        if not self._l.T360_getMapSize(self._h, map_index, C.byref(w), C.byref(h)):
            return None
        m = np.empty((h.value, w.value, 2), np.float32)
        if not self._l.T360_copyMap(self._h, map_index, m.ctypes.data):
            return None
        return m

    def segments(self, map_index):
        """[(left, top, width, height, kx, ky, fixed_point)] like the oracle's accessor."""
        out = []
        for i in range(self._l.T360_getSegmentCount(self._h, map_index)):
            rect, lens, fx = (C.c_int * 4)(), (C.c_int * 2)(), C.c_int()
            self._l.T360_getSegment(self._h, map_index, i, rect, lens, C.byref(fx))
            kx, ky = np.empty(lens[0], np.float32), np.empty(lens[1], np.float32)
            self._l.T360_copySegmentKernels(self._h, map_index, i, kx.ctypes.data, ky.ctypes.data)
            out.append((rect[0], rect[1], rect[2], rect[3], kx, ky, bool(fx.value)))
        return out


# ---- synthetic stream: byte i of a buffer = splitmix64(seed + i) >> 56 ----

_MASK = (1 << 64) - 1


def noise_bytes(nbytes, seed):
    """Host evaluation of the generator T360_fillNoise runs on the device."""
    x = (np.arange(nbytes, dtype=np.uint64) + np.uint64(seed & _MASK))
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(56)).astype(np.uint8)


def fill_noise(tensor, seed, stream=None):
    """Fill a flat uint8 CUDA tensor with the synthetic noise on `stream` (default: current)."""
    import torch
    L = _lib.load()
    s = stream if stream is not None else torch.cuda.current_stream()
    raw = getattr(s, "cuda_stream", s)
    if not L.T360_fillNoise(tensor.data_ptr(), tensor.numel(), seed & _MASK, raw or None):
        raise RuntimeError("T360_fillNoise failed")


def frame_seed(k, base=0x360):
    """Seed of frame k of the synthetic stream (distinct 2^40-byte windows per frame)."""
    return (base ^ (k << 40)) & _MASK
