"""Frame sharding of a synthetic stream across ranks (one process per GPU).

The remap path shards naturally: frames are independent and the handle's state is read-only after
init (reference VideoFrameTransform.cpp:707-794 touches only per-call buffers).
Frame k of a step goes to rank ``k // frames_per_rank`` (contiguous blocks), every rank rebuilds
its maps from the 112-byte context, and there is NO data-path collective.  Collectives are used
only around the path:

  * ``broadcast_context``  rank 0's FrameTransformContext -> all ranks (RCCL on GPUs, gloo on CPU)
  * ``gather_checksums``   per-frame output checksums -> rank 0, for verification

Backend-agnostic: the world_size-2 CPU tests run this module over ``gloo`` with the oracle as the
per-frame transform; bench.py runs it over ``nccl`` (= RCCL on ROCm) with the HIP path.
"""
import os

import numpy as np

from .abi import FrameTransformContext


def shard_range(n_frames, rank, world_size):
    """Contiguous block of frame indices owned by `rank`: [lo, hi).  The first
    ``n_frames % world_size`` ranks get one frame more."""
    base, extra = divmod(n_frames, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(frame, n_frames, world_size):
    """Rank that owns `frame` under shard_range."""
    base, extra = divmod(n_frames, world_size)
    edge = extra * (base + 1)
    if frame < edge:
        return frame // (base + 1)
    return extra + (frame - edge) // base if base else world_size - 1


def broadcast_context(ctx, dist=None, device=None, src=0):
    """Every rank ends up with rank `src`'s context (bitwise)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not os.environ.get("T360_FORCE_DIST")):
        return ctx
    import torch
    buf = torch.frombuffer(bytearray(bytes(ctx)), dtype=torch.uint8)
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=src)
    return FrameTransformContext.from_buffer_copy(bytes(buf.cpu().numpy()))


def frame_checksum(planes):
    """Order-sensitive 64-bit checksum of the output planes of one frame (FNV-1a over rows)."""
    h = np.uint64(1469598103934665603)
    prime = np.uint64(1099511628211)
    with np.errstate(over="ignore"):
        for p in planes:
            a = np.ascontiguousarray(p).reshape(-1)
            # fold 8 bytes at a time: fast and still position-sensitive
            pad = (-a.size) % 8
            if pad:
                a = np.concatenate([a, np.zeros(pad, np.uint8)])
            for w in a.view(np.uint64)[:: max(1, a.size // 8 // 4096)]:
                h = (h ^ w) * prime
            h = (h ^ np.uint64(int(a.view(np.uint64).sum(dtype=np.uint64)))) * prime
    return int(h)


def gather_checksums(local, n_frames, dist=None, device=None):
    """local: {frame index: checksum} of this rank.  Returns the full list on every rank
    (all_gather of a fixed-size int64 vector; unowned slots are zero and are summed away)."""
    vec = np.zeros(n_frames, np.int64)
    for k, v in local.items():
        vec[k] = np.int64(np.uint64(v).astype(np.int64))
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not os.environ.get("T360_FORCE_DIST")):
        return [int(np.uint64(np.int64(v))) for v in vec]
    import torch
    t = torch.from_numpy(vec)
    if device is not None:
        t = t.to(device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    total = torch.stack(parts).sum(dim=0).cpu().numpy()
    return [int(np.uint64(np.int64(v))) for v in total]


def run_sharded(n_frames, make_frame, transform_frame, dist=None, device=None):
    """Generic driver: every rank transforms the frames it owns.

    make_frame(k)       -> input frame k (any object the transform understands)
    transform_frame(f)  -> list of output planes (numpy arrays) of that frame
    Returns the per-frame checksums of the WHOLE stream on every rank."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    lo, hi = shard_range(n_frames, rank, world)
    local = {}
    for k in range(lo, hi):
        local[k] = frame_checksum(transform_frame(make_frame(k)))
    return gather_checksums(local, n_frames, dist, device)
