"""CPU suite for the host side: the C-ABI library loads and exports every symbol the headers
declare (no compute without a GPU), the headers are valid C with the reference's struct layout,
the filter's size logic, the synthetic stream generator, and the multi-rank frame sharding
(world_size 2 over gloo with the oracle standing in for the per-frame transform)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from transform360_amd import _lib, abi, sharding
from transform360_amd.handler import FrameLayout, frame_seed, noise_bytes

INCLUDE = os.path.join(ROOT, "include")


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.LIB_PATH


def declared_symbols():
    names = []
    for h in ("VideoFrameTransformHandler.h", "t360_device.h"):
        src = open(os.path.join(INCLUDE, "Transform360", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b((?:VideoFrameTransform|T360)_\w+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    decl = declared_symbols()
    assert set(_lib.REFERENCE_SYMBOLS) <= set(decl)
    missing = [s for s in decl if s not in exported]
    assert not missing, "declared in include/ but not exported: %s" % missing
    # python binding list and headers agree
    assert sorted(_lib.REFERENCE_SYMBOLS + _lib.ADDITIVE_SYMBOLS) == decl


def test_library_loads_and_fails_loudly_without_gpu(built_lib):
    L = ctypes.CDLL(built_lib)
    L.T360_version.restype = ctypes.c_char_p
    assert b"transform360" in L.T360_version()
    L.VideoFrameTransform_delete(None)           # delete(NULL) is a no-op (vf_transform360.c:334)
    L.VideoFrameTransform_new.restype = ctypes.c_void_p
    assert L.VideoFrameTransform_new(None) is None
    from tests.conftest import HAS_GPU
    if not HAS_GPU:
        # no CPU fallback: without a device the handle cannot be created at all
        ctx = abi.filter_defaults()
        assert L.VideoFrameTransform_new(ctypes.byref(ctx)) is None


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under transform360_amd/ may reference it."""
    pkg = os.path.join(ROOT, "transform360_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "t360_oracle" not in text and "from oracle" not in text and "import oracle" not in text, \
                    "%s references the oracle" % os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", _lib.LIB_PATH]).decode() if os.path.exists(_lib.LIB_PATH) else ""
    assert "oracle" not in out


def test_headers_compile_as_c_with_reference_layout(tmp_path):
    """include/ is plain C; FrameTransformContext is 112 bytes with the reference's field order
    (reference VideoFrameTransformHelper.h:56-90) and enum values (:18-54)."""
    src = tmp_path / "abi_check.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "Transform360/VideoFrameTransformHandler.h"
#include "Transform360/t360_device.h"
int main(void) {
  FrameTransformContext c;
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof c, offsetof(FrameTransformContext, interpolation_alg),
         offsetof(FrameTransformContext, fixed_cube_offcenter_z), offsetof(FrameTransformContext, enable_low_pass_filter),
         offsetof(FrameTransformContext, kernel_adjust_factor), sizeof(T360PlaneDesc));
  printf("%d %d %d %d %d %d %d %d\n", LAYOUT_CUBEMAP_32, LAYOUT_CUBEMAP_23_OFFCENTER, LAYOUT_FLAT_FIXED, LAYOUT_EQUIRECT,
         LAYOUT_BARREL, LAYOUT_BARREL_SPLIT, LAYOUT_EAC_32, LAYOUT_N);
  printf("%d %d %d %d %d %d %d %d\n", STEREO_FORMAT_TB, STEREO_FORMAT_LR, STEREO_FORMAT_MONO, STEREO_FORMAT_GUESS,
         NEAREST, LINEAR, CUBIC, LANCZOS4);
  printf("%d %d %d %d %d %d\n", RIGHT, LEFT, TOP, BOTTOM, FRONT, BACK);
  /* the filter's call sequence type-checks against the prototypes */
  VideoFrameTransform* (*fn_new)(FrameTransformContext*) = VideoFrameTransform_new;
  int (*fn_map)(VideoFrameTransform*, int, int, int, int, int) = VideoFrameTransform_generateMapForPlane;
  int (*fn_tx)(VideoFrameTransform*, uint8_t*, uint8_t*, int, int, int, int, int, int, int, int) =
      VideoFrameTransform_transformFramePlane;
  void (*fn_del)(VideoFrameTransform*) = VideoFrameTransform_delete;
  return !(fn_new && fn_map && fn_tx && fn_del);
}
''')
    exe = tmp_path / "abi_check"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", INCLUDE, str(src), "-o", str(exe),
                           "-L", os.path.dirname(_lib.LIB_PATH), "-lTransform360",
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    assert lines[0].split() == ["112", "28", "68", "76", "108", "48"]
    assert lines[1].split() == ["0", "1", "2", "3", "4", "5", "6", "7"]
    assert lines[2].split() == ["0", "1", "2", "3", "0", "1", "2", "4"]
    assert lines[3].split() == ["0", "1", "2", "3", "4", "5"]


def test_ctypes_mirror_matches_header():
    assert ctypes.sizeof(abi.FrameTransformContext) == 112
    assert abi.FrameTransformContext.interpolation_alg.offset == 28
    assert abi.FrameTransformContext.kernel_adjust_factor.offset == 108
    assert ctypes.sizeof(_lib.T360PlaneDesc) == 48
    d = abi.filter_defaults()
    # ffmpeg option defaults, vf_transform360.c:407-987
    assert (d.input_layout, d.output_layout) == (abi.LAYOUT_EQUIRECT, abi.LAYOUT_CUBEMAP_32)
    assert d.interpolation_alg == abi.CUBIC and d.enable_low_pass_filter == 1
    assert (d.num_vertical_segments, d.num_horizontal_segments, d.adjust_kernel) == (5, 1, 1)
    assert abs(d.expand_coef - 1.01) < 1e-6 and d.max_kernel_half_height == 10000.0


def test_config_output_rules():
    """vf_transform360.c:198-223, 293-299 and the GUESS rules :178-196."""
    assert abi.config_output(3840, 1920, 512) == (1536, 1024)
    assert abi.config_output(1920, 960, 256) == (768, 512)
    assert abi.config_output(3840, 1920, 520) == (1536, 1024)        # rounded down to a multiple of 16
    assert abi.config_output(7680, 3840, 1024, output_stereo_format=abi.STEREO_FORMAT_TB) == (3072, 4096)
    assert abi.config_output(3840, 1920, 512, output_layout=abi.LAYOUT_CUBEMAP_23_OFFCENTER,
                             output_stereo_format=abi.STEREO_FORMAT_LR) == (2048, 1536)
    assert abi.config_output(3840, 1920, 0, max_cube_edge_length=600) == (1776, 1184)   # 3840/4=960 -> 600 -> 592
    assert abi.guess_stereo(3840, 3840, abi.STEREO_FORMAT_GUESS, abi.STEREO_FORMAT_GUESS, abi.LAYOUT_CUBEMAP_32) == \
        (abi.STEREO_FORMAT_TB, abi.STEREO_FORMAT_TB)
    assert abi.guess_stereo(7680, 1920, abi.STEREO_FORMAT_GUESS, abi.STEREO_FORMAT_GUESS,
                            abi.LAYOUT_CUBEMAP_23_OFFCENTER) == (abi.STEREO_FORMAT_LR, abi.STEREO_FORMAT_LR)
    assert abi.guess_stereo(3840, 1920, abi.STEREO_FORMAT_GUESS, abi.STEREO_FORMAT_GUESS, 0)[0] == abi.STEREO_FORMAT_MONO
    assert abi.chroma_dims(1001, 499) == (501, 250)                  # FF_CEIL_RSHIFT


def test_frame_layout_and_noise():
    lay = FrameLayout(3840, 1920)
    assert lay.dims == [(3840, 1920), (1920, 960), (1920, 960)]
    assert lay.payload_bytes() == 11059200
    assert all(o % 256 == 0 for o in lay.offsets) and all(s % 64 == 0 for s in lay.strides)
    out = FrameLayout(1536, 1024)
    assert lay.payload_bytes() + out.payload_bytes() == 13418496     # algorithmic bytes per frame, config 2
    a = noise_bytes(1000, frame_seed(3))
    b = noise_bytes(1000, frame_seed(3))
    assert np.array_equal(a, b) and not np.array_equal(a, noise_bytes(1000, frame_seed(4)))
    assert abs(float(a.mean()) - 127.5) < 12
    buf = np.arange(lay.frame_bytes, dtype=np.uint32).astype(np.uint8)
    v = lay.plane_view(buf, 1)
    assert v.shape == (960, 1920) and v.strides == (lay.strides[1], 1)


def test_shard_range_partitions_the_stream():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = sharding.shard_range(n, r, w)
                seen += list(range(lo, hi))
                assert all(sharding.owner_of(k, n, w) == r for k in range(lo, hi))
            assert seen == list(range(n))
    assert sharding.shard_range(64, 3, 8) == (24, 32)                # config 5: 64 frames over 8 GPUs


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from oracle import t360_oracle as O
from transform360_amd import sharding
from transform360_amd.abi import filter_defaults, NEAREST, CUBIC
from transform360_amd.handler import FrameLayout, frame_seed, noise_bytes

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# rank 1 starts with a WRONG context: the broadcast must overwrite it
ctx = filter_defaults(interpolation_alg=CUBIC if rank == 0 else NEAREST, enable_low_pass_filter=0)
ctx = sharding.broadcast_context(ctx, dist)
assert ctx.interpolation_alg == CUBIC
lin, lout = FrameLayout(256, 128), FrameLayout(96, 64)
o = O.Oracle(ctx)
for idx, k in ((0, 0), (1, 1)):
    assert o.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)

def make_frame(k):
    return noise_bytes(lin.frame_bytes, frame_seed(k))

def transform(frame):
    outs = []
    for p in range(3):
        dst = np.zeros((lout.dims[p][1], lout.dims[p][0]), np.uint8)
        assert o.transformFramePlane(lin.plane_view(frame, p), dst, 1 if p else 0, p)
        outs.append(dst)
    return outs

n = 7
sums = sharding.run_sharded(n, make_frame, transform, dist)
lo, hi = sharding.shard_range(n, rank, world)
with open(os.path.join(%(out)r, "result_%%d.txt" %% rank), "w") as f:   # stdout of the ranks interleaves
    f.write("%%d %%d %%d %%s\n" %% (rank, lo, hi, " ".join(str(s) for s in sums)))
dist.destroy_process_group()
'''


def test_frame_sharding_world_size_2_gloo(tmp_path, oracle_mod):
    """Two ranks over gloo: context broadcast, contiguous frame shards, checksum all_gather;
    the sharded stream equals the unsharded one frame for frame."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29731", str(script)], env=env, stderr=subprocess.STDOUT, timeout=300)
    by_rank = {}
    for r in (0, 1):
        fields = (tmp_path / ("result_%d.txt" % r)).read_text().split()
        by_rank[int(fields[0])] = ["RESULT"] + fields
    assert (int(by_rank[0][2]), int(by_rank[0][3])) == (0, 4) and (int(by_rank[1][2]), int(by_rank[1][3])) == (4, 7)
    assert by_rank[0][4:] == by_rank[1][4:]                          # every rank sees the whole stream
    # unsharded run in this process
    from transform360_amd.abi import CUBIC, filter_defaults
    O = oracle_mod
    ctx = filter_defaults(interpolation_alg=CUBIC, enable_low_pass_filter=0)
    lin, lout = FrameLayout(256, 128), FrameLayout(96, 64)
    o = O.Oracle(ctx)
    for idx, k in ((0, 0), (1, 1)):
        assert o.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)

    def transform(frame):
        outs = []
        for p in range(3):
            dst = np.zeros((lout.dims[p][1], lout.dims[p][0]), np.uint8)
            assert o.transformFramePlane(lin.plane_view(frame, p), dst, 1 if p else 0, p)
            outs.append(dst)
        return outs

    want = sharding.run_sharded(7, lambda k: noise_bytes(lin.frame_bytes, frame_seed(k)), transform)
    assert [str(s) for s in want] == by_rank[0][4:]
    assert len(set(want)) == 7                                        # distinct frames, distinct sums


def test_bench_rank_function_world_size_2_gloo():
    """bench.py's run_rank() -- what every rank of the 8-GPU run executes: context broadcast, frame shards, barriers,
    MAX-reduced timing, the strong record, the overlapped output gather, the checksum all_gather -- under
    torch.distributed.run with two gloo ranks and a CPU stand-in for the transform (`--stub`)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", T360_DIST_BACKEND="gloo")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--frames", "6", "--steps", "3",
         "--warmup", "1", "--gather-outputs", "--scatter-inputs"], env=env, stderr=subprocess.DEVNULL, timeout=300, cwd=ROOT).decode()
    rec = json.loads([line for line in out.splitlines() if line.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["metric"].startswith("STUB")               # a rehearsal can never be mistaken for a result
    assert rec["verified"]["all_ranks_ok"] and rec["verified"]["max_abs_diff"] == 0
    assert len(rec["output_checksums"]) == 2 and rec["output_checksums"][0] != rec["output_checksums"][1]
    s = rec["strong_cfg5"]
    assert s["n_gpus"] == 2 and s["frames_per_gpu"] == 6 and "note" in s   # 64 / 2 = 32 > --frames 6: said so
    assert rec["input_ring"]["groups_of_F_frames"] == 2   # the timed steps rotated through two input batches ...
    assert rec["verified"]["frames"]                        # ... and the verified result is group 0's again
    g = rec["gather_outputs"]
    assert g["bytes_to_rank0_per_step"] > 0 and g["ms_per_step"] > 0 and "gloo" in g["collective"]
    x = rec["scatter_gather"]
    assert x["bytes_from_rank0_per_step"] > x["bytes_to_rank0_per_step"] > 0 and x["ms_per_step"] > 0


def test_bench_headline_is_strong_scaling_with_two_ranks():
    """With N > 1 ranks and the default 64 frames per step the line's `value` is BASELINE configs[4] as written -- 64 frames per
    step in total, 32 per rank here, "scaling": "strong" -- and the weak figure is a sub-record."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", T360_DIST_BACKEND="gloo")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29743", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "2",
         "--warmup", "1"], env=env, stderr=subprocess.DEVNULL, timeout=300, cwd=ROOT).decode()
    rec = json.loads([line for line in out.splitlines() if line.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["frames_per_step_total"] == 64 and rec["config"]["frames_per_step_per_gpu"] == 32
    s, w = rec["strong_cfg5"], rec["weak_64_frames_per_gpu"]
    assert rec["value"] == s["value"] == rec["strong_cfg5_value"] and rec["ms_per_step"] == s["ms_per_step"]
    assert w["scaling"] == "weak" and w["frames_per_step_per_gpu"] == 64 and rec["weak_value"] == w["value"]
    assert rec["roofline"]["algorithmic_bytes_per_launch"] > 0 and rec["roofline"]["avg_launch_ms"] > 0


def test_native_multi_gpu_driver_bookkeeping(tmp_path):
    """examples/t360_shard_plan.h -- the frame ranges, the per-step send / recv lists and the buffer alternation of the
    native multi-GPU driver (examples/t360_multi_gpu.cpp) -- compiled with g++ and checked without HIP or RCCL
    (tests/c/shard_plan_test.cpp): every frame owned once, every send met by one receive of its size, the sink gap-free."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "shard_plan_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "c", "shard_plan_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "shard plan ok" in out.stdout, out.stdout + out.stderr


def test_bench_states_the_cores_it_may_use():
    import bench
    a = bench.host_cpu_allowance()
    assert 1 <= a["usable_cores"] <= a["logical_cores_of_the_node"]
    assert a["sched_affinity_cores"] is None or a["usable_cores"] <= a["sched_affinity_cores"]
