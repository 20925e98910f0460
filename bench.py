#!/usr/bin/env python3
"""bench.py -- Mpix/s remapped on the BASELINE.json workload, with roofline, verification and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--config 2|3|1|4]

With --gpus N > 1 and no launcher environment the script starts its N ranks itself (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU);
under a launcher (RANK / WORLD_SIZE set) it is one rank.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): a synthetic stream of
3840x1920 8-bit yuv420p equirect frames -> 512-edge CUBEMAP_32 (1536x1024), bicubic, low-pass off.  Frames are
counter-hash noise generated on the device and RESIDENT IN HBM before the timed region starts; one "step" is one
pass of the hot path (all three planes, ONE fused kernel launch) over one batch of F frames (default 64: 708 MB of
input, far larger than the 256 MB Infinity Cache).

Timing: W warm-up steps, then R = 5 repeats of EXACTLY K steps, each repeat bracketed by a barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over the ranks; the MEDIAN repeat is reported (box-to-box
and run-to-run spread on the pool is a few percent).  With the driver's K = 20 that is 6 400 frames per rank.

Multi-GPU: whole frames are sharded across ranks, no data-path collective.  With N > 1 ranks the HEADLINE (`value`,
`ms_per_step`, `roofline`, "scaling": "strong") is BASELINE.json configs[4] as written: 64 frames per step IN TOTAL,
ceil(64/N) per rank, so a 1-2-4-8 GPU series of this line is the strong-scaling curve of configs[4]; the weak figure
(rank r owns frames r*F .. r*F+F-1 of every step, F = 64 per GPU) stays in the line as `weak_value` /
"weak_64_frames_per_gpu".  RCCL is used only around the path, through transform360_amd/sharding.py: broadcast of the
112-byte context from rank 0, all_gather of per-frame output checksums.
--gather-outputs adds SURVEY 8(e)(ii): the same steps with every step's output frames gathered to rank 0
(dist.gather = RCCL over xGMI), overlapped with the next step, next to the compute-only figure; --scatter-inputs adds
8(e)(iii): the inputs of the next step scattered from rank 0 as well.
Everything a rank does is run_rank(); `--stub` runs that same function with a CPU stand-in for the transform under
gloo (tests/test_host_cpu.py: the code the 8-GPU run executes is the code the CPU test covers).

Extra records of the default line (one GPU, config 2): "strong_cfg5.projected_8_gpus" = what one GPU of an 8-GPU node runs for
configs[4] (8 frames per step) timed on this GPU, and "two_streams" = the same steps alternating between two handles on two
HIP streams (independent batches; the tail of one launch overlaps the start of the next).  Neither enters `value`;
--no-two-streams leaves the overlapped launches out (kernel traces).

Rank 0 prints ONE JSON line:
  value        = frames/s (whole job) x output luma pixels / 1e6                              [Mpix/s]
  roofline     = algorithmic bytes of the step's kernel launch / its average duration (HIP events on the launch
                 stream inside the timed region) vs the 8 TB/s HBM peak; the kernel name is what the library
                 reports it launched; "traffic" = L2-miss (fabric) bytes per launch -- what the XCDs' L2s request from the memory side, Infinity-Cache
                 hits included: an upper bound of the HBM bytes -- from the committed PMC profile of THIS
                 library build (profiles/r*_traffic.json, sha256 of the .so checked), null with the reason otherwise
  verified     = frames of the LAST timed step's output compared, all planes, with the CPU oracle
  cpu_baseline = the CPU oracle (restatement of the reference's OpenCV path, NOT linked OpenCV) with the
                 reference's threading structure, timed on this host on a bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_BPS = 8.0e12  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
REPEATS = 5
RING_BYTES = 1400 << 20    # distinct input bytes the timed steps cycle through: several times the 256 MB Infinity Cache, and at
                           # least two batches (64 frames of config 2 are 708 MB: reading the SAME 708 MB every step measured 1 %
                           # faster than a ring of two or three batches, and let overlapping launches share lines: round 5, call 5)
CLOCK_WARMUP_S = 0.4   # untimed load before the contract's warm-up steps (GPU clocks ramp)


def workload(config):
    from transform360_amd.abi import (CUBIC, LANCZOS4, NEAREST, STEREO_FORMAT_TB)
    if config == 1:
        return dict(name="cfg1: 1920x960 yuv420p -> 256-edge cubemap, nearest, low-pass off",
                    in_w=1920, in_h=960, edge=256, ov=dict(interpolation_alg=NEAREST, enable_low_pass_filter=0))
    if config == 2:
        return dict(name="cfg2: 3840x1920 yuv420p MONO -> 512-edge cubemap (1536x1024), bicubic, low-pass off",
                    in_w=3840, in_h=1920, edge=512, ov=dict(interpolation_alg=CUBIC, enable_low_pass_filter=0))
    if config == 3:
        return dict(name="cfg3: 3840x1920 yuv420p MONO -> 512-edge cubemap, bicubic + segmented low-pass 32x15",
                    in_w=3840, in_h=1920, edge=512,
                    ov=dict(interpolation_alg=CUBIC, enable_low_pass_filter=1, num_horizontal_segments=32,
                            num_vertical_segments=15, adjust_kernel=1, enable_multi_threading=1))
    if config == 4:
        return dict(name="cfg4: 7680x3840 yuv420p TOP_BOTTOM -> 1024-edge cubemap TB (3072x4096), Lanczos4",
                    in_w=7680, in_h=3840, edge=1024,
                    ov=dict(interpolation_alg=LANCZOS4, enable_low_pass_filter=0,
                            input_stereo_format=STEREO_FORMAT_TB, output_stereo_format=STEREO_FORMAT_TB))
    raise SystemExit("unknown --config %r" % config)


def host_cpu_allowance():
    """What this process may actually use of the host: the scheduler affinity mask and the cgroup CPU quota (v2 cpu.max,
    v1 cfs_quota_us / cfs_period_us, of this process's own cgroup where /proc/self/cgroup names one).  os.cpu_count() is the
    node's logical cores, which a leased box usually does not get."""
    out = {"logical_cores_of_the_node": os.cpu_count() or 1}
    try:
        out["sched_affinity_cores"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        out["sched_affinity_cores"] = None
    quota = None
    paths = ["/sys/fs/cgroup"]
    try:
        with open("/proc/self/cgroup") as f:
            for line in f:
                parts = line.strip().split(":", 2)
                if len(parts) == 3 and parts[2] not in ("", "/"):
                    if parts[1] == "":
                        paths.insert(0, "/sys/fs/cgroup" + parts[2])
                    elif "cpu" in parts[1].split(","):
                        paths.insert(0, "/sys/fs/cgroup/cpu" + parts[2])
    except OSError:
        pass
    for base in paths + ["/sys/fs/cgroup/cpu"]:
        try:
            with open(os.path.join(base, "cpu.max")) as f:
                q, per = f.read().split()[:2]
                out["cgroup_cpu_max"] = "%s %s" % (q, per)
                if q != "max":
                    quota = float(q) / float(per)
                break
        except (OSError, ValueError):
            pass
        try:
            with open(os.path.join(base, "cpu.cfs_quota_us")) as f:
                q = int(f.read())
            with open(os.path.join(base, "cpu.cfs_period_us")) as f:
                per = int(f.read())
            out["cgroup_cfs_quota_us/period_us"] = "%d/%d" % (q, per)
            if q > 0 and per > 0:
                quota = q / per
            break
        except (OSError, ValueError):
            pass
    out["cgroup_cpu_quota_cores"] = round(quota, 2) if quota is not None else None
    usable = out["sched_affinity_cores"] or out["logical_cores_of_the_node"]
    if quota is not None:
        usable = min(usable, max(1, int(quota + 0.5)))
    out["usable_cores"] = usable
    return out


def cpu_baseline(wl, lin, lout, budget_s):
    """The oracle on the host cores: remap in row stripes over T threads, low-pass one task per
    segment (the reference's structure), planes sequentially (vf_transform360.c:368-397)."""
    import numpy as np

    from oracle import t360_oracle as O
    from transform360_amd.abi import filter_defaults
    from transform360_amd.handler import frame_seed, noise_bytes
    allowance = host_cpu_allowance()
    T = allowance["usable_cores"]
    ctx = filter_defaults(**wl["ov"])
    t0 = time.perf_counter()
    o = O.Oracle(ctx, threads=T)
    for idx, k in ((0, 0), (1, 1)):
        assert o.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
    init_s = time.perf_counter() - t0
    frame = noise_bytes(lin.frame_bytes, frame_seed(0))
    outs = [np.zeros((h, w), np.uint8) for (w, h) in lout.dims]

    def one_frame():
        for p in range(3):
            assert o.transformFramePlane(lin.plane_view(frame, p), outs[p], 1 if p else 0, p)

    def timed(threads, seconds):
        o.set_threads(threads)
        one_frame()  # warm (tables, page faults, worker pool)
        n, t0 = 0, time.perf_counter()
        while True:
            one_frame()
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                return n / el, n, el

    # T = hardware_concurrency is what the reference gets (cv::parallel_for_ + one std::thread per
    # segment); also try fewer threads, the striped remap of one 4K plane does not scale to
    # hundreds of cores.  The best of the sweep is reported, with its thread count.
    sweep = sorted({T, min(T, 64), min(T, 16), 1}, reverse=True)
    per = budget_s / (len(sweep) + 0.5)
    results = {th: timed(th, per if th > 1 else per / 2) for th in sweep}
    best_t = max(results, key=lambda th: results[th][0])
    fps, n, el = results[best_t]
    fps1 = results[1][0]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    mpix = lout.dims[0][0] * lout.dims[0][1] / 1e6
    o.close()
    # The other way to use a host: one single-threaded stream per core, frames independent (what the GPU sharding does
    # across devices).  Reported next to the reference's own per-frame threading, not instead of it.
    import threading
    streams = max(1, min(T, 256))
    per_stream = [0] * streams

    def stream(i):
        oi = O.Oracle(ctx, threads=1)
        for idx, k in ((0, 0), (1, 1)):
            assert oi.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
        oo = [np.zeros((h, w), np.uint8) for (w, h) in lout.dims]
        ready.wait()
        while time.perf_counter() < stop_at[0]:
            for p in range(3):
                assert oi.transformFramePlane(lin.plane_view(frame, p), oo[p], 1 if p else 0, p)
            per_stream[i] += 1
        oi.close()

    ready, stop_at = threading.Event(), [0.0]
    ths = [threading.Thread(target=stream, args=(i,)) for i in range(streams)]
    for th in ths:
        th.start()
    t_s = time.perf_counter()
    stop_at[0] = t_s + max(1.5, budget_s / 5)
    ready.set()
    for th in ths:
        th.join()
    el_s = time.perf_counter() - t_s
    frame_parallel = {"value": round(sum(per_stream) / el_s * mpix, 1), "unit": "Mpix/s", "streams": streams,
                      "frames": sum(per_stream), "seconds": round(el_s, 2),
                      "what": "one single-threaded oracle per logical core, every stream its own frames (ctypes releases the GIL)"}
    return {
        "value": round(fps * mpix, 3), "unit": "Mpix/s", "cores": best_t, "kind": "port",
        "frame_parallel_streams": frame_parallel,
        "sample": "%d frames of the same workload in %.1f s on %d threads, the best of a thread sweep %s on a host "
                  "that lets this process use %d cores (affinity mask %s, cgroup quota %s cores, node %d logical cores; oracle = "
                  "restatement of the reference's OpenCV path, not linked OpenCV); "
                  "map init %.2f s" % (n, el, best_t,
                                        {th: round(r[0] * mpix, 1) for th, r in sorted(results.items())}, T,
                                        allowance["sched_affinity_cores"], allowance["cgroup_cpu_quota_cores"],
                                        allowance["logical_cores_of_the_node"], init_s),
        "cpu_model": model, "host_cores": T, "host_cpu_allowance": allowance, "fps": round(fps, 3), "value_1thread": round(fps1 * mpix, 3),
    }


def verify_frames(wl, lin, lout, d_in, d_out, frames, seed_of):
    """Output frames `frames` of the device buffers against per-plane oracle calls (bit-exact is the bar)."""
    import numpy as np

    from oracle import t360_oracle as O
    from transform360_amd.abi import filter_defaults
    from transform360_amd.handler import noise_bytes
    ctx = filter_defaults(**wl["ov"])
    o = O.Oracle(ctx, threads=max(1, min(32, os.cpu_count() or 1)))
    for idx, k in ((0, 0), (1, 1)):
        assert o.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
    worst, differing = 0, 0
    for j in frames:
        fin = d_in[j * lin.frame_bytes:(j + 1) * lin.frame_bytes].cpu().numpy()
        # the device generator and its host restatement must agree, or the comparison proves nothing
        assert np.array_equal(fin, noise_bytes(lin.frame_bytes, seed_of(j))), "input frame %d is not the synthetic stream" % j
        fout = d_out[j * lout.frame_bytes:(j + 1) * lout.frame_bytes].cpu().numpy()
        for p in range(len(lout.dims)):
            want = np.zeros((lout.dims[p][1], lout.dims[p][0]), np.uint8)
            assert o.transformFramePlane(lin.plane_view(fin, p), want, 1 if p else 0, p)
            d = np.abs(lout.plane_view(fout, p).astype(np.int16) - want.astype(np.int16))
            worst = max(worst, int(d.max()))
            differing += int(np.count_nonzero(d))
    o.close()
    return {"frames": list(frames), "planes": len(lout.dims), "max_abs_diff": worst, "differing_pixels": differing,
            "against": "CPU oracle (oracle/), per-plane calls on the same synthetic frames"}


def host_abi_rate(wl, lin, lout, ctx):
    """The literal reference ABI: host (malloc'd) planes, one synchronous VideoFrameTransform_transformFramePlane call
    per plane, as vf_transform360.c:368-397 does.  PCIe-bound by construction; reported next to the HBM-resident
    number, never as `value`."""
    import numpy as np

    from transform360_amd import handler
    rng = np.random.default_rng(1)
    planes_in = [rng.integers(0, 256, (h, w), dtype=np.uint8) for (w, h) in lin.dims]
    planes_out = [np.zeros((h, w), np.uint8) for (w, h) in lout.dims]
    with handler.VideoFrameTransform(ctx) as t:
        for idx, k in ((0, 0), (1, 1)):
            assert t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)

        def frame():
            for k in range(len(planes_in)):
                assert t.transformFramePlane(planes_in[k], planes_out[k], 1 if k else 0, k)
        for _ in range(5):
            frame()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            frame()
        dt = (time.perf_counter() - t0) / n
    nbytes = sum(p.nbytes for p in planes_in) + sum(p.nbytes for p in planes_out)
    return {"frames_per_s": round(1 / dt, 1), "ms_per_frame": round(dt * 1e3, 4),
            "Mpix_s": round(lout.dims[0][0] * lout.dims[0][1] / dt / 1e6, 1),
            "pcie_GBps_in_plus_out": round(nbytes / dt / 1e9, 2),
            "note": "host pointers, 3 synchronous calls per frame (the ffmpeg filter's pattern); PCIe-inclusive, not `value`"}


def native_driver_leg(world):
    """The same steps driven by the repo's own C++ host (examples/t360_multi_gpu.cpp: ONE process, one thread + handle + stream
    per device, no Python anywhere near the loop): weak scaling (64 frames per device and step) and BASELINE configs[4] as
    written (64 frames per step sharded over the devices, pipelined calls).  Compute only -- no RCCL is initialised, the
    devices are used exactly as the ranks of this script use them.  Run by rank 0 after every rank's timed legs are over."""
    import re
    exe = os.path.join(ROOT, "examples", "t360_multi_gpu")
    if not os.path.exists(exe):
        return {"skipped": "examples/t360_multi_gpu is not built (make -C examples; __graft_entry__.build() does it)"}
    out_mpix = 1.572864
    import torch
    ndev = max(1, min(world, torch.cuda.device_count()))  # (a gloo rehearsal has more ranks than devices: workers then share)

    def run(label, extra, steps):
        cmd = [exe, "--devices", str(ndev), "--workers", str(world), "--ring-mb", "1440", "--steps", str(steps)] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=75)
        except subprocess.TimeoutExpired:
            return {"error": "timeout", "cmd": " ".join(cmd[1:])}
        m = re.search(r"\(([0-9.]+) frames/s\), ([0-9.]+) ms per step of (\d+) frames", r.stdout)
        if r.returncode != 0 or not m:
            return {"error": (r.stdout + r.stderr)[-300:], "cmd": " ".join(cmd[1:])}
        fps, ms, frames = float(m.group(1)), float(m.group(2)), int(m.group(3))
        return {"what": label, "cmd": "examples/t360_multi_gpu " + " ".join(cmd[1:]), "ms_per_step": ms, "frames_per_step": frames,
                "value": round(fps * out_mpix, 1), "unit": "Mpix/s"}

    rec = {"host": "C++ (examples/t360_multi_gpu.cpp), one process, one thread + handle + stream per device, compute only",
           "n_devices": ndev, "workers": world,
           "weak_64_frames_per_device": run("64 frames per device and step, plain calls, input rotating through HBM", ["--frames", "64"], 100),
           "strong_cfg5_64_frames_total": run("BASELINE configs[4]: 64 frames per step sharded over the devices, pipelined calls (depth 2)",
                                              ["--total-frames", "64", "--pipelined", "2"], 200 if world > 1 else 100)}
    if world == 1:
        rec["one_gpu_share_of_8"] = run("8 frames per step (one GPU's share at 8 GPUs), pipelined calls (depth 2)",
                                        ["--frames", "8", "--pipelined", "2"], 400)
        # like for like: the 64-frame steps through the same pipelined calls are "strong_cfg5_64_frames_total"
        # on one device; the plain-call ratio is kept under its own name
        a, p64, b = rec["weak_64_frames_per_device"], rec["strong_cfg5_64_frames_total"], rec["one_gpu_share_of_8"]
        if "ms_per_step" in p64 and "ms_per_step" in b:
            rec["projected_speedup_at_8_gpus"] = round(p64["ms_per_step"] / b["ms_per_step"], 2)
            rec["projected_speedup_is"] = "pipelined 64-frame step / pipelined 8-frame step"
        if "ms_per_step" in a and "ms_per_step" in b:
            rec["mixed_plain64_over_pipelined8"] = round(a["ms_per_step"] / b["ms_per_step"], 2)
    return rec


def self_launch(args):
    """--gpus N without a launcher: start the N ranks ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


class HipPath:
    """The product path on this rank's GPU: frames resident in HBM, one T360_transformFrames call per step."""
    name = "hip"

    def __init__(self, wl, ctx, lin, lout, F, rank):
        import torch

        from transform360_amd import handler
        self.torch, self.handler = torch, handler
        self.wl, self.ctx, self.lin, self.lout, self.F, self.rank = wl, ctx, lin, lout, F, rank
        # a stream of our own (non-blocking) as torch's current stream: the legacy NULL stream orders against every blocking
        # stream of the process, and the library's idle-stream test before a pipelined call (hipStreamQuery) is then not
        # the cheap question it is for an ordinary stream
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        torch.zeros(1, device="cuda")  # the HIP runtime's own start-up (100+ ms, once per process) is not map generation
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.t = handler.VideoFrameTransform(ctx)
        if os.environ.get("T360_BENCH_FUSED_LOWPASS"):   # --fused-lowpass: A/B of T360_setFusedLowpass (off by default)
            assert self.t.setFusedLowpass(True)
        for idx, k in ((0, 0), (1, 1)):
            assert self.t.generateMapForPlane(*lin.dims[k], *lout.dims[k], idx)
        assert self.t.setStream(self.stream)
        self.init_ms = (time.perf_counter() - t0) * 1e3
        # synthetic stream: this rank's frames of one step, resident in HBM.  When one step's input is smaller than the
        # 256 MB Infinity Cache (short steps, small frames) the timed steps rotate through `groups` such batches --
        # RING_BYTES of distinct input -- so that no step finds its source in a cache because the previous one read the
        # same bytes.  Group 0 is the batch every other leg (verification, checksums, gathers) works on.
        self.groups = max(1, -(-RING_BYTES // (F * lin.frame_bytes))) if F * lin.frame_bytes < RING_BYTES else 1
        self.ring = torch.empty(self.groups * F * lin.frame_bytes, dtype=torch.uint8, device="cuda")
        self.d_in = self.ring[:F * lin.frame_bytes]
        for j in range(self.groups * F):
            seed = self.seed_of(j) if j < F else handler.frame_seed(1_000_000 + rank * 100_000 + j)
            handler.fill_noise(self.ring[j * lin.frame_bytes:(j + 1) * lin.frame_bytes], seed)
        self.d_out = torch.zeros(F * lout.frame_bytes, dtype=torch.uint8, device="cuda")
        self.descs = self.t.plane_descs(lin, lout)

    def seed_of(self, j):
        return self.handler.frame_seed(self.rank * self.F + j)

    def second_handle(self):
        """a second handle on its own stream with its own output buffer: consecutive steps of a frame stream are
        independent, and alternating them between two handles lets step k+1 start while step k drains"""
        if getattr(self, "t2", None) is None:
            self.stream2 = self.torch.cuda.Stream()
            self.t2 = self.handler.VideoFrameTransform(self.ctx)
            for idx, k in ((0, 0), (1, 1)):
                assert self.t2.generateMapForPlane(*self.lin.dims[k], *self.lout.dims[k], idx)
            assert self.t2.setStream(self.stream2)
            self.d_out2 = self.torch.zeros_like(self.d_out)
        return self.t2

    def step2(self, n_frames, inp=None):
        """the same step on the second handle / stream / output buffer"""
        assert self.second_handle().transformFrames(self.d_in if inp is None else inp, self.lin.frame_bytes, self.d_out2,
                                                    self.lout.frame_bytes, n_frames, self.descs)

    def step(self, n_frames, events=None, out=None, inp=None):
        # one call = all three planes of n_frames frames; ONE fused launch of the tiled gather kernel
        # (plus the low-pass launches for config 3)
        if events is not None:
            events[0].record(self.stream)
        assert self.t.transformFrames(self.d_in if inp is None else inp, self.lin.frame_bytes,
                                      self.d_out if out is None else out, self.lout.frame_bytes, n_frames, self.descs)
        if events is not None:
            events[1].record(self.stream)

    def set_pipeline(self, depth):
        """T360_transformFramesPipelined on the ONE handle: consecutive (independent) steps go round-robin over `depth`
        internal streams of the library; output buffers rotate with the lanes (buffer 0 is d_out)"""
        assert self.t.setPipelineDepth(depth)
        self.pipe_depth = depth
        while len(getattr(self, "pipe_outs", [self.d_out])) < depth:
            self.pipe_outs = getattr(self, "pipe_outs", [self.d_out]) + [self.torch.zeros_like(self.d_out)]
        self.pipe_outs = getattr(self, "pipe_outs", [self.d_out])

    def step_pipelined(self, n_frames, inp=None):
        k = getattr(self, "k", 0)
        assert self.t.transformFramesPipelined(self.d_in if inp is None else inp, self.lin.frame_bytes,
                                               self.pipe_outs[k % self.pipe_depth], self.lout.frame_bytes, n_frames, self.descs)

    def fast_steps(self, n_frames, steps, groups, pipelined):
        """The timed loop with everything Python does per step taken out of it: raw pointers and the ctypes call are
        prepared once, the loop is `for a in calls: f(*a)`.  A step of 8 frames is 35-45 us of GPU time and the generic
        loop above (tensor slices, data_ptr(), attribute lookups) costs 15-25 us of host time per step -- host-bound for
        pipelined short steps; a C++ caller (examples/t360_multi_gpu.cpp) has no such cost.  Same calls, same work."""
        import ctypes as C
        L = self.t._l
        f = L.T360_transformFramesPipelined if pipelined else L.T360_transformFrames
        base = self.ring.data_ptr()
        outs = self.pipe_outs if pipelined else [self.d_out]
        h, nd = self.t._h, len(self.descs)
        calls = [(h, C.c_void_p(base + (k % groups) * n_frames * self.lin.frame_bytes), C.c_int64(self.lin.frame_bytes),
                  C.c_void_p(outs[k % len(outs)].data_ptr()), C.c_int64(self.lout.frame_bytes), n_frames, self.descs, nd)
                 for k in range(steps)]

        if pipelined:
            # the K pipelined calls of the timed region issued by the library's own loop (one ctypes call)
            ins = (C.c_void_p * steps)(*[a[1].value for a in calls])
            dsts = (C.c_void_p * steps)(*[a[3].value for a in calls])
            many = L.T360_transformFramesPipelinedMany

            def run_many():
                if not many(h, steps, ins, self.lin.frame_bytes, dsts, self.lout.frame_bytes, n_frames, self.descs, nd):
                    raise RuntimeError("transform call failed")
            return run_many

        def run():
            for a in calls:
                if not f(*a):
                    raise RuntimeError("transform call failed")
        return run

    def new_events(self, n):
        return [(self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)) for _ in range(n)]

    def mark(self, pair, i):
        pair[i].record(self.stream)

    @staticmethod
    def event_ms(pair):
        return pair[0].elapsed_time(pair[1])

    def sync(self):
        # (a device-wide wait: it covers the library's internal pipeline streams as well)
        self.torch.cuda.synchronize()

    def new_output(self):
        return self.torch.zeros_like(self.d_out)

    def new_input(self):
        return self.d_in.clone()

    def frames_of_rank(self, r):
        """the F input frames rank r owns, generated here (rank 0 of the scatter variant holds every rank's frames)"""
        buf = self.torch.empty_like(self.d_in)
        for j in range(self.F):
            self.handler.fill_noise(buf[j * self.lin.frame_bytes:(j + 1) * self.lin.frame_bytes], self.handler.frame_seed(r * self.F + j))
        return buf

    def frame_sums(self, n_frames):
        """byte sum of every output frame of this rank (device reduction)"""
        v = self.d_out[:n_frames * self.lout.frame_bytes].view(n_frames, self.lout.frame_bytes).to(self.torch.int64).sum(dim=1)
        return [int(x) for x in v.cpu()]

    def verify(self, frames):
        return verify_frames(self.wl, self.lin, self.lout, self.d_in, self.d_out, frames, self.seed_of)

    def kernel_name(self):
        return self.t.lastKernel()

    def plan_stats(self):
        return [self.t.planStats(0), self.t.planStats(1)]

    def close(self):
        self.t.close()
        if getattr(self, "t2", None) is not None:
            self.t2.close()


class StubPath:
    """CPU stand-in for the transform with the SAME interface, so that the rank function below -- sharding, barriers,
    timing, gathers, the records -- runs under gloo without a GPU (tests/test_host_cpu.py).  out = 255 - in."""
    name = "stub"

    def __init__(self, wl, ctx, lin, lout, F, rank):
        import torch
        self.torch = torch
        self.lin, self.lout, self.F, self.rank = lin, lout, F, rank
        self.init_ms = 0.0
        from transform360_amd.handler import frame_seed, noise_bytes
        # two groups of F frames, like a HipPath whose step would fit the Infinity Cache: the timed steps rotate
        self.groups = 2
        self.ring = torch.from_numpy(np_concat([noise_bytes(lin.frame_bytes, frame_seed(rank * F + j if j < F else 1_000_000 + rank * 100_000 + j))
                                                for j in range(self.groups * F)]))
        self.d_in = self.ring[:F * lin.frame_bytes]
        self.d_out = torch.zeros(F * lout.frame_bytes, dtype=torch.uint8)

    def step(self, n_frames, events=None, out=None, inp=None):
        t0 = time.perf_counter()
        dst = self.d_out if out is None else out
        d_in = self.d_in if inp is None else inp
        for j in range(n_frames):
            src = d_in[j * self.lin.frame_bytes:j * self.lin.frame_bytes + self.lout.frame_bytes]
            dst[j * self.lout.frame_bytes:(j + 1) * self.lout.frame_bytes] = 255 - src
        if events is not None:
            events[0], events[1] = t0, time.perf_counter()

    def new_events(self, n):
        return [[0.0, 0.0] for _ in range(n)]

    def mark(self, pair, i):
        pair[i] = time.perf_counter()

    @staticmethod
    def event_ms(pair):
        return (pair[1] - pair[0]) * 1e3

    def sync(self):
        pass

    def new_output(self):
        return self.torch.zeros_like(self.d_out)

    def new_input(self):
        return self.d_in.clone()

    def frames_of_rank(self, r):
        from transform360_amd.handler import frame_seed, noise_bytes
        return self.torch.from_numpy(np_concat([noise_bytes(self.lin.frame_bytes, frame_seed(r * self.F + j)) for j in range(self.F)]))

    def frame_sums(self, n_frames):
        v = self.d_out[:n_frames * self.lout.frame_bytes].view(n_frames, self.lout.frame_bytes).to(self.torch.int64).sum(dim=1)
        return [int(x) for x in v]

    def verify(self, frames):
        worst = 0
        for j in frames:
            src = self.d_in[j * self.lin.frame_bytes:j * self.lin.frame_bytes + self.lout.frame_bytes]
            got = self.d_out[j * self.lout.frame_bytes:(j + 1) * self.lout.frame_bytes]
            worst = max(worst, int((got.to(self.torch.int16) - (255 - src).to(self.torch.int16)).abs().max()))
        return {"frames": list(frames), "planes": 3, "max_abs_diff": worst, "differing_pixels": 0, "against": "the stub's definition"}

    def kernel_name(self):
        return "stub"

    def plan_stats(self):
        return [None, None]

    def close(self):
        pass


def np_concat(parts):
    import numpy as np
    return np.concatenate(parts)


def traffic_record(lib_path, kernel_name, frames, config):
    """L2-miss (fabric) bytes per launch from the committed PMC profile (profiles/rNN_traffic.json, written by
    tools/profile_round.sh on a GPU box: rocprofv3 --pmc TCC_EA0_RDREQ_* / WRREQ_*, separate passes) -- counters cannot be
    read inside a run.  Only reported when the profile was taken from THIS library build (sha256 of the .so), the same
    kernel, batch size and config; otherwise null with the reason."""
    import glob
    import hashlib
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True)
    suffix = "" if config == 2 else "_cfg%d" % config
    cands = [c for c in cands if (("_cfg" in os.path.basename(c)) == bool(suffix)) and (not suffix or suffix in os.path.basename(c))]
    if not cands:
        return None, "no profiles/r*%s_traffic.json" % suffix
    path = cands[0]
    try:
        with open(path) as f:
            tr = json.load(f)
        with open(lib_path, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
    except (OSError, ValueError) as e:
        return None, "%s: %s" % (os.path.relpath(path, ROOT), e)
    if tr.get("library_sha16") != sha:
        return None, "%s was profiled from another build of the library (%s, this one is %s)" % (
            os.path.relpath(path, ROOT), tr.get("library_sha16"), sha)
    if tr.get("frames") != frames or tr.get("config") != config or tr.get("kernel", "") not in kernel_name:
        return None, "%s is for config %s, %s frames, kernel %s" % (os.path.relpath(path, ROOT), tr.get("config"),
                                                                   tr.get("frames"), tr.get("kernel"))
    return int(tr["hbm_bytes_per_launch"]), "%s: L2-miss (fabric) bytes, rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}/WRREQ_{,64B} (in front of the Infinity Cache), mean over the " \
        "dispatches of a separate run of this build (library sha256[:16] %s): %d read + %d written" % (
            os.path.relpath(path, ROOT), sha, tr["hbm_read_bytes_per_launch"], tr["hbm_write_bytes_per_launch"])


def run_rank(args, Path, dist, rank, world, coll_dev, wl, ctx):
    """Everything one rank does; returns the record on rank 0 (None elsewhere).  `Path` is HipPath (the benchmark) or
    StubPath (the CPU rehearsal of this very function under gloo)."""
    import torch

    from transform360_amd import handler, sharding
    from transform360_amd.abi import config_output
    in_w, in_h = wl["in_w"], wl["in_h"]
    out_w, out_h = config_output(in_w, in_h, wl["edge"], ctx.output_layout, ctx.input_stereo_format, ctx.output_stereo_format)
    # init state: rank 0's 112-byte context is broadcast over RCCL; every rank rebuilds its maps from it
    ctx = sharding.broadcast_context(ctx, dist, coll_dev)
    lin, lout = handler.FrameLayout(in_w, in_h), handler.FrameLayout(out_w, out_h)
    F = args.frames
    path = Path(wl, ctx, lin, lout, F, rank)

    def barrier():
        path.sync()
        if dist is not None:
            dist.barrier()
            path.sync()

    busy = {"s": 0.0}   # wall-clock seconds this rank kept the GPU busy with transform steps (timed or not)

    def timed_run(n_frames, steps, with_events, after_step=None, rotate=False, before_step=None, alternate=False, pipelined=False):
        """REPEATS x (exactly `steps` steps between barriers); per repeat (elapsed max over ranks, [launch ms]).
        rotate: step k reads the k-th group of n_frames input frames of this rank's F (a short step must not find its
        input in the 256 MB Infinity Cache just because every step reads the same few frames)."""
        ring_frames = getattr(path, "groups", 1) * F
        groups = max(1, ring_frames // n_frames) if rotate else 1
        ring = getattr(path, "ring", None) if groups > 1 else None
        out = []
        fast = None
        if after_step is None and before_step is None and not alternate and hasattr(path, "fast_steps"):
            fast = path.fast_steps(n_frames, steps, groups if ring is not None else 1, pipelined)
        for _ in range(REPEATS):
            # ONE event pair around the K launches of the timed region (on the stream the kernels run on): the average
            # launch duration is its span / K, gaps between launches included.  A pair per step cost 7 us per step -- 3 %
            # of a 64-frame step that is measurement, not work.
            events = path.new_events(1)[0] if with_events else None
            barrier()
            t0 = time.perf_counter()
            if events is not None:
                path.mark(events, 0)
            if fast is not None:
                fast()
            for k in range(steps if fast is None else 0):
                path.k = k  # the double-buffered legs pick their buffers from the step number of THIS timed region
                if before_step is not None:
                    before_step(k)
                inp = ring[(k % groups) * n_frames * lin.frame_bytes:] if ring is not None else None
                if pipelined:
                    path.step_pipelined(n_frames, inp=inp)
                elif alternate and (k & 1):
                    path.step2(n_frames, inp=inp)
                elif inp is not None:
                    path.step(n_frames, inp=inp)
                else:
                    path.step(n_frames)
                if after_step is not None:
                    after_step(k)
            if events is not None:
                path.mark(events, 1)
            if after_step is not None:
                after_step(None)  # drain
            path.sync()
            elapsed = time.perf_counter() - t0
            busy["s"] += elapsed
            if dist is not None:
                dist.barrier()
                el = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
                elapsed = float(el.item())
            out.append((elapsed, [path.event_ms(events) / steps] if events is not None else []))
        return out

    # The device idles while the host plans the gather, and its clocks take a few hundred milliseconds of load to come
    # back: the first repeats used to read 5-15 % slow.  Untimed steps for CLOCK_WARMUP_S seconds first (they also build
    # the gather plan, which the library makes on the first call that needs it), then the W warm-up steps of the
    # contract, then the timed repeats.
    rotate = not os.environ.get("T360_BENCH_NO_ROTATE")

    def warm_steps(n_frames, count, start=0):
        """untimed steps that read the input the way the timed ones will (rotating through the ring when one step's input
        would fit the Infinity Cache): a kernel trace of the whole process then averages HBM-resident launches only"""
        ring_frames = getattr(path, "groups", 1) * F
        groups = max(1, ring_frames // n_frames) if rotate else 1
        ring = getattr(path, "ring", None) if groups > 1 else None
        for k in range(start, start + count):
            path.k = k
            if ring is not None:
                path.step(n_frames, inp=ring[(k % groups) * n_frames * lin.frame_bytes:])
            else:
                path.step(n_frames)

    clock_warmup_steps = 0
    t_first = time.perf_counter()
    path.step(F)
    path.sync()
    first_step_ms = (time.perf_counter() - t_first) * 1e3
    # what a stream that starts cold sees (VERDICT round 5, weak point 9): the 8 steps right behind the first one, before any
    # clock warm-up -- plans exist, clocks and caches do not
    t_c = time.perf_counter()
    warm_steps(F, 8)
    path.sync()
    cold_ms_per_step = (time.perf_counter() - t_c) * 1e3 / 8
    busy["s"] += time.perf_counter() - t_c
    t_w = time.perf_counter()
    while path.name == "hip" and time.perf_counter() - t_w < CLOCK_WARMUP_S:
        warm_steps(F, 8, clock_warmup_steps)
        path.sync()
        clock_warmup_steps += 8
    busy["s"] += time.perf_counter() - t_w
    warm_steps(F, args.warmup)
    runs = timed_run(F, args.steps, True, rotate=rotate)
    if getattr(path, "groups", 1) > 1:
        path.step(F)  # d_out holds group 0's result again
        path.sync()
    elapsed, launch_ms = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    kernel_name = path.kernel_name()

    pipelined = None
    if path.name == "hip" and not args.no_two_streams:
        path.set_pipeline(args.pipeline_depth)
        for k in range(2 * args.pipeline_depth):
            path.k = k
            path.step_pipelined(F)
        path.sync()
        pruns = timed_run(F, args.steps, False, rotate=rotate, pipelined=True)
        p_el = sorted(r[0] for r in pruns)[len(pruns) // 2]
        pipelined = {"what": "the headline steps through T360_transformFramesPipelined (one handle, %d internal streams, independent "
                             "batches, %d output buffers): step k+1 starts while step k drains; NOT the `value` above, whose "
                             "launches are back to back on one stream" % (args.pipeline_depth, args.pipeline_depth),
                     "ms_per_step": round(p_el / args.steps * 1e3, 4),
                     "value": round(args.steps * F * world / p_el * out_w * out_h / 1e6, 1), "unit": "Mpix/s",
                     "repeats_ms_per_step": [round(r[0] / args.steps * 1e3, 4) for r in pruns]}
        path.step(F)  # d_out holds group 0's full-batch result again
        path.sync()

    pipelined_64 = pipelined["ms_per_step"] * 1e-3 * args.steps if pipelined is not None and F == 64 else None

    # BASELINE configs[4] as written: 64 frames in total, frame-sharded -> ceil(64 / N) per rank (strong scaling).  No
    # events inside: a step of 8 frames is 40 us and two event records per step are 10 % of it.
    strong = None
    f5 = min(F, -(-64 // world))
    if args.config == 2:
        warm_steps(f5, max(2, args.warmup))
        # with N > 1 ranks this leg IS the headline (`value`, `roofline`): one event pair around its K launches then
        sruns = timed_run(f5, args.steps, world > 1, rotate=rotate)
        s_el, s_launch_ms = sorted(sruns, key=lambda r: r[0])[len(sruns) // 2]
        strong_kernel = path.kernel_name()
        strong = {"frames_total": 64, "frames_per_gpu": f5, "n_gpus": world, "scaling": "strong",
                  "ms_per_step": round(s_el / args.steps * 1e3, 4),
                  "value": round(min(64, f5 * world) * args.steps / s_el * out_w * out_h / 1e6, 1), "unit": "Mpix/s",
                  "repeats_ms_per_step": [round(r[0] / args.steps * 1e3, 4) for r in sruns]}
        if rotate and getattr(path, "groups", 1) * F > f5:
            # the same f5 input frames every step (88 MB for 8 frames: they stay in the 256 MB Infinity Cache)
            cruns = timed_run(f5, args.steps, False, rotate=False)
            c_el = sorted(r[0] for r in cruns)[len(cruns) // 2]
            strong["input"] = "every step reads other frames of the rank's %d-frame ring (HBM)" % (getattr(path, "groups", 1) * F)
            strong["ms_per_step_same_input_every_step"] = round(c_el / args.steps * 1e3, 4)
        if path.name == "hip" and not args.no_two_streams:
            # The same steps through T360_transformFramesPipelined (ONE handle, the library's internal streams): a frame
            # stream's consecutive batches are independent, so step k+1's workgroups start while step k's last ones drain
            # (the overlap is a product feature now, not two handles in the benchmark).
            path.set_pipeline(args.pipeline_depth)
            for k in range(2 * args.pipeline_depth):
                path.k = k
                path.step_pipelined(f5)
            path.sync()
            pruns = timed_run(f5, args.steps, False, rotate=rotate, pipelined=True)
            p_el = sorted(r[0] for r in pruns)[len(pruns) // 2]
            strong["pipelined"] = {"what": "the same steps through T360_transformFramesPipelined: one handle, %d internal streams, "
                                           "independent batches, %d output buffers" % (args.pipeline_depth, args.pipeline_depth),
                                   "ms_per_step": round(p_el / args.steps * 1e3, 4),
                                   "value": round(min(64, f5 * world) * args.steps / p_el * out_w * out_h / 1e6, 1), "unit": "Mpix/s"}
        if world == 1 and F >= 64 and path.name == "hip":
            # what ONE GPU of an 8-GPU node runs for configs[4]: 8 of the 64 frames per step.  Timed here on one GPU (input
            # rotating through the ring, so from HBM): the projected strong-scaling factor is t(64 frames) / t(8 frames).
            t_w8 = time.perf_counter()
            w8 = 0
            while time.perf_counter() - t_w8 < CLOCK_WARMUP_S / 2:   # the clocks settle for the lighter launches too
                warm_steps(8, 16, w8)
                path.sync()
                w8 += 16
            warm_steps(8, max(2, args.warmup))
            # 8 K steps of 8 frames per repeat: the same number of FRAMES inside a timed region as the headline's K steps of 64
            # (K short steps are ~1 ms between two device-wide syncs: pipeline fill and drain, and clocks that relax in the
            # gaps, are then a visible share of what is timed)
            k8 = 8 * args.steps
            e8 = sorted(r[0] for r in timed_run(8, k8, False, rotate=rotate))[REPEATS // 2] / 8
            t64 = elapsed  # the headline's own K steps of 64 frames (what `value` and `ms_per_step` are computed from)
            strong["projected_8_gpus"] = {
                "frames_per_gpu": 8, "steps_per_timed_region": k8,
                # like for like (ADVICE / VERDICT round 5): both sides plain T360_transformFrames calls back to back on one stream
                "one_stream_ms_per_step": round(e8 / args.steps * 1e3, 4),
                "one_stream_speedup_over_1_gpu": round(t64 / e8, 2),
                "what": "one GPU's share at 8 GPUs (8 frames per step, input rotating through HBM) timed on this GPU; every "
                        "speedup divides a 64-frame step time by an 8-frame step time ISSUED THE SAME WAY (no inter-GPU traffic "
                        "on the path: frames are sharded, SURVEY 8e)"}
            if not args.no_two_streams:
                for k in range(2 * args.pipeline_depth):
                    path.k = k
                    path.step_pipelined(8)
                path.sync()
                e8p = sorted(r[0] for r in timed_run(8, k8, False, rotate=rotate, pipelined=True))[REPEATS // 2] / 8
                p8 = strong["projected_8_gpus"]
                p8["pipelined_ms_per_step"] = round(e8p / args.steps * 1e3, 4)
                if pipelined_64 is not None:
                    # the headline ratio: a stream of 8-frame steps through T360_transformFramesPipelined (one handle, the
                    # library's internal streams, the K calls issued by T360_transformFramesPipelinedMany) against a stream of
                    # 64-frame steps through the SAME calls
                    p8["speedup_over_1_gpu"] = round(pipelined_64 / e8p, 2)
                    p8["speedup_is"] = "pipelined 64-frame step / pipelined 8-frame step (depth %d both)" % args.pipeline_depth
                # NOT like for like, kept for continuity with rounds 4-5 (their `speedup_over_1_gpu`): one-stream 64-frame
                # step / pipelined 8-frame step
                p8["mixed_plain64_over_pipelined8"] = round(t64 / e8p, 2)
            else:
                strong["projected_8_gpus"]["speedup_over_1_gpu"] = strong["projected_8_gpus"]["one_stream_speedup_over_1_gpu"]
                strong["projected_8_gpus"]["speedup_is"] = "one-stream 64-frame step / one-stream 8-frame step"
        if f5 * world != 64:
            strong["note"] = "64 frames do not divide over %d ranks (or --frames < 64/N): every rank ran %d" % (world, f5)

    two_handles = None
    if args.two_handles and path.name == "hip" and world == 1:
        # development comparison: the round-4 way of getting the overlap (two handles, two streams, no ordering events)
        path.second_handle()
        two_handles = {}
        for nfr in sorted({8, F}):
            for k in range(4):
                (path.step2 if k & 1 else path.step)(nfr)
            path.sync()
            th = sorted(r[0] for r in timed_run(nfr, args.steps, False, rotate=rotate, alternate=True))[REPEATS // 2]
            two_handles["%d_frames_ms_per_step" % nfr] = round(th / args.steps * 1e3, 4)

    # the same steps with every step's output frames gathered to rank 0 (RCCL gather over xGMI; a
    # device copy when there is one rank), overlapped with the next step: outputs alternate between two buffers and a
    # step waits only for the gather that used ITS buffer.
    gathered = None
    if args.gather_outputs:
        bufs = [path.d_out, path.new_output()]
        sink = [torch.empty_like(bufs[0]) for _ in range(world)] if rank == 0 else None
        pending = [None, None]

        def gather_step(k):
            if k is None:
                for w in pending:
                    if w is not None:
                        w.wait()
                return
            b = k & 1
            if dist is not None:
                path.sync() if path.name == "stub" else None
                pending[b] = dist.gather(bufs[b], sink if rank == 0 else None, dst=0, async_op=True)
            else:
                sink[0].copy_(bufs[b], non_blocking=True)
            nb = (k + 1) & 1
            if pending[nb] is not None:
                pending[nb].wait()
                pending[nb] = None

        step_plain = path.step

        def step_alt(n_frames, events=None, out=None):
            # step k writes bufs[k & 1], the buffer gather_step(k) sends: k is the step number inside the timed region
            # (a private counter that kept running through the warm-up steps sent the stale buffer)
            step_plain(n_frames, events, bufs[path.k & 1])

        path.step = step_alt
        for w in range(2 * max(1, args.warmup // 2)):  # an even number: the timed region starts on buffer 0
            path.k = w
            path.step(F)
            gather_step(w)
        gather_step(None)
        pending[0] = pending[1] = None
        gruns = timed_run(F, args.steps, False, after_step=gather_step)
        path.step = step_plain
        g_el = sorted(r[0] for r in gruns)[len(gruns) // 2]
        gathered = {"what": "each step's %d output frames of every rank gathered to rank 0, overlapped with the next step" % F,
                    "collective": ("dist.gather (%s)" % args.backend) if dist is not None else "device copy (one rank)",
                    "bytes_to_rank0_per_step": (world - 1) * F * lout.frame_bytes if world > 1 else F * lout.frame_bytes,
                    "ms_per_step": round(g_el / args.steps * 1e3, 4),
                    "value": round(args.steps * F * world / g_el * out_w * out_h / 1e6, 1), "unit": "Mpix/s",
                    "compute_only_ms_per_step": round(elapsed / args.steps * 1e3, 4)}
        path.step(F)  # d_out holds a full-batch result again
        path.sync()
    elif strong is not None:
        path.step(F)  # d_out holds a full-batch result again for the verification below
        path.sync()

    # the stream originates on rank 0 -- every step's input frames are scattered from rank 0
    # (RCCL scatter over xGMI; a device copy when there is one rank) into the buffer the NEXT step reads, while the
    # current step computes, and the outputs are gathered back as above.  Bound by rank 0's links: 7 x 153 GB/s.
    scattered = None
    if args.scatter_inputs:
        ins = [path.d_in, path.new_input()]
        outs = [path.d_out, path.new_output()]
        src_all = [path.frames_of_rank(r) for r in range(world)] if rank == 0 else None   # rank 0 holds every rank's frames
        sink = [torch.empty_like(outs[0]) for _ in range(world)] if rank == 0 else None
        pend_in, pend_out = [None, None], [None, None]

        def feed_step(k):
            """before step k is launched: the buffers it uses are free, and the NEXT step's input starts moving.  The
            scatter is issued before step k, so the collective's stream waits only for step k-1 (the last reader of
            that buffer) and runs beside step k; issued after step k it would queue behind it."""
            b, nb = k & 1, (k + 1) & 1
            for q in (pend_out, pend_in):      # step k reads ins[b] and writes outs[b]
                if q[b] is not None:
                    q[b].wait()
                    q[b] = None
            if dist is not None:
                pend_in[nb] = dist.scatter(ins[nb], src_all if rank == 0 else None, src=0, async_op=True)
            else:
                ins[nb].copy_(src_all[0], non_blocking=True)

        def move_step(k):
            if k is None:
                for w in pend_in + pend_out:
                    if w is not None:
                        w.wait()
                pend_in[0] = pend_in[1] = pend_out[0] = pend_out[1] = None
                return
            b = k & 1
            if dist is not None:
                pend_out[b] = dist.gather(outs[b], sink if rank == 0 else None, dst=0, async_op=True)
            else:
                sink[0].copy_(outs[b], non_blocking=True)

        step_plain = path.step

        def step_alt(n_frames, events=None, out=None, inp=None):
            step_plain(n_frames, events, outs[path.k & 1], ins[path.k & 1])

        path.step = step_alt
        for w in range(2 * max(1, args.warmup // 2)):
            path.k = w
            feed_step(w)
            path.step(F)
            move_step(w)
        move_step(None)
        xruns = timed_run(F, args.steps, False, after_step=move_step, before_step=feed_step)
        path.step = step_plain
        x_el = sorted(r[0] for r in xruns)[len(xruns) // 2]
        scattered = {"what": "each step's %d input frames per rank scattered from rank 0 into the next step's buffer and its "
                             "output frames gathered to rank 0, both overlapped with the computing step" % F,
                     "collective": ("dist.scatter + dist.gather (%s)" % args.backend) if dist is not None else "device copies (one rank)",
                     "bytes_from_rank0_per_step": (world - 1 if world > 1 else 1) * F * lin.frame_bytes,
                     "bytes_to_rank0_per_step": (world - 1 if world > 1 else 1) * F * lout.frame_bytes,
                     "ms_per_step": round(x_el / args.steps * 1e3, 4),
                     "value": round(args.steps * F * world / x_el * out_w * out_h / 1e6, 1), "unit": "Mpix/s",
                     "compute_only_ms_per_step": round(elapsed / args.steps * 1e3, 4)}
        path.step(F)
        path.sync()

    # verification outside the timed region: oracle comparison of frames of the last step + the checksum of every frame
    verified = None
    if not args.no_verify:
        frames = sorted({0, min(15, F - 1), min(16, F - 1), min(31, F - 1), min(32, F - 1), F - 1})
        if args.config == 4:
            frames = frames[:2]  # 12.6 Mpix Lanczos4 frames: seconds each on the host
        verified = path.verify(frames)
        ok = torch.tensor([1 if verified["max_abs_diff"] == 0 else 0], dtype=torch.int64, device=coll_dev)
        if dist is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        verified["all_ranks_ok"] = bool(int(ok.item()))
    sums = path.frame_sums(F)
    all_sums = sharding.gather_checksums({rank * F + j: v for j, v in enumerate(sums)}, F * world, dist, coll_dev)
    checksums = [sum(all_sums[r * F:(r + 1) * F]) for r in range(world)]

    res = None
    if rank == 0:
        if verified is not None and not (verified["max_abs_diff"] == 0 and verified["all_ranks_ok"]):
            print(json.dumps({"error": "output differs from the oracle: no throughput is reported", "verified": verified}))
            raise SystemExit(1)
        out_mpix = out_w * out_h / 1e6
        alg_frame = lin.payload_bytes() + lout.payload_bytes()
        weak_fps = args.steps * F * world / elapsed
        # Which leg is the headline.  One GPU: K steps of F frames (BASELINE configs[1]; with F = 64 also configs[4] at N = 1).
        # N > 1 GPUs on the default workload: BASELINE configs[4] AS WRITTEN -- 64 frames per step IN TOTAL, ceil(64 / N) per
        # rank, strong scaling -- so that a 1-2-4-8 GPU series of this line is the strong-scaling curve of configs[4] and not
        # eight independent copies of the one-GPU job.  The weak figure (F frames per GPU) stays
        # in the line as `weak_value` / "weak_64_frames_per_gpu".
        headline_strong = world > 1 and strong is not None and F == 64
        if headline_strong:
            h_elapsed, h_frames_step, h_launch_ms, h_kernel, h_F = s_el, min(64, f5 * world), s_launch_ms, strong_kernel, f5
        else:
            h_elapsed, h_frames_step, h_launch_ms, h_kernel, h_F = elapsed, F * world, launch_ms, kernel_name, F
        fps = args.steps * h_frames_step / h_elapsed
        launch_alg = h_F * alg_frame   # the fused launch moves every plane of h_F frames
        launch_avg_s = (sum(h_launch_ms) / len(h_launch_ms)) * 1e-3
        from transform360_amd import _lib
        traffic, traffic_source = (None, "stub") if path.name != "hip" else traffic_record(_lib.LIB_PATH, h_kernel, h_F, args.config)
        lib_sha = None
        if path.name == "hip":
            import hashlib
            with open(_lib.LIB_PATH, "rb") as f:
                lib_sha = hashlib.sha256(f.read()).hexdigest()[:16]
        res = {
            "metric": ("Mpix/s remapped (4K equirect→512-edge cubemap, bicubic)" if args.config == 2
                       else "Mpix/s remapped (%s)" % wl["name"]) if path.name == "hip" else "STUB (no transform ran)",
            "value": round(fps * out_mpix, 1), "unit": "Mpix/s",
            # both scaling figures at the head of the line: `strong_cfg5_value` = BASELINE configs[4] as written (64 frames
            # per step in total, plain calls; == `value` when N > 1), `weak_value` = F frames per GPU and step (== `value` at N = 1)
            "strong_cfg5_value": strong["value"] if strong is not None else None,
            "strong_cfg5_pipelined_value": (strong.get("pipelined") or {}).get("value") if strong is not None else None,
            "weak_value": round(weak_fps * out_mpix, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(h_elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if (headline_strong or (world == 1 and F == 64 and args.config == 2)) else "weak",
            "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"] + (" -- BASELINE configs[4]: 64 frames per step sharded over %d GPUs" % world if headline_strong else ""),
                       "frames_per_step_per_gpu": h_F, "frames_per_step_total": h_frames_step,
                       "pixel_format": "yuv420p 8-bit",
                       "in": "%dx%d" % (in_w, in_h), "out": "%dx%d" % (out_w, out_h),
                       "sharding": "whole frames per rank, no data-path collective", "input": "resident in HBM"},
            "fps": round(fps, 1),
            # cold and warm-up figures next to the timed ones (VERDICT round 5, weak point 9): the first call after init plans
            # the gather on the host; `clock_warmup_steps` untimed steps then bring the clocks up before the W warm-up steps
            "first_step_ms": round(first_step_ms, 1), "cold_ms_per_step_steps_2_to_9": round(cold_ms_per_step, 4),
            "clock_warmup_steps": clock_warmup_steps,
            "library_sha16": lib_sha,
            "repeats": REPEATS, "repeats_ms_per_step": [round(r[0] / args.steps * 1e3, 4) for r in (sruns if headline_strong else runs)],
            "input_ring": {"groups_of_F_frames": getattr(path, "groups", 1), "bytes": getattr(path, "groups", 1) * F * lin.frame_bytes,
                           "why": "timed steps rotate through this much distinct input, several times the 256 MB Infinity Cache: a step "
                                  "whose input fits the cache would find it there, and even re-reading ONE 708 MB batch of config 2 every "
                                  "step measured 1 % faster than rotating through three (0.2370 vs 0.2395 ms, round 5, "
                                  "profiles/r05_experiments/README.md call 5; rounds 1-4 did the former)"},
            "frames_timed_per_gpu": REPEATS * args.steps * h_F,
            # context for the timed region: `ms_per_step` x steps is a few milliseconds; over the
            # whole run this rank kept the GPU busy with transform steps (clock ramp, warm-up, every timed leg) for
            "gpu_busy_s_all_legs": round(busy["s"], 3),
            "frac_of_hbm_roofline_whole_job": round(alg_frame * fps / (HBM_PEAK_BPS * world), 4),
            "roofline": {
                "bound": "hbm",
                "kernel": "%s: Y+U+V planes of %d frames per launch%s" % (
                    h_kernel, h_F, " (+ the low-pass launches of the step)" if ctx.enable_low_pass_filter else ""),
                "achieved": round(launch_alg / launch_avg_s / 1e9, 1), "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                "frac": round(launch_alg / launch_avg_s / HBM_PEAK_BPS, 4),
                "algorithmic_bytes_per_launch": launch_alg, "avg_launch_ms": round(launch_avg_s * 1e3, 4),
                "avg_launch_ms_from": "one HIP event pair around the %d launches of the timed region / %d" % (args.steps, args.steps),
                "traffic": traffic, "traffic_source": traffic_source,
            },
            "verified": verified,
            "gather_plan": path.plan_stats(),
            "init_ms": round(path.init_ms, 1),
            "output_checksums": checksums,
        }
        if strong is not None:
            res["strong_cfg5"] = strong
        if headline_strong:
            res["weak_64_frames_per_gpu"] = {"value": round(weak_fps * out_mpix, 1), "unit": "Mpix/s", "scaling": "weak",
                                             "ms_per_step": round(elapsed / args.steps * 1e3, 4), "frames_per_step_per_gpu": F,
                                             "kernel": kernel_name,
                                             "repeats_ms_per_step": [round(r[0] / args.steps * 1e3, 4) for r in runs]}
        if pipelined is not None:
            res["pipelined"] = pipelined
        if two_handles is not None:
            res["two_handles"] = two_handles
        if gathered is not None:
            res["gather_outputs"] = gathered
        if scattered is not None:
            res["scatter_gather"] = scattered
    return res, path, (lin, lout)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)  # 32 x 64 = 2 048 frames inside one event bracket (>= 2 000)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=64, help="frames per step per GPU (BASELINE config 5: 64)")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison (development sweeps only)")
    ap.add_argument("--no-host-abi", action="store_true", help="skip the host-pointer ABI leg (profiling runs)")
    ap.add_argument("--gather-outputs", action="store_true",
                    help="also time the steps with every step's output frames gathered to rank 0, overlapped with the next step")
    ap.add_argument("--scatter-inputs", action="store_true",
                    help="also time the steps with the inputs scattered from rank 0 and the outputs gathered to it, overlapped")
    ap.add_argument("--no-two-streams", "--no-pipelined", dest="no_two_streams", action="store_true",
                    help="skip the legs that issue steps through T360_transformFramesPipelined (kernel traces: overlapped "
                         "launches of the hot kernel would enter its average duration)")
    ap.add_argument("--pipeline-depth", type=int, default=2, help="internal streams of the pipelined legs (1..4)")
    ap.add_argument("--no-native", action="store_true", help="skip the leg that runs the native C++ driver (examples/t360_multi_gpu)")
    ap.add_argument("--two-handles", action="store_true", help="also time steps alternating between two handles on two streams (development)")
    ap.add_argument("--fused-lowpass", action="store_true",
                    help="config 3: T360_setFusedLowpass(1) -- the low-pass inside the gather tiles (off by default: slower, DESIGN.md 5.2)")
    ap.add_argument("--stub", action="store_true",
                    help="CPU rehearsal of the rank function with a stand-in transform (tests; never a benchmark result)")
    args = ap.parse_args()
    if args.fused_lowpass:
        os.environ["T360_BENCH_FUSED_LOWPASS"] = "1"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)

    import torch

    from transform360_amd import _lib
    from transform360_amd.abi import filter_defaults

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # T360_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks then
    # share devices and the few collectives around the path run on CPU tensors); the driver's runs use
    # nccl (= RCCL), one rank per GPU
    backend = os.environ.get("T360_DIST_BACKEND", "gloo" if args.stub else "nccl")
    args.backend = backend
    if not args.stub:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists for the remap path)"
        if backend == "nccl" and world > torch.cuda.device_count():
            raise SystemExit("--gpus %d but only %d GPU(s) visible (T360_DIST_BACKEND=gloo rehearses the multi-rank "
                             "path on fewer devices)" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
        if backend != "nccl" and world > 1 and (args.gather_outputs or args.scatter_inputs):
            # gloo has no gather / scatter for tensors in device memory (the calls never complete: a 10-minute hang on
            # the GPU box, 2026-09-24); these legs move frames between GPUs and need RCCL -- their gloo rehearsal is --stub
            raise SystemExit("--gather-outputs / --scatter-inputs with frames in HBM need the nccl (RCCL) backend; "
                             "rehearse these legs on CPU with --stub")
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    # T360_FORCE_DIST=1: initialise the process group even for ONE rank, so that the launcher form of a multi-GPU run --
    # RCCL init, context broadcast, barriers, all_reduce, all_gather, the rendezvous-store wait -- can be rehearsed on a
    # one-GPU box (every collective is then trivial, every API call real)
    if world > 1 or os.environ.get("T360_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    build_flags = 0
    if not args.stub:
        L = _lib.load()
        build_flags = int(L.T360_buildFlags())
        if build_flags and not os.environ.get("T360_BENCH_ALLOW_INSTRUMENTED"):
            raise SystemExit("refusing to benchmark an instrumented library (%s): it reads tuning and wrong-pixel switches "
                             "from the environment" % _lib.LIB_PATH)

    wl = workload(args.config)
    if args.stub:
        wl = dict(wl, in_w=256, in_h=128, edge=32)
    ctx = filter_defaults(**wl["ov"])
    res, path, (lin, lout) = run_rank(args, StubPath if args.stub else HipPath, dist, rank, world, coll_dev, wl, ctx)

    # The repo's own C++ host on the same devices (north_star: "host side is the repo's own C++"): after every rank is done,
    # rank 0 runs the native driver over all `world` devices while the other ranks wait on the rendezvous store (a CPU-side
    # wait: an RCCL barrier would park a spinning kernel on every other GPU).
    native = None
    if not args.stub and args.config == 2 and not args.no_native and not build_flags:
        path.sync()
        store = None
        if dist is not None:
            dist.barrier()
            path.sync()
            store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            try:
                native = native_driver_leg(world)
            except Exception as e:  # the benchmark line never depends on the example binary
                native = {"error": repr(e)[:300]}
            if store is not None:
                store.set("t360_native_done", "1")
        elif store is not None:
            try:
                import datetime
                store.wait(["t360_native_done"], datetime.timedelta(seconds=240))
            except Exception:  # rank 0's leg is bounded by its own timeouts; never let a waiting rank fail the job
                pass
    if rank == 0:
        if native is not None:
            res["native_driver"] = native
        if not args.stub:
            res["library"] = {"path": os.path.relpath(_lib.LIB_PATH, ROOT), "build_flags": build_flags,
                              "version": _lib.load().T360_version().decode()}
        if world == 1 and not args.stub:
            # "achievable" HBM rate of this box for reference: a plain device-to-device copy (read + write)
            try:
                n = 1 << 30
                a_ = torch.empty(n, dtype=torch.uint8, device="cuda")
                b_ = torch.empty(n, dtype=torch.uint8, device="cuda")
                b_.copy_(a_)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(5):
                    b_.copy_(a_)
                ev1.record()
                torch.cuda.synchronize()
                res["hbm_copy_GBps_measured"] = round(5 * 2 * n / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
                del a_, b_
            except RuntimeError:
                pass
        res["Mpix_s_in"] = round(res["fps"] * wl["in_w"] * wl["in_h"] / 1e6, 1)
        if world == 1 and args.config == 2 and not args.no_host_abi and not args.stub:
            res["host_abi"] = host_abi_rate(wl, lin, lout, ctx)
        if not args.no_cpu_baseline and world == 1 and not args.stub:
            res["cpu_baseline"] = cpu_baseline(wl, lin, lout, args.cpu_seconds)
        print(json.dumps(res))
    path.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
