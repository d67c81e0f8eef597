"""Throughput of the multi-view triangulation path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload algebraic|volumetric|stream16|ransac]

A step is one pass of the path over one resident batch of synthetic frames:
  algebraic  (default: the headline configuration) 4096 frames x 4 views x 17 joints of 96x96 heatmaps per GPU:
             soft-argmax (multiplier folded in, normalised heatmaps written back as the reference returns them)
             -> confidence normalisation + keypoint upscale + DLT.
  volumetric 64 frames: 4 views x 32-channel 96x96 features unprojected into 64^3 (softmax
             aggregation), 17 volumes per frame soft-argmaxed with coordinates (normalised volumes written).
  stream16   16 views: p50 latency of one frame and throughput at 1024 frames.
  ransac     the reference's third triangulation model on the algebraic shapes: arg-max keypoints of 4096 x 4 x 17
             heatmaps, then RANSAC over every view pair + DLT over the inliers + Huber refinement per joint.
With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds its own batch (weak scaling,
frames sharded across ranks) and each step ends with the RCCL all-gather of the (frames, 17, 3) keypoints, which
runs on RCCL's stream while the next step computes; every all-gather has completed before the clock stops.
Rank 0 prints one JSON line; `value` is the whole-job frames/s with inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from poseestimation_amd import _cabi, sharding, synth  # noqa: E402
from poseestimation_amd.mvn.models.triangulation import AlgebraicTriangulationNet  # noqa: E402
from poseestimation_amd.mvn.utils import op  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak
ALL_GATHER_MS = None    # one keypoint all-gather on its own (N > 1), measured after the timed loop


def algebraic_head():
    net = AlgebraicTriangulationNet.__new__(AlgebraicTriangulationNet)
    torch.nn.Module.__init__(net)
    net.use_confidences, net.heatmap_softmax, net.heatmap_multiplier = True, True, 100.0
    return net


class EventTimer:
    """HIP events on torch's current stream (the stream the kernels are launched on), one pair per launch."""

    def __init__(self):
        self.pairs = []

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()
        return self

    def __exit__(self, *exc):
        self.b.record()
        self.pairs.append((self.a, self.b))

    def mean_ms(self):
        return float(np.mean([a.elapsed_time(b) for a, b in self.pairs])) if self.pairs else float("nan")


def measured_traffic(kernel, frames, views, writeback):
    """HBM bytes per launch of `kernel` from the rocprofv3 counter passes stored under profiles/ (collected
    with the same launch shape in a separate profiled run: counters cannot be read from inside this process)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[kernel]
    except (OSError, KeyError, ValueError):
        return None, None
    if (rec["frames"], rec["views"], rec["heatmap_writeback"]) != (frames, views, writeback):
        return None, None
    return rec["hbm_bytes_per_launch"], rec["source"]


def cpu_baseline_algebraic(n_views, sample_frames):
    """The torch-CPU port (oracle/torch_port.py) on a bounded sample of the same workload, host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_port
    threads = torch.get_num_threads()
    case = synth.algebraic_case(1234 + 1, sample_frames, n_views)
    secs = torch_port.time_algebraic(case, repeats=9, warmup=2)
    return {"value": sample_frames / secs, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames x %d views x 17 joints, soft-argmax + glue + per-point host SVD, median of 9 passes" %
                      (sample_frames, n_views)}


def cpu_baseline_volumetric(sample_frames):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_port
    threads = torch.get_num_threads()
    case = synth.volumetric_case(1234 + 2, sample_frames, 4)
    f, p, c, vc, vol = (torch.from_numpy(case[k]) for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    ts = []
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            torch_port.unproject(f, p, c, "softmax", vc)
            torch_port.soft_argmax_3d_coords(vol, c)
            if i >= 1:
                ts.append(time.perf_counter() - t0)
    secs = float(np.median(ts))
    return {"value": sample_frames / secs, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames, unproject (softmax) + 3-D soft-argmax, median of 2" % sample_frames}


def run_algebraic(args, dev, rank, world, n_views=4, frames=4096, writeback=True):
    case = synth.device_algebraic_case(1234 + 1 + rank, frames, n_views, dev)
    hm, conf, proj = case["heatmaps"], case["confidences"], case["proj"]
    net = algebraic_head()
    b, v, j, h, w = hm.shape
    lib = _cabi.load()
    k1 = EventTimer()
    k2 = EventTimer()
    kp_hm = torch.empty((b * v, j, 2), dtype=torch.float32, device=dev)
    hm_out = torch.empty_like(hm) if writeback else None
    kp_img = torch.empty((b, v, j, 2), dtype=torch.float32, device=dev)
    conf_n = torch.empty((b, v, j), dtype=torch.float32, device=dev)
    sg = StepGather((b, j, 3), dev, world)
    st = _cabi.stream(dev)

    def step(timed):
        # the two C-ABI calls AlgebraicTriangulationNet.triangulate_from_heatmaps makes, with outputs preallocated
        kp3d = sg.buffer()
        if timed:
            with k1:
                rc = lib.mvx_soft_argmax_2d(_cabi.ptr(hm), b * v * j, h, w, 100.0, 1, _cabi.ptr(kp_hm), _cabi.ptr(hm_out), st)
            with k2:
                rc |= lib.mvx_algebraic_glue_dlt(_cabi.ptr(proj), _cabi.ptr(kp_hm), _cabi.ptr(conf), b, v, j, 384.0 / w,
                                                 384.0 / h, _cabi.ptr(kp_img), _cabi.ptr(conf_n), _cabi.ptr(kp3d), st)
        else:
            rc = lib.mvx_soft_argmax_2d(_cabi.ptr(hm), b * v * j, h, w, 100.0, 1, _cabi.ptr(kp_hm), _cabi.ptr(hm_out), st)
            rc |= lib.mvx_algebraic_glue_dlt(_cabi.ptr(proj), _cabi.ptr(kp_hm), _cabi.ptr(conf), b, v, j, 384.0 / w,
                                             384.0 / h, _cabi.ptr(kp_img), _cabi.ptr(conf_n), _cabi.ptr(kp3d), st)
        if rc != 0:
            raise RuntimeError("mvx call failed: %s" % lib.mvx_last_error().decode())
        sg.submit()

    # the drop-in class path gives the same numbers as the preallocated calls
    ref3d = net.triangulate_from_heatmaps(hm[:8], conf[:8], proj[:8], (384, 384))[0]
    step(False)
    out = sg.gathered()
    own = out[rank * frames: rank * frames + 8] if world > 1 else out[:8]
    assert torch.equal(own, ref3d)
    # sanity: synthetic joints are recovered to within the blob noise
    err = (own.cpu().numpy() - case["joints_3d"][:8]).__abs__().max()
    assert err < 80.0, "triangulated joints are %g mm from the synthetic ground truth" % err

    ms = timed_loop(step, args, dev, world, sg)
    bytes_per_launch = frames * v * j * h * w * 4 * (2 if writeback else 1) + frames * v * j * 8
    roof = {"bound": "hbm", "kernel": "soft_argmax_2d_reg<9>", "achieved": bytes_per_launch / (k1.mean_ms() * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": k1.mean_ms()}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"], roof["traffic_source"] = measured_traffic("soft_argmax_2d_reg<9>", frames, v, writeback)
    extra = {"dlt_kernel_ms": k2.mean_ms(), "max_abs_error_vs_synthetic_truth_mm": float(err)}
    cfg = {"workload": "algebraic: soft-argmax 2D + confidence-weighted DLT, %d frames/GPU x %d views x 17 joints, 96x96 heatmaps"
                       % (frames, v), "frames_per_gpu": frames, "views": v, "joints": j, "heatmap_writeback": writeback,
           "parallelism": "frames sharded over %d GPU(s), all-gather of keypoints" % world}
    return ms, frames, roof, cfg, extra


def run_volumetric(args, dev, rank, world, frames=64):
    case = synth.device_volumetric_case(1234 + 2 + rank, frames, 4, dev)
    feats, proj, coord, vconf, vols = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    k3 = EventTimer()
    k4 = EventTimer()
    sg = StepGather((frames, 17, 3), dev, world)

    def step(timed):
        dst = sg.buffer()
        if timed:
            with k3:
                u = op.unproject_heatmaps(feats, proj, coord, "softmax", vconf)
            with k4:
                kp, _ = op.integrate_tensor_3d_with_coordinates(vols, coord, True)
        else:
            u = op.unproject_heatmaps(feats, proj, coord, "softmax", vconf)
            kp, _ = op.integrate_tensor_3d_with_coordinates(vols, coord, True)
        del u
        dst.copy_(kp)
        sg.submit()

    step(False)
    ms = timed_loop(step, args, dev, world, sg)
    n_vox = 64 ** 3
    b3 = frames * (4 * 32 * 96 * 96 * 4 + n_vox * 12 + 32 * n_vox * 4)
    b4 = frames * (17 * n_vox * 4 * 2 + n_vox * 12)
    roof = {"bound": "hbm", "kernel": "unproject (channel-last copy + unproject_tile32<softmax>)",
            "achieved": b3 / (k3.mean_ms() * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "algorithmic_bytes_per_launch": b3, "avg_launch_ms": k3.mean_ms()}
    roof["frac"] = roof["achieved"] / roof["peak"]
    # the unprojection gathers 4 taps x 32 channels x 4 B per voxel and view through the vector L1; a CU returns
    # 46 B/clk for loads of that shape when everything hits (tools/l1_bandwidth.hip: 28.2 TB/s over the chip)
    tap_bytes = frames * n_vox * 4 * 512
    extra = {"soft_argmax_3d_ms": k4.mean_ms(), "soft_argmax_3d_GBps": b4 / (k4.mean_ms() * 1e-3) / 1e9,
             "unproject_tap_TBps": tap_bytes / (k3.mean_ms() * 1e-3) / 1e12, "measured_l1_gather_ceiling_TBps": 28.2}
    cfg = {"workload": "volumetric: unproject 4 views x 32ch 96x96 -> 64^3 (softmax) + 3-D soft-argmax of 17 volumes, %d frames/GPU"
                       % frames, "frames_per_gpu": frames, "parallelism": "frames sharded over %d GPU(s)" % world}
    return ms, frames, roof, cfg, extra


def cpu_baseline_ransac(sample_frames):
    """The numpy/scipy restatement of RANSACTriangulationNet (oracle/mvn_oracle.py) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mvn_oracle
    from poseestimation_amd.mvn.models.triangulation import all_view_pairs
    case = synth.algebraic_case(1234 + 1, sample_frames, 4)
    pairs = all_view_pairs(sample_frames * 17, 4)
    t0 = time.perf_counter()
    kp = mvn_oracle.heatmap_argmax_keypoints(case["heatmaps"], (384, 384))
    mvn_oracle.ransac_batch(case["proj"], kp, pairs, 15, True)
    secs = time.perf_counter() - t0
    return {"value": sample_frames / secs, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames x 4 views x 17 joints: numpy arg-max, 6 pair hypotheses per joint with numpy SVD, "
                      "scipy least_squares(huber) refinement; one pass" % sample_frames}


def run_ransac(args, dev, rank, world, frames=4096):
    from poseestimation_amd.mvn.models.triangulation import all_view_pairs
    case = synth.device_algebraic_case(1234 + 1 + rank, frames, 4, dev)
    hm, proj = case["heatmaps"], case["proj"]
    b, v, j, h, w = hm.shape
    lib = _cabi.load()
    pairs = torch.from_numpy(np.ascontiguousarray(all_view_pairs(b * j, v))).to(dev)
    n_iters = pairs.shape[1]
    kp = torch.empty((b, v, j, 2), dtype=torch.int64, device=dev)
    kp_f64 = torch.empty((b, v, j, 2), dtype=torch.float64, device=dev)
    out = torch.empty((b, j, 3), dtype=torch.float64, device=dev)
    sg = StepGather((b, j, 3), dev, world)
    masks = torch.empty((b, j), dtype=torch.int32, device=dev)
    st = _cabi.stream(dev)
    ka, kr = EventTimer(), EventTimer()

    def launch_argmax():
        return lib.mvx_heatmap_argmax(_cabi.ptr(hm), b * v * j, h, w, 384.0 / w, 384.0 / h, _cabi.ptr(kp), st)

    def launch_ransac(kp3d):
        kp_f64.copy_(kp)
        rc = lib.mvx_triangulate_ransac(_cabi.ptr(proj), _cabi.ptr(kp_f64), _cabi.ptr(pairs), b, v, j, n_iters, 15.0, 1,
                                        _cabi.ptr(out), _cabi.ptr(masks), st)
        kp3d.copy_(out)
        return rc

    def step(timed):
        kp3d = sg.buffer()
        if timed:
            with ka:
                rc = launch_argmax()
            with kr:
                rc |= launch_ransac(kp3d)
        else:
            rc = launch_argmax() | launch_ransac(kp3d)
        if rc != 0:
            raise RuntimeError("mvx call failed: %s" % lib.mvx_last_error().decode())
        sg.submit()

    step(False)
    got = sg.gathered()
    own = got[rank * frames: rank * frames + 8] if world > 1 else got[:8]
    err = (own.cpu().numpy() - case["joints_3d"][:8]).__abs__().max()
    assert err < 80.0, "triangulated joints are %g mm from the synthetic ground truth" % err
    ms = timed_loop(step, args, dev, world, sg)
    bytes_per_launch = frames * v * j * (h * w * 4 + 16)
    roof = {"bound": "hbm", "kernel": "heatmap_argmax<9>", "achieved": bytes_per_launch / (ka.mean_ms() * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": ka.mean_ms()}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"], roof["traffic_source"] = measured_traffic("heatmap_argmax<9>", frames, v, False)
    extra = {"ransac_kernel_and_casts_ms": kr.mean_ms(), "max_abs_error_vs_synthetic_truth_mm": float(err),
             "mean_inlier_views": float(torch.bitwise_and(masks.view(-1, 1) >> torch.arange(4, device=dev), 1).sum(1).float().mean())}
    cfg = {"workload": "ransac: heatmap arg-max + RANSAC (all 6 view pairs) + inlier DLT + Huber refinement, "
                       "%d frames/GPU x 4 views x 17 joints, 96x96 heatmaps" % frames, "frames_per_gpu": frames,
           "views": v, "joints": j, "parallelism": "frames sharded over %d GPU(s), all-gather of keypoints" % world}
    return ms, frames, roof, cfg, extra


class StepGather:
    """The all-gather that ends a step, overlapped with the next step: step k writes its keypoints into buffer
    k % 2 and its collective runs (on RCCL's stream) while step k + 1 computes; a buffer is written again only
    after the collective that reads it has finished.  `drain()` completes everything before timing stops.
    In the single-GPU rehearsal the same logic runs over gloo on host copies."""

    def __init__(self, local_shape, dev, world):
        self.world = world
        self.dev = dev
        self.host = world > 1 and dist.get_backend() == "gloo"
        self.buffers = [torch.empty(local_shape, dtype=torch.float32, device=dev) for _ in range(2)]
        self.pipe = None
        if world > 1:
            self.pipe = sharding.PipelinedKeypointGather(local_shape, torch.float32, "cpu" if self.host else dev, depth=2)
            self.host_send = [torch.empty(local_shape, dtype=torch.float32) for _ in range(2)] if self.host else None
        self.step = 0
        self.last_slot = None

    def buffer(self):
        """The keypoint buffer of the step about to run (free of any collective still reading it)."""
        slot = self.step % 2
        if self.pipe is not None and self.pipe.pending[slot] is not None:
            self.pipe.result(slot)
        return self.buffers[slot]

    def submit(self):
        slot = self.step % 2
        self.step += 1
        if self.pipe is None:
            return self.buffers[slot]
        if self.host:
            self.host_send[slot].copy_(self.buffers[slot])
            self.last_slot = self.pipe.submit(self.host_send[slot])
        else:
            self.last_slot = self.pipe.submit(self.buffers[slot])
        return None

    def gathered(self):
        """All ranks' keypoints of the last submitted step (waits for its collective)."""
        if self.pipe is None:
            return self.buffers[(self.step - 1) % 2]
        return self.pipe.result(self.last_slot).to(self.dev)

    def drain(self):
        if self.pipe is not None:
            self.pipe.drain()

    def exposed_ms(self, repeats=10):
        """Duration of one all-gather on its own (nothing to overlap with), host-visible, median."""
        if self.pipe is None:
            return None
        self.drain()
        ts = []
        for _ in range(repeats):
            dist.barrier()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            self.submit()
            self.drain()
            torch.cuda.synchronize(self.dev)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts) * 1e3)


def timed_loop(step, args, dev, world, gather):
    for _ in range(args.warmup):
        step(False)
    gather.drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    gather.drain()   # every step's all-gather has completed inside the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    secs = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([secs], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
        global ALL_GATHER_MS
        ALL_GATHER_MS = gather.exposed_ms()
    return secs * 1e3 / args.steps


def latency_p50(dev, n_views=16, launches=1000):
    """One frame, 16 views: host-visible p50 of soft-argmax + DLT, each launch synchronised."""
    case = synth.device_algebraic_case(1234 + 5, 1, n_views, dev)
    net = algebraic_head()
    hm, conf, proj = case["heatmaps"], case["confidences"], case["proj"]
    for _ in range(50):
        net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(launches):
        t0 = time.perf_counter()
        net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    wrapper = (float(np.percentile(ts, 50) * 1e6), float(np.percentile(ts, 99) * 1e6))
    # the same frame through the bare C entry point with caller-owned outputs (what a serving loop would do)
    lib = _cabi.load()
    b, v, j, h, w = hm.shape
    hm_out = torch.empty_like(hm)
    kp_hm = torch.empty((b * v * j, 2), dtype=torch.float32, device=dev)
    kp_img = torch.empty((b, v, j, 2), dtype=torch.float32, device=dev)
    conf_n = torch.empty((b, v, j), dtype=torch.float32, device=dev)
    kp3d = torch.empty((b, j, 3), dtype=torch.float32, device=dev)
    args = (_cabi.ptr(hm), _cabi.ptr(conf), _cabi.ptr(proj), b, v, j, h, w, 100.0, 1, 384.0 / w, 384.0 / h,
            _cabi.ptr(hm_out), _cabi.ptr(kp_img), _cabi.ptr(conf_n), _cabi.ptr(kp3d), _cabi.ptr(kp_hm), _cabi.stream(dev))
    for _ in range(50):
        lib.mvx_algebraic_forward(*args)
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(launches):
        t0 = time.perf_counter()
        lib.mvx_algebraic_forward(*args)
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    return wrapper + (float(np.percentile(ts, 50) * 1e6),)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="algebraic", choices=["algebraic", "volumetric", "stream16", "ransac"])
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: the workload's)")
    ap.add_argument("--no-writeback", action="store_true", help="algebraic: skip the normalised-heatmap output")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MVX_BENCH_REHEARSAL=1 rehearses the multi-rank path on a single GPU: every rank uses cuda:0 and the
    # collectives run over gloo on host copies (RCCL needs one device per rank).  Not a measurement mode.
    rehearsal = os.environ.get("MVX_BENCH_REHEARSAL") == "1"
    if world > 1:
        dist.init_process_group("gloo" if rehearsal else "nccl", init_method="env://")
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    _cabi.load()  # fails loudly if libmvx.so was not built

    extra = {}
    if args.workload == "algebraic":
        ms, frames, roof, cfg, extra = run_algebraic(args, dev, rank, world, 4, args.frames or 4096, not args.no_writeback)
        metric = "triangulated frames/sec (4-view, 17-joint)"
    elif args.workload == "volumetric":
        ms, frames, roof, cfg, extra = run_volumetric(args, dev, rank, world, args.frames or 64)
        metric = "volumetric frames/sec (4-view unproject to 64^3 + 3-D soft-argmax, 17 joints)"
    elif args.workload == "ransac":
        ms, frames, roof, cfg, extra = run_ransac(args, dev, rank, world, args.frames or 4096)
        metric = "RANSAC-triangulated frames/sec (4-view, 17-joint)"
    else:
        ms, frames, roof, cfg, extra = run_algebraic(args, dev, rank, world, 16, args.frames or 1024, not args.no_writeback)
        metric = "triangulated frames/sec (16-view, 17-joint)"
        if rank == 0:
            p50, p99, c_p50 = latency_p50(dev)
            extra["batch1_16view_latency_p50_us"] = p50
            extra["batch1_16view_latency_p99_us"] = p99
            extra["batch1_16view_c_abi_latency_p50_us"] = c_p50

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        if args.workload == "volumetric":
            cpu = cpu_baseline_volumetric(4)
        elif args.workload == "ransac":
            cpu = cpu_baseline_ransac(16)
        else:
            cpu = cpu_baseline_algebraic(16 if args.workload == "stream16" else 4, 64 if args.workload == "stream16" else 256)

    if ALL_GATHER_MS is not None:
        extra["keypoint_all_gather_alone_ms"] = ALL_GATHER_MS
    if rank == 0:
        line = {"metric": metric, "value": frames * world / (ms * 1e-3), "unit": "frames/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg,
                "roofline": roof, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
