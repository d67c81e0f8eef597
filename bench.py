"""Throughput of the multi-view triangulation path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload algebraic|volumetric|stream16|ransac|sharded]

A step is one pass of the path over one resident batch of synthetic frames:
  algebraic  (default: the headline configuration) 4096 frames x 4 views x 17 joints of 96x96 heatmaps per GPU:
             soft-argmax (multiplier folded in, normalised heatmaps written back as the reference returns them)
             -> confidence normalisation + keypoint upscale + DLT.
  volumetric 64 frames: 4 views x 32-channel 96x96 features unprojected into 64^3 (softmax
             aggregation), 17 volumes per frame soft-argmaxed with coordinates (normalised volumes written).
  stream16   16 views: p50 latency of one frame and throughput at 1024 frames.
  ransac     the reference's third triangulation model on the algebraic shapes: arg-max keypoints of 4096 x 4 x 17
             heatmaps, then RANSAC over every view pair + DLT over the inliers + Huber refinement per joint.
  sharded    65 536 frames in all, split into contiguous shards over the ranks (strong scaling): every frame goes
             through the algebraic leg and then the volumetric leg (cuboid centred on the pelvis the algebraic leg
             found), and one all-gather returns both sets of keypoints to every rank.
`--gpus N` with N > 1 and no launcher around it starts the N ranks itself (fresh interpreters, one per GPU, before
this process has touched the GPU); under torch.distributed.run the ranks it started are used as they are.  With
N > 1 the algebraic / volumetric / ransac workloads give every rank its own batch (weak scaling) and end each step
with the RCCL all-gather of the (frames, 17, 3) keypoints, which runs on RCCL's stream while the next step computes;
every all-gather has completed before the clock stops.  Rank 0 prints one JSON line; `value` is the whole-job
frames/s with inputs resident in HBM.  The default run also measures the other configurations once the headline is
done and reports them under extra.configs (skip with --no-extra).
"""
import argparse
import copy
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from poseestimation_amd import _cabi, launch, sharding, synth  # noqa: E402
from poseestimation_amd.mvn.models.triangulation import AlgebraicTriangulationNet  # noqa: E402
from poseestimation_amd.mvn.utils import op  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak
ALL_GATHER_MS = None    # one keypoint all-gather on its own (process group up), measured after the timed loop
LAST_GATHERED = None    # sharded workload: the keypoints every rank holds after the last pass


def collective():
    """True when a process group is up: the keypoint all-gather then runs, also in a group of one rank (a launcher
    that starts a single rank exercises the same RCCL calls as N ranks do)."""
    return dist.is_available() and dist.is_initialized()


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


KERNEL_SOURCES = {   # the files a kernel is compiled from: a counter record is valid for exactly these bytes
    "soft_argmax_2d_reg<9>": ("soft_argmax_2d.hip", "mvx_common.h"),
    "heatmap_argmax<9>": ("ransac.hip", "mvx_common.h"),
    "unproject": ("unproject.hip", "unproject_taps.h", "grid_coords.h", "mvx_common.h"),
    "soft_argmax_3d": ("soft_argmax_3d.hip", "grid_coords.h", "mvx_common.h"),
}


def kernel_source_hash(kernel):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel]:
        with open(os.path.join(ROOT, "poseestimation_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_traffic(kernel, shape):
    """HBM bytes per launch of `kernel` from the rocprofv3 counter passes stored under profiles/ (collected
    with the same launch shape in a separate profiled run: counters cannot be read from inside this process).
    A record counts only if it was taken on the kernel sources this library was built from (their SHA-256 is
    stored with it) and on the same launch shape; otherwise the traffic is reported as unknown."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[kernel]
    except (OSError, KeyError, ValueError):
        return None, "no counter record for %s" % kernel
    if rec.get("shape") != shape:
        return None, "counter record is for another launch shape (%s)" % (rec.get("shape"),)
    try:
        if rec.get("source_sha256") != kernel_source_hash(kernel):
            return None, "stale: the kernel sources changed after the counter passes (%s)" % rec.get("source")
    except OSError:
        return None, "kernel sources not found"
    return rec["hbm_bytes_per_launch"], rec["source"]


def moved_bytes(roof, nbytes, note):
    """Adds to a roofline record the same rate on the bytes the launch has to move (inputs it reads + outputs it
    writes, each once), where that differs from the counted algorithmic bytes."""
    roof["bytes_moved_per_launch"] = int(nbytes)
    roof["achieved_on_bytes_moved"] = nbytes / (roof["avg_launch_ms"] * 1e-3) / 1e9
    roof["frac_on_bytes_moved"] = roof["achieved_on_bytes_moved"] / roof["peak"]
    roof["bytes_moved_note"] = note


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
    moved_bytes(roof, bytes_per_launch, "equal to the algorithmic bytes: every map is read once and written once")
    roof["traffic"], roof["traffic_source"] = measured_traffic(
        "soft_argmax_2d_reg<9>", {"maps": frames * v * j, "map": [h, w], "heatmap_writeback": writeback})  # the launch sees maps only
    extra = {"dlt_kernel_ms": k2.mean_ms(), "max_abs_error_vs_synthetic_truth_mm": float(err)}
    cfg = {"workload": "algebraic: soft-argmax 2D + confidence-weighted DLT, %d frames/GPU x %d views x 17 joints, 96x96 heatmaps"
                       % (frames, v), "frames_per_gpu": frames, "views": v, "joints": j, "heatmap_writeback": writeback,
           "timed_step": "the two C-ABI calls of AlgebraicTriangulationNet.triangulate_from_heatmaps (mvx_soft_argmax_2d, "
                         "mvx_algebraic_glue_dlt) with caller-owned outputs; bit-equality with the class path is asserted "
                         "on 8 frames before timing",
           "parallelism": "frames sharded over %d GPU(s), all-gather of keypoints" % world}
    return ms, frames, roof, cfg, extra


def torch_unproject_softmax(feats, proj, coord, frame, voxels):
    """Float64 torch evaluation on the device of the unprojection of `frame` at the flat voxel indices `voxels`
    (projection, the reference's pixel mapping ix = x / H * (W - 1), iy = y / W * (H - 1), bilinear taps with zero
    padding, zero behind the camera, softmax over views): an independent check of the kernel, not the oracle."""
    v, c, h, w = feats.shape[1:]
    pts = coord[frame].reshape(-1, 3)[voxels].double()
    pts = torch.cat([pts, torch.ones_like(pts[:, :1])], dim=1)
    per_view = []
    for vi in range(v):
        uvw = pts @ proj[frame, vi].double().t()
        behind = uvw[:, 2] <= 0
        z = torch.where(uvw[:, 2] == 0, torch.ones_like(uvw[:, 2]), uvw[:, 2])
        ix, iy = uvw[:, 0] / z / h * (w - 1), uvw[:, 1] / z / w * (h - 1)
        x0, y0 = torch.floor(ix), torch.floor(iy)
        img = feats[frame, vi].double()
        acc = torch.zeros((c, pts.shape[0]), dtype=torch.float64, device=feats.device)
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi = x0 + dx, y0 + dy
                wgt = (1 - (ix - xi).abs()) * (1 - (iy - yi).abs())
                ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1) & ~behind
                tap = img[:, yi.clamp(0, h - 1).long(), xi.clamp(0, w - 1).long()]
                acc += tap * (wgt * ok).unsqueeze(0)
        per_view.append(acc)
    stack = torch.stack(per_view)
    return (stack * torch.softmax(stack, dim=0)).sum(0)


def check_volumetric(feats, proj, coord, vols, unprojected, keypoints, frames_to_check):
    """The volumes and keypoints of a few frames against float64 torch on the device."""
    n_vox = coord.shape[1] * coord.shape[2] * coord.shape[3]
    voxels = torch.arange(0, n_vox, 997, device=feats.device)
    worst_u = worst_k = 0.0
    for f in frames_to_check:
        want = torch_unproject_softmax(feats, proj, coord, f, voxels)
        got = unprojected[f].reshape(unprojected.shape[1], -1)[:, voxels].double()
        worst_u = max(worst_u, float((got - want).abs().max()))
        p = torch.softmax(vols[f].reshape(vols.shape[1], -1).double(), dim=1)
        want_k = (p.unsqueeze(-1) * coord[f].reshape(1, -1, 3).double()).sum(1)   # (no GEMM: keeps the kernel profile clean)
        worst_k = max(worst_k, float((keypoints[f].double() - want_k).abs().max()))
    assert worst_u < 1e-4, "unprojected volumes are %g from the float64 evaluation" % worst_u
    assert worst_k < 1e-3, "volumetric keypoints are %g mm from the float64 evaluation" % worst_k
    return worst_u, worst_k


def library_grid(base_points, dev, side=2500.0, size=64):
    """The coordinate volumes of `base_points` (n, 3) as VolumetricTriangulationNet.build_coord_volumes makes them in
    eval mode: written by mvx_build_coord_volumes and tagged with their grid parameters, so that the ops evaluate
    the grid in their kernels instead of reading the tensor back."""
    lib = _cabi.load()
    n = base_points.shape[0]
    center = base_points.to(device=dev, dtype=torch.float32).contiguous()
    position = (center - side / 2).contiguous()
    rot = torch.eye(3, dtype=torch.float32, device=dev).reshape(1, 9).repeat(n, 1).contiguous()
    step = float(np.float32(side / (size - 1)))
    coord = torch.empty((n, size, size, size, 3), dtype=torch.float32, device=dev)
    rc = lib.mvx_build_coord_volumes(_cabi.ptr(position), _cabi.ptr(center), _cabi.ptr(rot), n, size, size, size, step, step,
                                     step, 0, _cabi.ptr(coord), _cabi.stream(dev))
    if rc != 0:
        raise RuntimeError("mvx_build_coord_volumes failed: %s" % lib.mvx_last_error().decode())
    return op.attach_grid(coord, op.VoxelGrid(position, center, rot, (step,) * 3, 0, (size,) * 3))


def run_volumetric(args, dev, rank, world, frames=64):
    """The two ops as VolumetricTriangulationNet.forward calls them: feature maps channel-last in memory (what its
    1 x 1 convolution leaves) and coordinate volumes it built itself (grid parameters attached)."""
    case = synth.device_volumetric_case(1234 + 2 + rank, frames, 4, dev)
    feats, proj, vconf, vols = (case[k] for k in ("features", "proj", "vol_confidences", "volumes"))
    coord = library_grid(torch.from_numpy(case["base_points"]), dev)
    b, v, c, h, w = feats.shape
    fcl = feats.view(b * v, c, h, w).contiguous(memory_format=torch.channels_last).view(b, v, c, h, w)
    k3 = EventTimer()
    k4 = EventTimer()
    sg = StepGather((frames, 17, 3), dev, world)

    def step(timed):
        dst = sg.buffer()
        if timed:
            with k3:
                u = op.unproject_heatmaps(fcl, proj, coord, "softmax", vconf)
            with k4:
                kp, _ = op.integrate_tensor_3d_with_coordinates(vols, coord, True)
        else:
            u = op.unproject_heatmaps(fcl, proj, coord, "softmax", vconf)
            kp, _ = op.integrate_tensor_3d_with_coordinates(vols, coord, True)
        dst.copy_(kp)
        sg.submit()
        return u

    # sanity before timing: frames from every part of the batch against float64 torch on the device
    u = step(False)
    own = sg.gathered()[rank * frames: (rank + 1) * frames] if world > 1 else sg.gathered()
    err_u, err_k = check_volumetric(feats, proj, coord, vols, u, own, sorted({0, frames // 7, frames // 2, frames - 1}))
    # the plain-API form of the same call (maps [view][channel][h][w], an untagged coordinate tensor) gives the same bits
    plain = coord.clone()
    assert torch.equal(op.unproject_heatmaps(feats, proj, plain, "softmax", vconf), u)
    del u
    ms = timed_loop(step, args, dev, world, sg)
    api = EventTimer()
    for _ in range(5):
        with api:
            op.unproject_heatmaps(feats, proj, plain, "softmax", vconf)
    torch.cuda.synchronize(dev)
    n_vox = 64 ** 3
    b3 = frames * (4 * 32 * 96 * 96 * 4 + n_vox * 12 + 32 * n_vox * 4)
    b4 = frames * (17 * n_vox * 4 * 2 + n_vox * 12)
    roof = {"bound": "hbm", "kernel": "unproject_brick32<softmax> (channel-last maps, grid parameters)",
            "achieved": b3 / (k3.mean_ms() * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "algorithmic_bytes_per_launch": b3, "avg_launch_ms": k3.mean_ms()}
    roof["frac"] = roof["achieved"] / roof["peak"]
    moved_bytes(roof, b3 - frames * n_vox * 12, "the grid-parameter call does not read the %d B per frame of voxel coordinates "
                                               "that the algorithmic figure counts" % (n_vox * 12))
    roof["traffic"], roof["traffic_source"] = measured_traffic(
        "unproject", {"frames": frames, "views": 4, "method": "softmax", "maps": "channel-last", "coordinates": "grid parameters"})
    t4, t4_source = measured_traffic(
        "soft_argmax_3d", {"frames": frames, "volumes": 17, "grid": 64, "write_volumes": True, "coordinates": "grid parameters"})
    extra = {"soft_argmax_3d_ms": k4.mean_ms(), "soft_argmax_3d_GBps": b4 / (k4.mean_ms() * 1e-3) / 1e9,
             "soft_argmax_3d_frac_of_hbm_peak": b4 / (k4.mean_ms() * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "soft_argmax_3d_frac_on_bytes_moved": (b4 - frames * n_vox * 12) / (k4.mean_ms() * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "soft_argmax_3d_traffic": t4, "soft_argmax_3d_traffic_source": t4_source,
             "soft_argmax_3d_algorithmic_bytes_per_launch": b4,
             "unproject_plain_api_ms": api.mean_ms(),
             "unproject_plain_api_note": "maps [view][channel][h][w] (layout copy inside) and an untagged coordinate tensor",
             "max_abs_error_unproject_vs_float64_torch": err_u, "max_abs_error_keypoints_vs_float64_torch_mm": err_k}
    cfg = {"workload": "volumetric: unproject 4 views x 32ch 96x96 -> 64^3 (softmax) + 3-D soft-argmax of 17 volumes, %d frames/GPU"
                       % frames, "frames_per_gpu": frames, "parallelism": "frames sharded over %d GPU(s)" % world}
    return ms, frames, roof, cfg, extra


def cpu_baseline_training(sample_frames):
    """Forward + backward of the two volumetric ops in the torch-CPU port (autograd through grid_sample, softmax and
    einsum, the reference's formulation) on a bounded sample, host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_port
    threads = torch.get_num_threads()
    case = synth.volumetric_case(1234 + 5, sample_frames, 4)
    f, p, c, vc, vol = (torch.from_numpy(case[k]) for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    up_u = torch.randn(sample_frames, 32, 64, 64, 64)
    up_k, up_p = torch.randn(sample_frames, 17, 3), torch.randn_like(vol)
    ts = []
    for i in range(2):
        t0 = time.perf_counter()
        fg = f.clone().requires_grad_(True)
        torch_port.unproject(fg, p, c, "softmax", vc).backward(up_u)
        vg = vol.clone().requires_grad_(True)
        torch.autograd.backward(list(torch_port.soft_argmax_3d_coords(vg, c)), [up_k, up_p])
        if i >= 1:
            ts.append(time.perf_counter() - t0)
    secs = float(np.median(ts))
    return {"value": sample_frames / secs, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frame(s), forward + backward of unproject (softmax) and 3-D soft-argmax, second of 2 passes" % sample_frames}


def run_training(args, dev, rank, world, frames=8):
    """Forward + backward of the two volumetric ops as a training step of VolumetricTriangulationNet runs them
    (train.py:281-290; a batch of a few frames): gradients of the unprojected volumes flow to the feature maps, those
    of the keypoints and of the normalised volumes to the volume net's output."""
    case = synth.device_volumetric_case(1234 + 5 + rank, frames, 4, dev)
    feats, proj, vconf, vols = (case[k] for k in ("features", "proj", "vol_confidences", "volumes"))
    coord = library_grid(torch.from_numpy(case["base_points"]), dev)
    b, v, c, h, w = feats.shape
    fcl = feats.view(b * v, c, h, w).contiguous(memory_format=torch.channels_last).view(b, v, c, h, w)
    gen = torch.Generator(device=dev).manual_seed(77 + rank)
    up_u = torch.randn((frames, 32, 64, 64, 64), device=dev, generator=gen)
    up_k = torch.randn((frames, 17, 3), device=dev, generator=gen)
    up_p = torch.randn(vols.shape, device=dev, generator=gen)
    timers = {k: EventTimer() for k in ("unproject_forward", "unproject_backward", "soft_argmax_3d_forward",
                                        "soft_argmax_3d_backward")}
    sg = StepGather((frames, 17, 3), dev, world)
    grads = {}

    def step(timed):
        dst = sg.buffer()
        f = fcl.detach().requires_grad_(True)
        x = vols.detach().requires_grad_(True)
        if timed:
            with timers["unproject_forward"]:
                u = op.unproject_heatmaps(f, proj, coord, "softmax", vconf)
            with timers["unproject_backward"]:
                u.backward(up_u)
            with timers["soft_argmax_3d_forward"]:
                kp, prob = op.integrate_tensor_3d_with_coordinates(x, coord, True)
            with timers["soft_argmax_3d_backward"]:
                torch.autograd.backward([kp, prob], [up_k, up_p])
        else:
            u = op.unproject_heatmaps(f, proj, coord, "softmax", vconf)
            u.backward(up_u)
            kp, prob = op.integrate_tensor_3d_with_coordinates(x, coord, True)
            torch.autograd.backward([kp, prob], [up_k, up_p])
        grads["maps"], grads["volumes"] = f.grad, x.grad
        dst.copy_(kp.detach())
        sg.submit()

    # sanity before timing: with an upstream gradient that is zero outside a few voxels of one frame, the map gradient
    # is what autograd gives through the float64 torch evaluation of those voxels; the volume gradient of one frame
    # against float64 torch
    frame = frames // 2
    voxels = torch.arange(0, 64 ** 3, 997, device=dev)
    sparse = torch.zeros_like(up_u)
    sparse[frame].view(32, -1)[:, voxels] = up_u[frame].view(32, -1)[:, voxels]
    f = fcl.detach().requires_grad_(True)
    op.unproject_heatmaps(f, proj, coord, "softmax", vconf).backward(sparse)
    f64 = feats.double().requires_grad_(True)
    (torch_unproject_softmax(f64, proj, coord, frame, voxels) * up_u[frame].view(32, -1)[:, voxels].double()).sum().backward()
    scale = float(f64.grad.abs().max())
    err_maps = float((f.grad.double() - f64.grad).abs().max()) / scale
    assert err_maps < 1e-5, "map gradients are %g (relative) from float64 autograd" % err_maps
    x64 = vols[frame].double().requires_grad_(True)
    p64 = torch.softmax(x64.reshape(17, -1), dim=1)
    k64 = (p64.unsqueeze(-1) * coord[frame].reshape(1, -1, 3).double()).sum(1)
    ((k64 * up_k[frame].double()).sum() + (p64 * up_p[frame].reshape(17, -1).double()).sum()).backward()
    step(False)
    err_vol = float((grads["volumes"][frame].double() - x64.grad).abs().max() / x64.grad.abs().max())
    assert err_vol < 1e-5, "volume gradients are %g (relative) from float64 autograd" % err_vol

    ms = timed_loop(step, args, dev, world, sg)
    n_vox = 64 ** 3
    tb = timers["unproject_backward"].mean_ms()
    # what the backward has to move: the upstream gradient and the coordinates in, the map gradients out (the view
    # statistics it writes and reads back, 16 B per channel pair and voxel, are its own device)
    moved = frames * (32 * n_vox * 4 + n_vox * 12 + 2 * 4 * 32 * 96 * 96 * 4)
    roof = {"bound": "hbm", "kernel": "unproject_view_stats<softmax> + unproject_backward_maps<softmax>",
            "achieved": moved / (tb * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "traffic_source": "not collected", "algorithmic_bytes_per_launch": moved, "avg_launch_ms": tb,
            "note": "bound by LDS atomic additions (float64, 4-7 lanes per clock per CU) and by re-sampling the maps, not "
                    "by HBM: the fraction is reported for completeness"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    extra = {k + "_ms": t.mean_ms() for k, t in timers.items()}
    extra["timers_note"] = "HIP events around the op call and around autograd's backward call (kernels plus the Python around them)"
    extra["map_gradient_relative_error_vs_float64_autograd"] = err_maps
    extra["volume_gradient_relative_error_vs_float64_autograd"] = err_vol
    cfg = {"workload": "training step of the volumetric ops: forward + backward of unproject (4 views x 32ch 96x96 -> 64^3, "
                       "softmax) and of the 3-D soft-argmax of 17 volumes, %d frames/GPU" % frames,
           "frames_per_gpu": frames, "parallelism": "frames sharded over %d GPU(s)" % world}
    return ms, frames, roof, cfg, extra


def run_sharded(args, dev, rank, world, total=65536, alg_block=4096, vol_block=64):
    """BASELINE configs[3]: `total` frames in contiguous shards over the ranks, algebraic then volumetric leg per
    frame, one all-gather of both keypoint sets.  The shard is streamed through resident input blocks (4096 frames
    of heatmaps, 64 frames of features and volumes: synthetic inputs are re-used block after block, outputs are
    written per frame), so that a shard of any size fits in HBM; every kernel runs on every frame."""
    lo, hi = sharding.shard_bounds(total, rank, world)
    n_local = hi - lo
    alg_block = max(1, min(alg_block, n_local))
    vol_block = max(1, min(vol_block, n_local))
    lib = _cabi.load()
    st = _cabi.stream(dev)
    a = synth.device_algebraic_case(1234 + 3 + rank, alg_block, 4, dev)
    hm, conf, proj = a["heatmaps"], a["confidences"], a["proj"]
    _, v, j, h, w = hm.shape
    vcase = synth.device_volumetric_case(1234 + 3 + rank, vol_block, 4, dev)
    feats, proj_hm, vols = vcase["features"], vcase["proj"], vcase["volumes"]
    feats = feats.view(-1, 32, 96, 96).contiguous(memory_format=torch.channels_last).view(feats.shape)  # as the model leaves them
    assert op._is_channels_last(feats)
    S, side = 64, 2500.0
    n_vox = S ** 3
    # outputs: per-frame keypoints of both legs side by side (one collective moves both)
    kp_all = torch.zeros((n_local, 2 * j, 3), dtype=torch.float32, device=dev)
    kp3d_blk = torch.empty((alg_block, j, 3), dtype=torch.float32, device=dev)
    kp_hm = torch.empty((alg_block * v, j, 2), dtype=torch.float32, device=dev)
    hm_out = torch.empty_like(hm)
    kp_img = torch.empty((alg_block, v, j, 2), dtype=torch.float32, device=dev)
    conf_n = torch.empty((alg_block, v, j), dtype=torch.float32, device=dev)
    coord = torch.empty((vol_block, S, S, S, 3), dtype=torch.float32, device=dev)
    unproj = torch.empty((vol_block, 32, S, S, S), dtype=torch.float32, device=dev)
    vols_out = torch.empty_like(vols)
    vkp_blk = torch.empty((vol_block, j, 3), dtype=torch.float32, device=dev)
    ws4 = torch.empty(lib.mvx_soft_argmax_3d_workspace_bytes(vol_block, j, S, S, S) // 8 + 1, dtype=torch.float64, device=dev)
    rot = torch.eye(3, dtype=torch.float32, device=dev).reshape(1, 9).repeat(vol_block, 1).contiguous()
    step_mm = float(np.float32(side / (S - 1)))
    t_alg, t_unp, t_k4 = EventTimer(), EventTimer(), EventTimer()
    t_leg_a, t_leg_v = EventTimer(), EventTimer()

    def fail(rc):
        if rc != 0:
            raise RuntimeError("mvx call failed: %s" % lib.mvx_last_error().decode())

    def algebraic_leg(timed):
        for f0 in range(0, n_local, alg_block):
            nf = min(alg_block, n_local - f0)
            tm = t_alg if (timed and nf == alg_block) else None
            if tm is not None:
                tm.__enter__()
            rc = lib.mvx_soft_argmax_2d(_cabi.ptr(hm), nf * v * j, h, w, 100.0, 1, _cabi.ptr(kp_hm), _cabi.ptr(hm_out), st)
            if tm is not None:
                tm.__exit__()
            rc |= lib.mvx_algebraic_glue_dlt(_cabi.ptr(proj), _cabi.ptr(kp_hm), _cabi.ptr(conf), nf, v, j, 384.0 / w,
                                             384.0 / h, _cabi.ptr(kp_img), _cabi.ptr(conf_n), _cabi.ptr(kp3d_blk), st)
            fail(rc)
            kp_all[f0:f0 + nf, :j].copy_(kp3d_blk[:nf])

    def volumetric_leg(timed):
        for f0 in range(0, n_local, vol_block):
            nf = min(vol_block, n_local - f0)
            center = kp_all[f0:f0 + nf, synth.PELVIS_JOINT].contiguous()   # the pelvis the algebraic leg found
            position = (center - side / 2).contiguous()
            fail(lib.mvx_build_coord_volumes(_cabi.ptr(position), _cabi.ptr(center), _cabi.ptr(rot), nf, S, S, S, step_mm,
                                             step_mm, step_mm, 0, _cabi.ptr(coord), st))
            tm = t_unp if (timed and nf == vol_block) else None
            if tm is not None:
                tm.__enter__()
            rc = lib.mvx_unproject_grid(_cabi.ptr(feats), 1, _cabi.ptr(proj_hm), _cabi.ptr(position), _cabi.ptr(center),
                                        _cabi.ptr(rot), step_mm, step_mm, step_mm, 0, None, nf, 4, 32, 96, 96, S, S, S,
                                        _cabi.AGG_SOFTMAX, _cabi.ptr(unproj), None, 0, st)
            if tm is not None:
                tm.__exit__()
            fail(rc)
            tm = t_k4 if (timed and nf == vol_block) else None
            if tm is not None:
                tm.__enter__()
            rc = lib.mvx_soft_argmax_3d_grid(_cabi.ptr(vols), _cabi.ptr(position), _cabi.ptr(center), _cabi.ptr(rot), step_mm,
                                             step_mm, step_mm, 0, nf, j, S, S, S, 1.0, 1, _cabi.ptr(vkp_blk),
                                             _cabi.ptr(vols_out), _cabi.ptr(ws4), ws4.numel() * 8, st)
            if tm is not None:
                tm.__exit__()
            fail(rc)
            kp_all[f0:f0 + nf, j:].copy_(vkp_blk[:nf])

    gather_host = collective() and dist.get_backend() == "gloo"
    gathered = [None]

    def exchange():
        if not collective():
            gathered[0] = kp_all
        elif gather_host:
            gathered[0] = sharding.all_gather_keypoints(kp_all.cpu(), total)
        else:
            gathered[0] = sharding.all_gather_keypoints(kp_all, total)

    def step(timed):
        if timed:
            with t_leg_a:
                algebraic_leg(True)
            with t_leg_v:
                volumetric_leg(True)
        else:
            algebraic_leg(False)
            volumetric_leg(False)
        exchange()

    # sanity before timing: the first blocks against the drop-in class path and float64 torch on the device
    step(False)
    torch.cuda.synchronize(dev)
    net = algebraic_head()
    nchk = min(8, alg_block)
    # the 8-frame reference call would take the 16-lanes-per-point DLT kernel, whose last float32 bit may differ from
    # the lane-per-point kernel the resident blocks run: switch it off so that both sides run the same kernel
    lib.mvx_set_kernel_selection(_cabi.SELECT_NO_DLT_ROWS)
    try:
        ref3d = net.triangulate_from_heatmaps(hm[:nchk], conf[:nchk], proj[:nchk], (384, 384))[0]
        torch.cuda.synchronize(dev)
    finally:
        lib.mvx_set_kernel_selection(0)
    if alg_block * j > 2048:   # the first block's launch is beyond the 16-lanes-per-point kernel's size
        assert torch.equal(kp_all[:nchk, :j], ref3d)
    else:   # a shard this small runs the 16-lanes-per-point kernel itself
        assert torch.allclose(kp_all[:nchk, :j], ref3d, rtol=0.0, atol=1e-3)
    err = float((kp_all[:nchk, :j].cpu().numpy() - a["joints_3d"][:nchk]).__abs__().max())
    assert err < 80.0, "triangulated joints are %g mm from the synthetic ground truth" % err
    # the last volumetric block is still in coord / unproj / vkp_blk
    last0 = ((n_local - 1) // vol_block) * vol_block
    nlast = n_local - last0
    err_u, err_k = check_volumetric(feats[:nlast], proj_hm[:nlast], coord[:nlast], vols[:nlast], unproj[:nlast],
                                    vkp_blk[:nlast], sorted({0, nlast - 1}))
    full = gathered[0]
    assert tuple(full.shape) == (total, 2 * j, 3)
    assert torch.equal(full[lo:hi].to(dev), kp_all)
    assert bool(torch.isfinite(full).all())

    class _NoPipe:
        def drain(self):
            pass

        def exposed_ms(self, repeats=5):
            if not collective():
                return None
            ts = []
            for _ in range(repeats):
                dist.barrier()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                exchange()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts) * 1e3)

    ms = timed_loop(step, args, dev, world, _NoPipe())
    global LAST_GATHERED
    LAST_GATHERED = gathered[0]
    b3 = vol_block * (4 * 32 * 96 * 96 * 4 + n_vox * 12 + 32 * n_vox * 4)
    b4 = vol_block * (17 * n_vox * 4 * 2 + n_vox * 12)
    b1 = alg_block * v * j * (h * w * 4 * 2 + 8)
    roof = {"bound": "hbm", "kernel": "unproject_brick32<softmax> (channel-last maps, grid parameters), %d-frame blocks" % vol_block,
            "achieved": b3 / (t_unp.mean_ms() * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "algorithmic_bytes_per_launch": b3, "avg_launch_ms": t_unp.mean_ms()}
    roof["frac"] = roof["achieved"] / roof["peak"]
    moved_bytes(roof, b3 - vol_block * n_vox * 12, "the grid-parameter call does not read the %d B per frame of voxel "
                                                  "coordinates that the algorithmic figure counts" % (n_vox * 12))
    # the 64-frame block launch is the volumetric workload's launch: same kernel, shape, layout and coordinate source
    roof["traffic"], roof["traffic_source"] = measured_traffic(
        "unproject", {"frames": vol_block, "views": 4, "method": "softmax", "maps": "channel-last", "coordinates": "grid parameters"})
    extra = {"frames_this_rank": n_local, "algebraic_leg_ms": t_leg_a.mean_ms(), "volumetric_leg_ms": t_leg_v.mean_ms(),
             "soft_argmax_2d_GBps": b1 / (t_alg.mean_ms() * 1e-3) / 1e9,
             "soft_argmax_3d_GBps": b4 / (t_k4.mean_ms() * 1e-3) / 1e9,
             "max_abs_error_vs_synthetic_truth_mm": err, "max_abs_error_unproject_vs_float64_torch": err_u,
             "max_abs_error_keypoints_vs_float64_torch_mm": err_k}
    cfg = {"workload": "sharded: %d frames in all, algebraic (4 views x 17 joints, heatmap write-back) then volumetric "
                       "(unproject to 64^3 + 3-D soft-argmax) per frame, one all-gather of both keypoint sets" % total,
           "frames_total": total, "frames_per_gpu": [b - a_ for a_, b in (sharding.shard_bounds(total, r, world)
                                                                         for r in range(world))],
           "parallelism": "contiguous frame shards over %d GPU(s), all-gather of (frames, 34, 3) keypoints" % world}
    return ms, n_local, roof, cfg, extra


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


def run_ransac(args, dev, rank, world, frames=4096, view_pairs="sampled"):
    """view_pairs "sampled": the reference's hypotheses -- per (frame, joint) ten pairs from Python's `random`
    (mvn/models/triangulation.py:96-97), drawn for the next step on a host thread while the GPU runs the current one;
    "all": every pair, no host work (not what the reference does)."""
    import random

    from poseestimation_amd.mvn.models.triangulation import ViewPairPrefetcher, all_view_pairs
    case = synth.device_algebraic_case(1234 + 1 + rank, frames, 4, dev)
    hm, proj = case["heatmaps"], case["proj"]
    b, v, j, h, w = hm.shape
    lib = _cabi.load()
    prefetch = None
    if view_pairs == "sampled":
        random.seed(1234 + rank)
        prefetch = ViewPairPrefetcher(b * j, v, 10, dev)
        prefetch.start()
        n_iters = 10
        pairs = None
    else:
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

    def launch_ransac(kp3d, step_pairs):
        kp_f64.copy_(kp)
        rc = lib.mvx_triangulate_ransac(_cabi.ptr(proj), _cabi.ptr(kp_f64), _cabi.ptr(step_pairs), b, v, j, n_iters, 15.0, 1,
                                        _cabi.ptr(out), _cabi.ptr(masks), st)
        kp3d.copy_(out)
        return rc

    def step(timed):
        kp3d = sg.buffer()
        step_pairs = pairs
        if prefetch is not None:
            step_pairs = prefetch.take()   # this step's hypotheses (waits for the host draw if it is the slower side)
            prefetch.start()               # the next step's are drawn while the GPU works on this one
        if timed:
            with ka:
                rc = launch_argmax()
            with kr:
                rc |= launch_ransac(kp3d, step_pairs)
        else:
            rc = launch_argmax() | launch_ransac(kp3d, step_pairs)
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
    roof["traffic"], roof["traffic_source"] = measured_traffic(
        "heatmap_argmax<9>", {"maps": frames * v * j, "map": [h, w], "heatmap_writeback": False})
    extra = {"ransac_kernel_and_casts_ms": kr.mean_ms(), "max_abs_error_vs_synthetic_truth_mm": float(err),
             "mean_inlier_views": float(torch.bitwise_and(masks.view(-1, 1) >> torch.arange(4, device=dev), 1).sum(1).float().mean())}
    if prefetch is not None:
        prefetch.take()   # the draw begun by the last step
        # the host side alone: one draw of a step's pairs, not overlapped with anything
        t0 = time.perf_counter()
        prefetch.start()
        prefetch.take()
        torch.cuda.synchronize(dev)
        extra["view_pair_draw_ms_per_step"] = (time.perf_counter() - t0) * 1e3
        extra["direct_optimization_tolerance_note"] = (
            "with direct_optimization the refined points are checked against the reference at 2e-2 mm (scipy's "
            "least_squares stops short of the minimiser; the kernel's cost is never above the reference's), not 1e-3 mm")
    how = ("10 view pairs per joint sampled as the reference samples them (Python's generator state continued natively, "
           "drawn one step ahead on a host thread)" if view_pairs == "sampled" else "all 6 view pairs")
    cfg = {"workload": "ransac: heatmap arg-max + RANSAC (%s) + inlier DLT + Huber refinement, "
                       "%d frames/GPU x 4 views x 17 joints, 96x96 heatmaps" % (how, frames), "frames_per_gpu": frames,
           "views": v, "joints": j, "view_pairs": view_pairs,
           "parallelism": "frames sharded over %d GPU(s), all-gather of keypoints" % world}
    return ms, frames, roof, cfg, extra


class StepGather:
    """The all-gather that ends a step, overlapped with the next step: step k writes its keypoints into buffer
    k % 2 and its collective runs (on RCCL's stream) while step k + 1 computes; a buffer is written again only
    after the collective that reads it has finished.  `drain()` completes everything before timing stops.
    In the single-GPU rehearsal the same logic runs over gloo on host copies."""

    def __init__(self, local_shape, dev, world):
        self.world = world
        self.dev = dev
        self.host = collective() and dist.get_backend() == "gloo"
        self.buffers = [torch.empty(local_shape, dtype=torch.float32, device=dev) for _ in range(2)]
        self.pipe = None
        if collective():
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
    if collective():
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    gather.drain()   # every step's all-gather has completed inside the timed region
    if collective():
        dist.barrier()
    torch.cuda.synchronize(dev)
    secs = time.perf_counter() - t0
    if collective():
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
    out = {"batch1_16view_latency_p50_us": float(np.percentile(ts, 50) * 1e6),
           "batch1_16view_latency_p99_us": float(np.percentile(ts, 99) * 1e6)}
    # the class path with its per-shape output arena switched on (a frame loop that is done with one frame's results
    # before it asks for the next: nothing is allocated per call)
    net.reuse_outputs = True
    for _ in range(50):
        net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(launches):
        t0 = time.perf_counter()
        net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    out["batch1_16view_reuse_outputs_latency_p50_us"] = float(np.percentile(ts, 50) * 1e6)
    out["batch1_16view_reuse_outputs_latency_p99_us"] = float(np.percentile(ts, 99) * 1e6)
    # one frame of the real-time loop end to end (the reference's real_pipeline_trt.py:336-399 without the backbone and
    # the socket): soft-argmax + DLT, then the pose into the window, Savitzky-Golay filter, joint order / axis map /
    # scale, the two message frames copied to the host and turned into the wire dictionaries
    from poseestimation_amd.realtime import RealtimeTail
    tail = RealtimeTail()
    for _ in range(50):
        tail.push(net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))[0][0])
    ts = []
    for _ in range(launches):
        t0 = time.perf_counter()
        msgs = tail.push(net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))[0][0])
        ts.append(time.perf_counter() - t0)
    assert len(msgs) == 2 and len(msgs[0]) == 17
    out["frame_loop_16view_latency_p50_us"] = float(np.percentile(ts, 50) * 1e6)
    out["frame_loop_16view_latency_p99_us"] = float(np.percentile(ts, 99) * 1e6)
    net.reuse_outputs = False
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
    out["batch1_16view_c_abi_latency_p50_us"] = float(np.percentile(ts, 50) * 1e6)
    out["batch1_16view_c_abi_latency_p99_us"] = float(np.percentile(ts, 99) * 1e6)
    return out


METRICS = {
    "algebraic": "triangulated frames/sec (4-view, 17-joint)",
    "volumetric": "volumetric frames/sec (4-view unproject to 64^3 + 3-D soft-argmax, 17 joints)",
    "ransac": "RANSAC-triangulated frames/sec (4-view, 17-joint)",
    "stream16": "triangulated frames/sec (16-view, 17-joint)",
    "sharded": "triangulated frames/sec (4-view, 17-joint; algebraic + volumetric leg per frame, 65536 frames sharded)",
    "training": "volumetric frames/sec, forward + backward of unproject and 3-D soft-argmax (4-view, 64^3, 17 joints)",
}


def cpu_baseline_for(name, small=False):
    """The CPU port on a bounded sample of workload `name` (host cores of this box).  `small`: the shorter samples
    used for the configurations reported under extra.configs, so that the default run stays within minutes."""
    if name == "volumetric":
        return cpu_baseline_volumetric(2 if small else 4)
    if name == "ransac":
        return cpu_baseline_ransac(16)
    if name == "stream16":
        return cpu_baseline_algebraic(16, 64)
    if name == "training":
        return cpu_baseline_training(1)
    if name == "sharded":
        # per frame the path runs both legs: frames/s = 1 / (1/algebraic + 1/volumetric)
        a = cpu_baseline_algebraic(4, 64 if small else 256)
        v = cpu_baseline_volumetric(2 if small else 4)
        return {"value": 1.0 / (1.0 / a["value"] + 1.0 / v["value"]), "unit": "frames/s", "cores": a["cores"],
                "kind": "port", "sample": "both legs per frame, harmonic combination of: [%s] and [%s]" % (a["sample"], v["sample"])}
    return cpu_baseline_algebraic(4, 256)


def run_workload(name, args, dev, rank, world):
    """-> one result record: metric, value (whole-job frames/s), ms_per_step, scaling, config, roofline, extra."""
    global ALL_GATHER_MS
    ALL_GATHER_MS = None
    extra = {}
    scaling = "weak"
    if name == "algebraic":
        ms, frames, roof, cfg, extra = run_algebraic(args, dev, rank, world, 4, args.frames or 4096, not args.no_writeback)
    elif name == "volumetric":
        ms, frames, roof, cfg, extra = run_volumetric(args, dev, rank, world, args.frames or 64)
    elif name == "ransac":
        ms, frames, roof, cfg, extra = run_ransac(args, dev, rank, world, args.frames or 4096, "sampled")
        ms_all, frames_all, _, _, extra_all = run_ransac(with_steps(args, min(args.steps, 10), 2), dev, rank, world,
                                                         args.frames or 4096, "all")
        extra["all_view_pairs"] = {"value": frames_all * world / (ms_all * 1e-3), "unit": "frames/s", "ms_per_step": ms_all,
                                   "ransac_kernel_and_casts_ms": extra_all["ransac_kernel_and_casts_ms"],
                                   "note": "every pair instead of the reference's ten sampled ones: no host work"}
    elif name == "training":
        ms, frames, roof, cfg, extra = run_training(args, dev, rank, world, args.frames or 8)
    elif name == "sharded":
        total = args.frames or 65536
        ms, frames, roof, cfg, extra = run_sharded(args, dev, rank, world, total)
        scaling = "strong"
    else:
        ms, frames, roof, cfg, extra = run_algebraic(args, dev, rank, world, 16, args.frames or 1024, not args.no_writeback)
        if rank == 0:
            lat = latency_p50(dev)
            extra.update(lat)
    units = (args.frames or 65536) if name == "sharded" else frames * world
    if ALL_GATHER_MS is not None:
        extra["keypoint_all_gather_alone_ms"] = ALL_GATHER_MS
    return {"metric": METRICS[name], "value": units / (ms * 1e-3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg, "roofline": roof, "extra": extra}


def with_steps(args, steps, warmup):
    a = copy.copy(args)
    a.steps, a.warmup, a.frames = steps, warmup, 0
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="algebraic", choices=sorted(METRICS))
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (sharded: frames in all); default: the workload's")
    ap.add_argument("--no-writeback", action="store_true", help="algebraic: skip the normalised-heatmap output")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="default run: skip the other configurations after the headline")
    ap.add_argument("--dump-keypoints", default="", help="sharded: rank 0 saves the gathered (frames, 34, 3) keypoints "
                                                         "of the last pass to this .npy file")
    args = ap.parse_args()

    # MVX_BENCH_REHEARSAL=1 rehearses the multi-rank path on a single GPU: every rank uses cuda:0 and the
    # collectives run over gloo on host copies (RCCL needs one device per rank).  Not a measurement mode.
    rehearsal = os.environ.get("MVX_BENCH_REHEARSAL") == "1"
    if args.gpus > 1 and not launch.in_launched_rank():
        # No launcher around us: start the ranks ourselves.  This process makes no GPU query of any kind (each rank
        # checks for its own device below); it only relays rank 0's line and the ranks' exit status.
        sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    visible = torch.cuda.device_count()
    if not rehearsal and visible < (local_rank + 1 if launch.in_launched_rank() else 1):
        if rank == 0:
            print("bench.py: --gpus %d but %d GPU(s) visible" % (args.gpus, visible), file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if launch.in_launched_rank():
        # under a launcher the process group comes up for any world size, one rank included, as the reference's
        # training script does (train.py:424-436): the keypoint all-gather is then the same RCCL call at N = 1
        dist.init_process_group("gloo" if rehearsal else "nccl", init_method="env://")
    _cabi.load()  # fails loudly if libmvx.so was not built

    line = run_workload(args.workload, args, dev, rank, world)

    want_cpu = rank == 0 and not args.no_cpu_baseline and world == 1
    line["cpu_baseline"] = cpu_baseline_for(args.workload) if want_cpu else None

    # the other configurations of the path, measured after the headline so that one driver-timed run covers them
    plain_default = args.workload == "algebraic" and not args.no_extra and not args.frames and not args.no_writeback
    if plain_default:
        others = {}
        todo = [("sharded_65536_frames", "sharded", with_steps(args, 2, 1))]
        if world == 1:
            todo = [("volumetric_64_frames", "volumetric", with_steps(args, 20, 3)),
                    ("stream16", "stream16", with_steps(args, 20, 3)),
                    ("training_8_frames", "training", with_steps(args, 20, 5))] + todo
        for key, name, sub in todo:
            torch.cuda.empty_cache()
            rec = run_workload(name, sub, dev, rank, world)
            others[key] = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                               "scaling", "config", "roofline", "extra")}
            others[key]["cpu_baseline"] = cpu_baseline_for(name, small=True) if want_cpu else None
        line["extra"]["configs"] = others

    if args.dump_keypoints and rank == 0 and LAST_GATHERED is not None:
        np.save(args.dump_keypoints, LAST_GATHERED.detach().cpu().numpy())
    if rank == 0:
        print(json.dumps(line), flush=True)
    if collective():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
