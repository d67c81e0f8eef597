"""Development tool: where the batch-1 latency goes (wrapper vs raw C-ABI calls vs kernels)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poseestimation_amd import _cabi, synth  # noqa: E402
from poseestimation_amd.mvn.models.triangulation import AlgebraicTriangulationNet  # noqa: E402

dev = torch.device("cuda:0")
lib = _cabi.load()
for V in (4, 16):
    case = synth.device_algebraic_case(1, 1, V, dev)
    hm, conf, proj = case["heatmaps"], case["confidences"], case["proj"]
    net = AlgebraicTriangulationNet.__new__(AlgebraicTriangulationNet)
    torch.nn.Module.__init__(net)
    net.use_confidences, net.heatmap_softmax, net.heatmap_multiplier = True, True, 100.0
    b, v, j, h, w = hm.shape
    kp_hm = torch.empty((b * v, j, 2), device=dev)
    hm_out = torch.empty_like(hm)
    kp_img = torch.empty((b, v, j, 2), device=dev)
    conf_n = torch.empty((b, v, j), device=dev)
    kp3d = torch.empty((b, j, 3), device=dev)
    st = _cabi.stream(dev)

    def raw():
        lib.mvx_soft_argmax_2d(_cabi.ptr(hm), b * v * j, h, w, 100.0, 1, _cabi.ptr(kp_hm), _cabi.ptr(hm_out), st)
        lib.mvx_algebraic_glue_dlt(_cabi.ptr(proj), _cabi.ptr(kp_hm), _cabi.ptr(conf), b, v, j, 4.0, 4.0,
                                   _cabi.ptr(kp_img), _cabi.ptr(conf_n), _cabi.ptr(kp3d), st)

    def wrapped():
        net.triangulate_from_heatmaps(hm, conf, proj, (384, 384))

    def p50(fn, n=2000):
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return np.percentile(ts, 50) * 1e6, np.percentile(ts, 99) * 1e6

    def gpu_only(fn, n=200):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) * 1e3 / n

    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st_side = _cabi.stream(dev)

        def raw_side():
            lib.mvx_soft_argmax_2d(_cabi.ptr(hm), b * v * j, h, w, 100.0, 1, _cabi.ptr(kp_hm), _cabi.ptr(hm_out), st_side)
            lib.mvx_algebraic_glue_dlt(_cabi.ptr(proj), _cabi.ptr(kp_hm), _cabi.ptr(conf), b, v, j, 4.0, 4.0,
                                       _cabi.ptr(kp_img), _cabi.ptr(conf_n), _cabi.ptr(kp3d), st_side)
        raw_side()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            raw_side()
    torch.cuda.synchronize()
    before = kp3d.clone()
    kp3d.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(before, kp3d)
    print("V=%d graph replay: p50 %.1f us p99 %.1f us, back-to-back %.1f us" % ((V,) + p50(graph.replay) + (gpu_only(graph.replay),)))
    print("V=%d raw C calls: p50 %.1f us p99 %.1f us | wrapper: p50 %.1f us p99 %.1f us | back-to-back raw %.1f us, wrapper %.1f us per frame"
          % ((V,) + p50(raw) + p50(wrapped) + (gpu_only(raw), gpu_only(wrapped))))
    t0 = time.perf_counter()
    for _ in range(2000):
        torch.cuda.synchronize()
    print("  empty synchronize: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))

# floor: one trivial launch (a 1-voxel coordinate volume) + synchronize, and two of them
one = torch.zeros(3, device=dev)
rot1 = torch.eye(3, device=dev).reshape(1, 9).contiguous()
out1 = torch.empty(3, device=dev)
st = _cabi.stream(dev)


def tiny():
    lib.mvx_build_coord_volumes(_cabi.ptr(one), _cabi.ptr(one), _cabi.ptr(rot1), 1, 1, 1, 1, 1.0, 1.0, 1.0, 0, _cabi.ptr(out1), st)


def tiny2():
    tiny()
    tiny()


for name, fn in (("one trivial launch", tiny), ("two trivial launches", tiny2)):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(2000):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("  %s + synchronize: p50 %.1f us p99 %.1f us" % (name, np.percentile(ts, 50) * 1e6, np.percentile(ts, 99) * 1e6))
