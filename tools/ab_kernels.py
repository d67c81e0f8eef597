"""Development tool: times kernel variants (selected through MVX_* environment knobs read per call)
interleaved in one process on cuda:0 and prints median / min milliseconds per variant.

    python tools/ab_kernels.py k1|k3|k4 [rounds]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poseestimation_amd import _cabi, synth  # noqa: E402
from poseestimation_amd.mvn.utils import op  # noqa: E402

dev = torch.device("cuda:0")


def time_variants(fn, env_key, variants, rounds=10, inner=3):
    res = {v: [] for v in variants}
    outs = {}
    for v in variants:                      # warm-up + correctness snapshot
        os.environ[env_key] = str(v)
        outs[v] = fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in variants:
            os.environ[env_key] = str(v)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                fn()
            b.record()
            torch.cuda.synchronize()
            res[v].append(a.elapsed_time(b) / inner)
    base = outs[variants[0]]
    for v in variants:
        diff = max(float((x.float() - y.float()).abs().max()) for x, y in zip(outs[v], base))
        print("%s=%-8s median %.4f ms  min %.4f ms  max|diff vs first| %.3g" %
              (env_key, v, np.median(res[v]), np.min(res[v]), diff))
    os.environ.pop(env_key, None)


def k1(rounds):
    case = synth.device_algebraic_case(3, 4096, 4, dev)
    hm = case["heatmaps"].view(-1, 17, 96, 96)
    lib = _cabi.load()
    coords = torch.empty((hm.shape[0], 17, 2), device=dev)
    out = torch.empty_like(hm)

    def fn():
        lib.mvx_soft_argmax_2d(_cabi.ptr(hm), hm.shape[0] * 17, 96, 96, 100.0, 1, _cabi.ptr(coords), _cabi.ptr(out), _cabi.stream(dev))
        return coords.clone(), out[:64].clone()
    time_variants(fn, "MVX_SAM2D_BLOCKS_PER_CU", [2, 1, 3, 4, 8], rounds)
    gb = hm.numel() * 8 / 1e9
    print("bytes per launch: %.3f GB" % gb)

    def fn_ro():
        lib.mvx_soft_argmax_2d(_cabi.ptr(hm), hm.shape[0] * 17, 96, 96, 100.0, 1, _cabi.ptr(coords), None, _cabi.stream(dev))
        return (coords,)
    print("-- coords only (%.3f GB)" % (gb / 2))
    time_variants(fn_ro, "MVX_SAM2D_BLOCKS_PER_CU", [8, 4, 2], rounds)
    print("-- torch copy_ of the same bytes")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        a.record()
        out.copy_(hm)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("copy_ median %.4f ms min %.4f ms" % (np.median(ts), np.min(ts)))


def k3(rounds):
    case = synth.device_volumetric_case(4, 64, 4, dev)
    f, p, c, vc = case["features"], case["proj"], case["coord_volumes"], case["vol_confidences"]
    for method in ("softmax", "conf", "sum", "max"):
        print("--", method)
        time_variants(lambda: (op.unproject_heatmaps(f, p, c, method, vc),), "MVX_UNPROJ_PATH", ["tile", "generic"], rounds, inner=2)
    B, V, C, h, w = f.shape
    fcl = f.view(B * V, C, h, w).contiguous(memory_format=torch.channels_last).view(B, V, C, h, w)
    print("-- softmax, channel-last input")
    time_variants(lambda: (op.unproject_heatmaps(fcl, p, c, "softmax", vc),), "MVX_UNPROJ_PATH", ["tile"], rounds, inner=2)



def k4(rounds):
    case = synth.device_volumetric_case(5, 64, 4, dev)
    v, c = case["volumes"], case["coord_volumes"]
    print("-- with write-back")
    time_variants(lambda: op.integrate_tensor_3d_with_coordinates(v, c), "MVX_UNUSED", [0], rounds, inner=2)
    print("-- coords only")
    time_variants(lambda: (op.integrate_tensor_3d_with_coordinates(v, c, return_volumes=False)[0],), "MVX_UNUSED", [0], rounds, inner=2)


if __name__ == "__main__":
    which = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    {"k1": k1, "k3": k3, "k4": k4}[which](rounds)
