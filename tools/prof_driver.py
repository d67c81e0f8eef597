"""Development tool: launches one kernel family a few times so that a profiler sees little else.

    python3 tools/prof_driver.py k1|k2|k3|k4|k4w|k3m|k4m [launches]     (k3m / k4m: as the volumetric model calls them)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poseestimation_amd import _cabi  # noqa: E402
from poseestimation_amd.mvn.utils import multiview, op  # noqa: E402

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
_cabi.load()
g = torch.Generator(device=dev)
g.manual_seed(1)

if which == "k1":
    frames = int(os.environ.get("FRAMES", "4096"))
    hm = torch.empty((frames * 4, 17, 96, 96), device=dev).normal_(generator=g)
    for _ in range(n):
        op.integrate_tensor_2d(hm, True, multiplier=100.0)
elif which == "argmax":
    frames = int(os.environ.get("FRAMES", "4096"))
    hm = torch.empty((frames, 4, 17, 96, 96), device=dev).normal_(generator=g)
    kp = torch.empty((frames, 4, 17, 2), dtype=torch.int64, device=dev)
    for _ in range(n):
        _cabi.load().mvx_heatmap_argmax(_cabi.ptr(hm), frames * 4 * 17, 96, 96, 4.0, 4.0, _cabi.ptr(kp), _cabi.stream(dev))
elif which == "k2":
    P = torch.empty((4096, 4, 3, 4), device=dev).normal_(generator=g)
    P[..., 2, 3] += 4000.0
    pts = torch.empty((4096, 4, 17, 2), device=dev).uniform_(0, 384, generator=g)
    for _ in range(n):
        multiview.triangulate_batch_of_points(P, pts)
elif which in ("k3", "k4", "k4w"):
    from poseestimation_amd import synth
    frames = int(os.environ.get("FRAMES", "16"))
    case = synth.device_volumetric_case(9, frames, 4, dev)
    for _ in range(n):
        if which == "k3":
            op.unproject_heatmaps(case["features"], case["proj"], case["coord_volumes"], os.environ.get("METHOD", "softmax"),
                                  case["vol_confidences"])
        else:
            op.integrate_tensor_3d_with_coordinates(case["volumes"], case["coord_volumes"], return_volumes=(which == "k4w"))
elif which in ("k3m", "k4m"):
    import bench
    from poseestimation_amd import synth
    frames = int(os.environ.get("FRAMES", "64"))
    case = synth.device_volumetric_case(9, frames, 4, dev)
    coord = bench.library_grid(torch.from_numpy(case["base_points"]), dev)
    feats = case["features"]
    b, v, c, h, w = feats.shape
    fcl = feats.view(b * v, c, h, w).contiguous(memory_format=torch.channels_last).view(b, v, c, h, w)
    for _ in range(n):
        if which == "k3m":
            op.unproject_heatmaps(fcl, case["proj"], coord, "softmax", case["vol_confidences"])
        else:
            op.integrate_tensor_3d_with_coordinates(case["volumes"], coord, True)
torch.cuda.synchronize()
print("done", which)
