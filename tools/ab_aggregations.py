"""Development tool: the brick unprojection timed per aggregation method (64 frames, channel-last maps, grid parameters)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from poseestimation_amd import _cabi, synth  # noqa: E402
from poseestimation_amd.mvn.utils import op  # noqa: E402

dev = torch.device("cuda:0")
_cabi.load()
frames = 64
case = synth.device_volumetric_case(5, frames, 4, dev)
feats, proj, vconf = case["features"], case["proj"], case["vol_confidences"]
fcl = feats.view(frames * 4, 32, 96, 96).contiguous(memory_format=torch.channels_last).view(frames, 4, 32, 96, 96)
coord = bench.library_grid(torch.from_numpy(case["base_points"]), dev)
for method in ("sum", "max", "conf", "softmax"):
    fn = lambda: op.unproject_heatmaps(fcl, proj, coord, method, vconf)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("%-8s %.3f ms" % (method, float(np.median(ts))))
