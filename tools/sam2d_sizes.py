"""Development tool: 2-D soft-argmax bandwidth for map sizes that do and do not hit the register kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseestimation_amd import _cabi
dev = torch.device("cuda:0")
lib = _cabi.load()
for h, w in ((96, 96), (64, 64), (128, 128), (80, 80), (72, 96), (56, 56), (112, 112), (48, 64), (33, 31)):
    n = int(4e9 // (h * w * 4))
    hm = torch.randn((n, h, w), device=dev)
    out = torch.empty_like(hm)
    coords = torch.empty((n, 2), device=dev)
    for wb in (True, False):
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.mvx_soft_argmax_2d(_cabi.ptr(hm), n, h, w, 100.0, 1, _cabi.ptr(coords), _cabi.ptr(out) if wb else None, _cabi.stream(dev))
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            assert rc == 0
        gb = hm.numel() * 4 * (2 if wb else 1) / 1e9
        print("%3dx%-3d %s: %.0f GB/s" % (h, w, "write-back " if wb else "coords only", gb / (np.median(ts[1:]) * 1e-3)))
