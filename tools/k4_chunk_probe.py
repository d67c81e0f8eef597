"""Development tool: 3-D soft-argmax with write-back over 64 frames, called in slices of F frames, to see whether
the write-back pass re-reads a slice from the Infinity Cache when the slice is small enough."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseestimation_amd import _cabi, synth
dev = torch.device("cuda:0")
case = synth.device_volumetric_case(5, 64, 4, dev)
v, c = case["volumes"], case["coord_volumes"]
lib = _cabi.load()
B, J, S = v.shape[0], v.shape[1], v.shape[2]
out = torch.empty((B, J, 3), device=dev)
vout = torch.empty_like(v)
ws_bytes = lib.mvx_soft_argmax_3d_workspace_bytes(B, J, S, S, S)
ws = torch.empty(ws_bytes // 4 + 64, device=dev)
st = _cabi.stream(dev)
def run(F):
    for f0 in range(0, B, F):
        n = min(F, B - f0)
        rc = lib.mvx_soft_argmax_3d(_cabi.ptr(v[f0:]), _cabi.ptr(c[f0:]), n, J, S, S, S, 1.0, 1, _cabi.ptr(out[f0:]),
                                    _cabi.ptr(vout[f0:]), _cabi.ptr(ws), ws_bytes, st)
        assert rc == 0
ref = None
for F in (64, 32, 16, 12, 8, 6, 4, 2):
    run(F); torch.cuda.synchronize()
    if ref is None: ref = vout.clone()
    else: assert torch.allclose(ref, vout, rtol=1e-5, atol=1e-12)
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(F); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("slices of %2d frames: %.3f ms per 64 frames" % (F, np.median(ts)))
