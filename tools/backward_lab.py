"""Development tool: times mvx_unproject_backward and mvx_soft_argmax_3d_backward alone (no autograd around them),
for one or more builds of the library.    python tools/backward_lab.py <frames> <aggregation 0..3> lib.so [lib.so ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poseestimation_amd import _cabi, synth  # noqa: E402

frames, agg = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
case = synth.volumetric_case(11, min(frames, 8), 4)
rep = (frames + 7) // 8
up = lambda k: torch.from_numpy(case[k]).to(dev).repeat((rep,) + (1,) * (case[k].ndim - 1))[:frames].contiguous()
feats, proj, coord, vconf, vols = up("features"), up("proj"), up("coord_volumes"), up("vol_confidences"), up("volumes")
feats_cl = feats.permute(0, 1, 3, 4, 2).contiguous()
g_vol = torch.randn(frames, 32, 64, 64, 64, device=dev)
g_cl = torch.empty_like(feats_cl)
g_conf = torch.empty_like(vconf)
probs = torch.softmax(vols.view(frames, 17, -1), dim=2).view_as(vols)
g_kp, g_p, g_in = torch.randn(frames, 17, 3, device=dev), torch.randn_like(vols), torch.empty_like(vols)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for path in sys.argv[3:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("mvx_unproject_backward", "mvx_soft_argmax_3d_backward", "mvx_unproject_backward_workspace_bytes",
                 "mvx_soft_argmax_3d_backward_workspace_bytes"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = _cabi.SIGNATURES[name]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = lib.mvx_unproject_backward_workspace_bytes(frames, 4, 32, 96, 96, 64, 64, 64, agg)
    ws = torch.empty(max(nb, 16) // 4, dtype=torch.float32, device=dev)
    nb2 = lib.mvx_soft_argmax_3d_backward_workspace_bytes(frames, 17, 64, 64, 64)
    ws2 = torch.empty(max(nb2, 8) // 8, dtype=torch.float64, device=dev)

    def unproject():
        rc = lib.mvx_unproject_backward(P(feats_cl), P(proj), P(coord), P(vconf), P(g_vol), frames, 4, 32, 96, 96, 64, 64, 64,
                                        agg, P(g_cl), P(g_conf) if agg == 1 else None, P(ws), ws.numel() * 4, st)
        assert rc == 0, rc

    def argmax():
        rc = lib.mvx_soft_argmax_3d_backward(P(probs), P(coord), frames, 17, 64, 64, 64, 1.0, 1, P(g_kp), P(g_p), P(g_in), P(ws2), ws2.numel() * 8, st)
        assert rc == 0, rc

    for what, fn in (("unproject backward", unproject), ("soft-argmax 3-D backward", argmax)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        print("%-40s %-26s %8.3f ms  (%d frames, aggregation %d)" % (os.path.basename(path), what, a.elapsed_time(b) / 10, frames, agg),
              flush=True)
