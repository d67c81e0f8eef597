"""Times the volumetric training step's two operations, forward + backward, on one GPU: the mvx kernels through the
op mirror against the reference's formulation written with torch's own GPU ops (grid_sample per view, softmax,
einsum -- what the reference's op.py would run on this GPU).  Development tool; prints one JSON record.

    python tools/ab_backward.py [frames] [method]
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poseestimation_amd import synth  # noqa: E402
from poseestimation_amd.mvn.utils import op  # noqa: E402

DEV = torch.device("cuda:0")


def torch_unproject(features, proj, coord, method, vconf):
    b, v, c, h, w = features.shape
    vshape = tuple(coord.shape[1:4])
    out = []
    for bi in range(b):
        pts = coord[bi].reshape(-1, 3)
        pts_h = torch.cat([pts, torch.ones(pts.shape[0], 1, device=pts.device)], dim=1)
        stack = []
        for vi in range(v):
            uvw = pts_h @ proj[bi, vi].t()
            behind = uvw[:, 2] <= 0.0
            z = torch.where(uvw[:, 2] == 0.0, torch.ones_like(uvw[:, 2]), uvw[:, 2])
            g = torch.stack([2 * (uvw[:, 0] / z / h - 0.5), 2 * (uvw[:, 1] / z / w - 0.5)], dim=1)
            samp = F.grid_sample(features[bi, vi].unsqueeze(0), g.view(1, -1, 1, 2), align_corners=True).view(c, -1)
            stack.append(torch.where(behind[None], torch.zeros_like(samp), samp))
        stack = torch.stack(stack)
        if method.startswith("conf"):
            out.append((stack * vconf[bi].view(v, c, 1)).sum(0))
        elif method == "sum":
            out.append(stack.sum(0))
        elif method == "max":
            out.append(stack.max(0)[0])
        else:
            out.append((stack * F.softmax(stack, dim=0)).sum(0))
    return torch.stack(out).view(b, c, *vshape)


def torch_soft_argmax(volumes, coord):
    b, n = volumes.shape[:2]
    p = F.softmax(volumes.reshape(b, n, -1), dim=2).reshape(volumes.shape)
    return torch.einsum("bnxyz,bxyzc->bnc", p, coord), p


def timed(fn, steps=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    method = sys.argv[2] if len(sys.argv) > 2 else "softmax"
    case = synth.volumetric_case(11, frames, 4)
    feats = torch.from_numpy(case["features"]).to(DEV)
    proj = torch.from_numpy(case["proj"]).to(DEV)
    coord = torch.from_numpy(case["coord_volumes"]).to(DEV)
    vconf = torch.from_numpy(case["vol_confidences"]).to(DEV)
    vols = torch.from_numpy(case["volumes"]).to(DEV)
    up_vol = torch.randn(frames, 32, 64, 64, 64, device=DEV)
    up_kp = torch.randn(frames, 17, 3, device=DEV)
    up_p = torch.randn_like(vols)
    rec = {"frames": frames, "views": 4, "method": method}

    def step_unproject(fn):
        f = feats.detach().requires_grad_(True)
        c = vconf.detach().requires_grad_(True)
        (fn(f, proj, coord, method, c) * up_vol).sum().backward()
        return f.grad

    def step_argmax(fn):
        v = vols.detach().requires_grad_(True)
        kp, p = fn(v, coord)
        ((kp * up_kp).sum() + (p * up_p).sum()).backward()
        return v.grad

    with torch.no_grad():
        rec["mvx_unproject_forward_ms"] = timed(lambda: op.unproject_heatmaps(feats, proj, coord, method, vconf))
        rec["mvx_soft_argmax_3d_forward_ms"] = timed(lambda: op.integrate_tensor_3d_with_coordinates(vols, coord))
    rec["mvx_unproject_forward_backward_ms"] = timed(lambda: step_unproject(op.unproject_heatmaps))
    rec["mvx_soft_argmax_3d_forward_backward_ms"] = timed(lambda: step_argmax(op.integrate_tensor_3d_with_coordinates))
    rec["torch_unproject_forward_backward_ms"] = timed(lambda: step_unproject(torch_unproject), steps=3, warmup=1)
    rec["torch_soft_argmax_3d_forward_backward_ms"] = timed(lambda: step_argmax(torch_soft_argmax), steps=3, warmup=1)
    ga, gb = step_unproject(op.unproject_heatmaps), step_unproject(torch_unproject)
    rec["unproject_grad_max_abs_difference"] = (ga - gb).abs().max().item()
    rec["unproject_grad_max_abs"] = gb.abs().max().item()
    ga, gb = step_argmax(op.integrate_tensor_3d_with_coordinates), step_argmax(torch_soft_argmax)
    rec["soft_argmax_3d_grad_max_abs_difference"] = (ga - gb).abs().max().item()
    rec["soft_argmax_3d_grad_max_abs"] = gb.abs().max().item()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
