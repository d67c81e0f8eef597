"""CPU baseline for the triangulation path: TEST/BENCH INFRASTRUCTURE ONLY.

A float32 torch-on-CPU restatement that keeps the *cost profile* of the
reference's implementation: eager torch ops with several passes over the
heatmaps, one host `np.linalg.svd` per (frame, joint) inside a Python double
loop, one `grid_sample` per (frame, view) with the per-view volumes
materialised before aggregation, and softmax + einsum for the volumetric
soft-argmax.  `bench.py` times it on the GPU box's host cores as the
`cpu_baseline` ("port": the reference's own Python cannot travel to that box);
`oracle/gen_golden.py` records, in the build container, that its outputs and
its timing match the reference imported from its own tree.

Never imported by the product package.
"""
import numpy as np
import torch
import torch.nn.functional as F


def soft_argmax_2d(heatmaps, softmax=True):
    """Cost profile of mvn/utils/op.py:11-47: softmax pass, two marginal sums, two weighted sums."""
    n, k, h, w = heatmaps.shape
    flat = heatmaps.reshape(n, k, h * w)
    flat = F.softmax(flat, dim=2) if softmax else F.relu(flat)
    maps = flat.reshape(n, k, h, w)
    col_mass = maps.sum(dim=2)
    row_mass = maps.sum(dim=3)
    xs = torch.arange(w, dtype=torch.float32)
    ys = torch.arange(h, dtype=torch.float32)
    x = (col_mass * xs).sum(dim=2, keepdim=True)
    y = (row_mass * ys).sum(dim=2, keepdim=True)
    if not softmax:
        x = x / col_mass.sum(dim=2, keepdim=True)
        y = y / row_mass.sum(dim=2, keepdim=True)
    return torch.cat((x, y), dim=2), maps


def dlt_point(proj, pts, conf=None):
    """Cost profile of mvn/utils/multiview.py:143-172: three tiny torch ops, then a host LAPACK SVD."""
    v = proj.shape[0]
    if conf is None:
        conf = torch.ones(v, dtype=torch.float32)
    a = proj[:, 2:3].expand(v, 2, 4) * pts.view(v, 2, 1)
    a = a - proj[:, :2]
    a = a * conf.view(v, 1, 1)
    vh = np.linalg.svd(a.reshape(2 * v, 4).numpy())[2]
    sol = torch.from_numpy(vh[3])
    return sol[:3] / sol[3]


def dlt_batch(proj, pts, conf=None):
    """Cost profile of mvn/utils/multiview.py:175-198: Python loop over frames and joints."""
    b, v, j = pts.shape[:3]
    out = torch.zeros(b, j, 3, dtype=torch.float32)
    for bi in range(b):
        for ji in range(j):
            c = None if conf is None else conf[bi, :, ji]
            out[bi, ji] = dlt_point(proj[bi], pts[bi, :, ji, :], c)
    return out


def algebraic_head(heatmaps, alg_conf, proj, multiplier=100.0, softmax=True, image_shape=(384, 384)):
    """Cost profile of mvn/models/triangulation.py:189-236 after the backbone."""
    b, v, j, h, w = heatmaps.shape
    kp, maps = soft_argmax_2d((heatmaps * multiplier).view(b * v, j, h, w), softmax)
    maps = maps.view(b, v, j, h, w)
    kp = kp.view(b, v, j, 2)
    conf = alg_conf.view(b, v, j)
    conf = conf / conf.sum(dim=1, keepdim=True)
    conf = conf + 1e-5
    kp_img = torch.zeros_like(kp)
    kp_img[..., 0] = kp[..., 0] * (image_shape[1] / w)
    kp_img[..., 1] = kp[..., 1] * (image_shape[0] / h)
    kp3d = dlt_batch(proj, kp_img, conf)
    return kp3d, kp_img, maps, conf


def unproject(features, proj, coord_volumes, method="sum", vol_conf=None):
    """Cost profile of mvn/utils/op.py:99-166: per (frame, view) projection + grid_sample,
    per-view volumes kept until the aggregation."""
    b, v, c, h, w = features.shape
    vshape = tuple(coord_volumes.shape[1:4])
    out = torch.zeros(b, c, *vshape)
    for bi in range(b):
        pts = coord_volumes[bi].reshape(-1, 3)
        pts_h = torch.cat([pts, torch.ones(pts.shape[0], 1)], dim=1)
        stack = torch.zeros(v, c, *vshape)
        for vi in range(v):
            uvw = pts_h @ proj[bi, vi].t()
            behind = uvw[:, 2] <= 0.0
            uvw[uvw[:, 2] == 0.0, 2] = 1.0
            uv = uvw[:, :2] / uvw[:, 2:3]
            g = torch.zeros_like(uv)
            g[:, 0] = 2 * (uv[:, 0] / h - 0.5)
            g[:, 1] = 2 * (uv[:, 1] / w - 0.5)
            samp = F.grid_sample(features[bi, vi].unsqueeze(0), g.view(1, -1, 1, 2), align_corners=True)
            samp = samp.view(c, -1)
            samp[:, behind] = 0.0
            stack[vi] = samp.view(c, *vshape)
        if method.startswith("conf"):
            out[bi] = (stack * vol_conf[bi].view(v, c, 1, 1, 1)).sum(0)
        elif method == "sum":
            out[bi] = stack.sum(0)
        elif method == "max":
            out[bi] = stack.max(0)[0]
        elif method == "softmax":
            wgt = F.softmax(stack.clone().view(v, -1), dim=0).view(v, c, *vshape)
            out[bi] = (stack * wgt).sum(0)
        else:
            raise ValueError("Unknown volume_aggregation_method: {}".format(method))
    return out


def soft_argmax_3d_coords(volumes, coord_volumes, softmax=True):
    """Cost profile of mvn/utils/op.py:84-96."""
    b, n = volumes.shape[:2]
    flat = volumes.reshape(b, n, -1)
    flat = F.softmax(flat, dim=2) if softmax else F.relu(flat)
    vols = flat.reshape(volumes.shape)
    return torch.einsum("bnxyz,bxyzc->bnc", vols, coord_volumes), vols


def time_algebraic(case, repeats=5, warmup=2, threads=None):
    """Median seconds for one `algebraic_head` pass over `case` (numpy arrays from synth.algebraic_case)."""
    import time
    if threads:
        torch.set_num_threads(threads)
    hm = torch.from_numpy(case["heatmaps"])
    conf = torch.from_numpy(case["confidences"])
    proj = torch.from_numpy(case["proj"])
    ts = []
    with torch.no_grad():
        for i in range(warmup + repeats):
            t0 = time.perf_counter()
            algebraic_head(hm, conf, proj)
            t1 = time.perf_counter()
            if i >= warmup:
                ts.append(t1 - t0)
    return float(np.median(ts))
