"""CPU oracle for the multi-view triangulation path: TEST INFRASTRUCTURE ONLY.

A numpy restatement, evaluated in float64, of the reference's torch functions
on the triangulation path.  It is the checker for the HIP kernels and must
never be imported by the product package (`poseestimation_amd`); only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may use it.

Parity pin: the reference ships no tests or golden vectors for this path, so
the pin is `tests/golden/*.npz`, produced by `oracle/gen_golden.py`, which
imports the reference's own functions (in the build container, where the
reference tree is mounted) on seeded synthetic inputs and stores their fp32
outputs and their outputs when evaluated in float64.  `tests/test_oracle_golden.py`
checks this restatement against those vectors.

Precision model.  Inputs are the float32 arrays the reference would receive.
Element-wise float32 steps whose rounding the reference fixes before any
reduction (the DLT design matrix, the confidence normalisation, the keypoint
upscale) are repeated here in float32 so that they are bit-identical; every
reduction, softmax, SVD and interpolation is evaluated in float64.
"""
import numpy as np


# ----------------------------------------------------------------------------
# soft-argmax over 2-D heatmaps        reference: mvn/utils/op.py:11-47
# ----------------------------------------------------------------------------
def integrate_tensor_2d(heatmaps, softmax=True):
    """heatmaps (N, n, h, w) -> (coordinates (N, n, 2) as [x, y], heatmaps (N, n, h, w)), float64.

    softmax=True: softmax over h*w (op.py:24-25); softmax=False: ReLU (op.py:27) and the
    coordinates are divided by the mass (op.py:40-42); the returned maps are then the
    ReLU'd ones, not normalised.
    """
    hm = np.asarray(heatmaps, dtype=np.float64)
    N, n, h, w = hm.shape
    flat = hm.reshape(N, n, -1)
    if softmax:
        m = flat.max(axis=2, keepdims=True)
        e = np.exp(flat - m)
        flat = e / e.sum(axis=2, keepdims=True)
    else:
        flat = np.maximum(flat, 0.0)
    hm = flat.reshape(N, n, h, w)
    mass_x = hm.sum(axis=2)  # (N, n, w)   op.py:31
    mass_y = hm.sum(axis=3)  # (N, n, h)   op.py:32
    x = (mass_x * np.arange(w, dtype=np.float64)).sum(axis=2, keepdims=True)
    y = (mass_y * np.arange(h, dtype=np.float64)).sum(axis=2, keepdims=True)
    if not softmax:
        with np.errstate(invalid="ignore", divide="ignore"):
            x = x / mass_x.sum(axis=2, keepdims=True)
            y = y / mass_y.sum(axis=2, keepdims=True)
    return np.concatenate([x, y], axis=2), hm


# ----------------------------------------------------------------------------
# soft-argmax over 3-D volumes, index space        reference: mvn/utils/op.py:50-81
# ----------------------------------------------------------------------------
def integrate_tensor_3d(volumes, softmax=True):
    vol = np.asarray(volumes, dtype=np.float64)
    B, n, sx, sy, sz = vol.shape
    flat = vol.reshape(B, n, -1)
    if softmax:
        m = flat.max(axis=2, keepdims=True)
        e = np.exp(flat - m)
        flat = e / e.sum(axis=2, keepdims=True)
    else:
        flat = np.maximum(flat, 0.0)
    vol = flat.reshape(B, n, sx, sy, sz)
    mass_x = vol.sum(axis=(3, 4))
    mass_y = vol.sum(axis=(2, 4))
    mass_z = vol.sum(axis=(2, 3))
    x = (mass_x * np.arange(sx)).sum(axis=2, keepdims=True)
    y = (mass_y * np.arange(sy)).sum(axis=2, keepdims=True)
    z = (mass_z * np.arange(sz)).sum(axis=2, keepdims=True)
    if not softmax:
        with np.errstate(invalid="ignore", divide="ignore"):
            x = x / mass_x.sum(axis=2, keepdims=True)
            y = y / mass_y.sum(axis=2, keepdims=True)
            z = z / mass_z.sum(axis=2, keepdims=True)
    return np.concatenate([x, y, z], axis=2), vol


# ----------------------------------------------------------------------------
# soft-argmax over 3-D volumes with world coordinates   reference: mvn/utils/op.py:84-96
# ----------------------------------------------------------------------------
def integrate_tensor_3d_with_coordinates(volumes, coord_volumes, softmax=True):
    """volumes (B, n, X, Y, Z), coord_volumes (B, X, Y, Z, 3) -> (coords (B, n, 3), volumes), float64.
    softmax=False is a plain ReLU with no renormalisation (op.py:90-91)."""
    vol = np.asarray(volumes, dtype=np.float64)
    B, n = vol.shape[:2]
    flat = vol.reshape(B, n, -1)
    if softmax:
        m = flat.max(axis=2, keepdims=True)
        e = np.exp(flat - m)
        flat = e / e.sum(axis=2, keepdims=True)
    else:
        flat = np.maximum(flat, 0.0)
    cv = np.asarray(coord_volumes, dtype=np.float64).reshape(B, -1, 3)
    coords = np.einsum("bnk,bkc->bnc", flat, cv)  # op.py:94
    return coords, flat.reshape(vol.shape)


# ----------------------------------------------------------------------------
# homogeneous helpers and projection     reference: mvn/utils/multiview.py:57-112
# ----------------------------------------------------------------------------
def euclidean_to_homogeneous(points):
    points = np.asarray(points)
    return np.hstack([points, np.ones((len(points), 1), dtype=points.dtype)])


def homogeneous_to_euclidean(points):
    points = np.asarray(points)
    return (points.T[:-1] / points.T[-1]).T


def project_3d_points_to_image_plane_without_distortion(proj_matrix, points_3d, convert_back_to_euclidean=True):
    result = euclidean_to_homogeneous(np.asarray(points_3d, dtype=np.float64)) @ np.asarray(proj_matrix, dtype=np.float64).T
    if convert_back_to_euclidean:
        result = homogeneous_to_euclidean(result)
    return result


# ----------------------------------------------------------------------------
# confidence-weighted DLT                reference: mvn/utils/multiview.py:143-198
# ----------------------------------------------------------------------------
def dlt_design_matrix(proj_matricies, points, confidences=None):
    """The (.., 2V, 4) matrix of multiview.py:161-163, with the reference's float32 roundings:
    A = fl(fl(fl(P[2] * x) - P[0:2]) * c)."""
    P = np.asarray(proj_matricies, dtype=np.float32)          # (..., V, 3, 4)
    pts = np.asarray(points, dtype=np.float32)                # (..., V, 2)
    V = P.shape[-3]
    A = P[..., 2:3, :] * pts[..., :, None]                    # (..., V, 2, 4), float32 product
    A = A - P[..., :2, :]
    if confidences is None:
        c = np.ones(pts.shape[:-1], dtype=np.float32)
    else:
        c = np.asarray(confidences, dtype=np.float32)
    A = A * c[..., None, None]
    assert A.dtype == np.float32
    return A.reshape(A.shape[:-3] + (2 * V, 4))


def triangulate_point_from_multiple_views_linear(proj_matricies, points, confidences=None):
    """One point from V views: right singular vector of the smallest singular value of the
    float32 design matrix, solved in float64, then divided by its last component
    (multiview.py:166-170; the two sign flips there cancel)."""
    assert len(proj_matricies) == len(points)
    A = dlt_design_matrix(proj_matricies, points, confidences).astype(np.float64)
    vh = np.linalg.svd(A, full_matrices=False)[2]
    h = vh[3]
    return h[:3] / h[3]


def triangulate_batch_of_points(proj_matricies_batch, points_batch, confidences_batch=None):
    """proj (B,V,3,4), points (B,V,J,2), conf (B,V,J) or None -> (B,J,3) float64 (multiview.py:175-198)."""
    P = np.asarray(proj_matricies_batch, dtype=np.float32)
    pts = np.asarray(points_batch, dtype=np.float32)
    B, V, J = pts.shape[:3]
    if B == 0 or J == 0:
        return np.zeros((B, J, 3))
    Pj = np.broadcast_to(P[:, None], (B, J, V, 3, 4))
    ptsj = np.transpose(pts, (0, 2, 1, 3))                    # (B,J,V,2)
    cj = None if confidences_batch is None else np.transpose(np.asarray(confidences_batch, dtype=np.float32), (0, 2, 1))
    A = dlt_design_matrix(Pj, ptsj, cj).astype(np.float64)    # (B,J,2V,4)
    vh = np.linalg.svd(A, full_matrices=False)[2]
    h = vh[..., 3, :]
    with np.errstate(invalid="ignore", divide="ignore"):
        return h[..., :3] / h[..., 3:4]


# ----------------------------------------------------------------------------
# glue of AlgebraicTriangulationNet.forward   reference: mvn/models/triangulation.py:190-220
# ----------------------------------------------------------------------------
def algebraic_glue(keypoints_2d_hm, alg_confidences, image_shape, heatmap_shape):
    """Float32 element-wise steps exactly as the reference orders them:
    conf = conf / sum_views(conf) + 1e-5 (triangulation.py:199-200);
    x *= W_img / W_hm, y *= H_img / H_hm (triangulation.py:208-211)."""
    conf = np.asarray(alg_confidences, dtype=np.float32)
    # the view sum is taken exactly and rounded once; torch's float32 summation order is unspecified
    # and differs from this by at most an ulp of the sum for V > 4
    conf = conf / conf.sum(axis=1, keepdims=True, dtype=np.float64).astype(np.float32)
    conf = conf + np.float32(1e-5)
    kp = np.asarray(keypoints_2d_hm, dtype=np.float32).copy()
    kp[..., 0] = kp[..., 0] * np.float32(image_shape[1] / heatmap_shape[1])
    kp[..., 1] = kp[..., 1] * np.float32(image_shape[0] / heatmap_shape[0])
    return kp, conf


def algebraic_forward(heatmaps_raw, alg_confidences, proj_matricies, multiplier=100.0, softmax=True,
                      image_shape=(384, 384)):
    """Everything AlgebraicTriangulationNet.forward does after the backbone (triangulation.py:189-236):
    returns (kp3d (B,J,3) f64, kp2d (B,V,J,2) f32, heatmaps (B,V,J,h,w) f64, conf (B,V,J) f32).
    The multiplier is applied in float32 (triangulation.py:190); the soft-argmax is exact on that
    float32 product; its coordinates are rounded to float32 once, as the reference's output is."""
    hm = np.asarray(heatmaps_raw, dtype=np.float32)
    B, V, J, h, w = hm.shape
    scaled = hm * np.float32(multiplier)
    coords, hm_out = integrate_tensor_2d(scaled.reshape(B * V, J, h, w), softmax)
    kp2d = coords.astype(np.float32).reshape(B, V, J, 2)
    kp2d, conf = algebraic_glue(kp2d, np.asarray(alg_confidences).reshape(B, V, J), image_shape, (h, w))
    kp3d = triangulate_batch_of_points(proj_matricies, kp2d, conf)
    return kp3d, kp2d, hm_out.reshape(B, V, J, h, w), conf


# ----------------------------------------------------------------------------
# feature unprojection                   reference: mvn/utils/op.py:99-166
# ----------------------------------------------------------------------------
def _grid_sample_bilinear_zeros(img, ix, iy):
    """img (C,H,W) f64; ix, iy (N,) f64 pixel coordinates (align_corners=True convention).
    Bilinear, zero padding per tap, as torch.nn.functional.grid_sample (op.py:135)."""
    C, H, W = img.shape
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    wx1 = ix - x0
    wy1 = iy - y0
    out = np.zeros((C, ix.shape[0]))
    for dy, wy in ((0, 1.0 - wy1), (1, wy1)):
        for dx, wx in ((0, 1.0 - wx1), (1, wx1)):
            xx = x0 + dx
            yy = y0 + dy
            ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            xi = np.clip(xx, 0, W - 1).astype(np.int64)
            yi = np.clip(yy, 0, H - 1).astype(np.int64)
            out += img[:, yi, xi] * (wx * wy * ok)[None, :]
    return out


def unproject_heatmaps(heatmaps, proj_matricies, coord_volumes, volume_aggregation_method="sum", vol_confidences=None,
                       voxel_subset=None):
    """heatmaps (B,V,C,H,W), proj (B,V,3,4) at heatmap scale, coord_volumes (B,X,Y,Z,3),
    vol_confidences (B,V,C) -> (B,C,X,Y,Z) float64.  `voxel_subset` (flat voxel indices) restricts
    the evaluation to those voxels and returns (B,C,len(subset)) instead.

    Per view (op.py:117-141): project the grid with P; invalid = z <= 0; z == 0 -> 1; divide;
    gx = 2(x/H - 0.5), gy = 2(y/W - 0.5) (H is used for x and W for y, op.py:128-129);
    grid_sample(align_corners=True) maps g to pixel (g+1)/2*(size-1); invalid voxels are set to 0
    before the aggregation.  Aggregation (op.py:150-162): conf* / sum / max / softmax over views.
    """
    method = volume_aggregation_method
    if not (method.startswith("conf") or method in ("sum", "max", "softmax")):
        raise ValueError("Unknown volume_aggregation_method: {}".format(method))
    hm = np.asarray(heatmaps, dtype=np.float64)
    P = np.asarray(proj_matricies, dtype=np.float64)
    cv = np.asarray(coord_volumes, dtype=np.float64)
    B, V, C, H, W = hm.shape
    vshape = cv.shape[1:4]
    out_list = []
    for b in range(B):
        grid = cv[b].reshape(-1, 3)
        if voxel_subset is not None:
            grid = grid[np.asarray(voxel_subset)]
        per_view = np.zeros((V, C, grid.shape[0]))
        for v in range(V):
            uvw = np.hstack([grid, np.ones((grid.shape[0], 1))]) @ P[b, v].T
            invalid = uvw[:, 2] <= 0.0
            z = np.where(uvw[:, 2] == 0.0, 1.0, uvw[:, 2])
            x = uvw[:, 0] / z
            y = uvw[:, 1] / z
            gx = 2.0 * (x / H - 0.5)
            gy = 2.0 * (y / W - 0.5)
            ix = (gx + 1.0) / 2.0 * (W - 1)
            iy = (gy + 1.0) / 2.0 * (H - 1)
            vals = _grid_sample_bilinear_zeros(hm[b, v], ix, iy)
            vals[:, invalid] = 0.0
            per_view[v] = vals
        if method.startswith("conf"):
            agg = (per_view * np.asarray(vol_confidences, dtype=np.float64)[b][:, :, None]).sum(0)
        elif method == "sum":
            agg = per_view.sum(0)
        elif method == "max":
            agg = per_view.max(0)
        else:
            m = per_view.max(0, keepdims=True)
            e = np.exp(per_view - m)
            agg = (per_view * (e / e.sum(0, keepdims=True))).sum(0)
        out_list.append(agg)
    out = np.stack(out_list, axis=0) if B else np.zeros((0, C, 0))
    if voxel_subset is None:
        out = out.reshape((B, C) + tuple(vshape))
    return out


# ----------------------------------------------------------------------------
# gradients of the volumetric operations: what torch autograd derives through
# mvn/utils/op.py:50-96 and :99-166 when the reference trains (train.py:281-290)
# ----------------------------------------------------------------------------
def _grid_sample_bilinear_zeros_transpose(grad, ix, iy, H, W):
    """Transpose of _grid_sample_bilinear_zeros: grad (C, N) scattered to (C, H, W) with the same weights;
    taps outside the map receive nothing (grid_sample's backward with zero padding)."""
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    wx1 = ix - x0
    wy1 = iy - y0
    out = np.zeros((grad.shape[0], H, W))
    for dy, wy in ((0, 1.0 - wy1), (1, wy1)):
        for dx, wx in ((0, 1.0 - wx1), (1, wx1)):
            xx = x0 + dx
            yy = y0 + dy
            ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            xi = np.clip(xx, 0, W - 1).astype(np.int64)
            yi = np.clip(yy, 0, H - 1).astype(np.int64)
            for c in range(grad.shape[0]):
                np.add.at(out[c], (yi, xi), grad[c] * wx * wy * ok)
    return out


def unproject_heatmaps_backward(heatmaps, proj_matricies, coord_volumes, grad_volumes, volume_aggregation_method="sum",
                                vol_confidences=None):
    """Gradients of unproject_heatmaps' output with respect to heatmaps (B,V,C,H,W) and vol_confidences (B,V,C),
    given grad_volumes (B,C,X,Y,Z): float64.  Chain rule through op.py:150-162 (aggregation), op.py:141 (voxels
    behind a camera are overwritten with 0: no gradient) and op.py:135 (grid_sample).  For "max" the gradient goes
    to the first view that holds the maximum (torch.max(0) returns the first maximal index)."""
    method = volume_aggregation_method
    hm = np.asarray(heatmaps, dtype=np.float64)
    P = np.asarray(proj_matricies, dtype=np.float64)
    cv = np.asarray(coord_volumes, dtype=np.float64)
    B, V, C, H, W = hm.shape
    g_hm = np.zeros_like(hm)
    g_conf = np.zeros((B, V, C)) if method.startswith("conf") else None
    for b in range(B):
        grid = cv[b].reshape(-1, 3)
        g = np.asarray(grad_volumes, dtype=np.float64)[b].reshape(C, -1)
        per_view = np.zeros((V, C, grid.shape[0]))
        pix, invalid = [], []
        for v in range(V):
            uvw = np.hstack([grid, np.ones((grid.shape[0], 1))]) @ P[b, v].T
            inv = uvw[:, 2] <= 0.0
            z = np.where(uvw[:, 2] == 0.0, 1.0, uvw[:, 2])
            ix = (uvw[:, 0] / z) / H * (W - 1)
            iy = (uvw[:, 1] / z) / W * (H - 1)
            vals = _grid_sample_bilinear_zeros(hm[b, v], ix, iy)
            vals[:, inv] = 0.0
            per_view[v] = vals
            pix.append((ix, iy))
            invalid.append(inv)
        if method.startswith("conf"):
            conf = np.asarray(vol_confidences, dtype=np.float64)[b]
            d = g[None] * conf[:, :, None]
            g_conf[b] = (g[None] * per_view).sum(2)
        elif method == "sum":
            d = np.broadcast_to(g[None], per_view.shape).copy()
        elif method == "max":
            first = per_view.argmax(0)
            d = np.where(np.arange(V)[:, None, None] == first[None], g[None], 0.0)
        elif method == "softmax":
            e = np.exp(per_view - per_view.max(0, keepdims=True))
            s = e / e.sum(0, keepdims=True)
            out = (per_view * s).sum(0, keepdims=True)
            d = g[None] * s * (1.0 + per_view - out)
        else:
            raise ValueError("Unknown volume_aggregation_method: {}".format(method))
        for v in range(V):
            dv = d[v].copy()
            dv[:, invalid[v]] = 0.0
            g_hm[b, v] = _grid_sample_bilinear_zeros_transpose(dv, pix[v][0], pix[v][1], H, W)
    return g_hm, g_conf


def integrate_tensor_3d_backward(volumes, coord_volumes, grad_coords, grad_volumes_out, softmax=True):
    """Gradient with respect to `volumes` of integrate_tensor_3d_with_coordinates (coord_volumes given) or
    integrate_tensor_3d (coord_volumes None: voxel indices, and the division by the mass for softmax=False,
    op.py:73-76), for upstream gradients of the coordinates (B,n,3) and of the returned volumes (either may be None)."""
    vol = np.asarray(volumes, dtype=np.float64)
    B, n, sx, sy, sz = vol.shape
    flat = vol.reshape(B, n, -1)
    if coord_volumes is not None:
        X = np.asarray(coord_volumes, dtype=np.float64).reshape(B, 1, -1, 3)
    else:
        idx = np.stack(np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij"), axis=-1)
        X = np.broadcast_to(idx.reshape(1, 1, -1, 3).astype(np.float64), (B, 1, sx * sy * sz, 3))
    gc = np.zeros((B, n, 3)) if grad_coords is None else np.asarray(grad_coords, dtype=np.float64)
    gp = np.zeros_like(flat) if grad_volumes_out is None else np.asarray(grad_volumes_out, dtype=np.float64).reshape(B, n, -1)
    if softmax:
        e = np.exp(flat - flat.max(axis=2, keepdims=True))
        p = e / e.sum(axis=2, keepdims=True)
        G = gp + np.einsum("bnc,bmkc->bnk", gc, X)
        out = p * (G - (p * G).sum(axis=2, keepdims=True))
    else:
        p = np.maximum(flat, 0.0)
        if coord_volumes is not None:
            G = gp + np.einsum("bnc,bmkc->bnk", gc, X)
        else:
            mass = p.sum(axis=2, keepdims=True)
            c = np.einsum("bnk,bmkc->bnc", p, X) / mass
            G = gp + np.einsum("bnc,bnkc->bnk", gc, X - c[:, :, None, :]) / mass
        out = np.where(flat > 0.0, G, 0.0)
    return out.reshape(vol.shape)


# ----------------------------------------------------------------------------
# camera and coordinate-volume helpers
# reference: mvn/utils/multiview.py:7-54, mvn/utils/volumetric.py:87-114,
#            mvn/models/triangulation.py:341-377
# ----------------------------------------------------------------------------
def camera_projection(K, R, t):
    """K [R|t] in float64 (multiview.py:48-54)."""
    return np.asarray(K, dtype=np.float64) @ np.hstack([np.asarray(R, dtype=np.float64), np.asarray(t, dtype=np.float64).reshape(3, 1)])


def camera_update_after_resize(K, image_shape, new_image_shape):
    """multiview.py:35-46: fx, cx scale with width; fy, cy with height."""
    K = np.array(K, dtype=np.float64).copy()
    h, w = image_shape
    nh, nw = new_image_shape
    K[0, 0] *= nw / w
    K[1, 1] *= nh / h
    K[0, 2] *= nw / w
    K[1, 2] *= nh / h
    return K


def camera_update_after_crop(K, bbox):
    """multiview.py:25-33: principal point moves by the crop's left/upper corner."""
    K = np.array(K, dtype=np.float64).copy()
    left, upper, right, lower = bbox
    K[0, 2] -= left
    K[1, 2] -= upper
    return K


def get_rotation_matrix(axis, theta):
    """Euler-Rodrigues rotation (volumetric.py:87-99)."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.sqrt(np.dot(axis, axis))
    a = np.cos(theta / 2.0)
    b, c, d = -axis * np.sin(theta / 2.0)
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c + a * d), 2 * (b * d - a * c)],
        [2 * (b * c - a * d), a * a + c * c - b * b - d * d, 2 * (c * d + a * b)],
        [2 * (b * d + a * c), 2 * (c * d - a * b), a * a + d * d - b * b - c * c]])


def build_coord_volume(base_point, cuboid_side, volume_size, theta=0.0, axis=(0, 0, 1)):
    """triangulation.py:341-377 for one frame, in float64: grid = position + side/(S-1) * idx,
    index order [x,y,z]; rotated by theta about `axis` around the base point."""
    S = volume_size
    base = np.asarray(base_point, dtype=np.float64)
    pos = base - cuboid_side / 2
    idx = np.arange(S, dtype=np.float64)
    g = np.zeros((S, S, S, 3))
    g[..., 0] = (pos[0] + cuboid_side / (S - 1) * idx)[:, None, None]
    g[..., 1] = (pos[1] + cuboid_side / (S - 1) * idx)[None, :, None]
    g[..., 2] = (pos[2] + cuboid_side / (S - 1) * idx)[None, None, :]
    rot = get_rotation_matrix(axis, theta)
    g = ((g - base).reshape(-1, 3) @ rot.T).reshape(S, S, S, 3) + base
    return g


# ----------------------------------------------------------------------------
# RANSAC triangulation                 reference: mvn/models/triangulation.py:19-148
# ----------------------------------------------------------------------------
def heatmap_argmax_keypoints(heatmaps, image_shape):
    """heatmaps (B, V, J, h, w) -> (B, V, J, 2) int64 [x, y] in image pixels.  The flat arg-max (first maximal
    entry; a NaN counts as maximal, as in torch.max) is split into column and row, and each is multiplied by
    image/heatmap size in float32 and truncated: an int64 tensor times a Python float is a float32 tensor in
    torch, and the reference assigns it into an int64 tensor.  Reference: triangulation.py:48-56."""
    B, V, J, h, w = heatmaps.shape
    flat = np.argmax(heatmaps.reshape(B, V, J, h * w), axis=-1)
    x, y = flat % w, flat // w
    sx, sy = np.float32(image_shape[1] / w), np.float32(image_shape[0] / h)
    out = np.empty((B, V, J, 2), dtype=np.int64)
    out[..., 0] = np.trunc(x.astype(np.float32) * sx).astype(np.int64)
    out[..., 1] = np.trunc(y.astype(np.float32) * sy).astype(np.int64)
    return out


def dlt_numpy(proj_matricies, points):
    """Unweighted DLT in float64 with numpy's SVD.  Reference: mvn/utils/multiview.py:115-140."""
    n = len(proj_matricies)
    A = np.zeros((2 * n, 4))
    for v in range(n):
        P = np.asarray(proj_matricies[v], dtype=np.float64)
        A[2 * v] = np.float64(points[v][0]) * P[2] - P[0]
        A[2 * v + 1] = np.float64(points[v][1]) * P[2] - P[1]
    vh = np.linalg.svd(A, full_matrices=False)[2]
    return vh[3, :3] / vh[3, 3]


def reprojection_errors(point_3d, points, proj_matricies):
    """HALF the pixel distance between each keypoint and the projection of `point_3d`, per view.
    Reference: mvn/utils/multiview.py:201-208 (note the 1/2 at :205)."""
    Xh = np.append(np.asarray(point_3d, dtype=np.float64), 1.0)
    errs = np.empty(len(points))
    for v in range(len(points)):
        q = np.asarray(proj_matricies[v], dtype=np.float64) @ Xh
        errs[v] = 0.5 * np.sqrt(np.sum((np.asarray(points[v], dtype=np.float64) - q[:2] / q[2]) ** 2))
    return errs


def triangulate_ransac(proj_matricies, points, view_pairs, reprojection_error_epsilon=15, direct_optimization=True):
    """One point.  `view_pairs` (n_iters, 2) are the sorted view pairs the reference would have drawn with
    random.sample (triangulation.py:97).  Returns (point (3,) float64, sorted inlier indices).
    Reference: triangulation.py:83-148; the refinement is scipy.optimize.least_squares(loss='huber',
    method='trf') on the vector of reprojection errors, as at :133-142."""
    proj_matricies = np.asarray(proj_matricies)
    points = np.asarray(points)
    n_views = len(points)
    inliers = set()
    for a, b in view_pairs:
        sampled = [int(a), int(b)]
        X = dlt_numpy(proj_matricies[sampled], points[sampled])
        errs = reprojection_errors(X, points, proj_matricies)
        new = set(sampled) | {v for v in range(n_views) if errs[v] < reprojection_error_epsilon}
        if len(new) > len(inliers):
            inliers = new
    if not inliers:
        inliers = set(range(n_views))
    idx = np.array(sorted(inliers))
    X = dlt_numpy(proj_matricies[idx], points[idx])
    if direct_optimization:
        from scipy.optimize import least_squares
        res = least_squares(lambda x: reprojection_errors(x, points[idx], proj_matricies[idx]), np.array(X),
                            loss="huber", method="trf")
        X = res.x
    return X, idx


def huber_minimiser(start, points, proj_matricies):
    """The point the refinement at triangulation.py:140 is heading for: the same least_squares problem run to
    machine-precision tolerances instead of scipy's default ftol = xtol = gtol = 1e-8 (at which it stops up to
    1.3e-2 mm short on the fixtures here)."""
    from scipy.optimize import least_squares
    res = least_squares(lambda x: reprojection_errors(x, points, proj_matricies), np.array(start, dtype=np.float64),
                        loss="huber", method="trf", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=5000)
    return res.x


def huber_cost(point_3d, points, proj_matricies):
    """The objective least_squares minimises at triangulation.py:140: 1/2 sum rho(r^2), rho Huber with unit scale."""
    s = reprojection_errors(point_3d, points, proj_matricies) ** 2
    return 0.5 * np.sum(np.where(s <= 1.0, s, 2.0 * np.sqrt(s) - 1.0))


def ransac_batch(proj_matricies, keypoints_2d, view_pairs, reprojection_error_epsilon=15, direct_optimization=True):
    """proj (B, V, 3, 4), keypoints_2d (B, V, J, 2), view_pairs (B*J, n_iters, 2) -> ((B, J, 3) float64,
    (B, J) inlier bit masks).  Reference loop: triangulation.py:67-73."""
    B, V, J = keypoints_2d.shape[:3]
    out = np.zeros((B, J, 3))
    masks = np.zeros((B, J), dtype=np.int64)
    for b in range(B):
        for j in range(J):
            X, idx = triangulate_ransac(proj_matricies[b], keypoints_2d[b, :, j], view_pairs[b * J + j],
                                        reprojection_error_epsilon, direct_optimization)
            out[b, j] = X
            masks[b, j] = sum(1 << int(v) for v in idx)
    return out, masks


# ----------------------------------------------------------------------------
# temporal Savitzky-Golay filter       reference: savgol_filer.py:7-70 (scipy.signal.savgol_filter inside)
# ----------------------------------------------------------------------------
def savgol_window(window_size, n_steps, polyorder=2):
    """The window and polynomial order the reference ends up using (savgol_filer.py:33-45): an even window
    loses one sample, a window longer than the sequence is cut to it, the order drops below the window."""
    window = window_size - 1 if window_size % 2 == 0 else window_size
    window = min(window, n_steps)
    order = window - 1 if window <= polyorder else polyorder
    assert order > 0 and window > order
    return window, order


def savgol_filter(x, window, order):
    """scipy.signal.savgol_filter(x, window, order, axis=0) with its default mode='interp', restated with
    explicit least-squares fits, float64: away from the ends each output is the value at the window centre of
    the polynomial fitted to the window around it; the first and last window//2 outputs are read off the
    polynomial fitted to the first and last `window` samples.  x (T, ...)."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[0]
    flat = x.reshape(T, -1)
    half = window // 2
    t = np.arange(window, dtype=np.float64)
    vander = np.vander(t, order + 1)                      # columns t^order ... t^0
    out = np.empty_like(flat)
    for c in range(half, T - half):
        if window % 2 == 0:
            break                                          # an even window equals T here: ends cover everything
        coef = np.linalg.lstsq(vander, flat[c - half:c + half + 1], rcond=None)[0]
        out[c] = np.polyval(coef, float(half))
    head = np.linalg.lstsq(vander, flat[:window], rcond=None)[0]
    tail = np.linalg.lstsq(vander, flat[T - window:], rcond=None)[0]
    for i in range(half):
        out[i] = np.polyval(head, float(i))
        out[T - half + i] = np.polyval(tail, float(window - half + i))
    return out.reshape(x.shape)


def savgol_smooth(x, window_size):
    """SAVGOLFilter(window_size)(x) for x (T, K, C): every coordinate series filtered along T.  Returns float64;
    the reference stores into an array of x's dtype (savgol_filer.py:54-60)."""
    window, order = savgol_window(window_size, x.shape[0])
    return savgol_filter(x, window, order)


# ----------------------------------------------------------------------------
# wire format of the real-time pipelines      reference: real_pipeline_trt.py:357-399
# (inline script code that cannot be imported; pinned by tests/golden/wire.npz, the messages that block produced
# when oracle/gen_golden.py executed it, from the file where it lies, on injected poses)
# ----------------------------------------------------------------------------
UNITY_JOINTS = ('Hip', 'RightHip', 'RightKnee', 'RightAnkle', 'LeftHip', 'LeftKnee', 'LeftAnkle', 'Spine', 'Thorax',
                'Neck', 'Head', 'LeftShoulder', 'LeftElbow', 'LeftWrist', 'RightShoulder', 'RightElbow', 'RightWrist')
UNITY_ORDER = (6, 2, 1, 0, 3, 4, 5, 7, 8, 16, 9, 13, 14, 15, 12, 11, 10)


def unity_points(points, order=UNITY_ORDER):
    """(J, 3) float32 -> joints reordered and axes mapped (x, y, z) -> (-x, z, -y).  real_pipeline_trt.py:371-384."""
    p = np.asarray(points, dtype=np.float32)[list(order)]
    return np.stack([-p[:, 0], p[:, 2], -p[:, 1]], axis=1)


def unity_scale(first_frame_points, unity_len=0.43):
    """unity_len over the length of the bone between reordered joints 2 and 3 of the first frame (:386-392)."""
    p = unity_points(first_frame_points)
    return unity_len / np.linalg.norm(p[2] - p[3])


def unity_message(points, scale, names=UNITY_JOINTS):
    """{joint name: [x, y, z]} of Python floats (:393-396)."""
    p = unity_points(points)
    return {names[j]: [float(p[j][0] * scale), float(p[j][1] * scale), float(p[j][2] * scale)] for j in range(len(p))}


def realtime_window_frames(window):
    """window (8, J, 3) float32 of the last frames -> the two frames the pipeline sends: the mean of the last two
    filtered frames and the last filtered frame (:360-367, filter window 8 -> 7)."""
    sm = savgol_smooth(np.asarray(window, dtype=np.float32).transpose(0, 2, 1), 8).astype(np.float32).transpose(0, 2, 1)
    return [(sm[-1] + sm[-2]) / 2, sm[-1]]
