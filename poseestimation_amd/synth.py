"""Synthetic Human3.6M-shaped inputs for the multi-view triangulation path.

Everything here is numpy on the host, driven by ``np.random.default_rng(seed)``
so that a (seed, shape) pair names one input set exactly; tests, the golden
fixture generator and ``bench.py`` all draw their inputs from these recipes.
For batches too large to build on the host, ``device_*`` helpers produce
inputs of the same distribution directly in HBM with torch's device RNG.

Geometry: V cameras on a ring of radius 4500 mm (+-0.1 rad jitter), 1500 mm
above the floor, looking at the middle of the capture volume; focal length
1145 px on a ~960 px crop, rescaled to the 384 px network input (x0.4) or to
the 96 px heatmap grid (x0.1); principal point at the image centre, jittered
per frame the way a per-frame crop would move it.
"""
import numpy as np

N_JOINTS = 17
IMAGE_SIZE = 384
HEATMAP_SIZE = 96
PELVIS_JOINT = 6  # "mpii" skeleton base joint


def _look_at_rotation(cam_pos, target):
    """World->camera rotation for a camera at cam_pos looking at target, image y down."""
    fwd = target - cam_pos
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.stack([right, down, fwd], axis=0)


def make_cameras(rng, n_frames, n_views, image_size=IMAGE_SIZE):
    """Returns K (B,V,3,3), R (V,3,3), t (V,3,1) in float64 for `image_size`-px images."""
    scale = image_size / 960.0
    angles = 2 * np.pi * np.arange(n_views) / n_views + rng.uniform(-0.1, 0.1, n_views)
    R = np.zeros((n_views, 3, 3))
    t = np.zeros((n_views, 3, 1))
    target = np.array([0.0, 0.0, 900.0])
    for v in range(n_views):
        pos = np.array([4500.0 * np.cos(angles[v]), 4500.0 * np.sin(angles[v]), 1500.0])
        R[v] = _look_at_rotation(pos, target)
        t[v] = -R[v] @ pos.reshape(3, 1)
    K = np.zeros((n_frames, n_views, 3, 3))
    f = 1145.0 * scale * rng.uniform(0.9, 1.1, (n_frames, n_views))
    K[:, :, 0, 0] = f
    K[:, :, 1, 1] = f
    K[:, :, 0, 2] = image_size / 2 + rng.uniform(-0.05, 0.05, (n_frames, n_views)) * image_size
    K[:, :, 1, 2] = image_size / 2 + rng.uniform(-0.05, 0.05, (n_frames, n_views)) * image_size
    K[:, :, 2, 2] = 1.0
    return K, R, t


def projection_matrices(K, R, t, rescale=1.0):
    """K[R|t] in float64 with the intrinsics rescaled by `rescale`, cast to float32 (B,V,3,4)."""
    Ks = K.copy()
    Ks[:, :, :2, :] *= rescale
    Rt = np.concatenate([R, t], axis=2)  # (V,3,4)
    P = np.einsum("bvij,vjk->bvik", Ks, Rt)
    return P.astype(np.float32)


def make_joints_3d(rng, n_frames, n_joints=N_JOINTS):
    xy = rng.uniform(-800.0, 800.0, (n_frames, n_joints, 2))
    z = rng.uniform(100.0, 1700.0, (n_frames, n_joints, 1))
    return np.concatenate([xy, z], axis=2)


def project(P, X):
    """P (B,V,3,4) float, X (B,J,3) -> (B,V,J,2) pixel coordinates (float64)."""
    Xh = np.concatenate([X, np.ones(X.shape[:-1] + (1,))], axis=-1)
    uvw = np.einsum("bvik,bjk->bvji", P.astype(np.float64), Xh)
    return uvw[..., :2] / uvw[..., 2:3]


def make_heatmaps(rng, centres_hm, size=HEATMAP_SIZE, noise=0.02):
    """Gaussian blobs (sigma in [1.5,3] px, amplitude in [0.3,1]) at centres_hm (...,2) plus N(0,noise)."""
    lead = centres_hm.shape[:-1]
    sig = rng.uniform(1.5, 3.0, lead)[..., None, None]
    amp = rng.uniform(0.3, 1.0, lead)[..., None, None]
    ys = np.arange(size, dtype=np.float64)[:, None]
    xs = np.arange(size, dtype=np.float64)[None, :]
    cx = centres_hm[..., 0][..., None, None]
    cy = centres_hm[..., 1][..., None, None]
    blob = amp * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sig * sig))
    blob = blob + rng.normal(0.0, noise, blob.shape)
    return blob.astype(np.float32)


def make_coord_volumes(base_points, cuboid_side=2500.0, volume_size=64):
    """(B,3) base points -> (B,S,S,S,3) float32 grid, index order [x,y,z], no rotation."""
    S = volume_size
    idx = np.arange(S, dtype=np.float32)
    step = np.float32(cuboid_side / (S - 1))
    out = np.zeros((base_points.shape[0], S, S, S, 3), dtype=np.float32)
    for b in range(base_points.shape[0]):
        pos = (base_points[b] - cuboid_side / 2).astype(np.float32)
        out[b, :, :, :, 0] = (pos[0] + step * idx)[:, None, None]
        out[b, :, :, :, 1] = (pos[1] + step * idx)[None, :, None]
        out[b, :, :, :, 2] = (pos[2] + step * idx)[None, None, :]
    return out


def algebraic_case(seed, n_frames, n_views, n_joints=N_JOINTS, worst_case=False):
    """Inputs of the algebraic path: raw heatmaps (before the x100 multiplier), confidences,
    image-pixel projection matrices and the ground-truth joints they were drawn from."""
    rng = np.random.default_rng(seed)
    K, R, t = make_cameras(rng, n_frames, n_views)
    proj_img = projection_matrices(K, R, t, 1.0)
    X = make_joints_3d(rng, n_frames, n_joints)
    centres = project(proj_img, X) * (HEATMAP_SIZE / IMAGE_SIZE)
    if worst_case:
        heatmaps = rng.standard_normal((n_frames, n_views, n_joints, HEATMAP_SIZE, HEATMAP_SIZE)).astype(np.float32)
    else:
        heatmaps = make_heatmaps(rng, centres)
    conf = rng.uniform(0.05, 1.0, (n_frames, n_views, n_joints)).astype(np.float32)
    return {"heatmaps": heatmaps, "confidences": conf, "proj": proj_img, "joints_3d": X}


def volumetric_case(seed, n_frames, n_views, n_channels=32, n_joints=N_JOINTS,
                    volume_size=64, cuboid_side=2500.0, heatmap_size=HEATMAP_SIZE, with_volumes=True):
    """Inputs of the volumetric path: feature maps, heatmap-pixel projection matrices, coordinate
    volumes around the pelvis, per-view volume confidences and a stand-in for the V2V output."""
    rng = np.random.default_rng(seed)
    K, R, t = make_cameras(rng, n_frames, n_views)
    proj_hm = projection_matrices(K, R, t, heatmap_size / IMAGE_SIZE)
    X = make_joints_3d(rng, n_frames, n_joints)
    base = X[:, PELVIS_JOINT if n_joints > PELVIS_JOINT else 0, :]
    coord = make_coord_volumes(base, cuboid_side, volume_size)
    feats = rng.standard_normal((n_frames, n_views, n_channels, heatmap_size, heatmap_size)).astype(np.float32)
    vconf = rng.uniform(0.0, 1.0, (n_frames, n_views, n_channels)).astype(np.float32)
    out = {"features": feats, "proj": proj_hm, "coord_volumes": coord, "vol_confidences": vconf,
           "joints_3d": X, "base_points": base}
    if with_volumes:
        S = volume_size
        vol = rng.standard_normal((n_frames, n_joints, S, S, S)).astype(np.float32)
        # a peak near each joint so the soft-argmax has something to find
        for b in range(n_frames):
            rel = (X[b] - (base[b] - cuboid_side / 2)) / (cuboid_side / (S - 1))
            for j in range(n_joints):
                c = np.clip(np.round(rel[j]).astype(int), 0, S - 1)
                vol[b, j, c[0], c[1], c[2]] += 12.0
        out["volumes"] = vol
    return out


def device_algebraic_case(seed, n_frames, n_views, device, n_joints=N_JOINTS, chunk=256):
    """Same distribution as `algebraic_case`, built in HBM chunk by chunk (torch device RNG).
    Camera geometry comes from the host recipe; heatmap pixels are drawn on the device."""
    import torch
    rng = np.random.default_rng(seed)
    K, R, t = make_cameras(rng, n_frames, n_views)
    proj_img = projection_matrices(K, R, t, 1.0)
    X = make_joints_3d(rng, n_frames, n_joints)
    centres = project(proj_img, X) * (HEATMAP_SIZE / IMAGE_SIZE)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hm = torch.empty((n_frames, n_views, n_joints, HEATMAP_SIZE, HEATMAP_SIZE), dtype=torch.float32, device=device)
    ys = torch.arange(HEATMAP_SIZE, dtype=torch.float32, device=device).view(1, 1, 1, -1, 1)
    xs = torch.arange(HEATMAP_SIZE, dtype=torch.float32, device=device).view(1, 1, 1, 1, -1)
    for s in range(0, n_frames, chunk):
        e = min(s + chunk, n_frames)
        c = torch.from_numpy(centres[s:e].astype(np.float32)).to(device)
        shape = (e - s, n_views, n_joints, 1, 1)
        sig = torch.empty(shape, device=device).uniform_(1.5, 3.0, generator=g)
        amp = torch.empty(shape, device=device).uniform_(0.3, 1.0, generator=g)
        cx = c[..., 0].unsqueeze(-1).unsqueeze(-1)
        cy = c[..., 1].unsqueeze(-1).unsqueeze(-1)
        blob = amp * torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sig * sig))
        noise = torch.empty(blob.shape, device=device).normal_(0.0, 0.02, generator=g)
        hm[s:e] = blob + noise
    conf = torch.empty((n_frames, n_views, n_joints), device=device).uniform_(0.05, 1.0, generator=g)
    return {"heatmaps": hm, "confidences": conf,
            "proj": torch.from_numpy(proj_img).to(device), "joints_3d": X}


def device_volumetric_case(seed, n_frames, n_views, device, n_channels=32, n_joints=N_JOINTS,
                           volume_size=64, cuboid_side=2500.0, heatmap_size=HEATMAP_SIZE):
    """Same distribution as `volumetric_case`, with the large tensors drawn in HBM."""
    import torch
    rng = np.random.default_rng(seed)
    K, R, t = make_cameras(rng, n_frames, n_views)
    proj_hm = projection_matrices(K, R, t, heatmap_size / IMAGE_SIZE)
    X = make_joints_3d(rng, n_frames, n_joints)
    base = X[:, PELVIS_JOINT if n_joints > PELVIS_JOINT else 0, :]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    S = volume_size
    feats = torch.empty((n_frames, n_views, n_channels, heatmap_size, heatmap_size), device=device).normal_(generator=g)
    vconf = torch.empty((n_frames, n_views, n_channels), device=device).uniform_(0.0, 1.0, generator=g)
    coord = torch.empty((n_frames, S, S, S, 3), dtype=torch.float32, device=device)
    idx = torch.arange(S, dtype=torch.float32, device=device)
    step = float(np.float32(cuboid_side / (S - 1)))
    for b in range(n_frames):
        pos = (base[b] - cuboid_side / 2).astype(np.float32)
        coord[b, :, :, :, 0] = (float(pos[0]) + step * idx).view(S, 1, 1)
        coord[b, :, :, :, 1] = (float(pos[1]) + step * idx).view(1, S, 1)
        coord[b, :, :, :, 2] = (float(pos[2]) + step * idx).view(1, 1, S)
    vol = torch.empty((n_frames, n_joints, S, S, S), device=device).normal_(generator=g)
    rel = (X - (base[:, None, :] - cuboid_side / 2)) / (cuboid_side / (S - 1))
    c = np.clip(np.round(rel).astype(np.int64), 0, S - 1)
    bi = torch.arange(n_frames, device=device).view(-1, 1).expand(n_frames, n_joints)
    ji = torch.arange(n_joints, device=device).view(1, -1).expand(n_frames, n_joints)
    ct = torch.from_numpy(c).to(device)
    vol[bi, ji, ct[..., 0], ct[..., 1], ct[..., 2]] += 12.0
    return {"features": feats, "proj": torch.from_numpy(proj_hm).to(device), "coord_volumes": coord,
            "vol_confidences": vconf, "volumes": vol, "joints_3d": X, "base_points": base}
