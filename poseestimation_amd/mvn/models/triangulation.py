"""RANSACTriangulationNet, AlgebraicTriangulationNet and VolumetricTriangulationNet with the reference's
constructor and `forward(images, proj_matricies, batch)` signatures and return tuples
(reference: mvn/models/triangulation.py:19-148, :151-236 and :240-403), running the triangulation maths on
the mvx HIP kernels.

The heatmap backbone and the V2V volume network are not part of this package: they are taken from the
reference's own `mvn.models.pose_resnet` / `mvn.models.v2v` when that tree is importable, or passed in
(`backbone=`, `volume_net=`).  State-dict keys (`backbone.*`, `process_features.*`, `volume_net.*`) are the
reference's, so its checkpoints load with strict=True.  The volumetric model trains as the reference's does
(train.py:281-290): the unprojection and the 3-D soft-argmax have backward kernels for the maps, the view
confidences and the volumes.  The algebraic and RANSAC heads are forward-only (soft-argmax + DLT have no backward
kernel): such a module in training mode called with autograd recording raises instead of returning detached results.
"""
import contextlib
import random
import time

import numpy as np
import torch
from torch import nn

from poseestimation_amd import _cabi
from poseestimation_amd.mvn.utils import op, volumetric


def _reference_backbone(backbone_config, device):
    try:
        from mvn.models import pose_resnet  # the reference's tree, if it is on sys.path
    except ImportError as e:
        raise ImportError(
            "no backbone given and the reference's mvn.models.pose_resnet is not importable; put the reference "
            "checkout on sys.path or pass backbone=<nn.Module returning (heatmaps, features, alg_conf, vol_conf)>"
        ) from e
    return pose_resnet.get_pose_net(backbone_config, device=device)


def _reference_volume_net(in_channels, out_channels):
    try:
        from mvn.models.v2v import V2VModel
    except ImportError as e:
        raise ImportError(
            "no volume_net given and the reference's mvn.models.v2v is not importable; put the reference checkout "
            "on sys.path or pass volume_net=<nn.Module mapping (B,32,S,S,S) to (B,J,S,S,S)>") from e
    return V2VModel(in_channels, out_channels)


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def draw_view_pairs_python(n_points, n_views, n_iters):
    """The draws of `draw_view_pairs` through the interpreter, one `random.sample` call each (the check for the
    native routine)."""
    views = range(n_views)
    sample = random.sample
    pairs = np.empty((n_points, n_iters, 2), dtype=np.uint8)
    for p in range(n_points):
        for i in range(n_iters):
            a, b = sample(views, 2)
            pairs[p, i, 0], pairs[p, i, 1] = (a, b) if a < b else (b, a)
    return pairs


def draw_view_pairs(n_points, n_views, n_iters, out=None):
    """(n_points, n_iters, 2) uint8 view pairs drawn exactly as the reference draws them: for each point in
    order, `n_iters` calls of `sorted(random.sample(views, 2))` on Python's global generator
    (mvn/models/triangulation.py:96-97), so that `random.seed(s)` before a forward gives the reference's
    hypotheses.  (The reference samples from a set, which Python 3.11 no longer accepts; a range of the same
    length consumes the generator identically.)  The draws are made by a native host routine that continues
    the generator's Mersenne Twister state and hands it back, so Python's `random` ends up where the
    reference's calls would have left it.  `out`: a C-contiguous uint8 array of that shape to fill (for instance
    the numpy view of a page-locked tensor) instead of a new one."""
    if not 2 <= n_views <= 255:
        raise ValueError("view pairs are stored as uint8: 2 to 255 views, got %d" % n_views)
    import ctypes
    version, internal, gauss_next = random.getstate()
    state = np.array(internal[:624], dtype=np.uint32)
    pos = ctypes.c_int32(internal[624])
    if out is None:
        pairs = np.empty((n_points, n_iters, 2), dtype=np.uint8)
    else:
        pairs = out
        if pairs.dtype != np.uint8 or pairs.shape != (n_points, n_iters, 2) or not pairs.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape %s" % ((n_points, n_iters, 2),))
    st = _cabi.load().mvx_draw_view_pairs(state.ctypes.data, ctypes.addressof(pos), int(n_views),
                                          int(n_points) * int(n_iters), pairs.ctypes.data)
    _cabi.check(st, "mvx_draw_view_pairs")
    random.setstate((version, tuple(state.tolist()) + (pos.value,), gauss_next))
    return pairs


class ViewPairPrefetcher:
    """Draws the view pairs of the NEXT batch on a host thread while the GPU works on the current one, and uploads
    them from page-locked memory (the draw is sequential host work -- about two Mersenne Twister words per index --
    and the native routine releases the interpreter lock).  The pairs are the ones `draw_view_pairs` gives in the
    same order of calls, so a loop that seeds `random` and then takes one set of pairs per batch sees the
    reference's hypotheses; nothing else may use Python's global generator while a draw is under way.

        pf = ViewPairPrefetcher(batch * joints, views, 10, device); pf.start()
        for batch in batches:
            pairs = pf.take(); pf.start()        # this batch's pairs (device tensor); begin the next draw
            ransac_triangulate_batch(proj, keypoints, pairs, ...)
    """

    def __init__(self, n_points, n_views, n_iters, device):
        self.shape = (int(n_points), int(n_iters), 2)
        self.n_views = int(n_views)
        self.device = torch.device(device)
        self._host = [torch.empty(self.shape, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self._dev = [torch.empty(self.shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._uploaded = [None, None]   # event after the last upload from host buffer i
        self._slot = 0
        self._thread = None
        self._error = None

    def _draw(self, slot):
        try:
            if self._uploaded[slot] is not None:
                self._uploaded[slot].synchronize()   # the upload that last read this buffer has finished
            draw_view_pairs(self.shape[0], self.n_views, self.shape[1], out=self._host[slot].numpy())
        except BaseException as e:   # handed to the thread that calls take()
            self._error = e

    def start(self):
        """Begins drawing the next set of pairs (returns at once)."""
        if self._thread is not None:
            raise RuntimeError("a draw is already under way: take() its result first")
        import threading
        self._thread = threading.Thread(target=self._draw, args=(self._slot,), daemon=True)
        self._thread.start()

    def take(self):
        """Waits for the draw begun by start() and returns its pairs as a uint8 tensor on the device (valid until
        the take() after next)."""
        if self._thread is None:
            raise RuntimeError("start() has not been called")
        self._thread.join()
        self._thread = None
        if self._error is not None:
            e, self._error = self._error, None
            raise e
        slot = self._slot
        self._slot ^= 1
        with torch.cuda.device(self.device):
            self._dev[slot].copy_(self._host[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self._uploaded[slot] = ev
        return self._dev[slot]


def all_view_pairs(n_points, n_views):
    """Every unordered view pair, the same for each point: a deterministic superset of what random sampling can
    draw (needs no host random numbers; not the reference's behaviour)."""
    pairs = np.array([(a, b) for a in range(n_views) for b in range(a + 1, n_views)], dtype=np.uint8)
    return np.broadcast_to(pairs, (n_points,) + pairs.shape)


def ransac_triangulate_batch(proj_matricies, keypoints_2d, view_pairs, reprojection_error_epsilon=15.0,
                             direct_optimization=True):
    """RANSAC triangulation of every (frame, joint) in one launch.

    proj_matricies (B, V, 3, 4) and keypoints_2d (B, V, J, 2) on a HIP device; view_pairs (B * J, n_iters, 2)
    uint8 (numpy or tensor).  Returns (keypoints_3d (B, J, 3) float64, inlier_masks (B, J) int32 bit masks)."""
    proj = _cabi.device_f32(proj_matricies, "proj_matricies")
    if not (torch.is_tensor(keypoints_2d) and keypoints_2d.is_cuda):
        raise _cabi.MvxError("keypoints_2d must be a tensor on a HIP device (no CPU path)")
    b, v, j = keypoints_2d.shape[:3]
    if tuple(proj.shape) != (b, v, 3, 4):
        raise ValueError("proj_matricies must have shape %s, got %s" % ((b, v, 3, 4), tuple(proj.shape)))
    if not 2 <= v <= 32:
        raise ValueError("RANSAC triangulation takes 2 to 32 views, got %d" % v)
    _cabi.same_device(proj, "proj_matricies", keypoints_2d=keypoints_2d)
    points = keypoints_2d.to(torch.float64).contiguous()
    pairs = torch.as_tensor(np.ascontiguousarray(view_pairs) if isinstance(view_pairs, np.ndarray) else view_pairs)
    if pairs.dtype != torch.uint8 or pairs.dim() != 3 or pairs.shape[0] != b * j or pairs.shape[2] != 2:
        raise ValueError("view_pairs must be uint8 of shape (%d, n_iters, 2), got %s %s"
                         % (b * j, pairs.dtype, tuple(pairs.shape)))
    pairs = pairs.to(proj.device).contiguous()
    out = torch.empty((b, j, 3), dtype=torch.float64, device=proj.device)
    masks = torch.empty((b, j), dtype=torch.int32, device=proj.device)
    with torch.cuda.device(proj.device):
        st = _cabi.load().mvx_triangulate_ransac(
            _cabi.ptr(proj), _cabi.ptr(points), _cabi.ptr(pairs), b, v, j, int(pairs.shape[1]),
            float(reprojection_error_epsilon), int(bool(direct_optimization)), _cabi.ptr(out), _cabi.ptr(masks),
            _cabi.stream(proj.device))
    _cabi.check(st, "mvx_triangulate_ransac")
    return out, masks


class RANSACTriangulationNet(nn.Module):
    """Heatmap arg-max keypoints, then RANSAC over view pairs, DLT over the inliers and (with
    config.model.direct_optimization) a Huber-loss refinement.  Reference: mvn/models/triangulation.py:19-148.

    `view_pairs="sampled"` (default) draws the hypotheses from Python's `random` like the reference, one host
    draw per (frame, joint, iteration); `view_pairs="all"` tests every pair instead and needs no host work."""

    def __init__(self, config, device='cuda:0', backbone=None, view_pairs="sampled"):
        super().__init__()
        config.model.backbone.alg_confidences = False
        config.model.backbone.vol_confidences = False
        self.backbone = backbone if backbone is not None else _reference_backbone(config.model.backbone, device)
        self.direct_optimization = config.model.direct_optimization
        if view_pairs not in ("sampled", "all"):
            raise ValueError("view_pairs must be 'sampled' or 'all', got %r" % (view_pairs,))
        self.view_pairs = view_pairs

    def keypoints_from_heatmaps(self, heatmaps, image_shape):
        """(B, V, J, h, w) heatmaps -> (B, V, J, 2) int64 keypoints in image pixels (triangulation.py:48-56)."""
        hm = _cabi.device_f32(heatmaps, "heatmaps")
        b, v, j, h, w = hm.shape
        keypoints = torch.empty((b, v, j, 2), dtype=torch.int64, device=hm.device)
        with torch.cuda.device(hm.device):
            st = _cabi.load().mvx_heatmap_argmax(_cabi.ptr(hm), b * v * j, h, w, float(image_shape[1] / w),
                                                 float(image_shape[0] / h), _cabi.ptr(keypoints),
                                                 _cabi.stream(hm.device))
        _cabi.check(st, "mvx_heatmap_argmax")
        return keypoints

    def forward(self, images, proj_matricies, batch):
        batch_size, n_views = images.shape[:2]
        images = images.view(-1, *images.shape[2:])
        heatmaps, _, _, _ = self.backbone(images)
        image_shape = tuple(images.shape[2:])
        heatmaps = heatmaps.view(batch_size, n_views, *heatmaps.shape[1:])
        n_joints = heatmaps.shape[2]
        with torch.no_grad():
            keypoints_2d = self.keypoints_from_heatmaps(heatmaps, image_shape)
            if self.view_pairs == "sampled":
                pairs = draw_view_pairs(batch_size * n_joints, n_views, 10)
            else:
                pairs = all_view_pairs(batch_size * n_joints, n_views)
            keypoints_3d, _ = ransac_triangulate_batch(proj_matricies, keypoints_2d, pairs, 15.0,
                                                       self.direct_optimization)
        confidences = torch.zeros((batch_size, n_views, n_joints), dtype=torch.float32, device=images.device)
        return keypoints_3d.float(), keypoints_2d, heatmaps, confidences

    def triangulate_ransac(self, proj_matricies, points, n_iters=10, reprojection_error_epsilon=15,
                           direct_optimization=True):
        """One point from (V, 3, 4) projections and (V, 2) keypoints (array-likes) -> (point (3,) float64 numpy,
        sorted inlier view indices).  Reference: mvn/models/triangulation.py:83-148."""
        assert len(proj_matricies) == len(points)
        assert len(points) >= 2
        device = torch.device("cuda", torch.cuda.current_device())
        # float32 projections, which is what forward hands the reference's numpy code as well
        proj = torch.as_tensor(np.asarray(proj_matricies, dtype=np.float32), device=device)[None]
        pts = torch.as_tensor(np.asarray(points, dtype=np.float64), device=device)[None, :, None, :]
        pairs = draw_view_pairs(1, len(points), n_iters)
        out, mask = ransac_triangulate_batch(proj, pts, pairs, reprojection_error_epsilon, direct_optimization)
        bits = int(mask.item()) & 0xffffffff
        inliers = np.array([v for v in range(len(points)) if (bits >> v) & 1])
        return out[0, 0].cpu().numpy(), inliers


def _inference_guard(module):
    """A module in eval mode runs the triangulation kernels without autograd, which is what an inference caller of
    the reference gets under torch.no_grad().  A module in training mode does not switch autograd off: the
    reference back-propagates through these steps (train.py:197-202, 281-290), so with autograd recording the
    volumetric ops record their backward, and the forward-only ones raise instead of handing back results that
    would silently train nothing."""
    return contextlib.nullcontext() if module.training else torch.no_grad()


class AlgebraicTriangulationNet(nn.Module):
    # Not in the reference.  False: every call returns freshly allocated tensors, as the reference does.  True (for
    # frame loops that finish with one frame's results before asking for the next, as real_pipeline_trt.py does): the
    # output tensors of a call are kept per input shape and handed out again by the next call of that shape, which
    # then allocates nothing -- results of an earlier call are overwritten by a later one.
    reuse_outputs = False

    def __init__(self, config, device='cuda:0', backbone=None):
        super().__init__()
        self.use_confidences = config.model.use_confidences
        # the reference rewrites these backbone flags in place (triangulation.py:157-161)
        config.model.backbone.alg_confidences = False
        config.model.backbone.vol_confidences = False
        if self.use_confidences:
            config.model.backbone.alg_confidences = True
        self.backbone = backbone if backbone is not None else _reference_backbone(config.model.backbone, device)
        self.heatmap_softmax = config.model.heatmap_softmax
        self.heatmap_multiplier = config.model.heatmap_multiplier

    def triangulate_from_heatmaps(self, heatmaps, alg_confidences, proj_matricies, image_shape):
        """Everything after the backbone: soft-argmax (multiplier folded in), confidence normalisation,
        keypoint upscale and DLT; one C-ABI call, two kernel launches.

        heatmaps (B, V, J, h, w) raw backbone output; alg_confidences (B, V, J); proj_matricies (B, V, 3, 4)
        for image pixels; image_shape (H, W).  Returns (keypoints_3d (B, J, 3), keypoints_2d (B, V, J, 2) in
        image pixels, normalised heatmaps (B, V, J, h, w), confidences (B, V, J)); see `reuse_outputs` for who owns
        them."""
        _cabi.forward_only("AlgebraicTriangulationNet.triangulate_from_heatmaps", heatmaps, alg_confidences,
                           proj_matricies)
        hm = _cabi.device_f32(heatmaps, "heatmaps")
        b, v, j, h, w = hm.shape
        conf = _cabi.device_f32(alg_confidences, "alg_confidences").reshape(b, v, j)
        proj = _cabi.device_f32(proj_matricies, "proj_matricies")
        if tuple(proj.shape) != (b, v, 3, 4):
            raise ValueError("proj_matricies must have shape %s, got %s" % ((b, v, 3, 4), tuple(proj.shape)))
        _cabi.same_device(hm, "heatmaps", alg_confidences=conf, proj_matricies=proj)
        index = hm.device.index
        key = (b, v, j, h, w, index)
        arena = self.__dict__.setdefault("_arena", {}) if self.reuse_outputs else None   # (shape, device) -> outputs
        slot = arena.get(key) if arena is not None else None
        if slot is None:
            hm_norm = torch.empty_like(hm)
            # one allocation for the small outputs: heatmap-pixel keypoints, image-pixel keypoints, confidences, joints
            n2, n1, n3 = b * v * j * 2, b * v * j, b * j * 3
            small = torch.empty(2 * n2 + n1 + n3, dtype=torch.float32, device=hm.device)
            _, kp_img, conf_norm, kp3d = small.split((n2, n2, n1, n3))
            base = small.data_ptr()
            # results, then the output pointers in the order mvx_algebraic_forward takes them
            slot = ((kp3d.view(b, j, 3), kp_img.view(b, v, j, 2), hm_norm, conf_norm.view(b, v, j)),
                    (hm_norm.data_ptr(), base + 4 * n2, base + 8 * n2, base + 8 * n2 + 4 * n1, base), small)
            if arena is not None:
                arena[key] = slot
        results, out_ptrs, _ = slot
        lib = _cabi.load()
        # (device pointers travel as plain integers: the argument types declared for the entry point convert them)
        if index == torch.cuda.current_device():   # the common case: no device switch around the call
            st = lib.mvx_algebraic_forward(hm.data_ptr(), conf.data_ptr(), proj.data_ptr(), b, v, j, h, w,
                                           self.heatmap_multiplier, 1 if self.heatmap_softmax else 0, image_shape[1] / w,
                                           image_shape[0] / h, *out_ptrs, _cabi.raw_stream(index))
        else:
            with torch.cuda.device(hm.device):
                st = lib.mvx_algebraic_forward(hm.data_ptr(), conf.data_ptr(), proj.data_ptr(), b, v, j, h, w,
                                               self.heatmap_multiplier, 1 if self.heatmap_softmax else 0,
                                               image_shape[1] / w, image_shape[0] / h, *out_ptrs, _cabi.raw_stream(index))
        if st != _cabi.MVX_OK:
            _cabi.check(st, "mvx_algebraic_forward")
        return results

    def forward(self, images, proj_matricies, batch):
        device = images.device
        batch_size, n_views = images.shape[:2]
        images = images.view(-1, *images.shape[2:])

        _sync(device)
        if self.use_confidences:
            heatmaps, _, alg_confidences, _ = self.backbone(images)
        else:
            heatmaps, _, _, _ = self.backbone(images)
            alg_confidences = torch.ones(batch_size * n_views, heatmaps.shape[1], dtype=torch.float32, device=device)
        _sync(device)
        t1 = time.time()

        image_shape = tuple(images.shape[2:])
        heatmaps = heatmaps.view(batch_size, n_views, *heatmaps.shape[1:])
        alg_confidences = alg_confidences.view(batch_size, n_views, *alg_confidences.shape[1:])
        with _inference_guard(self):
            keypoints_3d, keypoints_2d, heatmaps, alg_confidences = self.triangulate_from_heatmaps(
                heatmaps, alg_confidences, proj_matricies, image_shape)
        _sync(device)
        t2 = time.time()
        # same 6-tuple as the reference, including its two wall-clock stamps (triangulation.py:236)
        return keypoints_3d, keypoints_2d, heatmaps, alg_confidences, t1, t2


class VolumetricTriangulationNet(nn.Module):
    def __init__(self, config, device='cuda:0', backbone=None, volume_net=None):
        super().__init__()
        self.num_joints = config.model.backbone.num_joints
        self.volume_aggregation_method = config.model.volume_aggregation_method
        self.volume_softmax = config.model.volume_softmax
        self.volume_multiplier = config.model.volume_multiplier
        self.volume_size = config.model.volume_size
        self.cuboid_side = config.model.cuboid_side
        self.kind = config.model.kind
        self.use_gt_pelvis = config.model.use_gt_pelvis
        self.heatmap_softmax = config.model.heatmap_softmax
        self.heatmap_multiplier = config.model.heatmap_multiplier
        self.transfer_cmu_to_human36m = getattr(config.model, "transfer_cmu_to_human36m", False)

        config.model.backbone.alg_confidences = False
        config.model.backbone.vol_confidences = False
        if self.volume_aggregation_method.startswith('conf'):
            config.model.backbone.vol_confidences = True
        self.backbone = backbone if backbone is not None else _reference_backbone(config.model.backbone, device)
        final_layer = getattr(self.backbone, "final_layer", None)
        if final_layer is not None:
            for p in final_layer.parameters():
                p.requires_grad = False
        self.process_features = nn.Sequential(nn.Conv2d(256, 32, 1))
        self.volume_net = volume_net if volume_net is not None else _reference_volume_net(32, self.num_joints)

    def _base_points(self, batch, batch_size):
        """(B, 3) float64 host array: pelvis (joint 6) for "mpii", mid-hip (joints 11, 12) for "coco"."""
        key = 'keypoints_3d' if self.use_gt_pelvis else 'pred_keypoints_3d'
        pts = np.zeros((batch_size, 3), dtype=np.float64)
        for i in range(batch_size):
            kp = np.asarray(batch[key][i])
            if self.kind == "coco":
                pts[i] = (kp[11, :3] + kp[12, :3]) / 2
            elif self.kind == "mpii":
                pts[i] = kp[6, :3]
            else:
                raise ValueError("unknown skeleton kind: %s" % (self.kind,))
        return pts

    def build_coord_volumes(self, base_points_np, device):
        """Coordinate volumes (B, S, S, S, 3), float32, index order [x, y, z]: the cuboid of side
        `cuboid_side` centred on each base point, rotated about the vertical axis by a random angle in
        training mode (one np.random.uniform draw per frame, as the reference) and by 0 otherwise.
        Reference: mvn/models/triangulation.py:341-385 (a per-frame Python loop of meshgrid, scale, rotate
        there; one kernel launch here, same float32 operation order).  Returns (volumes, cuboid positions)."""
        S = self.volume_size
        B = base_points_np.shape[0]
        side = float(self.cuboid_side)
        position = base_points_np - side / 2                         # float64, host
        axis = [0, 1, 0] if self.kind == "coco" else [0, 0, 1]
        thetas = [np.random.uniform(0.0, 2 * np.pi) if self.training else 0.0 for _ in range(B)]
        rots = np.stack([volumetric.get_rotation_matrix(axis, th) for th in thetas]) if B else np.zeros((0, 3, 3))
        # position, centre and rotation travel as one small float32 tensor: [B][3 + 3 + 9]
        packed = np.concatenate([position, base_points_np, rots.reshape(B, 9)], axis=1).astype(np.float32)
        params = torch.from_numpy(packed).to(device)
        pos, center, rot = params[:, 0:3].contiguous(), params[:, 3:6].contiguous(), params[:, 6:15].contiguous()
        coord = torch.empty((B, S, S, S, 3), dtype=torch.float32, device=device)
        step = float(np.float32(side / (S - 1)))
        with torch.cuda.device(device):
            st = _cabi.load().mvx_build_coord_volumes(
                _cabi.ptr(pos), _cabi.ptr(center), _cabi.ptr(rot), B, S, S, S, step, step, step,
                int(bool(self.transfer_cmu_to_human36m)), _cabi.ptr(coord), _cabi.stream(device))
        _cabi.check(st, "mvx_build_coord_volumes")
        # the ops downstream evaluate the grid themselves instead of reading this tensor back (same values)
        op.attach_grid(coord, op.VoxelGrid(pos, center, rot, (step, step, step), self.transfer_cmu_to_human36m, (S, S, S)))
        return coord, position

    @staticmethod
    def _heatmap_projections(cameras, image_shape, heatmap_shape):
        """(B, V, 3, 4) float64 projection matrices of the cameras ([V][B] list) after rescaling their
        intrinsics from the image to the heatmap resolution, without touching the caller's objects.
        Reference: the deepcopy + Camera.update_after_resize + Camera.projection loop of
        mvn/models/triangulation.py:309-316, vectorised over all cameras."""
        n_views, batch_size = len(cameras), len(cameras[0]) if len(cameras) else 0
        K = np.empty((batch_size, n_views, 3, 3), dtype=np.float64)
        Rt = np.empty((batch_size, n_views, 3, 4), dtype=np.float64)
        for v in range(n_views):
            for b in range(batch_size):
                cam = cameras[v][b]
                K[b, v] = cam.K
                Rt[b, v, :, :3] = cam.R
                Rt[b, v, :, 3] = np.asarray(cam.t).reshape(3)
        rx, ry = heatmap_shape[1] / image_shape[1], heatmap_shape[0] / image_shape[0]
        K[:, :, 0, 0] *= rx    # fx, cx scale with the width ratio; fy, cy with the height ratio
        K[:, :, 0, 2] *= rx
        K[:, :, 1, 1] *= ry
        K[:, :, 1, 2] *= ry
        return np.matmul(K, Rt)

    def _process_features_channel_last(self, features):
        """`self.process_features(features)` (mvn/models/triangulation.py:388-390: a 1 x 1 convolution 256 -> 32)
        with the result laid out channel-last in memory, which is the layout the unprojection samples in place:
        the convolution is evaluated as one batched matrix product (N, HW, 256) x (256, 32) whose output rows are
        the pixels, so the channel-last maps come out of the product itself and no layout copy follows.  Same
        values up to the summation order of the product; the returned tensor has the usual logical shape
        (N, 32, h, w).  Anything else (another module in `process_features`, a module in training mode with autograd
        recording) goes through the module as it is."""
        conv = self.process_features[0] if len(self.process_features) == 1 else None
        plain = (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
                 and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1)
                 and features.is_cuda and features.dtype == torch.float32)
        if not plain or (self.training and torch.is_grad_enabled()):
            return self.process_features(features)
        n, c, h, w = features.shape
        features = features.detach()
        weight = conv.weight.detach().view(conv.out_channels, c)
        pixels = features.reshape(n, c, h * w).transpose(1, 2)                 # (N, HW, C_in): a view, not a copy
        if conv.bias is not None:
            out = torch.baddbmm(conv.bias.detach().view(1, 1, -1), pixels, weight.t().unsqueeze(0).expand(n, -1, -1))
        else:
            out = torch.matmul(pixels, weight.t())
        return out.view(n, h, w, conv.out_channels).permute(0, 3, 1, 2)        # logical (N, C_out, h, w)

    def forward(self, images, proj_matricies, batch):
        device = images.device
        batch_size, n_views = images.shape[:2]
        images = images.view(-1, *images.shape[2:])
        heatmaps, features, _, vol_confidences = self.backbone(images)

        image_shape, heatmap_shape = tuple(images.shape[2:]), tuple(heatmaps.shape[2:])
        features = features.view(batch_size, n_views, *features.shape[1:])
        if vol_confidences is not None:
            vol_confidences = vol_confidences.view(batch_size, n_views, *vol_confidences.shape[1:])
        if self.volume_aggregation_method == 'conf_norm':
            vol_confidences = vol_confidences / vol_confidences.sum(dim=1, keepdim=True)

        # cameras at heatmap resolution; the projection is built in float64 on the host and cast once
        proj = self._heatmap_projections(batch['cameras'], image_shape, heatmap_shape)
        proj_matricies = torch.from_numpy(proj).float().to(device)

        base_points_np = self._base_points(batch, batch_size)
        base_points = torch.from_numpy(base_points_np).to(device=device, dtype=torch.float32)
        coord_volumes, positions = self.build_coord_volumes(base_points_np, device)
        sides = np.array([self.cuboid_side, self.cuboid_side, self.cuboid_side])
        cuboids = [volumetric.Cuboid3D(positions[i], sides) for i in range(batch_size)]

        features = features.view(-1, *features.shape[2:])
        features = self._process_features_channel_last(features)
        features = features.view(batch_size, n_views, *features.shape[1:])

        with _inference_guard(self):
            volumes = op.unproject_heatmaps(features, proj_matricies, coord_volumes,
                                            volume_aggregation_method=self.volume_aggregation_method,
                                            vol_confidences=vol_confidences)
        volumes = self.volume_net(volumes)
        with _inference_guard(self):
            vol_keypoints_3d, volumes = op.integrate_tensor_3d_with_coordinates(
                volumes, coord_volumes, softmax=self.volume_softmax, multiplier=self.volume_multiplier)
        # same 7-tuple as the reference (triangulation.py:403)
        return vol_keypoints_3d, features, volumes, vol_confidences, cuboids, coord_volumes, base_points
