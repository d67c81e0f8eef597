"""Same functions and signatures as the reference's mvn/utils/op.py, running on the mvx HIP kernels.

All functions take HIP device tensors and return freshly allocated float32 tensors on the same device, launched
on torch's current stream.  The volumetric functions (unproject_heatmaps, integrate_tensor_3d,
integrate_tensor_3d_with_coordinates) are differentiable with respect to what the reference trains through them --
the maps, the view confidences and the volumes (train.py:281-290) -- by backward kernels of their own;
integrate_tensor_2d, and projection matrices or coordinate volumes that require grad, have no backward kernel and
raise when autograd is recording, rather than drop the gradient.
"""
import torch
from torch.autograd.function import once_differentiable

from poseestimation_amd import _cabi


class VoxelGrid:
    """Parameters of a regular voxel grid on a HIP device: what mvx_build_coord_volumes turns into a coordinate
    volume.  position, center (n_frames, 3) and rotation (n_frames, 9) are float32 tensors; step = voxel pitch per
    axis; swap_yz_flip = the CMU-to-Human3.6M axis transfer."""

    def __init__(self, position, center, rotation, step, swap_yz_flip, shape):
        self.position, self.center, self.rotation = position, center, rotation
        self.step = tuple(float(x) for x in step)
        self.swap_yz_flip = int(bool(swap_yz_flip))
        self.shape = tuple(int(x) for x in shape)


def attach_grid(coord_volumes, grid):
    """Marks `coord_volumes` (n_frames, x, y, z, 3) as the materialised form of `grid`.  The functions of this module
    then evaluate the grid inside their kernels instead of reading the tensor (same float32 operation order: the
    results are bit-identical), as long as the tensor has not been modified in place since and is passed on as the
    same object (a slice or copy is an ordinary tensor again)."""
    coord_volumes._mvx_grid = (grid, coord_volumes._version)
    return coord_volumes


def _grid_of(coord_volumes, n_frames, vshape):
    tag = getattr(coord_volumes, "_mvx_grid", None)
    if tag is None or tag[1] != coord_volumes._version:
        return None
    grid = tag[0]
    if grid.shape != tuple(vshape) or grid.position.shape[0] != n_frames or grid.position.device != coord_volumes.device:
        return None
    return grid


def _as_input_dtype(like, *results):
    """float64 inputs give float64 results, as the reference's torch arithmetic does (mvn/utils/op.py:34-35: the
    float32 index ranges are promoted to the dtype of the heatmaps).  The kernels read float32 (a float64 input is
    rounded to it first) and accumulate in float64, so such results carry float32 accuracy in a float64 tensor."""
    if torch.is_tensor(like) and like.dtype == torch.float64:
        return tuple(r.double() if r is not None else None for r in results)
    return results


def _soft_argmax_2d(heatmaps, softmax, multiplier, return_heatmaps):
    _cabi.forward_only("integrate_tensor_2d", heatmaps)
    hm = _cabi.device_f32(heatmaps, "heatmaps")
    if hm.dim() != 4:
        raise ValueError("heatmaps must have shape (batch, n_heatmaps, h, w), got %s" % (tuple(hm.shape),))
    n, k, h, w = hm.shape
    coords = torch.empty((n, k, 2), dtype=torch.float32, device=hm.device)
    out = torch.empty_like(hm) if return_heatmaps else None
    lib = _cabi.load()
    with torch.cuda.device(hm.device):
        st = lib.mvx_soft_argmax_2d(_cabi.ptr(hm), n * k, h, w, float(multiplier), int(bool(softmax)),
                                    _cabi.ptr(coords), _cabi.ptr(out), _cabi.stream(hm.device))
    _cabi.check(st, "mvx_soft_argmax_2d")
    return _as_input_dtype(heatmaps, coords, out)


def integrate_tensor_2d(heatmaps, softmax=True, *, multiplier=1.0, return_heatmaps=True):
    """Soft-argmax of (batch, n_heatmaps, h, w) heatmaps; reference: mvn/utils/op.py:11-47.

    Returns (coordinates (batch, n_heatmaps, 2) as [x, y], heatmaps): the softmax-normalised maps for
    softmax=True, the ReLU'd maps for softmax=False (coordinates are then divided by the mass).
    Keyword-only extras: `multiplier` folds the caller's `heatmaps * multiplier` into the kernel (each
    element is rounded to float32 after the multiply, as the separate torch op would);
    `return_heatmaps=False` skips the write of the normalised maps and returns None for them.
    """
    return _soft_argmax_2d(heatmaps, softmax, multiplier, return_heatmaps)


class _SoftArgmax3D(torch.autograd.Function):
    """The 3-D soft-argmax with the backward torch autograd derives for mvn/utils/op.py:50-96 (gradient of both
    outputs with respect to the volumes)."""

    @staticmethod
    def forward(ctx, volumes, coord_volumes, softmax, multiplier):
        coords, out = _soft_argmax_3d_run(volumes, coord_volumes, softmax, multiplier, True)
        cv = _cabi.device_f32(coord_volumes, "coord_volumes") if coord_volumes is not None else None
        ctx.save_for_backward(out, cv)
        ctx.softmax, ctx.multiplier = softmax, multiplier
        ctx.set_materialize_grads(False)
        return coords, out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_coords, grad_out):
        out, cv = ctx.saved_tensors
        if grad_coords is None and grad_out is None:
            return None, None, None, None
        b, n, sx, sy, sz = out.shape
        gc = _cabi.device_f32(grad_coords, "grad_coords") if grad_coords is not None else None
        go = _cabi.device_f32(grad_out, "grad_out") if grad_out is not None else None
        grad = torch.empty_like(out)
        lib = _cabi.load()
        ws = torch.empty(max(lib.mvx_soft_argmax_3d_backward_workspace_bytes(b, n, sx, sy, sz), 8) // 8,
                         dtype=torch.float64, device=out.device)
        with torch.cuda.device(out.device):
            st = lib.mvx_soft_argmax_3d_backward(
                _cabi.ptr(out), _cabi.ptr(cv), b, n, sx, sy, sz, float(ctx.multiplier), int(ctx.softmax), _cabi.ptr(gc),
                _cabi.ptr(go), _cabi.ptr(grad), _cabi.ptr(ws), ws.numel() * 8, _cabi.stream(out.device))
        _cabi.check(st, "mvx_soft_argmax_3d_backward")
        return grad, None, None, None


def _soft_argmax_3d(volumes, coord_volumes, softmax, multiplier, return_volumes):
    name = "integrate_tensor_3d" if coord_volumes is None else "integrate_tensor_3d_with_coordinates"
    _cabi.forward_only(name + " (coord_volumes)", coord_volumes)
    if _cabi.needs_grad(volumes):
        vol = volumes if volumes.dtype == torch.float32 else volumes.float()
        coords, out = _SoftArgmax3D.apply(vol, coord_volumes, bool(softmax), float(multiplier))
        return _as_input_dtype(volumes, coords, out if return_volumes else None)
    return _soft_argmax_3d_run(volumes, coord_volumes, softmax, multiplier, return_volumes)


def _soft_argmax_3d_run(volumes, coord_volumes, softmax, multiplier, return_volumes):
    vol = _cabi.device_f32(volumes, "volumes")
    if vol.dim() != 5:
        raise ValueError("volumes must have shape (batch, n_volumes, x, y, z), got %s" % (tuple(vol.shape),))
    b, n, sx, sy, sz = vol.shape
    cv = None
    if coord_volumes is not None:
        cv = _cabi.device_f32(coord_volumes, "coord_volumes")
        if tuple(cv.shape) != (b, sx, sy, sz, 3):
            raise ValueError("coord_volumes must have shape %s, got %s" % ((b, sx, sy, sz, 3), tuple(cv.shape)))
        _cabi.same_device(vol, "volumes", coord_volumes=cv)
    coords = torch.empty((b, n, 3), dtype=torch.float32, device=vol.device)
    out = torch.empty_like(vol) if return_volumes else None
    lib = _cabi.load()
    ws_bytes = lib.mvx_soft_argmax_3d_workspace_bytes(b, n, sx, sy, sz)
    ws = torch.empty((max(ws_bytes, 8) + 7) // 8, dtype=torch.float64, device=vol.device)
    grid = _grid_of(coord_volumes, b, (sx, sy, sz)) if coord_volumes is not None else None
    with torch.cuda.device(vol.device):
        if grid is not None:   # the grid is evaluated in the kernel: its 12 bytes per voxel are not read
            st = lib.mvx_soft_argmax_3d_grid(_cabi.ptr(vol), _cabi.ptr(grid.position), _cabi.ptr(grid.center),
                                             _cabi.ptr(grid.rotation), grid.step[0], grid.step[1], grid.step[2],
                                             grid.swap_yz_flip, b, n, sx, sy, sz, float(multiplier), int(bool(softmax)),
                                             _cabi.ptr(coords), _cabi.ptr(out), _cabi.ptr(ws), ws.numel() * 8,
                                             _cabi.stream(vol.device))
        else:
            st = lib.mvx_soft_argmax_3d(_cabi.ptr(vol), _cabi.ptr(cv), b, n, sx, sy, sz, float(multiplier),
                                        int(bool(softmax)), _cabi.ptr(coords), _cabi.ptr(out), _cabi.ptr(ws),
                                        ws.numel() * 8, _cabi.stream(vol.device))
    _cabi.check(st, "mvx_soft_argmax_3d")
    return _as_input_dtype(volumes, coords, out)


def integrate_tensor_3d(volumes, softmax=True, *, multiplier=1.0, return_volumes=True):
    """Soft-argmax in voxel-index space; reference: mvn/utils/op.py:50-81."""
    return _soft_argmax_3d(volumes, None, softmax, multiplier, return_volumes)


def integrate_tensor_3d_with_coordinates(volumes, coord_volumes, softmax=True, *, multiplier=1.0,
                                         return_volumes=True):
    """Expectation of the world coordinates under the (softmax of the) volumes; reference:
    mvn/utils/op.py:84-96.  volumes (batch, n, x, y, z), coord_volumes (batch, x, y, z, 3) ->
    (coordinates (batch, n, 3), volumes).  softmax=False is a ReLU without renormalisation, as in the
    reference.  Keyword-only extras as in `integrate_tensor_2d`."""
    if coord_volumes is None:
        raise TypeError("coord_volumes is required")
    return _soft_argmax_3d(volumes, coord_volumes, softmax, multiplier, return_volumes)


_AGGREGATIONS = {"sum": _cabi.AGG_SUM, "max": _cabi.AGG_MAX, "softmax": _cabi.AGG_SOFTMAX}


def _is_channels_last(t):
    """True for a float32 HIP tensor (batch, views, c, h, w) whose memory is (batch, views, h, w, c): what
    `.view(batch, views, c, h, w)` of a torch.channels_last (batch*views, c, h, w) tensor looks like."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 5):
        return False
    _, v, c, h, w = t.shape
    return (c % 4 == 0 and t.data_ptr() % 16 == 0 and min(v, c, h, w) > 1
            and tuple(t.stride()) == (v * h * w * c, h * w * c, 1, w * c, c))


def unproject_heatmaps(heatmaps, proj_matricies, coord_volumes, volume_aggregation_method='sum', vol_confidences=None):
    """Unprojects (batch, n_views, n_joints, h, w) maps into (batch, n_joints, x, y, z) volumes;
    reference: mvn/utils/op.py:99-166.

    proj_matricies (batch, n_views, 3, 4) project world points to map pixels; coord_volumes
    (batch, x, y, z, 3) hold the world position of every voxel; volume_aggregation_method is 'sum', 'max',
    'softmax' or anything starting with 'conf' (then vol_confidences (batch, n_views, n_joints) weight the
    views).  Any other method raises ValueError, as the reference does.  Maps that are channel-last in memory
    (the output of a backbone run in torch.channels_last) are sampled in place; others are first copied to
    that layout.
    """
    method = volume_aggregation_method
    _aggregation(method)
    _cabi.forward_only("unproject_heatmaps (proj_matricies, coord_volumes)", proj_matricies, coord_volumes)
    if _cabi.needs_grad(heatmaps, vol_confidences):
        hm, vc = heatmaps, vol_confidences
        if torch.is_tensor(hm) and hm.dtype != torch.float32:
            hm = hm.float()
        if torch.is_tensor(vc) and vc.dtype != torch.float32:
            vc = vc.float()
        volumes = _Unproject.apply(hm, proj_matricies, coord_volumes, vc, method)
        return _as_input_dtype(heatmaps, volumes)[0]
    return _unproject_run(heatmaps, proj_matricies, coord_volumes, method, vol_confidences)


def _aggregation(method):
    if method.startswith('conf'):
        return _cabi.AGG_CONF
    if method in _AGGREGATIONS:
        return _AGGREGATIONS[method]
    raise ValueError("Unknown volume_aggregation_method: {}".format(method))


class _Unproject(torch.autograd.Function):
    """The unprojection with the backward torch autograd derives for mvn/utils/op.py:99-166: gradients with respect
    to the maps and, for the confidence aggregations, to the view confidences."""

    @staticmethod
    def forward(ctx, heatmaps, proj_matricies, coord_volumes, vol_confidences, method):
        volumes = _unproject_run(heatmaps, proj_matricies, coord_volumes, method, vol_confidences)
        ctx.save_for_backward(heatmaps, proj_matricies, coord_volumes, vol_confidences)
        ctx.method = method
        return volumes

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_volumes):
        heatmaps, proj_matricies, coord_volumes, vol_confidences = ctx.saved_tensors
        agg = _aggregation(ctx.method)
        b, v, c, h, w = heatmaps.shape
        cp = (c + 3) // 4 * 4
        dev = heatmaps.device
        want_conf = agg == _cabi.AGG_CONF and ctx.needs_input_grad[3]
        feats_cl = None
        if agg in (_cabi.AGG_MAX, _cabi.AGG_SOFTMAX) or want_conf:   # the kernel samples the maps again
            if _is_channels_last(heatmaps):
                feats_cl = heatmaps
            else:   # channel-last copy, channels padded to a multiple of 4 with zeros
                feats_cl = torch.zeros((b, v, h, w, cp), dtype=torch.float32, device=dev)
                feats_cl[..., :c] = heatmaps.permute(0, 1, 3, 4, 2)
        proj = _cabi.device_f32(proj_matricies, "proj_matricies")
        cv = _cabi.device_f32(coord_volumes, "coord_volumes")
        vc = _cabi.device_f32(vol_confidences, "vol_confidences") if agg == _cabi.AGG_CONF else None
        grad = _cabi.device_f32(grad_volumes, "grad_volumes")
        grad_cl = torch.empty((b, v, h, w, cp), dtype=torch.float32, device=dev)
        grad_conf = torch.empty((b, v, c), dtype=torch.float32, device=dev) if want_conf else None
        vshape = tuple(cv.shape[1:4])
        lib = _cabi.load()
        ws_bytes = lib.mvx_unproject_backward_workspace_bytes(b, v, c, h, w, vshape[0], vshape[1], vshape[2], agg)
        ws = torch.empty((ws_bytes + 15) // 16 * 4, dtype=torch.float32, device=dev) if ws_bytes else None
        with torch.cuda.device(dev):
            st = lib.mvx_unproject_backward(
                _cabi.ptr(feats_cl), _cabi.ptr(proj), _cabi.ptr(cv), _cabi.ptr(vc), _cabi.ptr(grad), b, v, c, h, w,
                vshape[0], vshape[1], vshape[2], agg, _cabi.ptr(grad_cl), _cabi.ptr(grad_conf), _cabi.ptr(ws),
                ws.numel() * 4 if ws is not None else 0, _cabi.stream(dev))
        _cabi.check(st, "mvx_unproject_backward")
        # logical (batch, views, channels, h, w) over the channel-last memory the kernel wrote
        grad_maps = grad_cl[..., :c].permute(0, 1, 4, 2, 3) if ctx.needs_input_grad[0] else None
        return grad_maps, None, None, grad_conf, None


def _unproject_run(heatmaps, proj_matricies, coord_volumes, method, vol_confidences):
    agg = _aggregation(method)
    channels_last = _is_channels_last(heatmaps)
    feats = heatmaps if channels_last else _cabi.device_f32(heatmaps, "heatmaps")
    proj = _cabi.device_f32(proj_matricies, "proj_matricies")
    cv = _cabi.device_f32(coord_volumes, "coord_volumes")
    if feats.dim() != 5 or cv.dim() != 5 or cv.shape[-1] != 3:
        raise ValueError("heatmaps must be (batch, views, joints, h, w) and coord_volumes (batch, x, y, z, 3)")
    b, v, c, h, w = feats.shape
    if tuple(proj.shape) != (b, v, 3, 4) or cv.shape[0] != b:
        raise ValueError("proj_matricies must be %s and coord_volumes batch %d" % ((b, v, 3, 4), b))
    vshape = tuple(cv.shape[1:4])
    vc = None
    if agg == _cabi.AGG_CONF:
        if vol_confidences is None:
            raise ValueError("volume_aggregation_method '%s' needs vol_confidences" % method)
        vc = _cabi.device_f32(vol_confidences, "vol_confidences")
        if tuple(vc.shape) != (b, v, c):
            raise ValueError("vol_confidences must have shape %s, got %s" % ((b, v, c), tuple(vc.shape)))
    _cabi.same_device(feats, "heatmaps", proj_matricies=proj, coord_volumes=cv, vol_confidences=vc)
    volumes = torch.empty((b, c) + vshape, dtype=torch.float32, device=feats.device)
    lib = _cabi.load()
    grid = _grid_of(coord_volumes, b, vshape)
    ws, ws_bytes = None, 0
    if not channels_last:   # maps that are not channel-last in memory are first copied to that layout
        ws_bytes = lib.mvx_unproject_workspace_bytes(b, v, c, h, w)
        ws = torch.empty((max(ws_bytes, 16) + 15) // 16 * 4, dtype=torch.float32, device=feats.device)
        ws_bytes = ws.numel() * 4
    dev_stream = _cabi.stream(feats.device)
    with torch.cuda.device(feats.device):
        if grid is not None:   # the grid is evaluated in the kernel: its 12 bytes per voxel are not read
            what = "mvx_unproject_grid"
            st = lib.mvx_unproject_grid(_cabi.ptr(feats), int(channels_last), _cabi.ptr(proj), _cabi.ptr(grid.position),
                                        _cabi.ptr(grid.center), _cabi.ptr(grid.rotation), grid.step[0], grid.step[1],
                                        grid.step[2], grid.swap_yz_flip, _cabi.ptr(vc), b, v, c, h, w, vshape[0],
                                        vshape[1], vshape[2], agg, _cabi.ptr(volumes), _cabi.ptr(ws), ws_bytes, dev_stream)
        elif channels_last:    # the maps are sampled where they lie: no layout copy, no workspace
            what = "mvx_unproject_channels_last"
            st = lib.mvx_unproject_channels_last(_cabi.ptr(feats), _cabi.ptr(proj), _cabi.ptr(cv), _cabi.ptr(vc), b, v, c,
                                                 h, w, vshape[0], vshape[1], vshape[2], agg, _cabi.ptr(volumes), dev_stream)
        else:
            what = "mvx_unproject"
            st = lib.mvx_unproject(_cabi.ptr(feats), _cabi.ptr(proj), _cabi.ptr(cv), _cabi.ptr(vc), b, v, c, h, w,
                                   vshape[0], vshape[1], vshape[2], agg, _cabi.ptr(volumes), _cabi.ptr(ws), ws_bytes,
                                   dev_stream)
    _cabi.check(st, what)
    return _as_input_dtype(heatmaps, volumes)[0]


def gaussian_2d_pdf(coords, means, sigmas, normalize=True):
    """Axis-aligned 2-D Gaussian density at `coords`; reference: mvn/utils/op.py:169-175 (which
    normalises by 2*pi*sigma_x^2; kept as is)."""
    dx = (coords[:, 0] - means[:, 0]) / sigmas[:, 0]
    dy = (coords[:, 1] - means[:, 1]) / sigmas[:, 1]
    value = torch.exp(-(dx * dx + dy * dy) / 2)
    if normalize:
        value = value / (2 * torch.pi * sigmas[:, 0] * sigmas[:, 0])
    return value


def render_points_as_2d_gaussians(points, sigmas, image_shape, normalize=True):
    """One (h, w) Gaussian image per point; reference: mvn/utils/op.py:178-196."""
    h, w = image_shape
    ys = torch.arange(h, device=points.device, dtype=torch.float32).view(1, h, 1)
    xs = torch.arange(w, device=points.device, dtype=torch.float32).view(1, 1, w)
    dx = (xs - points[:, 0].view(-1, 1, 1)) / sigmas[:, 0].view(-1, 1, 1)
    dy = (ys - points[:, 1].view(-1, 1, 1)) / sigmas[:, 1].view(-1, 1, 1)
    images = torch.exp(-(dx * dx + dy * dy) / 2)
    if normalize:
        images = images / (2 * torch.pi * sigmas[:, 0] * sigmas[:, 0]).view(-1, 1, 1)
    return images
