"""Backward kernels of the volumetric operations (mvx_unproject_backward, mvx_soft_argmax_3d_backward) through
torch autograd on the op mirror, against the gradients the reference's own autograd produced
(tests/golden/grads.npz) and the float64 gradient oracle.

Tolerances.  Map gradients are sums of float32 products added by float32 atomics in no fixed order: 2e-5 absolute on
gradients of magnitude 1-3 for the small case (the float32 reference itself is 9e-6 from its float64 evaluation),
and 1e-5 of the largest entry for larger cases.  Volume gradients are float64 arithmetic rounded once: 1e-6 of the
largest entry."""
import os
import types

import numpy as np
import pytest
import torch

import golden_cases as gc
import mvn_oracle as orc
from poseestimation_amd import _cabi, synth
from poseestimation_amd.mvn.models.triangulation import VolumetricTriangulationNet
from poseestimation_amd.mvn.utils import multiview, op

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x.requires_grad_(True) if grad else x


def load(name):
    return np.load(os.path.join(gc.GOLDEN_DIR, name + ".npz"))


@pytest.fixture(params=["slices in LDS", "global atomics"])
def backward_kernel(request):
    """Runs a test once with each of the two unprojection backward kernels."""
    lib = _cabi.load()
    lib.mvx_set_kernel_selection(_cabi.SELECT_NO_BACKWARD_IN_LDS if request.param == "global atomics" else 0)
    yield request.param
    lib.mvx_set_kernel_selection(0)


def small_case():
    case = gc.small_volumetric_inputs()
    up = gc.gradient_upstreams()
    return (case["features"], case["proj"], case["coord_volumes"], case["vol_confidences"], case["volumes"]), up


@pytest.mark.parametrize("method", gc.UNPROJECT_METHODS)
def test_unproject_gradients_match_the_reference_autograd(method, backward_kernel):
    (feats, proj, coord, vconf, _), up = small_case()
    g = load("grads")
    f, c = t(feats, True), t(vconf, True)
    vol = op.unproject_heatmaps(f, t(proj), t(coord), method, c)
    assert vol.requires_grad
    # the forward values are those of the plain call
    assert torch.equal(vol.detach(), op.unproject_heatmaps(t(feats), t(proj), t(coord), method, t(vconf)))
    (vol * t(up["volumes"])).sum().backward()
    assert f.grad.shape == f.shape
    o_maps, o_conf = orc.unproject_heatmaps_backward(feats, proj, coord, up["volumes"], method, vconf)
    got = f.grad.cpu().numpy()
    assert np.abs(got - o_maps).max() < 2e-5, np.abs(got - o_maps).max()
    assert np.abs(got - g["unproj_%s_gfeat_ref32" % method]).max() < 5e-5
    assert np.abs(got - g["unproj_%s_gfeat_ref64" % method]).max() < 2e-5
    if method.startswith("conf"):
        gc_ = c.grad.cpu().numpy()
        assert np.abs(gc_ - o_conf).max() < 1e-4, np.abs(gc_ - o_conf).max()          # entries up to 21
        assert np.abs(gc_ - g["unproj_%s_gconf_ref32" % method]).max() < 2e-4
    else:
        assert c.grad is None


@pytest.mark.parametrize("softmax", [True, False])
@pytest.mark.parametrize("with_coordinates", [True, False])
def test_soft_argmax_3d_gradients_match_the_reference_autograd(softmax, with_coordinates):
    (_, _, coord, _, vols), up = small_case()
    g = load("grads")
    tag = "%s_%s" % ("int3dc" if with_coordinates else "int3d", "softmax" if softmax else "relu")
    if not with_coordinates and not softmax:
        vols = vols + np.float32(0.5)       # as the fixture: the division by the mass stays away from zero
    cv = coord if with_coordinates else None

    def run(v):
        if with_coordinates:
            return op.integrate_tensor_3d_with_coordinates(v, t(coord), softmax=softmax)
        return op.integrate_tensor_3d(v, softmax=softmax)

    v = t(vols, True)
    coords, probs = run(v)
    assert coords.requires_grad and probs.requires_grad
    with torch.no_grad():
        plain = run(t(vols))
    assert torch.equal(coords.detach(), plain[0]) and torch.equal(probs.detach(), plain[1])
    ((coords * t(up["coords"])).sum() + (probs * t(up["probs"])).sum()).backward()
    want = orc.integrate_tensor_3d_backward(vols, cv, up["coords"], up["probs"], softmax)
    scale = np.abs(want).max()
    assert np.abs(v.grad.cpu().numpy() - want).max() <= 1e-6 * scale
    assert np.abs(v.grad.cpu().numpy() - g[tag + "_gvol_ref64"]).max() <= 1e-6 * scale
    assert np.abs(v.grad.cpu().numpy() - g[tag + "_gvol_ref32"]).max() <= 2e-6 * scale
    # a loss that uses the coordinates alone, and one that uses the volumes alone
    v2 = t(vols, True)
    (run(v2)[0] * t(up["coords"])).sum().backward()
    want = orc.integrate_tensor_3d_backward(vols, cv, up["coords"], None, softmax)
    assert np.abs(v2.grad.cpu().numpy() - want).max() <= 1e-6 * np.abs(want).max()
    assert np.abs(v2.grad.cpu().numpy() - g[tag + "_gvol_coords_only_ref64"]).max() <= 1e-6 * np.abs(want).max()
    v3 = t(vols, True)
    (run(v3)[1] * t(up["probs"])).sum().backward()
    want = orc.integrate_tensor_3d_backward(vols, cv, None, up["probs"], softmax)
    assert np.abs(v3.grad.cpu().numpy() - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-30)
    # return_volumes=False and the folded multiplier: d/dv of f(m v) = m f'(m v)
    v4 = t(vols, True)
    if with_coordinates:
        c4, none = op.integrate_tensor_3d_with_coordinates(v4, t(coord), softmax=softmax, multiplier=3.0,
                                                           return_volumes=False)
    else:
        c4, none = op.integrate_tensor_3d(v4, softmax=softmax, multiplier=3.0, return_volumes=False)
    assert none is None
    (c4 * t(up["coords"])).sum().backward()
    want = 3.0 * orc.integrate_tensor_3d_backward(vols * np.float32(3.0), cv, up["coords"], None, softmax)
    assert np.abs(v4.grad.cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("views", [2, 3, 4])
@pytest.mark.parametrize("method", ["sum", "conf", "max", "softmax"])
def test_unproject_gradients_model_shapes(views, method, backward_kernel):
    """32 channels, 96 x 96 maps, a 16^3 grid (coarse enough for the numpy oracle), maps in both layouts, against
    the float64 oracle."""
    case = synth.volumetric_case(4200 + views, 2, views, volume_size=16, with_volumes=False)
    feats, proj, coord, vconf = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences"))
    rng = np.random.default_rng(77)
    up = rng.standard_normal((2, 32, 16, 16, 16)).astype(np.float32)
    o_maps, o_conf = orc.unproject_heatmaps_backward(feats, proj, coord, up, method, vconf)
    scale = np.abs(o_maps).max()
    results = []
    for layout in ("planar", "channel_last"):
        f = t(feats)
        if layout == "channel_last":
            f = f.view(2 * views, 32, 96, 96).contiguous(memory_format=torch.channels_last).view(2, views, 32, 96, 96)
        f.requires_grad_(True)
        c = t(vconf, True)
        vol = op.unproject_heatmaps(f, t(proj), t(coord), method, c)
        (vol * t(up)).sum().backward()
        got = f.grad.cpu().numpy()
        assert np.abs(got - o_maps).max() <= 1e-5 * scale, (layout, np.abs(got - o_maps).max(), scale)
        if method == "conf":
            assert np.abs(c.grad.cpu().numpy() - o_conf).max() <= 2e-5 * np.abs(o_conf).max()
        results.append(got)
    assert np.abs(results[0] - results[1]).max() <= 1e-5 * scale
    # float64 maps come back with float64 gradients (float32 accuracy), as autograd does for the cast
    f64 = t(feats).double().requires_grad_(True)
    vol = op.unproject_heatmaps(f64, t(proj), t(coord), method, t(vconf))
    assert vol.dtype == torch.float64
    (vol * t(up).double()).sum().backward()
    assert f64.grad.dtype == torch.float64
    assert np.abs(f64.grad.cpu().numpy() - o_maps).max() <= 1e-5 * scale


@pytest.mark.parametrize("method", ["sum", "conf", "max", "softmax"])
def test_unproject_gradients_of_maps_cut_into_bands(method, backward_kernel):
    """112 x 112 maps (12 544 pixels) do not fit one LDS slice: the slice kernel walks them in two bands of rows;
    taps whose two rows fall into different bands are split between the workgroups."""
    case = synth.volumetric_case(4400, 1, 3, n_channels=8, volume_size=16, heatmap_size=112, with_volumes=False)
    feats, proj, coord, vconf = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences"))
    up = np.random.default_rng(5).standard_normal((1, 8, 16, 16, 16)).astype(np.float32)
    f, c = t(feats, True), t(vconf, True)
    (op.unproject_heatmaps(f, t(proj), t(coord), method, c) * t(up)).sum().backward()
    want, want_conf = orc.unproject_heatmaps_backward(feats, proj, coord, up, method, vconf)
    assert np.abs(f.grad.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    if method == "conf":
        assert np.abs(c.grad.cpu().numpy() - want_conf).max() <= 2e-5 * np.abs(want_conf).max()


def test_unproject_gradients_many_views_take_the_any_shape_kernel():
    """20 views with the confidences' gradient wanted: more per-view sums than the pre-pass keeps in LDS, so the call
    goes to the global-atomics kernel by itself."""
    case = synth.volumetric_case(4500, 1, 20, n_channels=4, volume_size=8, heatmap_size=24, with_volumes=False)
    feats, proj, coord, vconf = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences"))
    up = np.random.default_rng(6).standard_normal((1, 4, 8, 8, 8)).astype(np.float32)
    f, c = t(feats, True), t(vconf, True)
    (op.unproject_heatmaps(f, t(proj), t(coord), "conf", c) * t(up)).sum().backward()
    want, want_conf = orc.unproject_heatmaps_backward(feats, proj, coord, up, "conf", vconf)
    assert np.abs(f.grad.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    assert np.abs(c.grad.cpu().numpy() - want_conf).max() <= 2e-5 * np.abs(want_conf).max()


def test_unproject_gradient_is_linear_and_deterministic_enough():
    """backward(2 g) = 2 backward(g) up to rounding, and two runs agree to rounding (the atomics' order varies)."""
    case = synth.volumetric_case(4300, 1, 4, volume_size=32, with_volumes=False)
    feats, proj, coord = (t(case[k]) for k in ("features", "proj", "coord_volumes"))
    up = torch.randn(1, 32, 32, 32, 32, device=DEV)
    grads = []
    for factor in (1.0, 1.0, 2.0):
        f = feats.clone().requires_grad_(True)
        (op.unproject_heatmaps(f, proj, coord, "softmax") * (up * factor)).sum().backward()
        grads.append(f.grad)
    scale = grads[0].abs().max().item()
    assert (grads[0] - grads[1]).abs().max().item() <= 1e-5 * scale
    assert (2 * grads[0] - grads[2]).abs().max().item() <= 2e-5 * scale
    # every pixel no voxel projects near keeps a zero gradient; something is non-zero
    assert scale > 0 and torch.isfinite(grads[0]).all()


def test_backward_on_two_overlapping_streams_matches_the_serial_result():
    """The backward kernels keep no state between calls (workspaces are per call): two training steps issued on two
    streams that overlap give what each gives alone."""
    cases = [synth.volumetric_case(4600 + i, 2, 4, n_joints=5, volume_size=32) for i in range(2)]
    ups = [torch.randn(2, 32, 32, 32, 32, device=DEV), torch.randn(2, 32, 32, 32, 32, device=DEV)]
    upk = [torch.randn(2, 5, 3, device=DEV), torch.randn(2, 5, 3, device=DEV)]

    def run(i):
        case = cases[i]
        f = t(case["features"], True)
        v = t(case["volumes"], True)
        proj, coord = t(case["proj"]), t(case["coord_volumes"])
        op.unproject_heatmaps(f, proj, coord, "softmax").backward(ups[i])
        kp, _ = op.integrate_tensor_3d_with_coordinates(v, coord)
        kp.backward(upk[i])
        return f, v

    serial = [tuple(x.grad.clone() for x in run(i)) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    held = []
    for rep in range(3):
        for i, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                held.append((i, run(i)))
    torch.cuda.synchronize()
    for i, (f, v) in held:
        scale = serial[i][0].abs().max().item()
        assert (f.grad - serial[i][0]).abs().max().item() <= 1e-6 * scale
        assert torch.equal(v.grad, serial[i][1])


def test_arguments_without_a_backward_kernel_still_raise():
    case = synth.volumetric_case(5, 1, 2, n_joints=3, volume_size=8, with_volumes=True)
    f, p, c, v = (t(case[k]) for k in ("features", "proj", "coord_volumes", "volumes"))
    with pytest.raises(_cabi.MvxError, match="forward-only"):
        op.unproject_heatmaps(f, p.clone().requires_grad_(True), c, "sum")
    with pytest.raises(_cabi.MvxError, match="forward-only"):
        op.unproject_heatmaps(f, p, c.clone().requires_grad_(True), "sum")
    with pytest.raises(_cabi.MvxError, match="forward-only"):
        op.integrate_tensor_3d_with_coordinates(v, c.clone().requires_grad_(True))
    with pytest.raises(ValueError, match="Unknown volume_aggregation_method"):
        op.unproject_heatmaps(f.clone().requires_grad_(True), p, c, "median")


class StubBackbone(torch.nn.Module):
    def __init__(self, heatmaps, features, vol_conf):
        super().__init__()
        self.out = (heatmaps, features, None, vol_conf)

    def forward(self, images):
        return self.out


@pytest.mark.parametrize("method", ["softmax", "conf_norm"])
def test_volumetric_model_trains_like_the_torch_ops(method):
    """One training step of VolumetricTriangulationNet (stub backbone, 1 x 1 x 1 volume net): the gradients that
    reach process_features, the volume net and the backbone's outputs, against the same step written with
    differentiable torch ops on the CPU in float64 (grid_sample, softmax, einsum: the reference's formulation)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(910)
    B, V, S, J, hm = 2, 3, 16, 17, 24
    ns = types.SimpleNamespace
    cfg = ns(model=ns(volume_aggregation_method=method, volume_softmax=True, volume_multiplier=1.0, volume_size=S,
                      cuboid_side=2500.0, kind="mpii", use_gt_pelvis=True, heatmap_softmax=True, heatmap_multiplier=100.0,
                      backbone=ns(num_joints=J, alg_confidences=None, vol_confidences=None)))
    img = 4 * hm
    K, R, tv = synth.make_cameras(rng, B, V, image_size=img)
    feats256 = (rng.standard_normal((B * V, 256, hm, hm)) * 0.3).astype(np.float32)
    heat = rng.standard_normal((B * V, J, hm, hm)).astype(np.float32)
    vconf = rng.uniform(0.2, 1.0, (B * V, 32)).astype(np.float32)
    kp3 = synth.make_joints_3d(rng, B, J)
    features_in = t(feats256, True)
    vconf_in = t(vconf, True) if method.startswith("conf") else None
    vol_net = torch.nn.Conv3d(32, J, 1, bias=False)
    net = VolumetricTriangulationNet(cfg, device=DEV, volume_net=vol_net,
                                     backbone=StubBackbone(t(heat), features_in, vconf_in)).to(DEV)
    net.train()
    cams = [[multiview.Camera(R[v], tv[v], K[b, v]) for b in range(B)] for v in range(V)]
    batch = {"cameras": cams, "keypoints_3d": [kp3[b] for b in range(B)]}
    np.random.seed(3)
    kp, f, vols, vc, cuboids, cvol, base = net(torch.zeros(B, V, 3, img, img, device=DEV), None, batch)
    assert kp.requires_grad and vols.requires_grad
    target = t(kp3.astype(np.float32))
    w_vol = torch.randn_like(vols)
    loss = ((kp - target) ** 2).mean() + (vols * w_vol).sum()
    loss.backward()

    # the same step with torch ops, float64, CPU; projections and the (rotated) coordinate volumes as the model built them
    proj = torch.from_numpy(net._heatmap_projections(cams, (img, img), (hm, hm))).float().double()
    coord = cvol.detach().cpu().double()
    pf_w = net.process_features[0].weight.detach().cpu().double().requires_grad_(True)
    pf_b = net.process_features[0].bias.detach().cpu().double().requires_grad_(True)
    vn_w = vol_net.weight.detach().cpu().double().requires_grad_(True)
    x = torch.from_numpy(feats256).double().requires_grad_(True)
    vcr = torch.from_numpy(vconf).double().requires_grad_(True)
    feats = F.conv2d(x, pf_w, pf_b).view(B, V, 32, hm, hm)
    conf = vcr.view(B, V, 32)
    if method == "conf_norm":
        conf = conf / conf.sum(dim=1, keepdim=True)
    volumes = []
    for b in range(B):
        pts = torch.cat([coord[b].reshape(-1, 3), torch.ones(S ** 3, 1, dtype=torch.float64)], dim=1)
        stack = []
        for v in range(V):
            uvw = pts @ proj[b, v].t()
            behind = uvw[:, 2] <= 0.0
            z = torch.where(uvw[:, 2] == 0.0, torch.ones_like(uvw[:, 2]), uvw[:, 2])
            gx = 2 * (uvw[:, 0] / z / hm - 0.5)
            gy = 2 * (uvw[:, 1] / z / hm - 0.5)
            samp = F.grid_sample(feats[b, v].unsqueeze(0), torch.stack([gx, gy], dim=1).view(1, -1, 1, 2),
                                 align_corners=True).view(32, -1)
            stack.append(torch.where(behind[None], torch.zeros_like(samp), samp))
        stack = torch.stack(stack)
        if method.startswith("conf"):
            volumes.append((stack * conf[b].view(V, 32, 1)).sum(0))
        else:
            volumes.append((stack * F.softmax(stack, dim=0)).sum(0))
    volumes = torch.stack(volumes).view(B, 32, S, S, S)
    volumes = F.conv3d(volumes, vn_w)
    p = F.softmax(volumes.view(B, J, -1), dim=2)
    kp_ref = torch.einsum("bnk,bkc->bnc", p, coord.view(B, -1, 3))
    loss_ref = ((kp_ref - target.cpu().double()) ** 2).mean() + (p.view_as(volumes) * w_vol.cpu().double()).sum()
    loss_ref.backward()

    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())

    def close(got, want, rel):
        want = want.numpy()
        err = np.abs(got.detach().cpu().numpy() - want).max()
        assert err <= rel * np.abs(want).max(), (err, np.abs(want).max())

    close(vol_net.weight.grad, vn_w.grad, 2e-4)
    close(net.process_features[0].weight.grad, pf_w.grad, 2e-4)
    close(net.process_features[0].bias.grad, pf_b.grad, 2e-4)
    close(features_in.grad, x.grad, 2e-4)
    if method.startswith("conf"):
        close(vconf_in.grad, vcr.grad, 2e-4)
