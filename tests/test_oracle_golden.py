"""The numpy float64 oracle against the golden vectors produced by the reference itself
(oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import golden_cases as gc
import mvn_oracle as orc


def load(name):
    return np.load(os.path.join(gc.GOLDEN_DIR, name + ".npz"))


@pytest.mark.parametrize("name", list(gc.ALGEBRAIC_CASES))
def test_algebraic_oracle_matches_reference(name):
    g = load(name)
    case = gc.algebraic_inputs(name)
    hm, conf, proj = case["heatmaps"], case["confidences"], case["proj"]
    assert gc.sha(hm, conf, proj) == str(g["input_sha"]), "synthetic inputs no longer regenerate identically"
    B, V, J, h, w = hm.shape
    scaled = hm * np.float32(100.0)
    coords, hm_out = orc.integrate_tensor_2d(scaled.reshape(B * V, J, h, w), case["softmax"])
    # float64 evaluation of the reference: agreement to rounding of float64 sums
    np.testing.assert_allclose(coords, g["coords_ref64"], rtol=0, atol=1e-9)
    sample = hm_out.reshape(B, V, J, h, w)[0, :, :, ::7, ::5]
    np.testing.assert_allclose(sample, g["hm_out_sample_ref64"], rtol=1e-10, atol=1e-300)
    # float32 reference: within its own float32 noise
    assert np.abs(coords - g["coords_ref32"]).max() < 2e-4
    kp3d, kp2d, _, cn = orc.algebraic_forward(hm, conf, proj, 100.0, case["softmax"])
    # the glue is float32 element-wise: bit-identical given the same float32 keypoints (view sum aside)
    kp2d_from_ref, cn2 = orc.algebraic_glue(g["coords_ref32"].reshape(B, V, J, 2), conf, (384, 384), (h, w))
    assert np.array_equal(cn2, cn)
    np.testing.assert_allclose(cn, g["conf_norm_ref32"], rtol=2.5e-7, atol=0)  # one ulp of the view sum
    assert np.array_equal(kp2d_from_ref, g["kp2d_img_ref32"])
    # DLT on the reference's own float32 keypoints: float64 solve vs the reference's float32 LAPACK solve
    p = orc.triangulate_batch_of_points(proj, g["kp2d_img_ref32"], g["conf_norm_ref32"])
    # 1e-3 mm absolute inside the capture volume; noise heatmaps put points tens of metres away, where
    # one float32 ulp of the reference's output is already 8e-3 mm, hence the relative term
    assert (np.abs(p - g["kp3d_ref32"]) <= 1e-3 + 1e-6 * np.abs(g["kp3d_ref32"])).all()
    # the reference fed float64 tensors never rounds A to float32, which moves the solution by more than
    # its float32 LAPACK noise (5.8e-3 mm at V=2): the float32-A / float64-solve oracle is the closer one
    assert (np.abs(p - g["kp3d_ref64"]) <= 2e-2 + 1e-6 * np.abs(g["kp3d_ref64"])).all()
    p_nc = orc.triangulate_batch_of_points(proj, g["kp2d_img_ref32"], None)
    assert (np.abs(p_nc - g["kp3d_noconf_ref32"]) <= 1e-3 + 1e-6 * np.abs(g["kp3d_noconf_ref32"])).all()
    # end to end (oracle's own keypoints, which differ from ref32's by float32 noise)
    if not (name.startswith("alg_worst") or name.startswith("alg_relu")):
        assert np.abs(kp3d - g["kp3d_ref32"]).max() < 5e-2


def test_single_point_functions():
    case = gc.algebraic_inputs("alg_b2v4")
    g = load("alg_b2v4")
    proj = case["proj"]
    kp, cn = g["kp2d_img_ref32"], g["conf_norm_ref32"]
    p = orc.triangulate_point_from_multiple_views_linear(proj[1], kp[1, :, 5], cn[1, :, 5])
    assert np.abs(p - g["kp3d_ref32"][1, 5]).max() < 1e-3
    with pytest.raises(AssertionError):
        orc.triangulate_point_from_multiple_views_linear(proj[1], kp[1, :3, 5])


def test_unproject_small_all_methods():
    g = load("vol_small")
    case = gc.small_volumetric_inputs()
    args = [case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes")]
    assert gc.sha(*args) == str(g["input_sha"])
    feats, proj, coord, vconf, vols = args
    for m in gc.UNPROJECT_METHODS:
        out = orc.unproject_heatmaps(feats, proj, coord, m, vconf)
        ref64 = g["unproj_%s_ref64" % m]      # float64 maths stored through the reference's float32 buffer
        ref32 = g["unproj_%s_ref32" % m]
        assert out.shape == ref32.shape == (2, 5, 6, 5, 7)
        np.testing.assert_allclose(out, ref64, rtol=0, atol=2e-6)
        assert np.abs(out - ref32).max() < 5e-4
        # the fixture exercises the edge cases it is meant to: voxels behind a camera and outside the maps
    assert (g["unproj_sum_ref32"] == 0).any()
    with pytest.raises(ValueError):
        orc.unproject_heatmaps(feats, proj, coord, "mean", vconf)
    sub = np.array([0, 3, 17, 200])
    part = orc.unproject_heatmaps(feats, proj, coord, "softmax", vconf, voxel_subset=sub)
    full = orc.unproject_heatmaps(feats, proj, coord, "softmax", vconf).reshape(2, 5, -1)
    np.testing.assert_array_equal(part, full[:, :, sub])


def test_integrate_3d_small():
    g = load("vol_small")
    case = gc.small_volumetric_inputs()
    vols, coord = case["volumes"], case["coord_volumes"]
    for sm, tag in ((True, "softmax"), (False, "relu")):
        c, v = orc.integrate_tensor_3d_with_coordinates(vols, coord, sm)
        np.testing.assert_allclose(c, g["int3dc_%s_coords_ref64" % tag], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(v, g["int3dc_%s_vol_ref64" % tag], rtol=1e-12, atol=1e-300)
        r32 = g["int3dc_%s_coords_ref32" % tag]   # ReLU sums are not normalised: ~3e5, so relative
        assert (np.abs(c - r32) <= 5e-2 + 2e-6 * np.abs(r32)).all()
        ci, vi = orc.integrate_tensor_3d(vols, sm)
        np.testing.assert_allclose(ci, g["int3d_%s_coords_ref64" % tag], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(vi, g["int3d_%s_vol_ref64" % tag], rtol=1e-12, atol=1e-300)


def test_volumetric_full_size_sample():
    g = load("vol_full")
    case = gc.full_volumetric_inputs()
    args = [case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes")]
    assert gc.sha(*args) == str(g["input_sha"])
    feats, proj, coord, vconf, vols = args
    sub = np.arange(0, 64 ** 3, gc.FULL_VOL_STRIDE)
    for m in ("softmax", "conf", "sum", "max"):
        out = orc.unproject_heatmaps(feats, proj, coord, m, vconf, voxel_subset=sub)[0]
        assert np.abs(out - g["unproj_%s_ref32" % m]).max() < 1e-4
        if m in ("softmax", "conf"):
            np.testing.assert_allclose(out, g["unproj_%s_ref64" % m], rtol=0, atol=2e-6)
    c, v = orc.integrate_tensor_3d_with_coordinates(vols, coord, True)
    np.testing.assert_allclose(c, g["int3dc_coords_ref64"], rtol=0, atol=1e-8)
    assert np.abs(c - g["int3dc_coords_ref32"]).max() < 5e-3   # the reference's own float32 noise
    np.testing.assert_allclose(v.reshape(17, -1)[:, ::gc.FULL_VOL_STRIDE], g["int3dc_vol_sample_ref64"], rtol=1e-10, atol=1e-300)


def test_helpers():
    g = load("helpers")
    K1 = orc.camera_update_after_crop(g["K"], (100, 50, 900, 850))
    np.testing.assert_array_equal(K1, g["K_crop"])
    K2 = orc.camera_update_after_resize(K1, (800, 800), (384, 384))
    np.testing.assert_allclose(K2, g["K_resize"], rtol=1e-15)
    P = orc.camera_projection(K2, g["R"], g["t"])
    np.testing.assert_allclose(P, g["P"], rtol=1e-14)
    np.testing.assert_allclose(orc.project_3d_points_to_image_plane_without_distortion(P, g["pts"]), g["proj2d"], rtol=1e-12)
    np.testing.assert_allclose(
        orc.project_3d_points_to_image_plane_without_distortion(P, g["pts"], False), g["proj3"], rtol=1e-12)
    np.testing.assert_allclose(orc.get_rotation_matrix([0, 0, 1], 0.7), g["rot_z"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(orc.get_rotation_matrix([1, 2, -1], 2.1), g["rot_a"], rtol=1e-14, atol=1e-16)
    rotated = (g["cv"].astype(np.float64).reshape(-1, 3) @ g["rot_z"].astype(np.float32).astype(np.float64).T).reshape(g["cv"].shape)
    np.testing.assert_allclose(rotated, g["rotated"], rtol=0, atol=1e-6)


def test_port_matches_reference_record():
    """The torch-CPU port used as bench.py's cpu_baseline was timed beside the reference in the
    build container; its outputs were identical and its timing within 2x."""
    import json
    rec = json.load(open(os.path.join(gc.GOLDEN_DIR, "port_vs_reference.json")))
    assert rec["max_abs_diff_kp3d_mm"] == 0.0 and rec["max_abs_diff_kp2d_px"] == 0.0
    assert 0.5 < rec["port_s"] / rec["reference_s"] < 2.0


def test_gradient_oracle_matches_the_reference_autograd():
    """The oracle's backward restatements against the gradients torch autograd derived through the reference's own
    unproject_heatmaps / integrate_tensor_3d[_with_coordinates] (tests/golden/grads.npz).  ref64 = the reference fed
    float64 tensors (its per-view buffers stay float32, hence 2e-7 rather than 1e-12 for softmax / conf)."""
    g = load("grads")
    case = gc.small_volumetric_inputs()
    up = gc.gradient_upstreams()
    feats, proj, coord, vconf, vols = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    assert gc.sha(feats, proj, coord, vconf, vols, up["volumes"], up["coords"], up["probs"]) == str(g["input_sha"])
    for m in gc.UNPROJECT_METHODS:
        g_maps, g_conf = orc.unproject_heatmaps_backward(feats, proj, coord, up["volumes"], m, vconf)
        np.testing.assert_allclose(g_maps, g["unproj_%s_gfeat_ref64" % m], rtol=0, atol=3e-7)
        assert np.abs(g_maps - g["unproj_%s_gfeat_ref32" % m]).max() < 5e-5
        if m.startswith("conf"):
            np.testing.assert_allclose(g_conf, g["unproj_%s_gconf_ref64" % m], rtol=0, atol=2e-6)
        else:
            assert g_conf is None
    for sm in (True, False):
        tag = "softmax" if sm else "relu"
        for cv, v, name in ((coord, vols, "int3dc"), (None, vols if sm else vols + np.float32(0.5), "int3d")):
            out = orc.integrate_tensor_3d_backward(v, cv, up["coords"], up["probs"], sm)
            np.testing.assert_allclose(out, g["%s_%s_gvol_ref64" % (name, tag)], rtol=1e-12, atol=1e-11)
            ref32 = g["%s_%s_gvol_ref32" % (name, tag)]
            assert np.abs(out - ref32).max() <= 1e-6 * np.abs(ref32).max()
            out = orc.integrate_tensor_3d_backward(v, cv, up["coords"], None, sm)
            np.testing.assert_allclose(out, g["%s_%s_gvol_coords_only_ref64" % (name, tag)], rtol=1e-12, atol=1e-11)
