"""Generates tests/golden/*.npz by running the REFERENCE's own functions on seeded inputs.

Run in the build container only (the reference tree is mounted read-only at
/root/reference there; it does not exist on the GPU box):

    python oracle/gen_golden.py

Each fixture stores the SHA-256 of the regenerated inputs, the reference's float32 outputs
("ref32") and the outputs of the same reference code fed float64 tensors ("ref64").
`cv2` is absent from the container and is replaced by an empty stub module: nothing on the
triangulation path calls it.  `torch.cuda.synchronize` is replaced by a no-op for the model
forward (the container has no GPU).  TEST INFRASTRUCTURE ONLY.
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

REFERENCE = os.environ.get("MVN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REFERENCE)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import golden_cases as gc  # noqa: E402
import torch_port  # noqa: E402
from mvn.utils import op, multiview, volumetric  # noqa: E402  (the reference)
from mvn.models import triangulation as ref_tri  # noqa: E402


def t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def save(name, **arrays):
    path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_algebraic():
    for name in gc.ALGEBRAIC_CASES:
        case = gc.algebraic_inputs(name)
        hm, conf, proj = case["heatmaps"], case["confidences"], case["proj"]
        B, V, J, h, w = hm.shape
        softmax = case["softmax"]
        out = {"input_sha": gc.sha(hm, conf, proj)}
        # float32 reference, as AlgebraicTriangulationNet.forward drives it
        scaled = t(hm).view(B * V, J, h, w) * 100.0
        kp, hm_out = op.integrate_tensor_2d(scaled, softmax)
        out["coords_ref32"] = kp.numpy()
        out["hm_out_sample_ref32"] = hm_out.view(B, V, J, h, w)[0, :, :, ::7, ::5].numpy()
        kp = kp.view(B, V, J, 2)
        c = t(conf)
        c = c / c.sum(dim=1, keepdim=True)
        c = c + 1e-5
        kpi = torch.zeros_like(kp)
        kpi[..., 0] = kp[..., 0] * (384 / w)
        kpi[..., 1] = kp[..., 1] * (384 / h)
        out["conf_norm_ref32"] = c.numpy()
        out["kp2d_img_ref32"] = kpi.numpy()
        out["kp3d_ref32"] = multiview.triangulate_batch_of_points(t(proj), kpi, c).numpy()
        out["kp3d_noconf_ref32"] = multiview.triangulate_batch_of_points(t(proj), kpi, None).numpy()
        # the same reference code evaluated in float64 on the same float32 inputs
        kp64, hm64 = op.integrate_tensor_2d(scaled.double(), softmax)
        out["coords_ref64"] = kp64.numpy()
        out["hm_out_sample_ref64"] = hm64.view(B, V, J, h, w)[0, :, :, ::7, ::5].numpy()
        p64 = multiview.triangulate_batch_of_points(t(proj).double(), kpi.double(), c.double())
        out["kp3d_ref64"] = p64.numpy()  # float64 maths, stored through the reference's float32 buffer
        save(name, **out)


def gen_volumetric_small():
    case = gc.small_volumetric_inputs()
    feats, proj, coord, vconf, vols = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    out = {"input_sha": gc.sha(feats, proj, coord, vconf, vols)}
    for m in gc.UNPROJECT_METHODS:
        out["unproj_%s_ref32" % m] = op.unproject_heatmaps(t(feats), t(proj), t(coord), m, t(vconf)).numpy()
        out["unproj_%s_ref64" % m] = op.unproject_heatmaps(
            t(feats).double(), t(proj).double(), t(coord).double(), m, t(vconf).double()).numpy()
    for sm in (True, False):
        tag = "softmax" if sm else "relu"
        c32, v32 = op.integrate_tensor_3d_with_coordinates(t(vols), t(coord), softmax=sm)
        c64, v64 = op.integrate_tensor_3d_with_coordinates(t(vols).double(), t(coord).double(), softmax=sm)
        out["int3dc_%s_coords_ref32" % tag] = c32.numpy()
        out["int3dc_%s_vol_ref32" % tag] = v32.numpy()
        out["int3dc_%s_coords_ref64" % tag] = c64.numpy()
        out["int3dc_%s_vol_ref64" % tag] = v64.numpy()
        i32, iv32 = op.integrate_tensor_3d(t(vols), softmax=sm)
        i64, iv64 = op.integrate_tensor_3d(t(vols).double(), softmax=sm)
        out["int3d_%s_coords_ref32" % tag] = i32.numpy()
        out["int3d_%s_coords_ref64" % tag] = i64.numpy()
        out["int3d_%s_vol_ref64" % tag] = iv64.numpy()
    save("vol_small", **out)


def gen_volumetric_full():
    case = gc.full_volumetric_inputs()
    feats, proj, coord, vconf, vols = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    out = {"input_sha": gc.sha(feats, proj, coord, vconf, vols), "stride": np.int64(gc.FULL_VOL_STRIDE)}
    timing = {}
    for m in ("softmax", "conf", "sum", "max"):
        t0 = time.perf_counter()
        r = op.unproject_heatmaps(t(feats), t(proj), t(coord), m, t(vconf))
        timing["unproject_%s_s_per_frame" % m] = time.perf_counter() - t0
        out["unproj_%s_ref32" % m] = r.numpy().reshape(32, -1)[:, ::gc.FULL_VOL_STRIDE]
        if m in ("softmax", "conf"):
            r64 = op.unproject_heatmaps(t(feats).double(), t(proj).double(), t(coord).double(), m, t(vconf).double())
            out["unproj_%s_ref64" % m] = r64.numpy().reshape(32, -1)[:, ::gc.FULL_VOL_STRIDE]
    t0 = time.perf_counter()
    c32, v32 = op.integrate_tensor_3d_with_coordinates(t(vols), t(coord), softmax=True)
    timing["integrate3d_s_per_frame"] = time.perf_counter() - t0
    c64, v64 = op.integrate_tensor_3d_with_coordinates(t(vols).double(), t(coord).double(), softmax=True)
    out["int3dc_coords_ref32"] = c32.numpy()
    out["int3dc_coords_ref64"] = c64.numpy()
    out["int3dc_vol_sample_ref64"] = v64.numpy().reshape(17, -1)[:, ::gc.FULL_VOL_STRIDE]
    save("vol_full", **out)
    return timing


def gen_gradients():
    """Gradients torch autograd derives through the reference's volumetric ops (what train.py:281-290 relies on) for
    the small volumetric case and seeded upstream gradients: float32 as the reference runs, and the same code fed
    float64."""
    case = gc.small_volumetric_inputs()
    feats, proj, coord, vconf, vols = (case[k] for k in ("features", "proj", "coord_volumes", "vol_confidences", "volumes"))
    up = gc.gradient_upstreams()
    out = {"input_sha": gc.sha(feats, proj, coord, vconf, vols, up["volumes"], up["coords"], up["probs"])}
    for dtype, tag in ((torch.float32, "ref32"), (torch.float64, "ref64")):
        for m in gc.UNPROJECT_METHODS:
            f = t(feats, dtype).requires_grad_(True)
            c = t(vconf, dtype).requires_grad_(True)
            vol = op.unproject_heatmaps(f, t(proj, dtype), t(coord, dtype), m, c)
            (vol * t(up["volumes"], dtype)).sum().backward()
            out["unproj_%s_gfeat_%s" % (m, tag)] = f.grad.numpy()
            if m.startswith("conf"):
                out["unproj_%s_gconf_%s" % (m, tag)] = c.grad.numpy()
        for sm in (True, False):
            for fn, name in ((lambda v: op.integrate_tensor_3d_with_coordinates(v, t(coord, dtype), softmax=sm), "int3dc"),
                             (lambda v: op.integrate_tensor_3d(v, softmax=sm), "int3d")):
                # the index-space ReLU divides by the mass: keep it away from zero
                v = t(vols if name == "int3dc" or sm else vols + 0.5, dtype).requires_grad_(True)
                coords, probs = fn(v)
                ((coords * t(up["coords"], dtype)).sum() + (probs * t(up["probs"], dtype)).sum()).backward()
                out["%s_%s_gvol_%s" % (name, "softmax" if sm else "relu", tag)] = v.grad.numpy()
                v2 = v.detach().clone().requires_grad_(True)   # the coordinates alone (the volumes unused by the loss)
                (fn(v2)[0] * t(up["coords"], dtype)).sum().backward()
                out["%s_%s_gvol_coords_only_%s" % (name, "softmax" if sm else "relu", tag)] = v2.grad.numpy()
    save("grads", **out)


def gen_helpers():
    rng = np.random.default_rng(1400)
    K = np.array([[1145.0, 0, 512.3], [0, 1143.2, 515.9], [0, 0, 1]])
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    tv = rng.uniform(-3000, 3000, (3, 1))
    cam = multiview.Camera(R, tv, K)
    cam.update_after_crop((100, 50, 900, 850))
    K_crop = cam.K.copy()
    cam.update_after_resize((800, 800), (384, 384))
    K_resize = cam.K.copy()
    P = cam.projection
    pts = rng.uniform(-1000, 1000, (9, 3))
    proj2d = multiview.project_3d_points_to_image_plane_without_distortion(P, pts)
    proj3 = multiview.project_3d_points_to_image_plane_without_distortion(P, pts, convert_back_to_euclidean=False)
    rot_z = volumetric.get_rotation_matrix([0, 0, 1], 0.7)
    rot_a = volumetric.get_rotation_matrix([1, 2, -1], 2.1)
    cv = rng.standard_normal((3, 4, 5, 3)).astype(np.float32)
    rotated = volumetric.rotate_coord_volume(t(cv), 0.7, [0, 0, 1]).numpy()
    kp3 = rng.uniform(-500, 500, (17, 3))
    kp2_list = [rng.uniform(0, 384, (17, 2)) for _ in range(3)]
    Ps = [rng.standard_normal((3, 4)) for _ in range(3)]
    for p in Ps:
        p[2, 3] += 4000.0
    reproj = multiview.calc_reprojection_error_matrix(kp3, kp2_list, Ps)
    one = multiview.triangulate_point_from_multiple_views_linear(
        np.stack([P, P + 0.01 * rng.standard_normal((3, 4))]), proj2d[:2])
    save("helpers", R=R, t=tv, K=K, K_crop=K_crop, K_resize=K_resize, P=P, pts=pts, proj2d=proj2d, proj3=proj3,
         rot_z=rot_z, rot_a=rot_a, cv=cv, rotated=rotated, kp3=kp3, kp2_list=np.stack(kp2_list), Ps=np.stack(Ps),
         reproj=reproj, dlt_numpy_point=one)


class _StubBackbone(torch.nn.Module):
    """Stands in for PoseResNet: returns preset tensors in the 4-tuple layout of pose_resnet.py:318."""

    def __init__(self, heatmaps, features, alg_conf, vol_conf):
        super().__init__()
        self.out = (heatmaps, features, alg_conf, vol_conf)

    def forward(self, images):
        return self.out


def gen_model_forward():
    # --- algebraic: the 6-tuple of triangulation.py:236 -------------------------------------
    case = gc.algebraic_inputs("alg_b2v4")
    hm, conf, proj = case["heatmaps"], case["confidences"], case["proj"]
    B, V, J, h, w = hm.shape
    net = ref_tri.AlgebraicTriangulationNet.__new__(ref_tri.AlgebraicTriangulationNet)
    torch.nn.Module.__init__(net)
    net.use_confidences = True
    net.heatmap_softmax = True
    net.heatmap_multiplier = 100.0
    net.backbone = _StubBackbone(t(hm).view(B * V, J, h, w), None, t(conf).view(B * V, J), None)
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        images = torch.zeros(B, V, 3, 384, 384)
        with torch.no_grad():
            kp3d, kp2d, hm_out, c, t1, t2 = net(images, t(proj), None)
    finally:
        torch.cuda.synchronize = real_sync
    out = {"alg_input_sha": gc.sha(hm, conf, proj), "alg_kp3d": kp3d.numpy(), "alg_kp2d": kp2d.numpy(),
           "alg_conf": c.numpy(), "alg_hm_sample": hm_out[0, :, :, ::7, ::5].numpy()}

    # --- volumetric: the 7-tuple of triangulation.py:403, small shapes ------------------------
    rng = np.random.default_rng(1500)
    B, V, S, hs, img = 2, 3, 8, 24, 96
    K, R, tv = gc.synth.make_cameras(rng, B, V, image_size=img)
    feats256 = rng.standard_normal((B * V, 256, hs, hs)).astype(np.float32)
    heat = rng.standard_normal((B * V, 17, hs, hs)).astype(np.float32)
    vconf = rng.uniform(0.1, 1.0, (B * V, 32)).astype(np.float32)
    w_pf = (rng.standard_normal((32, 256, 1, 1)) * 0.05).astype(np.float32)
    b_pf = (rng.standard_normal(32) * 0.1).astype(np.float32)
    w_vn = (rng.standard_normal((17, 32, 1, 1, 1)) * 0.5).astype(np.float32)
    kp3 = gc.synth.make_joints_3d(rng, B)
    out.update(vol_K=K, vol_R=R, vol_t=tv, vol_feats_sha=gc.sha(feats256, heat, vconf, w_pf, b_pf, w_vn), vol_kp3=kp3)
    for kind, method, training in (("mpii", "softmax", False), ("coco", "conf_norm", False), ("mpii", "sum", True)):
        net = ref_tri.VolumetricTriangulationNet.__new__(ref_tri.VolumetricTriangulationNet)
        torch.nn.Module.__init__(net)
        net.num_joints = 17
        net.volume_aggregation_method = method
        net.volume_softmax = True
        net.volume_multiplier = 1.0
        net.volume_size = S
        net.cuboid_side = 2500.0
        net.kind = kind
        net.use_gt_pelvis = True
        net.heatmap_softmax = True
        net.heatmap_multiplier = 100.0
        net.transfer_cmu_to_human36m = False
        net.backbone = _StubBackbone(t(heat), t(feats256), None, t(vconf) if method.startswith("conf") else None)
        net.process_features = torch.nn.Sequential(torch.nn.Conv2d(256, 32, 1))
        net.process_features[0].weight.data = t(w_pf)
        net.process_features[0].bias.data = t(b_pf)
        net.volume_net = torch.nn.Conv3d(32, 17, 1, bias=False)
        net.volume_net.weight.data = t(w_vn)
        net.train(training)
        cams = [[multiview.Camera(R[v], tv[v], K[b, v]) for b in range(B)] for v in range(V)]
        batch = {"cameras": cams, "keypoints_3d": [kp3[b] for b in range(B)]}
        np.random.seed(7)
        with torch.no_grad():
            kp, f, vols, vc, cuboids, cvol, base = net(torch.zeros(B, V, 3, img, img), None, batch)
        tag = "vol_%s_%s_%s" % (kind, method, "train" if training else "eval")
        out[tag + "_kp3d"] = kp.numpy()
        out[tag + "_coord_volumes"] = cvol.numpy()
        out[tag + "_base_points"] = base.numpy()
        out[tag + "_volumes"] = vols.numpy()
        out[tag + "_features"] = f.numpy()
        out[tag + "_cuboid_pos"] = np.stack([c.position for c in cuboids])
        if vc is not None:
            out[tag + "_vol_conf"] = vc.numpy()
    save("model_forward", **out)


def time_port_vs_reference():
    """cfg-1 shape (B=8, V=4): reference functions vs the torch port, same inputs, same threads."""
    case = gc.synth.algebraic_case(1234, 8, 4)
    hm, conf, proj = t(case["heatmaps"]), t(case["confidences"]), t(case["proj"])

    def ref_pass():
        kp, maps = op.integrate_tensor_2d((hm * 100.0).view(32, 17, 96, 96), True)
        kp = kp.view(8, 4, 17, 2)
        c = conf / conf.sum(dim=1, keepdim=True) + 1e-5
        kpi = torch.zeros_like(kp)
        kpi[..., 0] = kp[..., 0] * 4.0
        kpi[..., 1] = kp[..., 1] * 4.0
        return multiview.triangulate_batch_of_points(proj, kpi, c), kpi

    def med(fn):
        ts = []
        for i in range(7):
            t0 = time.perf_counter()
            fn()
            if i >= 2:
                ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    with torch.no_grad():
        t_ref = med(ref_pass)
        t_port = med(lambda: torch_port.algebraic_head(hm, conf, proj))
        a, kpa = ref_pass()
        b = torch_port.algebraic_head(hm, conf, proj)
    return {"shape": "B=8,V=4,J=17,96x96", "threads": torch.get_num_threads(),
            "reference_s": t_ref, "port_s": t_port,
            "reference_frames_per_s": 8 / t_ref, "port_frames_per_s": 8 / t_port,
            "max_abs_diff_kp3d_mm": float((a - b[0]).abs().max()), "max_abs_diff_kp2d_px": float((kpa - b[1]).abs().max())}


def gen_ransac():
    """RANSACTriangulationNet: arg-max keypoints, triangulate_ransac per point (both settings of
    direct_optimization) and the whole forward, with Python's random seeded before each."""
    import random
    import warnings
    warnings.simplefilter("ignore", DeprecationWarning)      # random.sample on a set (triangulation.py:97)
    case = gc.ransac_inputs()
    hm, proj, kp = case["heatmaps"], case["proj"], case["keypoints"]
    B, V, J, h, w = hm.shape
    net = ref_tri.RANSACTriangulationNet.__new__(ref_tri.RANSACTriangulationNet)
    torch.nn.Module.__init__(net)
    out = {"input_sha": gc.sha(hm, proj, kp)}
    for direct in (False, True):
        random.seed(gc.RANSAC_RANDOM_SEED)
        pts = np.zeros((B, J, 3))
        masks = np.zeros((B, J), dtype=np.int64)
        for b in range(B):
            for j in range(J):
                X, inl = net.triangulate_ransac(proj[b], kp[b, :, j], direct_optimization=direct)
                pts[b, j] = X
                masks[b, j] = sum(1 << int(v) for v in inl)
        tag = "refined" if direct else "dlt"
        out["points_" + tag] = pts
        out["masks_" + tag] = masks
    # forward with a stub backbone (heatmaps as given); 384-px images
    net.backbone = _StubBackbone(t(hm).view(B * V, J, h, w), None, None, None)
    for direct in (False, True):
        net.direct_optimization = direct
        random.seed(gc.RANSAC_RANDOM_SEED + 1)
        with torch.no_grad():
            kp3d, kp2d, hm_out, conf = net(torch.zeros(B, V, 3, 384, 384), t(proj), None)
        tag = "refined" if direct else "dlt"
        out["forward_kp3d_" + tag] = kp3d.numpy()
        out["forward_kp2d"] = kp2d.numpy()
        out["forward_conf"] = conf.numpy()
    save("ransac", **out)


def gen_datasets():
    from mvn.datasets import utils as ref_data
    items = gc.dataset_items(multiview.Camera)
    out = {}
    for rand in (True, False):
        tag = "rand" if rand else "all"
        np.random.seed(5)
        batch = ref_data.make_collate_fn(rand, 3, 6)(items + [None])
        images, kp, validity, proj = ref_data.prepare_batch(batch, "cpu", None)
        out["images_sha_" + tag] = gc.sha(images.numpy())
        out["detections_" + tag] = batch["detections"]
        out["kp_" + tag] = kp.numpy()
        out["validity_" + tag] = validity.numpy()
        out["proj_" + tag] = proj.numpy()
        np.random.seed(6)
        batch = ref_data.make_collate_fn_test(rand, 3, 6)(items)
        images, proj = ref_data.prepare_batch_test(batch, "cpu", None)
        out["test_images_sha_" + tag] = gc.sha(images.numpy())
        out["test_proj_" + tag] = proj.numpy()
    sample = {"inps": [torch.from_numpy(im.transpose(2, 0, 1).copy()).float() for im in items[0]["images"][:4]],
              "cameras": items[0]["cameras"][:4]}
    images, proj = ref_data.prepare_batch_video(sample, 4, "cpu")
    out["video_images_sha"] = gc.sha(images.numpy())
    out["video_proj"] = proj.numpy()
    save("datasets", **out)


def gen_savgol():
    sys.path.insert(0, REFERENCE)
    from savgol_filer import SAVGOLFilter   # the reference's module (its file name)
    out = {}
    for i, (T, window) in enumerate(gc.SAVGOL_CASES):
        x = gc.savgol_input(i)
        y, _ = SAVGOLFilter(window)(x)
        assert y.dtype == np.float32
        out["y%d" % i] = y
        yt, _ = SAVGOLFilter(window)(torch.from_numpy(x))
        assert torch.equal(yt, torch.from_numpy(y))
    save("savgol", **out)


def _reference_script_lines(path, first, last):
    """Lines first..last (1-based, inclusive) of a reference script, read where it lies, dedented."""
    import textwrap
    with open(path) as f:
        lines = f.readlines()
    return textwrap.dedent("".join(lines[first - 1:last]))


def gen_wire():
    """Runs the reference's own per-frame tail -- the block of real_pipeline_trt.py that appends the new pose,
    filters the last eight, maps joints and axes, scales and builds the message (lines 349-399; the joint names of
    lines 133-135 and the joint order of line 472) -- on a walk of injected poses, and stores the messages it hands
    to its sender.  The script cannot be imported (it opens cameras and TensorRT engines at import), so that block
    is compiled from the file where it lies and executed frame by frame in a namespace that supplies what the
    lines around it define: the pose of the frame, the reference's SAVGOLFilter, args.slide_window_size = 8 (its
    default, :84) and a recorder in place of the socket sender.  Only the resulting numbers are kept."""
    import types
    sys.path.insert(0, REFERENCE)
    from savgol_filer import SAVGOLFilter   # the reference's module (its file name)
    script = os.path.join(REFERENCE, "real_pipeline_trt.py")
    names_src = _reference_script_lines(script, 133, 135)
    order_src = _reference_script_lines(script, 472, 472)
    body = compile(_reference_script_lines(script, 349, 399), "real_pipeline_trt.py:349-399", "exec")
    assert names_src.lstrip().startswith("kpt_s = [") and order_src.lstrip().startswith("point_change = [")

    class Recorder:
        def __init__(self):
            self.saved = []

        def save(self, kpts):
            self.saved.append(kpts)

    msg = Recorder()
    ns = {"np": np, "torch": torch, "msg": msg, "args": types.SimpleNamespace(slide_window_size=8),
          "filter_model": SAVGOLFilter(window_size=8), "frame_num": 0}
    exec(names_src, ns)
    exec(order_src, ns)
    walk = gc.wire_input()
    counts = []
    for pose in walk:
        ns["keypoints_3d_pred"] = torch.from_numpy(pose[None].copy())   # (1, 17, 3), as the model returns it
        before = len(msg.saved)
        exec(body, ns)
        counts.append(len(msg.saved) - before)
    names = list(ns["kpt_s"])
    assert all(list(m) == names for m in msg.saved)
    assert all(isinstance(v, float) for m in msg.saved for xyz in m.values() for v in xyz)
    save("wire", input_sha=gc.sha(walk), names=np.array(names), point_change=np.array(ns["point_change"]),
         counts=np.array(counts), messages=np.array([[m[n] for n in names] for m in msg.saved], dtype=np.float64),
         frame_num_after=np.array(ns["frame_num"]), json_first=np.array(json.dumps(msg.saved[0])),
         json_last=np.array(json.dumps(msg.saved[-1])))


if __name__ == "__main__":
    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    if len(sys.argv) > 1:                     # regenerate selected fixtures only: ransac, datasets, savgol, wire, grads
        for name in sys.argv[1:]:
            {"ransac": gen_ransac, "datasets": gen_datasets, "savgol": gen_savgol, "wire": gen_wire,
             "grads": gen_gradients}[name]()
        sys.exit(0)
    gen_wire()
    gen_ransac()
    gen_datasets()
    gen_savgol()
    gen_algebraic()
    gen_volumetric_small()
    gen_gradients()
    vol_timing = gen_volumetric_full()
    gen_helpers()
    gen_model_forward()
    rec = time_port_vs_reference()
    rec["reference_volumetric_cpu"] = vol_timing
    rec["versions"] = {"torch": torch.__version__, "numpy": np.__version__}
    with open(os.path.join(gc.GOLDEN_DIR, "port_vs_reference.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))
