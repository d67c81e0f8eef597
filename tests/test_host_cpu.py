"""CPU-only checks: the C-ABI library loads and exports every symbol of include/mvx.h, the host-side mirror
(camera, helpers, rotation, error types) against the reference's golden vectors, and the frame sharding
including a world_size-2 gloo all-gather."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_cases as gc
from poseestimation_amd import _cabi, sharding, synth
from poseestimation_amd.mvn.utils import multiview, op, volumetric

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mvx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mvx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_cabi.SIGNATURES), "include/mvx.h and the ctypes table disagree"
    lib = _cabi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mvx_version() == int(re.search(r"#define MVX_VERSION (\d+)", header).group(1))
    assert lib.mvx_last_error() is not None


def test_argument_validation_without_a_gpu():
    """Shape checks run before any launch, so they can be exercised on a GPU-less host."""
    lib = _cabi.load()
    assert lib.mvx_soft_argmax_2d(None, 4, 96, 96, 1.0, 1, None, None, None) == _cabi.MVX_ERR_INVALID_ARGUMENT
    assert b"null" in lib.mvx_last_error()
    assert lib.mvx_soft_argmax_2d(None, 0, 96, 96, 1.0, 1, None, None, None) == _cabi.MVX_OK      # empty batch
    assert lib.mvx_soft_argmax_2d(None, -1, 96, 96, 1.0, 1, None, None, None) == _cabi.MVX_ERR_INVALID_ARGUMENT
    assert lib.mvx_triangulate_dlt(None, None, None, 3, 0, 17, None, None) == _cabi.MVX_ERR_INVALID_ARGUMENT
    assert lib.mvx_triangulate_dlt(None, None, None, 0, 4, 17, None, None) == _cabi.MVX_OK
    assert lib.mvx_unproject(None, None, None, None, 1, 4, 32, 96, 96, 4, 4, 4, 9, None, None, 0, None) == _cabi.MVX_ERR_INVALID_ARGUMENT
    assert lib.mvx_unproject(None, None, None, None, 1, 4, 32, 1, 96, 4, 4, 4, 0, None, None, 0, None) == _cabi.MVX_ERR_INVALID_ARGUMENT
    assert lib.mvx_unproject(None, None, None, None, 0, 4, 32, 96, 96, 4, 4, 4, 0, None, None, 0, None) == _cabi.MVX_OK
    assert lib.mvx_unproject_workspace_bytes(64, 4, 32, 96, 96) == 32 * 4 * 96 * 96 * 32 * 4  # rounds of 32 frames
    assert lib.mvx_unproject_workspace_bytes(2, 3, 5, 20, 28) == 2 * 3 * 20 * 28 * 8 * 4
    assert lib.mvx_soft_argmax_3d_workspace_bytes(2, 17, 64, 64, 64) == 2 * 17 * (1024 * 40 + 8)
    assert lib.mvx_soft_argmax_3d(None, None, 1, 17, 0, 64, 64, 1.0, 1, None, None, None, 0, None) == _cabi.MVX_ERR_INVALID_ARGUMENT


def test_hot_ops_refuse_cpu_tensors():
    with pytest.raises(_cabi.MvxError):
        op.integrate_tensor_2d(torch.zeros(1, 1, 8, 8))
    with pytest.raises(_cabi.MvxError):
        multiview.triangulate_batch_of_points(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 2))
    with pytest.raises(_cabi.MvxError):
        op.integrate_tensor_3d_with_coordinates(torch.zeros(1, 1, 2, 2, 2), torch.zeros(1, 2, 2, 2, 3))
    with pytest.raises(ValueError, match="Unknown volume_aggregation_method"):
        op.unproject_heatmaps(torch.zeros(1, 2, 4, 8, 8), torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 2, 2, 3), "mean")
    with pytest.raises(TypeError):
        op.integrate_tensor_2d(np.zeros((1, 1, 8, 8), np.float32))


def test_camera_and_helpers_match_reference():
    g = np.load(os.path.join(gc.GOLDEN_DIR, "helpers.npz"))
    cam = multiview.Camera(g["R"], g["t"], g["K"])
    K_before = g["K"].copy()
    cam.update_after_crop((100, 50, 900, 850))
    assert np.array_equal(cam.K, g["K_crop"]) and np.array_equal(g["K"], K_before)      # constructor copied K
    cam.update_after_resize((800, 800), (384, 384))
    assert np.array_equal(cam.K, g["K_resize"])
    assert np.array_equal(cam.projection, g["P"])
    assert cam.extrinsics.shape == (3, 4) and cam.t.shape == (3, 1)
    assert np.array_equal(multiview.project_3d_points_to_image_plane_without_distortion(g["P"], g["pts"]), g["proj2d"])
    assert np.array_equal(multiview.project_3d_points_to_image_plane_without_distortion(g["P"], g["pts"], False), g["proj3"])
    Pt, pt = torch.from_numpy(g["P"]), torch.from_numpy(g["pts"])
    np.testing.assert_allclose(multiview.project_3d_points_to_image_plane_without_distortion(Pt, pt).numpy(), g["proj2d"], rtol=1e-12)
    h = multiview.euclidean_to_homogeneous(g["pts"])
    assert h.shape == (9, 4) and (h[:, 3] == 1).all()
    assert np.array_equal(multiview.homogeneous_to_euclidean(multiview.euclidean_to_homogeneous(pt)).numpy(), g["pts"])
    for bad in ([1.0, 2.0], None):
        with pytest.raises(TypeError, match="Works only with numpy arrays and PyTorch tensors."):
            multiview.homogeneous_to_euclidean(bad)
    assert np.array_equal(multiview.calc_reprojection_error_matrix(g["kp3"], list(g["kp2_list"]), list(g["Ps"])), g["reproj"])
    P2 = np.stack([g["P"], g["P"]])
    with pytest.raises(AssertionError):
        multiview.triangulate_point_from_multiple_views_linear(P2, g["proj2d"][:3])
    with pytest.raises(AssertionError):
        multiview.Camera(np.eye(4), g["t"], g["K"])


def test_rotation_helpers_match_reference():
    g = np.load(os.path.join(gc.GOLDEN_DIR, "helpers.npz"))
    assert np.array_equal(volumetric.get_rotation_matrix([0, 0, 1], 0.7), g["rot_z"])
    assert np.array_equal(volumetric.get_rotation_matrix([1, 2, -1], 2.1), g["rot_a"])
    assert np.array_equal(volumetric.get_rotation_matrix([0, 0, 1], 0.0), np.eye(3))
    rotated = volumetric.rotate_coord_volume(torch.from_numpy(g["cv"]), 0.7, [0, 0, 1]).numpy()
    np.testing.assert_allclose(rotated, g["rotated"], rtol=0, atol=1e-6)
    box = volumetric.Cuboid3D(np.array([1.0, 2.0, 3.0]), np.array([10.0, 20.0, 30.0]))
    lines = box.build()
    assert len(lines) == 12
    ends = np.array([np.concatenate([l.start_point, l.end_point]) for l in lines])
    assert ends.min() == 1.0 and ends[:, [2, 5]].max() == 33.0
    assert lines[0].color == (255, 0, 0) and lines[3].color == (255, 255, 0)


def test_synthetic_inputs_are_reproducible():
    a = synth.algebraic_case(11, 2, 4)
    b = synth.algebraic_case(11, 2, 4)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["heatmaps"].shape == (2, 4, 17, 96, 96) and a["heatmaps"].dtype == np.float32
    # blobs sit at the projected joints: the arg-max is within a pixel or two of proj/4 where it is inside the map
    centres = synth.project(a["proj"], a["joints_3d"]) / 4
    inside = ((centres > 3) & (centres < 92)).all(-1)
    flat = a["heatmaps"].reshape(2, 4, 17, -1).argmax(-1)
    peak = np.stack([flat % 96, flat // 96], -1)
    assert np.abs(peak - centres)[inside].max() < 3.0
    v = synth.volumetric_case(12, 1, 4, volume_size=8)
    assert v["coord_volumes"].shape == (1, 8, 8, 8, 3) and v["volumes"].shape == (1, 17, 8, 8, 8)
    step = np.float32(2500.0 / 7)
    assert np.allclose(np.diff(v["coord_volumes"][0, :, 0, 0, 0]), step)


def test_shard_bounds_cover_all_frames():
    for n, w in ((65536, 8), (10, 4), (3, 8), (0, 2), (17, 1)):
        spans = [sharding.shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, w)
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)
    single = torch.arange(12.0).view(2, 2, 3)
    assert sharding.all_gather_keypoints(single, 2) is single      # not distributed: identity


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from poseestimation_amd import sharding
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
for total in (7, 8, 1):
    a, b = sharding.shard_bounds(total, rank, world)
    local = torch.arange(a, b, dtype=torch.float32).view(-1, 1, 1).expand(-1, 17, 3).contiguous() * 1.5
    full = sharding.all_gather_keypoints(local, total)
    want = torch.arange(total, dtype=torch.float32).view(-1, 1, 1).expand(-1, 17, 3) * 1.5
    assert full.shape == (total, 17, 3) and torch.equal(full, want), (rank, total)
# pipelined form: several steps in flight, each step's keypoints in its own buffer
pipe = sharding.PipelinedKeypointGather((5, 17, 3), torch.float32, "cpu", depth=2)
bufs = [torch.empty(5, 17, 3) for _ in range(2)]
slots = []
for step in range(5):
    buf = bufs[step % 2]
    if step >= 2:
        got = pipe.result(slots[step - 2])          # also frees the buffer this step is about to overwrite
        assert torch.equal(got[:5], torch.full((5, 17, 3), float(step - 2))) and \
            torch.equal(got[5:], torch.full((5, 17, 3), 100.0 + step - 2)), (rank, step)
    buf.fill_(float(step) + 100.0 * rank)
    slots.append(pipe.submit(buf))
pipe.drain()
assert torch.equal(pipe.recv[0][5:], torch.full((5, 17, 3), 104.0)) and torch.equal(pipe.recv[1][:5], torch.full((5, 17, 3), 3.0))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_all_gather_keypoints_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]


_STUB_RANK = r"""
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from poseestimation_amd import sharding
fail_rank = int(sys.argv[2])
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if rank == fail_rank:
    sys.exit(7)
total = 65536
a, b = sharding.shard_bounds(total, rank, world)
local = torch.arange(a, b, dtype=torch.float32).view(-1, 1, 1).expand(-1, 34, 3).contiguous()   # the stub "step"
full = sharding.all_gather_keypoints(local, total)
ok = bool(torch.equal(full[:, 0, 0], torch.arange(total, dtype=torch.float32)))
print("noise from rank", rank)
if rank == 0:
    print(json.dumps({"n_gpus": world, "frames": total, "ok": ok}))
dist.barrier()
dist.destroy_process_group()
"""


def test_spawn_ranks_runs_a_stub_step_over_gloo(tmp_path):
    """The launcher bench.py uses for --gpus N without an outer launcher: N fresh ranks with the env:// variables,
    rank 0's stdout relayed as the job's output, a failing rank fails the job."""
    import io
    import json

    from poseestimation_amd import launch
    script = tmp_path / "stub_rank.py"
    script.write_text(_STUB_RANK)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for world in (2, 3):
        out, err = io.StringIO(), io.StringIO()
        rc = launch.spawn_ranks(world, [sys.executable, str(script), ROOT, "-1"], timeout=180, env=env, stdout=out, stderr=err)
        assert rc == 0, err.getvalue()
        lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
        assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": world, "frames": 65536, "ok": True}
        assert "noise from rank 1" in err.getvalue() and "noise from rank 1" not in out.getvalue()
    out, err = io.StringIO(), io.StringIO()
    assert launch.spawn_ranks(2, [sys.executable, str(script), ROOT, "1"], timeout=180, env=env, stdout=out, stderr=err) == 7
    assert not launch.in_launched_rank({}) and launch.in_launched_rank({"RANK": "0", "WORLD_SIZE": "2"})
    with pytest.raises(ValueError):
        launch.spawn_ranks(0, ["true"])


def test_bench_refuses_a_gpu_count_it_cannot_deliver():
    """`bench.py --gpus N` never silently runs another number of ranks: with no launcher and fewer visible GPUs, or
    under a launcher that started a different world size, it exits non-zero before touching a GPU."""
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MVX_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 but 0 GPU(s) visible" in r.stderr and r.stdout == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655"))
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and r.stdout == ""


def test_wrappers_refuse_tensors_on_different_devices(monkeypatch):
    """Every wrapper checks that its tensors share one device before it takes raw pointers.  Driven on the CPU
    with the float32 / HIP check switched off and tensors on the `meta` and `cpu` devices."""
    from poseestimation_amd import _cabi
    from poseestimation_amd.mvn.models import triangulation as tri
    from poseestimation_amd.mvn.utils import op
    a = torch.empty(1, device="meta")
    _cabi.same_device(a, "a", b=torch.empty(2, device="meta"), c=None)
    with pytest.raises(_cabi.MvxError, match="b is on cpu but a is on meta"):
        _cabi.same_device(a, "a", b=torch.empty(2))
    assert issubclass(_cabi.MvxError, RuntimeError)      # what torch raises for mixed devices

    monkeypatch.setattr(_cabi, "device_f32", lambda t, name: t)
    monkeypatch.setattr(_cabi, "load", lambda: pytest.fail("a kernel was reached with tensors on different devices"))
    meta = lambda *shape: torch.empty(shape, device="meta")   # noqa: E731
    cpu = lambda *shape: torch.empty(shape)                   # noqa: E731
    with pytest.raises(_cabi.MvxError, match="proj_matricies is on cpu"):
        op.unproject_heatmaps(meta(1, 2, 4, 8, 8), cpu(1, 2, 3, 4), meta(1, 2, 2, 2, 3), "sum")
    with pytest.raises(_cabi.MvxError, match="coord_volumes is on cpu"):
        op.unproject_heatmaps(meta(1, 2, 4, 8, 8), meta(1, 2, 3, 4), cpu(1, 2, 2, 2, 3), "sum")
    with pytest.raises(_cabi.MvxError, match="vol_confidences is on cpu"):
        op.unproject_heatmaps(meta(1, 2, 4, 8, 8), meta(1, 2, 3, 4), meta(1, 2, 2, 2, 3), "conf", cpu(1, 2, 4))
    with pytest.raises(_cabi.MvxError, match="coord_volumes is on cpu"):
        op.integrate_tensor_3d_with_coordinates(meta(1, 3, 2, 2, 2), cpu(1, 2, 2, 2, 3))
    with pytest.raises(_cabi.MvxError, match="proj_matricies is on cpu"):
        multiview.triangulate_batch_of_points(cpu(2, 4, 3, 4), meta(2, 4, 17, 2))
    with pytest.raises(_cabi.MvxError, match="confidences is on cpu"):
        multiview.triangulate_batch_of_points(meta(2, 4, 3, 4), meta(2, 4, 17, 2), cpu(2, 4, 17))
    net = tri.AlgebraicTriangulationNet.__new__(tri.AlgebraicTriangulationNet)
    torch.nn.Module.__init__(net)
    net.use_confidences, net.heatmap_softmax, net.heatmap_multiplier = True, True, 100.0
    with pytest.raises(_cabi.MvxError, match="proj_matricies is on cpu"):
        net.triangulate_from_heatmaps(meta(1, 4, 17, 8, 8), meta(1, 4, 17), cpu(1, 4, 3, 4), (32, 32))
    with pytest.raises(_cabi.MvxError, match="alg_confidences is on cpu"):
        net.triangulate_from_heatmaps(meta(1, 4, 17, 8, 8), cpu(1, 4, 17), meta(1, 4, 3, 4), (32, 32))


def test_heatmap_projections_match_the_camera_loop():
    """The vectorised camera rescale + projection of the volumetric forward against the per-camera
    update_after_resize / projection it replaces; the caller's cameras stay untouched."""
    from copy import deepcopy

    from poseestimation_amd.mvn.models.triangulation import VolumetricTriangulationNet
    rng = np.random.default_rng(77)
    K, R, t = synth.make_cameras(rng, 3, 4)
    cams = [[multiview.Camera(R[v], t[v], K[b, v]) for b in range(3)] for v in range(4)]
    P = VolumetricTriangulationNet._heatmap_projections(cams, (384, 320), (96, 80))
    assert P.shape == (3, 4, 3, 4)
    for v in range(4):
        for b in range(3):
            cam = deepcopy(cams[v][b])
            cam.update_after_resize((384, 320), (96, 80))
            np.testing.assert_allclose(P[b, v], cam.projection, rtol=1e-15, atol=1e-12)
            assert np.array_equal(cams[v][b].K, K[b, v])


def test_patch_reference_swaps_functions_in_a_reference_like_package(tmp_path, monkeypatch):
    """patch_reference() against a stand-in package with the reference's module layout (the reference itself is
    not needed, and is not on the GPU box)."""
    import importlib
    import sys
    import textwrap

    import poseestimation_amd
    from poseestimation_amd.mvn.utils import multiview as mv, op as ops

    root = tmp_path / "fake_reference"
    for d in ("mvn", "mvn/utils", "mvn/datasets", "mvn/models"):
        (root / d).mkdir(parents=True)
        (root / d / "__init__.py").write_text("")
    (root / "mvn/utils/op.py").write_text(textwrap.dedent("""
        def integrate_tensor_2d(*a, **k): return "reference"
        def integrate_tensor_3d(*a, **k): return "reference"
        def integrate_tensor_3d_with_coordinates(*a, **k): return "reference"
        def unproject_heatmaps(*a, **k): return "reference"
        def untouched(): return "reference"
    """))
    (root / "mvn/utils/multiview.py").write_text(textwrap.dedent("""
        def triangulate_batch_of_points(*a, **k): return "reference"
        def triangulate_point_from_multiple_views_linear_torch(*a, **k): return "reference"
    """))
    (root / "mvn/datasets/utils.py").write_text("def prepare_batch(*a): return 'reference'\n"
                                               "def prepare_batch_test(*a): return 'reference'\n"
                                               "def prepare_batch_video(*a): return 'reference'\n")
    (root / "mvn/models/triangulation.py").write_text("import module_that_does_not_exist\n")
    (root / "savgol_filer.py").write_text("class SAVGOLFilter: pass\n")
    monkeypatch.syspath_prepend(str(root))
    for name in [m for m in sys.modules if m == "mvn" or m.startswith("mvn.") or m == "savgol_filer"]:
        monkeypatch.delitem(sys.modules, name)
    try:
        patched = poseestimation_amd.patch_reference()
        ref_op = importlib.import_module("mvn.utils.op")
        assert ref_op.unproject_heatmaps is ops.unproject_heatmaps and ref_op.untouched() == "reference"
        assert importlib.import_module("mvn.utils.multiview").triangulate_batch_of_points is mv.triangulate_batch_of_points
        from poseestimation_amd.mvn.datasets import utils as own_data
        from poseestimation_amd.savgol_filer import SAVGOLFilter
        assert importlib.import_module("mvn.datasets.utils").prepare_batch is own_data.prepare_batch
        assert importlib.import_module("savgol_filer").SAVGOLFilter is SAVGOLFilter
        assert "mvn.utils.op.integrate_tensor_2d" in patched and "savgol_filer.SAVGOLFilter" in patched
        # a reference module that does not import (missing optional dependency) is skipped, not fatal
        assert not any(p.startswith("mvn.models.triangulation") for p in patched)
    finally:
        for name in [m for m in sys.modules if m == "mvn" or m.startswith("mvn.") or m == "savgol_filer"]:
            sys.modules.pop(name, None)
