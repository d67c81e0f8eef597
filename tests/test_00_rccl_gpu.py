"""The RCCL path on the one GPU a test box has: a process group of a single rank, started the way N ranks are.

Collected first on purpose: the rank processes are started (poseestimation_amd.launch.spawn_ranks: fresh
interpreters with the env:// variables) before this pytest process has touched the GPU.  Covers
`dist.init_process_group("nccl")`, sharding.all_gather_keypoints and PipelinedKeypointGather on device tensors,
and the sharded workload of bench.py (65 536 frames in the benchmark; 256 here) with its keypoints checked
against the float64 oracle.
"""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _spawn_one_rank(command, timeout):
    from poseestimation_amd import launch
    out, err = io.StringIO(), io.StringIO()
    rc = launch.spawn_ranks(1, command, timeout=timeout, stdout=out, stderr=err)
    assert rc == 0, "rank exited with %d\n%s\n%s" % (rc, out.getvalue()[-2000:], err.getvalue()[-4000:])
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.getvalue()
    return json.loads(lines[0])


def test_rccl_process_group_of_one_rank_gathers_keypoints():
    rec = _spawn_one_rank([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_worker.py"), ROOT], 300)
    assert rec == {"backend": "nccl", "world": 1, "ok": True}


def test_sharded_workload_under_a_launcher_matches_the_oracle(tmp_path):
    """bench.py --workload sharded as one launched rank (process group "nccl" up, the all-gather of the
    (frames, 34, 3) keypoints is an RCCL call), 256 frames; the gathered keypoints of a few frames against the
    oracle: algebraic leg within 1e-3 mm, volumetric leg (cuboid centred on the pelvis the algebraic leg found)
    within 1e-3 mm."""
    dump = str(tmp_path / "keypoints.npy")
    rec = _spawn_one_rank([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "sharded",
                           "--frames", "256", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                           "--dump-keypoints", dump], 600)
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["config"]["frames_total"] == 256
    assert rec["extra"].get("keypoint_all_gather_alone_ms") is not None     # the collective really ran
    kp = np.load(dump)
    assert kp.shape == (256, 34, 3) and np.isfinite(kp).all()

    # the inputs bench.py drew (same seeds, same device generator), then the oracle on a few frames
    import torch

    import mvn_oracle as orc
    from poseestimation_amd import synth
    dev = torch.device("cuda:0")
    a = synth.device_algebraic_case(1234 + 3, 256, 4, dev)
    v = synth.device_volumetric_case(1234 + 3, 64, 4, dev)
    frames = [0, 63, 64, 200, 255]
    hm = a["heatmaps"][frames].cpu().numpy()
    conf = a["confidences"][frames].cpu().numpy()
    proj = a["proj"][frames].cpu().numpy()
    want3d = orc.algebraic_forward(hm, conf, proj)[0]
    assert np.abs(kp[frames, :17] - want3d).max() < 1e-3
    for i, f in enumerate(frames):
        vols = v["volumes"][f % 64].cpu().numpy()[None]                  # the 64-frame input block is re-used
        centre = kp[f, synth.PELVIS_JOINT].astype(np.float64)            # what the volumetric leg centred its cuboid on
        coord = orc.build_coord_volume(centre, 2500.0, 64)[None]
        want_v = orc.integrate_tensor_3d_with_coordinates(vols, coord)[0]
        assert np.abs(kp[f, 17:] - want_v[0]).max() < 1e-3, f
