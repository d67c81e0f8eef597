"""One rank of the RCCL checks that tests/test_00_rccl_gpu.py starts through launch.spawn_ranks (not collected by
pytest).  Brings up the process group the way the reference's training script does (train.py:424-436:
backend "nccl", init_method "env://"; on ROCm that backend is RCCL), then runs the keypoint all-gather and the
pipelined all-gather of poseestimation_amd.sharding on device tensors.

    python rccl_rank_worker.py <repo root>
"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, sys.argv[1])
from poseestimation_amd import _cabi, sharding  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local_rank = int(os.environ["LOCAL_RANK"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1"
dev = torch.device("cuda", local_rank)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="env://")
assert dist.get_backend() == "nccl" and dist.get_world_size() == world

# 1. one all-gather of unevenly sized shards (the padding rows must be dropped)
total = 256 * world + (1 if world > 1 else 0)
lo, hi = sharding.shard_bounds(total, rank, world)
local = torch.arange(lo, hi, dtype=torch.float32, device=dev).view(-1, 1, 1).expand(-1, 34, 3).contiguous()
full = sharding.all_gather_keypoints(local, total)
assert full.is_cuda and tuple(full.shape) == (total, 34, 3)
assert torch.equal(full[:, 5, 1], torch.arange(total, dtype=torch.float32, device=dev))

# 2. five pipelined steps: every step's keypoints come out of a kernel of the library on the current stream (the
#    confidence-normalising DLT front end), its all-gather runs on RCCL's stream while the next step's kernel runs,
#    and a buffer is rewritten only after the collective that reads it has been waited for
lib = _cabi.load()
frames, views, joints = 64, 4, 17
g = torch.Generator(device=dev)
g.manual_seed(5 + rank)
proj = torch.empty((frames, views, 3, 4), device=dev).normal_(generator=g)
proj[..., 2, 3] += 4000.0
pipe = sharding.PipelinedKeypointGather((frames, joints, 3), torch.float32, dev, depth=2)
bufs = [torch.empty((frames, joints, 3), dtype=torch.float32, device=dev) for _ in range(2)]
want, slots = [], []
for step in range(5):
    pts = torch.empty((frames, views, joints, 2), device=dev).uniform_(0.0, 384.0, generator=g)
    buf = bufs[step % 2]
    if step >= 2:
        got = pipe.result(slots[step - 2])       # also frees the buffer this step is about to overwrite
        assert torch.equal(got[rank * frames:(rank + 1) * frames], want[step - 2]), (rank, step)
    rc = lib.mvx_triangulate_dlt(_cabi.ptr(proj), _cabi.ptr(pts), None, frames, views, joints, _cabi.ptr(buf),
                                 _cabi.stream(dev))
    assert rc == 0, lib.mvx_last_error()
    want.append(buf.clone())
    slots.append(pipe.submit(buf))
pipe.drain()
assert torch.equal(pipe.recv[1][rank * frames:(rank + 1) * frames], want[3])
assert torch.equal(pipe.recv[0][rank * frames:(rank + 1) * frames], want[4])
torch.cuda.synchronize(dev)
if rank == 0:
    print(json.dumps({"backend": dist.get_backend(), "world": world, "ok": True}))
dist.barrier()
dist.destroy_process_group()
