"""Frame sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Every frame is independent in all four kernels, so the only exchange is an all-gather of the (frames, joints, 3)
keypoints: 204 bytes per frame, latency-bound.  Heatmaps, features and volumes never leave their GPU.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_frames, rank, world_size):
    """Contiguous block [start, stop) of frames for `rank`: the first n_frames % world_size ranks hold one
    extra frame, so shard sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world size %d" % (rank, world_size))
    base, extra = divmod(n_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_frames, world_size):
    return [shard_bounds(n_frames, r, world_size)[1] - shard_bounds(n_frames, r, world_size)[0]
            for r in range(world_size)]


def all_gather_keypoints(local_keypoints, n_frames_total, group=None):
    """local_keypoints (n_local, J, 3) -> (n_frames_total, J, 3) on every rank, in frame order.

    Shards produced by `shard_bounds` differ by at most one frame; each rank pads its block to the largest
    shard so that a single fixed-size all-gather (one RCCL call) moves everything, and the padding rows are
    dropped afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        if local_keypoints.shape[0] != n_frames_total:
            raise ValueError("not distributed: expected all %d frames locally" % n_frames_total)
        return local_keypoints
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_frames_total, world)
    if local_keypoints.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d frames, its shard has %d" % (rank, local_keypoints.shape[0], sizes[rank]))
    biggest = max(sizes)
    tail = local_keypoints.shape[1:]
    send = local_keypoints.contiguous()
    if send.shape[0] != biggest:
        pad = torch.zeros((biggest - send.shape[0],) + tuple(tail), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    gathered = torch.empty((world * biggest,) + tuple(tail), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(gathered, send, group=group)
    if all(s == biggest for s in sizes):
        return gathered
    parts = [gathered[r * biggest: r * biggest + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


class PipelinedKeypointGather:
    """All-gather of equally sized keypoint shards that overlaps with the next step's kernels.

    `submit(local_keypoints)` starts the collective (asynchronously, on the backend's own stream for RCCL) into
    one of `depth` receive buffers and returns the slot; the caller must hand over a different tensor for each
    slot in flight (a step writes its keypoints into buffer k % depth) because the collective reads it while
    the next step already runs.  Before a slot is reused its previous collective is waited for; `result(slot)`
    waits and returns the gathered (world * n_local, J, 3) tensor; `drain()` waits for everything outstanding
    (call it before timing stops)."""

    def __init__(self, local_shape, dtype, device, depth=2, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.depth = depth
        self.recv = [torch.empty((self.world * local_shape[0],) + tuple(local_shape[1:]), dtype=dtype, device=device)
                     for _ in range(depth)]
        self.pending = [None] * depth
        self.count = 0

    def submit(self, local_keypoints):
        slot = self.count % self.depth
        if self.pending[slot] is not None:
            self.pending[slot].wait()
        self.pending[slot] = dist.all_gather_into_tensor(self.recv[slot], local_keypoints, group=self.group,
                                                         async_op=True)
        self.count += 1
        return slot

    def result(self, slot):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        return self.recv[slot]

    def drain(self):
        for slot in range(self.depth):
            self.result(slot)
