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
