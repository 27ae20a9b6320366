"""Frame-level sharding of the hot path across the GPUs of a node.

The reference processes one frame per step in one process; frames are
independent, so the path shards by frame with NO data-path collective: rank r of W takes
frames r, r+W, r+2W, ...  torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo"
in the CPU tests) is used only for (a) the barrier + max-over-ranks timing of bench.py and
(b) gathering the small per-frame results (ROI counts / detections) onto rank 0.
"""
import torch


def frame_shard(num_frames, rank, world_size):
    """Indices of the frames rank `rank` owns (round-robin, so ragged counts differ by <= 1)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, num_frames, world_size))


def owner_of(frame, world_size):
    return frame % world_size


def max_over_ranks(value, dist=None, device="cpu"):
    """max of a python float over all ranks (the bench's step time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_frame_results(local, num_frames, dist=None, device="cpu"):
    """local: dict {frame index: 1-D int64 tensor} for the frames this rank owns.  Returns on
    every rank the list of per-frame tensors in frame order (None if no process group: local only).
    Small host-side gather (<= 300 detections x a few numbers per frame), never on the hot path."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[f] for f in range(num_frames)]
    world = dist.get_world_size()
    width = max([int(v.numel()) for v in local.values()] + [0])
    wt = torch.tensor([width], dtype=torch.int64, device=device)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    width = int(wt.item())
    per_rank = (num_frames + world - 1) // world
    buf = torch.full((per_rank, width + 1), -1, dtype=torch.int64, device=device)
    for slot, f in enumerate(frame_shard(num_frames, dist.get_rank(), world)):
        v = local[f].to(device=device, dtype=torch.int64).reshape(-1)
        buf[slot, 0] = v.numel()
        buf[slot, 1:1 + v.numel()] = v
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = []
    for f in range(num_frames):
        row = out[owner_of(f, world)][f // world]
        res.append(row[1:1 + int(row[0])].cpu())
    return res
