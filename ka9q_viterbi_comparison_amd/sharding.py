"""Frame sharding across the GPUs of one node: frames are independent, so each rank takes a
contiguous frame range and runs the whole hot path on it; there is NO data-path collective.  torch.distributed
(RCCL on GPUs, gloo in the CPU tests) is used only for the barrier and the max-over-ranks timing."""
import torch
import torch.distributed as dist


def shard_range(total_frames, rank, world):
    """Contiguous range [lo, hi) of global frame ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_range(frames_per_rank, rank):
    """Weak scaling: every rank decodes `frames_per_rank` frames with globally unique frame ids."""
    return rank * frames_per_rank, (rank + 1) * frames_per_rank


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    """The job's elapsed time is the slowest rank's."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
