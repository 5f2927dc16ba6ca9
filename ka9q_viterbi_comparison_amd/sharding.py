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


def run_sharded_job(make_shard, *, frames_per_rank, steps, warmup, rank=0, world=1, device=None, reduce_device=None):
    """The N-rank measurement loop of bench.py, independent of what decodes the frames (bench.py passes a HIP-backed
    shard; the world-size-2 gloo test passes a CPU one).

    `make_shard(frame_lo, frames)` builds this rank's shard: the frames with global ids [frame_lo, frame_lo + frames)
    resident where the decoder wants them.  The shard object provides
        one_pass(i)   enqueue / run one full decode pass (init + update + chainback) over every frame of the shard
        drain()       order everything enqueued so far before the barrier (pipelined decoders join their streams)
        stats()       dict: 'units_per_pass' (what `value` counts, e.g. coded symbols), 'bit_errors' (or -1), and
                      anything else the caller wants on the JSON line (taken from rank 0)
    Frames are independent, so there is NO data-path collective: torch.distributed only supplies the barrier, the
    MAX over ranks of the elapsed time and the SUM of the counters.  Returns, on rank 0, the core of the
    bench line {value (units/s over ALL ranks), n_gpus, steps, warmup, ms_per_step, scaling, bit_errors, stats};
    None on the other ranks."""
    import time

    frame_lo, frame_hi = weak_range(frames_per_rank, rank)
    shard = make_shard(frame_lo, frame_hi - frame_lo)
    for i in range(warmup):
        shard.one_pass(i)
    shard.drain()
    barrier(device)
    t0 = time.perf_counter()
    for i in range(steps):
        shard.one_pass(warmup + i)
    shard.drain()
    barrier(device)
    elapsed = max_over_ranks(time.perf_counter() - t0, reduce_device)
    st = shard.stats()
    units = sum_over_ranks(int(st["units_per_pass"]), reduce_device)  # whole job: every rank's frames
    nerr = int(st.get("bit_errors", -1))
    nerr_all = sum_over_ranks(max(nerr, 0), reduce_device)
    any_unknown = sum_over_ranks(1 if nerr < 0 else 0, reduce_device)
    if rank != 0:
        return None
    return {
        "value": units * steps / elapsed,
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "bit_errors": -1 if any_unknown else nerr_all,
        "units_per_pass_all_ranks": units,
        "elapsed_s": elapsed,
        "stats": st,
    }
