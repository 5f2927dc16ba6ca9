"""N>1 path on CPU: two gloo ranks take their frame shards, decode them (with the CPU oracle standing in for the
device, as tests may), and the union equals the single-process result; timing reduces with MAX, counters with SUM.
test_bench_job_loop_two_ranks drives the very function bench.py runs (sharding.run_sharded_job: shard ranges, warm-up,
barrier, timed passes, MAX / SUM reductions, the core of the JSON line) with a CPU shard in place of the HIP one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from ka9q_viterbi_comparison_amd.sharding import shard_range, weak_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_exactly():
    for total in (0, 1, 7, 64, 65536, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [weak_range(10, r) for r in range(3)] == [(0, 10), (10, 20), (20, 30)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from common import frames, spec_of
    from ka9q_viterbi_comparison_amd import codes as C
    from ka9q_viterbi_comparison_amd.sharding import barrier, max_over_ranks, shard_range, sum_over_ranks
    from oracle_lib import OracleDecoder

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code = C.KA9Q27
    spec = spec_of(code)
    B = 8
    steps = B * 8 + spec.K - 1
    lo, hi = shard_range(total, rank, world)
    payload, syms = frames(code, 0x5EED, hi - lo, B, spec.ebn0_db, frame0=lo)
    out = np.zeros((hi - lo, B), np.uint8)
    barrier()
    for i in range(hi - lo):
        o = OracleDecoder(code, spec.poly, steps)
        o.update(syms[i], steps)
        out[i], _ = o.chainback(B * 8)
        o.close()
    barrier()
    t = max_over_ranks(1.0 + rank)
    n = sum_over_ranks(hi - lo)
    np.save(os.path.join(out_dir, f"out{rank}.npy"), out)
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([lo, hi, t, n], dtype=np.float64))
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    from common import frames, spec_of
    from ka9q_viterbi_comparison_amd import codes as C
    from oracle_lib import OracleDecoder

    world, total = 2, 37
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"out{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    assert metas[0][0] == 0 and metas[0][1] == metas[1][0] and metas[1][1] == total
    assert all(m[2] == 2.0 for m in metas)      # MAX over ranks of (1+rank)
    assert all(m[3] == total for m in metas)    # SUM of shard sizes
    spec = spec_of(C.KA9Q27)
    B = 8
    steps = B * 8 + spec.K - 1
    payload, syms = frames(C.KA9Q27, 0x5EED, total, B, spec.ebn0_db)
    ref = np.zeros((total, B), np.uint8)
    for i in range(total):
        o = OracleDecoder(C.KA9Q27, spec.poly, steps)
        o.update(syms[i], steps)
        ref[i], _ = o.chainback(B * 8)
        o.close()
    assert np.array_equal(np.concatenate(outs), ref)


class _CpuShard:
    """What bench.py's HipShard is to run_sharded_job, with the CPU oracle as the decoder (test stand-in only)."""

    def __init__(self, frame_lo, nframes, rank, out_dir):
        from common import frames, spec_of
        from ka9q_viterbi_comparison_amd import codes as C

        self.code, self.spec = C.KA9Q27, spec_of(C.KA9Q27)
        self.B = 8
        self.steps = self.B * 8 + self.spec.K - 1
        self.frame_lo, self.nframes, self.rank, self.out_dir = frame_lo, nframes, rank, out_dir
        self.payload, self.syms = frames(self.code, 0x5EED, nframes, self.B, None, frame0=frame_lo)  # hard symbols: error free
        self.out = np.zeros((nframes, self.B), np.uint8)
        self.passes, self.drains = 0, 0

    def one_pass(self, i):
        import time

        from oracle_lib import OracleDecoder

        for f in range(self.nframes):
            o = OracleDecoder(self.code, self.spec.poly, self.steps)
            o.update(self.syms[f], self.steps)
            self.out[f], _ = o.chainback(self.B * 8)
            o.close()
        if self.rank == 1:
            time.sleep(0.05)  # the slow rank sets the job's elapsed time
        self.passes += 1

    def drain(self):
        self.drains += 1

    def stats(self):
        from common import bit_errors

        np.save(os.path.join(self.out_dir, f"job_out{self.rank}.npy"), self.out)
        return {"units_per_pass": self.nframes * self.steps * self.spec.R, "bit_errors": bit_errors(self.out, self.payload),
                "passes": self.passes, "drains": self.drains, "frame_lo": self.frame_lo}


def _job_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import time

    import torch.distributed as dist

    from ka9q_viterbi_comparison_amd.sharding import run_sharded_job

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.perf_counter()
    core = run_sharded_job(lambda lo, n: _CpuShard(lo, n, rank, out_dir), frames_per_rank=5, steps=3, warmup=1, rank=rank, world=world)
    wall = time.perf_counter() - t0
    if rank == 0:
        core["wall_s"] = wall
        json.dump(core, open(os.path.join(out_dir, "core.json"), "w"))
    else:
        assert core is None
    dist.destroy_process_group()


def test_bench_job_loop_two_ranks(tmp_path):
    import json

    from common import frames, spec_of
    from ka9q_viterbi_comparison_amd import codes as C

    world = 2
    mp.spawn(_job_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    core = json.load(open(tmp_path / "core.json"))
    spec = spec_of(C.KA9Q27)
    steps = 8 * 8 + spec.K - 1
    assert core["n_gpus"] == 2 and core["steps"] == 3 and core["warmup"] == 1 and core["scaling"] == "weak"
    assert core["units_per_pass_all_ranks"] == 2 * 5 * steps * spec.R  # SUM over ranks: whole-job units
    assert core["bit_errors"] == 0
    assert core["stats"]["passes"] == 4 and core["stats"]["drains"] == 2 and core["stats"]["frame_lo"] == 0
    # elapsed is the MAX over ranks (rank 1 sleeps 50 ms per pass) and value = units * steps / elapsed
    assert core["elapsed_s"] >= 3 * 0.05 and core["elapsed_s"] <= core["wall_s"]
    assert abs(core["value"] - core["units_per_pass_all_ranks"] * 3 / core["elapsed_s"]) < 1e-6 * core["value"]
    assert abs(core["ms_per_step"] - core["elapsed_s"] / 3 * 1e3) < 1e-9
    # weak scaling: globally unique frame ids -> rank 1 decoded frames 5..9 of the same generator
    payload, _ = frames(C.KA9Q27, 0x5EED, 10, 8, None)
    assert np.array_equal(np.load(tmp_path / "job_out0.npy"), payload[:5])
    assert np.array_equal(np.load(tmp_path / "job_out1.npy"), payload[5:])


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py as the driver launches it for N = 2 (python -m torch.distributed.run --nproc-per-node 2 ... --gpus 2), on the
    one GPU of the test box: VITERBI_BENCH_BACKEND=gloo lets both ranks share device 0 (RCCL refuses two ranks on one
    device), everything else is the N-GPU code path.  The launcher is a fresh child process (nothing in it has touched the
    GPU before torch.distributed.run starts the ranks).  Checked: one JSON line from rank 0 with n_gpus == 2 and twice the
    single-rank work, and the frame ranges of the two ranks are distinct -- the job's bit-error count equals the sum of the
    counts of frame ranges [0, F) and [F, 2F) decoded here one after the other, which differ from each other.  No data-path
    collective exists to check for: the only torch.distributed calls are barrier / MAX / SUM in sharding.py.
    An 8-GPU RCCL run is NOT covered by this (no such node was available to the builder)."""
    import argparse
    import json
    import subprocess

    import torch

    import bench
    from ka9q_viterbi_comparison_amd import codes as C

    F, bits, ebn0 = 4096, 512, 2.0
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--code", "27", "--frames", str(F), "--payload-bits", str(bits), "--ebn0", str(ebn0), "--no-cpu-baseline", "--no-extra-configs"]
    env = dict(os.environ, VITERBI_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_gpu"] == F
    assert "no collective" in d["config"]["parallelism"]
    assert d["payload_bits_total"] == 2 * F * bits
    # the same two frame ranges, decoded here by the same shard object
    spec = C.CODES["27"]
    args = argparse.Namespace(code="27", payload_bits=bits, ebn0=ebn0, hard=False, variant=0, chunk_frames=None, hbm_budget_gb=200.0,
                              pipeline_depth=2)
    errs = []
    for rank in range(2):
        shard = bench.HipShard(args, spec, torch.device("cuda", 0), rank * F, F)
        shard.drain()
        shard.one_pass(0)
        shard.one_pass(1)
        shard.drain()
        errs.append(shard.stats()["bit_errors"])
        shard.close()
    assert errs[0] != errs[1] and min(errs) > 0, errs
    assert d["bit_errors"] == errs[0] + errs[1], (d["bit_errors"], errs)
