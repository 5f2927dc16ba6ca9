"""N>1 path on CPU: two gloo ranks take their frame shards, decode them (with the CPU oracle standing in for the
device, as tests may), and the union equals the single-process result; timing reduces with MAX, counters with SUM."""
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
