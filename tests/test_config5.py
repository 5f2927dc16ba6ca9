"""BASELINE.json configs[4] -- the full sweep {27, 29, 47, 49, 615, 224} x 1M frames sharded over 8 GPUs -- at its per-GPU
size on ONE GPU (131072 frames per code; src/main.cpp:363-419 is the reference's block list).  No 8-GPU node is available
to the builder, so what one rank does is what can be checked: every frame of a noise-free batch decodes without error,
frames sampled from an AWGN batch -- from both ends, from the middle and from beyond the 2^31-byte offset of the
decision history -- equal the CPU oracle byte for byte, and for the K <= 9 codes every frame of that batch equals the
reference's own decoder (oracle/_ref) run over all 131072 frames on the host cores.  K=15 goes through the double-buffered handle in HBM-sized chunks
exactly as bench.HipShard does (131072 frames x 4.2 MB of decisions do not fit); K=24 runs the reduced share a rank
would get (stated below), compared with the oracle in full."""
import argparse

import numpy as np
import pytest
import torch

from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, count_bit_errors_dev
from oracle_lib import OracleDecoder

pytestmark = pytest.mark.gpu

FRAMES = 131072
BITS = 2048


def oracle_bytes(spec, syms, steps, nbits):
    o = OracleDecoder(spec.code, spec.poly, steps)
    o.update(syms, steps)
    data, _ = o.chainback(nbits)
    o.close()
    return data


def shard_args(name, hard):
    return argparse.Namespace(code=name, payload_bits=BITS, ebn0=None, hard=hard, variant=0, chunk_frames=None,
                              hbm_budget_gb=200.0, pipeline_depth=2)


@pytest.mark.parametrize("name", ["47", "29", "49", "615"])  # 27: test_full_size.py::test_config5_shard_size_k7_131072_frames
def test_config5_shard_per_code(name):
    import bench

    spec = C.CODES[name]
    steps = BITS + spec.K - 1
    B = BITS // 8
    dev = torch.device("cuda", 0)
    # (a) noise-free: zero bit errors on all 131072 frames (bench.HipShard: K=7 in chunks of 65536, K=15 of 16384, K=9 whole)
    shard = bench.HipShard(shard_args(name, True), spec, dev, 0, FRAMES)
    assert shard.frames == FRAMES and shard.chunk * shard.nchunks == FRAMES
    if spec.K == 15:
        assert shard.nchunks > 1 and shard.dec.pipeline_depth == 2
    shard.drain()
    shard.one_pass(0)
    shard.drain()
    shard.dec.sync()
    assert count_bit_errors_dev(shard.d_out[0], shard.d_payload, FRAMES * B, shard.stream.cuda_stream) == 0
    shard.close()
    del shard
    torch.cuda.empty_cache()
    # (b) AWGN: two passes agree (alternate output buffers), sampled frames equal the oracle
    shard = bench.HipShard(shard_args(name, False), spec, dev, 0, FRAMES)
    shard.drain()
    for i in range(2):
        shard.one_pass(i)
    shard.drain()
    shard.dec.sync()
    assert torch.equal(shard.d_out[0], shard.d_out[1])
    # decision-history bytes per frame in the handle: the first frame whose rows start beyond 2^31 bytes
    per_frame = (steps + spec.K - 1) * ((1 << (spec.K - 1)) // 8)
    beyond = min(shard.chunk - 1, (1 << 31) // per_frame + 1)
    sample = sorted({0, 1, beyond, min(beyond + 70, shard.chunk - 1), shard.chunk - 1, shard.chunk % FRAMES, FRAMES // 2 + 3, FRAMES - 1})
    if spec.K == 15:
        sample = [0, shard.chunk - 1, shard.chunk, FRAMES - 1]  # 6 ms of oracle per frame, 4.2 MB rows: every chunk boundary kind
    got = shard.d_out[0].view(FRAMES, B)
    sy = shard.d_syms.view(FRAMES, steps * spec.R)
    for f in sample:
        ref = oracle_bytes(spec, sy[f].cpu().numpy(), steps, BITS)
        assert np.array_equal(got[f].cpu().numpy(), ref), (name, f)
    if spec.K <= 9:
        # ... and for K <= 9 EVERY one of the 131072 frames against the reference's own decoder on the host cores (a few seconds;
        # K=15 would take a quarter of an hour and stays sampled)
        import os

        import oracle_lib as ol

        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count() or 1
        want, kind = ol.decode_batch_cpu(spec.code, spec.poly, sy.cpu().numpy(), steps, BITS, threads=max(1, min(16, cores)))
        bad = np.nonzero((got.cpu().numpy() != want).any(axis=1))[0]
        assert bad.size == 0, f"{bad.size} of {FRAMES} frames differ from the {kind} decoder, first {bad[:5]}"
    errs = count_bit_errors_dev(shard.d_out[0], shard.d_payload, FRAMES * B, shard.stream.cuda_stream)
    assert errs < FRAMES * BITS * 2e-3, errs  # a working decoder at the operating point, not garbage
    shard.close()


def test_config5_k24_share():
    """K=24's share of configs[4]: 1M frames x 87 steps x 33 MiB is ~3 PB of traffic, so a rank runs a stated, reduced
    count -- 12 frames of the reference's own size (64 info bits, src/main.cpp:411-418), three decodes in flight on the
    handle -- and every frame is compared with the oracle: rows sampled, final metrics, bytes of both chainback conventions."""
    from common import frames as gen

    spec = C.CODES["224"]
    nframes, B = 12, 8
    steps = B * 8 + spec.K - 1
    payload, syms = gen(spec.code, 0x224, nframes, B, spec.ebn0_db)
    dec = HipViterbi("224", steps, nframes=nframes)
    dec.reset()
    dec.update(syms, nbits=steps)
    data, _ = dec.chainback(B * 8)
    tail, _ = dec.chainback(steps)
    for f in range(nframes):
        o = OracleDecoder(spec.code, spec.poly, steps)
        o.update(syms[f], steps)
        assert np.array_equal(data[f], o.chainback(B * 8)[0]), f
        assert np.array_equal(tail[f], o.chainback(steps)[0]), f
        if f in (0, 5, 11):
            assert np.array_equal(dec.metrics(f), o.metrics()), f
            rows = o.rows(steps)
            for r in (0, 22, 23, 45, steps - 1):
                assert np.array_equal(dec.decision_rows(f, r, 1)[0], rows[r]), (f, r)
        o.close()
        assert np.array_equal(tail[f][:B], payload[f])
    dec.close()
