"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties:
encode -> (noise-free) channel -> decode round trip with zero bit errors, agreement between independent kernel
families on the whole batch (a checksum over every decoded byte), and exact agreement with the CPU oracle on
sampled frames.  The oracle cannot decode 65536 frames in seconds; the properties can be checked on all of them."""
import zlib

import numpy as np
import pytest
import torch

from common import spec_of
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, count_bit_errors_dev, gen_frames_dev, noise_q12
from ka9q_viterbi_comparison_amd import VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED, VARIANT_LDS, VARIANT_REGS
from oracle_lib import OracleDecoder

pytestmark = pytest.mark.gpu


def decode_batch(name, frames, payload_bits, hard, variant=0, seed=0x5EED, cb_bits=None):
    spec = C.CODES[name]
    B = payload_bits // 8
    steps = payload_bits + spec.K - 1
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    d_payload = torch.empty(frames * B, dtype=torch.uint8, device=dev)
    d_syms = torch.empty(frames * steps * spec.R, dtype=torch.uint8, device=dev)
    if hard:
        gen_frames_dev(spec, seed, 0, frames, B, C.HARD_AMP_Q16, 0, d_payload, d_syms, stream)
    else:
        gen_frames_dev(spec, seed, 0, frames, B, C.SOFT_AMP_Q16, noise_q12(spec.R, C.SOFT_AMP, spec.ebn0_db), d_payload, d_syms, stream)
    cb_bits = cb_bits or payload_bits
    d_out = torch.zeros(frames * ((cb_bits + 7) // 8), dtype=torch.uint8, device=dev)
    dec = HipViterbi(name, steps, nframes=frames, variant=variant, stream=stream)
    dec.reset()
    dec.update(d_syms, nbits=steps)
    dec.chainback(cb_bits, out=d_out)
    torch.cuda.synchronize()
    dec.close()
    return spec, d_payload, d_syms, d_out


def oracle_frame(spec, syms, steps, nbits):
    o = OracleDecoder(spec.code, spec.poly, steps)
    o.update(syms, steps)
    data, _ = o.chainback(nbits)
    o.close()
    return data


def test_config2_k7_65536_frames_roundtrip_and_cross_kernel():
    """configs[1]: K=7 r=1/2, 65536 frames x 2048 bits."""
    frames, bits = 65536, 2048
    spec, d_payload, d_syms, d_out = decode_batch("27", frames, bits, hard=True)
    assert count_bit_errors_dev(d_out, d_payload, frames * bits // 8) == 0
    # AWGN: the register kernels (L=1 and L=2) and the LDS kernel agree on every decoded byte of the batch
    sums = []
    for variant in (VARIANT_REGS | (1 << 8), VARIANT_REGS | (2 << 8), VARIANT_LDS):
        spec, d_payload, d_syms, d_out = decode_batch("27", frames, bits, hard=False, variant=variant)
        sums.append(zlib.crc32(d_out.cpu().numpy().tobytes()))
    assert len(set(sums)) == 1
    # ... and sampled frames equal the CPU oracle exactly
    steps = bits + spec.K - 1
    out = d_out.cpu().numpy().reshape(frames, bits // 8)
    syms = d_syms.cpu().numpy().reshape(frames, steps * spec.R)
    for f in (0, 1, 4095, 32767, 40000, 65535):
        assert np.array_equal(out[f], oracle_frame(spec, syms[f], steps, bits)), f
    errs = count_bit_errors_dev(d_out, d_payload, frames * bits // 8)
    assert 0 < errs < frames * bits * 1e-3  # Eb/N0 = 4 dB: a few errors per million bits, not a broken decoder


def test_config3_k15_4096_frames():
    """configs[2]: K=15 r=1/6, 4096 frames x 2048 bits (17 GB of decision history resident in HBM)."""
    frames, bits = 4096, 2048
    spec, d_payload, d_syms, d_out = decode_batch("615", frames, bits, hard=True)
    assert count_bit_errors_dev(d_out, d_payload, frames * bits // 8) == 0
    spec, d_payload, d_syms, d_out = decode_batch("615", frames, bits, hard=False)
    steps = bits + spec.K - 1
    out = d_out.cpu().numpy().reshape(frames, bits // 8)
    for f in (0, 2047, 4095):
        s = d_syms[f * steps * spec.R:(f + 1) * steps * spec.R].cpu().numpy()
        assert np.array_equal(out[f], oracle_frame(spec, s, steps, bits)), f


@pytest.mark.parametrize("name,frames", [("27", 65536), ("47", 65536), ("29", 32768), ("49", 32768), ("615", 4096), ("spiral615", 512)])
def test_whole_batch_equals_the_reference(name, frames):
    """configs[1] / configs[2] and the sweep codes at their bench sizes, EVERY frame: the batch the bench decodes (AWGN symbols
    generated on the device, the handle's own kernel selection, i.e. the production kernels) against the reference's own
    decoders (oracle/_ref -- the reference's translation units compiled by oracle/Makefile; the plain-C restatement where that
    library is absent) run over all frames on the host cores with the harness's call sequence.  All decoded bytes must be equal:
    65536 x 256 bytes for configs[1], 4096 x 256 for configs[2] (K=15: the 32-bit-word chainback)."""
    import os

    import oracle_lib as ol

    bits = 2048
    spec, d_payload, d_syms, d_out = decode_batch(name, frames, bits, hard=False)
    steps = bits + spec.K - 1
    syms = d_syms.cpu().numpy().reshape(frames, steps * spec.R)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    want, kind = ol.decode_batch_cpu(spec.code, spec.poly, syms, steps, bits, threads=max(1, min(16, cores)))
    got = d_out.cpu().numpy().reshape(frames, bits // 8)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {frames} frames differ from the {kind} decoder, first {bad[:5]}"
    errs = count_bit_errors_dev(d_out, d_payload, frames * bits // 8)
    assert errs < frames * bits * 2e-2  # a working decoder at the operating point (spiral615's 8-bit metrics: ~1e-2)


@pytest.mark.parametrize("bits", [2048, 16384])
def test_config4_k24_long_frame(bits):
    """configs[3]: K=24 single long frame.  Round trip through the correct chainback convention (nbits+K-1,
    SURVEY.md §0.4), and the per-step and fused kernel families agree on every decoded byte (AWGN input)."""
    spec = C.CODES["224"]
    steps = bits + spec.K - 1
    spec, d_payload, d_syms, d_out = decode_batch("224", 1, bits, hard=True, cb_bits=steps)
    assert count_bit_errors_dev(d_out, d_payload, bits // 8) == 0
    outs = []
    for variant in (VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED):
        spec, d_payload, d_syms, d_out = decode_batch("224", 1, bits, hard=False, variant=variant, cb_bits=steps)
        outs.append(d_out.cpu().numpy().copy())
        assert count_bit_errors_dev(d_out, d_payload, bits // 8) == 0  # K=24 at 4 dB: error free
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_config4_k24_full_frame_against_the_reference():
    """configs[3] at full size, held to the REFERENCE itself (oracle/_ref, the reference's own translation unit compiled by
    oracle/Makefile; the plain-C restatement where that library is absent): one 2048-bit K=24 frame at 4 dB -- 2071 trellis
    steps, about four renormalisations -- through the production kernel (HBM_TILED, what AUTO selects).  Compared: all 8 388 608
    final path metrics, every decoded byte of both chainback conventions (nbits as the harness calls it, nbits + K - 1 which
    decodes correctly: SURVEY.md §0.4), and decision rows sampled across the frame (first / last row, both sides of a
    23-row period boundary, rows around the renormalisations)."""
    import oracle_lib as ol
    from common import frames as gen

    spec = C.CODES["224"]
    bits = 2048
    B, steps = bits // 8, bits + spec.K - 1
    payload, syms = gen(spec.code, 0xC4, 1, B, spec.ebn0_db)
    ref = ol.RefDecoder(spec.code, spec.poly, steps) if ol.have_ref() else OracleDecoder(spec.code, spec.poly, steps)
    ref.init(0)
    ref.update(syms[0], steps)
    dec = HipViterbi("224", steps, nframes=1)
    assert dec.variant == VARIANT_HBM_TILED
    dec.reset()
    dec.update(syms, nbits=steps)
    assert np.array_equal(dec.metrics(0), ref.metrics()), "final path metrics"
    for nb in (bits, steps, bits - 5):
        got, _ = dec.chainback(nb)
        want, _ = ref.chainback(nb)
        assert np.array_equal(got[0], want), f"decoded bytes, nbits={nb}"
    assert np.array_equal(dec.chainback(steps)[0][0][:B], payload[0])
    rows = ref.rows(steps)
    for r in (0, 1, 8, 9, 22, 23, 24, 500, 1034, 1035, 1500, 2047, steps - 2, steps - 1):
        assert np.array_equal(dec.decision_rows(0, r, 1)[0], rows[r]), f"decision row {r}"
    dec.close()
    ref.close()


@pytest.mark.parametrize("name,frames", [("47", 65536), ("29", 32768), ("49", 32768)])
def test_sweep_codes_roundtrip(name, frames):
    bits = 2048
    spec, d_payload, d_syms, d_out = decode_batch(name, frames, bits, hard=True)
    assert count_bit_errors_dev(d_out, d_payload, frames * bits // 8) == 0
    sums = []
    for variant in (VARIANT_REGS, VARIANT_LDS):
        spec, d_payload, d_syms, d_out = decode_batch(name, frames, bits, hard=False, variant=variant)
        sums.append(zlib.crc32(d_out.cpu().numpy().tobytes()))
    assert sums[0] == sums[1]


@pytest.mark.parametrize("name", ["27", "47", "29", "49", "spiral27", "spiral29", "615", "spiral615"])
def test_pure_noise_cross_kernel_stress(name):
    """Uniform random symbols (no codeword underneath) drive the path metrics to their widest spread -- the regime of
    wrap-around / saturation / tie corner cases.  Large batches, every kernel family: decoded bytes and final path
    metrics must agree between the independent implementations (the LDS kernel is oracle-checked frame by frame in
    test_hip_parity.py), and sampled frames must equal the oracle."""
    spec = C.CODES[name]
    frames = 4096 if spec.K < 15 else 64
    bits = 1024 if spec.K < 15 else 256
    steps = bits + spec.K - 1
    steps -= steps % 2
    g = torch.Generator(device="cuda").manual_seed(1234 + spec.K * 10 + spec.R)
    d_syms = torch.randint(0, 256, (frames * steps * spec.R,), dtype=torch.uint8, device="cuda", generator=g)
    # a second batch that mixes extremes only: 0 / 255 / 127 / 128
    pal = torch.tensor([0, 255, 127, 128], dtype=torch.uint8, device="cuda")
    d_syms2 = pal[torch.randint(0, 4, (frames * steps * spec.R,), device="cuda", generator=g)]
    variants = [VARIANT_LDS, VARIANT_REGS] if spec.K == 15 else [VARIANT_LDS] + [VARIANT_REGS | ((lb + 1) << 8) for lb in (0, 1, 2)]
    for syms in (d_syms, d_syms2):
        results = []
        for variant in variants:
            dec = HipViterbi(name, steps, nframes=frames, variant=variant, stream=torch.cuda.current_stream().cuda_stream)
            out = torch.zeros(frames * (bits // 8), dtype=torch.uint8, device="cuda")
            dec.reset()
            dec.update(syms, nbits=steps)
            dec.chainback(bits, out=out)
            torch.cuda.synchronize()
            mets = np.stack([dec.metrics(f) for f in (0, 1, frames // 2, frames - 1)])
            results.append((out.cpu().numpy().copy(), mets))
            dec.close()
        for o, m in results[1:]:
            assert np.array_equal(o, results[0][0]), "decoded bytes differ between kernel families"
            assert np.array_equal(m, results[0][1]), "path metrics differ between kernel families"
        host = syms.cpu().numpy().reshape(frames, steps * spec.R)
        out0 = results[0][0].reshape(frames, bits // 8)
        for f in (0, frames - 1):
            assert np.array_equal(out0[f], oracle_frame(spec, host[f], steps, bits))


def test_config5_shard_size_k7_131072_frames():
    """configs[4] per-GPU shard: 1M frames / 8 GPUs = 131072 frames of K=7 r=1/2 x 2048 bits on one handle -- the first
    batch whose decision history (2.15 GB) has byte offsets beyond 2^31.  Round trip on every frame (noise-free: zero
    bit errors), the L=1 and L=2 register kernels agree on every decoded byte of the AWGN batch (CRC), and frames sampled
    from both ends and from beyond the 2^31-byte mark equal the CPU oracle exactly (sampled, not the whole batch)."""
    frames, bits = 131072, 2048
    spec, d_payload, d_syms, d_out = decode_batch("27", frames, bits, hard=True, seed=0x5EED)
    assert count_bit_errors_dev(d_out, d_payload, frames * bits // 8) == 0
    sums = []
    for variant in (VARIANT_REGS | (2 << 8), VARIANT_REGS | (1 << 8)):
        spec, d_payload, d_syms, d_out = decode_batch("27", frames, bits, hard=False, variant=variant)
        sums.append(zlib.crc32(d_out.cpu().numpy().tobytes()))
    assert sums[0] == sums[1]
    steps = bits + spec.K - 1
    out = d_out.cpu().numpy().reshape(frames, bits // 8)
    for f in (0, 65535, 65536, 130689, 130690, 130700, 131071):
        s = d_syms[f * steps * spec.R:(f + 1) * steps * spec.R].cpu().numpy()
        assert np.array_equal(out[f], oracle_frame(spec, s, steps, bits)), f
    # every frame against the reference's own decoder (one handle, 131072 frames, history offsets beyond 2^31 included)
    import os

    import oracle_lib as ol

    want, kind = ol.decode_batch_cpu(spec.code, spec.poly, d_syms.cpu().numpy().reshape(frames, steps * spec.R), steps, bits,
                                     threads=max(1, min(16, len(os.sched_getaffinity(0)))))
    bad = np.nonzero((out != want).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {frames} frames differ from the {kind} decoder, first {bad[:5]}"
    errs = count_bit_errors_dev(d_out, d_payload, frames * bits // 8)
    assert 0 < errs < frames * bits * 1e-3


@pytest.mark.parametrize("name,frames,chunk,bits", [("615", 256, 64, 256), ("27", 4096, 1024, 512), ("49", 512, 128, 256)])
def test_bench_chunked_shard(name, frames, chunk, bits):
    """bench.py's chunked path (nchunks > 1: batches whose decision history does not fit are decoded in chunks of frames
    that reuse one double-buffered handle): the very HipShard object bench.py drives.  Two passes agree with each other
    byte for byte, the decoded batch equals an unchunked single-handle decode, sampled frames equal the CPU oracle, and
    the live event timing saw every launch."""
    import argparse

    import bench

    spec = C.CODES[name]
    args = argparse.Namespace(code=name, payload_bits=bits, ebn0=None, hard=False, variant=0, chunk_frames=chunk,
                              hbm_budget_gb=200.0, pipeline_depth=2)
    dev = torch.device("cuda", 0)
    frame0 = 777
    shard = bench.HipShard(args, spec, dev, frame0, frames)
    assert shard.nchunks == frames // chunk and shard.dec.pipeline_depth == 2
    shard.drain()
    for i in range(3):
        shard.one_pass(i)
    shard.drain()
    st = shard.stats()  # asserts that the two output buffers (alternate passes) are identical
    assert st["timed_update_launches"] == 3 * shard.nchunks and st["update_ms"] > 0 and st["chainback_ms"] > 0
    steps = bits + spec.K - 1
    got = shard.d_out[0].cpu().numpy().reshape(frames, bits // 8)
    # unchunked, unpipelined decode of the same symbols
    one = HipViterbi(name, steps, nframes=frames, stream=torch.cuda.current_stream().cuda_stream)
    d_out = torch.zeros(frames * (bits // 8), dtype=torch.uint8, device=dev)
    one.reset()
    one.update(shard.d_syms, nbits=steps)
    one.chainback(bits, out=d_out)
    torch.cuda.synchronize()
    one.close()
    assert np.array_equal(got, d_out.cpu().numpy().reshape(frames, bits // 8))
    syms = shard.d_syms.cpu().numpy().reshape(frames, steps * spec.R)
    for f in (0, chunk - 1, chunk, frames - 1):
        assert np.array_equal(got[f], oracle_frame(spec, syms[f], steps, bits)), f
    assert st["bit_errors"] == count_bit_errors_dev(d_out, shard.d_payload, frames * bits // 8)
    shard.close()


@pytest.mark.parametrize("name", ["27", "47", "29", "49", "spiral27", "spiral29"])
def test_one_very_long_frame(name):
    """One frame of 600 000 bits through the handle's own choice for a single frame (the wave-per-frame kernels): hundreds of
    branch-metric table chunks, byte offsets past 2^22 in a frame's symbols, and -- spiral arithmetic -- the lazy offset of the
    saturating metrics folded back into the registers dozens of times (acs_wave.hip fold_offset).  Every decoded byte, the
    final metrics and a sample of decision rows against the oracle; then the same bytes from the register kernel."""
    from common import frames

    spec = C.CODES[name]
    B = 75000
    steps = B * 8 + spec.K - 1
    steps -= steps % 2 if spec.family.startswith("spiral") else 0
    _, syms = frames(spec.code, 4242, 1, B, spec.ebn0_db)
    syms = np.ascontiguousarray(syms[:, :steps * spec.R])
    o = OracleDecoder(spec.code, spec.poly, steps)
    o.update(syms[0], steps)
    ref_bytes = o.chainback(B * 8)[0]
    ref_metrics = o.metrics()
    ref_rows = o.rows(steps)
    o.close()
    got = None
    for variant in (0, VARIANT_REGS):
        dec = HipViterbi(name, steps, nframes=1, variant=variant)
        dec.reset()
        dec.update(syms, nbits=steps)
        data, _ = dec.chainback(B * 8)
        assert np.array_equal(data[0], ref_bytes), (name, variant)
        assert np.array_equal(dec.metrics(0), ref_metrics), (name, variant)
        for row0 in (0, 4999, steps // 2, steps - 700):
            assert np.array_equal(dec.decision_rows(0, row0, 600), ref_rows[row0:row0 + 600]), (name, variant, row0)
        dec.close()
