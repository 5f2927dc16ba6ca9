"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar: bit-exact decision rows, path metrics, decoded bytes and return codes (integer work).
"""
import numpy as np
import pytest

from common import GPU_CODES, bit_errors, frames, spec_of
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C
from ka9q_viterbi_comparison_amd import VARIANT_AUTO, VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED, VARIANT_LDS, VARIANT_REGS, VARIANT_WAVE
from oracle_lib import OracleDecoder

pytestmark = pytest.mark.gpu


def regs(lb):
    """REGS variant with 2^lb lanes per frame (bits 8.. of the variant word carry 1+lb)."""
    return VARIANT_REGS | ((lb + 1) << 8)


def variants_for(code):
    if code == C.KA9Q224:
        return [VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED]
    if code == C.KA9Q615:
        return [VARIANT_LDS, VARIANT_REGS]
    if code == C.SPIRAL615:
        return [VARIANT_LDS, VARIANT_REGS]
    if code in (C.KA9Q27, C.SPIRAL47, C.SPIRAL27):
        return [VARIANT_LDS, regs(0), regs(1), regs(2), VARIANT_WAVE]
    return [VARIANT_LDS, regs(0), regs(1), regs(2), VARIANT_WAVE]


def oracle_decode(code, syms, steps, nbits, endstate=0, start=0, splits=None):
    spec = spec_of(code)
    o = OracleDecoder(code, spec.poly, steps)
    o.init(start)
    if splits is None:
        o.update(syms, steps)
    else:
        off = 0
        for n in splits:
            o.update(syms[off * spec.R:], n)
            off += n
    data, rc = o.chainback(nbits, endstate)
    out = dict(rows=o.rows(), metrics=o.metrics(), data=data, rc=rc, renorms=o.renorms)
    o.close()
    return out


CASES = []
for _code in GPU_CODES:
    for _v in variants_for(_code):
        CASES.append((_code, _v))


@pytest.mark.parametrize("code,variant", CASES)
@pytest.mark.parametrize("ebn0", ["hard", "awgn"])
def test_batch_matches_oracle(code, variant, ebn0):
    """A batch of frames: every decision row, final metric, decoded byte and return code equals the oracle's."""
    spec = spec_of(code)
    B = 8 if spec.K == 24 else (24 if spec.K == 15 else 40)
    nframes = 1 if spec.K == 24 else (3 if spec.K == 15 else 70)
    if code == C.SPIRAL615:
        B = 25  # odd step count is dropped by spiral; keep steps even: 25*8+14  # 70: more than one wave of frames, ragged
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 77 + code, nframes, B, None if ebn0 == "hard" else spec.ebn0_db)
    dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
    dec.reset()
    dec.update(syms)
    data, rc = dec.chainback(B * 8)
    for f in range(nframes):
        ref = oracle_decode(code, syms[f], steps, B * 8)
        rows = dec.decision_rows(f, 0, ref["rows"].shape[0])
        assert np.array_equal(rows, ref["rows"]), f"decision rows differ, frame {f}"
        assert np.array_equal(dec.metrics(f), ref["metrics"]), f"metrics differ, frame {f}"
        assert np.array_equal(data[f], ref["data"]), f"decoded bytes differ, frame {f}"
        if code != C.KA9Q224:
            assert bit_errors(data[f], payload[f]) == 0 or ebn0 == "awgn"
    if nframes == 1:
        assert rc == oracle_decode(code, syms[0], steps, B * 8)["rc"]
    dec.close()


@pytest.mark.parametrize("code,variant", [c for c in CASES if c[0] in (C.KA9Q27, C.KA9Q29, C.KA9Q615)])
def test_incremental_update(code, variant):
    """ka9q update is incremental (viterbi27_sse2.cpp:121,174): ragged splits give the same rows as one call."""
    spec = spec_of(code)
    B = 16
    steps = B * 8 + spec.K - 1
    nframes = 5 if spec.K < 15 else 2
    _, syms = frames(code, 5, nframes, B, spec.ebn0_db)
    splits = [1, 7, 30, 2, steps - 40]
    dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
    dec.reset()
    off = 0
    s3 = syms.reshape(nframes, steps, spec.R)
    for n in splits:
        dec.update(np.ascontiguousarray(s3[:, off:off + n, :]), nbits=n)
        off += n
    assert dec.rows_written == steps
    data, _ = dec.chainback(B * 8)
    for f in range(nframes):
        ref = oracle_decode(code, syms[f], steps, B * 8)
        assert np.array_equal(dec.decision_rows(f, 0, steps), ref["rows"])
        assert np.array_equal(dec.metrics(f), ref["metrics"])
        assert np.array_equal(data[f], ref["data"])
    dec.close()


@pytest.mark.parametrize("code,variant", [c for c in CASES if c[0] in (C.SPIRAL47, C.SPIRAL49, C.SPIRAL27, C.SPIRAL29, C.SPIRAL615)])
def test_spiral_restart_and_odd_step(code, variant):
    """spiral update restarts at row 0 on every call and drops an odd last step (spiral47.cpp:536-538)."""
    spec = spec_of(code)
    B = 8
    steps = B * 8 + spec.K - 1  # even for K=7,9
    _, syms = frames(code, 9, 4, B, spec.ebn0_db)
    dec = HipViterbi(spec.name, steps, nframes=4, variant=variant)
    dec.reset()
    s3 = syms.reshape(4, steps, spec.R)
    dec.update(np.ascontiguousarray(s3[:, :11, :]), nbits=11)  # runs 10 steps
    assert dec.rows_written == 10
    dec.update(syms, nbits=steps)  # restarts at row 0, metrics continue
    for f in range(4):
        o = OracleDecoder(code, spec.poly, steps)
        o.update(s3[f, :11].reshape(-1), 11)
        o.update(syms[f], steps)
        assert np.array_equal(dec.decision_rows(f, 0, steps), o.rows(steps))
        assert np.array_equal(dec.metrics(f), o.metrics())
        o.close()
    dec.close()


@pytest.mark.parametrize("code,variant", CASES)
def test_endstate_start_state_and_ragged_bits(code, variant):
    """Non-zero starting state, non-zero end state, bit counts that are not multiples of 8."""
    spec = spec_of(code)
    # K=24 (init_viterbi224_sse2 / chainback_viterbi224_sse2, viterbi224_sse2.cpp:32-47,79-121): a short frame keeps the
    # oracle's 8M-state scalar walk to seconds; the oracle is pinned on these arguments by
    # test_oracle_vs_reference.py::test_k24_start_state_end_state_ragged_bits
    B = 4 if spec.K == 24 else 12
    nframes = 2 if spec.K == 24 else 3
    steps = B * 8 + spec.K - 1
    _, syms = frames(code, 21, nframes, B, spec.ebn0_db)
    N = 1 << (spec.K - 1)
    cases = [(5, 0, B * 8), (N - 1, 3, B * 8 - 3), (0, N + 9, 13)]
    if spec.K == 24:
        cases = [(5, 0, B * 8), (N - 1, 3, steps - 3), (0x2AAAAA, N + 0x155555, steps)]
    for start, end, nbits in cases:
        dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
        dec.reset(start)
        dec.update(syms)
        data, _ = dec.chainback(nbits, endstate=end)
        for f in range(nframes):
            ref = oracle_decode(code, syms[f], steps, nbits, endstate=end, start=start)
            assert np.array_equal(data[f], ref["data"])
            assert np.array_equal(dec.metrics(f), ref["metrics"])
            if spec.K == 24:
                assert np.array_equal(dec.decision_rows(f, 0, steps), ref["rows"])
        dec.close()


@pytest.mark.parametrize("variant", [VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED])
def test_k24_chainback_variants(variant):
    """chainback_viterbi224_sse2 has no tail skip: pin both the harness call (nbits) and the
    correct one (nbits+K-1)."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 8
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 3, 1, B, spec.ebn0_db)
    dec = HipViterbi("224", steps, variant=variant)
    dec.reset()
    dec.update(syms)
    o = OracleDecoder(code, spec.poly, steps)
    o.update(syms[0], steps)
    for nbits in (B * 8, steps, 13):
        d, rc = dec.chainback(nbits)
        r, rrc = o.chainback(nbits)
        assert np.array_equal(d[0], r) and rc == rrc
    d, _ = dec.chainback(steps)
    assert bit_errors(d[0][:B], payload[0]) == 0
    o.close()
    dec.close()


def test_k615_forced_renormalisation_and_return_value():
    """Noisy K=15 frames renormalise several times (viterbi615_sse2.cpp:160-183) and chainback returns the end
    state's path metric (:76,90)."""
    code = C.KA9Q615
    spec = spec_of(code)
    B = 64
    steps = B * 8 + spec.K - 1
    _, syms = frames(code, 1234, 1, B, spec.ebn0_db)
    ref = oracle_decode(code, syms[0], steps, B * 8)
    assert ref["renorms"] >= 2
    dec = HipViterbi("615", steps)
    dec.reset()
    dec.update(syms)
    data, rc = dec.chainback(B * 8)
    assert rc == ref["rc"]
    assert np.array_equal(data[0], ref["data"])
    assert np.array_equal(dec.metrics(0), ref["metrics"])
    dec.close()


@pytest.mark.parametrize("variant", [VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED])
def test_k24_renormalisation_and_incremental(variant):
    """A K=24 frame long enough to renormalise (viterbi224_sse2.cpp:226-246), fed in ragged pieces so that the
    speculative replay and the phase bookkeeping of the fused passes are exercised; two frames per handle."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 56
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 11, 2, B, spec.ebn0_db)
    dec = HipViterbi("224", steps, nframes=2, variant=variant)
    dec.reset()
    s3 = syms.reshape(2, steps, spec.R)
    off = 0
    for n in (3, 30, 100, steps - 133):
        dec.update(np.ascontiguousarray(s3[:, off:off + n, :]), nbits=n)
        off += n
    data, _ = dec.chainback(steps)
    for f in range(2):
        ref = oracle_decode(code, syms[f], steps, steps)
        assert ref["renorms"] >= 1
        assert np.array_equal(dec.metrics(f), ref["metrics"])
        assert np.array_equal(data[f], ref["data"])
        assert bit_errors(data[f][:B], payload[f]) == 0
        for r in (0, 1, 22, 23, 100, steps - 1):
            assert np.array_equal(dec.decision_rows(f, r, 1), ref["rows"][r:r + 1]), f"row {r}"
    dec.close()


def test_k24_renormalisation_at_call_boundaries():
    """The tiled K=24 passes renormalise among themselves: the replay of the pass that raised the flag forms the minimum and
    the next pass subtracts it while it loads (acs_k24t.hip ctl).  The row after which viterbi224_sse2.cpp:226-246
    renormalises is located with the oracle, and update calls are cut so that the renormalisation falls behind the LAST row
    of a call (no pass left in that call: the separate min / subtract kernels run), one row before the end of a call (the
    subtracting pass is a single row) and at the first row of a call."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 56
    steps = B * 8 + spec.K - 1
    _, syms = frames(code, 11, 1, B, spec.ebn0_db)
    s2 = syms.reshape(1, steps, spec.R)
    o = OracleDecoder(code, spec.poly, steps)
    rstar = None
    for r in range(steps):
        o.update(s2[0, r].reshape(-1), 1)
        if o.renorms >= 1:
            rstar = r
            break
    assert rstar is not None and 25 < rstar < steps - 30
    o.update(s2[0, rstar + 1:].reshape(-1), steps - rstar - 1)
    want_metrics, want_rows = o.metrics(), o.rows(steps)
    want_data, _ = o.chainback(steps, 0)
    o.close()
    for variant in (VARIANT_HBM_TILED, VARIANT_HBM_FUSED):
        for first in (rstar + 1, rstar + 2, rstar, rstar - 1, rstar + 1 - 23):
            dec = HipViterbi("224", steps, variant=variant)
            dec.reset()
            dec.update(np.ascontiguousarray(s2[:, :first, :]), nbits=first)
            dec.update(np.ascontiguousarray(s2[:, first:, :]), nbits=steps - first)
            data, _ = dec.chainback(steps)
            assert np.array_equal(dec.metrics(0), want_metrics), f"variant {variant}: metrics, first call {first} rows (renormalisation after row {rstar})"
            for r in (rstar - 1, rstar, rstar + 1, rstar + 2, rstar + 24, steps - 1):
                assert np.array_equal(dec.decision_rows(0, r, 1), want_rows[r:r + 1]), f"variant {variant} first {first} row {r}"
            assert np.array_equal(data[0], want_data)
            dec.close()


def test_k24_many_frames_per_handle():
    """Seven K=24 frames on one handle: with VARIANT_HBM_FUSED each of the three in-flight decode slots takes several
    frames in turn, every one through a renormalisation (viterbi224_sse2.cpp:226-246) and its speculative replay.
    Frames 0 and 6 are checked against the oracle (rows sampled, metrics, bytes); all seven against the per-step
    kernel family (VARIANT_HBM, oracle-checked above) on metrics and decoded bytes."""
    code = C.KA9Q224
    spec = spec_of(code)
    B, nframes = 56, 7
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 31, nframes, B, spec.ebn0_db)
    out = {}
    for variant in (VARIANT_HBM_TILED, VARIANT_HBM_FUSED, VARIANT_HBM):
        dec = HipViterbi("224", steps, nframes=nframes, variant=variant)
        dec.reset()
        dec.update(syms)
        data, _ = dec.chainback(steps)
        out[variant] = (data.copy(), np.stack([dec.metrics(f) for f in range(nframes)]))
        if variant != VARIANT_HBM:
            for f in (0, nframes - 1):
                ref = oracle_decode(code, syms[f], steps, steps)
                assert ref["renorms"] >= 1
                assert np.array_equal(out[variant][1][f], ref["metrics"])
                assert np.array_equal(data[f], ref["data"])
                for r in (0, 22, 23, 200, steps - 1):
                    assert np.array_equal(dec.decision_rows(f, r, 1), ref["rows"][r:r + 1]), f"frame {f} row {r}"
        dec.close()
    for v in (VARIANT_HBM_FUSED, VARIANT_HBM_TILED):
        assert np.array_equal(out[v][0], out[VARIANT_HBM][0])
        assert np.array_equal(out[v][1], out[VARIANT_HBM][1])
    for f in range(nframes):
        assert bit_errors(out[VARIANT_HBM_TILED][0][f][:B], payload[f]) == 0


@pytest.mark.parametrize("name,depth,nframes", [("27", 2, 300), ("27", 3, 70), ("47", 2, 130), ("29", 2, 70), ("615", 2, 3), ("spiral615", 2, 2)])
def test_pipelined_handle_matches_oracle(name, depth, nframes):
    """vhip_set_pipeline_depth: consecutive decodes on one handle rotate through `depth` decision-history buffers and
    internal streams.  Five decodes of five DIFFERENT symbol batches are issued back to back with no host sync in
    between (device-pointer API), each into its own output buffer; after vhip_join every output equals the oracle's for
    ITS batch (sampled frames) and a strictly serial handle's (all frames), and the live timing saw every launch."""
    import torch

    spec = C.CODES[name]
    code = spec.code
    B = 20 if spec.K < 15 else 12
    steps = B * 8 + spec.K - 1
    steps -= 0 if spec_is_incremental(code) else steps % 2
    stream = torch.cuda.current_stream().cuda_stream
    dec = HipViterbi(name, steps, nframes=nframes, stream=stream, pipeline_depth=depth)
    assert dec.pipeline_depth == depth
    dec.enable_timing(True)
    ser = HipViterbi(name, steps, nframes=nframes, stream=stream)
    batches, outs, refs = [], [], []
    for k in range(5):
        _, syms = frames(code, 900 + k, nframes, B, spec.ebn0_db)
        syms = np.ascontiguousarray(syms[:, :steps * spec.R])
        batches.append((syms, torch.from_numpy(syms).cuda()))
        outs.append(torch.zeros(nframes * B, dtype=torch.uint8, device="cuda"))
        refs.append(torch.zeros(nframes * B, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    for k in range(5):
        dec.reset()
        dec.update(batches[k][1], nbits=steps)
        dec.chainback(B * 8, out=outs[k])
    dec.join()
    for k in range(5):
        ser.reset()
        ser.update(batches[k][1], nbits=steps)
        ser.chainback(B * 8, out=refs[k])
    torch.cuda.synchronize()
    su, nu, sc, nc = dec.read_timing()
    assert nu == 5 and nc == 5 and su > 0 and sc > 0
    for k in range(5):
        got = outs[k].cpu().numpy().reshape(nframes, B)
        assert np.array_equal(got, refs[k].cpu().numpy().reshape(nframes, B)), f"decode {k} differs from the serial handle"
        for f in sorted({0, nframes // 2, nframes - 1}):
            assert np.array_equal(got[f], oracle_decode(code, batches[k][0][f], steps, B * 8)["data"]), (k, f)
    # introspection reads the CURRENT buffer set (decode 4)
    ref = oracle_decode(code, batches[4][0][0], steps, B * 8)
    assert np.array_equal(dec.decision_rows(0, 0, ref["rows"].shape[0]), ref["rows"])
    assert np.array_equal(dec.metrics(0), ref["metrics"])
    # host-pointer calls still work on a pipelined handle (they drain it first)
    dec.reset()
    dec.update(batches[1][0], nbits=steps)
    data, _ = dec.chainback(B * 8)
    assert np.array_equal(data, refs[1].cpu().numpy().reshape(nframes, B))
    dec.close()
    ser.close()


@pytest.mark.parametrize("name", ["615", "spiral615"])
def test_first_chainback_of_a_fresh_pipelined_handle(name):
    """Regression for the one intermittent wrong decode of round 2 (commit 7a7caf7): the first chainback of a K >= 15 handle
    with <= 64 frames allocates and clears the scratch words of the segment-parallel walk; the clear was a hipMemset on the
    null stream while the walk ran on a NON-BLOCKING internal stream of the pipelined handle, so the walk could start on
    words that were cleared under its feet.  The order that raced, on purpose: fresh handle, set_pipeline_depth, then init /
    update / chainback through the device-pointer API with no host synchronisation anywhere before the join -- for the first
    decode AND for the decode that first uses the second slot.  Several fresh handles, each compared with the oracle."""
    import torch

    spec = C.CODES[name]
    code = spec.code
    B, nframes = 64, 5  # 512 bits = 8 segments of 64 bits per frame
    steps = B * 8 + spec.K - 1
    steps -= 0 if spec_is_incremental(code) else steps % 2
    stream = torch.cuda.current_stream().cuda_stream
    _, syms = frames(code, 4242, nframes, B, spec.ebn0_db)
    syms = np.ascontiguousarray(syms[:, :steps * spec.R])
    d_syms = torch.from_numpy(syms).cuda()
    refs = [oracle_decode(code, syms[f], steps, B * 8)["data"] for f in range(nframes)]
    torch.cuda.synchronize()
    for trial in range(6):
        dec = HipViterbi(name, steps, nframes=nframes, stream=stream, pipeline_depth=2)
        outs = [torch.zeros(nframes * B, dtype=torch.uint8, device="cuda") for _ in range(3)]
        for k in range(3):  # slot 1, slot 0, slot 1
            dec.reset()
            dec.update(d_syms, nbits=steps)
            dec.chainback(B * 8, out=outs[k])
        dec.join()
        torch.cuda.synchronize()
        rewalked, nseg = dec.chainback_rewalked()
        assert nseg == 8
        for k in range(3):
            got = outs[k].cpu().numpy().reshape(nframes, B)
            for f in range(nframes):
                assert np.array_equal(got[f], refs[f]), (name, trial, k, f)
        dec.close()


def test_pipeline_depth_argument_checks():
    from ka9q_viterbi_comparison_amd._lib import VhipError

    with pytest.raises(VhipError):
        HipViterbi("224", 40, pipeline_depth=2)  # K=24 keeps several frames in flight by itself
    with pytest.raises(VhipError):
        HipViterbi("27", 40, pipeline_depth=4)
    d = HipViterbi("27", 40, nframes=2, pipeline_depth=2)
    assert d.status == 0
    with pytest.raises(VhipError):
        d.update(np.zeros((2, 400 * 2), np.uint8), nbits=400)  # more steps than the handle was created for
    assert d.status == -1
    d.reset()
    assert d.status == 0
    d.close()


def spec_is_incremental(code):
    return code in (C.KA9Q27, C.KA9Q29, C.KA9Q615, C.KA9Q224)


@pytest.mark.parametrize("code,variant", CASES)
def test_adversarial_symbols(code, variant):
    """Saturation / wrap-around corner cases: all-0, all-255, erasure-like mid-scale symbols, alternating extremes,
    ramps and uniform noise -- one pattern per frame of a batch, every row / metric / byte against the oracle."""
    spec = spec_of(code)
    B = 16 if spec.K < 15 else (10 if spec.K == 15 else 3)
    steps = B * 8 + spec.K - 1
    if not spec_is_incremental(code):
        steps -= steps % 2
    n = steps * spec.R
    rng = np.random.default_rng(99)
    pats = np.stack([
        np.zeros(n, np.uint8), np.full(n, 255, np.uint8), np.full(n, 128, np.uint8), np.full(n, 127, np.uint8),
        np.tile(np.array([0, 255], np.uint8), n)[:n], (np.arange(n) % 256).astype(np.uint8),
        rng.integers(0, 256, n, dtype=np.uint8), rng.choice(np.array([0, 255], np.uint8), n),
        rng.choice(np.array([0, 127, 128, 255], np.uint8), n),
    ])
    if spec.K == 24:
        pats = pats[[0, 1, 2, 4, 6]]  # each oracle frame is an 8M-state scalar walk
    dec = HipViterbi(spec.name, steps, nframes=len(pats), variant=variant)
    dec.reset()
    dec.update(pats, nbits=steps)
    nb = (B * 8) - 5
    data, _ = dec.chainback(nb)
    for f in range(len(pats)):
        ref = oracle_decode(code, pats[f], steps, nb)
        assert np.array_equal(dec.decision_rows(f, 0, steps), ref["rows"]), f"rows, pattern {f}"
        assert np.array_equal(dec.metrics(f), ref["metrics"]), f"metrics, pattern {f}"
        assert np.array_equal(data[f], ref["data"]), f"bytes, pattern {f}"
    dec.close()


@pytest.mark.parametrize("code", GPU_CODES)
def test_empty_and_degenerate_calls(code):
    """Zero-length update / chainback are no-ops; chainback before any update reads all-zero rows like the oracle;
    more steps than the handle was created for is an error, not a buffer overrun."""
    from ka9q_viterbi_comparison_amd._lib import VhipError

    spec = spec_of(code)
    steps = 40 + spec.K - 1 if spec.K < 24 else 30
    dec = HipViterbi(spec.name, steps, nframes=2)
    dec.reset()
    dec.update(np.zeros((2, 0), np.uint8), nbits=0)
    assert dec.rows_written == 0
    data, rc = dec.chainback(0)
    assert data.shape == (2, 0)
    data, _ = dec.chainback(16, endstate=5)
    o = OracleDecoder(code, spec.poly, steps)
    ref, _ = o.chainback(16, 5)
    o.close()
    assert np.array_equal(data[0], ref) and np.array_equal(data[1], ref)
    too_many = steps + spec.K + 1
    with pytest.raises(VhipError):
        dec.update(np.zeros((2, too_many * spec.R), np.uint8), nbits=too_many)
    dec.close()


def test_two_handles_two_streams_and_reuse():
    """Distinct handles are independent (SURVEY.md §8b threading): two decoders on two streams, interleaved calls,
    each reused for a second batch after reset(); device-pointer API throughout."""
    import torch

    code = C.KA9Q27
    spec = spec_of(code)
    B, nframes = 24, 200
    steps = B * 8 + spec.K - 1
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d1 = HipViterbi("27", steps, nframes=nframes, stream=s1.cuda_stream)
    d2 = HipViterbi("47", steps, nframes=nframes, stream=s2.cuda_stream)
    for rnd in range(2):
        p1, y1 = frames(C.KA9Q27, 100 + rnd, nframes, B, 4.0)
        p2, y2 = frames(C.SPIRAL47, 200 + rnd, nframes, B, 2.0)
        t1, t2 = torch.from_numpy(y1).cuda(), torch.from_numpy(y2).cuda()
        o1 = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
        o2 = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        d1.reset()
        d2.reset()
        d1.update(t1, nbits=steps)
        d2.update(t2, nbits=steps)
        d2.chainback(B * 8, out=o2)
        d1.chainback(B * 8, out=o1)
        d1.sync()
        d2.sync()
        r1, r2 = o1.cpu().numpy().reshape(nframes, B), o2.cpu().numpy().reshape(nframes, B)
        for f in (0, 63, 64, 199):
            assert np.array_equal(r1[f], oracle_decode(C.KA9Q27, y1[f], steps, B * 8)["data"])
            assert np.array_equal(r2[f], oracle_decode(C.SPIRAL47, y2[f], steps, B * 8)["data"])
    d1.close()
    d2.close()


def _random_poly(spec, rng):
    return [int(rng.integers(0, 1 << (spec.K - 2))) * 2 + 1 + (1 << (spec.K - 1)) for _ in range(spec.R)]


@pytest.mark.parametrize("code", [C.KA9Q27, C.SPIRAL47, C.KA9Q29, C.SPIRAL49, C.SPIRAL27, C.KA9Q615, C.SPIRAL615, C.KA9Q224])
def test_arbitrary_polynomials(code):
    """Polynomials other than the harness set (the reference builds its table from any `poly`, viterbi27_sse2.cpp:57-75).
    Branch tables are per handle, so handles with different polynomials coexist (the reference keeps one process-global
    table, :26-28,61-70).  Three routes, all held to the oracle on every row / metric / byte:
      * default: the FAST kernel, compiled for these polynomials when the handle is created (jit.hip) -- REGS / HBM_TILED;
      * the any-polynomial kernels (acs_lds / acs_k24) when asked for, or when the run-time build is switched off;
      * with VHIP_JIT=0, asking for the fast variant explicitly is an error, not a silent fallback."""
    import os

    from ka9q_viterbi_comparison_amd._lib import VhipError
    from ka9q_viterbi_comparison_amd.decoder import gen_frames_host, noise_q12

    spec = spec_of(code)
    rng = np.random.default_rng(1000 + code)
    B = 8 if spec.K == 24 else 16
    steps = B * 8 + spec.K - 1
    steps -= 0 if spec_is_incremental(code) else steps % 2
    fast = VARIANT_HBM_TILED if spec.K == 24 else VARIANT_REGS
    generic = VARIANT_HBM if spec.K == 24 else VARIANT_LDS
    for trial in range(2):
        poly = _random_poly(spec, rng)
        nframes = 1 if spec.K == 24 else 5
        payload, syms = gen_frames_host(spec, 7 + trial, 0, nframes, B, C.SOFT_AMP_Q16, noise_q12(spec.R, 64.0, spec.ebn0_db), poly=poly)
        syms = np.ascontiguousarray(syms[:, :steps * spec.R])
        refs = []
        for f in range(nframes):
            o = OracleDecoder(code, poly, steps)
            o.update(syms[f], steps)
            refs.append((o.rows(steps), o.metrics(), o.chainback(B * 8)[0]))
            o.close()
        for route in ("default", "fast", "generic"):
            want = {"default": VARIANT_AUTO, "fast": fast, "generic": generic}[route]
            dec = HipViterbi(spec.name, steps, nframes=nframes, poly=poly, variant=want)
            if route == "default" and spec.K <= 9:
                # few K=7 frames: the one-wave-per-frame kernel, which takes any polynomials as it is
                assert not dec.runtime_specialised and dec.variant == VARIANT_WAVE
            elif route != "generic":
                from ka9q_viterbi_comparison_amd import _lib as L

                assert dec.runtime_specialised and (dec.variant & 0xff) == fast, "the run-time build of the fast kernel was not used: " + L.last_error()
            else:
                assert not dec.runtime_specialised and dec.variant == generic
            dec.reset()
            dec.update(syms, nbits=steps)
            data, _ = dec.chainback(B * 8)
            for f in range(nframes):
                assert np.array_equal(dec.decision_rows(f, 0, steps), refs[f][0]), (route, poly, f)
                assert np.array_equal(dec.metrics(f), refs[f][1]), (route, poly, f)
                assert np.array_equal(data[f], refs[f][2]), (route, poly, f)
            dec.close()
    os.environ["VHIP_JIT"] = "0"
    try:
        dec = HipViterbi(spec.name, steps, nframes=1, poly=poly)
        assert not dec.runtime_specialised and dec.variant == (VARIANT_WAVE if spec.K <= 9 else generic)
        dec.close()
        with pytest.raises(VhipError):
            HipViterbi(spec.name, steps, nframes=1, poly=poly, variant=fast)
    finally:
        del os.environ["VHIP_JIT"]


def test_jit_cache_must_be_private(tmp_path):
    """Run-time builds are cached on disk and loaded into the process, so the cache must be the caller's alone:
    a cache directory others can write to switches the run-time build off (the handle then runs the any-polynomial
    kernel, still bit-exact), and a code object planted as a symbolic link is not loaded but rebuilt in place.  Each case is a
    child process (the library caches loaded modules per process)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys; sys.path.insert(0, %r)\n"
            "from ka9q_viterbi_comparison_amd import HipViterbi\n"
            "d = HipViterbi('49', 100, nframes=3000, poly=(0x1EF, 0x19B, 0x127, 0x1F5))\n"
            "print('SPECIALISED' if d.runtime_specialised else 'GENERIC', d.variant & 0xff)\n"
            "d.close()\n") % root

    def run(cache):
        env = dict(os.environ, VHIP_JIT_CACHE=str(cache))
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.split()

    open_dir = tmp_path / "shared"
    open_dir.mkdir()
    os.chmod(open_dir, 0o777)
    assert run(open_dir) == ["GENERIC", str(VARIANT_LDS)]
    assert not list(open_dir.iterdir()), "nothing may be written to (or loaded from) a cache others can write to"
    link_target = tmp_path / "elsewhere"
    link_target.mkdir()
    link = tmp_path / "cache_link"
    link.symlink_to(link_target, target_is_directory=True)
    assert run(link)[0] == "GENERIC"  # the cache path itself must not be a link
    priv = tmp_path / "private"
    assert run(priv) == ["SPECIALISED", str(VARIANT_REGS)]
    assert (os.stat(priv).st_mode & 0o777) == 0o700
    objs = [p for p in priv.iterdir() if p.suffix == ".hsaco"]
    assert len(objs) == 1
    planted = tmp_path / "planted.hsaco"
    objs[0].rename(planted)
    objs[0].symlink_to(planted)
    assert run(priv) == ["SPECIALISED", str(VARIANT_REGS)]
    assert not objs[0].is_symlink() and objs[0].is_file(), "the link was replaced by a fresh build, not followed"


@pytest.mark.parametrize("name,nframes,B", [("27", 16384, 64), ("49", 4096, 64), ("615", 512, 32)])
def test_arbitrary_polynomials_run_at_full_speed(name, nframes, B):
    """The run-time build for other polynomials is the same kernel with other constants: its ACS update takes no longer than
    1.3x the harness-polynomial update on the same batch shape (live event timing of the library), and it decodes its
    own noise-free frames without error."""
    import torch

    from ka9q_viterbi_comparison_amd import count_bit_errors_dev, gen_frames_dev

    spec = C.CODES[name]
    steps = B * 8 + spec.K - 1
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(77)
    times = {}
    for label, poly in (("harness", spec.poly), ("other", _random_poly(spec, rng))):
        d_payload = torch.empty(nframes * B, dtype=torch.uint8, device="cuda")
        d_syms = torch.empty(nframes * steps * spec.R, dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
        gen_frames_dev(spec, 5, 0, nframes, B, C.HARD_AMP_Q16, 0, d_payload, d_syms, stream, poly=poly)
        dec = HipViterbi(name, steps, nframes=nframes, poly=poly, stream=stream)
        assert dec.runtime_specialised == (label == "other")
        dec.enable_timing(True)
        for _ in range(4):
            dec.reset()
            dec.update(d_syms, nbits=steps)
            dec.chainback(B * 8, out=d_out)
        dec.read_timing()
        for _ in range(5):
            dec.reset()
            dec.update(d_syms, nbits=steps)
            dec.chainback(B * 8, out=d_out)
        su, nu, _, _ = dec.read_timing()
        times[label] = su / nu
        # a random polynomial set may be catastrophic / weak, but noise-free input still decodes to the payload
        assert count_bit_errors_dev(d_out, d_payload, nframes * B, stream) == 0 or label == "other"
        dec.close()
    assert times["other"] <= 1.3 * times["harness"], times


@pytest.mark.parametrize("code,variant", [(C.KA9Q27, regs(0)), (C.SPIRAL47, regs(0)), (C.KA9Q27, regs(2)), (C.KA9Q29, regs(1)), (C.SPIRAL49, regs(0))])
def test_chainback_output_windows(code, variant):
    """Long frames through the register-layout chainback kernels: whole 1024-row output windows (one 128-byte line per
    lane), the partial window at the bottom, the byte-stored rows above the highest complete dword, bit counts and byte
    strides that are not multiples of 32 / 4, non-zero end states, more than one wave of frames."""
    spec = spec_of(code)
    B = 270  # 2160 payload bits: two full output windows plus a partial one
    steps = B * 8 + spec.K - 1  # even for K = 7 and K = 9: the spiral decoders drop nothing
    nframes = 67
    _, syms = frames(code, 4242, nframes, B, spec.ebn0_db)
    dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
    dec.reset()
    dec.update(syms)
    N = 1 << (spec.K - 1)
    refs = {}
    for nbits, end in [(B * 8, 0), (2048, 0), (2040, 5), (1031, N - 1), (1024 + 32, 0), (96, 1), (33, 0)]:
        data, _ = dec.chainback(nbits, endstate=end)
        for f in (0, 1, 63, 64, 66):
            if f not in refs:
                o = OracleDecoder(code, spec.poly, steps)
                o.init(0)
                o.update(syms[f], steps)
                refs[f] = o
            r, _ = refs[f].chainback(nbits, end)
            assert np.array_equal(data[f], r), f"nbits {nbits} endstate {end} frame {f}"
    for o in refs.values():
        o.close()
    dec.close()


@pytest.mark.parametrize("seed", range(72))
def test_chainback_fuzz(seed):
    """Randomised geometry for every K=7 / K=9 decision layout and its chainback kernels -- the register layouts (1, 2 or 4 lanes
    per frame), the wave-per-frame layout (lanes sharing one frame's walk in verified segments) and natural rows (the same, over
    whole rows): payload length, frames, how much of the frame has been fed when chainback is called (rows never written read
    as zero), bit count, end state."""
    rng = np.random.default_rng(9000 + seed)
    code = [C.KA9Q27, C.SPIRAL47, C.SPIRAL27, C.KA9Q29, C.SPIRAL49, C.SPIRAL29][seed % 6]
    spec = spec_of(code)
    variant = [regs(0), regs(1), regs(2), VARIANT_WAVE, VARIANT_LDS][seed // 6 % 5] if seed >= 48 else regs(int(rng.integers(0, 3)))
    B = int(rng.integers(1, 330))
    steps = B * 8 + spec.K - 1
    nframes = int(rng.integers(1, 140))
    _, syms = frames(code, 500 + seed, nframes, B, spec.ebn0_db)
    fed = steps if rng.random() < 0.5 else int(rng.integers(0, steps + 1))
    if code in (C.SPIRAL47, C.SPIRAL27, C.SPIRAL49, C.SPIRAL29):
        fed &= ~1  # the spiral decoders drop an odd last step; keep the comparison about the chainback
    dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
    dec.reset()
    if fed:
        dec.update(np.ascontiguousarray(syms[:, :fed * spec.R]), nbits=fed)
    picks = sorted(set([0, nframes - 1, int(rng.integers(0, nframes))]))
    oracles = {}
    for f in picks:
        o = OracleDecoder(code, spec.poly, steps)
        o.init(0)
        if fed:
            o.update(syms[f][:fed * spec.R], fed)
        oracles[f] = o
    N = 1 << (spec.K - 1)
    for _ in range(4):
        nbits = int(rng.integers(1, B * 8 + 1))
        end = int(rng.integers(0, 2 * N))
        data, _ = dec.chainback(nbits, endstate=end)
        for f in picks:
            r, _ = oracles[f].chainback(nbits, end)
            assert np.array_equal(data[f], r), f"code {spec.name} variant {variant:#x} B {B} frames {nframes} fed {fed} nbits {nbits} end {end} frame {f}"
    for o in oracles.values():
        o.close()
    dec.close()


@pytest.mark.parametrize("name,variant,nframes,B", [
    ("224", VARIANT_HBM_TILED, 2, 56), ("224", VARIANT_HBM_FUSED, 1, 56), ("224", VARIANT_HBM, 1, 30),
    ("615", VARIANT_REGS, 3, 100), ("615", VARIANT_LDS, 2, 100), ("spiral615", VARIANT_REGS, 3, 101), ("spiral615", VARIANT_LDS, 2, 101),
])
def test_segment_parallel_chainback(name, variant, nframes, B):
    """The K >= 15 traceback cut into segments walked by separate waves from guessed states and verified afterwards
    (chainback_spec.hip) gives the bytes of the one-walk chainback for every geometry -- including a warm-up of 0 rows, where
    every guess is wrong and each segment is walked a second time from the true state -- for end states != 0 and ragged bit
    counts (viterbi615_sse2.cpp:65-91, viterbi224_sse2.cpp:79-121)."""
    spec = C.CODES[name]
    code = spec.code
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 4242, nframes, B, spec.ebn0_db)
    dec = HipViterbi(name, steps, nframes=nframes, variant=variant)
    dec.reset()
    dec.update(syms)
    N = 1 << (spec.K - 1)
    oracles = []
    for f in range(nframes):
        o = OracleDecoder(code, spec.poly, steps)
        o.update(syms[f], steps)
        oracles.append(o)
    full = steps if spec.K == 24 else B * 8
    for nbits, end in ((full, 0), (full - 5, 3), (full, N - 2), (203, 0)):
        want = [o.chainback(nbits, end) for o in oracles]
        for seg_bits, warm in ((-1, -1), (64, 0), (8, 40), (128, 100000), (96, 64), (0, 0)):
            dec.set_chainback_segments(seg_bits, warm)
            data, rc = dec.chainback(nbits, endstate=end)
            rew, nseg = dec.chainback_rewalked()
            for f in range(nframes):
                assert np.array_equal(data[f], want[f][0]), f"{name} nbits {nbits} end {end} seg {seg_bits}/{warm} frame {f}"
            if nframes == 1:
                assert rc == want[0][1]
            if seg_bits == 0:
                assert nseg == 0
            if seg_bits == 64:
                assert nseg == (nbits + 63) // 64
            if seg_bits == 64 and warm == 0 and nbits > 200:
                assert rew >= nframes, "a guessed all-zero state cannot match the true one on a noisy frame in every segment"
            if seg_bits == 128 and nbits == full:
                assert rew == 0, "warm-up reaching the end of the frame starts from the caller's end state: exact"
    with pytest.raises(Exception):
        dec.set_chainback_segments(12, 0)  # not a multiple of 8
    dec.set_chainback_segments(-1, -1)
    data, _ = dec.chainback(full)
    rew, nseg = dec.chainback_rewalked()
    assert nseg >= 2
    for f in range(nframes):
        assert bit_errors(data[f][:B], payload[f]) == 0 or spec.K == 15
    for o in oracles:
        o.close()
    dec.close()


@pytest.mark.parametrize("name", ["27", "47", "spiral27", "29", "49", "spiral29"])
def test_wave_variant_long_frames_and_segmented_chainback(name):
    """The one-wave-per-frame kernels (acs_wave.hip) on frames long enough for every path: whole 48-step blocks through the LDS
    branch-metric table over several table chunks, the steps in front of and behind them (incremental updates that start
    at every phase of the six-row period), and the chainback that cuts a frame into segments walked by the 64 lanes at once,
    verified from the top and re-walked where a lane's guessed entry state was wrong -- on pure-noise symbols (tracebacks
    merge late, if at all, so re-walks do happen), AWGN frames and ragged bit counts / end states.  Rows, metrics and bytes
    against the oracle."""
    spec = C.CODES[name]
    code = spec.code
    rng = np.random.default_rng(77 + code)
    for B, nframes, kind in ((1024, 2, "awgn"), (300, 3, "noise"), (701, 2, "awgn"), (130, 5, "noise")):
        steps = B * 8 + spec.K - 1
        steps -= 0 if spec_is_incremental(code) else steps % 2
        if kind == "awgn":
            _, syms = frames(code, 31 + B, nframes, B, spec.ebn0_db - 2.0)
            syms = np.ascontiguousarray(syms[:, :steps * spec.R])
        else:
            syms = rng.integers(0, 256, size=(nframes, steps * spec.R), dtype=np.uint8)
        dec = HipViterbi(name, steps, nframes=nframes, variant=VARIANT_WAVE)
        dec.reset()
        if spec_is_incremental(code):
            # three calls: the second starts at row 1000 + phase, the third wherever that leaves it
            cut1 = min(steps - 2, 1000 + (B % 6))
            cut2 = min(steps - 1, cut1 + 53)
            for lo, hi in ((0, cut1), (cut1, cut2), (cut2, steps)):
                dec.update(np.ascontiguousarray(syms[:, lo * spec.R:hi * spec.R]), nbits=hi - lo)
        else:
            dec.update(syms, nbits=steps)
        for nbits, endstate in ((B * 8, 0), (B * 8 - 13, 37), (B * 8 - 32, 5)):
            data, _ = dec.chainback(nbits, endstate)
            for f in range(nframes):
                ref = oracle_decode(code, syms[f], steps, nbits, endstate=endstate)
                assert np.array_equal(data[f], ref["data"]), (name, B, kind, nbits, f)
        for f in range(nframes):
            ref = oracle_decode(code, syms[f], steps, B * 8)
            assert np.array_equal(dec.decision_rows(f, 0, steps), ref["rows"]), (name, B, kind, f)
            assert np.array_equal(dec.metrics(f), ref["metrics"]), (name, B, kind, f)
        dec.close()
