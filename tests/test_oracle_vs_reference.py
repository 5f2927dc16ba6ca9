"""Pins the CPU oracle (oracle/viterbi_oracle.c) against the GENUINE reference decoders compiled from
/root/reference by oracle/Makefile into oracle/_ref/ (ka9q_libfec_port/*.cpp, spiral/*.cpp).

The reference ships no tests or golden vectors: running its own objects is the only pin there is.
Compared per frame: every decision row, the final path metrics, the decoded bytes and the return code.
Runs wherever oracle/_ref/*.so exists (the build container, and the GPU box because built .so files travel).
"""
import numpy as np
import pytest

from common import ORACLE_CODES, bit_errors, frames, spec_of
from ka9q_viterbi_comparison_amd import codes as C
from oracle_lib import OracleDecoder, RefDecoder, have_ref, ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference): make -C oracle")

ALL = sorted(ORACLE_CODES)
SMALL = [c for c in ALL if c != C.KA9Q224]


def both(code, steps):
    spec = spec_of(code)
    return OracleDecoder(code, spec.poly, steps), RefDecoder(code, spec.poly, steps, w32=(code == C.KA9Q615))


def compare(o, r, nrows, nbits, endstate=0):
    assert o.rows_written == nrows
    assert np.array_equal(o.rows(nrows), r.rows(nrows)), "decision rows"
    assert np.array_equal(o.metrics(), r.metrics()), "path metrics"
    do, rco = o.chainback(nbits, endstate)
    dr, rcr = r.chainback(nbits, endstate)
    assert np.array_equal(do, dr), "decoded bytes"
    assert rco == rcr, "return code"
    return dr


@pytest.mark.parametrize("code", SMALL)
@pytest.mark.parametrize("ebn0", ["hard", "awgn", "awgn-3dB"])
def test_frames_match_reference(code, ebn0):
    spec = spec_of(code)
    B = 16 if spec.K == 15 else 128
    nframes = 3 if spec.K == 15 else 40
    steps = B * 8 + spec.K - 1
    db = None if ebn0 == "hard" else (spec.ebn0_db if ebn0 == "awgn" else spec.ebn0_db - 3.0)
    payload, syms = frames(code, 1000 + code, nframes, B, db)
    for f in range(nframes):
        o, r = both(code, steps)
        o.update(syms[f], steps)
        r.update(syms[f], steps)
        dec = compare(o, r, steps, B * 8)
        if ebn0 == "hard":
            assert bit_errors(dec, payload[f]) == 0  # noise-free decode is error free: pins the encoder convention
        o.close()
        r.close()


def test_k24_matches_reference_both_chainback_conventions():
    """K=24 at the harness frame size (8 bytes, src/main.cpp:414): rows, metrics, and chainback with nbits (the
    harness call, wrong bits but deterministic) and nbits+K-1 (correct) -- SURVEY.md §0.4."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 8
    steps = B * 8 + spec.K - 1
    for db in (None, spec.ebn0_db):
        payload, syms = frames(code, 7, 1, B, db)
        o, r = both(code, steps)
        o.update(syms[0], steps)
        r.update(syms[0], steps)
        short = compare(o, r, steps, B * 8)
        assert bit_errors(short, payload[0]) > 0  # the reference's own harness call does NOT decode K=24 correctly
        long_o, _ = o.chainback(steps)
        long_r, _ = r.chainback(steps)
        assert np.array_equal(long_o, long_r)
        assert bit_errors(long_r[:B], payload[0]) == 0
        o.close()
        r.close()


def test_k24_renormalisation_event():
    """A frame long enough for new[0] >= 25000 (viterbi224_sse2.cpp:226): exercises the wrapping subtract."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 56
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 11, 1, B, spec.ebn0_db)
    o, r = both(code, steps)
    o.update(syms[0], steps)
    r.update(syms[0], steps)
    assert o.renorms >= 1, "input did not trigger a renormalisation; lengthen the frame"
    assert np.array_equal(o.metrics(), r.metrics())
    # compare a sample of rows (each is 1 MiB) and the decode
    ro, rr = o.rows(steps), r.rows(steps)
    assert np.array_equal(ro, rr)
    do, _ = o.chainback(steps)
    dr, _ = r.chainback(steps)
    assert np.array_equal(do, dr)
    assert bit_errors(dr[:B], payload[0]) == 0
    o.close()
    r.close()


@pytest.mark.parametrize("code", [C.KA9Q27, C.KA9Q29, C.KA9Q615])
def test_incremental_update_matches_reference(code):
    """ka9q update appends at dp (viterbi27_sse2.cpp:121,174): ragged splits == reference called the same way."""
    spec = spec_of(code)
    B = 24
    steps = B * 8 + spec.K - 1
    _, syms = frames(code, 5, 1, B, spec.ebn0_db)
    o, r = both(code, steps)
    off = 0
    for n in (1, 2, 5, 64, steps - 72):
        o.update(syms[0][off * spec.R:], n)
        r.update(syms[0][off * spec.R:], n)
        off += n
    compare(o, r, steps, B * 8)
    o.close()
    r.close()


@pytest.mark.parametrize("code", [C.SPIRAL47, C.SPIRAL49, C.SPIRAL27, C.SPIRAL29, C.SPIRAL615])
def test_spiral_restart_and_odd_count(code):
    """spiral update restarts at decisions[0] on every call and runs nbits/2 double steps (spiral47.cpp:536-538)."""
    spec = spec_of(code)
    B = 8
    steps = B * 8 + spec.K - 1
    _, syms = frames(code, 6, 1, B, spec.ebn0_db)
    o, r = both(code, steps)
    o.update(syms[0], 11)
    r.update(syms[0], 11)
    assert o.rows_written == 10
    assert np.array_equal(o.rows(10), r.rows(10))
    o.update(syms[0], steps)
    r.update(syms[0], steps)
    n = (steps // 2) * 2
    assert np.array_equal(o.rows(n), r.rows(n))
    assert np.array_equal(o.metrics(), r.metrics())
    o.close()
    r.close()


@pytest.mark.parametrize("code", SMALL)
def test_start_state_end_state_ragged_bits(code):
    spec = spec_of(code)
    B = 12
    steps = B * 8 + spec.K - 1
    N = 1 << (spec.K - 1)
    _, syms = frames(code, 8, 1, B, spec.ebn0_db)
    for start, end, nbits in [(5, 0, B * 8), (N - 1, 3, B * 8 - 3), (0, N + 9, 13), (N + 2, 1, 1)]:
        o, r = both(code, steps)
        o.init(start)
        r.init(start)
        o.update(syms[0], steps)
        r.update(syms[0], steps)
        compare(o, r, (steps // 2) * 2 if code >= C.SPIRAL47 else steps, nbits, end)
        o.close()
        r.close()


def test_k24_start_state_end_state_ragged_bits():
    """init_viterbi224_sse2(p, starting_state != 0) (viterbi224_sse2.cpp:32-47) and chainback_viterbi224_sse2 with
    endstate != 0 / ragged bit counts (:79-121, endstate masked at :90): a short frame keeps the 8M-state scalar
    walk to seconds.  Rows, metrics, decoded bytes and return code against the genuine reference object."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 5
    steps = B * 8 + spec.K - 1
    N = 1 << (spec.K - 1)
    _, syms = frames(code, 8, 1, B, spec.ebn0_db)
    for start, end, nbits in [(5, 0, B * 8), (N - 1, 3, steps - 3), (N + 2, N + 9, 13), (0x2AAAAA, 0x555555, steps)]:
        o, r = both(code, steps)
        assert o.init(start) == r.init(start)
        o.update(syms[0], steps)
        r.update(syms[0], steps)
        compare(o, r, steps, nbits, end)
        # a second chainback on the same history with another end state (the walk does not modify it)
        do, _ = o.chainback(nbits, end ^ 0x40001)
        dr, _ = r.chainback(nbits, end ^ 0x40001)
        assert np.array_equal(do, dr)
        o.close()
        r.close()


def test_k24_adversarial_symbols():
    """Saturating adds of the K=24 family at their corners (all-0 / all-255 / mid-scale / alternating symbols)."""
    code = C.KA9Q224
    spec = spec_of(code)
    B = 3
    steps = B * 8 + spec.K - 1
    n = steps * spec.R
    rng = np.random.default_rng(7)
    for s in [np.zeros(n, np.uint8), np.full(n, 255, np.uint8), np.full(n, 128, np.uint8),
              np.tile(np.array([0, 255], np.uint8), n)[:n], rng.integers(0, 256, n, dtype=np.uint8)]:
        o, r = both(code, steps)
        o.update(s, steps)
        r.update(s, steps)
        compare(o, r, steps, steps)
        o.close()
        r.close()


def test_ka9q615_lp64_chainback_defect_is_documented():
    """the shipped K=15 chainback indexes 64-bit words on LP64 and decodes wrongly even without
    noise; update (rows, metrics) is unaffected.  The oracle follows the 32-bit-word semantics."""
    code = C.KA9Q615
    spec = spec_of(code)
    assert ref(False).ref_sizeof_long() == 8 and ref(True).ref_sizeof_long() == 4
    B = 32
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 3, 1, B, None)
    shipped = RefDecoder(code, spec.poly, steps, w32=False)
    fixed = RefDecoder(code, spec.poly, steps, w32=True)
    shipped.update(syms[0], steps)
    fixed.update(syms[0], steps)
    assert np.array_equal(shipped.rows(steps), fixed.rows(steps))
    assert np.array_equal(shipped.metrics(), fixed.metrics())
    good, _ = fixed.chainback(B * 8)
    assert bit_errors(good, payload[0]) == 0
    shipped.close()
    fixed.close()


@pytest.mark.parametrize("code", SMALL)
def test_adversarial_symbols(code):
    """Extreme and patterned symbols (all 0, all 255, alternating, ramps): saturation / wrap-around corner cases."""
    spec = spec_of(code)
    B = 16
    steps = B * 8 + spec.K - 1
    n = steps * spec.R
    rng = np.random.default_rng(99)
    patterns = [
        np.zeros(n, np.uint8), np.full(n, 255, np.uint8), np.full(n, 128, np.uint8), np.full(n, 127, np.uint8),
        np.tile(np.array([0, 255], np.uint8), n)[:n], (np.arange(n) % 256).astype(np.uint8),
        rng.integers(0, 256, n, dtype=np.uint8), rng.choice(np.array([0, 255], np.uint8), n),
    ]
    for s in patterns:
        o, r = both(code, steps)
        o.update(s, steps)
        r.update(s, steps)
        compare(o, r, (steps // 2) * 2 if code >= C.SPIRAL47 else steps, B * 8)
        o.close()
        r.close()


@pytest.mark.parametrize("code", [C.KA9Q27, C.SPIRAL49, C.KA9Q615])
def test_batch_helper_equals_single_frame_calls(code):
    """oracle_lib.decode_batch_cpu (what the whole-batch GPU tests compare with: every frame decoded on the host cores, several decoder
    objects in threads) gives the bytes of one-frame-at-a-time calls on the restatement, frame for frame."""
    import oracle_lib as ol

    spec = spec_of(code)
    B, n = 24, 37
    steps = B * 8 + spec.K - 1
    steps -= 0 if spec.family.startswith("ka9q") else steps % 2
    _, syms = frames(code, 99, n, B, spec.ebn0_db - 2.0)
    syms = np.ascontiguousarray(syms[:, :steps * spec.R])
    got, kind = ol.decode_batch_cpu(code, spec.poly, syms, steps, B * 8, threads=3)
    assert kind == ("reference" if ol.have_ref() else "port")
    for f in range(n):
        o = OracleDecoder(code, spec.poly, steps)
        o.update(syms[f], steps)
        assert np.array_equal(got[f], o.chainback(B * 8)[0]), f
        o.close()
