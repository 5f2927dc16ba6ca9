"""Synthetic frame generator (host twin in libviterbi_hip.so; analogue of src/util.h:8-62)."""
import numpy as np
import pytest

from common import frames, spec_of
from ka9q_viterbi_comparison_amd import codes as C
from ka9q_viterbi_comparison_amd.decoder import gen_frames_host, noise_q12
from oracle_lib import encode


@pytest.mark.parametrize("name", sorted(C.CODES))
def test_hard_symbols_equal_oracle_encoder(name):
    """amp=127.5, no noise -> the reference's 0/255 symbols (src/util.h:36) of the App. A.1 encoder."""
    spec = C.CODES[name]
    payload, syms = gen_frames_host(spec, 99, 3, 4, 20, C.HARD_AMP_Q16, 0)
    for f in range(4):
        bits = encode(spec.K, spec.R, spec.poly, payload[f])
        assert np.array_equal(syms[f], bits * 255)


def test_deterministic_and_frame_indexed():
    spec = C.CODES["27"]
    nq = noise_q12(2, 64.0, 4.0)
    p1, s1 = gen_frames_host(spec, 1, 0, 8, 16, C.SOFT_AMP_Q16, nq)
    p2, s2 = gen_frames_host(spec, 1, 5, 3, 16, C.SOFT_AMP_Q16, nq)  # frames 5..7
    assert np.array_equal(p1[5:], p2) and np.array_equal(s1[5:], s2)
    p3, s3 = gen_frames_host(spec, 2, 0, 8, 16, C.SOFT_AMP_Q16, nq)
    assert not np.array_equal(p1, p3)
    assert len({bytes(r) for r in p1}) == 8  # distinct payloads per frame


def test_noise_statistics():
    """Irwin-Hall(8) term scaled to the requested sigma: mean ~ 127.5 +- amp, std ~ sigma (before clamping)."""
    spec = C.CODES["27"]
    amp, ebn0 = 32.0, 6.0  # small amplitude + high SNR: clamping negligible
    nq = noise_q12(spec.R, amp, ebn0)
    sigma = amp / np.sqrt(2.0 * 10 ** (ebn0 / 10) / spec.R)
    payload, syms = gen_frames_host(spec, 7, 0, 64, 256, int(amp * 65536), nq)
    bits = np.stack([encode(spec.K, spec.R, spec.poly, payload[f]) for f in range(64)])
    resid = syms.astype(np.float64) - (127.5 + amp * (2.0 * bits - 1.0))
    assert abs(resid.mean()) < 0.1
    assert abs(resid.std() - np.sqrt(sigma ** 2 + 1 / 12.0)) / sigma < 0.02
    # roughly normal: kurtosis of Irwin-Hall(8) is 3 - 0.15
    k = ((resid - resid.mean()) ** 4).mean() / resid.var() ** 2
    assert 2.7 < k < 3.0


def test_common_frames_helper_shapes():
    for code in (C.KA9Q27, C.SPIRAL49, C.KA9Q615):
        spec = spec_of(code)
        p, s = frames(code, 3, 2, 5, spec.ebn0_db)
        assert p.shape == (2, 5) and s.shape == (2, (40 + spec.K - 1) * spec.R)
