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


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["27", "47", "29", "49", "615", "224"])
@pytest.mark.parametrize("mode", ["hard", "awgn"])
def test_device_generator_equals_host_twin(name, mode):
    """vhip_gen_frames_dev (csrc/framegen.hip) == vhip_gen_frames_host byte for byte: payload and symbols, hard and
    AWGN, every harness polynomial set, frame0 != 0, a frame count that is not a multiple of the launch geometry.
    bench.py and the full-size tests decode what the DEVICE generator produced, so this is what ties them to the host
    frames the oracle is run on."""
    import torch

    from ka9q_viterbi_comparison_amd import gen_frames_dev

    spec = C.CODES[name]
    B = 9 if spec.K == 24 else 37
    nframes, frame0, seed = 131, 1000003, 0x5EED
    steps = B * 8 + spec.K - 1
    amp_q16, nq = (C.HARD_AMP_Q16, 0) if mode == "hard" else (C.SOFT_AMP_Q16, noise_q12(spec.R, C.SOFT_AMP, spec.ebn0_db))
    hp, hs = gen_frames_host(spec, seed, frame0, nframes, B, amp_q16, nq)
    d_payload = torch.full((nframes * B,), 0xAA, dtype=torch.uint8, device="cuda")
    d_syms = torch.full((nframes * steps * spec.R,), 0xAA, dtype=torch.uint8, device="cuda")
    gen_frames_dev(spec, seed, frame0, nframes, B, amp_q16, nq, d_payload, d_syms, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_payload.cpu().numpy().reshape(nframes, B), hp)
    assert np.array_equal(d_syms.cpu().numpy().reshape(nframes, steps * spec.R), hs)
    if mode == "awgn":
        assert len(np.unique(hs)) > 100  # really soft symbols, not a degenerate noise scale


@pytest.mark.gpu
def test_device_bit_error_count_equals_numpy_popcount():
    """vhip_count_bit_errors_dev (the BER reduction of src/util.h:64-73) against numpy on ragged sizes."""
    import torch

    from ka9q_viterbi_comparison_amd import count_bit_errors_dev

    rng = np.random.default_rng(5)
    for n in (1, 3, 255, 4096, 1000003):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        b = a.copy()
        flip = rng.random(n) < 0.3
        b[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        got = count_bit_errors_dev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), n, torch.cuda.current_stream().cuda_stream)
        assert got == want, n
    z = torch.zeros(16, dtype=torch.uint8, device="cuda")
    assert count_bit_errors_dev(z, z, 16) == 0
