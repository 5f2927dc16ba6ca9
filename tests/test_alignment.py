"""Device buffers at odd addresses.  The reference takes any `unsigned char *` (its loads are _mm_loadu / byte reads,
viterbi27_sse2.cpp:131-136); the kernels here fetch symbols as dwords and store decoded bytes as dwords where they can, so a
symbol or output pointer that is not 4-byte aligned -- a frame in the middle of a caller's larger buffer -- must take the
byte-wise paths and give the same rows, metrics and bytes."""
import numpy as np
import pytest
import torch

from common import frames, spec_of
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C
from ka9q_viterbi_comparison_amd import VARIANT_AUTO, VARIANT_LDS, VARIANT_REGS, VARIANT_WAVE

pytestmark = pytest.mark.gpu

CASES = [
    (C.KA9Q27, 1, VARIANT_AUTO), (C.KA9Q27, 3, VARIANT_WAVE), (C.KA9Q27, 130, VARIANT_REGS), (C.KA9Q27, 5, VARIANT_LDS),
    (C.SPIRAL47, 2, VARIANT_WAVE), (C.SPIRAL47, 70, VARIANT_REGS),
    (C.KA9Q29, 2, VARIANT_WAVE), (C.KA9Q29, 67, VARIANT_REGS), (C.SPIRAL49, 3, VARIANT_AUTO), (C.SPIRAL49, 66, VARIANT_REGS),
    (C.KA9Q615, 3, VARIANT_AUTO), (C.SPIRAL615, 2, VARIANT_AUTO), (C.KA9Q224, 1, VARIANT_AUTO),
]


@pytest.mark.parametrize("code,nframes,variant", CASES)
def test_unaligned_symbol_and_output_pointers(code, nframes, variant):
    spec = spec_of(code)
    B = 7 if spec.K == 24 else 37  # odd byte counts: the per-frame strides are odd as well
    steps = B * 8 + spec.K - 1
    steps -= steps % 2 if spec.family.startswith("spiral") else 0
    nbits = B * 8 if spec.K != 24 else steps  # K=24: the payload comes out of the nbits+K-1 call
    nbytes = (nbits + 7) // 8
    payload, syms = frames(code, 21 + code, nframes, B, ebn0_db=spec.ebn0_db + 2.0)
    syms = np.ascontiguousarray(syms[:, :steps * spec.R])
    flat = torch.from_numpy(syms.reshape(-1))
    results = []
    for off_in, off_out in ((0, 0), (1, 0), (2, 3), (3, 1), (0, 2)):
        buf = torch.zeros(flat.numel() + 8, dtype=torch.uint8, device="cuda")
        buf[off_in:off_in + flat.numel()] = flat.cuda()
        d_syms = buf[off_in:off_in + flat.numel()]
        obuf = torch.zeros(nframes * nbytes + 8, dtype=torch.uint8, device="cuda")
        d_out = obuf[off_out:off_out + nframes * nbytes]
        assert d_syms.data_ptr() % 4 == off_in % 4 and d_out.data_ptr() % 4 == off_out % 4
        dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
        dec.reset()
        dec.update(d_syms, nbits=steps)
        dec.chainback(nbits, out=d_out)
        dec.sync()
        out = d_out.cpu().numpy().reshape(nframes, nbytes).copy()
        met = np.stack([dec.metrics(f) for f in range(min(nframes, 3))])
        # nothing outside the output window was written
        guard = obuf.cpu().numpy()
        assert not guard[:off_out].any() and not guard[off_out + nframes * nbytes:].any()
        dec.close()
        results.append((out, met))
    for out, met in results[1:]:
        assert np.array_equal(out, results[0][0])
        assert np.array_equal(met, results[0][1])
    if spec.K != 24:
        # and the aligned result is the oracle's
        from oracle_lib import OracleDecoder

        for f in range(min(nframes, 3)):
            o = OracleDecoder(code, spec.poly, steps)
            o.update(syms[f], steps)
            assert np.array_equal(o.chainback(nbits)[0], results[0][0][f])
            o.close()


@pytest.mark.parametrize("name,nframes,nbits", [("27", 70, 1000), ("27", 3, 61), ("47", 65, 400), ("49", 17, 333), ("615", 3, 300)])
def test_unaligned_pointers_fused_windowed_decode(name, nframes, nbits):
    """The fused sliding-window decode stores four decoded bytes at a time where the output allows it (K=7 steady state): the
    same bytes must come out at every output offset, nothing outside the window may be written."""
    spec = C.CODES[name]
    steps = nbits + spec.K - 1  # the spiral decoders drop an odd last step themselves (spiral47.cpp:536-538)
    B = (nbits + 7) // 8
    _, syms = frames(spec.code, 5, nframes, B, ebn0_db=spec.ebn0_db)
    syms = np.ascontiguousarray(syms[:, :steps * spec.R])
    flat = torch.from_numpy(syms.reshape(-1))
    nbytes = (nbits + 7) // 8
    results = []
    for off_in, off_out in ((0, 0), (1, 1), (2, 3), (3, 2)):
        buf = torch.zeros(flat.numel() + 8, dtype=torch.uint8, device="cuda")
        buf[off_in:off_in + flat.numel()] = flat.cuda()
        obuf = torch.full((nframes * nbytes + 8,), 0xA5, dtype=torch.uint8, device="cuda")
        d_out = obuf[off_out:off_out + nframes * nbytes]
        dec = HipViterbi(name, steps, nframes=nframes)
        assert dec.window is not None
        dec.decode_windowed(buf[off_in:off_in + flat.numel()], nbits, d_out)
        dec.sync()
        guard = obuf.cpu().numpy()
        assert (guard[:off_out] == 0xA5).all() and (guard[off_out + nframes * nbytes:] == 0xA5).all()
        results.append(d_out.cpu().numpy().copy())
        dec.close()
    for r in results[1:]:
        assert np.array_equal(r, results[0])
