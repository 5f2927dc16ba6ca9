"""Call granularity and batch shape at their extremes.  The reference's update takes any number of steps per call
(viterbi27_sse2.cpp:119-175 walks `nbits` steps from wherever the previous call stopped); the kernels here unroll a period of
K-1 steps, run whole 48-step blocks through a table, regroup every 7 steps (K=15) or run 9 + 14 steps per pass pair (K=24) --
so every call boundary inside such a unit is a separate path: one step per call walks through all of them."""
import numpy as np
import pytest

from common import frames, spec_of
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C
from ka9q_viterbi_comparison_amd import VARIANT_AUTO, VARIANT_HBM, VARIANT_LDS, VARIANT_REGS, VARIANT_WAVE
from oracle_lib import OracleDecoder

pytestmark = pytest.mark.gpu


def _oracle(code, syms, steps, nbits):
    spec = spec_of(code)
    o = OracleDecoder(code, spec.poly, steps)
    o.update(syms, steps)
    out = (o.chainback(nbits)[0], o.metrics(), o.rows(steps))
    o.close()
    return out


@pytest.mark.parametrize("code,variant,nframes", [
    (C.KA9Q27, VARIANT_WAVE, 2), (C.KA9Q27, VARIANT_REGS, 70), (C.KA9Q27, VARIANT_LDS, 3),
    (C.KA9Q29, VARIANT_WAVE, 2), (C.KA9Q29, VARIANT_REGS, 20), (C.KA9Q615, VARIANT_AUTO, 2), (C.KA9Q615, VARIANT_LDS, 1),
    (C.KA9Q224, VARIANT_AUTO, 1), (C.KA9Q224, VARIANT_HBM, 1)])
def test_one_step_per_call(code, variant, nframes):
    spec = spec_of(code)
    B = 3 if spec.K == 24 else 14
    steps = B * 8 + spec.K - 1
    nbits = steps if spec.K == 24 else B * 8
    _, syms = frames(code, 77 + code, nframes, B, spec.ebn0_db)
    s3 = np.ascontiguousarray(syms).reshape(nframes, steps, spec.R)
    dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
    dec.reset()
    for i in range(steps):
        dec.update(np.ascontiguousarray(s3[:, i:i + 1, :]), nbits=1)
    assert dec.rows_written == steps
    data, _ = dec.chainback(nbits)
    for f in range(nframes):
        want = _oracle(code, syms[f], steps, nbits)
        assert np.array_equal(data[f], want[0]), (code, variant, f)
        assert np.array_equal(dec.metrics(f), want[1]), (code, variant, f)
        assert np.array_equal(dec.decision_rows(f, 0, steps), want[2]), (code, variant, f)
    dec.close()


@pytest.mark.parametrize("code,variant", [(C.KA9Q27, VARIANT_WAVE), (C.KA9Q27, VARIANT_REGS), (C.KA9Q29, VARIANT_WAVE),
                                          (C.KA9Q29, VARIANT_REGS), (C.KA9Q615, VARIANT_AUTO)])
def test_every_split_point(code, variant):
    """Two calls, the cut at every step of a window that covers more than one unrolled unit of each kernel (two periods, a
    whole 48-step table block and its neighbours)."""
    spec = spec_of(code)
    B = 20
    steps = B * 8 + spec.K - 1
    nframes = 2
    _, syms = frames(code, 5, nframes, B, spec.ebn0_db)
    s3 = np.ascontiguousarray(syms).reshape(nframes, steps, spec.R)
    want = [_oracle(code, syms[f], steps, B * 8) for f in range(nframes)]
    dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant)
    for cut in list(range(1, 64)) + [95, 96, 97, 101, steps - 2, steps - 1]:
        dec.reset()
        dec.update(np.ascontiguousarray(s3[:, :cut, :]), nbits=cut)
        dec.update(np.ascontiguousarray(s3[:, cut:, :]), nbits=steps - cut)
        data, _ = dec.chainback(B * 8)
        for f in range(nframes):
            assert np.array_equal(data[f], want[f][0]), (code, variant, cut, f)
            assert np.array_equal(dec.metrics(f), want[f][1]), (code, variant, cut, f)
        assert np.array_equal(dec.decision_rows(0, 0, steps), want[0][2]), (code, variant, cut)
    dec.close()


@pytest.mark.parametrize("name,nframes", [("27", 300000), ("47", 200000), ("29", 150000), ("49", 100000), ("615", 30000), ("spiral615", 20000)])
def test_very_many_very_short_frames(name, nframes):
    """One payload byte per frame: the batch is almost all tail, every frame ends inside its first period / block / regroup,
    the chainbacks decode 8 bits, and the grids are as wide as they get.  Noise-free: every frame must come back exactly; a
    sample of frames also against the oracle on AWGN symbols."""
    import torch

    from ka9q_viterbi_comparison_amd import count_bit_errors_dev, gen_frames_dev, noise_q12

    spec = C.CODES[name]
    B = 1
    steps = 8 + spec.K - 1
    steps -= steps % 2 if spec.family.startswith("spiral") else 0
    stream = torch.cuda.current_stream().cuda_stream
    d_payload = torch.empty(nframes * B, dtype=torch.uint8, device="cuda")
    d_syms = torch.empty(nframes * (8 + spec.K - 1) * spec.R, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
    for ebn0 in (None, spec.ebn0_db):
        if ebn0 is None:
            gen_frames_dev(spec, 3, 0, nframes, B, C.HARD_AMP_Q16, 0, d_payload, d_syms, stream)
        else:
            gen_frames_dev(spec, 3, 0, nframes, B, C.SOFT_AMP_Q16, noise_q12(spec.R, C.SOFT_AMP, ebn0), d_payload, d_syms, stream)
        full = d_syms.view(nframes, (8 + spec.K - 1) * spec.R)
        use = full[:, :steps * spec.R].contiguous() if steps != 8 + spec.K - 1 else d_syms
        dec = HipViterbi(name, steps, nframes=nframes)
        dec.reset()
        dec.update(use, nbits=steps)
        dec.chainback(8, out=d_out)
        dec.sync()
        if ebn0 is None:
            assert count_bit_errors_dev(d_out, d_payload, nframes * B, stream=stream) == 0
        else:
            host = use.cpu().numpy().reshape(nframes, steps * spec.R)
            out = d_out.cpu().numpy()
            for f in (0, 1, 63, 64, nframes // 2 + 1, nframes - 65, nframes - 1):
                o = OracleDecoder(spec.code, spec.poly, steps)
                o.update(host[f], steps)
                assert o.chainback(8)[0][0] == out[f], (name, f)
                assert np.array_equal(o.metrics(), dec.metrics(f)), (name, f)
                o.close()
        dec.close()
