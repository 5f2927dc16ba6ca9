"""Golden fixtures (tests/golden/*.npz, written by oracle/make_golden.py from the compiled reference decoders):
the CPU oracle must reproduce them (runs anywhere), and so must the HIP path (-m gpu)."""
import hashlib
import os

import numpy as np
import pytest

from ka9q_viterbi_comparison_amd import codes as C
from oracle_lib import OracleDecoder

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = {C.KA9Q27: "ka9q27", C.KA9Q29: "ka9q29", C.KA9Q615: "ka9q615", C.KA9Q224: "ka9q224", C.SPIRAL47: "spiral47",
         C.SPIRAL49: "spiral49", C.SPIRAL27: "spiral27", C.SPIRAL29: "spiral29", C.SPIRAL615: "spiral615"}
HIP_NAMES = {C.KA9Q27: "27", C.KA9Q29: "29", C.KA9Q615: "615", C.KA9Q224: "224", C.SPIRAL47: "47", C.SPIRAL49: "49",
             C.SPIRAL27: "spiral27", C.SPIRAL29: "spiral29", C.SPIRAL615: "spiral615"}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def load(code):
    return np.load(os.path.join(GOLD, NAMES[code] + ".npz"), allow_pickle=False)


def check_case(g, name, rows, metrics, data, rc, data_ragged, data_tail=None):
    K = int(g["K"])
    assert np.array_equal(sha(rows), g[f"{name}_rows_sha256"]), "decision rows"
    if K <= 9:
        assert np.array_equal(rows, g[f"{name}_rows"])
    else:
        each = g[f"{name}_row_sha256_each"]
        bad = [i for i in range(rows.shape[0]) if not np.array_equal(sha(rows[i]), each[i])]
        assert not bad, f"rows differ first at {bad[:3]}"
    if K < 24:
        assert np.array_equal(metrics, g[f"{name}_metrics"]), "metrics"
    else:
        assert np.array_equal(sha(metrics), g[f"{name}_metrics"]), "metrics"
    assert np.array_equal(data, g[f"{name}_data"]), "decoded bytes"
    assert rc == int(g[f"{name}_rc"]), "return code"
    assert np.array_equal(data_ragged, g[f"{name}_data_ragged"]), "ragged chainback"
    if data_tail is not None:
        assert np.array_equal(data_tail, g[f"{name}_data_with_tail"])


@pytest.mark.parametrize("code", sorted(NAMES))
def test_oracle_reproduces_golden(code):
    g = load(code)
    steps, nrows, B = int(g["steps"]), int(g["nrows"]), int(g["payload_bytes"])
    for name in g["case_names"]:
        o = OracleDecoder(code, [int(x) for x in g["poly"]], steps)
        o.update(g[f"{name}_syms"], steps)
        data, rc = o.chainback(B * 8)
        ragged, _ = o.chainback(B * 8 - 3, 5)
        tail = o.chainback(steps)[0] if int(g["K"]) == 24 else None
        check_case(g, name, o.rows(nrows), o.metrics(), data, rc, ragged, tail)
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("code", sorted(HIP_NAMES))
def test_hip_reproduces_golden(code):
    from ka9q_viterbi_comparison_amd import HipViterbi

    g = load(code)
    steps, nrows, B = int(g["steps"]), int(g["nrows"]), int(g["payload_bytes"])
    variants = [0]
    if int(g["K"]) == 15:
        variants = [1, 2]
    if int(g["K"]) == 24:
        variants = [3, 4, 5]  # 5 = HBM_TILED, what AUTO picks for the harness polynomials
    if int(g["K"]) <= 9:
        variants = [1] + [2 | ((lb + 1) << 8) for lb in (0, 1, 2)] + [6]  # 6 = wave(s) per frame (acs_wave.hip)
    for variant in variants:
        for name in g["case_names"]:
            d = HipViterbi(HIP_NAMES[code], steps, nframes=1, variant=variant)
            d.reset()
            d.update(g[f"{name}_syms"].reshape(1, -1))
            data, rc = d.chainback(B * 8)
            ragged, _ = d.chainback(B * 8 - 3, 5)
            tail = d.chainback(steps)[0][0] if int(g["K"]) == 24 else None
            check_case(g, name, d.decision_rows(0, 0, nrows), d.metrics(0), data[0], rc, ragged[0], tail)
            d.close()
