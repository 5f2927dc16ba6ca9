"""Soak test of the K <= 9 kernels and their chainbacks under the handle's own kernel selection -- the wave-per-frame kernels
(acs_wave.hip) for the small batches, the register kernels (acs_regs.hip) for the large ones: random codes, frame counts,
lengths (one in four long enough for several branch-metric table chunks), start / end states, ragged bit counts, (ka9q)
update-call boundaries and pipelined handles; decision rows, metrics and decoded bytes of sampled frames compared with the
CPU oracle.  python tools/soak_k9.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from common import frames
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C
from oracle_lib import OracleDecoder

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(777)
t0 = time.time()
rounds = checked = 0
while time.time() - t0 < budget:
    name = str(rng.choice(["27", "29", "47", "49", "spiral27", "spiral29"]))
    spec = C.CODES[name]
    ka9q = name in ("27", "29")
    B = int(rng.integers(2, 80)) if rng.random() < 0.75 else int(rng.integers(150, 500))
    steps = B * 8 + spec.K - 1
    nframes = int(rng.choice([1, 3, 64, 65, 130, 700, 3000, 5000])) if B < 150 else int(rng.choice([1, 2, 5, 70]))
    ebn0 = float(rng.choice([spec.ebn0_db, 0.0, -5.0]))
    _, syms = frames(spec.code, int(rng.integers(1, 1 << 30)), nframes, B, ebn0)
    N = 1 << (spec.K - 1)
    start = int(rng.integers(0, N))
    depth = int(rng.choice([1, 2]))
    dec = HipViterbi(name, steps, nframes=nframes, pipeline_depth=depth)
    dec.reset(start)
    bounds = [0, steps]
    if ka9q:
        cuts = sorted(set(int(x) for x in rng.integers(1, steps, size=int(rng.integers(0, 4)))))
        bounds = [0] + cuts + [steps]
    s3 = syms.reshape(nframes, steps, spec.R)
    for a, b in zip(bounds[:-1], bounds[1:]):
        dec.update(np.ascontiguousarray(s3[:, a:b, :]), nbits=b - a)
    nbits = int(rng.integers(1, B * 8 + 1))
    end = int(rng.integers(0, 2 * N))
    data, _ = dec.chainback(nbits, endstate=end)
    for f in sorted(set([0, nframes - 1] + [int(x) for x in rng.integers(0, nframes, size=3)])):
        o = OracleDecoder(spec.code, spec.poly, steps)
        o.init(start)
        o.update(syms[f], steps)
        want, _ = o.chainback(nbits, end)
        oks = (np.array_equal(data[f], want), np.array_equal(dec.metrics(f), o.metrics()), np.array_equal(dec.decision_rows(f, 0, steps), o.rows(steps)))
        ok = all(oks)
        o.close()
        if not ok:
            print("MISMATCH (bytes, metrics, rows ok?)", oks, name, nframes, B, start, bounds, nbits, end, depth, f, "variant", dec.variant, "round", rounds, flush=True)
            sys.exit(1)
        checked += 1
    dec.close()
    rounds += 1
print(f"soak ok: {rounds} K<=9 batch decodes, {checked} frames checked against the oracle (rows, metrics, bytes), {time.time() - t0:.0f} s")
