"""Soak test of the K=24 tiled passes: random frame lengths, start states and update-call boundaries on noisy frames (several
renormalisations each, viterbi224_sse2.cpp:226-246), metrics / decoded bytes / sampled decision rows compared with the CPU
oracle.  python tools/soak_k24.py [seconds]   (the scalar oracle walks 8M states per step: ~10 s per frame)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from common import frames
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, VARIANT_HBM_TILED
from oracle_lib import OracleDecoder

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2024)
spec = C.CODES["224"]
t0 = time.time()
rounds = renorms = 0
while time.time() - t0 < budget:
    B = int(rng.integers(30, 90))
    steps = B * 8 + spec.K - 1
    nframes = int(rng.choice([1, 1, 2, 4]))
    ebn0 = float(rng.choice([spec.ebn0_db, 0.0, -6.0]))  # noisier input: metrics grow faster, more renormalisations
    _, syms = frames(spec.code, int(rng.integers(1, 1 << 30)), nframes, B, ebn0)
    start = int(rng.integers(0, 1 << 23))
    cuts = sorted(set(int(x) for x in rng.integers(1, steps, size=int(rng.integers(0, 5)))))
    bounds = [0] + cuts + [steps]
    dec = HipViterbi("224", steps, nframes=nframes, variant=VARIANT_HBM_TILED)
    dec.reset(start)
    s3 = syms.reshape(nframes, steps, spec.R)
    for a, b in zip(bounds[:-1], bounds[1:]):
        dec.update(np.ascontiguousarray(s3[:, a:b, :]), nbits=b - a)
    end = int(rng.integers(0, 1 << 23))
    data, _ = dec.chainback(steps, endstate=end)
    f = int(rng.integers(0, nframes))
    o = OracleDecoder(spec.code, spec.poly, steps)
    o.init(start)
    o.update(syms[f], steps)
    want, _ = o.chainback(steps, end)
    ok = np.array_equal(dec.metrics(f), o.metrics()) and np.array_equal(data[f], want)
    rows = o.rows(steps)
    for r in [0, steps - 1] + [int(x) for x in rng.integers(0, steps, size=6)]:
        ok = ok and np.array_equal(dec.decision_rows(f, r, 1), rows[r:r + 1])
    renorms += o.renorms
    o.close()
    dec.close()
    if not ok:
        print("MISMATCH", B, nframes, ebn0, start, bounds, end, f, flush=True)
        sys.exit(1)
    rounds += 1
print(f"soak ok: {rounds} K=24 decodes checked against the oracle, {renorms} renormalisations, {time.time() - t0:.0f} s")
