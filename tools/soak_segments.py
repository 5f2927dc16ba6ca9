"""Soak test of the segment-parallel chainback: random geometries, bit counts, end states and frame counts for the K=15 codes,
every result compared with the CPU oracle's chainback over the same decision history.  python tools/soak_segments.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from common import frames
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, VARIANT_LDS, VARIANT_REGS
from oracle_lib import OracleDecoder

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(12345)
t0 = time.time()
rounds = checks = rewalks = 0
while time.time() - t0 < budget:
    name = rng.choice(["615", "spiral615"])
    spec = C.CODES[name]
    B = int(rng.integers(20, 140))
    if name == "spiral615" and (B * 8 + 14) % 2:
        B += 1
    nframes = int(rng.integers(1, 5))
    steps = B * 8 + spec.K - 1
    variant = int(rng.choice([VARIANT_REGS, VARIANT_LDS]))
    ebn0 = float(rng.choice([spec.ebn0_db, spec.ebn0_db - 2.0, -3.0]))
    _, syms = frames(spec.code, int(rng.integers(1, 1 << 30)), nframes, B, ebn0)
    start = int(rng.integers(0, 1 << 14))
    dec = HipViterbi(name, steps, nframes=nframes, variant=variant)
    dec.reset(start)
    dec.update(syms)
    oracles = []
    for f in range(nframes):
        o = OracleDecoder(spec.code, spec.poly, steps)
        o.init(start)
        o.update(syms[f], steps)
        oracles.append(o)
    for _ in range(6):
        nbits = int(rng.integers(1, B * 8 + 1))
        end = int(rng.integers(0, 1 << 15))
        seg = int(rng.choice([-1, 8, 16, 64, 104, 256]))
        warm = int(rng.choice([-1, 0, 7, 50, 160, 5000]))
        dec.set_chainback_segments(seg, warm)
        data, _ = dec.chainback(nbits, endstate=end)
        rew, nseg = dec.chainback_rewalked()
        rewalks += rew
        for f in range(nframes):
            want, _ = oracles[f].chainback(nbits, end)
            if not np.array_equal(data[f], want):
                print("MISMATCH", name, variant, B, nframes, nbits, end, seg, warm, f, flush=True)
                sys.exit(1)
            checks += 1
    for o in oracles:
        o.close()
    dec.close()
    rounds += 1
print(f"soak ok: {rounds} decodes, {checks} chainbacks checked against the oracle, {rewalks} segments re-walked, {time.time() - t0:.0f} s")
