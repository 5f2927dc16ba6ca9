"""Soak test of the fused sliding-window decode against the oracle's windowed traceback: random codes, frame counts, payload
sizes (ragged, shorter than the window) and noise levels.  python tools/soak_windowed.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from common import frames
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C
from oracle_lib import OracleDecoder

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(99)
stream = torch.cuda.current_stream().cuda_stream
t0 = time.time()
rounds = checked = 0
while time.time() - t0 < budget:
    name = str(rng.choice(["27", "47", "spiral27", "29", "49", "spiral29", "615", "spiral615"]))
    spec = C.CODES[name]
    big = spec.K == 15
    B = int(rng.integers(1, 40 if big else 90))
    nbits = int(rng.integers(max(1, B * 8 - 7), B * 8 + 1))
    steps = nbits + spec.K - 1
    nframes = int(rng.choice([1, 2, 7] if big else [1, 17, 64, 65, 200, 1100]))
    ebn0 = rng.choice([None, spec.ebn0_db, spec.ebn0_db - 3.0])
    _, syms = frames(spec.code, int(rng.integers(1, 1 << 30)), nframes, B, None if ebn0 is None else float(ebn0))
    syms = np.ascontiguousarray(syms[:, :steps * spec.R])
    dec = HipViterbi(name, steps, nframes=nframes, stream=stream)
    depth, block = dec.window
    stride = (nbits + 7) // 8
    out = torch.full((nframes * stride,), 0xEE, dtype=torch.uint8, device="cuda")
    dec.decode_windowed(torch.from_numpy(syms).cuda(), nbits, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(nframes, stride)
    for f in sorted({0, nframes - 1, int(rng.integers(0, nframes)), int(rng.integers(0, nframes))}):
        o = OracleDecoder(spec.code, spec.poly, steps)
        o.update(syms[f], steps)
        ref = o.chainback_windowed(nbits, depth, block)
        o.close()
        if not np.array_equal(got[f], ref):
            print("MISMATCH", name, nframes, B, nbits, ebn0, f, flush=True)
            sys.exit(1)
        checked += 1
    dec.close()
    rounds += 1
print(f"soak ok: {rounds} fused windowed decodes, {checked} frames checked against the windowed oracle, {time.time() - t0:.0f} s")
