"""Soak test of the double-buffered handles (the path bench.py measures): random codes, pipeline depths, frame counts and
lengths; several decodes of DIFFERENT symbol batches are issued back to back through the device-pointer API with no host
synchronisation, then every output is compared with the oracle on sampled frames and with a strictly serial handle on all
frames.  python tools/soak_pipeline.py [seconds]"""
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
rng = np.random.default_rng(4711)
stream = torch.cuda.current_stream().cuda_stream
t0 = time.time()
rounds = checked = 0
while time.time() - t0 < budget:
    name = str(rng.choice(["27", "29", "47", "49", "spiral27", "spiral29", "615", "spiral615"]))
    spec = C.CODES[name]
    big = spec.K == 15
    B = int(rng.integers(2, 30 if big else 60))
    steps = B * 8 + spec.K - 1
    if name.startswith("spiral") or name in ("47", "49"):
        steps -= steps % 2  # the spiral decoders drop an odd last step
    nframes = int(rng.choice([1, 2, 5, 40] if big else [1, 64, 65, 300, 2000, 4097]))
    depth = int(rng.choice([2, 2, 3]))
    ndec = int(rng.integers(2, 7))
    dec = HipViterbi(name, steps, nframes=nframes, stream=stream, pipeline_depth=depth)
    ser = HipViterbi(name, steps, nframes=nframes, stream=stream)
    nbits = int(rng.integers(1, B * 8 + 1))
    stride = (nbits + 7) // 8
    batches, outs, refs = [], [], []
    for k in range(ndec):
        _, syms = frames(spec.code, int(rng.integers(1, 1 << 30)), nframes, B, float(rng.choice([spec.ebn0_db, -2.0])))
        syms = np.ascontiguousarray(syms[:, :steps * spec.R])
        batches.append((syms, torch.from_numpy(syms).cuda()))
        outs.append(torch.zeros(nframes * stride, dtype=torch.uint8, device="cuda"))
        refs.append(torch.zeros(nframes * stride, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    for k in range(ndec):
        dec.reset()
        dec.update(batches[k][1], nbits=steps)
        dec.chainback(nbits, out=outs[k])
    dec.join()
    for k in range(ndec):
        ser.reset()
        ser.update(batches[k][1], nbits=steps)
        ser.chainback(nbits, out=refs[k])
    torch.cuda.synchronize()
    for k in range(ndec):
        got = outs[k].cpu().numpy().reshape(nframes, stride)
        if not np.array_equal(got, refs[k].cpu().numpy().reshape(nframes, stride)):
            print("MISMATCH vs serial handle", name, nframes, B, depth, ndec, nbits, k, flush=True)
            sys.exit(1)
        for f in sorted({0, nframes - 1, int(rng.integers(0, nframes))}):
            o = OracleDecoder(spec.code, spec.poly, steps)
            o.update(batches[k][0][f], steps)
            want, _ = o.chainback(nbits, 0)
            o.close()
            if not np.array_equal(got[f], want):
                print("MISMATCH vs oracle", name, nframes, B, depth, ndec, nbits, k, f, flush=True)
                sys.exit(1)
            checked += 1
    dec.close()
    ser.close()
    rounds += 1
print(f"soak ok: {rounds} pipelined handles, {checked} frames checked against the oracle, every decode against a serial handle, {time.time() - t0:.0f} s")
