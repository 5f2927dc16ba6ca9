#!/usr/bin/env python3
"""Update + chainback time of small batches: LDS kernel (one workgroup per frame), register kernels (frames across lanes) and the
wave-per-frame kernels (acs_wave.hip): where should the automatic variant switch?
python tools/small_batch_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, gen_frames_dev  # noqa: E402

stream = torch.cuda.current_stream().cuda_stream
for name in ("27", "47", "29", "49"):
    spec = C.CODES[name]
    B = 256
    steps = B * 8 + spec.K - 1
    for nframes in (1, 16, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        d_payload = torch.empty(nframes * B, dtype=torch.uint8, device="cuda")
        d_syms = torch.empty(nframes * steps * spec.R, dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
        gen_frames_dev(spec, 1, 0, nframes, B, C.HARD_AMP_Q16, 0, d_payload, d_syms, stream)
        row = []
        for variant in (1, 2, 6):
            dec = HipViterbi(name, steps, nframes=nframes, variant=variant, stream=stream)
            dec.enable_timing(True)
            for _ in range(3):
                dec.reset(); dec.update(d_syms, nbits=steps); dec.chainback(B * 8, out=d_out)
            dec.read_timing()
            for _ in range(5):
                dec.reset(); dec.update(d_syms, nbits=steps); dec.chainback(B * 8, out=d_out)
            su, nu, sc, nc = dec.read_timing()
            row.append((su / nu, sc / nc))
            dec.close()
        line = f"{name} frames {nframes:6d}: LDS update {row[0][0]:8.3f} ms chainback {row[0][1]:7.3f} ms | REGS update {row[1][0]:8.3f} ms chainback {row[1][1]:7.3f} ms"
        if len(row) > 2:
            line += f" | WAVE update {row[2][0]:8.3f} ms chainback {row[2][1]:7.3f} ms"
        print(line, flush=True)
