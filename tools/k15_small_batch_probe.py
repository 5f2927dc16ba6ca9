import sys, torch
sys.path.insert(0, '.')
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, gen_frames_dev
stream = torch.cuda.current_stream().cuda_stream
for name in ("615", "spiral615"):
    spec = C.CODES[name]; B = 256; steps = B*8 + spec.K - 1
    for nframes in (1, 4, 16, 64, 256):
        d_payload = torch.empty(nframes*B, dtype=torch.uint8, device="cuda"); d_syms = torch.empty(nframes*steps*spec.R, dtype=torch.uint8, device="cuda"); d_out = torch.zeros(nframes*B, dtype=torch.uint8, device="cuda")
        gen_frames_dev(spec, 1, 0, nframes, B, C.HARD_AMP_Q16, 0, d_payload, d_syms, stream)
        row = []
        for variant in (1, 2):
            dec = HipViterbi(name, steps, nframes=nframes, variant=variant, stream=stream); dec.enable_timing(True)
            for _ in range(2): dec.reset(); dec.update(d_syms, nbits=steps); dec.chainback(B*8, out=d_out)
            dec.read_timing()
            for _ in range(4): dec.reset(); dec.update(d_syms, nbits=steps); dec.chainback(B*8, out=d_out)
            su, nu, sc, nc = dec.read_timing(); row.append((su/nu, sc/nc)); dec.close()
        print(f"{name} frames {nframes}: LDS update {row[0][0]:.3f} ms cb {row[0][1]:.3f} | REGS update {row[1][0]:.3f} ms cb {row[1][1]:.3f}", flush=True)
