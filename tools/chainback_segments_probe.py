"""Segment-parallel chainback of the large codes: time and re-walk counts per geometry.
python tools/chainback_segments_probe.py [code=224] [frames=1] [bits=2048] [ebn0 dB, default the code's]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ka9q_viterbi_comparison_amd import codes as C
from ka9q_viterbi_comparison_amd.decoder import HipViterbi, gen_frames_dev, noise_q12

code = sys.argv[1] if len(sys.argv) > 1 else "224"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
spec = C.CODES[code]
ebn0 = float(sys.argv[4]) if len(sys.argv) > 4 else spec.ebn0_db
nsteps = bits + spec.K - 1
nb = nsteps if spec.K == 24 else bits  # chainback_viterbi224 walks the tail rows too when given nbits + K - 1
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
d_payload = torch.empty(frames * bits // 8, dtype=torch.uint8, device=dev)
d_syms = torch.empty(frames * nsteps * spec.R, dtype=torch.uint8, device=dev)
d_out = torch.zeros(frames * ((nb + 7) // 8), dtype=torch.uint8, device=dev)
gen_frames_dev(spec, 1, 0, frames, bits // 8, C.SOFT_AMP_Q16, noise_q12(spec.R, C.SOFT_AMP, ebn0), d_payload, d_syms, st.cuda_stream)
dec = HipViterbi(code, nsteps, nframes=frames, stream=st.cuda_stream)
dec.reset()
dec.update(d_syms, nbits=nsteps)
torch.cuda.synchronize()
ref = None
print(f"{code}: {frames} frame(s) x {nb} decoded bits, Eb/N0 {ebn0} dB")
for seg, warm in [(0, 0), (-1, -1), (64, 128), (64, 192), (64, 256), (64, 320), (128, 128), (128, 192), (128, 256), (128, 320), (128, 448), (256, 192),
                  (256, 256), (256, 320), (512, 320), (32, 256), (32, 320), (64, 0)]:
    dec.set_chainback_segments(seg, warm)
    ts = []
    for k in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        dec.chainback(nb, out=d_out)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    rew, nseg = dec.chainback_rewalked()
    got = d_out.cpu()
    if ref is None:
        ref = got
    same = bool(torch.equal(got, ref))
    t = min(ts)
    print(f"seg {seg:4d} warm-up {warm:4d}: {t*1e3:8.1f} us  {frames * nb / t / 1e3:8.2f} Mbit/s  segments/frame {nseg:3d}  re-walked {rew:3d}  same bytes as one walk: {same}")
errs = (torch.bitwise_xor(ref[: frames * ((nb + 7) // 8)].view(frames, -1)[:, : bits // 8], d_payload.cpu().view(frames, -1)).to(torch.int32)).sum().item()
print("payload byte xor sum (0 = error free):", errs)
