"""Time the chainback kernel alone (history already in HBM, not just written): python tools/chainback_only.py [code] [frames]"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ka9q_viterbi_comparison_amd import codes as C
from ka9q_viterbi_comparison_amd.decoder import HipViterbi, gen_frames_dev, noise_q12

code = sys.argv[1] if len(sys.argv) > 1 else "27"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
spec = C.CODES[code]
nsteps = bits + spec.K - 1
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
d_payload = torch.empty(frames * bits // 8, dtype=torch.uint8, device=dev)
d_syms = torch.empty(frames * nsteps * spec.R, dtype=torch.uint8, device=dev)
d_out = torch.zeros(frames * bits // 8, dtype=torch.uint8, device=dev)
gen_frames_dev(spec, 1, 0, frames, bits // 8, C.SOFT_AMP_Q16, noise_q12(spec.R, C.SOFT_AMP, spec.ebn0_db), d_payload, d_syms, st.cuda_stream)
dec = HipViterbi(code, nsteps, nframes=frames, stream=st.cuda_stream)
dec.reset()
dec.update(d_syms, nbits=nsteps)
junk = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
junk.fill_(1)  # push the freshly written history out of the Infinity Cache
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
for k in range(11):
    ev[k].record(st)
    dec.chainback(bits, out=d_out)
ev[11].record(st)
torch.cuda.synchronize()
ts = [ev[k].elapsed_time(ev[k + 1]) for k in range(11)]
hist = frames * nsteps * ((1 << (spec.K - 1)) // 8)
print(code, frames, "chainback ms:", " ".join(f"{t:.4f}" for t in ts), f"| history {hist/1e9:.3f} GB -> {hist/1e9/min(ts):.0f} GB/s... {hist / (min(ts) * 1e-3) / 1e12:.2f} TB/s")
