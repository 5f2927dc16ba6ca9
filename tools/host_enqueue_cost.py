import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from ka9q_viterbi_comparison_amd import codes as C
from ka9q_viterbi_comparison_amd.decoder import HipViterbi, gen_frames_dev, noise_q12
spec = C.CODES["27"]; frames = 65536; bits = 2048; nsteps = bits + 6
dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
d_payload = torch.empty(frames * bits // 8, dtype=torch.uint8, device=dev)
d_syms = torch.empty(frames * nsteps * 2, dtype=torch.uint8, device=dev)
d_out = torch.zeros(frames * bits // 8, dtype=torch.uint8, device=dev)
gen_frames_dev(spec, 1, 0, frames, bits // 8, C.SOFT_AMP_Q16, noise_q12(2, C.SOFT_AMP, 4.0), d_payload, d_syms, st.cuda_stream)
dec = HipViterbi("27", nsteps, nframes=frames, stream=st.cuda_stream)
for _ in range(3):
    dec.reset(); dec.update(d_syms, nbits=nsteps); dec.chainback(bits, out=d_out)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        dec.reset(); dec.update(d_syms, nbits=nsteps); dec.chainback(bits, out=d_out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue 20 steps: {(t1-t0)*1e3:.3f} ms host, total {(t2-t0)*1e3:.3f} ms")
