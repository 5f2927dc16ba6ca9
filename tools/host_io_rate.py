#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer API (vhip_update / vhip_chainback copy symbols in and decoded bytes out
around the kernels) for BASELINE config 2.  Never the headline value: bench.py times device-resident input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, gen_frames_host, noise_q12

spec = C.CODES["27"]
frames, bits = 65536, 2048
steps = bits + spec.K - 1
_, one = gen_frames_host(spec, 1, 0, 256, bits // 8, C.SOFT_AMP_Q16, noise_q12(2, 64.0, 4.0))
syms = np.ascontiguousarray(np.tile(one, (frames // 256, 1)))
dec = HipViterbi("27", steps, nframes=frames)
for rep in range(3):
    t0 = time.perf_counter()
    dec.reset()
    dec.update(syms, nbits=steps)
    t1 = time.perf_counter()
    data, _ = dec.chainback(bits)
    t2 = time.perf_counter()
    print(f"host-pointer API: update {1e3*(t1-t0):.2f} ms ({syms.nbytes/1e6:.0f} MB in), chainback {1e3*(t2-t1):.2f} ms "
          f"({data.nbytes/1e6:.1f} MB out): {frames*steps*2/(t2-t0)/1e6:.0f} Msym/s PCIe-inclusive", flush=True)
