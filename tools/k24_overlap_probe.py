#!/usr/bin/env python3
"""Probe: do two K=24 decodes on two streams overlap (i.e. is a single decode leaving the chip under-occupied)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, gen_frames_dev, noise_q12

spec = C.CODES["224"]
bits = 2048
steps = bits + spec.K - 1
dev = torch.device("cuda", 0)
s = [torch.cuda.Stream(), torch.cuda.Stream()]
syms = torch.empty(steps * 2, dtype=torch.uint8, device=dev)
pay = torch.empty(bits // 8, dtype=torch.uint8, device=dev)
gen_frames_dev(spec, 1, 0, 1, bits // 8, C.SOFT_AMP_Q16, noise_q12(2, 64.0, 4.0), pay, syms, 0)
torch.cuda.synchronize()
decs = [HipViterbi("224", steps, stream=x.cuda_stream) for x in s]

import threading
def run(d):
    d.reset(); d.update(syms, nbits=steps); d.sync()

for mode in ("serial", "threads"):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "serial":
            run(decs[0]); run(decs[1])
        else:
            th = [threading.Thread(target=run, args=(d,)) for d in decs]
            [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        print(mode, f"{(time.perf_counter()-t0)*1e3:.2f} ms for 2 decodes", flush=True)
