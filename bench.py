#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Viterbi decode hot path on MI355X.

One "step" = one full decode pass of the hot path over one batch of synthetic frames already resident in
HBM: init (reset metrics) + ACS update + chainback, exactly the three calls the reference times per sample
(src/main.cpp:264-278).  Default workload = BASELINE.json configs[1]: K=7 r=1/2 (viterbi27), 65 536 frames x
2048 info bits per GPU, AWGN soft symbols.  Frames are sharded across ranks (no data-path collective;
torch.distributed is used only for the barrier and the max-over-ranks timing) -> "scaling": "weak".

metric / value: coded symbols (incl. tail, the reference's definition scripts/tabulate_data.py:33) decoded per
second over the whole job, Msymbols/s.  The JSON line also carries the ACS-update-only and chainback-only
rates from HIP events, the HBM roofline of the dominant (ACS) kernel, and a CPU baseline timed on this host.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import concurrent.futures as cf
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ka9q_viterbi_comparison_amd import HipViterbi, codes as C, count_bit_errors_dev, gen_frames_dev, noise_q12  # noqa: E402
from ka9q_viterbi_comparison_amd.decoder import gen_frames_host  # noqa: E402
from ka9q_viterbi_comparison_amd.sharding import barrier as shard_barrier, max_over_ranks, sum_over_ranks, weak_range  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_frame_step(spec):
    """R symbol bytes read + 2^(K-1)/8 decision bytes written per frame-step (metrics on chip);
    K=24: 2*16 MiB metric read+write + 1 MiB row + 2."""
    if spec.K == 24:
        return 2 * (1 << 23) * 2 + (1 << 23) // 8 + 2
    return spec.R + (1 << (spec.K - 1)) // 8


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(spec, payload_bits, budget_s=12.0):
    """Times the CPU decoder on this host on a bounded sample of the same workload: the genuine reference objects
    (oracle/_ref/libref.so, built from /root/reference in the build container) when present -> kind "reference";
    else the plain-C restatement -> kind "port".  One frame per call, reset+update+chainback per frame
    (src/main.cpp:257-280), timed inside C (oracle/ref_shim.cpp ref_bench_loop): first 1 thread, then one decoder
    per host core in parallel Python threads (ctypes releases the GIL for the whole loop)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    use_ref = ol.have_ref()
    payload_bytes = payload_bits // 8
    steps = payload_bits + spec.K - 1
    nsample = 64 if spec.K < 15 else (8 if spec.K == 15 else 1)
    nq = noise_q12(spec.R, C.SOFT_AMP, spec.ebn0_db)
    _, syms = gen_frames_host(spec, 0xC0FFEE, 0, nsample, payload_bytes, C.SOFT_AMP_Q16, nq)
    syms = np.ascontiguousarray(syms)
    w32 = spec.code == C.KA9Q615

    def make():
        return ol.RefDecoder(spec.code, spec.poly, steps, w32=w32) if use_ref else ol.OracleDecoder(spec.code, spec.poly, steps)

    def work(dec, seconds):
        el = ctypes.c_double(0.0)
        sp = syms.ctypes.data_as(ctypes.c_void_p)
        if use_ref:
            n = dec.lib.ref_bench_loop(spec.code, dec.h, sp, nsample, syms.shape[1], steps, payload_bits, seconds, ctypes.byref(el))
        else:
            n = dec.lib.vo_bench_loop(dec.h, sp, nsample, syms.shape[1], steps, payload_bits, seconds, ctypes.byref(el))
        return n, el.value

    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box gives one GPU a 16-core share
    if spec.K == 24:
        cores = min(cores, 4)      # each K=24 handle holds 32 MiB of metrics + 1 MiB per step
    decs = [make() for _ in range(cores)]  # created serially: the reference's table init is not thread-safe
    n1, t1 = work(decs[0], budget_s * 0.4)
    single = n1 * steps * spec.R / t1 / 1e6
    with cf.ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        res = list(ex.map(lambda d: work(d, budget_s * 0.6), decs))
        wall = time.perf_counter() - t0
    nall = sum(r[0] for r in res)
    multi = nall * steps * spec.R / wall / 1e6
    for d in decs:
        d.close()
    return {
        "value": round(multi, 6),
        "unit": "Msymbols/s",
        "cores": cores,
        "kind": "reference" if use_ref else "port",
        "single_thread_value": round(single, 6),
        "sample": f"{n1} (1 thread) + {nall} ({cores} threads) one-frame decodes (reset+update+chainback) of K={spec.K} "
                  f"r=1/{spec.R} x {payload_bits} bits, AWGN Eb/N0={spec.ebn0_db} dB, {budget_s:.0f} s budget",
    }


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--code", default="27", choices=sorted(C.CODES))
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (default: BASELINE config for the code)")
    ap.add_argument("--payload-bits", type=int, default=None)
    ap.add_argument("--ebn0", type=float, default=None, help="Eb/N0 in dB; 'hard' symbols if --hard")
    ap.add_argument("--hard", action="store_true", help="reference-style noise-free 0/255 symbols")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--chunk-frames", type=int, default=None, help="frames per handle (default: all, halved until the history fits)")
    ap.add_argument("--hbm-budget-gb", type=float, default=200.0, help="decision-history budget per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="(default for K <= 15) also report value_pipelined: the same steps double-buffered over two handles")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the value_pipelined extra")
    ap.add_argument("--pipeline-priority", action="store_true", help="value_pipelined with the priority-stream scheme instead of two unordered streams")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the decode path has no CPU fallback)"
    # One rank per GPU.  VITERBI_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks: the ranks then
    # share the visible devices and the (scalar) reductions run on CPU tensors; RCCL refuses two ranks on one device.
    backend = os.environ.get("VITERBI_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    spec = C.CODES[args.code]
    # BASELINE.json configs: K=7 x 65536 frames, K=15 x 4096 frames, K=24 single long frame; K=9 sized in between
    defaults = {7: (65536, 2048), 9: (32768, 2048), 15: (4096, 2048), 24: (1, 2048)}
    frames = args.frames or defaults[spec.K][0]
    payload_bits = args.payload_bits or defaults[spec.K][1]
    payload_bytes = payload_bits // 8
    nsteps = payload_bits + spec.K - 1
    ebn0 = spec.ebn0_db if args.ebn0 is None else args.ebn0

    stream = torch.cuda.current_stream()
    dev = torch.device("cuda", dev_index)
    rdev = dev if backend == "nccl" else None  # where the scalar reductions live
    d_payload = torch.empty(frames * payload_bytes, dtype=torch.uint8, device=dev)
    d_syms = torch.empty(frames * nsteps * spec.R, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(frames * payload_bytes, dtype=torch.uint8, device=dev)
    if args.hard:
        amp_q16, nq = C.HARD_AMP_Q16, 0
    else:
        amp_q16, nq = C.SOFT_AMP_Q16, noise_q12(spec.R, C.SOFT_AMP, ebn0)
    # synthetic frames generated on the device, distinct per rank (frame ids rank*frames ...)
    frame_lo, _ = weak_range(frames, rank)
    gen_frames_dev(spec, 0x5EED, frame_lo, frames, payload_bytes, amp_q16, nq, d_payload, d_syms, stream.cuda_stream)
    # The decision history (N/8 bytes per frame-step) must fit in HBM next to the symbols: large K=15 batches are
    # decoded in chunks of frames that reuse one handle (SURVEY.md §7 "K=15 capacity"); a step still covers all frames.
    dec_bytes_per_frame = (nsteps + spec.K) * ((1 << (spec.K - 1)) // 8)
    budget = int(args.hbm_budget_gb * 1e9)
    chunk = frames if args.chunk_frames is None else args.chunk_frames
    while chunk > 64 and chunk * dec_bytes_per_frame > budget:
        chunk = (chunk + 1) // 2
    nchunks = (frames + chunk - 1) // chunk
    assert frames % chunk == 0, "frames must be a multiple of the chunk size"
    dec = HipViterbi(args.code, nsteps, nframes=chunk, variant=args.variant, stream=stream.cuda_stream)
    # K=24's own chainback call convention needs nbits+K-1 to decode correctly; the harness call
    # (nbits = payload bits) is what is timed, as in the reference.
    cb_bits = payload_bits
    sym_chunk, out_chunk = chunk * nsteps * spec.R, chunk * payload_bytes

    def one_pass(ev=None):
        for c in range(nchunks):
            first, last = c == 0, c == nchunks - 1
            dec.reset()
            if ev and first:
                ev[0].record(stream)
            dec.update(d_syms[c * sym_chunk:(c + 1) * sym_chunk], nbits=nsteps)
            if ev and nchunks == 1:
                ev[1].record(stream)
            dec.chainback(cb_bits, out=d_out[c * out_chunk:(c + 1) * out_chunk])
            if ev and last:
                if nchunks > 1:
                    ev[1].record(stream)  # chunked: per-kernel split not available, report the whole pass as update
                ev[2].record(stream)

    def barrier():
        shard_barrier(dev)

    for _ in range(args.warmup):
        one_pass()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_pass(events[i])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, rdev)

    # Extra (never `value`): the same K steps double-buffered over two handles, so that the HBM-bound chainback of one
    # batch overlaps the VALU-bound update of the next (steady-state serving throughput; +20 % for K=7 on one MI355X).
    pipelined = None
    if not args.no_pipeline and spec.K <= 15 and nchunks == 1 and 2 * chunk * dec_bytes_per_frame <= budget:
        # Two handles (two decision-history buffers).  All ACS updates go, in order, to one high-priority stream, so
        # the next update is dispatched first and spreads evenly over the CUs (exactly one wave per SIMD: a
        # chainback workgroup that got there first would make two update workgroups share a CU and double their
        # time); every chainback goes to a normal-priority stream behind an event and fills in beside the next update.
        hi = torch.cuda.Stream(device=dev, priority=-1)
        lo = torch.cuda.Stream(device=dev, priority=0)
        dec2 = HipViterbi(args.code, nsteps, nframes=frames, variant=args.variant, stream=hi.cuda_stream)
        d_out2 = torch.zeros_like(d_out)
        lanes = [(dec, d_out), (dec2, d_out2)]
        cb_done = [None, None]
        torch.cuda.synchronize()

        free_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

        def pass_on(k):
            dk, ok = lanes[k & 1]
            if not args.pipeline_priority:
                # default: two independent in-order streams, no cross-stream ordering at all -- the two decodes drift apart
                # and one's chainback / launch ramps fill the other's update (K=7: +20 %; the priority scheme below: +4 %)
                dk.set_stream(free_streams[k & 1].cuda_stream)
                dk.reset()
                dk.update(d_syms, nbits=nsteps)
                dk.chainback(cb_bits, out=ok)
                return
            if cb_done[k & 1] is not None:
                hi.wait_event(cb_done[k & 1])  # this handle's previous history has been walked
            dk.set_stream(hi.cuda_stream)
            dk.reset()
            dk.update(d_syms, nbits=nsteps)
            upd_done = hi.record_event()
            lo.wait_event(upd_done)
            dk.set_stream(lo.cuda_stream)
            dk.chainback(cb_bits, out=ok)
            cb_done[k & 1] = lo.record_event()

        for k in range(2):
            pass_on(k)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            pass_on(k)
        barrier()
        pipelined = max_over_ranks(time.perf_counter() - t0, rdev)
        assert torch.equal(d_out, d_out2)
        dec.set_stream(stream.cuda_stream)
        dec2.close()

    upd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    cb_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
    # correctness guard outside the timed region: decoded bytes vs transmitted payload (BER over the batch)
    nerr = count_bit_errors_dev(d_out, d_payload, frames * payload_bytes, stream.cuda_stream) if spec.K != 24 else -1
    nerr = sum_over_ranks(nerr, rdev) if spec.K != 24 else -1

    if rank == 0:
        total_syms = frames * nsteps * spec.R * n_gpus
        ms_per_step = elapsed / args.steps * 1e3
        value = total_syms * args.steps / elapsed / 1e6
        abytes = algorithmic_bytes_per_frame_step(spec) * frames * nsteps  # per launch (one rank)
        achieved = abytes / (upd_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.code}.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        valu = None
        vpath = os.path.join(ROOT, "profiles", f"valu_{args.code}.json")
        if os.path.exists(vpath):
            try:
                v = json.load(open(vpath))
                # instructions per launch come from the committed PMC pass; the time is this run's
                valu = {"wave_instr_per_launch": v["valu_wave_instr_per_launch"],
                        "achieved_ginstr_per_s": round(v["valu_wave_instr_per_launch"] / (upd_ms * 1e-3) / 1e9, 2),
                        "issue_ceiling_ginstr_per_s": round(v["issue_ceiling_ginstr_per_s"], 2),
                        "frac": round(v["valu_wave_instr_per_launch"] / (upd_ms * 1e-3) / 1e9 / v["issue_ceiling_ginstr_per_s"], 4)}
            except Exception:
                valu = None
        out = {
            "metric": f"Msymbols/s decoded, K={spec.K} r=1/{spec.R} (init + ACS update + chainback)",
            "value": round(value, 6),
            "unit": "Msymbols/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if spec.family != "ka9q-i16-sat" else "i16",
            "data": "synthetic" + (" hard 0/255 symbols" if args.hard else f" AWGN Eb/N0={ebn0} dB, amplitude {C.SOFT_AMP}") + ", generated on device",
            "config": {"workload": f"viterbi{spec.name}: K={spec.K} r=1/{spec.R}, {frames} frames/GPU x {payload_bits} info bits "
                                   f"({nsteps} trellis steps, {nsteps * spec.R} symbols/frame)",
                       "frames_per_gpu": frames, "chunk_frames": chunk, "payload_bits": payload_bits, "variant": dec.variant,
                       "parallelism": f"frame-shard x{n_gpus}, no collective"},
            "update_msym_s": round(frames * nsteps * spec.R * n_gpus / (upd_ms * 1e-3) / 1e6, 6),
            "chainback_mbit_s": round(frames * cb_bits * n_gpus / (cb_ms * 1e-3) / 1e6, 3),
            "update_ms": round(upd_ms, 4),
            "chainback_ms": round(cb_ms, 4),
            "bit_errors": int(nerr),
            "value_pipelined": (round(total_syms * args.steps / pipelined / 1e6, 3) if pipelined else None),
            "payload_bits_total": frames * payload_bits * n_gpus,
            "roofline": {"bound": "hbm", "kernel": "acs_update", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes},
            # the binding limit of the K<=15 ACS kernels: packed-integer VALU issue, not HBM
            "valu_issue": valu,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, payload_bits, args.cpu_budget)
        print(json.dumps(out), flush=True)
    dec.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
