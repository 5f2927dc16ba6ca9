#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Viterbi decode hot path on MI355X.

One "step" = one full decode pass of the hot path over one batch of synthetic frames already resident in
HBM: init (reset metrics) + ACS update + chainback, exactly the three calls the reference times per sample
(src/main.cpp:264-278).  Default workload = BASELINE.json configs[1]: K=7 r=1/2 (viterbi27), 65 536 frames x
2048 info bits per GPU, AWGN soft symbols.  Frames are sharded across ranks (no data-path collective;
torch.distributed is used only for the barrier and the max-over-ranks timing) -> "scaling": "weak".

The steps are issued back to back on ONE handle.  For K <= 15 the handle is double-buffered inside the library
(vhip_set_pipeline_depth(2): two decision-history buffers and two internal streams), so the HBM-bound chainback of
step i overlaps the VALU-bound update of step i+1; --pipeline-depth 1 gives the strictly serial schedule.

metric / value: coded symbols (incl. tail, the reference's definition scripts/tabulate_data.py:33) decoded per
second over the whole job, Msymbols/s.  The JSON line also carries the ACS-update-only and chainback-only
rates from HIP events recorded by the library on the streams its kernels run on (vhip_enable_timing), the HBM
roofline of the dominant (ACS) kernel, and a CPU baseline timed on this host.

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
from ka9q_viterbi_comparison_amd import VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_LDS, VARIANT_REGS  # noqa: E402
from ka9q_viterbi_comparison_amd.decoder import gen_frames_host  # noqa: E402
from ka9q_viterbi_comparison_amd.sharding import run_sharded_job  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_hash import kernel_family, kernel_source_hash  # noqa: E402


def algorithmic_bytes_per_frame_step(spec):
    """R symbol bytes read + 2^(K-1)/8 decision bytes written per frame-step (metrics on chip);
    K=24: 2*16 MiB metric read+write + 1 MiB row + 2."""
    if spec.K == 24:
        return 2 * (1 << 23) * 2 + (1 << 23) // 8 + 2
    return spec.R + (1 << (spec.K - 1)) // 8


def moved_bytes_per_frame_step(spec, variant, k24_passes_per_period=None):
    """Bytes the ACS kernels really move per frame-step.  K <= 15: the algorithmic figure (metrics stay on chip; the PMC
    traffic confirms 1.02x).  K=24: the multi-step passes move the 16 MiB metric array twice per PASS, not per step:
    passes_per_23_steps x 32 MiB + 23 x (1 MiB row + 2 symbol bytes), per 23 steps."""
    if spec.K != 24:
        return float(algorithmic_bytes_per_frame_step(spec))
    if variant == VARIANT_HBM:
        return float(algorithmic_bytes_per_frame_step(spec))
    passes = k24_passes_per_period or 5
    return (passes * 2 * (1 << 24) + 23 * ((1 << 20) + 2)) / 23.0


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
        cores_available = len(os.sched_getaffinity(0))
    except Exception:
        cores_available = os.cpu_count() or 1
    cores = max(1, min(cores_available, 16))  # the GPU box gives one GPU a 16-core share
    if spec.K == 24:
        cores = min(cores, 4)      # each K=24 handle holds 32 MiB of metrics + 1 MiB per step
    flags_file = os.path.join(ROOT, "oracle", "_ref", "build_flags.txt")
    if use_ref:
        build = open(flags_file).read().strip() if os.path.exists(flags_file) else "g++ -std=c++17 -O3 -msse4.1 (oracle/Makefile REF_CXXFLAGS)"
        build += "; the reference's own presets add -march=native -ffast-math (CMakePresets.json:36-56), not used: the objects must run on any host"
    else:
        build = "gcc -O2 -std=gnu99 (oracle/Makefile CFLAGS), scalar C"
    decs = [make() for _ in range(cores)]  # created serially: the reference's table init is not thread-safe
    n1, t1 = work(decs[0], budget_s * 0.4)
    single = n1 * steps * spec.R / t1 / 1e6
    # one decoder per thread, each thread pinned to its own CPU of the allowed set (unpinned, the all-core figure of the same
    # 16 threads swung between 2.6 and 4.2 Gsym/s from run to run as the scheduler moved them around a 256-CPU host)
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        cpus = []

    def pinned(k):
        if len(cpus) >= cores:
            try:
                os.sched_setaffinity(0, {cpus[k]})  # pid 0 = the calling thread
            except Exception:
                pass
        return work(decs[k], budget_s * 0.6)

    with cf.ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        res = list(ex.map(pinned, range(cores)))
        wall = time.perf_counter() - t0
    nall = sum(r[0] for r in res)
    multi = nall * steps * spec.R / wall / 1e6
    for d in decs:
        d.close()
    return {
        "value": round(multi, 6),
        "unit": "Msymbols/s",
        "cores": cores,
        "cores_available": cores_available,
        "pinned": len(cpus) >= cores,
        "kind": "reference" if use_ref else "port",
        "build": build,
        "single_thread_value": round(single, 6),
        "sample": f"{n1} (1 thread) + {nall} ({cores} threads) one-frame decodes (reset+update+chainback) of K={spec.K} "
                  f"r=1/{spec.R} x {payload_bits} bits, AWGN Eb/N0={spec.ebn0_db} dB, {budget_s:.0f} s budget",
    }


# ------------------------------------------------------------------------------------------------ the HIP shard
class HipShard:
    """This rank's frames, resident in HBM, and the handle that decodes them (the object run_sharded_job drives)."""

    def __init__(self, args, spec, dev, frame_lo, frames):
        self.args, self.spec, self.dev, self.frames = args, spec, dev, frames
        self.stream = torch.cuda.current_stream()
        defaults = {7: 2048, 9: 2048, 15: 2048, 24: 2048}
        self.payload_bits = args.payload_bits or defaults[spec.K]
        self.payload_bytes = self.payload_bits // 8
        self.nsteps = self.payload_bits + spec.K - 1
        self.ebn0 = spec.ebn0_db if args.ebn0 is None else args.ebn0
        R = spec.R
        self.d_payload = torch.empty(frames * self.payload_bytes, dtype=torch.uint8, device=dev)
        self.d_syms = torch.empty(frames * self.nsteps * R, dtype=torch.uint8, device=dev)
        if args.hard:
            amp_q16, nq = C.HARD_AMP_Q16, 0
        else:
            amp_q16, nq = C.SOFT_AMP_Q16, noise_q12(R, C.SOFT_AMP, self.ebn0)
        # synthetic frames generated on the device, distinct per rank (global frame ids frame_lo ...)
        gen_frames_dev(spec, 0x5EED, frame_lo, frames, self.payload_bytes, amp_q16, nq, self.d_payload, self.d_syms, self.stream.cuda_stream)
        # The decision history (N/8 bytes per frame-step, times the pipeline depth) must fit in HBM next to the symbols:
        # large K=15 batches are decoded in chunks of frames that reuse one handle (SURVEY.md §7 "K=15 capacity"); a
        # step still covers all frames.
        self.windowed = bool(getattr(args, "windowed", False))
        self.depth = 1 if (spec.K == 24 or self.windowed) else max(1, args.pipeline_depth)
        dec_bytes_per_frame = (64 if self.windowed else self.nsteps + spec.K) * ((1 << (spec.K - 1)) // 8) * self.depth
        budget = int(args.hbm_budget_gb * 1e9)
        chunk = frames if args.chunk_frames is None else min(args.chunk_frames, frames)
        if args.chunk_frames is None and spec.K == 7 and not self.windowed and frames > 65536 and frames % 65536 == 0:
            # the K=7 register kernel is tuned for exactly one wave per SIMD (1024 waves x 64 frames): bigger
            # shards go through the double-buffered handle in chunks of 65536 frames (131072 frames: 298 vs 278 Gsym/s)
            chunk = 65536
        while chunk > 64 and chunk * dec_bytes_per_frame > budget:
            chunk = (chunk + 1) // 2
        assert frames % chunk == 0, "frames must be a multiple of the chunk size"
        self.chunk, self.nchunks = chunk, frames // chunk
        # the fused decode keeps its decisions in an on-chip / cache-resident ring: such a handle never allocates a history
        self.dec = HipViterbi(args.code, self.nsteps, nframes=chunk, variant=args.variant, stream=self.stream.cuda_stream,
                              pipeline_depth=self.depth)
        self.dec.enable_timing(True)
        # K=24's own chainback call convention needs nbits+K-1 to decode correctly; the harness call
        # (nbits = payload bits) is what is timed, as in the reference.
        self.cb_bits = self.payload_bits
        # consecutive decodes of a pipelined handle need different output buffers: passes alternate between two
        nout = 2 if self.depth > 1 else 1
        self.d_out = [torch.zeros(frames * self.payload_bytes, dtype=torch.uint8, device=dev) for _ in range(nout)]
        self.timed_from = None

    def one_pass(self, i):
        sc, oc = self.chunk * self.nsteps * self.spec.R, self.chunk * self.payload_bytes
        out = self.d_out[i % len(self.d_out)]
        for c in range(self.nchunks):
            if self.windowed:
                self.dec.decode_windowed(self.d_syms[c * sc:(c + 1) * sc], self.payload_bits, out[c * oc:(c + 1) * oc])
                continue
            self.dec.reset()
            self.dec.update(self.d_syms[c * sc:(c + 1) * sc], nbits=self.nsteps)
            self.dec.chainback(self.cb_bits, out=out[c * oc:(c + 1) * oc])

    def drain(self):
        self.dec.join()
        if self.timed_from is None:
            self.dec.read_timing()  # discard the warm-up launches: what is read in stats() is the timed region only
            self.timed_from = True

    def stats(self):
        su, nu, sc, nc = self.dec.read_timing()
        passes = max(1, nu // self.nchunks)
        nerr = -1
        if self.spec.K == 24 and self.nchunks == 1:
            # the timed chainback is the harness's own call (nbits = payload bits), which does not decode K=24 correctly
            # (SURVEY.md §0.4); the bit errors come from one extra nbits+K-1 call on the history of the last pass
            lb = (self.nsteps + 7) // 8
            d_long = torch.zeros(self.frames * lb, dtype=torch.uint8, device=self.dev)
            self.dec.chainback(self.nsteps, out=d_long)
            self.dec.sync()
            self.dec.read_timing()
            nerr = 0
            for f in range(self.frames):
                nerr += count_bit_errors_dev(d_long[f * lb:f * lb + self.payload_bytes],
                                             self.d_payload[f * self.payload_bytes:(f + 1) * self.payload_bytes], self.payload_bytes, self.stream.cuda_stream)
        if self.spec.K != 24:
            errs = [count_bit_errors_dev(o, self.d_payload, self.frames * self.payload_bytes, self.stream.cuda_stream) for o in self.d_out]
            assert len(set(errs)) == 1 and all(torch.equal(self.d_out[0], o) for o in self.d_out), "passes disagree"
            nerr = errs[0]
        # After the timed region: the same kernels one at a time (nothing else on the device), for the per-kernel roofline --
        # in the timed region the update of step i+1 shares the chip with the chainback of step i.
        alone_u = alone_c = 0.0
        nalone = 3
        sc_, oc_ = self.chunk * self.nsteps * self.spec.R, self.chunk * self.payload_bytes
        for _ in range(0 if self.windowed else nalone):
            for c in range(self.nchunks):
                self.dec.reset()
                self.dec.sync()
                self.dec.update(self.d_syms[c * sc_:(c + 1) * sc_], nbits=self.nsteps)
                self.dec.sync()
                self.dec.chainback(self.cb_bits, out=self.d_out[0][c * oc_:(c + 1) * oc_])
                self.dec.sync()
        au, _, ac, _ = self.dec.read_timing()
        alone_u, alone_c = (su / passes, 0.0) if self.windowed else (au / nalone, ac / nalone)
        return {
            "units_per_pass": self.frames * self.nsteps * self.spec.R,
            "bit_errors": nerr,
            "update_ms_alone": alone_u,
            "chainback_ms_alone": alone_c,
            "update_ms": su / passes,          # per pass (all chunks), kernels only, on the stream they ran on
            "chainback_ms": sc / passes,
            "timed_update_launches": nu,
            "variant": self.dec.variant,
        }

    def close(self):
        self.dec.close()


def replayed_counter(kind, code, family):
    """profiles/<kind>_<code>.json from a committed rocprofv3 --pmc pass: quoted only while the kernel sources it was taken
    on are unchanged (hash recorded by tools/summarize_pmc.py / summarize_valu.py); otherwise None + reason."""
    path = os.path.join(ROOT, "profiles", f"{kind}_{code}.json")
    if not os.path.exists(path):
        return None, "no committed PMC pass"
    try:
        v = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable: {e}"
    if v.get("kernel_source_sha256") != kernel_source_hash(family):
        return None, f"stale: {os.path.relpath(path, ROOT)} was taken on other kernel sources"
    return v, f"replayed from {os.path.relpath(path, ROOT)} (rocprofv3 --pmc, sources {v.get('kernel_source_sha256')}); not measured in this run"


# ------------------------------------------------------------------------------------------------ one JSON object per workload
def report(args, spec, frames, core, shard, with_cpu, cpu_budget):
    """The bench line of one workload from the job loop's core figures and the shard's kernel timings."""
    st = core.pop("stats")
    n_gpus, nsteps, payload_bits = core["n_gpus"], shard.nsteps, shard.payload_bits
    upd_ms, cb_ms = st["update_ms"], st["chainback_ms"]
    variant = st["variant"]
    family = kernel_family(spec.K)
    abytes = algorithmic_bytes_per_frame_step(spec) * frames * nsteps  # per pass of one rank
    mbytes = moved_bytes_per_frame_step(spec, variant, {4: 5, 5: 2}.get(variant)) * frames * nsteps
    achieved = mbytes / (upd_ms * 1e-3) / 1e9
    tv, tsrc = replayed_counter("traffic", args.code, family)
    vv, vsrc = replayed_counter("valu", args.code, family)
    valu = None
    if vv is not None:
        # instructions per launch come from the committed PMC pass; the time is this run's
        rate = vv["valu_wave_instr_per_launch"] / (upd_ms * 1e-3) / 1e9
        valu = {"wave_instr_per_launch": vv["valu_wave_instr_per_launch"], "achieved_ginstr_per_s": round(rate, 2),
                "issue_ceiling_ginstr_per_s": round(vv["issue_ceiling_ginstr_per_s"], 2),
                "frac": round(rate / vv["issue_ceiling_ginstr_per_s"], 4),
                # the same kernel with nothing else on the device (roofline.alone)
                "frac_alone": round(vv["valu_wave_instr_per_launch"] / (st["update_ms_alone"] * 1e-3) / 1e9 / vv["issue_ceiling_ginstr_per_s"], 4),
                "source": vsrc}
    out = {
        "metric": f"Msymbols/s decoded, K={spec.K} r=1/{spec.R} " + ("(fused sliding-window decode: NOT the reference chainback semantics)" if shard.windowed else "(init + ACS update + chainback)"),
        "value": round(core["value"] / 1e6, 6),
        "unit": "Msymbols/s",
        "n_gpus": n_gpus,
        "steps": core["steps"],
        "warmup": core["warmup"],
        "ms_per_step": round(core["ms_per_step"], 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8" if spec.family != "ka9q-i16-sat" else "i16",
        "data": "synthetic" + (" hard 0/255 symbols" if args.hard else f" AWGN Eb/N0={shard.ebn0} dB, amplitude {C.SOFT_AMP}") + ", generated on device",
        "config": {"workload": f"viterbi{spec.name}: K={spec.K} r=1/{spec.R}, {frames} frames/GPU x {payload_bits} info bits "
                               f"({nsteps} trellis steps, {nsteps * spec.R} symbols/frame)",
                   "frames_per_gpu": frames, "chunk_frames": shard.chunk, "payload_bits": payload_bits, "variant": variant,
                   "pipeline_depth": shard.depth,
                   "parallelism": f"frame-shard x{n_gpus}, no collective"},
        "update_msym_s": round(frames * nsteps * spec.R * n_gpus / (upd_ms * 1e-3) / 1e6, 6),
        "chainback_mbit_s": round(frames * shard.cb_bits * n_gpus / (cb_ms * 1e-3) / 1e6, 3) if cb_ms > 0 else None,
        "update_ms": round(upd_ms, 4),
        "chainback_ms": round(cb_ms, 4),
        "kernel_timing": "HIP events recorded by the library around its launches, on the streams they run on, timed region only"
                         + ("; update and chainback of consecutive steps overlap (pipeline depth 2), so the two do not add up to ms_per_step" if shard.depth > 1 else ""),
        "bit_errors": int(core["bit_errors"]),
        "payload_bits_total": frames * payload_bits * n_gpus,
        # achieved = bytes the ACS kernels move per pass / their event-timed duration; for K <= 15 that IS the
        # algorithmic figure of SURVEY.md §8d, for the multi-step K=24 passes it is what the passes really move
        "roofline": {"bound": "hbm", "kernel": "acs_update", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": (tv or {}).get("hbm_bytes_per_launch"), "traffic_source": tsrc,
                     "bytes_per_launch": int(mbytes), "algorithmic_bytes_per_launch": int(abytes),
                     # On a double-buffered handle consecutive update kernels overlap each other and the previous
                     # chainback, so one launch lasts longer than the interval at which launches complete:
                     # `frac` above divides by the launch duration (the contract), this one by ms_per_step
                     "per_step": {"achieved": round(mbytes / (core["ms_per_step"] * 1e-3) / 1e9, 2),
                                  "frac": round(mbytes / (core["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                     # the same kernel with nothing else on the device (3 passes after the timed region)
                     "alone": {"update_ms": round(st["update_ms_alone"], 4), "achieved": round(mbytes / (st["update_ms_alone"] * 1e-3) / 1e9, 2),
                               "frac": round(mbytes / (st["update_ms_alone"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                               "chainback_ms": round(st["chainback_ms_alone"], 4)}},
        # the binding limit of the K<=15 ACS kernels: packed-integer VALU issue, not HBM
        "valu_issue": valu,
    }
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(spec, payload_bits, cpu_budget)
    return out


# BASELINE.json configs[2] and configs[3] ride on the same line as the headline (configs[1]) under "extra.configs", so the
# driver-run record carries every single-GPU configuration: fewer steps and a shorter CPU sample each, same fields.
EXTRA_CONFIGS = [("615", 10, 2, 4.0), ("224", 10, 2, 1.0)]  # (code, steps, warmup, cpu budget s; one K=24 CPU decode takes ~5 s whatever the budget)


def run_extra_config(base_args, code, steps, warmup, cpu_budget, dev):
    args = argparse.Namespace(**vars(base_args))
    args.code, args.frames, args.payload_bits, args.variant, args.chunk_frames, args.windowed = code, None, None, 0, None, False
    args.ebn0, args.hard = None, False
    spec = C.CODES[code]
    frames = DEFAULT_FRAMES[spec.K]
    holder = {}

    def make_shard(frame_lo, nframes):
        holder["shard"] = HipShard(args, spec, dev, frame_lo, nframes)
        return holder["shard"]

    core = run_sharded_job(make_shard, frames_per_rank=frames, steps=steps, warmup=warmup, rank=0, world=1, device=dev, reduce_device=None)
    shard = holder["shard"]
    try:
        return report(args, spec, frames, core, shard, not base_args.no_cpu_baseline, min(cpu_budget, base_args.cpu_budget))
    finally:
        shard.close()
        del holder["shard"], shard
        torch.cuda.empty_cache()


# The reference's OWN measurement (src/main.cpp:264-278: one frame per call, host buffers, each of the three calls timed by
# itself) through the five-function C ABI that replaces its decoders -- the geometry a maintainer sees after swapping the
# library in without batching anything.  Not a throughput claim: one frame cannot fill the chip, these are latency figures.
ONE_FRAME_HANDLES = [("27", "viterbi27", "_blk"), ("47", "spiral47", ""), ("29", "viterbi29", "_blk"), ("49", "spiral49", ""),
                     ("615", "viterbi615", "_blk"), ("224", "viterbi224", "_blk")]


def one_frame_handles(budget_s=0.4):
    from ka9q_viterbi_comparison_amd import _lib

    lib = _lib.load()
    rows = []
    for code, stem, blk in ONE_FRAME_HANDLES:
        spec = C.CODES[code]
        nbits = spec.ref_payload_bytes * 8  # the reference's frame for this decoder (main.cpp:364-418)
        steps = nbits + spec.K - 1
        payload, syms = gen_frames_host(spec, 77, 0, 1, spec.ref_payload_bytes, C.SOFT_AMP_Q16, noise_q12(spec.R, 64.0, spec.ebn0_db))
        create, init = getattr(lib, f"create_{stem}_hip"), getattr(lib, f"init_{stem}_hip")
        update, chainback, delete = getattr(lib, f"update_{stem}{blk}_hip"), getattr(lib, f"chainback_{stem}_hip"), getattr(lib, f"delete_{stem}_hip")
        poly = (ctypes.c_int * spec.R)(*spec.poly)
        h = create(poly, steps)  # the reference creates with the transmit length (main.cpp:247)
        if not h:
            rows.append({"code": code, "error": _lib.last_error()})
            continue
        try:
            out = np.zeros(spec.ref_payload_bytes, dtype=np.uint8)
            sp, op = syms.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)
            t_upd = t_cb = 0.0
            n = 0
            t_end = time.perf_counter() + budget_s
            for it in range(-2, 100000):  # two untimed passes first
                init(h, 0)
                t0 = time.perf_counter()
                update(h, sp, steps)
                t1 = time.perf_counter()
                chainback(h, op, nbits, 0)
                t2 = time.perf_counter()
                if it >= 0:
                    t_upd, t_cb, n = t_upd + (t1 - t0), t_cb + (t2 - t1), n + 1
                    if t2 > t_end and n >= 3:
                        break
            if spec.K == 24:
                # chainback_viterbi224 does not skip the tail rows: the payload comes out of the nbits+K-1 call
                long_out = np.zeros((steps + 7) // 8, dtype=np.uint8)
                chainback(h, long_out.ctypes.data_as(ctypes.c_void_p), steps, 0)
                out = long_out[:spec.ref_payload_bytes]
            errors = int(np.unpackbits(out ^ payload[0]).sum())
            rows.append({"code": code, "K": spec.K, "R": spec.R, "payload_bits": nbits, "decodes": n, "bit_errors": errors,
                         "update_msym_s": round(steps * spec.R * n / t_upd / 1e6, 3), "update_ms": round(t_upd / n * 1e3, 4),
                         "chainback_mbit_s": round(nbits * n / t_cb / 1e6, 3), "chainback_ms": round(t_cb / n * 1e3, 4)})
        finally:
            delete(h)
    return rows


# BASELINE.json configs: K=7 x 65536 frames, K=15 x 4096 frames, K=24 single long frame; K=9 sized in between
DEFAULT_FRAMES = {7: 65536, 9: 32768, 15: 4096, 24: 1}


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
    ap.add_argument("--pipeline-depth", type=int, default=2, help="decision-history buffers behind the handle (K<=15): 2 = double-buffered (default), 1 = strictly serial")
    ap.add_argument("--no-pipeline", action="store_true", help="same as --pipeline-depth 1")
    ap.add_argument("--windowed", action="store_true", help="K<=9: time the fused sliding-window decode (one kernel, no decision history in HBM) "
                                                            "instead of init + update + chainback; results are NOT those of the reference chainback")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--no-extra-configs", action="store_true", help="only the headline workload (default: at N=1 with the default workload, BASELINE "
                                                                    "configs[2] K=15 x 4096 and configs[3] K=24 x 1 follow under extra.configs)")
    args = ap.parse_args()
    if args.no_pipeline:
        args.pipeline_depth = 1

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
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    spec = C.CODES[args.code]
    frames = args.frames or DEFAULT_FRAMES[spec.K]
    dev = torch.device("cuda", dev_index)
    rdev = dev if backend == "nccl" else None  # where the scalar reductions live
    holder = {}

    def make_shard(frame_lo, nframes):
        holder["shard"] = HipShard(args, spec, dev, frame_lo, nframes)
        return holder["shard"]

    core = run_sharded_job(make_shard, frames_per_rank=frames, steps=args.steps, warmup=args.warmup, rank=rank, world=world,
                           device=dev, reduce_device=rdev)
    shard = holder["shard"]
    out = None
    if rank == 0:
        out = report(args, spec, frames, core, shard, core["n_gpus"] == 1 and not args.no_cpu_baseline, args.cpu_budget)
    shard.close()
    del holder["shard"], shard
    default_workload = args.code == "27" and args.frames is None and args.payload_bits is None and not args.windowed and args.variant == 0
    if rank == 0 and world == 1 and default_workload and not args.no_extra_configs:
        torch.cuda.empty_cache()
        extras = []
        for code, steps, warmup, cpu_budget in EXTRA_CONFIGS:
            try:
                extras.append(run_extra_config(args, code, steps, warmup, cpu_budget, dev))
            except Exception as e:  # noqa: BLE001 -- the headline stands on its own; a failed extra says so
                extras.append({"config": {"workload": f"viterbi{code}"}, "error": f"{type(e).__name__}: {e}"})
        out["extra"] = {"note": "BASELINE.json configs[2] (K=15 x 4096 frames) and configs[3] (K=24, one 2048-bit frame), measured after the "
                                "headline in the same process: same fields, fewer steps, a shorter CPU sample",
                        "configs": extras}
        try:
            out["extra"]["one_frame_handles"] = {
                "note": "the reference's own measurement (src/main.cpp:264-278): ONE frame of its size per call through the five-function "
                        "C ABI with host buffers, update and chainback timed separately around the blocking calls (PCIe copies and "
                        "launch latency included) -- latency figures for an unbatched drop-in, not throughput",
                "decoders": one_frame_handles()}
        except Exception as e:  # noqa: BLE001
            out["extra"]["one_frame_handles"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
