#!/usr/bin/env python3
"""README results table from profiles/<tag>_sweep.jsonl (+ <tag>_windowed_decode.jsonl, <tag>_config5_shard_size.jsonl).
    python tools/readme_table.py r02"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def si(x, unit):
    for u, d in (("T", 1e12), ("G", 1e9), ("M", 1e6), ("k", 1e3)):
        if x >= d:
            return f"{x / d:.3g} {u}{unit}"
    return f"{x:.3g} {unit}"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", f"{tag}_sweep.jsonl"))]
    print("| code | batch | decode = `value` (double-buffered handle) | ACS update alone | chainback alone | HBM-roofline fraction of the ACS kernel (alone / in the pipeline) | reference CPU, 1 thread / all cores |")
    print("|---|---|---|---|---|---|---|")
    for d in rows:
        c = d["config"]
        wl = c["workload"].split(":")[0].replace("viterbi", "")
        kr = c["workload"].split(": ")[1].split(",")[0]
        al = d["roofline"]["alone"]
        n = c["frames_per_gpu"]
        syms = d["update_msym_s"] * d["update_ms"]  # Msym per pass * 1e-3
        upd_alone = syms / al["update_ms"] * 1e6
        cb_alone = n * c["payload_bits"] / (al["chainback_ms"] * 1e-3) if al["chainback_ms"] else 0
        cb = d.get("cpu_baseline", {})
        print(f"| {kr} ({wl}) | {n} × {c['payload_bits']} bit | {si(d['value'] * 1e6, 'sym/s')} | {si(upd_alone, 'sym/s')} ({al['update_ms']:.3g} ms) | "
              f"{si(cb_alone, 'bit/s')} ({al['chainback_ms']:.3g} ms) | {al['frac']:.3f} / {d['roofline']['frac']:.3f} | "
              f"{si(cb.get('single_thread_value', 0) * 1e6, 'sym/s')} / {si(cb.get('value', 0) * 1e6, 'sym/s')} ({cb.get('cores', '-')} thr) |")
    for extra, title in ((f"{tag}_config5_shard_size.jsonl", "config-5 per-GPU shard (131 072 frames per code; K=24: 12 frames of the reference's 64-bit size)"),
                         (f"{tag}_windowed_decode.jsonl", "fused sliding-window decode (no decision history in HBM; NOT the reference chainback semantics)")):
        path = os.path.join(ROOT, "profiles", extra)
        if not os.path.exists(path):
            continue
        print(f"\n{title}:\n")
        print("| workload | `value` | ms per step | chunk frames | bit errors |")
        print("|---|---|---|---|---|")
        for l in open(path):
            d = json.loads(l)
            print(f"| {d['config']['workload'].split(' (')[0]} | {si(d['value'] * 1e6, 'sym/s')} | {d['ms_per_step']:.4g} | {d['config']['chunk_frames']} | {d['bit_errors']} |")


if __name__ == "__main__":
    main()
