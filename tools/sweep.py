#!/usr/bin/env python3
"""Runs bench.py for every code (BASELINE.json config 5 style sweep on one GPU) and writes the JSON lines plus a
markdown table.   python tools/sweep.py [--out profiles/r02_sweep] [--steps 20] [--cpu]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_sweep"))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu", action="store_true", help="also time the CPU baseline per code (adds ~12 s each)")
    ap.add_argument("--codes", default="27,47,29,49,615,224,spiral27,spiral29,spiral615")
    args = ap.parse_args()
    lines = []
    for code in args.codes.split(","):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--code", code, "--steps", str(args.steps), "--warmup", "3", "--no-extra-configs"]
        if not args.cpu:
            cmd.append("--no-cpu-baseline")
        r = subprocess.run(cmd, capture_output=True, text=True)
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not js:
            print(f"bench failed for {code}: {r.stderr[-500:]}", file=sys.stderr)
            continue
        lines.append(json.loads(js[-1]))
        print(js[-1][:160], flush=True)
    with open(args.out + ".jsonl", "w") as f:
        for d in lines:
            f.write(json.dumps(d) + "\n")
    with open(args.out + ".md", "w") as f:
        f.write("| workload | decode Msym/s | update Msym/s | chainback Mbit/s | update ms | chainback ms | HBM roofline frac | CPU 1 thread Msym/s | CPU all cores Msym/s |\n")
        f.write("|---|---|---|---|---|---|---|---|---|\n")
        for d in lines:
            cb = d.get("cpu_baseline", {})
            f.write(f"| {d['config']['workload']} | {d['value']:.4g} | {d['update_msym_s']:.4g} | {d['chainback_mbit_s']:.4g} | "
                    f"{d['update_ms']:.4g} | {d['chainback_ms']:.4g} | {d['roofline']['frac']:.3f} | "
                    f"{cb.get('single_thread_value', '-')} | {cb.get('value', '-')} ({cb.get('cores', '-')} thr) |\n")
    print("wrote", args.out + ".jsonl", args.out + ".md")


if __name__ == "__main__":
    main()
