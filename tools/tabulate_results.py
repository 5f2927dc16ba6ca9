#!/usr/bin/env python3
"""Markdown tables from the harness JSON (harness/viterbi_bench -o file.json), in the spirit of the reference's
scripts/tabulate_data.py: update rate = total_output_symbols / update_ns (tabulate_data.py:33), chainback rate =
total_input_bytes*8 / chainback_ns (:54), mean +- std over the samples.  Accepts any file in the reference's schema
(src/main.cpp:80-118), so CPU results produced by the reference harness can be tabulated next to "hip" entries.

    python tools/tabulate_results.py data/benchmark_hip.json [more.json ...]
"""
import json
import sys

import numpy as np


def si(x):
    for unit, div in (("G", 1e9), ("M", 1e6), ("k", 1e3)):
        if abs(x) >= div:
            return f"{x / div:.3g}{unit}"
    return f"{x:.3g}"


def main():
    entries = []
    for path in sys.argv[1:]:
        entries += json.load(open(path))
    names = sorted({e["name"] for e in entries})
    codes = sorted({(e["K"], e["R"]) for e in entries})
    for title, num_key, den_key, unit in (("Symbol update", "total_output_symbols", "update_ns", "sym/s"),
                                          ("Chainback", "total_input_bytes", "chainback_ns", "bit/s")):
        print(f"### {title} ({unit})\n")
        print("| K | R | " + " | ".join(f"{n} (frames per call)" for n in names) + " |")
        print("|---|---|" + "---|" * len(names))
        for K, R in codes:
            row = []
            for n in names:
                m = [e for e in entries if e["name"] == n and (e["K"], e["R"]) == (K, R)]
                if not m:
                    row.append("-")
                    continue
                e = m[-1]
                num = e[num_key] * (8 if den_key == "chainback_ns" else 1)
                rate = num / (np.array(e[den_key], dtype=np.float64) * 1e-9)
                row.append(f"{si(rate.mean())} ± {si(rate.std())} ({e.get('frames', 1)})")
            print(f"| {K} | {R} | " + " | ".join(row) + " |")
        print()


if __name__ == "__main__":
    main()
