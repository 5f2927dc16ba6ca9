#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (collected in SEPARATE passes, as MI355X_MICROARCH.md
§HBM prescribes) into profiles/traffic_<code>.json, which bench.py reads for roofline.traffic.

    python tools/summarize_pmc.py <code> <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> [name]

(name: file stem, default <code>; e.g. 615_chainback for the chainback kernel of the K=15 run)

Units and corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read, so it is doubled; WRITE_SIZE is exact for streaming stores."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mean_counter(path, kernel_sub, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    code, fetch_csv, write_csv, ksub = sys.argv[1:5]
    name = sys.argv[5] if len(sys.argv) > 5 else code
    if code == "224":
        # K=24: one update is many pass launches.  Mean counters of the full-pass kernels of each kind (H: 9 steps, L: 14
        # steps), added up per 23-step period and scaled to the bench's 2071-step update.
        steps = 2048 + 23
        f = w = 0.0
        nf = nw = 0
        for kind in ("pass_h_kernel<true", "pass_l_kernel<true"):
            fk, n1 = mean_counter(fetch_csv, kind, "FETCH_SIZE")
            wk, n2 = mean_counter(write_csv, kind, "WRITE_SIZE")
            f += fk * steps / 23.0
            w += wk * steps / 23.0
            nf += n1
            nw += n2
    else:
        f, nf = mean_counter(fetch_csv, ksub, "FETCH_SIZE")
        w, nw = mean_counter(write_csv, ksub, "WRITE_SIZE")
    from kernel_hash import kernel_family, kernel_source_hash
    from ka9q_viterbi_comparison_amd.codes import CODES

    out = {
        "kernel": ksub,
        "kernel_source_sha256": kernel_source_hash(kernel_family(CODES[code].K)),
        "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "launches_averaged": [nf, nw],
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": int((2.0 * f + w) * 1024),
        "note": "separate --pmc passes; FETCH_SIZE x2 (gfx950 half-count of wide reads), WRITE_SIZE exact; KiB -> bytes",
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(os.environ.get("PROFILES_DIR", os.path.join(root, "profiles")), f"traffic_{name}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, out)


if __name__ == "__main__":
    main()
