#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE pass into profiles/valu_<code>.json:
wave-level VALU instructions per launch of the ACS kernel, to set against the measured issue ceiling of the chip
(tools/valu_rate.hip -> profiles/r01_valu_rate.txt: one VOP3P wave64 instruction per ~1.8 ns per SIMD, 1024 SIMDs).

    python tools/summarize_valu.py <code> <counter_collection.csv> <kernel substring> <kernel ms>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    code, path, ksub, ms = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    vals = {}
    for r in csv.DictReader(open(path)):
        if ksub in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in vals.items()}
    valu = mean.get("SQ_INSTS_VALU", 0.0)
    peak = 1024 / 1.8e-9  # wave64 VOP3P instructions per second, whole chip (measured issue interval)
    from kernel_hash import kernel_family, kernel_source_hash
    from ka9q_viterbi_comparison_amd.codes import CODES

    out = {"kernel": ksub, "counters_mean": mean, "kernel_ms": ms,
           "kernel_source_sha256": kernel_source_hash(kernel_family(CODES[code].K)),
           "valu_wave_instr_per_launch": valu,
           "achieved_ginstr_per_s": valu / (ms * 1e-3) / 1e9,
           "issue_ceiling_ginstr_per_s": peak / 1e9,
           "frac_of_issue_ceiling": valu / (ms * 1e-3) / peak,
           "note": "ceiling = 1024 SIMDs x one VOP3P wave64 instruction per 1.8 ns (profiles/r01_valu_rate.txt); "
                   "full-rate VOP2 instructions issue faster, so a mixed stream can exceed 1.0 slightly"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(os.environ.get("PROFILES_DIR", os.path.join(root, "profiles")), f"valu_{code}.json")
    json.dump(out, open(p, "w"), indent=1)
    print(p, json.dumps(out)[:400])


if __name__ == "__main__":
    main()
