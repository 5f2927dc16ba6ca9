"""Fingerprint of the source files an ACS kernel family is built from.  rocprofv3 --pmc passes are collected separately
from the bench run (MI355X_MICROARCH.md §HBM: one counter set per pass), so their summaries under profiles/ are REPLAYED
into the bench line; the fingerprint recorded with a summary lets bench.py drop it once the kernel has been edited."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join("ka9q_viterbi_comparison_amd", "csrc")
KERNEL_SOURCES = {
    "regs": [os.path.join(CSRC, "acs_regs.hip"), os.path.join(CSRC, "viterbi_codes.h")],
    "k15": [os.path.join(CSRC, "acs_k15.hip"), os.path.join(CSRC, "k15_layout.h"), os.path.join(CSRC, "viterbi_codes.h")],
    "k24": [os.path.join(CSRC, "acs_k24f.hip"), os.path.join(CSRC, "acs_k24t.hip"), os.path.join(CSRC, "k24f_layout.h"),
            os.path.join(CSRC, "k24t_layout.h"), os.path.join(CSRC, "viterbi_codes.h")],
}


def kernel_family(K):
    return "k24" if K == 24 else ("k15" if K == 15 else "regs")


def kernel_source_hash(family):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[family]:
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            h.update(rel.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def fnv1a64_of_files(paths):
    """FNV-1a/64 over the concatenated contents: the fingerprint csrc/Makefile bakes into libviterbi_hip.so
    (jit_sources_hash.h) and jit.hip recomputes before every run-time build."""
    h = 0xCBF29CE484222325
    for path in paths:
        for b in open(path, "rb").read():
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 2 and sys.argv[1] == "--fnv":
        print(f"{fnv1a64_of_files(sys.argv[2:]):016x}")
    else:
        for fam in KERNEL_SOURCES:
            print(fam, kernel_source_hash(fam))
