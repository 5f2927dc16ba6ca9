"""The C-ABI shared library: loads, exports every symbol include/viterbi_hip.h declares, validates arguments
and fails loudly without a GPU (no CPU fallback).  No device compute here."""
import ctypes as C
import os
import re

import pytest

from ka9q_viterbi_comparison_amd import _lib, codes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "viterbi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(vhip_[A-Za-z0-9_]+)\s*\(", src))
    # the five reference-shaped functions per code come from the VHIP_DECLARE_FIVE(...) invocations
    for m in re.finditer(r"VHIP_DECLARE_FIVE\(([^)]*)\)", src):
        args = [a.strip() for a in m.group(1).replace("\n", " ").split(",")]
        if args[0] == "T":
            continue  # the macro definition itself
        names.update(args[1:])
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 23 + 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/viterbi_hip.h but not exported"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) <= bound, f"ctypes binding is missing {set(names) - bound}"


def test_library_exports_nothing_else():
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table IS the header -- no vh:: internals, kernel stubs,
    std:: instantiations or HIP registration symbols a host program could collide with."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == header_functions(), sorted(set(exported) ^ set(header_functions()))


def test_shipping_library_has_no_timing_modes():
    """The K=24 timing experiments (VHIP_K24T_MODE, results wrong by design) live in `make timing` only."""
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"VHIP_K24T_MODE", b"VHIP_K24T_NT", b"VHIP_CHAINBACK_SIMPLE", b"VHIP_CHAINBACK_PIPE", b"VHIP_K24_FUSED", b"VHIP_REGS_LB"):
        assert name not in blob, name


def test_jit_fingerprint_matches_the_sources():
    """jit.hip refuses run-time builds unless the kernel sources next to the library are the ones it was built from."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_hash import fnv1a64_of_files

    csrc = os.path.dirname(_lib.LIB_PATH)
    names = ["acs_regs.hip", "acs_k15.hip", "acs_k24t.hip", "kernels.h", "viterbi_codes.h", "k24t_layout.h", "k15_layout.h"]
    baked = re.search(r"VH_JIT_SOURCES_HASH 0x([0-9a-f]+)ull", open(os.path.join(csrc, "jit_sources_hash.h")).read()).group(1)
    assert int(baked, 16) == fnv1a64_of_files([os.path.join(csrc, n) for n in names])
    # ... and the library, hashing the files itself, agrees (this caught a mistyped FNV basis that refused every run-time build)
    assert _lib.load().vhip_runtime_build_sources_ok() == 1, _lib.last_error()
    # the same list, in the same order, on both sides
    assert all(n in open(os.path.join(csrc, "jit.hip")).read() for n in names)


def test_code_table():
    lib = _lib.load()
    for spec in codes.CODES.values():
        assert lib.vhip_code_K(spec.code) == spec.K and lib.vhip_code_R(spec.code) == spec.R
    assert lib.vhip_code_K(99) == 0


def test_create_rejects_bad_arguments_loudly():
    lib = _lib.load()
    poly = (C.c_int * 2)(0x6D, 0x4F)
    assert not lib.vhip_create(99, poly, 100, 1)
    assert b"unknown code" in lib.vhip_last_error()
    assert not lib.vhip_create(codes.KA9Q27, poly, -1, 1)
    assert not lib.vhip_create(codes.KA9Q27, poly, 100, 0)
    bad = (C.c_int * 2)(0x6C, 0x4F)  # bit 0 clear: butterfly symmetry broken
    assert not lib.vhip_create(codes.KA9Q27, bad, 100, 1)
    assert b"polynomial" in lib.vhip_last_error()
    assert not lib.create_viterbi27_hip(bad, 100)


def test_no_gpu_means_failure_not_fallback():
    lib = _lib.load()
    if lib.vhip_device_count() > 0:
        pytest.skip("a GPU is present")
    poly = (C.c_int * 2)(0x6D, 0x4F)
    assert not lib.vhip_create(codes.KA9Q27, poly, 100, 1)
    assert b"no HIP device" in lib.vhip_last_error()
    with pytest.raises(_lib.VhipError):
        from ka9q_viterbi_comparison_amd import HipViterbi

        HipViterbi("27", 100)


def test_null_handles_are_safe():
    lib = _lib.load()
    lib.vhip_delete(None)  # delete(NULL) is a no-op (viterbi27_sse2.cpp:108-115)
    lib.delete_viterbi27_hip(None)
    assert lib.vhip_init(None, 0) == -1  # viterbi224_sse2.cpp:36-37
    assert lib.vhip_update_dev(None, None, 10) == -1
    assert lib.vhip_chainback_dev(None, None, 10, 0) == -1


def test_noise_scale_helper():
    lib = _lib.load()
    q = lib.vhip_noise_q12_from_ebn0(2, 64.0, 4.0)
    sigma = 64.0 / (2 * 10 ** 0.4 / 2) ** 0.5
    assert abs(q - sigma * 65536 * 4096 / 107020.0) <= 1
