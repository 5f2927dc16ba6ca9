"""ctypes binding of the C ABI in include/viterbi_hip.h (libviterbi_hip.so).

The library is built in-tree by ``make -C ka9q_viterbi_comparison_amd/csrc`` (see ``__graft_entry__.build``).
If it is missing this module raises: there is no Python or CPU fallback for the decode path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITERBI_HIP_LIB: tools/k24t_probe.sh points this at csrc/libviterbi_hip_timing.so (`make timing`); nothing else should.
LIB_PATH = os.environ.get("VITERBI_HIP_LIB") or os.path.join(_HERE, "csrc", "libviterbi_hip.so")

_lib = None


class VhipError(RuntimeError):
    pass


# every symbol include/viterbi_hip.h declares: (name, restype, argtypes)
_u8p = C.POINTER(C.c_ubyte)
_i32p = C.POINTER(C.c_int)
_FIVE = []
for _t, _c, _i, _u, _cb, _d in [
    ("v27", "create_viterbi27_hip", "init_viterbi27_hip", "update_viterbi27_blk_hip", "chainback_viterbi27_hip", "delete_viterbi27_hip"),
    ("v29", "create_viterbi29_hip", "init_viterbi29_hip", "update_viterbi29_blk_hip", "chainback_viterbi29_hip", "delete_viterbi29_hip"),
    ("v615", "create_viterbi615_hip", "init_viterbi615_hip", "update_viterbi615_blk_hip", "chainback_viterbi615_hip", "delete_viterbi615_hip"),
    ("v224", "create_viterbi224_hip", "init_viterbi224_hip", "update_viterbi224_blk_hip", "chainback_viterbi224_hip", "delete_viterbi224_hip"),
    ("spiral47", "create_spiral47_hip", "init_spiral47_hip", "update_spiral47_hip", "chainback_spiral47_hip", "delete_spiral47_hip"),
    ("spiral49", "create_spiral49_hip", "init_spiral49_hip", "update_spiral49_hip", "chainback_spiral49_hip", "delete_spiral49_hip"),
    ("spiral27", "create_spiral27_hip", "init_spiral27_hip", "update_spiral27_hip", "chainback_spiral27_hip", "delete_spiral27_hip"),
    ("spiral29", "create_spiral29_hip", "init_spiral29_hip", "update_spiral29_hip", "chainback_spiral29_hip", "delete_spiral29_hip"),
    ("spiral615", "create_spiral615_hip", "init_spiral615_hip", "update_spiral615_hip", "chainback_spiral615_hip", "delete_spiral615_hip"),
]:
    _FIVE += [
        (_c, C.c_void_p, [_i32p, C.c_int]),
        (_i, C.c_int, [C.c_void_p, C.c_int]),
        (_u, None, [C.c_void_p, C.c_void_p, C.c_int]),
        (_cb, C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
        (_d, None, [C.c_void_p]),
    ]

SYMBOLS = [
    ("vhip_create", C.c_void_p, [C.c_int, _i32p, C.c_int, C.c_int]),
    ("vhip_init", C.c_int, [C.c_void_p, C.c_int]),
    ("vhip_update", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("vhip_chainback", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    ("vhip_update_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("vhip_chainback_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]),
    ("vhip_delete", None, [C.c_void_p]),
    ("vhip_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("vhip_sync", C.c_int, [C.c_void_p]),
    ("vhip_device_count", C.c_int, []),
    ("vhip_last_error", C.c_char_p, []),
    ("vhip_decode_windowed_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]),
    ("vhip_window_depth", C.c_int, [C.c_void_p]),
    ("vhip_window_block", C.c_int, [C.c_void_p]),
    ("vhip_status", C.c_int, [C.c_void_p]),
    ("vhip_enable_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("vhip_read_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("vhip_set_pipeline_depth", C.c_int, [C.c_void_p, C.c_int]),
    ("vhip_get_pipeline_depth", C.c_int, [C.c_void_p]),
    ("vhip_join", C.c_int, [C.c_void_p]),
    ("vhip_set_chainback_segments", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("vhip_chainback_rewalked", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("vhip_set_variant", C.c_int, [C.c_void_p, C.c_int]),
    ("vhip_get_variant", C.c_int, [C.c_void_p]),
    ("vhip_is_runtime_specialised", C.c_int, [C.c_void_p]),
    ("vhip_runtime_build_sources_ok", C.c_int, []),
    ("vhip_read_decision_rows", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("vhip_read_metrics", C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    ("vhip_rows_written", C.c_int, [C.c_void_p]),
    ("vhip_code_K", C.c_int, [C.c_int]),
    ("vhip_code_R", C.c_int, [C.c_int]),
    ("vhip_device_bytes", C.c_size_t, [C.c_void_p]),
    ("vhip_noise_q12_from_ebn0", C.c_int, [C.c_int, C.c_double, C.c_double]),
    ("vhip_gen_frames_host", C.c_int, [C.c_int, C.c_int, _i32p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("vhip_gen_frames_dev", C.c_int, [C.c_int, C.c_int, _i32p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("vhip_count_bit_errors_dev", C.c_longlong, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
] + _FIVE


def load():
    """Load libviterbi_hip.so (once) and attach signatures.  Raises VhipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VhipError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    try:
        # When torch is in the process its bundled libamdhip64.so.7 must be the one HIP runtime in use
        # (device pointers and streams are shared with it): import torch first if it is installed.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the pure C-ABI use
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().vhip_last_error().decode()


def check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise VhipError(f"{what} failed: {last_error()}")
    return rc
