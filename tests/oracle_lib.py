"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when built, the compiled reference
(oracle/_ref/libref.so, oracle/_ref/libref_ka9q615_w32.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
REF_W32_SO = os.path.join(ROOT, "oracle", "_ref", "libref_ka9q615_w32.so")

_oracle = None
_refs = {}


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle`")
        lib = C.CDLL(ORACLE_SO)
        lib.vo_create.restype = C.c_void_p
        lib.vo_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
        lib.vo_init.argtypes = [C.c_void_p, C.c_int]
        lib.vo_update_blk.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.vo_update_blk.restype = None
        lib.vo_chainback.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
        lib.vo_chainback_windowed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]
        lib.vo_delete.argtypes = [C.c_void_p]
        lib.vo_delete.restype = None
        lib.vo_decision_rows.restype = C.c_void_p
        lib.vo_decision_rows.argtypes = [C.c_void_p]
        lib.vo_row_bytes.restype = C.c_size_t
        lib.vo_row_bytes.argtypes = [C.c_void_p]
        lib.vo_rows_written.argtypes = [C.c_void_p]
        lib.vo_num_states.argtypes = [C.c_void_p]
        lib.vo_renorm_count.argtypes = [C.c_void_p]
        lib.vo_get_metrics.argtypes = [C.c_void_p, C.c_void_p]
        lib.vo_get_metrics.restype = None
        lib.vo_code_K.argtypes = [C.c_int]
        lib.vo_code_R.argtypes = [C.c_int]
        lib.vo_encode.restype = C.c_size_t
        lib.vo_encode.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vo_decode_batch.restype = None
        lib.vo_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_uint, C.c_void_p, C.c_long]
        lib.vo_bench_loop.restype = C.c_long
        lib.vo_bench_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_uint, C.c_double, C.POINTER(C.c_double)]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO) and os.path.exists(REF_W32_SO)


def ref(w32=False):
    path = REF_W32_SO if w32 else REF_SO
    if path not in _refs:
        lib = C.CDLL(path)
        lib.ref_create.restype = C.c_void_p
        lib.ref_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
        lib.ref_init.argtypes = [C.c_int, C.c_void_p, C.c_int]
        lib.ref_update.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_update.restype = None
        lib.ref_chainback.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
        lib.ref_delete.argtypes = [C.c_int, C.c_void_p]
        lib.ref_delete.restype = None
        lib.ref_rows.restype = C.c_void_p
        lib.ref_rows.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
        lib.ref_metrics.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        lib.ref_metrics.restype = None
        lib.ref_sizeof_long.restype = C.c_int
        lib.ref_decode_batch.restype = None
        lib.ref_decode_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_uint, C.c_void_p, C.c_long]
        lib.ref_bench_loop.restype = C.c_long
        lib.ref_bench_loop.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_uint, C.c_double, C.POINTER(C.c_double)]
        _refs[path] = lib
    return _refs[path]


def _poly(poly):
    return (C.c_int * len(poly))(*poly)


class OracleDecoder:
    """Plain-C restatement (oracle/viterbi_oracle.c)."""

    def __init__(self, code, poly, length):
        self.lib = oracle()
        self.code = code
        self.K = self.lib.vo_code_K(code)
        self.R = self.lib.vo_code_R(code)
        self.h = self.lib.vo_create(code, _poly(poly), length)
        assert self.h

    def init(self, start=0):
        return self.lib.vo_init(self.h, start)

    def update(self, syms, nbits):
        s = np.ascontiguousarray(syms, dtype=np.uint8)
        assert s.size >= nbits * self.R
        self.lib.vo_update_blk(self.h, s.ctypes.data_as(C.c_void_p), nbits)

    def chainback(self, nbits, endstate=0):
        data = np.zeros((nbits + 7) // 8, dtype=np.uint8)
        rc = self.lib.vo_chainback(self.h, data.ctypes.data_as(C.c_void_p), nbits, endstate)
        return data, rc

    def chainback_windowed(self, nbits, depth, block):
        data = np.zeros((nbits + 7) // 8, dtype=np.uint8)
        rc = self.lib.vo_chainback_windowed(self.h, data.ctypes.data_as(C.c_void_p), nbits, depth, block)
        assert rc == 0
        return data

    def rows(self, nrows=None):
        n = self.lib.vo_rows_written(self.h) if nrows is None else nrows
        rb = self.lib.vo_row_bytes(self.h)
        buf = (C.c_ubyte * (n * rb)).from_address(self.lib.vo_decision_rows(self.h))
        return np.frombuffer(buf, dtype=np.uint8).reshape(n, rb).copy()

    def metrics(self):
        out = np.zeros(self.lib.vo_num_states(self.h), dtype=np.int32)
        self.lib.vo_get_metrics(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    @property
    def renorms(self):
        return self.lib.vo_renorm_count(self.h)

    @property
    def rows_written(self):
        return self.lib.vo_rows_written(self.h)

    def close(self):
        if self.h:
            self.lib.vo_delete(self.h)
            self.h = None

    def __del__(self):
        self.close()


class RefDecoder:
    """The genuine reference decoder objects (oracle/_ref), driven through oracle/ref_shim.cpp."""

    def __init__(self, code, poly, length, w32=False):
        self.lib = ref(w32)
        self.code = code
        o = oracle()
        self.K = o.vo_code_K(code)
        self.R = o.vo_code_R(code)
        self.N = 1 << (self.K - 1)
        self.h = self.lib.ref_create(code, _poly(poly), length)
        assert self.h

    def init(self, start=0):
        return self.lib.ref_init(self.code, self.h, start)

    def update(self, syms, nbits):
        s = np.ascontiguousarray(syms, dtype=np.uint8).copy()
        self.lib.ref_update(self.code, self.h, s.ctypes.data_as(C.c_void_p), nbits)

    def chainback(self, nbits, endstate=0):
        data = np.zeros((nbits + 7) // 8, dtype=np.uint8)
        rc = self.lib.ref_chainback(self.code, self.h, data.ctypes.data_as(C.c_void_p), nbits, endstate)
        return data, rc

    def rows(self, nrows):
        stride = C.c_size_t(0)
        base = self.lib.ref_rows(self.code, self.h, C.byref(stride))
        rb = self.N // 8
        buf = (C.c_ubyte * (nrows * stride.value)).from_address(base)
        a = np.frombuffer(buf, dtype=np.uint8).reshape(nrows, stride.value)
        return a[:, :rb].copy()

    def metrics(self):
        out = np.zeros(self.N, dtype=np.int32)
        self.lib.ref_metrics(self.code, self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def close(self):
        if self.h:
            self.lib.ref_delete(self.code, self.h)
            self.h = None

    def __del__(self):
        self.close()


def encode(K, R, poly, payload):
    """Coded bits (0/1), step-major, via the oracle's encoder (SURVEY.md App. A.1 convention)."""
    p = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros((p.size * 8 + K - 1) * R, dtype=np.uint8)
    n = oracle().vo_encode(K, R, _poly(poly), p.ctypes.data_as(C.c_void_p), p.size, out.ctypes.data_as(C.c_void_p))
    assert n == out.size
    return out


def decode_batch_cpu(code, poly, syms, steps, nbits, threads=8):
    """Every frame of `syms` ([nframes, steps*R] uint8) decoded on the CPU -- by the compiled reference where oracle/_ref is
    present (kind "reference"), else by the plain-C restatement -- with reset + update + chainback per frame, as the harness calls
    them.  Frames are split over `threads` decoder objects (ctypes releases the GIL).  Returns (bytes [nframes, ceil(nbits/8)], kind)."""
    import concurrent.futures as cf

    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    nframes, stride = syms.shape
    nb = (nbits + 7) // 8
    out = np.zeros((nframes, nb), dtype=np.uint8)
    use_ref = have_ref()
    w32 = code == 2  # ka9q615: the 32-bit-word chainback
    decs = [(RefDecoder(code, poly, steps, w32=w32) if use_ref else OracleDecoder(code, poly, steps)) for _ in range(threads)]  # serially: table init is not thread-safe
    bounds = np.linspace(0, nframes, threads + 1).astype(int)

    def work(k):
        lo, hi = int(bounds[k]), int(bounds[k + 1])
        if hi <= lo:
            return
        d = decs[k]
        sp = C.c_void_p(syms.ctypes.data + lo * stride)
        op = C.c_void_p(out.ctypes.data + lo * nb)
        if use_ref:
            d.lib.ref_decode_batch(code, d.h, sp, hi - lo, stride, steps, nbits, op, nb)
        else:
            d.lib.vo_decode_batch(d.h, sp, hi - lo, stride, steps, nbits, op, nb)

    with cf.ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    for d in decs:
        d.close()
    return out, ("reference" if use_ref else "port")
