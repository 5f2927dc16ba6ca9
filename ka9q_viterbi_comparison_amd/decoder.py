"""Host-side mirror of the reference decoder adapter, over the C ABI.

`HipViterbi` has the shape of `ka9q_viterbi_interface` (src/ka9q_interface.h:21-56): constructed from
(poly, transmit_bits), then `reset()`, `update(symbols)`, `chainback(total_bits)`; with an extra `nframes`
so one handle decodes a batch of independent frames.  Every call goes through libviterbi_hip.so; numpy (host
pointers, blocking) and torch CUDA tensors (device pointers, asynchronous on the current stream) are accepted.
"""
import ctypes as C

import numpy as np

from . import _lib
from .codes import BY_ID, CODES

VARIANT_AUTO, VARIANT_LDS, VARIANT_REGS, VARIANT_HBM, VARIANT_HBM_FUSED, VARIANT_HBM_TILED, VARIANT_WAVE = 0, 1, 2, 3, 4, 5, 6


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class HipViterbi:
    def __init__(self, code, transmit_bits, nframes=1, poly=None, variant=VARIANT_AUTO, stream=None, pipeline_depth=1):
        spec = CODES[code] if isinstance(code, str) else BY_ID[code]
        self.spec = spec
        self.K, self.R = spec.K, spec.R
        self.poly = tuple(poly) if poly is not None else spec.poly
        self.transmit_bits = int(transmit_bits)
        self.nframes = int(nframes)
        self._lib = _lib.load()
        arr = (C.c_int * len(self.poly))(*self.poly)
        self._h = self._lib.vhip_create(spec.code, arr, self.transmit_bits, self.nframes)
        if not self._h:
            raise _lib.VhipError(f"vhip_create failed: {_lib.last_error()}")
        if variant != VARIANT_AUTO:
            _lib.check(self._lib.vhip_set_variant(self._h, variant), "vhip_set_variant")
        if stream is not None:
            self.set_stream(stream)
        if pipeline_depth != 1:
            _lib.check(self._lib.vhip_set_pipeline_depth(self._h, pipeline_depth), "vhip_set_pipeline_depth")

    # -- lifetime ----------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.vhip_delete(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing ----------------------------------------------------------------------------------
    def set_stream(self, stream):
        """stream: int hipStream_t handle (torch.cuda.current_stream().cuda_stream) or None."""
        _lib.check(self._lib.vhip_set_stream(self._h, C.c_void_p(stream or 0)), "vhip_set_stream")

    def sync(self):
        _lib.check(self._lib.vhip_sync(self._h), "vhip_sync")

    def join(self):
        """Pipelined handles: order everything in flight before what the caller enqueues next on the handle's stream."""
        _lib.check(self._lib.vhip_join(self._h), "vhip_join")

    def set_chainback_segments(self, seg_bits=-1, warmup_rows=-1):
        """Geometry of the segment-parallel chainback (K >= 15, <= 64 frames); 0 = one walk per frame, < 0 = defaults."""
        _lib.check(self._lib.vhip_set_chainback_segments(self._h, seg_bits, warmup_rows), "vhip_set_chainback_segments")

    def chainback_rewalked(self):
        """(segments walked twice, segments per frame) of the last chainback; (0, 0) if it ran as one walk."""
        n = C.c_int(0)
        r = self._lib.vhip_chainback_rewalked(self._h, C.byref(n))
        _lib.check(min(r, 0), "vhip_chainback_rewalked")
        return r, n.value

    def enable_timing(self, on=True):
        _lib.check(self._lib.vhip_enable_timing(self._h, int(on)), "vhip_enable_timing")

    def read_timing(self):
        """Waits for the handle to go idle; -> (update_ms_sum, n_update, chainback_ms_sum, n_chainback) since the last read."""
        su, sc, nu, nc = C.c_double(0), C.c_double(0), C.c_int(0), C.c_int(0)
        _lib.check(self._lib.vhip_read_timing(self._h, C.byref(su), C.byref(nu), C.byref(sc), C.byref(nc)), "vhip_read_timing")
        return su.value, nu.value, sc.value, nc.value

    @property
    def pipeline_depth(self):
        return self._lib.vhip_get_pipeline_depth(self._h)

    @property
    def status(self):
        return self._lib.vhip_status(self._h)

    @property
    def variant(self):
        return self._lib.vhip_get_variant(self._h)

    @property
    def runtime_specialised(self):
        """True when the handle runs fast kernels compiled at create time for its (non-harness) polynomials."""
        return self._lib.vhip_is_runtime_specialised(self._h) == 1

    @property
    def rows_written(self):
        return self._lib.vhip_rows_written(self._h)

    @property
    def device_bytes(self):
        return self._lib.vhip_device_bytes(self._h)

    # -- the adapter's three calls (src/ka9q_interface.h:45-55) -------------------------------------
    def reset(self, starting_state=0):
        _lib.check(self._lib.vhip_init(self._h, starting_state), "vhip_init")

    def update(self, symbols, nbits=None):
        """symbols: uint8, nframes*nbits*R values frame-major (numpy -> blocking; torch cuda -> async)."""
        total = symbols.numel() if _is_torch(symbols) else symbols.size
        if nbits is None:
            assert total % (self.R * self.nframes) == 0  # src/ka9q_interface.h:49
            nbits = total // (self.R * self.nframes)
        assert total >= self.nframes * nbits * self.R
        if _is_torch(symbols):
            assert symbols.is_cuda and symbols.is_contiguous() and symbols.element_size() == 1
            _lib.check(self._lib.vhip_update_dev(self._h, C.c_void_p(symbols.data_ptr()), nbits), "vhip_update_dev")
        else:
            s = np.ascontiguousarray(symbols, dtype=np.uint8)
            _lib.check(self._lib.vhip_update(self._h, s.ctypes.data_as(C.c_void_p), nbits), "vhip_update")

    def chainback(self, total_bits, endstate=0, out=None):
        """Returns (data, rc): decoded bytes [nframes, ceil(total_bits/8)], MSB-first."""
        nbytes = (total_bits + 7) // 8
        if out is not None and _is_torch(out):
            assert out.is_cuda and out.is_contiguous() and out.numel() >= self.nframes * nbytes
            rc = self._lib.vhip_chainback_dev(self._h, C.c_void_p(out.data_ptr()), total_bits, endstate)
            _lib.check(rc, "vhip_chainback_dev")
            return out, rc
        data = np.zeros((self.nframes, nbytes), dtype=np.uint8)  # caller pre-zeroes in the reference (main.cpp:262)
        rc = self._lib.vhip_chainback(self._h, data.ctypes.data_as(C.c_void_p), total_bits, endstate)
        # ka9q615 with one frame legitimately returns a (usually negative) path metric (viterbi615_sse2.cpp:90), so the
        # error status travels separately (vhip_status)
        if self._lib.vhip_status(self._h) != 0:
            raise _lib.VhipError(f"vhip_chainback failed: {_lib.last_error()}")
        return data, rc

    # -- fused sliding-window decode (no decision history in HBM) ------------------------------------------
    @property
    def window(self):
        """(traceback depth, block bits) of decode_windowed, or None when the code has no fused decode."""
        d = self._lib.vhip_window_depth(self._h)
        return (d, self._lib.vhip_window_block(self._h)) if d >= 0 else None

    def decode_windowed(self, symbols, total_bits, out):
        """symbols / out: torch cuda uint8 tensors (nframes x (total_bits+K-1)*R symbols in, nframes x ceil(total_bits/8) bytes out)."""
        assert symbols.is_cuda and symbols.is_contiguous() and out.is_cuda and out.is_contiguous()
        assert symbols.numel() >= self.nframes * (total_bits + self.K - 1) * self.R and out.numel() >= self.nframes * ((total_bits + 7) // 8)
        _lib.check(self._lib.vhip_decode_windowed_dev(self._h, C.c_void_p(symbols.data_ptr()), total_bits, C.c_void_p(out.data_ptr())),
                   "vhip_decode_windowed_dev")
        return out

    # -- introspection for parity tests --------------------------------------------------------------
    def decision_rows(self, frame, row0, nrows):
        n = 1 << (self.K - 1)
        out = np.zeros((nrows, n // 8), dtype=np.uint8)
        _lib.check(self._lib.vhip_read_decision_rows(self._h, frame, row0, nrows, out.ctypes.data_as(C.c_void_p)),
                   "vhip_read_decision_rows")
        return out

    def metrics(self, frame=0):
        out = np.zeros(1 << (self.K - 1), dtype=np.int32)
        _lib.check(self._lib.vhip_read_metrics(self._h, frame, out.ctypes.data_as(C.c_void_p)), "vhip_read_metrics")
        return out


# ---------------------------------------------------------------------------------------------------------
# synthetic frames (host twin; the device twin is vhip_gen_frames_dev)
def noise_q12(R, amp, ebn0_db):
    return _lib.load().vhip_noise_q12_from_ebn0(R, float(amp), float(ebn0_db))


def gen_frames_host(spec, seed, frame0, nframes, payload_bytes, amp_q16, noise_q12_val, poly=None):
    """Returns (payload[nframes, payload_bytes], syms[nframes, steps*R]) as numpy uint8."""
    lib = _lib.load()
    poly = tuple(poly) if poly is not None else spec.poly
    steps = payload_bytes * 8 + spec.K - 1
    payload = np.zeros((nframes, payload_bytes), dtype=np.uint8)
    syms = np.zeros((nframes, steps * spec.R), dtype=np.uint8)
    arr = (C.c_int * len(poly))(*poly)
    _lib.check(lib.vhip_gen_frames_host(spec.K, spec.R, arr, seed, frame0, nframes, payload_bytes, amp_q16, noise_q12_val,
                                        payload.ctypes.data_as(C.c_void_p), syms.ctypes.data_as(C.c_void_p)),
               "vhip_gen_frames_host")
    return payload, syms


def gen_frames_dev(spec, seed, frame0, nframes, payload_bytes, amp_q16, noise_q12_val, d_payload, d_syms, stream=0, poly=None):
    """Fill torch cuda uint8 tensors d_payload [nframes*payload_bytes] / d_syms [nframes*steps*R] on `stream`."""
    lib = _lib.load()
    poly = tuple(poly) if poly is not None else spec.poly
    arr = (C.c_int * len(poly))(*poly)
    pp = C.c_void_p(d_payload.data_ptr()) if d_payload is not None else C.c_void_p(0)
    sp = C.c_void_p(d_syms.data_ptr()) if d_syms is not None else C.c_void_p(0)
    _lib.check(lib.vhip_gen_frames_dev(spec.K, spec.R, arr, seed, frame0, nframes, payload_bytes, amp_q16, noise_q12_val,
                                       pp, sp, C.c_void_p(stream or 0)), "vhip_gen_frames_dev")


def count_bit_errors_dev(a, b, nbytes, stream=0):
    return _lib.check(_lib.load().vhip_count_bit_errors_dev(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nbytes,
                                                            C.c_void_p(stream or 0)), "vhip_count_bit_errors_dev")
