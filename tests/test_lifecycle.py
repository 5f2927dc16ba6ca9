"""Handle lifecycle on the device: everything a handle allocates -- metrics, decision histories of every pipeline slot, staging
buffers of the host-pointer calls, the segment-parallel chainback's scratch, the windowed decode's ring, the run-time built
module of a non-harness polynomial, internal streams and events -- goes back when the handle is deleted
(delete_viterbi27_sse2 and friends free their one allocation, viterbi27_sse2.cpp:107-113; a drop-in must not do worse over the
thousands of create/delete cycles a long-running caller makes)."""
import numpy as np
import pytest
import torch

from common import frames, spec_of
from ka9q_viterbi_comparison_amd import HipViterbi, codes as C
from ka9q_viterbi_comparison_amd import VARIANT_AUTO, VARIANT_HBM, VARIANT_LDS, VARIANT_REGS, VARIANT_WAVE

pytestmark = pytest.mark.gpu


def _use(dec, code, nframes, B, syms, host):
    spec = spec_of(code)
    steps = B * 8 + spec.K - 1
    dec.reset()
    if host:
        dec.update(syms, nbits=steps)
        data, _ = dec.chainback(B * 8)
        return data
    d_syms = torch.from_numpy(syms).cuda()
    d_out = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
    dec.update(d_syms, nbits=steps)
    dec.chainback(B * 8, out=d_out)
    dec.sync()
    return d_out.cpu().numpy().reshape(nframes, B)


# (code, nframes, payload bytes, variant, pipeline depth): every allocation path of vhip_create / set_variant / set_pipeline_depth
CASES = [
    (C.KA9Q27, 1, 64, VARIANT_AUTO, 1),      # wave kernel, one frame
    (C.KA9Q27, 5000, 16, VARIANT_AUTO, 2),   # register kernel, two decision histories
    (C.SPIRAL47, 70, 32, VARIANT_LDS, 1),    # natural rows
    (C.KA9Q29, 3, 40, VARIANT_WAVE, 3),
    (C.SPIRAL49, 2100, 8, VARIANT_REGS, 2),
    (C.KA9Q615, 3, 24, VARIANT_AUTO, 2),     # segment-parallel chainback scratch per slot
    (C.SPIRAL615, 70, 8, VARIANT_AUTO, 1),
    (C.KA9Q224, 1, 8, VARIANT_AUTO, 1),      # tiled passes, pinned report words
    (C.KA9Q224, 1, 6, VARIANT_HBM, 1),
]


@pytest.mark.timeout(600)
def test_create_use_delete_returns_the_device_memory():
    torch.cuda.init()
    inputs = []
    for code, nframes, B, variant, depth in CASES:
        payload, syms = frames(code, 11, nframes, B)
        inputs.append((payload, np.ascontiguousarray(syms)))

    def cycle(check):
        for (code, nframes, B, variant, depth), (payload, syms) in zip(CASES, inputs):
            spec = spec_of(code)
            steps = B * 8 + spec.K - 1
            for host in (False, True):
                dec = HipViterbi(spec.name, steps, nframes=nframes, variant=variant, pipeline_depth=depth)
                out = _use(dec, code, nframes, B, syms, host)
                if check and spec.K != 24:  # K=24: the payload comes out of the nbits+K-1 call, not checked here
                    assert np.array_equal(out[:, :B], payload), (code, host)
                assert dec.device_bytes > 0
                dec.close()
        # a non-harness polynomial: fast kernel built (or fetched from the cache) at create time, module unloaded with the handle
        dec = HipViterbi("27", 64 + 6, nframes=3000, poly=(0o133, 0o171))
        dec.close()
        # the fused windowed decode's ring (K=15)
        dec = HipViterbi("615", 256 + 14, nframes=2, pipeline_depth=2)
        if dec.window is not None:
            d_syms = torch.zeros(2 * (256 + 14) * 6, dtype=torch.uint8, device="cuda")
            d_out = torch.zeros(2 * 32, dtype=torch.uint8, device="cuda")
            dec.decode_windowed(d_syms, 256, d_out)
            dec.sync()
        dec.close()
        torch.cuda.synchronize()

    cycle(check=True)  # first use: module loads, the JIT cache, lazily created runtime state
    cycle(check=False)
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        cycle(check=False)
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    # 12 rounds x 20 handles: one buffer of 1.4 MiB leaked per round would trip this (a K=24 handle holds 32 MiB of metrics, the
    # 5000-frame case 5 MiB of decisions per slot)
    assert free0 - free1 < 16 << 20, f"device memory not returned: {free0 - free1} bytes over 12 rounds"


def test_delete_with_work_in_flight():
    """delete while the handle's kernels are still queued: the library drains its streams first (nothing the caller enqueued
    later may see freed buffers reused)."""
    code, nframes, B = C.KA9Q29, 4000, 64
    spec = spec_of(code)
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 3, nframes, B)
    d_syms = torch.from_numpy(np.ascontiguousarray(syms)).cuda()
    d_out = torch.zeros(nframes * B, dtype=torch.uint8, device="cuda")
    for depth in (1, 2):
        dec = HipViterbi(spec.name, steps, nframes=nframes, pipeline_depth=depth)
        for _ in range(3):
            dec.reset()
            dec.update(d_syms, nbits=steps)
            dec.chainback(B * 8, out=d_out)
        dec.close()  # no sync before
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().reshape(nframes, B), payload)


@pytest.mark.timeout(600)
def test_handles_on_concurrent_host_threads():
    """One handle per host thread, all threads at once (distinct handles are independent; the reference's decoders
    share one process-global branch table, ours keep theirs per handle): blocking host-pointer calls of four different kernel
    families interleave on the device, two threads create handles for the same non-harness polynomial at the same moment (one
    run-time build, one cache entry), and every decode is the payload."""
    import threading

    jobs = [(C.KA9Q27, 1, 128, None), (C.SPIRAL47, 3000, 16, None), (C.KA9Q615, 2, 32, None), (C.KA9Q224, 1, 8, None),
            (C.KA9Q27, 2100, 16, (0o135, 0o147)), (C.KA9Q27, 2500, 8, (0o135, 0o147))]  # register kernels: built at run time
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(code, nframes, B, poly):
        try:
            spec = spec_of(code)
            steps = B * 8 + spec.K - 1
            from ka9q_viterbi_comparison_amd.decoder import gen_frames_host

            payload, syms = gen_frames_host(spec, 5 + code, 0, nframes, B, C.HARD_AMP_Q16, 0, poly=poly)
            syms = np.ascontiguousarray(syms)
            start.wait()
            for it in range(12):
                dec = HipViterbi(spec.name, steps, nframes=nframes, poly=poly)
                for _ in range(3):
                    dec.reset()
                    dec.update(syms, nbits=steps)
                    if spec.K == 24:  # the payload comes out of the nbits+K-1 call
                        data, _ = dec.chainback(steps)
                    else:
                        data, _ = dec.chainback(B * 8)
                    if not np.array_equal(data[:, :B], payload):
                        errors.append(f"code {code} x {nframes}: wrong bytes in iteration {it}")
                        return
                dec.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(f"code {code} x {nframes}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_allocation_failure_is_an_error_not_a_crash():
    """A decision history that cannot fit the card: the call that needs it fails with a message (the reference: create returns
    NULL when its posix_memalign fails, viterbi27_sse2.cpp:60-66; here the history is allocated by the first update, so that a
    handle used only for the fused windowed decode never holds one), nothing is left behind, and the next handle works."""
    from ka9q_viterbi_comparison_amd._lib import VhipError, last_error

    free0, total = torch.cuda.mem_get_info()
    for name, steps, nframes in (("29", 1 << 20, 1 << 20), ("615", 1 << 16, 1 << 18), ("27", (1 << 31) - 64, 70000)):
        try:
            dec = HipViterbi(name, steps, nframes=nframes)
        except VhipError:
            continue
        d_syms = torch.zeros(nframes * dec.R, dtype=torch.uint8, device="cuda")
        assert dec._lib.vhip_update_dev(dec._h, d_syms.data_ptr(), 1) != 0, (name, "an update without a history succeeded")
        assert "alloc" in last_error().lower() or "memory" in last_error().lower(), last_error()
        dec.close()
        del d_syms
    # a second decision history that does not fit: the call fails, the handle stays at depth 1 (vhip_set_pipeline_depth frees
    # what it had allocated for the slot)
    spec = C.CODES["29"]
    steps = 8 * 8 + spec.K - 1
    per_frame = (steps + spec.K - 1) * 32  # rows of 2^(K-1)/8 bytes
    nframes = int(free0 * 0.45 / per_frame)
    dec = HipViterbi("29", steps, nframes=nframes)
    assert dec._lib.vhip_set_pipeline_depth(dec._h, 2) != 0
    assert dec.pipeline_depth == 1
    dec.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"a failed create kept {free0 - free1} bytes"
    payload, syms = frames(C.KA9Q27, 1, 3, 16)
    dec = HipViterbi("27", 16 * 8 + 6, nframes=3)
    dec.reset()
    dec.update(np.ascontiguousarray(syms), nbits=16 * 8 + 6)
    data, _ = dec.chainback(16 * 8)
    assert np.array_equal(data, payload)
    dec.close()


def test_errors_of_other_calls_do_not_leak_into_a_decode():
    """(1) Another library's failed runtime call leaves a 'last error' in the thread; the decoder's launches must not report it as
    theirs.  (2) The decoder's own refused call -- an update beyond the handle's capacity -- is an error for that call only: the
    same handle decodes correctly afterwards."""
    import ctypes

    from ka9q_viterbi_comparison_amd._lib import VhipError

    hip = ctypes.CDLL("libamdhip64.so")
    code, nframes, B = C.SPIRAL47, 5, 24
    spec = spec_of(code)
    steps = B * 8 + spec.K - 1
    payload, syms = frames(code, 9, nframes, B)
    syms = np.ascontiguousarray(syms)
    dec = HipViterbi(spec.name, steps, nframes=nframes)

    def decode_ok():
        dec.reset()
        dec.update(syms, nbits=steps)
        data, _ = dec.chainback(B * 8)
        assert np.array_equal(data, payload)

    decode_ok()
    ptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(1 << 60)) != 0  # refused: stays behind as the thread's last error
    decode_ok()
    assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(1 << 60)) != 0
    other = HipViterbi("615", 64 + 14, nframes=2)  # create launches the metric initialisation
    other.close()
    # beyond capacity: an error, nothing written, the handle still good
    too_many = np.zeros((nframes, (steps + 40) * spec.R), dtype=np.uint8)
    dec.reset()
    with pytest.raises(VhipError):
        dec.update(too_many, nbits=steps + 40)
    assert dec.rows_written <= steps
    decode_ok()
    dec.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_one_handle_through_every_kernel_family(depth):
    """vhip_set_variant between decodes of ONE handle: the decision history, metrics and scratch of the handle serve every kernel
    family in turn (their row layouts differ; the history is sized once), in both orders, and each decode is the oracle's."""
    from ka9q_viterbi_comparison_amd._lib import check
    from oracle_lib import OracleDecoder

    for code, nframes, B, variants in (
        (C.KA9Q27, 3, 40, [VARIANT_WAVE, VARIANT_REGS, VARIANT_LDS, VARIANT_REGS | (2 << 8), VARIANT_WAVE, VARIANT_LDS, VARIANT_AUTO]),
        (C.SPIRAL47, 70, 24, [VARIANT_REGS, VARIANT_WAVE, VARIANT_LDS, VARIANT_REGS | (3 << 8), VARIANT_AUTO]),
        (C.KA9Q29, 5, 32, [VARIANT_LDS, VARIANT_WAVE, VARIANT_REGS, VARIANT_WAVE]),
        (C.SPIRAL49, 2, 16, [VARIANT_WAVE, VARIANT_LDS, VARIANT_REGS, VARIANT_AUTO]),
        (C.KA9Q615, 2, 16, [VARIANT_REGS, VARIANT_LDS, VARIANT_REGS]),
    ):
        spec = spec_of(code)
        steps = B * 8 + spec.K - 1
        _, syms = frames(code, 50 + code, nframes, B, ebn0_db=spec.ebn0_db)
        syms = np.ascontiguousarray(syms[:, :steps * spec.R])
        refs = []
        for f in range(nframes):
            o = OracleDecoder(code, spec.poly, steps)
            o.update(syms[f], steps)
            refs.append((o.chainback(B * 8)[0], o.metrics(), o.rows(steps)))
            o.close()
        dec = HipViterbi(spec.name, steps, nframes=nframes, pipeline_depth=depth)
        for v in variants:
            dec.reset()
            check(dec._lib.vhip_set_variant(dec._h, v), "vhip_set_variant")
            dec.update(syms, nbits=steps)
            data, _ = dec.chainback(B * 8)
            for f in range(nframes):
                assert np.array_equal(data[f], refs[f][0]), (code, v, f)
                assert np.array_equal(dec.metrics(f), refs[f][1]), (code, v, f)
                assert np.array_equal(dec.decision_rows(f, 0, steps), refs[f][2]), (code, v, f)
        dec.close()


@pytest.mark.parametrize("name,variants", [("27", (0, 1, 2, 6)), ("47", (0, 2, 6)), ("29", (0, 1, 2, 6)), ("49", (0, 6)), ("615", (0, 1)), ("224", (0,))])
def test_chainback_beyond_what_was_fed(name, variants):
    """A chainback over more bits than steps were fed since the last init -- or than the handle can hold -- is a caller's bug the
    reference answers with whatever its decision array holds (stale rows of the previous frame, or memory past the array,
    viterbi27_sse2.cpp:97-103).  Here it is safe and defined: rows never written read as zero decisions, nothing is read
    outside the history, no error is raised, and the handle decodes correctly afterwards."""
    spec = C.CODES[name]
    for v in variants:
        for nframes in ((1,) if spec.K == 24 else (1, 70)):
            B = 4 if spec.K == 24 else 40
            steps = B * 8 + spec.K - 1
            payload, syms = frames(spec.code, 3, nframes, B)
            syms = np.ascontiguousarray(syms)
            nb = steps if spec.K == 24 else B * 8
            dec = HipViterbi(name, steps, nframes=nframes, variant=v)
            dec.reset()
            dec.update(syms, nbits=steps)
            first, _ = dec.chainback(nb)
            assert np.array_equal(first[:, :B], payload)
            dec.reset()
            dec.update(np.ascontiguousarray(syms[:, :16 * spec.R]), nbits=16)
            dec.chainback(nb)                                          # more bits than steps fed since the init
            dec.chainback(steps if spec.K == 24 else B * 8 + 5000)     # and more than the handle holds
            assert dec.status == 0
            dec.reset()
            dec.update(syms, nbits=steps)
            again, _ = dec.chainback(nb)
            assert np.array_equal(again, first), (name, v, nframes)
            dec.close()
