"""Shared helpers for the parity tests: standard codes, synthetic frames, reference-call conventions."""
import numpy as np

from ka9q_viterbi_comparison_amd import codes as C
from ka9q_viterbi_comparison_amd.decoder import gen_frames_host, noise_q12

# oracle code id -> (K, R, poly, hot-path spec name)
ORACLE_CODES = {
    C.KA9Q27: ("27", C.CODES["27"]),
    C.KA9Q29: ("29", C.CODES["29"]),
    C.KA9Q615: ("615", C.CODES["615"]),
    C.KA9Q224: ("224", C.CODES["224"]),
    C.SPIRAL47: ("47", C.CODES["47"]),
    C.SPIRAL49: ("49", C.CODES["49"]),
    C.SPIRAL27: ("spiral27", C.CODES["spiral27"]),
    C.SPIRAL29: ("spiral29", C.CODES["spiral29"]),
    C.SPIRAL615: ("spiral615", C.CODES["spiral615"]),
}
GPU_CODES = [C.KA9Q27, C.KA9Q29, C.KA9Q615, C.KA9Q224, C.SPIRAL47, C.SPIRAL49, C.SPIRAL27, C.SPIRAL29, C.SPIRAL615]


def spec_of(code):
    return ORACLE_CODES[code][1]


def frames(code, seed, nframes, payload_bytes, ebn0_db=None, amp=C.SOFT_AMP, frame0=0):
    """(payload[nframes, B], syms[nframes, steps*R]).  ebn0_db=None -> the reference's hard 0/255 symbols."""
    spec = spec_of(code)
    if ebn0_db is None:
        return gen_frames_host(spec, seed, frame0, nframes, payload_bytes, C.HARD_AMP_Q16, 0)
    nq = noise_q12(spec.R, amp, ebn0_db)
    return gen_frames_host(spec, seed, frame0, nframes, payload_bytes, int(amp * 65536), nq)


def bit_errors(a, b):
    return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())
