"""MI355X-native Viterbi decode backend (ACS update + chainback) behind the ka9q/spiral decoder API.

Only what the hot path needs lives here: `csrc/` (hand-written HIP kernels for gfx950 + the C ABI declared in
include/viterbi_hip.h), `_lib.py` (ctypes binding), `decoder.py` (host mirror of src/ka9q_interface.h) and
`codes.py` (the polynomials / frame sizes of src/main.cpp).  Importing the package does not load the HIP
library; the first decoder does, and raises if it has not been built.
"""
from .codes import CODES, BY_ID  # noqa: F401
from .decoder import (  # noqa: F401
    HipViterbi,
    VARIANT_AUTO,
    VARIANT_LDS,
    VARIANT_REGS,
    VARIANT_HBM,
    VARIANT_HBM_FUSED,
    VARIANT_HBM_TILED,
    VARIANT_WAVE,
    gen_frames_host,
    gen_frames_dev,
    count_bit_errors_dev,
    noise_q12,
)
