"""The six (K, R) codes on the hot path with the polynomials and frame sizes the reference harness uses.

Source: src/main.cpp:363-419 (one block per code: K, R, total_input_bytes, poly).
"""
from collections import namedtuple

# code ids shared by include/viterbi_hip.h (enum vhip_code) and oracle/viterbi_oracle.h (enum vo_code)
KA9Q27, KA9Q29, KA9Q615, KA9Q224, SPIRAL47, SPIRAL49 = range(6)
# spiral arithmetic for the codes whose primary GPU parity target is ka9q
SPIRAL27, SPIRAL29, SPIRAL615 = 6, 7, 8

CodeSpec = namedtuple("CodeSpec", "name code K R poly ref_payload_bytes family ebn0_db")

CODES = {
    # name       code      K   R   poly (src/main.cpp)                                  bytes  arithmetic      Eb/N0 for AWGN tests
    "27": CodeSpec("27", KA9Q27, 7, 2, (0x6D, 0x4F), 1024, "ka9q-u8-mod", 4.0),                       # main.cpp:364-371
    "47": CodeSpec("47", SPIRAL47, 7, 4, (121, 117, 91, 111), 1024, "spiral-u8-sat", 2.0),            # main.cpp:373-380
    "29": CodeSpec("29", KA9Q29, 9, 2, (0x1AF, 0x11D), 512, "ka9q-u8-mod", 4.0),                      # main.cpp:382-390
    "49": CodeSpec("49", SPIRAL49, 9, 4, (501, 441, 331, 315), 512, "spiral-u8-sat", 2.0),            # main.cpp:392-399
    "615": CodeSpec("615", KA9Q615, 15, 6, (0o42631, 0o47245, 0o56507, 0o73363, 0o77267, 0o64537), 256, "ka9q-i16-sat", 1.0),  # main.cpp:401-409
    "224": CodeSpec("224", KA9Q224, 24, 2, (0o62650457, 0o62650455), 8, "ka9q-i16-sat", 4.0),         # main.cpp:411-418
}
# the remaining spiral arithmetic variants of the same codes (src/main.cpp:370,388,408: test_spiral<...> lines)
CODES["spiral27"] = CodeSpec("spiral27", SPIRAL27, 7, 2, (0x6D, 0x4F), 1024, "spiral-u8-sat", 4.0)
CODES["spiral29"] = CodeSpec("spiral29", SPIRAL29, 9, 2, (0x1AF, 0x11D), 512, "spiral-u8-sat", 4.0)
CODES["spiral615"] = CodeSpec("spiral615", SPIRAL615, 15, 6, (0o42631, 0o47245, 0o56507, 0o73363, 0o77267, 0o64537), 256, "spiral-u8-sat", 1.0)
BY_ID = {c.code: c for c in CODES.values()}

# soft-symbol constants: ka9q offset binary 0..255 (src/viterbi_configs.h:15-20)
HARD_AMP_Q16 = int(127.5 * 65536)   # with noise_q12 = 0 reproduces the reference's 0 / 255 symbols (src/util.h:36)
SOFT_AMP = 64.0                     # AWGN operating amplitude
SOFT_AMP_Q16 = int(SOFT_AMP * 65536)


def steps_for(spec, payload_bytes):
    """Trellis steps incl. tail: total_transmit_bits of src/main.cpp:126-129."""
    return payload_bytes * 8 + spec.K - 1
