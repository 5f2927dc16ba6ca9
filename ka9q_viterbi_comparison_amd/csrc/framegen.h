// framegen.h -- synthetic frame generator shared by host and device (bit-identical on both).
//
// Stands in for the reference's test-input path (src/util.h:8-12 generate_random_bytes with an unseeded
// std::rand, src/util.h:14-62 encode_data through the un-vendored ConvolutionalEncoder, and the ka9q
// offset-binary mapping src/viterbi_configs.h:15-20).  The reference has no noise source; BASELINE.json
// asks for AWGN soft symbols, so this generator adds an integer-only approximately-Gaussian term:
// Irwin-Hall sum of eight 16-bit uniforms (exactly reproducible on CPU and GPU; after the 8-bit clamp its
// bounded +-4.9 sigma tails are indistinguishable from a true normal).
//
// Encoder convention: sr = (sr<<1)|bit, coded bit r = parity(sr & poly[r]), payload
// bytes consumed MSB-first, K-1 zero tail bits, symbols step-major.
#pragma once
#include <stdint.h>

#include "viterbi_codes.h"

namespace vh {

static VH_HD uint64_t fg_mix64(uint64_t z) {  // splitmix64 output function
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static VH_HD uint64_t fg_frame_key(uint64_t seed, uint64_t frame) { return fg_mix64(seed ^ fg_mix64(frame * 0xd6e8feb86659fd93ull)); }
static VH_HD unsigned fg_payload_byte(uint64_t key, uint32_t i) {
    uint64_t w = fg_mix64(key + 0x8000000000000000ull + (uint64_t)(i >> 3));
    return (unsigned)(w >> (8 * (i & 7))) & 0xffu;
}
// centred Irwin-Hall(8) sample, scaled by 2: integer in [-524280, 524280], sigma = 107020.0
static VH_HD int32_t fg_noise_c2(uint64_t key, uint32_t sym_index) {
    uint64_t a = fg_mix64(key + 2ull * sym_index + 1ull);
    uint64_t b = fg_mix64(key + 2ull * sym_index + 2ull);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s += (uint32_t)(a >> (16 * k)) & 0xffffu;
        s += (uint32_t)(b >> (16 * k)) & 0xffffu;
    }
    return (int32_t)(2u * s) - 8 * 65535;
}
#define VH_FG_SIGMA_C2 107020.0
// soft symbol for coded bit c: clamp(round(127.5 + amp*(2c-1) + noise), 0, 255), all in Q16
static VH_HD unsigned fg_symbol(unsigned c, int amp_q16, int noise_q12, int32_t c2) {
    int64_t v = ((int64_t)255 << 15) + (c ? (int64_t)amp_q16 : -(int64_t)amp_q16);
    v += ((int64_t)c2 * (int64_t)noise_q12) >> 12;
    v = (v + 32768) >> 16;
    return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

}  // namespace vh
