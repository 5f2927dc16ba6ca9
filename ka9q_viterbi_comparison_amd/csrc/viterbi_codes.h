// viterbi_codes.h -- arithmetic "families" of the six decoders on the hot path, shared by host and device.
//
// Each traits struct states, with the reference line it mirrors, everything an ACS kernel must reproduce
// bit-for-bit: metric type and initial values, branch-metric formula, add flavour, tie-break, survivor and
// renormalisation rule.
#pragma once
#include <stdint.h>

#include "../../include/viterbi_hip.h"

#if defined(__HIPCC__)
#define VH_HD __host__ __device__ __forceinline__
#else
#define VH_HD inline
#endif

namespace vh {

enum Metric { U8MOD = 0, U8SAT = 1, I16SAT = 2 };

// ka9q K=7/9 r=1/2: viterbi27_sse2.cpp:46-52 (init 63 / 0), :137-158 (butterfly), no renormalisation.
template <int K_>
struct Ka9qU8Mod {
    static constexpr int K = K_, R = 2;
    static constexpr Metric metric = U8MOD;
    static constexpr int init_all = 63, init_start = 0;
    static constexpr int bm_comp = 15;
    static constexpr bool tie_upper = false;
    static constexpr bool renorm = false;
    static constexpr int renorm_thr = 0;
    static constexpr bool incremental = true;
    // t = ((a0+a1+1)>>1)>>4, a_r = s_r ^ (class bit r ? 255 : 0)            viterbi27_sse2.cpp:137-146
    static VH_HD int bm(const int *s, unsigned c) {
        int a0 = s[0] ^ ((c & 1u) ? 255 : 0), a1 = s[1] ^ ((c & 2u) ? 255 : 0);
        return ((a0 + a1 + 1) >> 1) >> 4;
    }
    static VH_HD int bm_tc(int t) { return 15 - t; }  // m_metric = 15 - metric          viterbi27_sse2.cpp:146
};

// ka9q K=15 r=1/6: viterbi615_sse2.cpp:33-39, :132-148 (sum of six, adds_epi16, min + cmpeq),
// renormalise when new[0] >= SHRT_MAX-12750 (:160-183).
struct Ka9q615 {
    static constexpr int K = 15, R = 6;
    static constexpr Metric metric = I16SAT;
    static constexpr int init_all = -32768 + 1000, init_start = -32768;
    static constexpr int bm_comp = 1530;
    static constexpr bool tie_upper = true;
    static constexpr bool renorm = true;
    static constexpr int renorm_thr = 32767 - 12750;
    static constexpr bool incremental = true;
    static VH_HD int bm(const int *s, unsigned c) {
        int t = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) t += s[r] ^ (((c >> r) & 1u) ? 255 : 0);
        return t;
    }
    static VH_HD int bm_tc(int t) { return 1530 - t; }
};

// ka9q K=24 r=1/2: viterbi224_sse2.cpp:39-45, :159-194 (cmpgt before min: tie -> lower), renorm >= 25000 (:226).
struct Ka9q224 {
    static constexpr int K = 24, R = 2;
    static constexpr Metric metric = I16SAT;
    static constexpr int init_all = -32768 + 5000, init_start = -32768;
    static constexpr int bm_comp = 510;
    static constexpr bool tie_upper = false;
    static constexpr bool renorm = true;
    static constexpr int renorm_thr = 25000;
    static constexpr bool incremental = true;
    static VH_HD int bm(const int *s, unsigned c) {
        return (s[0] ^ ((c & 1u) ? 255 : 0)) + (s[1] ^ ((c & 2u) ? 255 : 0));
    }
    static VH_HD int bm_tc(int t) { return 510 - t; }
};

// spiral K=7/9 r=1/4: spiral47.cpp:54-61, :164-227 (6-bit terms, adds_epu8, min_epu8 + cmpeq: tie -> upper),
// renormalise when new[0] > 126 (47, :313) / > 103 (49, spiral49.cpp:790) by subs_epu8 of the minimum.
template <int K_, int THR_>
struct SpiralR4 {
    static constexpr int K = K_, R = 4;
    static constexpr Metric metric = U8SAT;
    static constexpr int init_all = 63, init_start = 0;
    static constexpr int bm_comp = 63;
    static constexpr bool tie_upper = true;
    static constexpr bool renorm = true;
    static constexpr int renorm_thr = THR_;
    static constexpr bool incremental = false;
    static VH_HD int bm(const int *s, unsigned c) {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) q += ((s[r] ^ (((c >> r) & 1u) ? 255 : 0)) >> 2) & 63;  // <= 252: never saturates
        return (q >> 2) & 63;
    }
    static VH_HD int bm_tc(int t) { return 63 - t; }
};

// spiral K=7/9 r=1/2: spiral27.cpp:164-173 (avg_epu8, >>2, &63), renormalise when new[0] > 210 (spiral27.cpp:236,
// spiral29.cpp:507); otherwise as SpiralR4.
template <int K_>
struct SpiralR2 {
    static constexpr int K = K_, R = 2;
    static constexpr Metric metric = U8SAT;
    static constexpr int init_all = 63, init_start = 0;
    static constexpr int bm_comp = 63;
    static constexpr bool tie_upper = true;
    static constexpr bool renorm = true;
    static constexpr int renorm_thr = 210;
    static constexpr bool incremental = false;
    static VH_HD int bm(const int *s, unsigned c) {
        int a0 = s[0] ^ ((c & 1u) ? 255 : 0), a1 = s[1] ^ ((c & 2u) ? 255 : 0);
        return (((a0 + a1 + 1) >> 1) >> 2) & 63;
    }
    static VH_HD int bm_tc(int t) { return 63 - t; }
};

// spiral K=15 r=1/6: spiral615.cpp:149-242 (six 6-bit terms through adds_epu8: the sum saturates at 255; >>2, &63),
// t' = subs_epu8(94, t) (clamps at 0), scalar renormalize(.., 74) (spiral615.cpp:31-40,269,408).
struct Spiral615 {
    static constexpr int K = 15, R = 6;
    static constexpr Metric metric = U8SAT;
    static constexpr int init_all = 63, init_start = 0;
    static constexpr int bm_comp = 94;
    static constexpr bool tie_upper = true;
    static constexpr bool renorm = true;
    static constexpr int renorm_thr = 74;
    static constexpr bool incremental = false;
    static VH_HD int bm(const int *s, unsigned c) {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) q += ((s[r] ^ (((c >> r) & 1u) ? 255 : 0)) >> 2) & 63;
        q = q > 255 ? 255 : q;  // every partial sum of adds_epu8 saturates; the terms are non-negative
        return (q >> 2) & 63;
    }
    static VH_HD int bm_tc(int t) { return 94 - t < 0 ? 0 : 94 - t; }
};

using Code27 = Ka9qU8Mod<7>;
using Code29 = Ka9qU8Mod<9>;
using Code615 = Ka9q615;
using Code224 = Ka9q224;
using Code47 = SpiralR4<7, 126>;
using Code49 = SpiralR4<9, 103>;
using CodeS27 = SpiralR2<7>;
using CodeS29 = SpiralR2<9>;
using CodeS615 = Spiral615;

struct CodeInfo {
    int K, R;
    int incremental;
};
static inline CodeInfo code_info(int code) {
    switch (code) {
    case VHIP_KA9Q27: return {7, 2, 1};
    case VHIP_KA9Q29: return {9, 2, 1};
    case VHIP_KA9Q615: return {15, 6, 1};
    case VHIP_KA9Q224: return {24, 2, 1};
    case VHIP_SPIRAL47: return {7, 4, 0};
    case VHIP_SPIRAL49: return {9, 4, 0};
    case VHIP_SPIRAL27: return {7, 2, 0};
    case VHIP_SPIRAL29: return {9, 2, 0};
    case VHIP_SPIRAL615: return {15, 6, 0};
    }
    return {0, 0, 0};
}

// parity of the set bits of x (src/parity.h:46-55 folds to a byte and looks up a table)
static VH_HD unsigned parity_u32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__popc(x) & 1u;
#else
    return (unsigned)__builtin_popcount(x) & 1u;
#endif
}

// branch-table class of butterfly j: bit r = parity((2j) & poly[r])      viterbi27_sse2.cpp:64-67
template <int R>
static VH_HD unsigned bt_class(uint32_t j, const int *poly) {
    unsigned c = 0;
#pragma unroll
    for (int r = 0; r < R; r++) c |= parity_u32((2u * j) & (uint32_t)poly[r]) << r;
    return c;
}

}  // namespace vh
