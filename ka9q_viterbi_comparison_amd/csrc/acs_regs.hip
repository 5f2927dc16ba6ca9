// acs_regs.hip -- production ACS update kernel for K <= 9: frames across lanes, path metrics packed in VGPRs.
//
// Replaces update_viterbi27_blk_sse2 / update_viterbi29_blk_sse2 (ka9q_libfec_port/viterbi27_sse2.cpp:119-175,
// viterbi29_sse2.cpp:108-164) and update_spiral47 / update_spiral49 (spiral/spiral47.cpp:131-538,
// spiral49.cpp:132-1540) for the polynomials the harness uses (src/main.cpp:367,377,386,396).
//
// Design (MI355X-first, not a translation of the SSE2 butterflies):
//  * L = 2^LB adjacent lanes own one frame (L = 1, 2 or 4, picked so the batch fills >= 2 waves per SIMD);
//    a wave therefore advances 64/L independent frames per instruction and there is no LDS, no barrier.
//  * The N = 2^(K-1) metrics of a frame are 16-bit fields packed two per VGPR (v_pk_* integer ops):
//    N/(2L) registers per lane.  u8-modular metrics (ka9q) sit in the HIGH byte of each field so that the
//    16-bit wrap of v_pk_add_u16 IS the mod-256 wrap of _mm_add_epi8 and the sign of a 16-bit difference IS
//    the sign of the 8-bit one; u8-saturating metrics (spiral) are the signed field (m<<7)|0x7f so that the clamp of
//    v_pk_add_i16 ... clamp IS _mm_adds_epu8 (255 <-> 0x7fff) even for a negative addend: the minimum a renormalisation
//    (spiral47.cpp:313-331) subtracts is owed to the next step's branch metrics instead (RegsStep::run, `pend`).
//  * In-place trellis with a rotating index map: new[2j] is written where old[j] lived and new[2j+1] where
//    old[j+H] lived, so position p holds state rotl^t(p) before step t.  No metric ever moves between
//    registers; the butterfly partner of step t differs in position bit (K-2 - t mod (K-1)), which is a
//    register-index bit (pure in-lane), the half bit (one v_perm), or a lane bit (one DPP quad_perm move).
//    The K-1 phases are fully unrolled, so every register index and every branch-table class is a
//    compile-time constant; the branch metric of a butterfly is a static pick from 2^R per-step values.
//  * Decisions: min(positive-part, 1<<k) turns a packed ACS by-product into bit k / bit 16+k, OR-ed into
//    accumulators; one 32-bit word per 16 registers leaves per step, laid out [group][row][word][lane] so a
//    wave store is 256 contiguous bytes.  the chainback kernels below walk exactly this layout.
//
// Algorithmic HBM bytes per frame-step are those of the reference layout: R symbol bytes in, N/8 decision
// bytes out; the metrics never leave the register file during a launch.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// polynomials of src/main.cpp:367,377,386,396
struct Poly27 { static constexpr int v[2] = {0x6d, 0x4f}; };
struct Poly47 { static constexpr int v[4] = {121, 117, 91, 111}; };
struct Poly29 { static constexpr int v[2] = {0x1af, 0x11d}; };
struct Poly49 { static constexpr int v[4] = {501, 441, 331, 315}; };

constexpr unsigned popcnt_c(unsigned x) {
    unsigned n = 0;
    while (x) {
        n += x & 1u;
        x >>= 1;
    }
    return n;
}
template <int NB>
constexpr unsigned rotl_c(unsigned x, int s) {
    s %= NB;
    return s == 0 ? x : (((x << s) | (x >> (NB - s))) & ((1u << NB) - 1u));
}
// branch-table class of state value j: bit r = parity((2j) & poly[r])   (viterbi27_sse2.cpp:64-67)
template <class P, int R>
constexpr unsigned cls_c(unsigned j) {
    unsigned c = 0;
    for (int r = 0; r < R; r++) c |= (popcnt_c((2u * j) & (unsigned)P::v[r]) & 1u) << r;
    return c;
}

__device__ __forceinline__ unsigned as_u32(u16x2 x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ u16x2 as_v(unsigned x) { return __builtin_bit_cast(u16x2, x); }

template <class C, class P, int LB>
struct RegsCfg {
    static constexpr int K = C::K, R = C::R, NB = K - 1;
    static constexpr int N = 1 << NB, L = 1 << LB, FPW = 64 / L;
    static constexpr int NR = N / (2 * L);          // packed registers per lane
    static constexpr int NC = 1 << R;               // branch-table classes
    static constexpr int NRW = NR < 16 ? NR : 16;   // registers per decision word
    static constexpr int DW = NR / NRW;             // decision words per lane per row
    static constexpr int WBYTES = NRW == 16 ? 4 : 2;
    static constexpr int SW = NB * R / 4;           // symbol dwords per period of NB steps
    static constexpr bool SAT = C::metric == U8SAT;
    static_assert((NB * R) % 4 == 0 && NR >= 8 && (NR % 8) == 0, "unsupported geometry");
};

// packed metric field <-> natural value.  Modular family: m << 8 (the byte wraps with the field).  Saturating family: the
// SIGNED field (m << 7) | 0x7f -- v_pk_add_i16 clamp saturates where adds_epu8 does (255 <-> 0x7fff) and the addend may be
// negative, which lets the renormalisation's subtract ride in the next step's branch metrics (RegsStep::run, `pend`).
template <bool SAT>
__device__ __forceinline__ constexpr unsigned field_from(int m) { return SAT ? (((unsigned)m << 7) | 0x7fu) : ((unsigned)m << 8); }
template <bool SAT>
__device__ __forceinline__ int field_to(unsigned f) { return (int)((f >> (SAT ? 7 : 8)) & 0xffu); }

template <bool SAT>
__device__ __forceinline__ u16x2 madd(u16x2 a, u16x2 t) {
    if constexpr (SAT) return (u16x2)__builtin_elementwise_add_sat((i16x2)a, (i16x2)t);  // v_pk_add_i16 clamp == adds_epu8 on bits 7..14
    else return a + t;                                                                    // v_pk_add_u16 wrap  == add_epi8  on the high byte
}

// One packed add-compare-select: two new states.  `acc` collects decision bit k (low field) / 16+k (high field);
// for the saturating family the collected bit is the COMPLEMENT of the decision (fixed up once per word).
template <bool SAT, int KBIT>
__device__ __forceinline__ u16x2 acs_pk(u16x2 lower, u16x2 upper, unsigned &acc) {
    static_assert(SAT ? KBIT <= 7 : (KBIT >= 1 && KBIT <= 8), "the collected bit must not exceed the smallest non-zero difference");
    const u16x2 one = {(unsigned short)(1u << KBIT), (unsigned short)(1u << KBIT)};
    if constexpr (SAT) {
        // min_epu8(upper,lower); decision = (min == upper) <=> upper <= lower       spiral47.cpp:224-227
        const u16x2 z = __builtin_elementwise_sub_sat(upper, lower);  // 0 <=> upper <= lower; else a multiple of 128
        unsigned bit = 0x10001u << KBIT;
        if constexpr (KBIT == 0) asm("" : "+s"(bit));  // min(z, 1) with a constant 1 is turned into compare + select + perm
        acc |= as_u32(__builtin_elementwise_min(z, as_v(bit)));
        return __builtin_elementwise_min(lower, upper);
    } else {
        // decision = (int8)(lower-upper) > 0, survivor = decision ? upper : lower    viterbi27_sse2.cpp:155-158
        const i16x2 diff = (i16x2)(lower - upper);
        const i16x2 zero = {0, 0};
        const u16x2 pd = (u16x2)__builtin_elementwise_max(diff, zero);  // 0 or a multiple of 256 in [0x100,0x7f00]
        acc |= as_u32(__builtin_elementwise_min(pd, one));
        return lower - pd;
    }
}

template <int CTRL>
__device__ __forceinline__ u16x2 dpp_quad(u16x2 x) {
    return as_v((unsigned)__builtin_amdgcn_mov_dpp((int)as_u32(x), CTRL, 0xf, 0xf, true));
}
template <int BIT>
__device__ __forceinline__ u16x2 dpp_xor(u16x2 x) {
    static_assert(BIT == 0 || BIT == 1, "lane stages stay inside a quad");
    if constexpr (BIT == 0) return dpp_quad<0xB1>(x);  // quad_perm [1,0,3,2]
    else return dpp_quad<0x4E>(x);                     // quad_perm [2,3,0,1]
}

// branch metrics of one step for all 2^R classes, as a 16-bit field value, per lane: t << 8 (modular family), or
// ((t - pending minimum) << 7) mod 2^16 (saturating family; P = pending minimum << 7)
template <class C, int NC>
__device__ __forceinline__ void branch_fields(const unsigned (&s)[C::R], unsigned (&T)[NC], unsigned P) {
    if constexpr (C::metric == U8MOD) {
        // t = ((a0+a1+1)>>1)>>4                                                   viterbi27_sse2.cpp:137-146
        const unsigned x0 = s[0] ^ 255u, x1 = s[1] ^ 255u;
        const unsigned a0[2] = {s[0], x0}, a1[2] = {s[1], x1};
#pragma unroll
        for (int c = 0; c < 4; c++) T[c] = ((a0[c & 1] + a1[c >> 1] + 1u) >> 5) << 8;
    } else if constexpr (C::R == 2) {
        // t = (((a0+a1+1)>>1)>>2)&63                                                  spiral27.cpp:164-169
        const unsigned x0 = s[0] ^ 255u, x1 = s[1] ^ 255u;
        const unsigned a0[2] = {s[0], x0}, a1[2] = {s[1], x1};
#pragma unroll
        for (int c = 0; c < 4; c++) T[c] = (((((a0[c & 1] + a1[c >> 1] + 1u) >> 3) & 63u) << 7) - P) & 0xffffu;
    } else {
        // t = (sum_r ((a_r>>2)&63)) >> 2; (s^255)>>2 == 63-(s>>2)                     spiral47.cpp:164-219
        unsigned g[4], h[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            g[r] = s[r] >> 2;
            h[r] = g[r] ^ 63u;
        }
        // floor(sum / 4) - pending == floor((sum - 4 pending) / 4): the pending minimum goes into four partial sums
        unsigned p01[4], p23[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            p01[c] = ((c & 1) ? h[0] : g[0]) + ((c & 2) ? h[1] : g[1]);
            p23[c] = ((c & 1) ? h[2] : g[2]) + ((c & 2) ? h[3] : g[3]) - (P >> 5);
        }
#pragma unroll
        for (int c = 0; c < 16; c++) T[c] = ((p01[c & 3] + p23[c >> 2]) << 5) & 0xff80u;
    }
}

// Packed branch-metric fields for a register / lane stage: TP[c] = field(t(c)) | field(t(c ^ CH)) << 16.  For the r=1/4
// spiral code the 16 sums are formed on both fields at once (sums <= 252 per field, no carry between them), which
// halves the scalar-style arithmetic per class: (sum >> 2) << 7 per field is (sum2 << 5) & 0xff80ff80.
template <class C, int NC, unsigned CH>
__device__ __forceinline__ void branch_fields_paired(const unsigned (&s)[C::R], const unsigned (&T)[NC], unsigned (&TP)[NC], unsigned P) {
    if constexpr (C::metric == U8SAT && C::R == 4) {
        unsigned g[4], h[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            g[r] = s[r] >> 2;
            h[r] = g[r] ^ 63u;
        }
        // floor(sum / 4) - pending == floor((sum - 4 pending) / 4): the pending minimum goes into four partial sums, together
        // with a bias of 2048 that keeps every field sum positive (plain 32-bit adds, no borrow between the fields) and drops
        // out below: (sum + 2048) << 5 carries the bias into bit 16 of its own field -- bit 0 of the high field, which the
        // mask clears, or out of the dword
        const unsigned bias = 2048u - (P >> 5);
        unsigned p01[4], p23[4], x01[4], x23[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            p01[c] = ((c & 1) ? h[0] : g[0]) + ((c & 2) ? h[1] : g[1]);
            p23[c] = ((c & 1) ? h[2] : g[2]) + ((c & 2) ? h[3] : g[3]) + bias;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            x01[c] = p01[c] | (p01[c ^ (CH & 3u)] << 16);
            x23[c] = p23[c] | (p23[c ^ (CH >> 2)] << 16);
        }
#pragma unroll
        for (int c = 0; c < 16; c++) TP[c] = ((x01[c & 3] + x23[c >> 2]) << 5) & 0xff80ff80u;
    } else {
#pragma unroll
        for (int c = 0; c < NC; c++) TP[c] = T[c] | (T[c ^ CH] << 16);
    }
}

template <class C, class P, int LB>
struct RegsStep {
    using G = RegsCfg<C, P, LB>;
    static constexpr int NB = G::NB, R = G::R, NR = G::NR, NC = G::NC, L = G::L;
    static constexpr bool SAT = G::SAT;
    static constexpr unsigned COMP1 = field_from<false>(C::bm_comp) >> (SAT ? 1 : 0);  // t' = COMP - t, as a field value
    static constexpr unsigned COMP2 = COMP1 * 0x10001u;
    static constexpr int KB0 = SAT ? 0 : 1;  // first decision bit collected in a 16-bit field (acs_pk)

    // One trellis step at phase PHI (absolute row index mod NB).  M: packed metrics; sraw: this step's R symbols;
    // lam: lane index inside the frame's lane group; words: decision words of this row.  pend (saturating family): the frame
    // minimum the previous step's renormalisation owes, << 7, in both 16-bit fields; it is subtracted inside this step's branch
    // metrics -- (m - min) + t == m + (t - min), m >= min -- and replaced by what this step owes.  Whoever reads the metrics
    // out of the registers settles the last one (acs_regs_body).
    template <int PHI>
    static __device__ __forceinline__ void run(u16x2 (&M)[NR], const unsigned (&sraw)[R], unsigned lam, unsigned (&words)[G::DW], unsigned &pend) {
        constexpr int b = NB - 1 - PHI;  // position bit paired at this phase
        // per-lane class offset from the lane bits of the position (linear in GF(2)); fold it into the symbols
        unsigned s[R];
        {
            unsigned cl = 0;
            static_for<LB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i != b) {
                    constexpr unsigned ci = cls_c<P, R>(rotl_c<NB>(1u << i, PHI));
                    cl ^= ((lam >> i) & 1u) ? ci : 0u;
                }
            });
#pragma unroll
            for (int r = 0; r < R; r++) s[r] = sraw[r] ^ (((cl >> r) & 1u) ? 255u : 0u);
        }
        const unsigned PD = SAT ? (pend & 0xffffu) : 0u;
        // t' = COMP - t - pending = (COMP - 2 pending) - (t - pending), per field mod 2^16
        const unsigned CQ1 = SAT ? ((COMP1 - 2u * PD) & 0xffffu) : COMP1, CQ2 = CQ1 * 0x10001u;
        auto comp_of = [&](unsigned tp) { return SAT ? as_u32(as_v(CQ2) - as_v(tp)) : COMP2 - tp; };
        unsigned T[NC];
        branch_fields<C, NC>(s, T, PD);
        unsigned acc[NR / 8];
#pragma unroll
        for (int i = 0; i < NR / 8; i++) acc[i] = 0;

        if constexpr (b > LB) {
            // ---- register-bit stage: partner lives in another register of the same lane
            constexpr int rb = b - (LB + 1);
            constexpr unsigned ch = cls_c<P, R>(rotl_c<NB>(1u << LB, PHI));
            unsigned TP[NC], TQ[NC], TE[NC];
            branch_fields_paired<C, NC, ch>(s, T, TP, PD);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                TQ[c] = comp_of(TP[c]);
                TE[c] = as_u32(as_v(TP[c]) - as_v(TQ[c]));  // t - t' per field (mod 2^16)
            }
            static_for<NR / 2>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int r0 = ((i >> rb) << (rb + 1)) | (i & ((1 << rb) - 1));
                constexpr int r1 = r0 | (1 << rb);
                constexpr unsigned cr = cls_c<P, R>(rotl_c<NB>((unsigned)r0 << (LB + 1), PHI));
                const u16x2 A = M[r0], B = M[r1];
                const u16x2 t = as_v(TP[cr]), tc = as_v(TQ[cr]);
                if constexpr (SAT) {
                    const u16x2 m0 = madd<SAT>(A, t), m1 = madd<SAT>(B, tc), m2 = madd<SAT>(A, tc), m3 = madd<SAT>(B, t);
                    M[r0] = acs_pk<SAT, (r0 & 7) + KB0>(m0, m1, acc[r0 >> 3]);
                    M[r1] = acs_pk<SAT, (r1 & 7) + KB0>(m2, m3, acc[r1 >> 3]);
                } else {
                    // modular family: everything is exact mod 2^16, so the two differences share A-B:
                    //   m0-m1 = (A-B) + (t-t'),  m2-m3 = (A-B) - (t-t')      (9 packed ops per butterfly pair instead of 10)
                    const u16x2 e = as_v(TE[cr]);
                    const u16x2 dab = A - B;
                    const i16x2 zero = {0, 0};
                    const u16x2 p0 = (u16x2)__builtin_elementwise_max((i16x2)(dab + e), zero);
                    const u16x2 p1 = (u16x2)__builtin_elementwise_max((i16x2)(dab - e), zero);
                    const u16x2 k0 = {(unsigned short)(1u << ((r0 & 7) + 1)), (unsigned short)(1u << ((r0 & 7) + 1))};
                    const u16x2 k1 = {(unsigned short)(1u << ((r1 & 7) + 1)), (unsigned short)(1u << ((r1 & 7) + 1))};
                    acc[r0 >> 3] |= as_u32(__builtin_elementwise_min(p0, k0));
                    acc[r1 >> 3] |= as_u32(__builtin_elementwise_min(p1, k1));
                    M[r0] = (A + t) - p0;   // survivor = decision ? upper : lower = lower - positive part
                    M[r1] = (A + tc) - p1;
                }
            });
        } else if constexpr (b == LB) {
            // ---- half stage: old[j] is the low field, old[j+H] the high field of the same register
            static_for<NR>([&](auto I) {
                constexpr int r0 = decltype(I)::value;
                constexpr unsigned cr = cls_c<P, R>(rotl_c<NB>((unsigned)r0 << (LB + 1), PHI));
                const unsigned t = T[cr], tc = SAT ? ((CQ1 - t) & 0xffffu) : COMP1 - t;
                const u16x2 A = M[r0];
                // broadcasting one field to both lanes of a packed add is an op_sel modifier, not an instruction
                const u16x2 Alo = {A.x, A.x}, Ahi = {A.y, A.y};
                const u16x2 lower = madd<SAT>(Alo, as_v(t | (tc << 16)));   // (m0, m2) = old[j] + (t, t')
                const u16x2 upper = madd<SAT>(Ahi, as_v(tc | (t << 16)));   // (m1, m3) = old[j+H] + (t', t)
                M[r0] = acs_pk<SAT, (r0 & 7) + KB0>(lower, upper, acc[r0 >> 3]);
            });
        } else {
            // ---- lane stage: partner register lives in lane ^ (1<<b) of the same quad
            constexpr unsigned ch = cls_c<P, R>(rotl_c<NB>(1u << LB, PHI));
            unsigned TP[NC], TQ[NC];
            branch_fields_paired<C, NC, ch>(s, T, TP, PD);
            // The X lane of a pair holds old[j] and will hold new[2j] = ACS(old[j]+t, old[j+H]+t'); the Y lane holds
            // old[j+H] and will hold new[2j+1] = ACS(old[j]+t', old[j+H]+t).  Both lanes fetch old[j] and old[j+H] with two
            // broadcasting DPP moves, and the lane-dependent choice between t and t' is made once per class and step
            // instead of twice per register.
            const bool isY = (lam >> b) & 1u;
            unsigned TL[NC], TU[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                TQ[c] = comp_of(TP[c]);
                TL[c] = isY ? TQ[c] : TP[c];
                TU[c] = isY ? TP[c] : TQ[c];
            }
            constexpr int XSEL = b == 0 ? 0xA0 : 0x44, YSEL = b == 0 ? 0xF5 : 0xEE;  // quad_perm [0,0,2,2]/[1,1,3,3], [0,1,0,1]/[2,3,2,3]
            static_for<NR>([&](auto I) {
                constexpr int r0 = decltype(I)::value;
                constexpr unsigned cr = cls_c<P, R>(rotl_c<NB>((unsigned)r0 << (LB + 1), PHI));
                const u16x2 lo_src = dpp_quad<XSEL>(M[r0]);  // old[j]
                const u16x2 up_src = dpp_quad<YSEL>(M[r0]);  // old[j+H]
                const u16x2 lower = madd<SAT>(lo_src, as_v(TL[cr]));
                const u16x2 upper = madd<SAT>(up_src, as_v(TU[cr]));
                M[r0] = acs_pk<SAT, (r0 & 7) + KB0>(lower, upper, acc[r0 >> 3]);
            });
        }

        // ---- decision words: bit (rho % NRW) + NRW*half of word rho / NRW
        if constexpr (G::NRW == 16) {
#pragma unroll
            for (int w = 0; w < G::DW; w++) {
                // accumulators use bits KB0..KB0+7 / 16+KB0..16+KB0+7
                const unsigned v = (acc[2 * w] >> KB0) | (acc[2 * w + 1] << (8 - KB0));
                words[w] = SAT ? ~v : v;
            }
        } else {
            const unsigned v = ((acc[0] >> KB0) & 0xffu) | ((acc[0] >> (8 + KB0)) & 0xff00u);
            words[0] = SAT ? (v ^ 0xffffu) : v;
        }

        // ---- renormalisation (spiral only): if new[0] > thr the frame minimum is owed (subtracted by the next step's branch
        // metrics; the reference subtracts it here, saturating -- never clamping, it is the minimum)
        if constexpr (C::renorm) {
            unsigned v0 = as_u32(M[0]) & 0xffffu;  // state 0 always sits at position 0: register 0, low field, lane 0
            if constexpr (LB == 1) v0 = (unsigned)__builtin_amdgcn_mov_dpp((int)v0, 0xA0, 0xf, 0xf, true);  // [0,0,2,2]
            if constexpr (LB == 2) v0 = (unsigned)__builtin_amdgcn_mov_dpp((int)v0, 0x00, 0xf, 0xf, true);  // [0,0,0,0]
            const bool fire = v0 > field_from<true>(C::renorm_thr);  // spiral47.cpp:313
            // A single serial min chain makes every v_pk_min_u16 wait for its predecessor (hipcc pads it with s_nop).  A
            // pairwise tree is what hipcc schedules best around the survivors' arithmetic for every geometry except
            // K=7 r=1/2, where four interleaved chains (fewer live values) win -- measured, update ms tree / chains:
            // spiral47 1.22 / 1.27, spiral49 2.46 / 2.67, spiral29 2.20 / 2.44, spiral27 1.25 / 1.18.
            u16x2 mn;
            if constexpr (!(R == 2 && NB == 6)) {
                u16x2 tr[NR / 2];
#pragma unroll
                for (int i = 0; i < NR / 2; i++) tr[i] = __builtin_elementwise_min(M[2 * i], M[2 * i + 1]);
#pragma unroll
                for (int w = NR / 4; w >= 1; w /= 2) {
#pragma unroll
                    for (int i = 0; i < w; i++) tr[i] = __builtin_elementwise_min(tr[2 * i], tr[2 * i + 1]);
                }
                mn = tr[0];
            } else {
                u16x2 ch4[4] = {M[0], M[1], M[2], M[3]};
#pragma unroll
                for (int i = 4; i < NR; i++) ch4[i & 3] = __builtin_elementwise_min(ch4[i & 3], M[i]);
                mn = __builtin_elementwise_min(__builtin_elementwise_min(ch4[0], ch4[1]), __builtin_elementwise_min(ch4[2], ch4[3]));
            }
            const u16x2 sw = {mn.y, mn.x};
            mn = __builtin_elementwise_min(mn, sw);
            if constexpr (LB >= 1) mn = __builtin_elementwise_min(mn, dpp_xor<0>(mn));
            if constexpr (LB >= 2) mn = __builtin_elementwise_min(mn, dpp_xor<1>(mn));
            pend = fire ? (as_u32(mn) & 0xff80ff80u) : 0u;  // subs_epu8  spiral47.cpp:327-330, one step later
        }
    }
};

// symbol fetch, slow form: SW dwords covering one period, byte-assembled and guarded at both ends of this
// call's per-frame chunk (partial first/last period, or a chunk that is not 4-byte aligned)
template <int SW>
__device__ __forceinline__ void load_period_guarded(const unsigned char *sp, long off, long lim, unsigned (&d)[SW]) {
#pragma unroll
    for (int w = 0; w < SW; w++) {
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long ob = off + 4 * w + k;
            if (ob >= 0 && ob < lim) v |= (unsigned)sp[ob] << (8 * k);
        }
        d[w] = v;
    }
}
// fast form: the whole period lies inside the chunk and is dword aligned -> SW plain dword loads (merged by the
// compiler into dwordx2/x3/x4), no branches, so the loads stay in flight under the previous period's arithmetic
template <int SW>
__device__ __forceinline__ void load_period_fast(const unsigned char *sp, long off, unsigned (&d)[SW]) {
    const unsigned *p = reinterpret_cast<const unsigned *>(sp + off);
#pragma unroll
    for (int w = 0; w < SW; w++) d[w] = p[w];
}

// Where the decision words of a row go: the [group][row][word][lane] history in HBM (the exact path) ...
template <int DW, int WBYTES>
struct GlobalSink {
    unsigned char *dp;  // this lane's slot of the current row
    __device__ __forceinline__ void put(int w, unsigned word) {
        // written once, read by the chainback much later: non-temporal stores keep the history from lingering
        // dirty in L2 / the Infinity Cache (chainback right after the update: 0.219 -> 0.194 ms for K=7)
        if constexpr (WBYTES == 4) __builtin_nontemporal_store(word, reinterpret_cast<unsigned *>(dp) + w * 64);
        else __builtin_nontemporal_store((unsigned short)word, reinterpret_cast<unsigned short *>(dp) + w * 64);
    }
    __device__ __forceinline__ void next_row(int) { dp += (long)DW * 64 * WBYTES; }
};
// ... or a wave-private ring of the last RING rows in LDS, same [row][word][lane] order (the fused windowed decode)
template <int DW, int WBYTES, int RING>
struct RingSink {
    unsigned char *base;  // this lane's slot of ring row 0
    int slot;             // row & (RING - 1) of the row being written
    __device__ __forceinline__ void put(int w, unsigned word) {
        unsigned char *q = base + ((long)slot * DW + w) * 64 * WBYTES;
        if constexpr (WBYTES == 4) *reinterpret_cast<unsigned *>(q) = word;
        else *reinterpret_cast<unsigned short *>(q) = (unsigned short)word;
    }
    __device__ __forceinline__ void next_row(int) { slot = (slot + 1) & (RING - 1); }
};

// one period = NB consecutive trellis steps at phases 0..NB-1; GUARD: skip steps outside [row0,row_end)
template <class C, class P, int LB, bool GUARD, class Sink, int NR_, int SW_>
__device__ __forceinline__ void run_period(u16x2 (&M)[NR_], const unsigned (&cur)[SW_], int rbase, int row0, int row_end,
                                           unsigned lam, Sink &sink, unsigned &pend) {
    using G = RegsCfg<C, P, LB>;
    using S = RegsStep<C, P, LB>;
    constexpr int R = G::R, DW = G::DW;
    static_for<G::NB>([&](auto I) {
        constexpr int PHI = decltype(I)::value;
        const int r = rbase + PHI;
        if (!GUARD || (r >= row0 && r < row_end)) {  // wave-uniform
            unsigned sraw[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int idx = PHI * R + q;
                sraw[q] = (cur[idx >> 2] >> (8 * (idx & 3))) & 255u;
            }
            unsigned words[DW];
            S::template run<PHI>(M, sraw, lam, words, pend);
#pragma unroll
            for (int w = 0; w < DW; w++) sink.put(w, words[w]);
            sink.next_row(r);
        }
    });
}

template <class C, class P, int LB>
__device__ __forceinline__ void acs_regs_body(const AcsRegsArgs &a) {
    using G = RegsCfg<C, P, LB>;
    constexpr int NB = G::NB, R = G::R, NR = G::NR, L = G::L, FPW = G::FPW, DW = G::DW, SW = G::SW;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned lam = lane & (L - 1);
    // 4 independent waves per workgroup (one per SIMD of the CU); no LDS, no barrier
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * FPW >= a.nframes) return;
    const long f = wave * FPW + (lane >> LB);
    const bool fvalid = f < a.nframes;
    const long fc = fvalid ? f : (long)a.nframes - 1;

    const int row0 = a.row0, row_end = a.row0 + a.nsteps;
    const int phi0 = row0 % NB;
    int16_t *gm = a.metrics + fc * (long)G::N;

    // position p holds state rotl^phi(p): load the canonical metrics into the rotated in-register layout
    u16x2 M[NR];
#pragma unroll
    for (int r0 = 0; r0 < NR; r0++) {
        unsigned fld[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned p = ((unsigned)r0 << (LB + 1)) | ((unsigned)h << LB) | lam;
            const unsigned st = phi0 == 0 ? p : (((p << phi0) | (p >> (NB - phi0))) & (G::N - 1));
            fld[h] = field_from<G::SAT>(gm[st]);
        }
        M[r0] = as_v(fld[0] | (fld[1] << 16));
    }

    const unsigned char *sp = a.syms + fc * (long)a.sym_stride;
    const long lim = (long)a.nsteps * R;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.syms) | a.sym_stride) & 3) == 0 && ((phi0 * R) & 3) == 0;
    GlobalSink<DW, G::WBYTES> sink{a.dec + ((wave * a.cap_rows + row0) * (long)DW * 64 + lane) * G::WBYTES};
    unsigned pend = 0;  // saturating family: the minimum the last renormalisation owes (RegsStep::run)

    int rbase = row0 - phi0;
    while (rbase < row_end) {
        const bool full = rbase >= row0 && rbase + NB <= row_end;
        if (!(aligned && full)) {
            // slow period: partial head / tail, or unaligned symbols
            unsigned cur[SW];
            load_period_guarded<SW>(sp, (long)(rbase - row0) * R, lim, cur);
            run_period<C, P, LB, true>(M, cur, rbase, row0, row_end, lam, sink, pend);
            rbase += NB;
            continue;
        }
        // fast run over all remaining full periods; the next period's symbols are fetched while this one computes
        const int nfull = (row_end - rbase) / NB;
        long off = (long)(rbase - row0) * R;
        const long last_off = off + (long)(nfull - 1) * NB * R;
        unsigned cur[SW], nxt[SW];
        load_period_fast<SW>(sp, off, cur);
        // Land the first period's symbols BEFORE the loop: otherwise the loop header joins "load pending" (from
        // here) with "stores pending" (from the back edge) and hipcc emits s_waitcnt vmcnt(0) at the top of every
        // iteration, draining the decision stores of the previous period (measured: ~2x the kernel time).
#pragma unroll
        for (int w = 0; w < SW; w++) asm volatile("" : "+v"(cur[w]));
        for (int i = 0; i < nfull; i++) {
            const long noff = off + NB * R < last_off ? off + NB * R : last_off;  // clamped: no branch around the load
            load_period_fast<SW>(sp, noff, nxt);
            run_period<C, P, LB, false>(M, cur, rbase, row0, row_end, lam, sink, pend);
#pragma unroll
            for (int w = 0; w < SW; w++) cur[w] = nxt[w];
            off += NB * R;
            rbase += NB;
        }
    }

    if (fvalid) {
        const int phie = row_end % NB;
#pragma unroll
        for (int r0 = 0; r0 < NR; r0++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned p = ((unsigned)r0 << (LB + 1)) | ((unsigned)h << LB) | lam;
                const unsigned st = phie == 0 ? p : (((p << phie) | (p >> (NB - phie))) & (G::N - 1));
                gm[st] = (int16_t)(field_to<G::SAT>(as_u32(M[r0]) >> (16 * h)) - (int)((pend >> 7) & 0xffu));
            }
        }
    }
}

// Two entry points over the same body.  With >= 32 metric registers per lane hipcc's occupancy-driven scheduler
// serialises the packed ACS chains and pads them with s_nop (303 per period for K=7, L=1); telling it that at most
// two waves per SIMD will ever be resident lets it interleave independent butterflies instead (0 s_nop).
template <class C, class P, int LB>
__global__ __launch_bounds__(256) void acs_regs_kernel(AcsRegsArgs a) {
    acs_regs_body<C, P, LB>(a);
}
template <class C, class P, int LB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void acs_regs_kernel_ilp(AcsRegsArgs a) {
    acs_regs_body<C, P, LB>(a);
}

#ifndef VH_JIT_KERNEL  // everything from here to the end of the namespace is launch code and polynomial-independent kernels
template <class C, class P, int LB>
static hipError_t launch_regs(const AcsRegsArgs &a, hipStream_t stream) {
    using G = RegsCfg<C, P, LB>;
    const int waves = (a.nframes + G::FPW - 1) / G::FPW;
    if constexpr (G::NR >= 32)
        hipLaunchKernelGGL((acs_regs_kernel_ilp<C, P, LB>), dim3((waves + 3) / 4), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((acs_regs_kernel<C, P, LB>), dim3((waves + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Fused sliding-window decode: init + ACS update + traceback in ONE kernel.  The trellis steps
// are those of acs_regs_body, but the decision words go to a wave-private LDS ring of the last RING rows instead of
// HBM, and whenever the row `DEPTH` steps beyond a block of BLOCK payload bits has been written that block is traced
// back out of the ring -- start state 0, the walk of chainback_viterbi27_sse2 (viterbi27_sse2.cpp:97-103) over the ring
// rows -- and its bytes are stored.  No decision history and no path metrics ever reach memory.  Semantics (which differ
// from the whole-frame chainback unless DEPTH >= the frame): oracle/viterbi_oracle.c vo_chainback_windowed.
struct DecodeWindowedArgs {
    const unsigned char *syms;
    size_t sym_stride;
    int nsteps;   // trellis steps per frame = nbits + K - 1
    int nframes;
    unsigned char *data;
    size_t data_stride;
    unsigned nbits;
};

// (32 metric registers per lane in every instantiation: the same scheduling hint as acs_regs_kernel_ilp)
template <class C, class P, int LB, int DEPTH, int BLOCK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void decode_windowed_kernel(DecodeWindowedArgs a) {
    using G = RegsCfg<C, P, LB>;
    constexpr int NB = G::NB, R = G::R, NR = G::NR, L = G::L, FPW = G::FPW, DW = G::DW, SW = G::SW, WB = G::WBYTES, NRW = G::NRW;
    constexpr int RING = 64;
    static_assert(DEPTH + BLOCK + NB <= RING && BLOCK % 8 == 0 && BLOCK <= 32, "ring too small for this window");
    constexpr int ROWB = DW * 64 * WB;  // bytes of one ring row of one wave
    static_assert(4 * RING * ROWB <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[4 * RING * ROWB];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned lam = lane & (L - 1);
    unsigned char *ring = ring_all + (long)wv * RING * ROWB;
    const long wave = (long)blockIdx.x * 4 + wv;
    if (wave * FPW >= a.nframes) return;
    const long f = wave * FPW + (lane >> LB);
    const bool fvalid = f < a.nframes;
    const long fc = fvalid ? f : (long)a.nframes - 1;
    const int T = a.nsteps;

    // init_viterbi27_sse2 (viterbi27_sse2.cpp:46-52): every state init_all, state 0 (= position 0 at phase 0) init_start
    u16x2 M[NR];
#pragma unroll
    for (int r0 = 0; r0 < NR; r0++) M[r0] = as_v(field_from<G::SAT>(C::init_all) * 0x10001u);
    if (lam == 0) M[0] = as_v(field_from<G::SAT>(C::init_start) | (field_from<G::SAT>(C::init_all) << 16));

    const unsigned char *sp = a.syms + fc * (long)a.sym_stride;
    const long lim = (long)T * R;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.syms) | a.sym_stride) & 3) == 0;
    RingSink<DW, WB, RING> sink{ring + lane * WB, 0};
    unsigned pend = 0;
    unsigned char *out = a.data + fc * (long)a.data_stride;
    const unsigned nblocks = (a.nbits + BLOCK - 1) / BLOCK;
    unsigned nextb = 0;

    // one block: payload bits [lo, hi), walked from row top-1 (state 0) down to row lo+NB out of the ring
    auto emit = [&](unsigned b) {
        const unsigned lo = b * BLOCK, hi = lo + BLOCK < a.nbits ? lo + BLOCK : a.nbits;
        const int top = (int)hi + NB + DEPTH < T ? (int)hi + NB + DEPTH : T;
        unsigned st = 0, v = 0;
        int rot = top % NB;  // (r + 1) mod NB at the first row visited, r = top - 1
        const unsigned lane0 = lane & ~(unsigned)(L - 1);  // first lane of this frame's lane group
        for (int r = top - 1; r >= (int)lo + NB; r--) {
            const unsigned p = rot == 0 ? st : (((st >> rot) | (st << (NB - rot))) & (G::N - 1));
            const unsigned lm = p & (L - 1), h = (p >> LB) & 1u, rho = p >> (LB + 1);
            const unsigned w = rho / NRW, bit = (rho % NRW) + NRW * h;
            const unsigned char *q = ring + ((long)(r & (RING - 1)) * DW + w) * 64 * WB + (lane0 + lm) * WB;
            const unsigned word = WB == 4 ? *reinterpret_cast<const unsigned *>(q) : *reinterpret_cast<const unsigned short *>(q);
            const unsigned k = (word >> bit) & 1u;
            st = (st >> 1) | (k << (NB - 1));
            const int i = r - NB;
            if (i < (int)hi) v |= k << (31 - (i - (int)lo));  // MSB-first inside the block
            rot = rot == 0 ? NB - 1 : rot - 1;
        }
        if (fvalid && lam == 0) {
            const unsigned nbytes = (hi - lo + 7) / 8;
#pragma unroll
            for (unsigned m = 0; m < BLOCK / 8; m++)
                if (m < nbytes) out[lo / 8 + m] = (unsigned char)(v >> (24 - 8 * m));
        }
    };
    // K=7, one lane per frame, 8-bit blocks -- the steady state (a whole block, the full depth available).  The 64-bit
    // decision row of a frame is the lane's own two ring words, so the reads do not depend on the state: all DEPTH + BLOCK
    // rows are fetched up front (ds_read2st64_b32) and the walk is the position-space walk of chainback_k7_lds_kernel
    // below -- three dependent ALU instructions per row instead of a dependent LDS round trip -- unrolled for each of the
    // six phases the first row can have.  After the walk the top byte of h is the block's output byte; four blocks leave
    // as one aligned dword.
    constexpr bool FAST = NB == 6 && LB == 0 && BLOCK == 8;
    const bool out_aligned = ((reinterpret_cast<uintptr_t>(a.data) | a.data_stride) & 3) == 0;
    unsigned obuf = 0;
    bool steady = true;  // no block has taken the general walk yet
    auto emit_fast = [&](unsigned b, auto R0) {
        constexpr int ROT0 = decltype(R0)::value;  // rot at the first row visited
        constexpr int LW = DEPTH + BLOCK;
        constexpr int PI[6] = {4, 0, 1, 2, 3, 5};  // position bit -> bit of the row index
        const int top = (int)(b + 1) * BLOCK + NB + DEPTH;
        const unsigned *rw = reinterpret_cast<const unsigned *>(ring) + lane;
        unsigned q = 0, h = 0;  // state 0 sits at position 0 in every phase
        static_for<2>([&](auto HALF) {
            constexpr int d0 = decltype(HALF)::value * (LW / 2);
            unsigned long long W[LW / 2];
#pragma unroll
            for (int d = 0; d < LW / 2; d++) {
                const unsigned *wp = rw + ((top - 1 - (d0 + d)) & (RING - 1)) * (DW * 64);
                W[d] = (unsigned long long)wp[0] | ((unsigned long long)wp[64] << 32);
            }
            // keep the reads ahead of the walk: left alone, hipcc sinks every ds_read next to its use and the walk then
            // waits out one LDS round trip per row (s_waitcnt lgkmcnt(1) in front of every shift: 2.4 instead of 1.1 ms)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < LW / 2; d++) {
                const int rotd = ((ROT0 - (d0 + d)) % NB + NB) % NB;
                const int jb = PI[(NB - rotd) % NB];
                const unsigned t = (unsigned)(W[d] >> q);  // (word select + 32-bit shift instead: measured slower)
                q = (q & ~(1u << jb)) | ((t << jb) & (1u << jb));
                h = (h >> 1) | (t << 31);
            }
        });
        const unsigned byte = h >> 24;
        if (out_aligned) {
            obuf = (obuf << 8) | byte;
            if ((b & 3u) == 3u && fvalid) *reinterpret_cast<unsigned *>(out + (b - 3u)) = __builtin_bswap32(obuf);
        } else if (fvalid) {
            out[b] = (unsigned char)byte;
        }
    };
    auto emit_ready = [&](int rows_done) {
        while (nextb < nblocks) {
            const unsigned hi = (nextb + 1) * BLOCK < a.nbits ? (nextb + 1) * BLOCK : a.nbits;
            const int top = (int)hi + NB + DEPTH < T ? (int)hi + NB + DEPTH : T;
            if (top > rows_done) break;
            bool fast = false;
            if constexpr (FAST) fast = (nextb + 1) * BLOCK <= a.nbits && (int)hi + NB + DEPTH <= T;
            if constexpr (FAST) {
                if (fast) {
                    switch (top % NB) {  // uniform
                    case 0: emit_fast(nextb, std::integral_constant<int, 0>{}); break;
                    case 1: emit_fast(nextb, std::integral_constant<int, 1>{}); break;
                    case 2: emit_fast(nextb, std::integral_constant<int, 2>{}); break;
                    case 3: emit_fast(nextb, std::integral_constant<int, 3>{}); break;
                    case 4: emit_fast(nextb, std::integral_constant<int, 4>{}); break;
                    default: emit_fast(nextb, std::integral_constant<int, 5>{}); break;
                    }
                    nextb++;
                    continue;
                }
                // leaving the steady state (the frame's last blocks): bytes of an unfinished dword go out one by one
                if (steady && out_aligned && (nextb & 3u) != 0 && fvalid) {
                    const unsigned pend = nextb & 3u;
                    for (unsigned m = 0; m < pend; m++) out[nextb - pend + m] = (unsigned char)(obuf >> (8 * (pend - 1 - m)));
                }
                steady = false;
            }
            emit(nextb++);
        }
    };

    int rbase = 0;
    const int nfull = T / NB;
    if (aligned && nfull > 0) {
        // whole periods: the next period's symbols are fetched while this one computes (as acs_regs_body; without the
        // prefetch every period waits out a memory round trip: 1.13 instead of 0.93 ms for the trellis part of config 2)
        const long last_off = (long)(nfull - 1) * NB * R;
        long off = 0;
        unsigned cur[SW], nxt[SW];
        load_period_fast<SW>(sp, 0, cur);
#pragma unroll
        for (int w = 0; w < SW; w++) asm volatile("" : "+v"(cur[w]));
        for (int i = 0; i < nfull; i++) {
            const long noff = off + NB * R < last_off ? off + NB * R : last_off;  // clamped: no branch around the load
            load_period_fast<SW>(sp, noff, nxt);
            run_period<C, P, LB, false>(M, cur, rbase, 0, T, lam, sink, pend);
#pragma unroll
            for (int w = 0; w < SW; w++) cur[w] = nxt[w];
            off += NB * R;
            rbase += NB;
            emit_ready(rbase);
        }
    }
    while (rbase < T) {  // the last partial period, or symbols that are not dword aligned
        unsigned cur[SW];
        load_period_guarded<SW>(sp, (long)rbase * R, lim, cur);
        run_period<C, P, LB, true>(M, cur, rbase, 0, T, lam, sink, pend);
        rbase += NB;
        emit_ready(rbase < T ? rbase : T);
    }
}

template <class C, class P, int LB, int DEPTH, int BLOCK>
static hipError_t launch_windowed(const DecodeWindowedArgs &a, hipStream_t stream) {
    using G = RegsCfg<C, P, LB>;
    const int waves = (a.nframes + G::FPW - 1) / G::FPW;
    hipLaunchKernelGGL((decode_windowed_kernel<C, P, LB, DEPTH, BLOCK>), dim3((waves + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// K=7: one lane per frame (512-byte ring rows), depth 48, blocks of 8 bits; K=9: four lanes per frame (the only geometry
// whose ring fits the LDS), depth 40, blocks of 16 bits
void windowed_params(int code, int *depth, int *block, int *lb) {
    const bool k7 = code_info(code).K == 7;
    *depth = k7 ? 48 : 40;
    *block = k7 ? 8 : 16;
    *lb = k7 ? 0 : 2;
}

hipError_t launch_decode_windowed(int code, const unsigned char *syms, size_t sym_stride, int nsteps, int nframes, unsigned char *data,
                                  size_t data_stride, unsigned nbits, hipStream_t stream) {
    const DecodeWindowedArgs a{syms, sym_stride, nsteps, nframes, data, data_stride, nbits};
    switch (code) {
    case VHIP_KA9Q27: return launch_windowed<Code27, Poly27, 0, 48, 8>(a, stream);
    case VHIP_SPIRAL47: return launch_windowed<Code47, Poly47, 0, 48, 8>(a, stream);
    case VHIP_SPIRAL27: return launch_windowed<CodeS27, Poly27, 0, 48, 8>(a, stream);
    case VHIP_KA9Q29: return launch_windowed<Code29, Poly29, 2, 40, 16>(a, stream);
    case VHIP_SPIRAL49: return launch_windowed<Code49, Poly49, 2, 40, 16>(a, stream);
    case VHIP_SPIRAL29: return launch_windowed<CodeS29, Poly29, 2, 40, 16>(a, stream);
    }
    return hipErrorInvalidValue;
}

bool regs_poly_supported(int code, const int *poly) {
    auto eq = [&](const int *ref, int R) {
        for (int r = 0; r < R; r++)
            if (poly[r] != ref[r]) return false;
        return true;
    };
    switch (code) {
    case VHIP_KA9Q27: return eq(Poly27::v, 2);
    case VHIP_SPIRAL47: return eq(Poly47::v, 4);
    case VHIP_KA9Q29: return eq(Poly29::v, 2);
    case VHIP_SPIRAL49: return eq(Poly49::v, 4);
    case VHIP_SPIRAL27: return eq(Poly27::v, 2);
    case VHIP_SPIRAL29: return eq(Poly29::v, 2);
    }
    return false;
}

bool regs_lanes_supported(int code, int lb) {
    switch (code) {
    case VHIP_KA9Q27: case VHIP_SPIRAL47: case VHIP_SPIRAL27: return lb >= 0 && lb <= 2;
    case VHIP_KA9Q29: case VHIP_SPIRAL49: case VHIP_SPIRAL29: return lb >= 0 && lb <= 2;
    }
    return false;
}

RegsLayout regs_layout(int code, int lb) {
    RegsLayout g{};
    const CodeInfo ci = code_info(code);
    const int N = 1 << (ci.K - 1), L = 1 << lb;
    g.lb = lb;
    g.fpw = 64 / L;
    g.nr = N / (2 * L);
    g.nrw = g.nr < 16 ? g.nr : 16;
    g.dw = g.nr / g.nrw;
    g.wbytes = g.nrw == 16 ? 4 : 2;
    return g;
}

hipError_t launch_acs_regs(int code, int lb, const AcsRegsArgs &a, hipStream_t stream) {
    switch (code) {
    case VHIP_KA9Q27:
        if (lb == 0) return launch_regs<Code27, Poly27, 0>(a, stream);
        if (lb == 1) return launch_regs<Code27, Poly27, 1>(a, stream);
        if (lb == 2) return launch_regs<Code27, Poly27, 2>(a, stream);
        break;
    case VHIP_SPIRAL47:
        if (lb == 0) return launch_regs<Code47, Poly47, 0>(a, stream);
        if (lb == 1) return launch_regs<Code47, Poly47, 1>(a, stream);
        if (lb == 2) return launch_regs<Code47, Poly47, 2>(a, stream);
        break;
    case VHIP_KA9Q29:
        if (lb == 0) return launch_regs<Code29, Poly29, 0>(a, stream);
        if (lb == 1) return launch_regs<Code29, Poly29, 1>(a, stream);
        if (lb == 2) return launch_regs<Code29, Poly29, 2>(a, stream);
        break;
    case VHIP_SPIRAL49:
        if (lb == 0) return launch_regs<Code49, Poly49, 0>(a, stream);
        if (lb == 1) return launch_regs<Code49, Poly49, 1>(a, stream);
        if (lb == 2) return launch_regs<Code49, Poly49, 2>(a, stream);
        break;
    case VHIP_SPIRAL27:
        if (lb == 0) return launch_regs<CodeS27, Poly27, 0>(a, stream);
        if (lb == 1) return launch_regs<CodeS27, Poly27, 1>(a, stream);
        if (lb == 2) return launch_regs<CodeS27, Poly27, 2>(a, stream);
        break;
    case VHIP_SPIRAL29:
        if (lb == 0) return launch_regs<CodeS29, Poly29, 0>(a, stream);
        if (lb == 1) return launch_regs<CodeS29, Poly29, 1>(a, stream);
        if (lb == 2) return launch_regs<CodeS29, Poly29, 2>(a, stream);
        break;
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------------
// chainback over the [group][row][word][lane] decision layout: one thread per frame, one wave per 64 frames.
// Same walk as chainback_viterbi27_sse2 (viterbi27_sse2.cpp:78-105) / chainback_spiral47 (spiral47.cpp:84-121);
// the decision of new state n at row r sits at position rotr^((r+1) mod NB)(n).  In this layout the bytes holding a
// frame's decisions for row r do not depend on the state (all N/8 bytes of the row belong to the frame's own L lanes),
// so the history is streamed ahead of the serial state recurrence.
//
// LDS-staged form: one wave per 64 frames streams the decision history of its groups through a double-buffered LDS
// ring with direct global->LDS loads (global_load_lds_dwordx4, 1 KiB per instruction, no VGPR staging), D rows per
// buffer; the loads of block n+1 are in flight while block n is walked out of LDS.  Same walk, same bytes.
template <int K, int LB, int D>
__global__ __launch_bounds__(64) void chainback_regs_lds_kernel(ChainbackRegsArgs a) {
    constexpr int NB = K - 1, L = 1 << LB;
    constexpr unsigned N = 1u << NB;
    constexpr int NR = N / (2 * L), NRW = NR < 16 ? NR : 16, DW = NR / NRW, WB = NRW == 16 ? 4 : 2;
    constexpr int FPW = 64 / L, GPW = L;   // frames per group, groups per wave
    constexpr int RS = DW * 64 * WB;       // bytes of one row of one group
    constexpr int GB = D * RS;             // bytes of one group's block
    constexpr int BUF = GPW * GB;          // bytes per buffer
    static_assert(GB % 1024 == 0 && 2 * BUF + 8192 <= 65536, "block geometry");
    constexpr int add = (NB < 8) ? 8 - NB : 0, sub = (NB > 8) ? NB - 8 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char ring[2 * BUF];
    __shared__ unsigned stash[32 * 64];  // [dword of the 128-byte output window][lane]

    const unsigned lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * 64 + lane;
    const bool active = f0 < a.nframes;
    const long f = active ? f0 : (long)a.nframes - 1;
    const unsigned gl = lane / FPW, fl = (unsigned)(f0 % FPW);
    unsigned char *out = a.data + f * (long)a.data_stride;
    const unsigned char *wave_dec = a.dec + ((long)blockIdx.x * GPW * a.cap_rows) * RS;  // group gw0 = blockIdx.x*GPW
    const unsigned char *gbase = a.dec + ((f / FPW) * a.cap_rows) * (long)RS + (fl * L) * WB;  // for the remainder walk
    unsigned e = (a.endstate % N) << add;
    int rot = (int)(a.nbits % NB);

    // Decoded bytes: neighbouring lanes' bytes are a frame apart, so a byte store per 8 rows is 64 separate lines per
    // instruction and every one a read-modify-write (see chainback_k7_lds_kernel below for the measurements).  h holds
    // the last 32 decisions (its top byte is the reference's output register); every 32 rows it is one dword of the
    // output, parked in LDS, and a 1024-row window leaves as one whole 128-byte line per lane.  Rows at or above
    // dword_top belong to a dword whose upper bytes lie beyond nbits: they keep the reference's byte stores.
    static_assert(sub == 0 && K - 2 + add == 7, "e is the 8-bit output register itself");
    const bool out_aligned = ((reinterpret_cast<uintptr_t>(a.data) | a.data_stride) & 3) == 0;
    const unsigned last_byte_row = a.nbits ? ((a.nbits - 1u) & ~7u) : 0u;
    const unsigned dword_top = (!out_aligned || last_byte_row < 24u) ? 0u : ((last_byte_row - 24u) / 32u + 1u) * 32u;
    unsigned h = e << 24;
    auto advance = [&](unsigned i, unsigned k) {
        e = (e >> 1) | (k << (K - 2 + add));
        h = (h >> 1) | (k << 31);
        if (i >= dword_top) {
            if ((i & 7u) == 0 && active) out[i >> 3] = (unsigned char)(e >> sub);
        } else if ((i & 31u) == 0) {
            stash[((i >> 5) & 31u) * 64 + lane] = __builtin_bswap32(h);
            if ((i & 1023u) == 0 && active) {  // window [i, i+1024) below dword_top is complete
                const unsigned top = i + 1024u < dword_top ? i + 1024u : dword_top, cnt = (top - i) / 32u;
                const unsigned *sp = stash + lane;
                unsigned *o = reinterpret_cast<unsigned *>(out + (i >> 3));
                if (cnt == 32u) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        reinterpret_cast<uint4 *>(o)[q] = make_uint4(sp[(4 * q) * 64], sp[(4 * q + 1) * 64], sp[(4 * q + 2) * 64], sp[(4 * q + 3) * 64]);
                } else {
                    for (unsigned q = 0; q < cnt; q++) o[q] = sp[q * 64];
                }
            }
        }
        rot = rot == 0 ? NB - 1 : rot - 1;
    };
    auto locate = [&](unsigned &woff, unsigned &bit) {  // byte offset inside a group row, bit inside the word
        const unsigned st = e >> add;
        const unsigned p = rot == 0 ? st : (((st >> rot) | (st << (NB - rot))) & (N - 1));
        const unsigned lam = p & (L - 1), h = (p >> LB) & 1u, rho = p >> (LB + 1);
        woff = ((rho / NRW) * 64 + fl * L + lam) * WB;
        bit = (rho % NRW) + NRW * h;
    };
    // rows [r_lo, r_lo + D) of every group of this wave -> buffer b
    auto issue = [&](int b, long r_lo) {
#pragma unroll
        for (int g = 0; g < GPW; g++) {
            const unsigned char *src = wave_dec + ((long)g * a.cap_rows + r_lo) * RS + lane * 16;
#pragma unroll
            for (int c = 0; c < GB / 1024; c++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 1024),
                                                 (__attribute__((address_space(3))) void *)(ring + b * BUF + g * GB + c * 1024), 16, 0, 2);
        }
    };

    unsigned i = a.nbits;
    while (i > 0 && (long)(i - 1) + NB >= a.rows_written) {  // rows never written read as zero
        --i;
        advance(i, 0);
    }
    const unsigned nblk = i / D;
    if (nblk > 0) {
        issue(0, (long)(i - D) + NB);
        for (unsigned b = 0; b < nblk; b++) {
            __syncthreads();  // block b has landed (hipcc drains vmcnt here); every lane is done with the other buffer
            if (b + 1 < nblk) issue((int)((b + 1) & 1u), (long)(i - 2 * D) + NB);
            const unsigned char *blk = ring + (b & 1u) * BUF + gl * GB;
            // Rows [i-D, i).  Almost every block lies wholly below dword_top and inside one 1024-row output window: its
            // rows then need no decision about the output at all -- h is parked on every row (the row that completes
            // a dword writes last, rows descend), which keeps scalar branches off the per-row path.
            const bool lean = i <= dword_top && ((i - 1u) >> 10) == ((i - D) >> 10) && ((i - D) & 1023u) != 0;
            if (lean) {
#pragma unroll 4
                for (int d = 0; d < D; d++) {
                    unsigned woff, bit;
                    locate(woff, bit);
                    const unsigned char *wp = blk + (D - 1 - d) * RS + woff;
                    const unsigned word = WB == 4 ? *reinterpret_cast<const unsigned *>(wp) : *reinterpret_cast<const unsigned short *>(wp);
                    const unsigned k = (word >> bit) & 1u;
                    e = (e >> 1) | (k << (K - 2 + add));
                    h = (h >> 1) | (k << 31);
                    stash[(((i - 1 - d) >> 5) & 31u) * 64 + lane] = __builtin_bswap32(h);
                    rot = rot == 0 ? NB - 1 : rot - 1;
                }
            } else {
#pragma unroll 2
                for (int d = 0; d < D; d++) {
                    unsigned woff, bit;
                    locate(woff, bit);
                    const unsigned char *wp = blk + (D - 1 - d) * RS + woff;
                    const unsigned word = WB == 4 ? *reinterpret_cast<const unsigned *>(wp) : *reinterpret_cast<const unsigned short *>(wp);
                    advance(i - 1 - d, (word >> bit) & 1u);
                }
            }
            i -= D;
        }
    }
    while (i > 0) {  // fewer than D rows left: read them from global memory
        --i;
        unsigned woff, bit;
        locate(woff, bit);
        const unsigned char *wp = gbase + ((long)i + NB) * RS + woff - (fl * L) * WB;
        const unsigned word = WB == 4 ? *reinterpret_cast<const unsigned *>(wp) : *reinterpret_cast<const unsigned short *>(wp);
        advance(i, (word >> bit) & 1u);
    }
}

// K=7, one lane per frame (LB = 0): the whole 64-bit decision row of a frame is the lane's own two words, so the LDS
// reads do not depend on the state at all and are issued ahead of the walk; only ALU work is left on the serial chain.
// The walk runs in POSITION space: position p = rotr^rot(state) holds the state (rot = (row+1) mod 6), and one
// traceback step  state' = (state >> 1) | (k << 5)  is  p' = p with bit (6 - rot) mod 6 replaced by k  -- the same
// rotating map the ACS kernel uses, so nothing is rotated per row.  q = the bit index of p inside the row (a fixed
// permutation of p's bits) is what is kept: k = (row >> q) & 1, then one bit of q is replaced: three dependent
// instructions per decoded bit instead of a dependent LDS round trip.  A block is 32 rows = one output dword; the phase
// of the six-row period at its first row is uniform, so each of the six phases gets its own unrolled walk with
// compile-time bit indices.
// Decoded bytes: a lane's bytes are 256 B apart from its neighbours', so a byte store per 8 rows is 64 separate lines per
// instruction, and any store narrower than a 32-byte sector is a read-modify-write once the streaming history has pushed
// the line out of L2 -- measured for 65536 frames: byte stores 0.27 ms, 12-byte stores 0.23 ms, 32-byte 0.21 ms, 64-byte
// 0.195 ms, 128-byte 0.188 ms, no stores 0.18 ms for the same walk (tools/hbm_read_probe.hip shows the same on a bare
// stream).  The dwords of a 1024-row window are parked in LDS and leave as one whole 128-byte line per lane.  Same bytes out as chainback_viterbi27_sse2 (viterbi27_sse2.cpp:78-105): the top byte of
// the 32-bit history is the reference's 8-bit register e.
template <int D, int WD>
__global__ __launch_bounds__(64) void chainback_k7_lds_kernel(ChainbackRegsArgs a) {
    constexpr int K = 7, NB = 6, RS = 512, GB = D * RS, add = 2;
    static_assert(WD == 8 || WD == 16 || WD == 32, "output window in dwords");
    constexpr unsigned N = 64;
    static_assert(D == 32 && GB % 1024 == 0, "one dword of output per block, whole DMA chunks");
    constexpr int NBUF = 2;  // a three-deep ring (two blocks in flight behind s_waitcnt vmcnt(n)) measured the same
    __shared__ __attribute__((aligned(16))) unsigned char ring[NBUF * GB];
    __shared__ unsigned stash[WD * 64];  // [dword of the output window][lane]

    const unsigned lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * 64 + lane;
    const bool active = f0 < a.nframes;
    unsigned char *out = a.data + (active ? f0 : (long)a.nframes - 1) * (long)a.data_stride;
    const unsigned char *wave_dec = a.dec + ((long)blockIdx.x * a.cap_rows) * RS;
    unsigned e = (a.endstate % N) << add;
    int rot = (int)(a.nbits % NB);

    auto advance = [&](unsigned i, unsigned k) {
        e = (e >> 1) | (k << (K - 2 + add));
        if ((i & 7u) == 0 && active) out[i >> 3] = (unsigned char)e;
        rot = rot == 0 ? NB - 1 : rot - 1;
    };
    // bit index of position p inside the 64-bit row: word p>>5, bit ((p>>1) & 15) + 16 * (p & 1)
    auto row_bit = [](unsigned p) { return (p & 32u) | ((p & 1u) << 4) | ((p >> 1) & 15u); };
    auto position = [&]() {
        const unsigned st = e >> add;
        return rot == 0 ? st : (((st >> rot) | (st << (NB - rot))) & (N - 1));
    };
    auto issue = [&](int b, long r_lo) {
        const unsigned char *src = wave_dec + r_lo * RS + lane * 16;
#pragma unroll
        for (int c = 0; c < GB / 1024; c++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 1024),
                                             (__attribute__((address_space(3))) void *)(ring + b * GB + c * 1024), 16, 0, 2);
    };
    // cnt <= 64 rows below i, through LDS with a dependent read per row and the reference's byte stores
    auto slow_rows = [&](unsigned &i, unsigned cnt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned lo = i - cnt;
        const unsigned char *src = wave_dec + ((long)lo + NB) * RS + lane * 16;
        for (unsigned c = 0; c < cnt / 2; c++)  // two rows per 1 KiB instruction
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 1024),
                                             (__attribute__((address_space(3))) void *)(ring + c * 1024), 16, 0, 0);
        if ((cnt & 1u) && lane < 32)  // an odd last row: half an instruction, nothing is read beyond row NB+i-1
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (cnt / 2) * 1024),
                                             (__attribute__((address_space(3))) void *)(ring + (cnt / 2) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned *blk = reinterpret_cast<const unsigned *>(ring) + lane;
        while (i > lo) {
            --i;
            const unsigned qq = row_bit(position());
            const unsigned word = blk[(i - lo) * (RS / 4) + (qq >> 5) * 64];
            advance(i, (word >> (qq & 31u)) & 1u);
        }
    };

    unsigned i = a.nbits;
    while (i > 0 && (long)(i - 1) + NB >= a.rows_written) {  // rows never written read as zero
        --i;
        advance(i, 0);
    }
    const bool out_aligned = ((reinterpret_cast<uintptr_t>(a.data) | a.data_stride) & 3) == 0;
    if (out_aligned && i >= (unsigned)D + (i & 31u)) {
        if (i & 31u) slow_rows(i, i & 31u);  // blocks start on a dword of the output
        const unsigned nblk = i / D;
        unsigned q = row_bit(position());
        unsigned h = e << 24;  // the last 32 decisions, newest at bit 31
        // one block: rows i-1 ... i-D out of buffer blk; ROT0 = rot at the block's first row
        auto walk = [&](auto R0, const unsigned *blk) {
            constexpr int ROT0 = decltype(R0)::value;
            constexpr int PI[6] = {4, 0, 1, 2, 3, 5};  // position bit -> row-index bit
            unsigned long long W[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                const unsigned *wp = blk + (D - 1 - d) * (RS / 4);
                W[d] = (unsigned long long)wp[0] | ((unsigned long long)wp[64] << 32);
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int rotd = ((ROT0 - d) % NB + NB) % NB;    // rot at row i-1-d
                const int jb = PI[(NB - rotd) % NB];             // the position bit replaced there
                const unsigned t = (unsigned)(W[d] >> q);        // bit 0 = the decision
                q = (q & ~(1u << jb)) | ((t << jb) & (1u << jb));
                h = (h >> 1) | (t << 31);
            }
        };
        // block n covers rows i0-1-n*D ... i0-(n+1)*D and lives in buffer n % NBUF
        const unsigned i0 = i;
        issue(0, (long)(i0 - D) + NB);
        unsigned parked = 0;  // dwords of the current 256-row window waiting in the stash
        for (unsigned b = 0; b < nblk; b++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // block b has landed (single wave: no barrier needed)
            if (b + 1 < nblk) issue((int)((b + 1) & 1u), (long)(i0 - (b + 2) * D) + NB);
            const unsigned *blk = reinterpret_cast<const unsigned *>(ring + (b & 1u) * GB) + lane;
            switch (rot) {  // uniform
            case 0: walk(std::integral_constant<int, 0>{}, blk); break;
            case 1: walk(std::integral_constant<int, 1>{}, blk); break;
            case 2: walk(std::integral_constant<int, 2>{}, blk); break;
            case 3: walk(std::integral_constant<int, 3>{}, blk); break;
            case 4: walk(std::integral_constant<int, 4>{}, blk); break;
            default: walk(std::integral_constant<int, 5>{}, blk); break;
            }
            rot = (rot + NB * 6 - D) % NB;
            i -= D;
            // h = rows i ... i+31, row i on top = the four bytes at out[i/8 ...], first byte in the top bits
            stash[((i >> 5) & (WD - 1)) * 64 + lane] = __builtin_bswap32(h);
            parked++;
            if ((i & (32u * WD - 1)) == 0 || b + 1 == nblk) {
                if (active) {
                    const unsigned *sp = stash + ((i >> 5) & (WD - 1)) * 64 + lane;
                    unsigned *o = reinterpret_cast<unsigned *>(out + (i >> 3));
                    if (parked == (unsigned)WD) {  // the whole window: WD * 4 contiguous, aligned bytes per lane
#pragma unroll
                        for (int k = 0; k < WD / 4; k++)
                            reinterpret_cast<uint4 *>(o)[k] = make_uint4(sp[(4 * k) * 64], sp[(4 * k + 1) * 64], sp[(4 * k + 2) * 64], sp[(4 * k + 3) * 64]);
                    } else {
                        for (unsigned k = 0; k < parked; k++) o[k] = sp[k * 64];
                    }
                }
                parked = 0;
            }
        }
        e = h >> 24;
    }
    while (i > 0) slow_rows(i, i < 64u ? i : 64u);  // what is left
}

hipError_t launch_chainback_regs(const ChainbackRegsArgs &a, hipStream_t stream) {
    const int blocks = (a.nframes + 63) / 64;
    if (a.K == 7 && a.lay.lb == 0) {
        hipLaunchKernelGGL((chainback_k7_lds_kernel<32, 32>), dim3(blocks), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
#define VH_CBL(KK, LBB, DD)                                                                                      \
    if (a.K == KK && a.lay.lb == LBB) {                                                                          \
        hipLaunchKernelGGL((chainback_regs_lds_kernel<KK, LBB, DD>), dim3(blocks), dim3(64), 0, stream, a);      \
        return hipGetLastError();                                                                                \
    }
    VH_CBL(7, 1, 32) VH_CBL(7, 2, 32) VH_CBL(9, 0, 12) VH_CBL(9, 1, 12) VH_CBL(9, 2, 12)
#undef VH_CBL
    return hipErrorInvalidValue;  // no such decision layout
}

#endif  // !VH_JIT_KERNEL

}  // namespace vh

#ifdef VH_JIT_KERNEL
// Runtime specialisation (jit.hip): this file is compiled again by hipcc --genco with the caller's polynomials as the
// compile-time constants the unrolled phases need -- VH_JIT_CODE (a traits type of viterbi_codes.h), VH_JIT_POLY ({a,b,..}),
// VH_JIT_LB -- and yields exactly one kernel.
namespace vh {
struct PolyJit { static constexpr int v[8] = VH_JIT_POLY; };
}
extern "C" __global__ __launch_bounds__(256) VH_JIT_ATTR void vh_jit_acs_regs(vh::AcsRegsArgs a) {
    vh::acs_regs_body<vh::VH_JIT_CODE, vh::PolyJit, VH_JIT_LB>(a);
}
#endif
