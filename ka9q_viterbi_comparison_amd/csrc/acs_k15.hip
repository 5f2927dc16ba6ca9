// acs_k15.hip -- production ACS update kernel for K=15 r=1/6 (ka9q i16 saturating family).
//
// Replaces update_viterbi615_blk_sse2 (ka9q_libfec_port/viterbi615_sse2.cpp:104-191) for the harness
// polynomials (src/main.cpp:405).
//
// One 128-thread workgroup (2 waves) per frame; several workgroups per CU.  The 16384 i16 metrics of a frame use
// the same in-place rotating trellis as acs_regs.hip (position p holds state rotl^phi(p) before the step of phase
// phi = row mod 14; the butterfly partner differs in position bit 13-phi), but the positions are split
//      7 "free" bits held in registers (128 positions = 64 packed VGPRs per thread)  x  7 bits = thread id.
// Phases 0..6 pair position bits 13..7, phases 7..13 pair bits 6..0, so a thread runs SEVEN trellis steps on
// registers alone, then the workgroup transposes through a 32 KiB LDS image (free bits <-> thread bits) and runs
// the other seven.  LDS traffic is 2 x 32 KiB per 7 steps instead of per step, and no metric touches HBM.
//
// Arithmetic (viterbi615_sse2.cpp:132-148): t = sum of six (sym ^ table) terms, t' = 1530 - t, adds_epi16,
// min_epi16, decision = (min == upper) i.e. tie -> upper; packed as v_pk_add_i16 clamp / v_pk_min_i16, with the
// decision taken from the sign of the saturating difference lower - upper (sign set <=> decision 0).
// Branch metrics: the 64 table classes factor as TA[c & 7] + TB[c >> 3] (two 8-entry tables per step); the class
// contribution of the thread-id bits is folded into the symbols (XOR with 255), so table indices are static.
// Renormalisation (viterbi615_sse2.cpp:160-183) is exact and immediate: thread 0 owns state 0 (position 0 in every
// phase), publishes new[0] >= 20017 through LDS behind the per-step barrier, and the workgroup min-reduces and
// subtracts with 16-bit wrap-around.
//
// Decision layout: row r = 512 words [w][thread], bit (rho & 15) + 16*half of word rho >> 4, where (thread, rho,
// half) follow from position rotr^((r+1) mod 14)(n) of new state n and the phase group of r (see chainback below).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "k15_layout.h"
#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {
namespace k15 {

typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int K = 15, R = 6, NB = 14, N = 1 << NB;
static_assert(NB == K - 1, "position bits");
constexpr int THREADS = 128, NR = 64;
#ifdef VH_JIT_KERNEL
constexpr int POLY[6] = VH_JIT_POLY;  // runtime specialisation (jit.hip): the caller's polynomials
#else
constexpr int POLY[6] = {042631, 047245, 056507, 073363, 077267, 064537};  // src/main.cpp:405
#endif

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int NN, class F>
__device__ __forceinline__ void sfor(F &&f) {
    sfor_impl(f, std::make_integer_sequence<int, NN>{});
}

constexpr unsigned popc(unsigned x) {
    unsigned n = 0;
    while (x) {
        n += x & 1u;
        x >>= 1;
    }
    return n;
}
constexpr unsigned rotl14(unsigned x, int s) {
    s %= NB;
    return s == 0 ? x : (((x << s) | (x >> (NB - s))) & (N - 1u));
}
// branch-table class of state j: bit r = parity((2j) & poly[r])                  viterbi615_sse2.cpp:52-56
constexpr unsigned cls(unsigned j) {
    unsigned c = 0;
    for (int r = 0; r < R; r++) c |= (popc((2u * j) & (unsigned)POLY[r]) & 1u) << r;
    return c;
}

__device__ __forceinline__ unsigned as_u32(i16x2 x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ i16x2 as_v(unsigned x) { return __builtin_bit_cast(i16x2, x); }

// packed ACS for two new states (ka9q615): w < 0  <=>  lower < upper  <=>  decision 0 (tie -> upper, viterbi615_sse2.cpp:145-148)
__device__ __forceinline__ i16x2 acs(i16x2 lower, i16x2 upper, i16x2 &w) {
    w = __builtin_elementwise_sub_sat(lower, upper);
    return __builtin_elementwise_min(lower, upper);  // min_epi16
}
// Decision bits of the register pair (RE, RE+1), RE even: one v_perm_b32 gathers the four sign bytes
// [RE.low, RE.high, RE+1.low, RE+1.high], one shift + one v_and_or_b32 drops the four sign bits into the word of the
// 16-register group (1.5 instructions per register instead of shift + and + or per register).  acc collects the
// COMPLEMENT of the decisions at bit k15_decision_bit(true, rho, half) (k15_layout.h).
template <int RE, int NA>
__device__ __forceinline__ void put_signs(i16x2 we, i16x2 wo, unsigned (&acc)[NA], const SignMasks &sm) {
    static_assert((RE & 1) == 0, "pairs start at an even register");
    constexpr int i8 = (RE & 15) >> 1;
    // v_perm_b32 selectors 8..11 replicate the SIGN of bytes 1 / 3 of either source over the whole destination byte:
    // one instruction turns the four 16-bit sign bits of (we, wo) into four 0x00 / 0xff bytes, and the pair's bit of each
    // byte is kept by a mask -- no shift (perm + and + or per register pair)
    const unsigned P = __builtin_amdgcn_perm(as_u32(wo), as_u32(we), 0x0b0a0908u);
    acc[RE >> 4] = (P & sm.m[i8]) | acc[RE >> 4];
}

// spiral615 flavour (spiral/spiral615.cpp:220-227): u8 saturating metrics held as (m<<7)|0x7f in SIGNED 16-bit fields so that
// v_pk_add_i16 clamp == adds_epu8 at the top (255 <-> 0x7fff) while the branch metric may be negative: the minimum that
// renormalize() subtracts after step n (spiral615.cpp:31-40) is folded into the branch metrics of step n+1 instead
// ("pending", see sp_field / the step loop) -- (m - min) + t == m + (t - min), and m >= min keeps the bottom clamp out of it.
// decision = (min == upper); acc collects the COMPLEMENT at bit KB / 16+KB (KB = 0..7: the min(z, 1<<KB) trick of
// acs_regs.hip needs z >= 1<<KB whenever it is non-zero, and differences are multiples of 128).
constexpr unsigned sp_field(unsigned m) { return (m << 7) | 0x7fu; }
template <int KB>
__device__ __forceinline__ i16x2 acs_u8(i16x2 lower, i16x2 upper, unsigned &acc) {
    const u16x2 lo = (u16x2)lower, up = (u16x2)upper;
    const u16x2 z = __builtin_elementwise_sub_sat(up, lo);  // 0 <=> upper <= lower <=> decision 1
    unsigned bit = 0x10001u << KB;
    if constexpr (KB == 0) asm("" : "+s"(bit));  // min(z, 1) as a constant is turned into compare + select + perm
    acc |= __builtin_bit_cast(unsigned, __builtin_elementwise_min(z, __builtin_bit_cast(u16x2, bit)));
    return (i16x2)__builtin_elementwise_min(lo, up);
}
template <bool SP>
__device__ __forceinline__ i16x2 madd(i16x2 a, unsigned t) {
    return __builtin_elementwise_add_sat(a, as_v(t));  // adds_epi16; spiral: adds_epu8 in the (m<<7)|0x7f fields
}

// Branch-table class offset of this thread's thread-id bits at phase PHI (linear in GF(2); folded into the symbols as a
// conditional complement).  It depends on the thread and the phase only.  Left to itself hipcc hoists the 14 x 6 complement
// masks out of the step loop, runs out of registers (the kernel sits at 256 VGPRs) and spills them: every trellis step then
// began with three scratch reloads and an s_waitcnt vmcnt(0) that also drained the previous step's decision stores.  So the 14
// six-bit offsets are formed once per kernel, kept packed in three registers, and the masks are rebuilt on every step behind
// an opaque asm that keeps them from being hoisted (12 VALU instructions per step).
template <int PHI>
__device__ __forceinline__ unsigned class_offset(unsigned tid) {
    constexpr int b = NB - 1 - PHI;
    constexpr int TSH = b >= 7 ? 0 : 7;  // thread bits <-> position bits: p_t = tid << TSH
    unsigned cl = 0;
    sfor<7>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr unsigned ci = cls(rotl14(1u << (i + TSH), PHI));
        cl ^= ((tid >> i) & 1u) ? ci : 0u;
    });
    return cl;
}
struct ClassOffsets {
    unsigned w[3];
    __device__ __forceinline__ explicit ClassOffsets(unsigned tid) {
        w[0] = w[1] = w[2] = 0;
        sfor<NB>([&](auto P) {
            constexpr int PHI = decltype(P)::value;
            w[PHI / 5] |= class_offset<PHI>(tid) << (6 * (PHI % 5));
        });
    }
    template <int PHI>
    __device__ __forceinline__ unsigned get() const {
        unsigned cl = (w[PHI / 5] >> (6 * (PHI % 5))) & 63u;
        asm volatile("" : "+v"(cl));  // not loop-invariant as far as the compiler can tell
        return cl;
    }
};

// One trellis step at phase PHI on the 128 positions this thread holds.  SP selects the spiral615 arithmetic; `pend` is the
// minimum the previous step's renormalisation still owes (spiral615 only, workgroup-uniform, 0..255).
template <bool SP, int PHI>
__device__ __forceinline__ void stage(i16x2 (&M)[NR], const unsigned (&sraw)[R], const ClassOffsets &co, unsigned (&words)[4], const SignMasks &sm,
                                      unsigned pend = 0) {
    constexpr int b = NB - 1 - PHI;            // position bit paired in this phase
    constexpr bool GA = b >= 7;                // group A: free bits 7..13, thread bits 0..6; group B: the reverse
    constexpr int kf = GA ? b - 7 : b;         // local free-bit index of the paired bit (0 = the half bit)
    constexpr int FSH = GA ? 7 : 0;            // local position q <-> position bits: p_q = q << FSH

    // class offset of the thread-id bits, folded into the symbols (conditional complement)
    const unsigned cl = co.template get<PHI>();
    unsigned s[R];
#pragma unroll
    for (int r = 0; r < R; r++) s[r] = sraw[r] ^ (((cl >> r) & 1u) ? 255u : 0u);

    // t(c) = TA[c & 7] + TB[c >> 3]                                                  viterbi615_sse2.cpp:132-135
    // spiral615: the terms are ((sym ^ table) >> 2) & 63 (spiral615.cpp:149-215); (s^255)>>2 == 63 - (s>>2)
    if constexpr (SP) {
#pragma unroll
        for (int r = 0; r < R; r++) s[r] >>= 2;
    }
    constexpr unsigned FLIP = SP ? 63u : 255u;
    unsigned TA[8], TB[8];
    {
        const unsigned x0 = s[0] ^ FLIP, x1 = s[1] ^ FLIP, x2 = s[2] ^ FLIP;
        const unsigned y0 = s[3] ^ FLIP, y1 = s[4] ^ FLIP, y2 = s[5] ^ FLIP;
        const unsigned a01[4] = {s[0] + s[1], x0 + s[1], s[0] + x1, x0 + x1};
        const unsigned b01[4] = {s[3] + s[4], y0 + s[4], s[3] + y1, y0 + y1};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            TA[c] = a01[c & 3] + ((c & 4) ? x2 : s[2]);
            TB[c] = b01[c & 3] + ((c & 4) ? y2 : s[5]);
        }
    }
    unsigned acc[SP ? 8 : 4] = {};
    constexpr unsigned COMP = SP ? 94u : (unsigned)Code615::bm_comp;
    // spiral615: (t - pend) << 7 and (94 - t - pend) << 7 as wrapping 16-bit fields = t * (+-128) + these two (v_pk_mad_u16)
    const unsigned short spn = (unsigned short)((0u - pend) << 7), spq = (unsigned short)((COMP - pend) << 7);
    unsigned mul_up = 0x00800080u, mul_dn = 0xff80ff80u;  // +-128 per field, opaque: as constants the multiply becomes a shift and
    asm("" : "+s"(mul_up), "+s"(mul_dn));                 // the v_pk_mad_u16 two instructions
    // branch-metric fields of a register pair from the table sums (both 16-bit fields at once)
    auto fields = [&](unsigned sum, unsigned &tp, unsigned &tq) {
        if constexpr (SP) {
            // adds_epu8 chain saturates at 255, then (>>2)&63: t = min(63, sum>>2); t' = subs_epu8(94, t)   spiral615.cpp:216-242
            // (94 - t >= 31 > 0: that saturating subtract never clamps); both minus the pending minimum
            const u16x2 c63 = {63, 63}, up = __builtin_bit_cast(u16x2, mul_up), dn = __builtin_bit_cast(u16x2, mul_dn), vn = {spn, spn}, vq = {spq, spq};
            const u16x2 t = __builtin_elementwise_min((u16x2)(__builtin_bit_cast(u16x2, sum) >> 2), c63);
            tp = __builtin_bit_cast(unsigned, (u16x2)(t * up + vn));
            tq = __builtin_bit_cast(unsigned, (u16x2)(t * dn + vq));
        } else {
            tp = sum;                          // both fields <= 1530: no carry between them
            tq = COMP * 0x10001u - tp;         // t' = 1530 - t in both fields
        }
    };

    if constexpr (kf > 0) {
        constexpr int rb = kf - 1;
        constexpr unsigned ch = cls(rotl14(1u << FSH, PHI));  // class of the half bit (local bit 0)
        unsigned TAp[8], TBp[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            TAp[c] = TA[c] | (TA[c ^ (ch & 7u)] << 16);
            TBp[c] = TB[c] | (TB[c ^ (ch >> 3)] << 16);
        }
        // one butterfly pair of registers (r0, r1 = r0 | 1<<rb); W0/W1 receive the decision differences (ka9q615)
        auto pair = [&](auto R0, i16x2 &W0, i16x2 &W1) {
            constexpr int r0 = decltype(R0)::value;
            constexpr int r1 = r0 | (1 << rb);
            constexpr unsigned cr = cls(rotl14(((unsigned)r0 << 1) << FSH, PHI));
            unsigned tp, tq;
            fields(TAp[cr & 7u] + TBp[cr >> 3], tp, tq);
            const i16x2 A = M[r0], B = M[r1];
            const i16x2 m0 = madd<SP>(A, tp), m1 = madd<SP>(B, tq), m2 = madd<SP>(A, tq), m3 = madd<SP>(B, tp);
            if constexpr (SP) {
                M[r0] = acs_u8<(r0 & 7)>(m0, m1, acc[r0 >> 3]);
                M[r1] = acs_u8<(r1 & 7)>(m2, m3, acc[r1 >> 3]);
            } else {
                M[r0] = acs(m0, m1, W0);
                M[r1] = acs(m2, m3, W1);
            }
        };
        if constexpr (SP) {
            sfor<NR / 2>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int r0 = ((i >> rb) << (rb + 1)) | (i & ((1 << rb) - 1));
                i16x2 w0, w1;
                pair(std::integral_constant<int, r0>{}, w0, w1);
            });
        } else if constexpr (rb == 0) {
            // the butterfly pair (2i, 2i+1) is also the sign-gathering pair
            sfor<NR / 2>([&](auto I) {
                constexpr int r0 = 2 * decltype(I)::value;
                i16x2 w0, w1;
                pair(std::integral_constant<int, r0>{}, w0, w1);
                put_signs<r0>(w0, w1, acc, sm);
            });
        } else {
            // two butterfly pairs (r0, r1), (r0+1, r1+1) feed the sign-gathering pairs (r0, r0+1) and (r1, r1+1)
            sfor<NR / 4>([&](auto I) {
                constexpr int i = 2 * decltype(I)::value;
                constexpr int r0 = ((i >> rb) << (rb + 1)) | (i & ((1 << rb) - 1));
                constexpr int r1 = r0 | (1 << rb);
                static_assert((r0 & 1) == 0, "even/odd neighbours");
                i16x2 wa0, wa1, wb0, wb1;
                pair(std::integral_constant<int, r0>{}, wa0, wa1);
                pair(std::integral_constant<int, r0 + 1>{}, wb0, wb1);
                put_signs<r0>(wa0, wb0, acc, sm);
                put_signs<r1>(wa1, wb1, acc, sm);
            });
        }
    } else {
        // half stage: old[j] is the low field, old[j+H] the high field of the same register
        auto half = [&](auto R0, i16x2 &W) {
            constexpr int r0 = decltype(R0)::value;
            constexpr unsigned cr = cls(rotl14(((unsigned)r0 << 1) << FSH, PHI));
            unsigned tpp, tqq;
            fields(TA[cr & 7u] + TB[cr >> 3], tpp, tqq);  // low fields: t and t' of this butterfly
            const unsigned t = tpp & 0xffffu, tc = tqq & 0xffffu;
            const i16x2 A = M[r0];
            // broadcasting one field to both lanes of a packed add is an op_sel modifier, not an instruction
            const i16x2 Alo = {A.x, A.x}, Ahi = {A.y, A.y};
            const i16x2 lower = madd<SP>(Alo, t | (tc << 16));  // (m0, m2) = old[j] + (t, t')
            const i16x2 upper = madd<SP>(Ahi, tc | (t << 16));  // (m1, m3) = old[j+H] + (t', t)
            if constexpr (SP) M[r0] = acs_u8<(r0 & 7)>(lower, upper, acc[r0 >> 3]);
            else M[r0] = acs(lower, upper, W);
        };
        sfor<NR / 2>([&](auto I) {
            constexpr int r0 = 2 * decltype(I)::value;
            i16x2 w0, w1;
            half(std::integral_constant<int, r0>{}, w0);
            half(std::integral_constant<int, r0 + 1>{}, w1);
            if constexpr (!SP) put_signs<r0>(w0, w1, acc, sm);
        });
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if constexpr (SP) words[w] = ~(acc[2 * w] | (acc[2 * w + 1] << 8));  // accumulators use bits 0..7 / 16..23
        else words[w] = ~acc[w];
    }
}

// LDS image of a frame's metrics: 128 rows (position bits 13..7) of 128 metrics (bits 6..0), each row padded by 16 bytes.
// Group B threads read and write their own row 16 bytes at a time; with rows exactly 256 bytes apart all 64 lanes of such an
// access meet in the same four banks and the transposes cost 22 % of the kernel (18.4 ms with, 14.4 ms without them in a
// timing build).  272-byte rows put 16 consecutive lanes on all 64 banks; group A's 16-bit accesses are unaffected (a
// row is still contiguous, the row offset is an immediate).
constexpr int ROWP = 136;  // int16 per padded row
__device__ __forceinline__ constexpr unsigned img_at(unsigned p) { return (p >> 7) * ROWP + (p & 127u); }
struct Smem {
    alignas(16) int16_t img[128 * ROWP];   // 34 KiB: metrics by position
    int flag[2];      // renormalisation request of the current step (double-buffered by row parity)
    int red[4];       // per-wave minima (double-buffered by row parity in the spiral flavour)
};

// Fused sliding-window decode (the K=15 half): the WIN instantiation of the body below runs the same
// trellis steps from freshly initialised metrics, but its decision rows go to a workgroup-private RING of the last WIN_RING
// rows in global memory -- 320 KiB per resident workgroup, at most 1024 of them: the ring lives in the L2 / Infinity Cache,
// and the 4.2 MB-per-frame history of the exact path is never written -- and whenever the row WIN_DEPTH steps beyond a block
// of WIN_BLOCK payload bits exists, wave 0 walks that block out of the ring with the speculative tree of chainback_spec.hip
// (start state 0; six decisions per cache round trip) and stores its four bytes.  Semantics:
// oracle/viterbi_oracle.c vo_chainback_windowed (depth 96, blocks of 32 bits).
constexpr int WIN_DEPTH = 96, WIN_BLOCK = 32, WIN_RING = 160;  // 160 rows x 2 KiB = 320 KiB per resident workgroup
static_assert(WIN_DEPTH + WIN_BLOCK + 2 * NB <= WIN_RING, "ring too small");
struct DecodeWindowedK15Args {
    const unsigned char *syms;
    size_t sym_stride;
    int nsteps, nframes;
    unsigned char *data;
    size_t data_stride;
    unsigned nbits;
    unsigned *ring;  // [gridDim.x][WIN_RING][512] words
};

// one block of the windowed traceback, by ONE wave: payload bits [lo, hi) walked from row top-1 (state 0) down to row lo+NB
template <bool SIGN_BYTES>
__device__ __forceinline__ unsigned walk_block_k15(const unsigned *ring, int top, unsigned lo, unsigned hi, unsigned lane) {
    constexpr int DEPTH = 6;
    const unsigned node = lane + 1;
    const int d = 31 - __clz(node);
    const unsigned path = node - (1u << d);
    const bool live = lane < 63;
    unsigned st0 = 0, v = 0;
    int i = top - NB;  // payload-bit index + 1 of the first row visited (row = bit + NB)
    while (i > (int)lo) {
        unsigned st = st0;
        for (int s = 0; s < DEPTH - 1; s++)
            if (s < d) st = (st >> 1) | (((path >> (d - 1 - s)) & 1u) << (NB - 1));
        const int bi = i - 1 - d;
        unsigned k = 0;
        if (live && bi >= (int)lo) {
            const int r = bi + NB;
            const int rot = (r + 1) % NB;
            const unsigned p = rot == 0 ? st : (((st >> rot) | (st << (NB - rot))) & (N - 1u));
            const int phi = rot == 0 ? NB - 1 : rot - 1;
            const unsigned t = phi < 7 ? (p & 127u) : (p >> 7), q = phi < 7 ? (p >> 7) : (p & 127u);
            const unsigned rho = q >> 1, h = q & 1u;
            // written by this workgroup a moment ago: read past the (not yet refreshed) vector L1
            const unsigned w = __builtin_nontemporal_load(ring + (long)(r % WIN_RING) * 512 + (rho >> 4) * 128 + t);
            k = (w >> k15_decision_bit(SIGN_BYTES, rho, h)) & 1u;
        }
        unsigned cur = 1;
        const int left = i - (int)lo, nres = left < DEPTH ? left : DEPTH;
        for (int s = 0; s < nres; s++) {
            const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)k, (int)(cur - 1)) & 1u;
            --i;
            st0 = (st0 >> 1) | (kb << (NB - 1));
            if (i < (int)hi) v |= kb << (31 - (i - (int)lo));  // MSB-first inside the block
            cur = 2 * cur + kb;
        }
    }
    return v;
}

// WIN = false: the exact path (AcsK15Args).  WIN = true: the fused windowed decode (DecodeWindowedK15Args), persistent
// over frames blockIdx.x, blockIdx.x + gridDim.x, ...
template <bool SP, bool WIN, class Args>
__device__ __forceinline__ void acs_k15_body(const Args &a) {
    __shared__ Smem sm;
    const SignMasks sgm;
    const unsigned tid = threadIdx.x;
    // group-A transposes (see run_group): byte address of this lane's dword in row (tid & 1), and the v_perm_b32 selector that
    // picks (own dword's half, partner dword's half) for an even lane / (partner's high half, own high half) for an odd one
    const unsigned pair_addr = (tid & 1u) * (ROWP * 2) + (tid & ~1u) * 2;
    const unsigned pair_sel = (tid & 1u) ? 0x07060302u : 0x01000504u;
    const ClassOffsets co(tid);
  for (long f = blockIdx.x; f < (WIN ? (long)a.nframes : (long)blockIdx.x + 1); f += gridDim.x) {
    int row0, row_end, phi0;
    int16_t *gm = nullptr;
    unsigned *drow = nullptr;
    unsigned *ring = nullptr;
    if constexpr (WIN) {
        row0 = 0;
        row_end = a.nsteps;
        phi0 = 0;
        ring = a.ring + (long)blockIdx.x * WIN_RING * 512;
        // init_viterbi615_sse2 (viterbi615_sse2.cpp:33-39): every state init_all, state 0 init_start; spiral: (m<<7)|0x7f fields
        using CT = std::conditional_t<SP, Spiral615, Code615>;
        const int ia = SP ? (int)sp_field((unsigned)CT::init_all) : CT::init_all;
        const int is = SP ? (int)sp_field((unsigned)CT::init_start) : CT::init_start;
        __syncthreads();  // the previous frame's last reads of the image are done
        for (unsigned p = tid; p < (unsigned)N; p += THREADS) sm.img[img_at(p)] = (int16_t)(p == 0 ? is : ia);
    } else {
        row0 = a.row0;
        row_end = a.row0 + a.nsteps;
        phi0 = row0 % NB;
        gm = a.metrics + f * (long)N;
        for (unsigned p = tid; p < (unsigned)N; p += THREADS) {
            const unsigned st = phi0 == 0 ? p : (((p << phi0) | (p >> (NB - phi0))) & (N - 1u));
            sm.img[img_at(p)] = SP ? (int16_t)sp_field((unsigned)gm[st] & 255u) : gm[st];  // spiral: (m<<7)|0x7f fields
        }
        drow = reinterpret_cast<unsigned *>(a.dec) + (f * a.cap_rows + row0) * 512L + tid;
    }
    const unsigned char *sp = a.syms + f * (long)a.sym_stride;
    const long lim = (long)a.nsteps * R;
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.syms) | a.sym_stride) & 3) == 0 && ((phi0 * R) & 3) == 0;
    unsigned nextb = 0;  // WIN: next block to emit
    unsigned pend = 0;   // spiral615: the minimum the last renormalisation owes, subtracted inside the next step's branch metrics
    __syncthreads();

    i16x2 M[NR];
    for (int rbase = row0 - phi0; rbase < row_end; rbase += NB) {
        // this period's 14 x 6 symbol bytes (uniform across the workgroup)
        unsigned cur[21];
        const long off = (long)(rbase - row0) * R;
        if (aligned && off >= 0 && off + 84 <= lim) {
            const unsigned *p32 = reinterpret_cast<const unsigned *>(sp + off);
#pragma unroll
            for (int w = 0; w < 21; w++) cur[w] = p32[w];
        } else {
#pragma unroll
            for (int w = 0; w < 21; w++) {
                unsigned v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const long ob = off + 4 * w + k;
                    if (ob >= 0 && ob < lim) v |= (unsigned)sp[ob] << (8 * k);
                }
                cur[w] = v;
            }
        }
        auto run_group = [&](auto GSEL) {
            constexpr int G = decltype(GSEL)::value;  // 0: phases 0..6 (free bits 7..13), 1: phases 7..13 (free bits 0..6)
            // gather this thread's 128 positions
            if constexpr (G == 0) {
                // A thread of this group owns one COLUMN of the image: 128 metrics 272 bytes apart.  Reading them one by one
                // is 128 16-bit LDS instructions, and the LDS pipe takes as long for a 16-bit wave access as for a 32-bit one
                // (the group's two transposes were 2.2 of 16.5 ms).  Lanes t and t ^ 1 own neighbouring columns, i.e. the
                // two halves of the same dwords: the even lane fetches the dword of row 2 r0, the odd lane that of row
                // 2 r0 + 1, the pair swaps them with one DPP move and each lane picks its two halves with one v_perm_b32.
                // (The fused windowed decode keeps the 16-bit accesses: with its ring bookkeeping live as well the extra
                // registers of the pair exchange cost it more than the LDS instructions saved -- 2.37 vs 2.21 Gsym/s.)
#pragma unroll
                for (int r0 = 0; r0 < NR; r0++) {
                    if constexpr (WIN) {
                        const unsigned lo = (unsigned short)sm.img[(2 * r0) * ROWP + tid];
                        const unsigned hi = (unsigned short)sm.img[(2 * r0 + 1) * ROWP + tid];
                        M[r0] = as_v(lo | (hi << 16));
                    } else {
                        const unsigned x = *reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned char *>(sm.img) + pair_addr + (2 * r0) * ROWP * 2);
                        const unsigned px = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
                        M[r0] = as_v(__builtin_amdgcn_perm(x, px, pair_sel));
                    }
                }
            } else {
                const uint4 *src = reinterpret_cast<const uint4 *>(&sm.img[tid * ROWP]);
#pragma unroll
                for (int q = 0; q < NR / 4; q++) {
                    const uint4 v = src[q];
                    M[4 * q] = as_v(v.x);
                    M[4 * q + 1] = as_v(v.y);
                    M[4 * q + 2] = as_v(v.z);
                    M[4 * q + 3] = as_v(v.w);
                }
            }
            __syncthreads();  // every thread has its registers before anyone overwrites the image
            sfor<7>([&](auto I) {
                constexpr int PHI = G * 7 + decltype(I)::value;
                const int r = rbase + PHI;
                if (r >= row0 && r < row_end) {  // workgroup-uniform
                    unsigned sraw[R];
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        const int idx = PHI * R + q;
                        sraw[q] = (cur[idx >> 2] >> (8 * (idx & 3))) & 255u;
                    }
                    unsigned words[4];
                    stage<SP, PHI>(M, sraw, co, words, sgm, pend);
                    if constexpr (WIN) {
                        unsigned *rr = ring + (long)(r % WIN_RING) * 512 + tid;
#pragma unroll
                        for (int w = 0; w < 4; w++) rr[w * 128] = words[w];
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; w++) __builtin_nontemporal_store(words[w], drow + w * 128);  // written once, read much later
                        drow += 512;
                    }
                    if constexpr (SP) {
                        // spiral615.cpp:31-40,269: if new[0] > 74 subtract the minimum -- it fires on nearly every step, so
                        // the workgroup minimum is formed unconditionally behind the one barrier of the step.  Nothing is
                        // subtracted here: the registers keep this step's values and the next step's branch metrics carry
                        // the minimum (64 v_pk_sub per thread and step saved); the final store below settles the last one.
                        u16x2 mq[4] = {(u16x2)M[0], (u16x2)M[1], (u16x2)M[2], (u16x2)M[3]};  // four chains: a single one is 63 dependent instructions
#pragma unroll
                        for (int i = 4; i < NR; i++) mq[i & 3] = __builtin_elementwise_min(mq[i & 3], (u16x2)M[i]);
                        const u16x2 mn = __builtin_elementwise_min(__builtin_elementwise_min(mq[0], mq[1]), __builtin_elementwise_min(mq[2], mq[3]));
                        const unsigned m = wave_min(min((unsigned)mn.x, (unsigned)mn.y));
                        if ((tid & 63u) == 0) sm.red[2 * (r & 1) + (tid >> 6)] = (int)m;
                        if (tid == 0) sm.flag[r & 1] = (as_u32(M[0]) & 0xffffu) > sp_field((unsigned)Spiral615::renorm_thr);
                        __syncthreads();
                        const unsigned owed = sm.flag[r & 1] ? (unsigned)min(sm.red[2 * (r & 1)], sm.red[2 * (r & 1) + 1]) >> 7 : 0u;
                        pend = (unsigned)__builtin_amdgcn_readfirstlane((int)owed);
                    } else {
                    // renormalise when new[0] >= SHRT_MAX-12750; state 0 is position 0 = thread 0, register 0, low field
                    if (tid == 0) sm.flag[r & 1] = ((int)(short)(as_u32(M[0]) & 0xffffu)) >= Code615::renorm_thr;
                    __syncthreads();
                    if (sm.flag[r & 1]) {
                        i16x2 mn = M[0];
#pragma unroll
                        for (int i = 1; i < NR; i++) mn = __builtin_elementwise_min(mn, M[i]);
                        int m = wave_min(min((int)mn.x, (int)mn.y));
                        if ((tid & 63u) == 0) sm.red[tid >> 6] = m;
                        __syncthreads();
                        m = min(sm.red[0], sm.red[1]);
                        const unsigned adj = (unsigned)(m + 32768) & 0xffffu;  // min - SHRT_MIN       :166-172
                        const u16x2 av = {(unsigned short)adj, (unsigned short)adj};
#pragma unroll
                        for (int i = 0; i < NR; i++) M[i] = (i16x2)((u16x2)M[i] - av);  // sub_epi16 wraps   :181-182
                    }
                    }
                }
            });
            // scatter back to the position image
            if constexpr (G == 0) {
                // the reverse: the pair swaps registers, the even lane writes the dword of row 2 r0 (its own low half, the
                // partner's low half), the odd lane that of row 2 r0 + 1 (the partner's high half, its own)
#pragma unroll
                for (int r0 = 0; r0 < NR; r0++) {
                    if constexpr (WIN) {
                        sm.img[(2 * r0) * ROWP + tid] = M[r0].x;
                        sm.img[(2 * r0 + 1) * ROWP + tid] = M[r0].y;
                    } else {
                        const unsigned m = as_u32(M[r0]);
                        const unsigned pm = (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0xB1, 0xf, 0xf, true);
                        *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(sm.img) + pair_addr + (2 * r0) * ROWP * 2) = __builtin_amdgcn_perm(m, pm, pair_sel);
                    }
                }
            } else {
                uint4 *dst = reinterpret_cast<uint4 *>(&sm.img[tid * ROWP]);
#pragma unroll
                for (int q = 0; q < NR / 4; q++)
                    dst[q] = make_uint4(as_u32(M[4 * q]), as_u32(M[4 * q + 1]), as_u32(M[4 * q + 2]), as_u32(M[4 * q + 3]));
            }
            __syncthreads();
        };
        run_group(std::integral_constant<int, 0>{});
        run_group(std::integral_constant<int, 1>{});
        if constexpr (WIN) {
            // every block whose top row now exists; the last barrier of run_group has drained this period's ring stores
            const int rows_done = rbase + NB < row_end ? rbase + NB : row_end;
            const unsigned nblocks = (a.nbits + WIN_BLOCK - 1) / WIN_BLOCK;
            while (nextb < nblocks) {
                const unsigned lo = nextb * WIN_BLOCK, hi = lo + WIN_BLOCK < a.nbits ? lo + WIN_BLOCK : a.nbits;
                const int top = (int)hi + NB + WIN_DEPTH < row_end ? (int)hi + NB + WIN_DEPTH : row_end;
                if (top > rows_done) break;
                if (tid < 64) {  // wave 0 walks; wave 1 waits at the next period's first barrier
                    const unsigned v = walk_block_k15<!SP>(ring, top, lo, hi, tid);
                    if (tid == 0) {
                        unsigned char *out = a.data + f * (long)a.data_stride + lo / 8;
                        const unsigned nbytes = (hi - lo + 7) / 8;
                        for (unsigned m = 0; m < nbytes; m++) out[m] = (unsigned char)(v >> (24 - 8 * m));
                    }
                }
                nextb++;
            }
        }
    }

    if constexpr (!WIN) {
        const int phie = row_end % NB;
        for (unsigned p = tid; p < (unsigned)N; p += THREADS) {
            const unsigned st = phie == 0 ? p : (((p << phie) | (p >> (NB - phie))) & (N - 1u));
            gm[st] = SP ? (int16_t)((((unsigned)(unsigned short)sm.img[img_at(p)]) >> 7) - pend) : sm.img[img_at(p)];
        }
    }
  }
}

template <bool SP>
__global__ __launch_bounds__(THREADS, 2) void acs_k15_kernel(AcsK15Args a) {
    acs_k15_body<SP, false>(a);
}

template <bool SP>
__global__ __launch_bounds__(THREADS, 2) void decode_windowed_k15_kernel(DecodeWindowedK15Args a) {
    acs_k15_body<SP, true>(a);
}

}  // namespace k15

#ifndef VH_JIT_KERNEL
bool k15_poly_supported(const int *poly) {
    for (int r = 0; r < 6; r++)
        if (poly[r] != k15::POLY[r]) return false;
    return true;
}

hipError_t launch_acs_k15(const AcsK15Args &a, bool spiral, hipStream_t stream) {
    if (spiral) hipLaunchKernelGGL(k15::acs_k15_kernel<true>, dim3(a.nframes), dim3(k15::THREADS), 0, stream, a);
    else hipLaunchKernelGGL(k15::acs_k15_kernel<false>, dim3(a.nframes), dim3(k15::THREADS), 0, stream, a);
    return hipGetLastError();
}

// fused windowed decode: a persistent grid of at most 1024 workgroups (four per CU), each with its own decision ring
size_t windowed_k15_ring_bytes(int nframes) { return (size_t)(nframes < 1024 ? nframes : 1024) * k15::WIN_RING * 2048; }
void windowed_k15_params(int *depth, int *block) {
    *depth = k15::WIN_DEPTH;
    *block = k15::WIN_BLOCK;
}
hipError_t launch_decode_windowed_k15(bool spiral, const unsigned char *syms, size_t sym_stride, int nsteps, int nframes, unsigned char *data,
                                      size_t data_stride, unsigned nbits, unsigned *ring, hipStream_t stream) {
    const k15::DecodeWindowedK15Args a{syms, sym_stride, nsteps, nframes, data, data_stride, nbits, ring};
    const int grid = nframes < 1024 ? nframes : 1024;
    if (spiral) hipLaunchKernelGGL(k15::decode_windowed_k15_kernel<true>, dim3(grid), dim3(k15::THREADS), 0, stream, a);
    else hipLaunchKernelGGL(k15::decode_windowed_k15_kernel<false>, dim3(grid), dim3(k15::THREADS), 0, stream, a);
    return hipGetLastError();
}

#endif  // !VH_JIT_KERNEL

}  // namespace vh

#ifdef VH_JIT_KERNEL
extern "C" __global__ __launch_bounds__(128, 2) void vh_jit_acs_k15(vh::AcsK15Args a) { vh::k15::acs_k15_body<VH_JIT_SPIRAL, false>(a); }
#endif
