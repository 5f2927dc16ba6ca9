// acs_k24t.hip -- K=24 r=1/2 ACS update in TWO passes over the 16 MiB metric array per 23 trellis steps.
//
// Replaces update_viterbi224_blk_sse2 (ka9q_libfec_port/viterbi224_sse2.cpp:135-258) for the harness polynomials
// (src/main.cpp:415).  acs_k24.hip moves the metrics through HBM once per step, acs_k24f.hip once per 4 or 7 steps
// (5 passes per 23 steps); here a workgroup keeps a whole tile of the rotating in-place trellis on chip -- positions in
// VGPRs (64 per thread), regrouped through a tile-sized LDS image -- and runs 9 (pass H) or 14 (pass L) consecutive phases before the
// tile goes back: 2 x 32 MiB of metric traffic + 23 MiB of decisions per 23 steps instead of 5 x 32 + 23 (k24f) or
// 23 x 33 (k24).  Geometry, register groups and the decision-row layout: k24t_layout.h.
//
// What makes two passes possible with whole 128-byte lines on both sides: the tile of pass H carries the six lowest
// position bits as passengers (runs of 64 positions), and those bits belong to pass L's tile, so HBM keeps the natural
// position order between the passes and both passes read and write complete lines.  Every LDS access of the regrouping
// is 32 or 64 bits wide (the half bit of a packed register is position bit 0 in every group), swizzled where lanes would
// otherwise meet in a bank.
//
// Arithmetic exactly as acs_k24f.hip: adds_epi16 / cmpgt / min_epi16 (viterbi224_sse2.cpp:159-194), decisions from the
// sign of the saturating difference; renormalisation (:226-246) is speculative -- the thread that owns position 0
// raises flags[PENDING] = row + 1 when new[0] >= 25000, later passes return at once, the host replays the raising pass
// up to that row, renormalises and continues.  Passes ping-pong between two buffers so that replay is possible.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "k24t_layout.h"
#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {
namespace k24t {

typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr int NB = 23;
constexpr unsigned N = 1u << NB;
#ifdef VH_JIT_KERNEL
constexpr int POLY[2] = VH_JIT_POLY;  // runtime specialisation (jit.hip): the caller's polynomials
#else
constexpr int POLY[2] = {062650457, 062650455};  // src/main.cpp:415
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
constexpr unsigned rotl23(unsigned x, int s) {
    s %= NB;
    return s == 0 ? x : (((x << s) | (x >> (NB - s))) & (N - 1u));
}
constexpr unsigned cls(unsigned j) {  // bit r = parity((2j) & poly[r])                 viterbi224_sse2.cpp:68-73
    return (popc((2u * j) & (unsigned)POLY[0]) & 1u) | ((popc((2u * j) & (unsigned)POLY[1]) & 1u) << 1);
}
__device__ __forceinline__ unsigned as_u32(i16x2 x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ i16x2 as_v(unsigned x) { return __builtin_bit_cast(i16x2, x); }

// packed ACS, tie -> lower (cmpgt_epi16 then min_epi16, viterbi224_sse2.cpp:190-194); w < 0 <=> decision 1
__device__ __forceinline__ i16x2 acs(i16x2 lower, i16x2 upper, i16x2 &w) {
    w = __builtin_elementwise_sub_sat(upper, lower);
    return __builtin_elementwise_min(lower, upper);
}
// sign bytes of the register pair (RE, RE+1) -> bits k24t_decision_bit(rho, half) of the 16-register word
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

// One trellis step at phase PHI on the registers of group G.  pt = thread part of the position (k24t_thread_base).
template <int G, int PHI>
__device__ __forceinline__ void stage(i16x2 (&M)[k24t_nr(G)], unsigned s0, unsigned s1, unsigned pt, unsigned (&acc)[k24t_nr(G) / 16], const SignMasks &sm) {
    constexpr int b = NB - 1 - PHI;  // position bit paired at this phase
    constexpr int NR = k24t_nr(G);
    // class offset of the thread part, folded into the symbols (conditional complement)
    const unsigned jt = PHI == 0 ? pt : (((pt << PHI) | (pt >> (NB - PHI))) & (N - 1u));
    const unsigned c0 = __popc((2u * jt) & (unsigned)POLY[0]) & 1u, c1 = __popc((2u * jt) & (unsigned)POLY[1]) & 1u;
    const unsigned a0 = s0 ^ (c0 ? 255u : 0u), a1 = s1 ^ (c1 ? 255u : 0u);
    const unsigned x0 = a0 ^ 255u, x1 = a1 ^ 255u;
    const unsigned T[4] = {a0 + a1, x0 + a1, a0 + x1, x0 + x1};  // xor + add   viterbi224_sse2.cpp:159
    constexpr unsigned COMP = (unsigned)Code224::bm_comp;
#pragma unroll
    for (int i = 0; i < NR / 16; i++) acc[i] = 0;

    if constexpr (b >= 1) {
        constexpr int rb = k24t_regbit(G, b);
        static_assert(rb >= 0 && (1 << rb) < NR, "phase outside this group");
        constexpr unsigned ch = cls(rotl23(1u, PHI));  // class of the half bit (position bit 0)
        unsigned TP[4], TQ[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            TP[c] = T[c] | (T[c ^ ch] << 16);
            TQ[c] = COMP * 0x10001u - TP[c];
        }
        auto pair = [&](auto R0, i16x2 &W0, i16x2 &W1) {
            constexpr int r0 = decltype(R0)::value;
            constexpr int r1 = r0 | (1 << rb);
            constexpr unsigned cr = cls(rotl23(k24t_spos(G, (unsigned)r0), PHI));
            const i16x2 A = M[r0], B = M[r1];
            const i16x2 tp = as_v(TP[cr]), tq = as_v(TQ[cr]);
            const i16x2 m0 = __builtin_elementwise_add_sat(A, tp), m1 = __builtin_elementwise_add_sat(B, tq);  // adds_epi16 :163-166
            const i16x2 m2 = __builtin_elementwise_add_sat(A, tq), m3 = __builtin_elementwise_add_sat(B, tp);
            M[r0] = acs(m0, m1, W0);
            M[r1] = acs(m2, m3, W1);
        };
        if constexpr (rb == 0) {
            sfor<NR / 2>([&](auto I) {
                constexpr int r0 = 2 * decltype(I)::value;
                i16x2 w0, w1;
                pair(std::integral_constant<int, r0>{}, w0, w1);
                put_signs<r0>(w0, w1, acc, sm);
            });
        } else {
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
        // half stage (position bit 0): old[j] low field, old[j+H] high field
        auto half = [&](auto R0, i16x2 &W) {
            constexpr int r0 = decltype(R0)::value;
            constexpr unsigned cr = cls(rotl23(k24t_spos(G, (unsigned)r0), PHI));
            const unsigned t = T[cr], tc = COMP - t;
            const i16x2 A = M[r0];
            const i16x2 Alo = {A.x, A.x}, Ahi = {A.y, A.y};  // op_sel broadcast, not an instruction
            const i16x2 lower = __builtin_elementwise_add_sat(Alo, as_v(t | (tc << 16)));  // (m0, m2) = old[j] + (t, t')
            const i16x2 upper = __builtin_elementwise_add_sat(Ahi, as_v(tc | (t << 16)));  // (m1, m3) = old[j+H] + (t', t)
            M[r0] = acs(lower, upper, W);
        };
        sfor<NR / 2>([&](auto I) {
            constexpr int r0 = 2 * decltype(I)::value;
            i16x2 w0, w1;
            half(std::integral_constant<int, r0>{}, w0);
            half(std::integral_constant<int, r0 + 1>{}, w1);
            put_signs<r0>(w0, w1, acc, sm);
        });
    }
}

// The stages [s_lo, s_hi) of the PASS (0-based inside the pass) that belong to group G, run on M.  PF = first phase of the pass.
// MODE (timing experiments only, tools/k24t_probe.sh; results are wrong unless MODE == 0): 1 = data movement only (loads,
// LDS regrouping, stores; no trellis stages), 2 = no global metric loads / stores (stages and row stores only), 3 = global
// loads and stores only, 4 = as 2 without the decision-row stores, 5 = as 2 without the LDS regrouping traffic
template <int G, int PF, bool FULL, int MODE, int NSY>
__device__ __forceinline__ void run_group(i16x2 (&M)[k24t_nr(G)], const unsigned (&sy)[NSY], unsigned pt, unsigned tile, unsigned tid,
                                          unsigned char *__restrict__ rows, int rel_row0, int s_lo, int s_hi, int *__restrict__ flags,
                                          int &pending, const SignMasks &sm) {
    constexpr int NR = k24t_nr(G), WPT = NR / 16, P0 = k24t_first_phase(G), NP = k24t_nphases(G);
    if constexpr (MODE == 1 || MODE == 3) return;
    sfor<NP>([&](auto I) {
        constexpr int PHI = P0 + decltype(I)::value;
        constexpr int S = PHI - PF;  // stage index inside the pass
        if (FULL || (S >= s_lo && S < s_hi)) {  // grid-uniform
            unsigned acc[WPT];
            stage<G, PHI>(M, sy[S] & 255u, sy[S] >> 8, pt, acc, sm);
            unsigned *row = reinterpret_cast<unsigned *>(rows + (size_t)S * (N / 8)) + ((size_t)tile * k24t_threads(G) + tid) * WPT;
            // written once, read by the chainback much later: keep the rows out of the caches the metrics live in
            static_assert(WPT == 2, "two words per thread and row");
            if (MODE != 4 || (acc[0] ^ acc[1]) == 0x12345u) {
                __builtin_nontemporal_store(acc[0], row);
                __builtin_nontemporal_store(acc[1], row + 1);
            }
            if (MODE == 0 && tile == 0 && tid == 0) {  // state 0 is position 0 in every phase: tile 0, thread 0, register 0, low field
                const int new0 = (int)(short)(as_u32(M[0]) & 0xffffu);
                if (new0 >= Code224::renorm_thr && pending == 0) flags[K24F_PENDING] = pending = rel_row0 + S + 1;
            }
        }
    });
}

// symbols of the stages that run, fetched once next to the metric loads (clamped: a truncated pass reads nothing else)
template <int NP, bool FULL>
__device__ __forceinline__ void load_symbols(const unsigned char *syms, int s_lo, int s_hi, unsigned (&sy)[NP]) {
#pragma unroll
    for (int S = 0; S < NP; S++) {
        const int sc = FULL ? S : min(max(S, s_lo), s_hi - 1);
        sy[S] = (unsigned)syms[2 * sc] | ((unsigned)syms[2 * sc + 1] << 8);
    }
}

// ---------------------------------------------------------------------------------------------------- pass H (phases 0..8)
// LDS image of the tile: 512 rows (position bits 22..14) x 32 dwords (position bits 5..1; bit 0 = field of the dword).
// H2 reads 8 bytes per lane with 16 lanes per row and neighbouring 16-lane groups 16 rows apart: rows r and r ^ 16 swap
// their parity so that such a pair lands in different halves of the 64 banks.
__device__ __forceinline__ unsigned h_row(unsigned row) { return row ^ ((row >> 4) & 1u); }

// NT: bit 0 = metric loads, bit 1 = metric stores carry the non-temporal hint.  Stores: a pass writes its 16 MiB of metrics
// in one burst at its very end, and with plain (write-back) stores the dirty lines are still in the L2s when the kernel ends,
// so the end-of-kernel write-back -- which the next pass has to wait for -- pays for them again: 3.95 -> 3.39 ms per 2071-step
// frame with the hint.  Loads: no difference (kept as a switch, VHIP_K24T_NT, for the record).
template <class T>
__device__ __forceinline__ T ld_metric(const T *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Renormalisation folded into the passes (viterbi224_sse2.cpp:226-246: when new[0] >= 25000 the minimum over all metrics,
// minus SHRT_MIN, is subtracted from every metric).  The host replays the pass that raised the flag up to the flagged row;
// that replay also forms the minimum of its final metrics (per workgroup, one atomicMin each) and the NEXT pass subtracts it
// while it loads -- no separate min / subtract kernels, no host synchronisation.  `ctl` (0 for an ordinary pass; only the
// FULL = false instantiations look at it):
//   bits 1..0  1 + slot that receives this pass's minimum            (the replay of the raising pass)
//   bits 3..2  1 + slot whose minimum is subtracted at load          (the first pass behind a renormalised row)
//   bits 5..4  1 + slot to reset to "no minimum yet"                 (slots rotate 0,1,2: the pass that applies event n's
//                                                                     minimum clears the slot event n+2 will use)
//   bit  6     clear the pending flag at the end                     (the replay: younger cancelled passes are behind it)
enum { K24T_CTL_MIN = 0, K24T_CTL_ADJ = 2, K24T_CTL_RESET = 4, K24T_CTL_CLEAR = 64 };
__device__ __forceinline__ void ctl_apply_adjust(i16x2 (&M)[32], const int *flags, int ctl) {
    const int slot = ((ctl >> K24T_CTL_ADJ) & 3) - 1;
    if (slot >= 0) {
        const unsigned short adj = (unsigned short)(flags[K24F_MIN + slot] + 32768);  // min - SHRT_MIN          :240
        const u16x2 av = {adj, adj};
#pragma unroll
        for (int q = 0; q < 32; q++) M[q] = (i16x2)((u16x2)M[q] - av);                 // sub_epi16 wraps         :245-246
    }
}
template <int NWAVES>
__device__ __forceinline__ void ctl_finish(const i16x2 (&M)[32], int *flags, int ctl, unsigned tile, unsigned tid, int *wmin) {
    const int mslot = ((ctl >> K24T_CTL_MIN) & 3) - 1;
    if (mslot >= 0) {
        i16x2 mn = M[0];
#pragma unroll
        for (int q = 1; q < 32; q++) mn = __builtin_elementwise_min(mn, M[q]);
        int m = wave_min(min((int)mn.x, (int)mn.y));
        if ((tid & 63u) == 0) wmin[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < NWAVES; w++) m = min(m, wmin[w]);
            atomicMin(&flags[K24F_MIN + mslot], m);
        }
    }
    if (tile == 0 && tid == 0) {
        const int rslot = ((ctl >> K24T_CTL_RESET) & 3) - 1;
        if (rslot >= 0) flags[K24F_MIN + rslot] = 0x7fffffff;
        if (ctl & K24T_CTL_CLEAR) flags[K24F_PENDING] = 0;
    }
}

// Timing build only (tools/k24t_probe.sh): delay the workgroups selected by `pattern` before they issue their metric loads, so
// that one co-resident workgroup's loads and stores fall into the other's arithmetic.  ctl bits 8..15: s_sleep units of 64
// cycles; bits 16..17: which half waits (1: blockIdx >= grid/2, 2: odd (blockIdx >> 3), 3: odd blockIdx).
__device__ __forceinline__ void timing_stagger(int ctl) {
#ifdef VHIP_TIMING_BUILD
    const int units = (ctl >> 8) & 255, pattern = (ctl >> 16) & 3;
    const bool late = pattern == 1 ? blockIdx.x >= gridDim.x / 2 : pattern == 2 ? ((blockIdx.x >> 3) & 1u) : pattern == 3 ? (blockIdx.x & 1u) : false;
    if (late)
        for (int i = 0; i < units; i++) __builtin_amdgcn_s_sleep(1);
#else
    (void)ctl;
#endif
}

template <bool FULL, int MODE, int NT = 2>
__device__ __forceinline__ void pass_h_body(const int16_t *oldm, int16_t *__restrict__ newm, unsigned char *__restrict__ rows,
                                            const unsigned char *syms, int rel_row0, int s_lo, int s_hi, int *flags,
                                            K24Report mirror, int ctl) {
    // rows/syms point at the row of the pass's first phase; rel_row0 = that row's index in the call
    int pending = flags[K24F_PENDING];  // looked at below, behind the metric loads
    const SignMasks sm;
    __shared__ unsigned img[512 * 32];
    __shared__ int wmin[8];
    const unsigned tile = blockIdx.x, tid = threadIdx.x;
    unsigned sy[9];
    load_symbols<9, FULL>(syms, s_lo, s_hi, sy);
    timing_stagger(ctl);
    i16x2 M[32];
    {
        const unsigned pt = k24t_thread_base(K24T_H1, tile, tid);
        // issue order (0, 16, 1, 17, ...): the pass's first phase pairs register q with q ^ 16, and the 16 MiB of a pass arrive
        // over ~3 us, so its first butterflies can start on the first lines instead of waiting for the last
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int q = (i >> 1) | ((i & 1) << 4);
            M[q] = (MODE == 2 || MODE == 4 || MODE == 5) ? as_v(pt + q) : as_v(ld_metric(reinterpret_cast<const unsigned *>(oldm + (pt | ((unsigned)q << 18))), NT & 1));
        }
        asm volatile("" : "+s"(pending) : : "memory");  // the flag (a dependent scalar load) is examined behind the loads
        if (pending != 0 && pending - 1 < rel_row0 + s_lo) {
            if (tile == 0 && tid == 0) k24_report(mirror, pending);
            return;
        }
        if constexpr (!FULL) ctl_apply_adjust(M, flags, ctl);
        run_group<K24T_H1, 0, FULL, MODE>(M, sy, pt, tile, tid, rows, rel_row0, s_lo, s_hi, flags, pending, sm);
        // regroup: row = bits 22..14 = (q << 4) | t4, dword = t5
        const unsigned t4 = tid >> 5, t5 = tid & 31u;
        if constexpr (MODE != 3 && MODE != 5)
#pragma unroll
        for (int q = 0; q < 32; q++) img[h_row(((unsigned)q << 4) | t4) * 32 + t5] = as_u32(M[q]);
    }
    if constexpr (MODE != 3) __syncthreads();
    {
        const unsigned pt = k24t_thread_base(K24T_H2, tile, tid);
        const unsigned t5 = tid >> 4, t4 = tid & 15u;
        if constexpr (MODE != 3 && MODE != 5)
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const uint2 d = *reinterpret_cast<const uint2 *>(&img[h_row((t5 << 4) | (unsigned)v) * 32 + 2 * t4]);
            M[2 * v] = as_v(d.x);
            M[2 * v + 1] = as_v(d.y);
        }
        run_group<K24T_H2, 0, FULL, MODE>(M, sy, pt, tile, tid, rows, rel_row0, s_lo, s_hi, flags, pending, sm);
        if (tile == 0 && tid == 0) k24_report(mirror, pending);  // the last pass of a host batch reports the flag
        if constexpr (!FULL) ctl_finish<8>(M, flags, ctl, tile, tid, wmin);
#pragma unroll
        for (int v = 0; v < 16; v++)
            if ((MODE != 2 && MODE != 4 && MODE != 5) || as_u32(M[2 * v]) == 0x12345u) {
                unsigned *dst = reinterpret_cast<unsigned *>(newm + (pt | ((unsigned)v << 14)));
                if constexpr (NT & 2) {
                    const u32x2 val = {as_u32(M[2 * v]), as_u32(M[2 * v + 1])};
                    __builtin_nontemporal_store(val, reinterpret_cast<u32x2 *>(dst));
                } else {
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(as_u32(M[2 * v]), as_u32(M[2 * v + 1]));
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------- pass L (phases 9..22)
// LDS image of the tile: 8192 dwords, dword d = position >> 1 (bits 12..8 <-> position bits 13..9, bits 7..3 <-> 8..4,
// bits 2..0 <-> 3..1).  L1 writes with lanes along bits 7..0, L2 reads / writes with lanes along bits 12..8 and 2..0:
// XOR-ing dword bits 7..3 with bits 12..8 keeps every 32-lane group of both shapes on 32 different banks and leaves bits
// 2..0 alone, so L3's 16-byte reads stay aligned quads.
__device__ __forceinline__ unsigned l_swz(unsigned d) { return d ^ (((d >> 8) & 31u) << 3); }

template <bool FULL, int MODE, int NT = 2>
__global__ __launch_bounds__(512) void acs_k24t_pass_h_kernel(const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                                                              int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, int ctl) {
    pass_h_body<FULL, MODE, NT>(oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);
}

template <bool FULL, int MODE, int NT = 2>
__device__ __forceinline__ void pass_l_body(const int16_t *oldm, int16_t *__restrict__ newm, unsigned char *__restrict__ rows,
                                            const unsigned char *syms, int rel_row0, int s_lo, int s_hi, int *flags,
                                            K24Report mirror, int ctl) {
    int pending = flags[K24F_PENDING];
    const SignMasks sm;
    __shared__ __attribute__((aligned(16))) unsigned img[8192];
    __shared__ int wmin[4];
    const unsigned tile = blockIdx.x, tid = threadIdx.x;
    unsigned sy[14];
    load_symbols<14, FULL>(syms, s_lo, s_hi, sy);
    timing_stagger(ctl);
    i16x2 M[32];
    {
        const unsigned pt = k24t_thread_base(K24T_L1, tile, tid);
#pragma unroll
        for (int i = 0; i < 32; i++) {  // issue order (0, 16, 1, 17, ...), as in pass H
            const int q = (i >> 1) | ((i & 1) << 4);
            M[q] = (MODE == 2 || MODE == 4 || MODE == 5) ? as_v(pt + q) : as_v(ld_metric(reinterpret_cast<const unsigned *>(oldm + (pt | ((unsigned)q << 9))), NT & 1));
        }
        asm volatile("" : "+s"(pending) : : "memory");
        if (pending != 0 && pending - 1 < rel_row0 + s_lo) {
            if (tile == 0 && tid == 0) k24_report(mirror, pending);
            return;
        }
        if constexpr (!FULL) ctl_apply_adjust(M, flags, ctl);
        run_group<K24T_L1, 9, FULL, MODE>(M, sy, pt, tile, tid, rows, rel_row0, s_lo, s_hi, flags, pending, sm);
        if constexpr (MODE != 3 && MODE != 5)
#pragma unroll
        for (int q = 0; q < 32; q++) img[l_swz(((unsigned)q << 8) | tid)] = as_u32(M[q]);
    }
    if constexpr (MODE != 3) __syncthreads();
    {
        const unsigned pt = k24t_thread_base(K24T_L2, tile, tid);
        const unsigned t5 = tid >> 3, t3 = tid & 7u;
        if constexpr (MODE != 3 && MODE != 5)
#pragma unroll
        for (int q = 0; q < 32; q++) M[q] = as_v(img[l_swz((t5 << 8) | ((unsigned)q << 3) | t3)]);
        run_group<K24T_L2, 9, FULL, MODE>(M, sy, pt, tile, tid, rows, rel_row0, s_lo, s_hi, flags, pending, sm);
        // every thread writes back exactly the dwords it read: no barrier needed in front of these stores
        if constexpr (MODE != 3 && MODE != 5)
#pragma unroll
        for (int q = 0; q < 32; q++) img[l_swz((t5 << 8) | ((unsigned)q << 3) | t3)] = as_u32(M[q]);
    }
    if constexpr (MODE != 3) __syncthreads();
    {
        // thread = position bits 11..4; registers: rho = (q2 << 3) | k3 holds positions (q2 << 12) | (tid << 4) | (k3 << 1) | field
        const unsigned pt = k24t_thread_base(K24T_L3, tile, tid);
        if constexpr (MODE != 3 && MODE != 5)
#pragma unroll
        for (int q2 = 0; q2 < 4; q2++) {
            const unsigned d = l_swz(((unsigned)q2 << 11) | (tid << 3));
            const uint4 lo = *reinterpret_cast<const uint4 *>(&img[d]), hi = *reinterpret_cast<const uint4 *>(&img[d + 4]);
            M[8 * q2 + 0] = as_v(lo.x); M[8 * q2 + 1] = as_v(lo.y); M[8 * q2 + 2] = as_v(lo.z); M[8 * q2 + 3] = as_v(lo.w);
            M[8 * q2 + 4] = as_v(hi.x); M[8 * q2 + 5] = as_v(hi.y); M[8 * q2 + 6] = as_v(hi.z); M[8 * q2 + 7] = as_v(hi.w);
        }
        run_group<K24T_L3, 9, FULL, MODE>(M, sy, pt, tile, tid, rows, rel_row0, s_lo, s_hi, flags, pending, sm);
        if (tile == 0 && tid == 0) k24_report(mirror, pending);
        if constexpr (!FULL) ctl_finish<4>(M, flags, ctl, tile, tid, wmin);
#pragma unroll
        for (int q2 = 0; q2 < 4; q2++) {
            if ((MODE != 2 && MODE != 4 && MODE != 5) || as_u32(M[8 * q2]) == 0x12345u) {
                if constexpr (NT & 2) {
                    u32x4 *dst = reinterpret_cast<u32x4 *>(newm + (pt | ((unsigned)q2 << 12)));
                    const u32x4 v0 = {as_u32(M[8 * q2 + 0]), as_u32(M[8 * q2 + 1]), as_u32(M[8 * q2 + 2]), as_u32(M[8 * q2 + 3])};
                    const u32x4 v1 = {as_u32(M[8 * q2 + 4]), as_u32(M[8 * q2 + 5]), as_u32(M[8 * q2 + 6]), as_u32(M[8 * q2 + 7])};
                    __builtin_nontemporal_store(v0, dst);
                    __builtin_nontemporal_store(v1, dst + 1);
                } else {
                    uint4 *dst = reinterpret_cast<uint4 *>(newm + (pt | ((unsigned)q2 << 12)));
                    dst[0] = make_uint4(as_u32(M[8 * q2 + 0]), as_u32(M[8 * q2 + 1]), as_u32(M[8 * q2 + 2]), as_u32(M[8 * q2 + 3]));
                    dst[1] = make_uint4(as_u32(M[8 * q2 + 4]), as_u32(M[8 * q2 + 5]), as_u32(M[8 * q2 + 6]), as_u32(M[8 * q2 + 7]));
                }
            }
        }
    }
}

template <bool FULL, int MODE, int NT = 2>
__global__ __launch_bounds__(256, 2) void acs_k24t_pass_l_kernel(const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                                                                 int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, int ctl) {
    pass_l_body<FULL, MODE, NT>(oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);
}

}  // namespace k24t

#ifndef VH_JIT_KERNEL
bool k24t_poly_supported(const int *poly) { return poly[0] == k24t::POLY[0] && poly[1] == k24t::POLY[1]; }

// one pass: stages [s_lo, s_hi) of pass `pass` (0 = H: phases 0..8, 1 = L: phases 9..22); rows/syms are those of the pass's
// first phase
template <int MODE, int NT = 2>
static hipError_t launch_pass_mode(int pass, const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                                   int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, hipStream_t stream, int ctl) {
    if (pass == 0) {
        const bool full = s_lo == 0 && s_hi == 9 && (ctl & 255) == 0;
        if (full) hipLaunchKernelGGL((k24t::acs_k24t_pass_h_kernel<true, MODE, NT>), dim3(256), dim3(512), 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);
        else hipLaunchKernelGGL((k24t::acs_k24t_pass_h_kernel<false, MODE, NT>), dim3(256), dim3(512), 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);
    } else if (pass == 1) {
        const bool full = s_lo == 0 && s_hi == 14 && (ctl & 255) == 0;
        if (full) hipLaunchKernelGGL((k24t::acs_k24t_pass_l_kernel<true, MODE, NT>), dim3(512), dim3(256), 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);
        else hipLaunchKernelGGL((k24t::acs_k24t_pass_l_kernel<false, MODE, NT>), dim3(512), dim3(256), 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_k24t_pass(int pass, const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                            int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, hipStream_t stream, bool nt_stores, int ctl) {
#ifdef VHIP_TIMING_BUILD
    // tools/k24t_probe.sh only (make -C csrc timing -> libviterbi_hip_timing.so): the MODE instantiations skip stores or
    // regrouping and give WRONG results by design; none of this is compiled into libviterbi_hip.so
    static const int mode = getenv("VHIP_K24T_MODE") ? atoi(getenv("VHIP_K24T_MODE")) : 0;
    if (mode == 1) return launch_pass_mode<1>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    if (mode == 3) return launch_pass_mode<3>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    if (mode == 2) return launch_pass_mode<2>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    if (mode == 4) return launch_pass_mode<4>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    if (mode == 5) return launch_pass_mode<5>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    // VHIP_K24T_STAGGER="<pass mask>:<pattern>:<sleep units>": see timing_stagger()
    static const char *stg = getenv("VHIP_K24T_STAGGER");
    if (stg) {
        int pm = 0, pat = 0, units = 0;
        if (sscanf(stg, "%d:%d:%d", &pm, &pat, &units) == 3 && ((pm >> pass) & 1)) ctl |= ((units & 255) << 8) | ((pat & 3) << 16);
    }
    static const int nt_env = getenv("VHIP_K24T_NT") ? atoi(getenv("VHIP_K24T_NT")) : -1;  // A/B switch for the non-temporal hints
    if (nt_env == 3) return launch_pass_mode<0, 3>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    if (nt_env >= 0) nt_stores = nt_env != 0;
#endif
    if (!nt_stores) return launch_pass_mode<0, 0>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
    return launch_pass_mode<0>(pass, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, stream, ctl);
}

#endif  // !VH_JIT_KERNEL

}  // namespace vh

#ifdef VH_JIT_KERNEL
#ifndef VH_JIT_NT
#define VH_JIT_NT 2  // metric stores non-temporal (a lone decode); 0 = plain (several decodes share the chip)
#endif
#define VH_JIT_PASS(NAME, BODY, FULL, BOUNDS)                                                                                    \
    extern "C" __global__ BOUNDS void NAME(const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,  \
                                           int rel_row0, int s_lo, int s_hi, int *flags, vh::K24Report mirror, int ctl) {        \
        vh::k24t::BODY<FULL, 0, VH_JIT_NT>(oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror, ctl);                            \
    }
VH_JIT_PASS(vh_jit_k24t_h_full, pass_h_body, true, __launch_bounds__(512))
VH_JIT_PASS(vh_jit_k24t_h_part, pass_h_body, false, __launch_bounds__(512))
VH_JIT_PASS(vh_jit_k24t_l_full, pass_l_body, true, __launch_bounds__(256, 2))
VH_JIT_PASS(vh_jit_k24t_l_part, pass_l_body, false, __launch_bounds__(256, 2))
#endif
