// acs_k24f.hip -- K=24 r=1/2 ACS update, several trellis steps per pass over the 16 MiB metric array.
//
// Replaces update_viterbi224_blk_sse2 (ka9q_libfec_port/viterbi224_sse2.cpp:135-258) for the harness polynomials
// (src/main.cpp:415).  acs_k24.hip moves 32 MiB of metrics through HBM for every trellis step; here the in-place
// rotating trellis (position p holds state rotl^phi(p), phi = row mod 23; the butterfly partner of phase phi
// differs in position bit 22-phi) lets a thread keep 128 positions (64 packed VGPRs) in registers and run every
// phase whose partner bit it holds before writing back:
//      group 0..3 : position bits 22-19, 18-15, 14-11, 10-7   -> 4 steps per pass  (16 strided 16-byte vectors)
//      group 4    : position bits 6-0                          -> 7 steps per pass  (128 contiguous positions)
// i.e. 5 passes per 23 steps: 5 x 32 MiB of metric traffic + 23 MiB of decisions instead of 23 x 33 MiB.
// Passes ping-pong between two buffers (never in place) so that a pass can be replayed: renormalisation
// (viterbi224_sse2.cpp:226-246) is speculative exactly as in acs_k24.hip -- the thread that owns state 0 (position 0
// in every phase) raises flags[PENDING] = row+1 when new[0] >= 25000, later passes return at once, and the host
// replays the raising pass up to that row, renormalises, and continues.
//
// The 4-step passes keep only 64 positions per thread (16 strided 8-byte vectors, 32 packed VGPRs) so that the chip
// holds two waves per SIMD and one wave's loads/stores overlap another's arithmetic (two concurrent decodes ran
// 1.47x faster than two serial ones with 128 positions per thread -- tools/k24_overlap_probe.py).
//
// Decision row r (1 MiB) is stored as the kernels produce it: [thread u of the pass][accumulator words], 32 bits per
// 16 registers (bit k24f_decision_bit(rho, half)).  k24f_locate() in k24f_layout.h maps a position to (word, bit); the decision
// of new state n at row r is at position rotr^((r+1) mod 23)(n).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "k24f_layout.h"
#include "k24t_layout.h"
#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {
namespace k24f {

typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int NB = 23;
constexpr unsigned N = 1u << NB;
constexpr int POLY[2] = {062650457, 062650455};  // src/main.cpp:415

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

// group geometry (k24f_layout.h): first phase, number of phases, lowest vector-index bit, positions per vector
constexpr int group_first_phase(int g) { return g < 4 ? 4 * g : 16; }
constexpr int group_nphases(int g) { return g < 4 ? 4 : 7; }
constexpr int group_bshift(int g) { return k24f_bshift(g); }
constexpr int group_vw(int g) { return k24f_vw(g); }          // 2 or 4 for the strided groups, 8 for group 4
constexpr int group_nr(int g) { return 16 * group_vw(g) / 2; }  // packed registers per thread

// packed ACS, tie -> lower (cmpgt_epi16 then min_epi16, viterbi224_sse2.cpp:190-194); acc gets the decision bits
__device__ __forceinline__ i16x2 acs(i16x2 lower, i16x2 upper, i16x2 &w) {
    w = __builtin_elementwise_sub_sat(upper, lower);  // < 0  <=>  upper < lower  <=>  decision 1
    return __builtin_elementwise_min(lower, upper);
}
// Decision bits of the register pair (RE, RE+1), RE even: one v_perm_b32 gathers the four sign bytes
// [RE.low, RE.high, RE+1.low, RE+1.high]; a shift and a masked OR drop the four sign bits into the word of the
// 16-register group at bit k24f_decision_bit(rho, half) (k24f_layout.h).
template <int RE, int NA>
__device__ __forceinline__ void put_signs(i16x2 we, i16x2 wo, unsigned (&acc)[NA]) {
    static_assert((RE & 1) == 0, "pairs start at an even register");
    constexpr int i8 = (RE & 15) >> 1;
    const unsigned P = __builtin_amdgcn_perm(as_u32(wo), as_u32(we), 0x07050301u);
    acc[RE >> 4] = ((P >> i8) & (0x80808080u >> i8)) | acc[RE >> 4];
}

// One trellis step at phase PHI.  pt = the thread-id part of the position (vector-index and low bits zero).
// Register rho = (v << KB) | k holds vector v, elements 2k (low field) and 2k+1 (high field); KB = log2(VW) - 1.
template <int G, int PHI>
__device__ __forceinline__ void stage(i16x2 (&M)[group_nr(G)], unsigned s0, unsigned s1, unsigned pt, unsigned (&acc)[group_nr(G) / 16]) {
    constexpr int b = NB - 1 - PHI;
    constexpr int BS = group_bshift(G), NR = group_nr(G), KB = k24f_lw(G) - 1;
    // static position of register rho, low field
    auto spos = [](int rho) constexpr -> unsigned {
        return ((unsigned)(rho >> KB) << BS) | ((unsigned)(rho & ((1 << KB) - 1)) << 1);
    };
    // class offset of the thread part, folded into the symbols
    const unsigned jt = PHI == 0 ? pt : (((pt << PHI) | (pt >> (NB - PHI))) & (N - 1u));
    const unsigned c0 = __popc((2u * jt) & (unsigned)POLY[0]) & 1u, c1 = __popc((2u * jt) & (unsigned)POLY[1]) & 1u;
    const unsigned a0 = s0 ^ (c0 ? 255u : 0u), a1 = s1 ^ (c1 ? 255u : 0u);
    const unsigned x0 = a0 ^ 255u, x1 = a1 ^ 255u;
    const unsigned T[4] = {a0 + a1, x0 + a1, a0 + x1, x0 + x1};  // xor + add   viterbi224_sse2.cpp:159
    constexpr unsigned COMP = (unsigned)Code224::bm_comp;
#pragma unroll
    for (int i = 0; i < NR / 16; i++) acc[i] = 0;

    if constexpr (b >= 1) {
        // register stage: rho bits 0..KB-1 <-> position bits 1..KB, rho bits KB..KB+3 <-> position bits BS..BS+3
        constexpr int rb = (b >= BS) ? (KB + b - BS) : (b - 1);
        static_assert(rb >= 0 && rb < KB + 4 && (b >= BS || b <= KB), "phase outside this group");
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
            constexpr unsigned cr = cls(rotl23(spos(r0), PHI));
            const i16x2 A = M[r0], B = M[r1];
            const i16x2 tp = as_v(TP[cr]), tq = as_v(TQ[cr]);
            const i16x2 m0 = __builtin_elementwise_add_sat(A, tp), m1 = __builtin_elementwise_add_sat(B, tq);  // adds_epi16 :163-166
            const i16x2 m2 = __builtin_elementwise_add_sat(A, tq), m3 = __builtin_elementwise_add_sat(B, tp);
            M[r0] = acs(m0, m1, W0);
            M[r1] = acs(m2, m3, W1);
        };
        if constexpr (rb == 0) {
            // the butterfly pair (2i, 2i+1) is also the sign-gathering pair
            sfor<NR / 2>([&](auto I) {
                constexpr int r0 = 2 * decltype(I)::value;
                i16x2 w0, w1;
                pair(std::integral_constant<int, r0>{}, w0, w1);
                put_signs<r0>(w0, w1, acc);
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
                put_signs<r0>(wa0, wb0, acc);
                put_signs<r1>(wa1, wb1, acc);
            });
        }
    } else {
        // half stage (position bit 0): old[j] low field, old[j+H] high field
        auto half = [&](auto R0, i16x2 &W) {
            constexpr int r0 = decltype(R0)::value;
            constexpr unsigned cr = cls(rotl23(spos(r0), PHI));
            const unsigned t = T[cr], tc = COMP - t;
            const i16x2 A = M[r0];
            // broadcasting one field to both lanes of a packed add is an op_sel modifier, not an instruction
            const i16x2 Alo = {A.x, A.x}, Ahi = {A.y, A.y};
            const i16x2 lower = __builtin_elementwise_add_sat(Alo, as_v(t | (tc << 16)));  // (m0, m2) = old[j] + (t, t')
            const i16x2 upper = __builtin_elementwise_add_sat(Ahi, as_v(tc | (t << 16)));  // (m1, m3) = old[j+H] + (t', t)
            M[r0] = acs(lower, upper, W);
        };
        sfor<NR / 2>([&](auto I) {
            constexpr int r0 = 2 * decltype(I)::value;
            i16x2 w0, w1;
            half(std::integral_constant<int, r0>{}, w0);
            half(std::integral_constant<int, r0 + 1>{}, w1);
            put_signs<r0>(w0, w1, acc);
        });
    }
}

// FULL: every stage of the group runs (s_lo == 0, s_hi == NP) -- the steady state.  The stage branches then vanish and
// hipcc can see that nothing but row stores is outstanding after the first stage, so no later s_waitcnt drains them.
template <int G, bool FULL>
__global__ __launch_bounds__(256) void acs_k24f_pass_kernel(const int16_t *oldm, int16_t *__restrict__ newm,
                                                            unsigned char *__restrict__ rows, const unsigned char *syms,
                                                            int rel_row0, int s_lo, int s_hi, int *__restrict__ flags,
                                                            K24Report mirror) {
    // rows/syms point at the first row of THIS pass's phase 0 of the group; rel_row0 = that row's index in the call.
    // Stages s in [s_lo, s_hi) of the group run; a pending renormalisation of an earlier row cancels the pass.
    int pending = flags[K24F_PENDING];  // looked at below, behind the metric loads
    constexpr int BS = group_bshift(G), P0 = group_first_phase(G), NP = group_nphases(G);
    constexpr int NR = group_nr(G), WPT = NR / 16;
    const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;  // N / (16*VW) threads
    const unsigned pt = k24f_thread_base(G, u);                // thread part of the position
    i16x2 M[NR];
    // Every symbol of the pass is fetched here, next to the metric loads: a fetch at the top of each stage would put a
    // full memory round trip (and, through vmcnt(0), the acknowledgement of the previous row store) on the critical
    // path of every trellis step.  Indices are clamped to the stages that run, so a truncated pass reads nothing else.
    unsigned sy[NP];
#pragma unroll
    for (int S = 0; S < NP; S++) {
        const int sc = FULL ? S : min(max(S, s_lo), s_hi - 1);
        sy[S] = (unsigned)syms[2 * sc] | ((unsigned)syms[2 * sc + 1] << 8);
    }
    // Group 4 holds 16 CONTIGUOUS vectors per thread (256 B), so a direct load would touch 64 separate 256-byte
    // chunks per wave instruction.  Instead each wave streams its 16 KiB tile in lane-linear order (1 KiB per
    // instruction) and transposes it through a wave-private, XOR-swizzled LDS tile (conflict-free both ways).
    __shared__ uint4 tile[G == 4 ? 4 * 1024 : 1];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if constexpr (G == 4) {
        const uint4 *src = reinterpret_cast<const uint4 *>(oldm) + (size_t)(u - lane) * 16;  // wave tile: 1024 vectors
        uint4 *tw = tile + wv * 1024;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned gv = i * 64 + lane;  // vector index inside the tile: thread gv>>4, slot gv&15
            tw[(gv & ~15u) | ((gv ^ (gv >> 4)) & 15u)] = src[gv];
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const uint4 q = tw[lane * 16 + ((v ^ lane) & 15u)];
            M[4 * v] = as_v(q.x);
            M[4 * v + 1] = as_v(q.y);
            M[4 * v + 2] = as_v(q.z);
            M[4 * v + 3] = as_v(q.w);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int v = 0; v < 16; v++) {
            if constexpr (group_vw(G) == 4) {
                const uint2 q = *reinterpret_cast<const uint2 *>(oldm + (pt | ((unsigned)v << BS)));
                M[2 * v] = as_v(q.x);
                M[2 * v + 1] = as_v(q.y);
            } else {
                M[v] = as_v(*reinterpret_cast<const unsigned *>(oldm + (pt | ((unsigned)v << BS))));
            }
        }
    }
    // The flag is a dependent scalar load: looking at it only here keeps its latency under the metric loads.  The empty
    // asm pins that order (hipcc otherwise sinks the loads below the branch, or hoists the compare above them), which
    // is also why oldm and syms are not __restrict__.
    asm volatile("" : "+s"(pending) : : "memory");
    if (pending != 0 && pending - 1 < rel_row0 + s_lo) {
        if (blockIdx.x == 0 && threadIdx.x == 0) k24_report(mirror, pending);
        return;
    }
    sfor<NP>([&](auto I) {
        constexpr int S = decltype(I)::value;
        constexpr int PHI = P0 + S;
        if (FULL || (S >= s_lo && S < s_hi)) {  // grid-uniform
            const unsigned sy0 = sy[S] & 255u, sy1 = sy[S] >> 8;
            unsigned acc[WPT];
            stage<G, PHI>(M, sy0, sy1, pt, acc);
            unsigned *row = reinterpret_cast<unsigned *>(rows + (size_t)S * (N / 8)) + (size_t)u * WPT;
            // decision words are written once and read by the chainback much later: keep them out of the caches the
            // 32 MiB of metrics live in
#pragma unroll
            for (int w = 0; w < WPT; w++) __builtin_nontemporal_store(acc[w], row + w);
            if (u == 0) {  // state 0 is position 0 in every phase: thread 0, register 0, low field
                const int new0 = (int)(short)(as_u32(M[0]) & 0xffffu);
                if (new0 >= Code224::renorm_thr && pending == 0) flags[K24F_PENDING] = pending = rel_row0 + S + 1;
            }
        }
    });
    // the last pass of a host batch also reports the flag to pinned host memory (thread 0 is the only writer of the flag)
    if (u == 0) k24_report(mirror, pending);
    if constexpr (G == 4) {
        uint4 *tw = tile + wv * 1024;
#pragma unroll
        for (int v = 0; v < 16; v++)
            tw[lane * 16 + ((v ^ lane) & 15u)] = make_uint4(as_u32(M[4 * v]), as_u32(M[4 * v + 1]), as_u32(M[4 * v + 2]), as_u32(M[4 * v + 3]));
        __syncthreads();
        uint4 *dst = reinterpret_cast<uint4 *>(newm) + (size_t)(u - lane) * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned gv = i * 64 + lane;
            // non-temporal: the dirty lines of a plain store would be written back once more when the kernel ends (acs_k24t.hip)
            const uint4 t4 = tw[(gv & ~15u) | ((gv ^ (gv >> 4)) & 15u)];
            const u32x4 v4 = {t4.x, t4.y, t4.z, t4.w};
            __builtin_nontemporal_store(v4, reinterpret_cast<u32x4 *>(dst) + gv);
        }
    } else {
#pragma unroll
        for (int v = 0; v < 16; v++) {
            if constexpr (group_vw(G) == 4) {
                const u32x2 v2 = {as_u32(M[2 * v]), as_u32(M[2 * v + 1])};
                __builtin_nontemporal_store(v2, reinterpret_cast<u32x2 *>(newm + (pt | ((unsigned)v << BS))));
            } else {
                __builtin_nontemporal_store(as_u32(M[v]), reinterpret_cast<unsigned *>(newm + (pt | ((unsigned)v << BS))));
            }
        }
    }
}

}  // namespace k24f

bool k24f_poly_supported(const int *poly) { return poly[0] == k24f::POLY[0] && poly[1] == k24f::POLY[1]; }

// one pass: stages [s_lo, s_hi) of group g; rows/syms are those of the group's first phase
hipError_t launch_k24f_pass(int g, const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                            int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, hipStream_t stream) {
    const dim3 block(256);
    const dim3 grid((k24f::N / (16u * (unsigned)k24f_vw(g))) / 256u);  // 16 vectors of VW positions per thread
    const bool full = s_lo == 0 && s_hi == (g < 4 ? 4 : 7);
    switch (g) {
    case 0:
        if (full) hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<0, true>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        else hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<0, false>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        break;
    case 1:
        if (full) hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<1, true>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        else hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<1, false>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        break;
    case 2:
        if (full) hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<2, true>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        else hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<2, false>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        break;
    case 3:
        if (full) hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<3, true>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        else hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<3, false>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        break;
    case 4:
        if (full) hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<4, true>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        else hipLaunchKernelGGL((k24f::acs_k24f_pass_kernel<4, false>), grid, block, 0, stream, oldm, newm, rows, syms, rel_row0, s_lo, s_hi, flags, mirror);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vh
