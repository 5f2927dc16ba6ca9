// acs_wave.hip -- the LATENCY geometry for K=7 and K=9: whole wavefronts per frame, one trellis state per lane.
//
// The reference's own methodology decodes ONE frame per handle, one blocking call at a time (src/main.cpp:257-280), and
// update_viterbi27_blk_sse2 / update_spiral47 (ka9q_libfec_port/viterbi27_sse2.cpp:119-175, spiral/spiral47.cpp:131-538) walk its
// trellis steps one after the other.  A frame is a single chain of dependent steps, so what bounds it on a GPU is the number of
// instructions ONE wave has to issue per step (a lone wave issues one instruction per ~4-5 cycles whatever its kind): the
// throughput kernels of acs_regs.hip spend ~70 instructions per step on a frame held by four lanes (0.18 us per step).  Here
// the 64 lanes of a wave hold 64 path metrics and a K=7 step is ~14 instructions:
//   * the rotating in-place trellis of acs_regs.hip: position p (= lane p) holds state rotl^phi(p) before the step of phase
//     phi = row mod 6, the butterfly partner is lane p ^ (1 << (5 - phi)) -- one DPP operand (quad_perm), two bank-masked DPP
//     row shifts, or one v_permlane16/32_swap (gfx950) -- and no metric ever moves;
//   * the branch metrics do not depend on the path metrics, so the OTHER three waves of the workgroup compute them ahead, for
//     every class of the branch table, into an LDS table (double-buffered chunks): on the serial chain a step's {t, t'} is one
//     ds_read_b64 at a per-lane class offset;
//   * ka9q: metrics in the top byte of a dword -- a 32-bit add wraps like _mm_add_epi8 and the sign of a 32-bit difference is that of
//     the 8-bit one; spiral: plain integers with a lazy offset, so that "subtract the minimum" after (nearly) every step is a scalar
//     assignment and _mm_adds_epu8's saturation a minimum with a scalar cap (struct Lazy below);
//   * the 64 decisions of a step are one ballot; 48 of them are parked in two VGPRs (v_writelane) and leave as one store.
// Three kernels: acs_wave_kernel (K=7, one wave per frame), acs_wave9_kernel (ka9q K=9: four waves per frame, the two wave-bit
// phases through an LDS exchange), acs_wave9s_kernel (spiral K=9: one wave, four states per lane, so that the minimum the spiral
// decoders subtract after every step needs no barrier).
// Decisions: [frame][row][N/64] 64-bit words in POSITION order (bit p & 63 of word p >> 6 of row r = decision of the new state
// rotl^((r+1) mod (K-1))(p)): exactly N/8 bytes per frame-row like every other layout.  Any polynomial set works (classes are
// computed per lane at run time).  Arithmetic per family exactly as acs_lds.hip.  chainback:
// chainback_wave_kernel below (the 64 lanes share one frame's walk).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {
namespace wave7 {

constexpr int K = 7, NB = 6;
constexpr unsigned N = 64, H = 32;
static_assert(NB == K - 1 && N == 1u << NB && H == N / 2, "one state per lane");
constexpr int BLK = 48;        // steps per unrolled block = 8 periods; their rows leave as one 384-byte store
constexpr int THREADS = 256;   // wave 0 walks the trellis, waves 1..3 fill the branch-metric table of the next chunk
constexpr int TBL_BYTES = 30720;  // per buffer: CH steps x 2^R classes x 8 bytes, CH a multiple of BLK

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int NN, class F>
__device__ __forceinline__ void sfor(F &&f) {
    sfor_impl(f, std::make_integer_sequence<int, NN>{});
}

__host__ __device__ constexpr unsigned rotl6(unsigned x, int s) {
    s %= NB;
    return s == 0 ? x : (((x << s) | (x >> (NB - s))) & (N - 1u));
}

// value of lane (l ^ (1 << B))
template <int B>
__device__ __forceinline__ unsigned partner(unsigned x, unsigned lane) {
    if constexpr (B == 0) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    } else if constexpr (B == 1) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    } else if constexpr (B == 2) {
        // banks (4 lanes) 0 and 2 of every row of 16 read four lanes up, banks 1 and 3 four lanes down
        int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0x5, false);    // row_shl:4
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);  // row_shr:4
    } else if constexpr (B == 3) {
        int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x108, 0xf, 0x3, false);    // row_shl:8
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x118, 0xf, 0xc, false);  // row_shr:8
    } else if constexpr (B == 4) {
        // v_permlane16_swap: rows 1,3 of the first operand <-> rows 0,2 of the second.  Both = x: the first comes back as
        // [x0 x0 x2 x2], the second as [x1 x1 x3 x3] (xr = row r of x): odd rows take the first, even rows the second
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16u) ? r[0] : r[1];
    } else {
        // v_permlane32_swap: upper half of the first operand <-> lower half of the second
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32u) ? r[0] : r[1];
    }
}

// lane LANE of acc <- a wave-uniform value.  clang has no builtin for v_writelane_b32, so the LLVM intrinsic is declared
// directly -- NOT inline assembly: an SGPR written by a VALU instruction (the ballot's v_cmp) needs wait states before
// v_writelane may read it, and the compiler's hazard recogniser does not look into inline assembly (a first version did
// that and parked stale ballots).
extern "C" __device__ int vh_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
template <int LANE>
__device__ __forceinline__ unsigned writelane(unsigned acc, unsigned sval) {
    return (unsigned)vh_writelane_i32((int)sval, LANE, (int)acc);
}

// Metric representation.  ka9q (u8 modular): the metric in the TOP byte of a dword -- a 32-bit add wraps like _mm_add_epi8 and the
// sign of a 32-bit difference is that of the 8-bit one.  spiral (u8 saturating, minimum subtracted after nearly every step):
// plain integers with a LAZY offset -- the register holds true + off, the saturation of _mm_adds_epu8 at 255 becomes a minimum with
// cap = 255 + off, and "subtract the minimum" is off = min(registers): a scalar assignment instead of a vector subtract, and the
// wave reduction it needs leaves the serial chain of the next step's adds (only its final minimum waits for the cap).  Exact:
// the register is true + off after every step, the fire test and the capped comparisons are the reference's shifted by off.
template <class C>
constexpr int metric_shift() { return C::metric == U8MOD ? 24 : 0; }
struct Lazy {
    unsigned off = 0, cap = 255;
};
template <class C>
__device__ __forceinline__ unsigned to_lane_metric(int m) {
    return ((unsigned)m & 255u) << metric_shift<C>();
}
template <class C>
__device__ __forceinline__ int from_lane_metric(unsigned M, const Lazy &lz) {
    if constexpr (C::metric == U8MOD) return (int)(M >> 24);
    else return (int)(M - lz.off);
}

// Add-compare-select of one lane.  M: this lane's metric, X: its butterfly partner's, tv / tcv: the butterfly's branch metrics
// t and t' in the top byte, up: this lane holds old[j + H].  Returns the decision of the new state that now lives in this lane.
template <class C>
__device__ __forceinline__ bool acs(unsigned &M, unsigned X, unsigned tv, unsigned tcv, bool up, const Lazy &lz) {
    if constexpr (C::metric == U8MOD) {
        // lower position: m0 = old[j] + t (self), m1 = old[j+H] + t' (partner); upper position: m2 = old[j] + t' (partner),
        // m3 = old[j+H] + t (self)                                                       viterbi27_sse2.cpp:149-152
        const unsigned cs = M + tv, co = X + tcv;
        const unsigned lo = up ? co : cs, hi = up ? cs : co;
        const int diff = (int)(lo - hi);
        const bool d = diff > 0;            // cmpgt_epi8(sub_epi8(m0, m1), 0): tie -> lower      :155-156
        // d ? hi : lo without waiting for the compare's lane mask: lo - max(diff, 0) (diff = INT_MIN: d = 0, lo stays)  :157-158
        M = lo - (unsigned)max(diff, 0);
        return d;
    } else {
        // adds_epu8 (spiral47.cpp:220-223) = min(x + t, 255) on the true values = min(register + t, cap) here; the upper
        // predecessor's candidate is the partner's for a lower position, this lane's own for an upper one
        const unsigned cs = M + tv, co = X + tcv;
        const unsigned nw = min(min(cs, co), lz.cap);   // min_epu8 of the two saturated candidates          :224-225
        const unsigned upper = min(up ? cs : co, lz.cap);
        const bool d = nw == upper;                     // cmpeq(min, upper): tie -> upper                   :226-227
        M = nw;
        return d;
    }
}
// One trellis step of the K=7 kernel at phase PHI: the partner is lane ^ (1 << (5 - PHI)).
template <class C, int PHI>
__device__ __forceinline__ bool step(unsigned &M, unsigned tv, unsigned tcv, bool up, unsigned lane, const Lazy &lz) {
    return acs<C>(M, partner<NB - 1 - PHI>(M, lane), tv, tcv, up, lz);
}

// spiral47.cpp:313-331: after every step, if new[0] > threshold, subtract the minimum over all states (saturating; nothing
// is below the minimum).  State 0 is position 0 in every phase: lane 0.
// It fires on most steps of a noisy frame, so the reduction is kept short: four v_min_u32 with a DPP operand leave each row of
// 16 lanes with its minimum, the four rows meet in scalar registers (v_readlane + s_min), and the subtrahend is an SGPR.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define VH_MIN_DPP(ctrl) v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, true))
    VH_MIN_DPP(0xB1);   // quad_perm [1,0,3,2]
    VH_MIN_DPP(0x4E);   // quad_perm [2,3,0,1]
    VH_MIN_DPP(0x141);  // row_half_mirror
    VH_MIN_DPP(0x140);  // row_mirror
#undef VH_MIN_DPP
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
// spiral47.cpp:313-331: after every step, if new[0] > threshold, subtract the minimum over all states.  State 0 is position 0 in
// every phase: lane 0.  With the lazy offset: the test is shifted by off, and the subtraction is off = minimum.
template <class C>
__device__ __forceinline__ void renormalise(unsigned &M, Lazy &lz) {
    if constexpr (C::renorm) {
        const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)M);
        if (m0 > (unsigned)C::renorm_thr + lz.off) {
            lz.off = wave_min_u32(M);
            lz.cap = lz.off + 255u;
        }
    }
}
// off grows by at most 63 per step; once per block of steps it is folded back into the registers long before it can wrap (at
// 2^20 -- every ~17 000 steps or more -- rather than near 2^32, so that frames of ordinary length take this path too and the
// tests reach it: tests/test_full_size.py::test_one_very_long_frame)
template <class C, int NM>
__device__ __forceinline__ void fold_offset(unsigned (&M)[NM], Lazy &lz) {
    if constexpr (C::renorm) {
        if (lz.off > (1u << 20)) {
#pragma unroll
            for (int r = 0; r < NM; r++) M[r] -= lz.off;
            lz.off = 0;
            lz.cap = 255u;
        }
    }
}

struct Args {
    const unsigned char *syms;
    size_t sym_stride;
    int nsteps, row0, cap_rows, nframes;
    unsigned long long *dec;  // [nframes][cap_rows] position-ordered rows
    int16_t *metrics;         // [nframes][64] canonical path metrics (natural units)
    int poly[8];
};

template <class C>
__global__ __launch_bounds__(THREADS) void acs_wave_kernel(Args a) {
    constexpr int R = C::R, NC = 1 << R;
    constexpr int STEP_BYTES = NC * 8;
    constexpr int CH = (TBL_BYTES / STEP_BYTES / BLK) * BLK;  // table steps per chunk (960 for r=1/2, 240 for r=1/4)
    static_assert(CH >= BLK && CH * STEP_BYTES <= TBL_BYTES, "chunk geometry");
    __shared__ __attribute__((aligned(16))) unsigned char tbl[2][TBL_BYTES];

    const int f = blockIdx.x;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    // the wave index as a SCALAR: "wave == 0" must be a uniform branch for the compiler, or every scalar the worker wave carries from
    // step to step (the lazy offset of the spiral arithmetic) becomes a per-lane value under a divergent branch
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned char *syms = a.syms + (size_t)f * a.sym_stride;
    unsigned long long *rows = a.dec + (size_t)f * a.cap_rows + a.row0;
    const int phi0 = a.row0 % NB;
    const int pre = min(a.nsteps, (NB - phi0) % NB);               // steps up to the first phase-0 row: slow path
    const int nmain = ((a.nsteps - pre) / BLK) * BLK;              // whole blocks through the table
    const int nchunks = (nmain + CH - 1) / CH;

    // table of chunk c: for each of its steps and every branch-table class {t << 24, t' << 24}
    auto fill = [&](int buf, int c) {
        const int s0 = pre + c * CH, cnt = min(CH, nmain - c * CH);
        for (int i = (int)tid - 64; i < cnt; i += THREADS - 64) {
            int s[R];
#pragma unroll
            for (int r = 0; r < R; r++) s[r] = syms[(size_t)(s0 + i) * R + r];
            uint2 *e = reinterpret_cast<uint2 *>(&tbl[buf][i * STEP_BYTES]);
#pragma unroll
            for (int cl = 0; cl < NC; cl++) {
                const int t = C::bm(s, (unsigned)cl);
                e[cl] = make_uint2((unsigned)t << metric_shift<C>(), (unsigned)C::bm_tc(t) << metric_shift<C>());
            }
        }
    };

    // ---- wave 0: the serial chain
    unsigned M = 0;
    Lazy lz;
    unsigned cls[NB];     // class of this lane's butterfly at each phase
    bool up[NB];          // this lane is the upper predecessor at each phase
    unsigned aoff[NB];    // byte offset of the class entry inside a step's record
    if (wave == 0) {
#pragma unroll
        for (int p = 0; p < NB; p++) {
            const unsigned st = rotl6(lane, p);                       // old state held at phase p
            cls[p] = bt_class<R>(st & (H - 1u), a.poly);
            up[p] = (st >> (NB - 1)) & 1u;
            aoff[p] = cls[p] * 8u;
        }
        M = to_lane_metric<C>(a.metrics[(size_t)f * N + rotl6(lane, phi0)]);
    }
    // one step outside the table (the few steps in front of the first phase-0 row and behind the last whole block)
    auto slow_step = [&](int i) {
        int s[R];
#pragma unroll
        for (int r = 0; r < R; r++) s[r] = syms[(size_t)i * R + r];
        const int phi = (a.row0 + i) % NB;
        bool d = false;
        sfor<NB>([&](auto P) {
            constexpr int PHI = decltype(P)::value;
            if (phi == PHI) {  // uniform
                const int t = C::bm(s, cls[PHI]);
                d = step<C, PHI>(M, (unsigned)t << metric_shift<C>(), (unsigned)C::bm_tc(t) << metric_shift<C>(), up[PHI], lane, lz);
            }
        });
        renormalise<C>(M, lz);
        const unsigned long long row = __builtin_amdgcn_ballot_w64(d);
        if (lane == 0) rows[i] = row;
    };

    if (wave == 0) {
        for (int i = 0; i < pre; i++) slow_step(i);
    } else if (nchunks > 0) {
        fill(0, 0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        if (wave == 0) {
            const int cnt = min(CH, nmain - c * CH);
            const unsigned char *tb = tbl[c & 1];
            for (int b0 = 0; b0 < cnt; b0 += BLK) {
                unsigned acc_lo = 0, acc_hi = 0;
                const unsigned char *blk = tb + b0 * STEP_BYTES;
                {
                    unsigned m1[1] = {M};
                    fold_offset<C, 1>(m1, lz);
                    M = m1[0];
                }
                // the table entries of a whole period are fetched one period ahead: an LDS read issued a step ahead is not back in
                // time for a ~90-cycle step
                uint2 en[NB];
#pragma unroll
                for (int q = 0; q < NB; q++) en[q] = *reinterpret_cast<const uint2 *>(blk + aoff[q] + q * STEP_BYTES);
                sfor<BLK>([&](auto J) {
                    constexpr int j = decltype(J)::value, PHI = j % NB;
                    const uint2 e = en[PHI];
                    if constexpr (j + NB < BLK) en[PHI] = *reinterpret_cast<const uint2 *>(blk + aoff[PHI] + (j + NB) * STEP_BYTES);
                    const bool d = step<C, PHI>(M, e.x, e.y, up[PHI], lane, lz);
                    renormalise<C>(M, lz);
                    const unsigned long long row = __builtin_amdgcn_ballot_w64(d);
                    acc_lo = writelane<j>(acc_lo, (unsigned)row);
                    acc_hi = writelane<j>(acc_hi, (unsigned)(row >> 32));
                });
                if (lane < (unsigned)BLK) rows[pre + c * CH + b0 + (int)lane] = ((unsigned long long)acc_hi << 32) | acc_lo;
            }
        } else if (c + 1 < nchunks) {
            fill((c + 1) & 1, c + 1);
        }
        __syncthreads();
    }
    if (wave == 0) {
        for (int i = pre + nmain; i < a.nsteps; i++) slow_step(i);
        // position p now holds state rotl^((row0 + nsteps) mod 6)(p)
        a.metrics[(size_t)f * N + rotl6(lane, (a.row0 + a.nsteps) % NB)] = (int16_t)from_lane_metric<C>(M, lz);
    }
}

// ---------------------------------------------------------------------------------------------------- K = 9: four waves per frame
// 256 states = 4 waves x 64 lanes, position p = wave * 64 + lane in the same rotating layout (period 8).  Phases 2..7 pair the six
// lane bits: exactly the K=7 step above, per wave, no LDS, no barrier.  Phases 0 and 1 pair the two wave bits: the partner is
// the same lane of wave ^ 2 / wave ^ 1, fetched through a 1 KiB LDS exchange (one barrier; two buffers alternate so that nobody
// waits twice).  All four waves work, so they fill the branch-metric table of a chunk themselves (1-5 % of a chunk's
// instructions).  spiral49 (spiral49.cpp:790, 1490: after EVERY step, if new[0] > 103 subtract the minimum over all 256 states):
// each wave forms its minimum, wave 0 adds the verdict on state 0, and one barrier per step carries both through LDS.
// Decisions: [frame][row][wave] 64-bit ballots = 32 bytes per row, bit (p & 63) of word (p >> 6).
namespace k9 {
constexpr int K = 9, NB = 8;
constexpr unsigned N = 256, H = 128;
static_assert(NB == K - 1 && N == 1u << NB && H == N / 2, "four states per lane");
__host__ __device__ constexpr unsigned rotl8(unsigned x, int s) {
    s %= NB;
    return s == 0 ? x : (((x << s) | (x >> (NB - s))) & (N - 1u));
}
}  // namespace k9

template <class C>
__global__ __launch_bounds__(256) void acs_wave9_kernel(Args a) {
    using namespace k9;
    static_assert(C::metric == U8MOD && !C::renorm, "ka9q arithmetic: the spiral codes take acs_wave9s_kernel (their per-step minimum would need a barrier per step here)");
    constexpr int R = C::R, NC = 1 << R;
    constexpr int STEP_BYTES = NC * 8;
    constexpr int CH = (TBL_BYTES / STEP_BYTES / BLK) * BLK;
    static_assert(BLK % k9::NB == 0 && CH >= BLK, "a block is whole periods");
    __shared__ __attribute__((aligned(16))) unsigned char tbl[TBL_BYTES];
    __shared__ unsigned xchg[2][4][64];   // wave-bit phases: every wave's metrics, buffer = phase

    const int f = blockIdx.x;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    // the wave index as a SCALAR: "wave == 0" must be a uniform branch for the compiler, or every scalar the worker wave carries from
    // step to step (the lazy offset of the spiral arithmetic) becomes a per-lane value under a divergent branch
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned char *syms = a.syms + (size_t)f * a.sym_stride;
    unsigned long long *rows = a.dec + ((size_t)f * a.cap_rows + a.row0) * 4 + wave;  // this wave's column; row stride 4 words
    const int phi0 = a.row0 % k9::NB;
    const int pre = min(a.nsteps, (k9::NB - phi0) % k9::NB);
    const int nmain = ((a.nsteps - pre) / BLK) * BLK;
    const int nchunks = (nmain + CH - 1) / CH;

    unsigned cls[k9::NB], aoff[k9::NB];
    bool up[k9::NB];
#pragma unroll
    for (int p = 0; p < k9::NB; p++) {
        const unsigned st = rotl8(tid, p);  // old state held at phase p
        cls[p] = bt_class<R>(st & (k9::H - 1u), a.poly);
        up[p] = (st >> (k9::NB - 1)) & 1u;
        aoff[p] = cls[p] * 8u;
    }
    unsigned M = to_lane_metric<C>(a.metrics[(size_t)f * k9::N + rotl8(tid, phi0)]);
    const Lazy lz;  // unused by the modular arithmetic

    // partner of this lane at phase PHI (uniform): lane bits through the K=7 partner fetch, wave bits through LDS
    auto fetch = [&](auto P) -> unsigned {
        constexpr int PHI = decltype(P)::value;
        if constexpr (PHI >= 2) {
            return partner<k9::NB - 1 - PHI>(M, lane);
        } else {
            xchg[PHI][wave][lane] = M;
            __syncthreads();
            return xchg[PHI][wave ^ (PHI == 0 ? 2u : 1u)][lane];
        }
    };
    auto slow_step = [&](int i) {
        int s[R];
#pragma unroll
        for (int r = 0; r < R; r++) s[r] = syms[(size_t)i * R + r];
        const int phi = (a.row0 + i) % k9::NB;
        bool d = false;
        sfor<k9::NB>([&](auto P) {
            constexpr int PHI = decltype(P)::value;
            if (phi == PHI) {  // uniform
                const int t = C::bm(s, cls[PHI]);
                const unsigned X = fetch(P);
                d = acs<C>(M, X, (unsigned)t << 24, (unsigned)C::bm_tc(t) << 24, up[PHI], lz);
            }
        });
        const unsigned long long row = __builtin_amdgcn_ballot_w64(d);
        if (lane == 0) rows[(size_t)i * 4] = row;
    };

    for (int i = 0; i < pre; i++) slow_step(i);
    for (int c = 0; c < nchunks; c++) {
        const int s0 = pre + c * CH, cnt = min(CH, nmain - c * CH);
        __syncthreads();  // everybody is done with the previous chunk's table
        for (int i = (int)tid; i < cnt; i += 256) {
            int s[R];
#pragma unroll
            for (int r = 0; r < R; r++) s[r] = syms[(size_t)(s0 + i) * R + r];
            uint2 *e = reinterpret_cast<uint2 *>(&tbl[i * STEP_BYTES]);
#pragma unroll
            for (int cl = 0; cl < NC; cl++) {
                const int t = C::bm(s, (unsigned)cl);
                e[cl] = make_uint2((unsigned)t << 24, (unsigned)C::bm_tc(t) << 24);
            }
        }
        __syncthreads();
        for (int b0 = 0; b0 < cnt; b0 += BLK) {
            unsigned acc_lo = 0, acc_hi = 0;
            const unsigned char *blk = tbl + b0 * STEP_BYTES;
            uint2 en[k9::NB];  // table entries fetched one period ahead (see acs_wave_kernel)
#pragma unroll
            for (int q = 0; q < k9::NB; q++) en[q] = *reinterpret_cast<const uint2 *>(blk + aoff[q] + q * STEP_BYTES);
            sfor<BLK>([&](auto J) {
                constexpr int j = decltype(J)::value, PHI = j % k9::NB;
                const uint2 e = en[PHI];
                if constexpr (j + k9::NB < BLK) en[PHI] = *reinterpret_cast<const uint2 *>(blk + aoff[PHI] + (j + k9::NB) * STEP_BYTES);
                const unsigned X = fetch(std::integral_constant<int, PHI>{});
                const bool d = acs<C>(M, X, e.x, e.y, up[PHI], lz);
                const unsigned long long row = __builtin_amdgcn_ballot_w64(d);
                acc_lo = writelane<j>(acc_lo, (unsigned)row);
                acc_hi = writelane<j>(acc_hi, (unsigned)(row >> 32));
            });
            if (lane < (unsigned)BLK) rows[(size_t)(s0 + b0 + (int)lane) * 4] = ((unsigned long long)acc_hi << 32) | acc_lo;
        }
    }
    for (int i = pre + nmain; i < a.nsteps; i++) slow_step(i);
    a.metrics[(size_t)f * k9::N + rotl8(tid, (a.row0 + a.nsteps) % k9::NB)] = (int16_t)(M >> 24);
}

// ---------------------------------------------------------------------------------------------------- K = 9, spiral: one wave, four states per lane
// The spiral decoders subtract the minimum over ALL states after every step (spiral49.cpp:790): with four waves per frame that is a
// barrier round trip per step, more than half of the step.  Here one wave holds all 256 metrics, register r of lane l = position
// r * 64 + l -- the same position <-> (word, bit) map as acs_wave9_kernel, so the decision layout and the chainback are shared --:
// phases 0 and 1 pair register-index bits (registers r and r + 2, r and r + 1: no cross-lane traffic at all), phases 2..7 the lane
// bits (the K=7 partner fetch, four times), the minimum is three v_min_u32 in the lane and one wave reduction, no barrier.
// Waves 1..3 fill the branch-metric table ahead, as in the K=7 kernel.
template <class C>
__global__ __launch_bounds__(THREADS) void acs_wave9s_kernel(Args a) {
    using namespace k9;
    constexpr int R = C::R, NC = 1 << R;
    constexpr int STEP_BYTES = NC * 8;
    constexpr int CH = (TBL_BYTES / STEP_BYTES / BLK) * BLK;
    static_assert(BLK % k9::NB == 0 && CH >= BLK, "a block is whole periods");
    __shared__ __attribute__((aligned(16))) unsigned char tbl[2][TBL_BYTES];

    const int f = blockIdx.x;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    // the wave index as a SCALAR: "wave == 0" must be a uniform branch for the compiler, or every scalar the worker wave carries from
    // step to step (the lazy offset of the spiral arithmetic) becomes a per-lane value under a divergent branch
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned char *syms = a.syms + (size_t)f * a.sym_stride;
    unsigned long long *rows = a.dec + ((size_t)f * a.cap_rows + a.row0) * 4;
    const int phi0 = a.row0 % k9::NB;
    const int pre = min(a.nsteps, (k9::NB - phi0) % k9::NB);
    const int nmain = ((a.nsteps - pre) / BLK) * BLK;
    const int nchunks = (nmain + CH - 1) / CH;

    auto fill = [&](int buf, int c) {
        const int s0 = pre + c * CH, cnt = min(CH, nmain - c * CH);
        for (int i = (int)tid - 64; i < cnt; i += THREADS - 64) {
            int s[R];
#pragma unroll
            for (int r = 0; r < R; r++) s[r] = syms[(size_t)(s0 + i) * R + r];
            uint2 *e = reinterpret_cast<uint2 *>(&tbl[buf][i * STEP_BYTES]);
#pragma unroll
            for (int cl = 0; cl < NC; cl++) {
                const int t = C::bm(s, (unsigned)cl);
                e[cl] = make_uint2((unsigned)t << metric_shift<C>(), (unsigned)C::bm_tc(t) << metric_shift<C>());
            }
        }
    };

    unsigned M[4] = {0, 0, 0, 0};
    Lazy lz;
    unsigned cls[k9::NB][4], aoff[k9::NB][4];
    bool up[k9::NB];  // phases 2..7: this lane holds the upper predecessor (the same for its four registers)
    if (wave == 0) {
#pragma unroll
        for (int p = 0; p < k9::NB; p++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned st = rotl8((unsigned)r * 64u + lane, p);
                cls[p][r] = bt_class<R>(st & (k9::H - 1u), a.poly);
                aoff[p][r] = cls[p][r] * 8u;
            }
            up[p] = p >= 2 ? ((lane >> (k9::NB - 1 - p)) & 1u) != 0 : false;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) M[r] = to_lane_metric<C>(a.metrics[(size_t)f * k9::N + rotl8((unsigned)r * 64u + lane, phi0)]);
    }
    // one trellis step at phase PHI: e[r] = {t, t'} of register r's butterfly; d[r] = decision of the new state now in register r
    auto step9 = [&](auto P, const uint2 (&e)[4], bool (&d)[4]) {
        constexpr int PHI = decltype(P)::value;
        if constexpr (PHI >= 2) {
#pragma unroll
            for (int r = 0; r < 4; r++) d[r] = acs<C>(M[r], partner<k9::NB - 1 - PHI>(M[r], lane), e[r].x, e[r].y, up[PHI], lz);
        } else {
            constexpr int S = PHI == 0 ? 2 : 1;  // partner register: r ^ 2 (position bit 7) or r ^ 1 (bit 6)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int lo = PHI == 0 ? k : 2 * k, hi = lo + S;
                const unsigned old_lo = M[lo], old_hi = M[hi];
                d[lo] = acs<C>(M[lo], old_hi, e[lo].x, e[lo].y, false, lz);
                d[hi] = acs<C>(M[hi], old_lo, e[hi].x, e[hi].y, true, lz);
            }
        }
        if constexpr (C::renorm) {  // state 0 is position 0: register 0 of lane 0; lazy offset as in renormalise()
            const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)M[0]);
            if (m0 > (unsigned)C::renorm_thr + lz.off) {
                lz.off = wave_min_u32(min(min(M[0], M[1]), min(M[2], M[3])));
                lz.cap = lz.off + 255u;
            }
        }
    };
    auto slow_step = [&](int i) {
        int s[R];
#pragma unroll
        for (int r = 0; r < R; r++) s[r] = syms[(size_t)i * R + r];
        const int phi = (a.row0 + i) % k9::NB;
        bool d[4] = {false, false, false, false};
        sfor<k9::NB>([&](auto P) {
            constexpr int PHI = decltype(P)::value;
            if (phi == PHI) {  // uniform
                uint2 e[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int t = C::bm(s, cls[PHI][r]);
                    e[r] = make_uint2((unsigned)t << metric_shift<C>(), (unsigned)C::bm_tc(t) << metric_shift<C>());
                }
                step9(P, e, d);
            }
        });
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned long long row = __builtin_amdgcn_ballot_w64(d[r]);
            if (lane == 0) rows[(size_t)i * 4 + r] = row;
        }
    };

    if (wave == 0) {
        for (int i = 0; i < pre; i++) slow_step(i);
    } else if (nchunks > 0) {
        fill(0, 0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        if (wave == 0) {
            const int cnt = min(CH, nmain - c * CH);
            const unsigned char *tb = tbl[c & 1];
            for (int b0 = 0; b0 < cnt; b0 += BLK) {
                unsigned acc[4][2] = {};
                const unsigned char *blk = tb + b0 * STEP_BYTES;
                fold_offset<C, 4>(M, lz);
                uint2 en[k9::NB][4];  // table entries fetched one period ahead (see acs_wave_kernel)
#pragma unroll
                for (int q = 0; q < k9::NB; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) en[q][r] = *reinterpret_cast<const uint2 *>(blk + aoff[q][r] + q * STEP_BYTES);
                sfor<BLK>([&](auto J) {
                    constexpr int j = decltype(J)::value, PHI = j % k9::NB;
                    uint2 e[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        e[r] = en[PHI][r];
                        if constexpr (j + k9::NB < BLK) en[PHI][r] = *reinterpret_cast<const uint2 *>(blk + aoff[PHI][r] + (j + k9::NB) * STEP_BYTES);
                    }
                    bool d[4];
                    step9(std::integral_constant<int, PHI>{}, e, d);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned long long row = __builtin_amdgcn_ballot_w64(d[r]);
                        acc[r][0] = writelane<j>(acc[r][0], (unsigned)row);
                        acc[r][1] = writelane<j>(acc[r][1], (unsigned)(row >> 32));
                    }
                });
                if (lane < (unsigned)BLK) {  // lane j holds the four words of row j: 32 contiguous bytes
                    uint4 *dst = reinterpret_cast<uint4 *>(rows + (size_t)(pre + c * CH + b0 + (int)lane) * 4);
                    dst[0] = make_uint4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
                    dst[1] = make_uint4(acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
                }
            }
        } else if (c + 1 < nchunks) {
            fill((c + 1) & 1, c + 1);
        }
        __syncthreads();
    }
    if (wave == 0) {
        for (int i = pre + nmain; i < a.nsteps; i++) slow_step(i);
#pragma unroll
        for (int r = 0; r < 4; r++)
            a.metrics[(size_t)f * k9::N + rotl8((unsigned)r * 64u + lane, (a.row0 + a.nsteps) % k9::NB)] = (int16_t)from_lane_metric<C>(M[r], lz);
    }
}

// ---------------------------------------------------------------------------------------------------- chainback
// chainback_viterbi27_sse2 (viterbi27_sse2.cpp:78-105) / chainback_spiral47 (spiral47.cpp:84-121) over position-ordered rows:
// one wave per frame.  The walk runs in position space (chainback_k7_lds_kernel in acs_regs.hip explains it): p =
// rotr^rot(state), and one traceback step replaces bit (6 - rot) mod 6 of p by the decision.
//
// One frame's walk is a chain of dependent steps, so the 64 lanes SHARE it: the frame is cut into at most 64 segments of S
// bits (a multiple of 96 = lcm(32 output bits, 6 phases), so every lane is at the same phase in the same loop step and a
// segment is whole dwords of the output).  Lane 0 walks the top segment from the caller's end state -- exact.  Every other lane
// starts W = 96 rows above its segment from a guess (state 0): tracebacks from any state merge with the survivor path within a
// few constraint lengths, and a guess is never trusted: a lane records the state it had when it entered its segment and the
// one it left it with, and while some lane's entry state differs from what the lane above it really ended with, those lanes walk
// their segment again from that state.  Lane 0 is exact, so each round makes at least one more lane exact; the walk is a
// deterministic function of (row, state), so when no lane differs the bytes are those of the reference's single walk.
// (The same idea as the segment-parallel chainback of the K >= 15 codes in chainback_spec.hip, inside one wave.)
// Short frames, ragged top bits and unaligned outputs take the reference's bit-at-a-time form.
struct CbArgs {
    const unsigned long long *dec;
    int cap_rows, rows_written, nframes;
    unsigned char *data;
    size_t data_stride;
    unsigned nbits, endstate;
};

// KK = 7: one 64-bit word per row, segments of 96 bits (lcm of 32 output bits and 6 phases), 96 warm-up rows, 24 rows per round
// trip.  KK = 9 (acs_wave9_kernel below): four words per row (word = position >> 6), segments of 32 bits, 128 warm-up rows, 8 rows
// per round trip.
template <int KK>
__global__ __launch_bounds__(64) void chainback_wave_kernel(CbArgs a) {
    constexpr int NBK = KK - 1, WORDS = (1 << NBK) / 64;
    constexpr unsigned NK = 1u << NBK;
    constexpr int add = (NBK < 8) ? 8 - NBK : 0;  // ADDSHIFT                         spiral47.cpp:92-101
    constexpr int SEG_UNIT = KK == 7 ? 96 : 32, SEG_WARM = KK == 7 ? 96 : 128, SEG_BATCH = KK == 7 ? 24 : 8;
    static_assert(KK - 2 + add == 7 && SEG_UNIT % 32 == 0 && SEG_UNIT % NBK == 0 && SEG_WARM % SEG_BATCH == 0 && SEG_BATCH % NBK == 0 && SEG_UNIT % SEG_BATCH == 0, "geometry");
    const int f = blockIdx.x;
    const unsigned lane = threadIdx.x;
    const unsigned long long *rows = a.dec + (size_t)f * a.cap_rows * WORDS;
    unsigned char *out = a.data + (size_t)f * a.data_stride;
    unsigned e = (a.endstate % NK) << add;
    int rot = (int)(a.nbits % NBK);  // (r + 1) mod NB at the first row visited, r = nbits - 1 + NB

    auto slow = [&](unsigned i) {  // decoded bit i from row i + NB, every lane the same walk
        const long r = (long)i + NBK;
        const unsigned st = e >> add;
        const unsigned p = rot == 0 ? st : (((st >> rot) | (st << (NBK - rot))) & (NK - 1u));
        unsigned k = 0;
        if (r < a.rows_written) k = (unsigned)(rows[r * WORDS + (p >> 6)] >> (p & 63u)) & 1u;
        e = (e >> 1) | (k << 7);
        if ((i & 7u) == 0 && lane == 0) out[i >> 3] = (unsigned char)e;
        rot = rot == 0 ? NBK - 1 : rot - 1;
    };

    unsigned i = a.nbits;
    const bool out_aligned = (reinterpret_cast<uintptr_t>(out) & 3) == 0;  // this frame's bytes (dword stores)
    if (out_aligned && a.nbits >= 192u) {
        while (i & 31u) slow(--i);  // the ragged top: segments are whole dwords of the output
        const unsigned nb0 = i;
        const unsigned S = SEG_UNIT * ((nb0 + SEG_UNIT * 64u - 1u) / (SEG_UNIT * 64u));  // at most 64 segments
        const long hi = (long)nb0 - (long)lane * S;            // this lane decodes bits [lo, hi)
        const long lo = hi - (long)S > 0 ? hi - (long)S : 0;
        const bool active = hi > 0;
        // position of the exact state at bit nb0 - 1
        const unsigned st0 = e >> add;
        const unsigned q_exact = rot == 0 ? st0 : (((st0 >> rot) | (st0 << (NBK - rot))) & (NK - 1u));
        const int rot0 = rot;  // phase of loop step 0 of every (re-)walk: all segment tops are congruent mod NB
        // lanes whose warm-up would start above bit nb0 - 1 start AT it, with the exact state
        const bool from_top = hi + SEG_WARM > (long)nb0;
        unsigned q = from_top ? q_exact : 0u;
        unsigned q_in = q, q_out = 0u, h = e << 24;

        // steps [g_lo, g_hi) of the walk that starts W rows above the segment: bit index hi + W - 1 - g for this lane
        auto walk = [&](int g_lo, int g_hi, bool emit, bool enabled) __attribute__((always_inline)) {
            for (int g0 = g_lo; g0 < g_hi; g0 += SEG_BATCH) {
                unsigned long long w[SEG_BATCH][WORDS];
                sfor<SEG_BATCH>([&](auto D) {
                    constexpr int d = decltype(D)::value;
                    const long bi = hi + SEG_WARM - 1 - (g0 + d);
                    const long r = bi + NBK;
                    const bool ok = enabled && bi >= lo && bi < (long)nb0 && r < a.rows_written;
                    const unsigned long long *rp = rows + (ok ? r : 0) * WORDS;
                    if constexpr (WORDS == 1) {
                        w[d][0] = ok ? rp[0] : 0ull;
                    } else {
                        const ulonglong2 v0 = reinterpret_cast<const ulonglong2 *>(rp)[0], v1 = reinterpret_cast<const ulonglong2 *>(rp)[1];
                        w[d][0] = ok ? v0.x : 0ull; w[d][1] = ok ? v0.y : 0ull; w[d][2] = ok ? v1.x : 0ull; w[d][3] = ok ? v1.y : 0ull;
                    }
                });
                // jb of step g: (NB - rot_g) mod NB with rot_g = (rot0 - g) mod NB; g0 is a multiple of NB
                sfor<SEG_BATCH>([&](auto D) {
                    constexpr int d = decltype(D)::value;
                    const long bi = hi + SEG_WARM - 1 - (g0 + d);
                    const bool ok = enabled && bi >= lo && bi < (long)nb0;
                    const int jb = (NBK - ((rot0 - d) % NBK + NBK) % NBK) % NBK;  // uniform
                    unsigned long long word;
                    if constexpr (WORDS == 1) {
                        word = w[d][0];
                    } else {
                        const unsigned long long a0 = (q & 64u) ? w[d][1] : w[d][0], a1 = (q & 64u) ? w[d][3] : w[d][2];
                        word = (q & 128u) ? a1 : a0;
                    }
                    const unsigned tt = (unsigned)(word >> (q & 63u));
                    const unsigned nq = (q & ~(1u << jb)) | ((tt & 1u) << jb);
                    const unsigned nh = (h >> 1) | (tt << 31);
                    q = ok ? nq : q;
                    h = ok ? nh : h;
                    if (emit && ok && (bi & 31) == 0) *reinterpret_cast<unsigned *>(out + (bi >> 3)) = __builtin_bswap32(h);
                });
            }
        };
        walk(0, SEG_WARM, false, active);
        q_in = q;
        walk(SEG_WARM, SEG_WARM + (int)S, true, active);
        q_out = q;
        // verify from the top down: lane l entered with q_in, the lane above left with q_out
        for (int round = 0; round < 64; round++) {
            unsigned above = (unsigned)__shfl_up((int)q_out, 1);
            const bool redo = active && lane > 0 && !from_top && q_in != above;
            if (__builtin_amdgcn_ballot_w64(redo) == 0ull) break;
            q = above;
            q_in = redo ? above : q_in;
            const unsigned keep = q_out;
            walk(SEG_WARM, SEG_WARM + (int)S, true, redo);
            q_out = redo ? q : keep;
        }
        return;
    }
    while (i > 0) slow(--i);
}

}  // namespace wave7

bool wave_code_supported(int code) {
    return code == VHIP_KA9Q27 || code == VHIP_SPIRAL47 || code == VHIP_SPIRAL27 || code == VHIP_KA9Q29 || code == VHIP_SPIRAL49 || code == VHIP_SPIRAL29;
}

hipError_t launch_acs_wave(int code, const AcsLdsArgs &l, hipStream_t stream) {
    wave7::Args a;
    a.syms = l.syms;
    a.sym_stride = l.sym_stride;
    a.nsteps = l.nsteps;
    a.row0 = l.row0;
    a.cap_rows = l.cap_rows;
    a.nframes = l.nframes;
    a.dec = reinterpret_cast<unsigned long long *>(l.dec);
    a.metrics = l.metrics;
    for (int r = 0; r < 8; r++) a.poly[r] = l.poly[r];
    switch (code) {
    case VHIP_KA9Q27: hipLaunchKernelGGL(wave7::acs_wave_kernel<Code27>, dim3(a.nframes), dim3(wave7::THREADS), 0, stream, a); break;
    case VHIP_SPIRAL47: hipLaunchKernelGGL(wave7::acs_wave_kernel<Code47>, dim3(a.nframes), dim3(wave7::THREADS), 0, stream, a); break;
    case VHIP_SPIRAL27: hipLaunchKernelGGL(wave7::acs_wave_kernel<CodeS27>, dim3(a.nframes), dim3(wave7::THREADS), 0, stream, a); break;
    case VHIP_KA9Q29: hipLaunchKernelGGL(wave7::acs_wave9_kernel<Code29>, dim3(a.nframes), dim3(256), 0, stream, a); break;
    // spiral K=9: one wave with four states per lane (no barrier behind the per-step minimum)
    case VHIP_SPIRAL49: hipLaunchKernelGGL(wave7::acs_wave9s_kernel<Code49>, dim3(a.nframes), dim3(wave7::THREADS), 0, stream, a); break;
    case VHIP_SPIRAL29: hipLaunchKernelGGL(wave7::acs_wave9s_kernel<CodeS29>, dim3(a.nframes), dim3(wave7::THREADS), 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_chainback_wave(const ChainbackRowsArgs &c, hipStream_t stream) {
    wave7::CbArgs a;
    a.dec = reinterpret_cast<const unsigned long long *>(c.dec);
    a.cap_rows = c.cap_rows;
    a.rows_written = c.rows_written;
    a.nframes = c.nframes;
    a.data = c.data;
    a.data_stride = c.data_stride;
    a.nbits = c.nbits;
    a.endstate = c.endstate;
    if (c.K == 7) hipLaunchKernelGGL(wave7::chainback_wave_kernel<7>, dim3(a.nframes), dim3(64), 0, stream, a);
    else if (c.K == 9) hipLaunchKernelGGL(wave7::chainback_wave_kernel<9>, dim3(a.nframes), dim3(64), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace vh
