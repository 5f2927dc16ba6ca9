// chainback.hip -- traceback over natural decision rows: one thread per frame, or (K <= 9) one wave per frame with the
// lanes sharing the frame's walk (chainback_rows_seg_kernel).
//
// Replaces chainback_viterbi27_sse2 (viterbi27_sse2.cpp:78-105), chainback_viterbi29_sse2
// (viterbi29_sse2.cpp:69-94), chainback_viterbi615_sse2 (viterbi615_sse2.cpp:65-91, with the 32-bit word
// semantics the author measured -- SURVEY.md §0.3), chainback_spiral47/49 (spiral47.cpp:84-121) and
// chainback_viterbi224_sse2 (viterbi224_sse2.cpp:79-121, which does NOT skip the tail rows and emits the bit
// that falls off the right end of the state -- SURVEY.md §0.4; reproduced as is).
//
// The walk is a chain of dependent loads (the next word address depends on the bit just read), so the only
// parallelism is across frames; per decoded bit it reads one 32-bit decision word and every 8th step it
// stores one byte (4 1/8 algorithmic bytes per bit).
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "kernels.h"

namespace vh {

namespace {
template <class F, int... Is>
__device__ __forceinline__ void cb_sfor_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int NN, class F>
__device__ __forceinline__ void cb_sfor(F &&f) {  // compile-time loop: the row registers below must never be indexed dynamically
    cb_sfor_impl(f, std::make_integer_sequence<int, NN>{});
}
}  // namespace

__global__ __launch_bounds__(64) void chainback_rows_kernel(ChainbackRowsArgs a) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.nframes) return;
    const int K = a.K;
    const unsigned N = 1u << (K - 1);
    const size_t row_bytes = N / 8;
    const unsigned char *rows = a.dec + (size_t)f * a.cap_rows * row_bytes;
    unsigned char *out = a.data + (size_t)f * a.data_stride;

    if (a.k224) {
        // viterbi224_sse2.cpp:90-104
        unsigned e = a.endstate & (N - 1);
        unsigned dbyte = 0;
        for (unsigned i = a.nbits; i-- > 0;) {
            dbyte = ((e & 1u) << 7) | (dbyte >> 1);
            if ((i & 7u) == 0) out[i >> 3] = (unsigned char)dbyte;
            unsigned bit = 0;
            if ((int)i < a.rows_written) {
                const unsigned w = *reinterpret_cast<const unsigned *>(rows + (size_t)i * row_bytes + (size_t)(e >> 5) * 4);
                bit = (w >> (e & 31u)) & 1u;
            }
            e = (bit << (K - 2)) | (e >> 1);
        }
        return;
    }
    // spiral47.cpp:92-119 generic form; ka9q27 hard-codes add=2 (viterbi27_sse2.cpp:90-102), ka9q29 add=sub=0
    // (viterbi29_sse2.cpp:80-91), ka9q615 sub=6 (viterbi615_sse2.cpp:74-88)
    const int add = (K - 1 < 8) ? 8 - (K - 1) : 0;
    const int sub = (K - 1 > 8) ? (K - 1) - 8 : 0;
    unsigned e = (a.endstate % N) << add;
    for (unsigned i = a.nbits; i-- > 0;) {
        const unsigned st = e >> add;
        const size_t r = (size_t)i + (size_t)(K - 1);  // look past the tail
        unsigned k = 0;
        if ((long long)r < (long long)a.rows_written) {
            const unsigned w = *reinterpret_cast<const unsigned *>(rows + r * row_bytes + (size_t)(st >> 5) * 4);
            k = (w >> (st & 31u)) & 1u;
        }
        e = (e >> 1) | (k << (K - 2 + add));
        if ((i & 7u) == 0) out[i >> 3] = (unsigned char)(e >> sub);  // last store to each byte wins in the reference
    }
}

// K <= 9 (whole rows of 8 or 32 bytes): one WAVE per frame, the 64 lanes sharing the frame's walk.  The frame is cut into at
// most 64 segments of whole output dwords; lane 0 walks the top one from the caller's end state, every other lane starts WARM
// rows above its segment from state 0 -- tracebacks merge with the survivor path within a few constraint lengths -- and records
// the state it entered its segment with and the one it left it with.  A guess is never trusted: while some lane's entry state
// differs from what the lane above it really ended with, those lanes walk their segment again from that state; lane 0 is
// exact, every round makes at least one more lane exact, and the walk is a deterministic function of (row, state), so the
// bytes are those of the single walk above (same scheme as chainback_wave_kernel in acs_wave.hip and, across waves, as
// chainback_spec.hip).  A row does not depend on the state, so each lane fetches WHOLE rows ahead of its walk (24 per round
// trip) and only the word / bit selection is on the dependent chain: a one-frame K=9 handle decodes 4096 bits in ~10 us
// instead of 0.5 ms of dependent loads.
template <int K>
__global__ __launch_bounds__(64) void chainback_rows_seg_kernel(ChainbackRowsArgs a) {
    constexpr int NB = K - 1, add = (NB < 8) ? 8 - NB : 0, RW = (1 << NB) / 32;  // RW dwords per row (2 or 8)
    constexpr unsigned N = 1u << NB;
    constexpr int WARM = 16 * NB, BATCH = 24 / (RW / 2);  // rows in flight per lane: 24 (K=7), 6 (K=9)
    static_assert(K - 2 + add == 7 && (RW == 2 || RW == 8), "the 8-bit register of the reference is the last eight decisions");
    const int f = blockIdx.x;
    const unsigned lane = threadIdx.x;
    const unsigned *rows = reinterpret_cast<const unsigned *>(a.dec) + (size_t)f * a.cap_rows * RW;
    unsigned char *out = a.data + (size_t)f * a.data_stride;
    unsigned e = (a.endstate % N) << add;
    unsigned i = a.nbits;
    while (i & 31u) {  // the ragged top, bit at a time, every lane the same walk (segments are whole dwords of the output)
        --i;
        const long r = (long)i + NB;
        const unsigned st = e >> add;
        unsigned k = 0;
        if (r < a.rows_written) k = (rows[r * RW + (st >> 5)] >> (st & 31u)) & 1u;
        e = (e >> 1) | (k << 7);
        if ((i & 7u) == 0 && lane == 0) out[i >> 3] = (unsigned char)e;
    }
    const unsigned nb0 = i;
    if (nb0 == 0) return;
    const unsigned S = 32u * ((nb0 + 32u * 64u - 1u) / (32u * 64u));
    const long hi = (long)nb0 - (long)lane * S, lo = hi - (long)S > 0 ? hi - (long)S : 0;
    const bool active = hi > 0;
    const bool from_top = hi + WARM > (long)nb0;  // the warm-up would start above the first bit: start AT it, exactly
    unsigned st = from_top ? (e >> add) : 0u, h = e << 24;

    auto walk = [&](int g_lo, int g_hi, bool emit, bool enabled) __attribute__((always_inline)) {  // steps of the walk that starts WARM rows above the segment
        for (int g0 = g_lo; g0 < g_hi; g0 += BATCH) {
            unsigned w[BATCH][RW];
            cb_sfor<BATCH>([&](auto D) {
                constexpr int d = decltype(D)::value;
                const long bi = hi + WARM - 1 - (g0 + d), r = bi + NB;
                const bool ok = enabled && g0 + d < g_hi && bi >= lo && bi < (long)nb0 && r < a.rows_written;
                const unsigned *rp = rows + (ok ? r : 0) * RW;
                if constexpr (RW == 2) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(rp);
                    w[d][0] = ok ? v.x : 0u;
                    w[d][1] = ok ? v.y : 0u;
                } else {
                    const uint4 v0 = reinterpret_cast<const uint4 *>(rp)[0], v1 = reinterpret_cast<const uint4 *>(rp)[1];
                    w[d][0] = ok ? v0.x : 0u; w[d][1] = ok ? v0.y : 0u; w[d][2] = ok ? v0.z : 0u; w[d][3] = ok ? v0.w : 0u;
                    w[d][4] = ok ? v1.x : 0u; w[d][5] = ok ? v1.y : 0u; w[d][6] = ok ? v1.z : 0u; w[d][7] = ok ? v1.w : 0u;
                }
            });
            cb_sfor<BATCH>([&](auto D) {
                constexpr int d = decltype(D)::value;
                const long bi = hi + WARM - 1 - (g0 + d);
                const bool ok = enabled && g0 + d < g_hi && bi >= lo && bi < (long)nb0;
                unsigned word;
                if constexpr (RW == 2) {
                    // (a select between two elements of w would be turned into an indexed access and put w into scratch)
                    word = (unsigned)(((((unsigned long long)w[d][1]) << 32) | w[d][0]) >> (st & 32u));
                } else {
                    const unsigned a0 = (st & 32u) ? w[d][1] : w[d][0], a1 = (st & 32u) ? w[d][3] : w[d][2];
                    const unsigned a2 = (st & 32u) ? w[d][5] : w[d][4], a3 = (st & 32u) ? w[d][7] : w[d][6];
                    const unsigned b0 = (st & 64u) ? a1 : a0, b1 = (st & 64u) ? a3 : a2;
                    word = (st & 128u) ? b1 : b0;
                }
                const unsigned k = (word >> (st & 31u)) & 1u;
                const unsigned ns = (st >> 1) | (k << (NB - 1)), nh = (h >> 1) | (k << 31);
                st = ok ? ns : st;
                h = ok ? nh : h;
                if (emit && ok && (bi & 31) == 0) *reinterpret_cast<unsigned *>(out + (bi >> 3)) = __builtin_bswap32(h);
            });
        }
    };
    walk(0, WARM, false, active);
    unsigned st_in = st;
    walk(WARM, WARM + (int)S, true, active);
    unsigned st_out = st;
    for (int round = 0; round < 64; round++) {
        const unsigned above = (unsigned)__shfl_up((int)st_out, 1);
        const bool redo = active && lane > 0 && !from_top && st_in != above;
        if (__builtin_amdgcn_ballot_w64(redo) == 0ull) break;
        const unsigned keep = st_out;
        st = above;
        st_in = redo ? above : st_in;
        walk(WARM, WARM + (int)S, true, redo);
        st_out = redo ? st : keep;
    }
}

hipError_t launch_chainback_rows(const ChainbackRowsArgs &a, hipStream_t stream) {
    // dword stores of whole segments need this frame's bytes 4-byte aligned (every frame: base and stride)
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.data) | (a.nframes > 1 ? a.data_stride : 0)) & 3) == 0;
    // beyond ~4096 frames one thread per frame has enough frames to hide its dependent loads and does less work in total
    // (K=9: 4096 frames 0.28 ms against 0.48, 8192 frames 0.58 against 0.52)
    if (!a.k224 && aligned && a.nbits >= 192u && a.nframes <= 4096 && (a.K == 7 || a.K == 9)) {
        if (a.K == 7) hipLaunchKernelGGL(chainback_rows_seg_kernel<7>, dim3(a.nframes), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL(chainback_rows_seg_kernel<9>, dim3(a.nframes), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    const int blocks = (a.nframes + 63) / 64;
    hipLaunchKernelGGL(chainback_rows_kernel, dim3(blocks), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace vh
