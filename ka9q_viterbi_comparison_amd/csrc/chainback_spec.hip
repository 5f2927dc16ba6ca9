// chainback_spec.hip -- wave-parallel speculative traceback for the large codes (K=15, K=24).
//
// Same walk as chainback_viterbi615_sse2 (ka9q_libfec_port/viterbi615_sse2.cpp:65-91, 32-bit word semantics) and
// chainback_viterbi224_sse2 (viterbi224_sse2.cpp:79-121), bit for bit.  The walk is a chain of dependent loads:
// the row-(i-1) address depends on the bit just read from row i, and for K>=15 the rows are 2 KiB..1 MiB apart, so
// one thread per frame pays a full DRAM round trip per decoded bit (what the CPU code pays as a cache miss per bit).
// Here ONE WAVE serves a frame and explores the binary tree of the next 6 decisions at once: lane l is tree node
// l+1 (depth d = floor(log2(l+1)), path = the low d bits), assumes those d decision bits, derives the state it would
// be in, and loads ITS decision bit from row i-d.  All 63 loads are in flight together; the true path is then
// resolved with six v_readlane steps.  Six decoded bits per DRAM round trip instead of one, 63 four-byte loads
// instead of six -- the history is far larger than the bytes touched either way.
//
// Two variants were built for K=24 in round 2 and dropped (bit-exact, slower than this kernel's 5.2 Mbit/s): rounds aligned with
// the register groups of k24t_layout.h (inside a group the word offset of a row does not depend on the decisions, so one
// 8-byte load per row covers the group and the group below can be speculated over <= 32 candidates: 8-10 decisions per round
// trip, but the per-round index arithmetic and the serial resolve cost more than the round trips saved: 1.9 Mbit/s), and
// touching the rows 96 steps ahead to warm their address translations (rows are 1 MiB apart: 4.8 Mbit/s, i.e. translation
// misses are not what a round waits for).
#include <hip/hip_runtime.h>

#include "k15_layout.h"
#include "k24f_layout.h"
#include "k24t_layout.h"
#include "kernels.h"

namespace vh {

enum { LAY_NATURAL = 0, LAY_K15 = 1, LAY_K24F = 2, LAY_K15_SIGN_BYTES = 3, LAY_K24T = 4 };

// decision bit of new state `st` at row r for each layout
template <int LAY, int K>
__device__ __forceinline__ unsigned fetch_bit(const unsigned char *rows, long r, unsigned st) {
    constexpr int NB = K - 1;
    constexpr unsigned N = 1u << NB;
    if constexpr (LAY == LAY_NATURAL) {
        const unsigned w = *reinterpret_cast<const unsigned *>(rows + r * (long)(N / 8) + (long)(st >> 5) * 4);
        return (w >> (st & 31u)) & 1u;
    } else {
        const int rot = (int)((r + 1) % NB);
        const unsigned p = rot == 0 ? st : (((st >> rot) | (st << (NB - rot))) & (N - 1u));
        if constexpr (LAY == LAY_K15 || LAY == LAY_K15_SIGN_BYTES) {
            const int phi = rot == 0 ? NB - 1 : rot - 1;
            const unsigned t = phi < 7 ? (p & 127u) : (p >> 7), q = phi < 7 ? (p >> 7) : (p & 127u);
            const unsigned rho = q >> 1, h = q & 1u;
            const unsigned w = reinterpret_cast<const unsigned *>(rows)[r * 512L + (rho >> 4) * 128 + t];
            return (w >> k15_decision_bit(LAY == LAY_K15_SIGN_BYTES, rho, h)) & 1u;
        } else {
            unsigned widx, wbit;
            if constexpr (LAY == LAY_K24T) k24t_locate(p, rot == 0 ? NB - 1 : rot - 1, widx, wbit);
            else k24f_locate(p, rot == 0 ? NB - 1 : rot - 1, widx, wbit);
            return (reinterpret_cast<const unsigned *>(rows + r * (long)(N / 8))[widx] >> wbit) & 1u;
        }
    }
}

template <int LAY, int K, bool K224>
__global__ __launch_bounds__(64) void chainback_spec_kernel(ChainbackRowsArgs a) {
    constexpr int NB = K - 1, DEPTH = 6;
    constexpr unsigned N = 1u << NB;
    constexpr int add = (NB < 8) ? 8 - NB : 0, sub = (NB > 8) ? NB - 8 : 0;
    const long f = blockIdx.x;
    const unsigned lane = threadIdx.x;
    const unsigned char *rows = a.dec + f * (long)a.cap_rows * (long)(N / 8);
    unsigned char *out = a.data + f * (long)a.data_stride;
    const long tail = K224 ? 0 : NB;  // chainback_viterbi224_sse2 does not skip the tail rows

    // tree node of this lane: depth d, assumed bits b_1..b_d = bits d-1..0 of `path` (first assumed bit = MSB)
    const unsigned node = lane + 1;
    const int d = 31 - __clz(node);
    const unsigned path = node - (1u << d);
    const bool live = lane < 63;

    unsigned e = K224 ? (a.endstate & (N - 1u)) : ((a.endstate % N) << add);  // the reference's own register
    unsigned dbyte = 0;
    long i = (long)a.nbits;  // bits i-1 ... 0 remain
    while (i > 0) {
        const unsigned st0 = K224 ? e : (e >> add);
        // state this lane would be in after its d assumed decisions
        unsigned st = st0;
        for (int s = 0; s < DEPTH - 1; s++)
            if (s < d) st = (st >> 1) | (((path >> (d - 1 - s)) & 1u) << (NB - 1));
        const long bi = i - 1 - d;  // decoded-bit index this lane serves
        unsigned k = 0;
        if (live && bi >= 0 && bi + tail < a.rows_written) k = fetch_bit<LAY, K>(rows, bi + tail, st);
        // resolve the true path: node 1 -> 2*node + k
        unsigned cur = 1;
        const int nres = i < DEPTH ? (int)i : DEPTH;
        for (int s = 0; s < nres; s++) {
            const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)k, (int)(cur - 1)) & 1u;
            --i;
            if (K224) {
                // viterbi224_sse2.cpp:96-103
                dbyte = ((e & 1u) << 7) | (dbyte >> 1);
                if ((i & 7) == 0 && lane == 0) out[i >> 3] = (unsigned char)dbyte;
                e = (kb << (K - 2)) | (e >> 1);
            } else {
                // viterbi615_sse2.cpp:86-88 / spiral47.cpp:116-118
                e = (e >> 1) | (kb << (K - 2 + add));
                if ((i & 7) == 0 && lane == 0) out[i >> 3] = (unsigned char)(e >> sub);
            }
            cur = 2 * cur + kb;
        }
    }
}

hipError_t launch_chainback_spec(int layout, const ChainbackRowsArgs &a, hipStream_t stream) {
    const dim3 grid(a.nframes), block(64);
    if (a.K == 15 && layout == LAY_NATURAL) hipLaunchKernelGGL((chainback_spec_kernel<LAY_NATURAL, 15, false>), grid, block, 0, stream, a);
    else if (a.K == 15 && layout == LAY_K15) hipLaunchKernelGGL((chainback_spec_kernel<LAY_K15, 15, false>), grid, block, 0, stream, a);
    else if (a.K == 15 && layout == LAY_K15_SIGN_BYTES) hipLaunchKernelGGL((chainback_spec_kernel<LAY_K15_SIGN_BYTES, 15, false>), grid, block, 0, stream, a);
    else if (a.K == 24 && layout == LAY_NATURAL) hipLaunchKernelGGL((chainback_spec_kernel<LAY_NATURAL, 24, true>), grid, block, 0, stream, a);
    else if (a.K == 24 && layout == LAY_K24F) hipLaunchKernelGGL((chainback_spec_kernel<LAY_K24F, 24, true>), grid, block, 0, stream, a);
    else if (a.K == 24 && layout == LAY_K24T) hipLaunchKernelGGL((chainback_spec_kernel<LAY_K24T, 24, true>), grid, block, 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace vh
