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

// One wave walks decoded-bit indices i = i_from, i_from-1, ..., i_to+1 (the register `e` is the reference's own: the state,
// shifted for K < 9) and returns the register it ends with.  OUT: bytes are stored as the reference stores them (every
// time i reaches a multiple of 8); a walk without OUT only carries the register forward.
template <int LAY, int K, bool K224, bool OUT>
__device__ __forceinline__ unsigned spec_walk(const ChainbackRowsArgs &a, const unsigned char *rows, unsigned char *out, unsigned e,
                                              long i_from, long i_to) {
    constexpr int NB = K - 1, DEPTH = 6;
    constexpr int add = (NB < 8) ? 8 - NB : 0, sub = (NB > 8) ? NB - 8 : 0;
    const unsigned lane = threadIdx.x;
    const long tail = K224 ? 0 : NB;  // chainback_viterbi224_sse2 does not skip the tail rows

    // tree node of this lane: depth d, assumed bits b_1..b_d = bits d-1..0 of `path` (first assumed bit = MSB)
    const unsigned node = lane + 1;
    const int d = 31 - __clz(node);
    const unsigned path = node - (1u << d);
    const bool live = lane < 63;

    unsigned dbyte = 0;
    long i = i_from;  // bits i-1 ... i_to remain
    while (i > i_to) {
        const unsigned st0 = K224 ? e : (e >> add);
        // state this lane would be in after its d assumed decisions
        unsigned st = st0;
        for (int s = 0; s < DEPTH - 1; s++)
            if (s < d) st = (st >> 1) | (((path >> (d - 1 - s)) & 1u) << (NB - 1));
        const long bi = i - 1 - d;  // decoded-bit index this lane serves
        unsigned k = 0;
        if (live && bi >= i_to && bi + tail < a.rows_written) k = fetch_bit<LAY, K>(rows, bi + tail, st);
        // resolve the true path: node 1 -> 2*node + k
        unsigned cur = 1;
        const int nres = i - i_to < DEPTH ? (int)(i - i_to) : DEPTH;
        for (int s = 0; s < nres; s++) {
            const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)k, (int)(cur - 1)) & 1u;
            --i;
            if (K224) {
                // viterbi224_sse2.cpp:96-103
                dbyte = ((e & 1u) << 7) | (dbyte >> 1);
                if (OUT && (i & 7) == 0 && lane == 0) out[i >> 3] = (unsigned char)dbyte;
                e = (kb << (K - 2)) | (e >> 1);
            } else {
                // viterbi615_sse2.cpp:86-88 / spiral47.cpp:116-118
                e = (e >> 1) | (kb << (K - 2 + add));
                if (OUT && (i & 7) == 0 && lane == 0) out[i >> 3] = (unsigned char)(e >> sub);
            }
            cur = 2 * cur + kb;
        }
    }
    return e;
}

template <int K, bool K224>
__device__ __forceinline__ unsigned spec_first_register(unsigned endstate) {
    constexpr int NB = K - 1;
    constexpr unsigned N = 1u << NB;
    constexpr int add = (NB < 8) ? 8 - NB : 0;
    return K224 ? (endstate & (N - 1u)) : ((endstate % N) << add);  // the reference's own register
}

template <int LAY, int K, bool K224>
__global__ __launch_bounds__(64) void chainback_spec_kernel(ChainbackRowsArgs a) {
    constexpr unsigned N = 1u << (K - 1);
    const long f = blockIdx.x;
    const unsigned char *rows = a.dec + f * (long)a.cap_rows * (long)(N / 8);
    unsigned char *out = a.data + f * (long)a.data_stride;
    (void)spec_walk<LAY, K, K224, true>(a, rows, out, spec_first_register<K, K224>(a.endstate), (long)a.nbits, 0);
}

// Segment-parallel form for FEW frames (K=24: one frame; the reference-style one-frame handles of K=15), where the walk above is a
// single chain of DRAM round trips.  The frame's nbits are cut into segments of seg_bits; segment j is walked by its own wave
// from i = hi_j + seg_ovl with a GUESSED register (0): tracebacks from any state merge with the survivor path, so after the
// seg_ovl warm-up rows the register is, almost always, the one the full walk has at i = hi_j.  "Almost always" is not
// bit-exact, so it is never trusted: every wave records the register it had at hi_j and the one it ended with at lo_j, and the
// last wave of the frame to finish (an atomic ticket; nobody waits for anybody) checks the chain from the top segment -- which
// starts from the caller's end state and is exact -- downwards: where the register a segment assumed at hi_j differs from the
// one the segment above really ended with, that segment is walked again from the true register, overwriting its bytes.  The walk
// is a deterministic function of (i, register), so equal registers at hi_j prove the segment's bytes.
template <int LAY, int K, bool K224>
__global__ __launch_bounds__(64) void chainback_spec_seg_kernel(ChainbackRowsArgs a) {
    constexpr unsigned N = 1u << (K - 1);
    const int nseg = a.nseg;
    const long f = blockIdx.x / nseg;
    const int j = nseg - 1 - (int)(blockIdx.x % nseg);  // top segment first
    const unsigned char *rows = a.dec + f * (long)a.cap_rows * (long)(N / 8);
    unsigned char *out = a.data + f * (long)a.data_stride;
    // scratch: [frame][ticket, re-walked segments] at a place that does not depend on the geometry (the ticket must read 0 when a
    // launch begins, whatever the previous launch's nseg was), then [frame][assumed[nseg], ended[nseg]]
    unsigned *tk = a.seg_scratch + f * 2;
    unsigned *sc = a.seg_scratch + 2 * (long)a.nframes + f * (long)(2 * nseg);
    const long lo = (long)j * a.seg_bits;
    const long hi = lo + a.seg_bits < (long)a.nbits ? lo + a.seg_bits : (long)a.nbits;

    unsigned e;
    if (hi + a.seg_ovl >= (long)a.nbits) e = spec_walk<LAY, K, K224, false>(a, rows, out, spec_first_register<K, K224>(a.endstate), (long)a.nbits, hi);
    else e = spec_walk<LAY, K, K224, false>(a, rows, out, 0u, hi + a.seg_ovl, hi);
    const unsigned assumed = e;
    e = spec_walk<LAY, K, K224, true>(a, rows, out, e, hi, lo);

    unsigned ticket = 0;
    if (threadIdx.x == 0) {
        __hip_atomic_store(sc + j, assumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sc + nseg + j, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);  // releases the bytes and the two words
    }
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    if (ticket != (unsigned)(nseg - 1)) return;

    // last wave of this frame: every other segment's words and bytes are visible
    unsigned truth = __hip_atomic_load(sc + nseg + (nseg - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned rewalked = 0;
    for (int s = nseg - 2; s >= 0; s--) {
        const unsigned as = __hip_atomic_load(sc + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (as == truth) {
            truth = __hip_atomic_load(sc + nseg + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long slo = (long)s * a.seg_bits;
            truth = spec_walk<LAY, K, K224, true>(a, rows, out, truth, slo + a.seg_bits, slo);
            rewalked++;
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(tk + 1, rewalked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ticket counter ready for the next launch
    }
}

template <int LAY, int K, bool K224>
static void launch_spec(const ChainbackRowsArgs &a, hipStream_t stream) {
    if (a.seg_scratch && a.nseg > 1)
        hipLaunchKernelGGL((chainback_spec_seg_kernel<LAY, K, K224>), dim3((unsigned)a.nframes * (unsigned)a.nseg), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL((chainback_spec_kernel<LAY, K, K224>), dim3(a.nframes), dim3(64), 0, stream, a);
}

hipError_t launch_chainback_spec(int layout, const ChainbackRowsArgs &a, hipStream_t stream) {
    if (a.K == 15 && layout == LAY_NATURAL) launch_spec<LAY_NATURAL, 15, false>(a, stream);
    else if (a.K == 15 && layout == LAY_K15) launch_spec<LAY_K15, 15, false>(a, stream);
    else if (a.K == 15 && layout == LAY_K15_SIGN_BYTES) launch_spec<LAY_K15_SIGN_BYTES, 15, false>(a, stream);
    else if (a.K == 24 && layout == LAY_NATURAL) launch_spec<LAY_NATURAL, 24, true>(a, stream);
    else if (a.K == 24 && layout == LAY_K24F) launch_spec<LAY_K24F, 24, true>(a, stream);
    else if (a.K == 24 && layout == LAY_K24T) launch_spec<LAY_K24T, 24, true>(a, stream);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace vh
