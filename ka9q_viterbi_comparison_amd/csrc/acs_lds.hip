// acs_lds.hip -- generic ACS update kernel: one workgroup per frame, path metrics ping-pong in LDS.
//
// Replaces update_viterbi{27,29,615}_blk_sse2 (ka9q_libfec_port/viterbi27_sse2.cpp:119-175,
// viterbi29_sse2.cpp:108-164, viterbi615_sse2.cpp:104-191) and update_spiral4{7,9}
// (spiral/spiral47.cpp:131-538, spiral49.cpp:132-1540) for any polynomial set.
//
// Mapping: lane <-> new state.  A thread owns new states n = tid + i*THREADS; it reads the two predecessor
// metrics old[n>>1], old[(n>>1)+H] from LDS, adds the branch metric of butterfly n>>1, compares, writes
// new[n] to the other LDS buffer, and the wave's 64 decisions leave as ONE 64-bit __ballot word -- which is
// exactly 8 bytes of the reference's decision bitmap row (bit n = new state n, viterbi27_sse2.cpp:161-162).
// Renormalisation (viterbi615_sse2.cpp:160-183, spiral47.cpp:313-331) is a workgroup-uniform branch on
// new[0] followed by a wave-shuffle + LDS min reduction.
//
// HBM traffic per frame-step: R symbol bytes in (staged through LDS in chunks), N/8 decision bytes out.
// This is the any-polynomial fallback and the K=15 production kernel; K<=9 production kernels are in
// acs_regs.hip.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {

template <class C>
__device__ __forceinline__ void acs_select(int lower, int upper, int &surv, int &d) {
    if constexpr (C::metric == U8MOD) {
        // cmpgt_epi8(sub_epi8(m0,m1),0): signed 8-bit difference, tie -> lower     viterbi27_sse2.cpp:155-158
        d = ((int)(int8_t)(uint8_t)(lower - upper)) > 0;
        surv = d ? upper : lower;
    } else if constexpr (C::tie_upper) {
        // min + cmpeq(min, upper): tie -> upper       viterbi615_sse2.cpp:145-148, spiral47.cpp:224-227
        d = upper <= lower;
        surv = d ? upper : lower;
    } else {
        // cmpgt(lower, upper) then min: tie -> lower                                viterbi224_sse2.cpp:190-194
        d = lower > upper;
        surv = d ? upper : lower;
    }
}

template <class C>
__device__ __forceinline__ int metric_add(int m, int t) {
    if constexpr (C::metric == U8MOD) return (m + t) & 255;                       // add_epi8
    else if constexpr (C::metric == U8SAT) return min(m + t, 255);                // adds_epu8
    else return max(-32768, min(m + t, 32767));                                   // adds_epi16
}

constexpr int SYM_CHUNK_STEPS = 128;

template <class C, int THREADS>
__global__ __launch_bounds__(THREADS) void acs_lds_kernel(AcsLdsArgs a) {
    constexpr int K = C::K, R = C::R;
    constexpr int N = 1 << (K - 1), H = N / 2;
    constexpr int S = N / THREADS;  // new states per thread
    static_assert(N % THREADS == 0 && THREADS % 64 == 0, "bad geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int16_t *mbuf0 = reinterpret_cast<int16_t *>(smem);
    int16_t *mbuf1 = mbuf0 + N;
    unsigned char *symbuf = reinterpret_cast<unsigned char *>(mbuf1 + N);  // SYM_CHUNK_STEPS*R bytes
    int *red = reinterpret_cast<int *>(symbuf + SYM_CHUNK_STEPS * R + ((16 - (SYM_CHUNK_STEPS * R) % 16) % 16));

    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    if (f >= a.nframes) return;

    unsigned cls[S];
#pragma unroll
    for (int i = 0; i < S; i++) cls[i] = bt_class<R>((uint32_t)((tid + i * THREADS) >> 1), a.poly);

    int16_t *gm = a.metrics + (size_t)f * N;
    for (int i = tid; i < N; i += THREADS) mbuf0[i] = gm[i];
    int16_t *oldm = mbuf0, *newm = mbuf1;

    const unsigned char *fsyms = a.syms + (size_t)f * a.sym_stride;
    unsigned char *frow = a.dec + ((size_t)f * a.cap_rows + a.row0) * (size_t)(N / 8);

    for (int t0 = 0; t0 < a.nsteps; t0 += SYM_CHUNK_STEPS) {
        const int chunk = min(SYM_CHUNK_STEPS, a.nsteps - t0);
        __syncthreads();  // previous chunk fully consumed (and initial metric load visible)
        for (int i = tid; i < chunk * R; i += THREADS) symbuf[i] = fsyms[(size_t)t0 * R + i];
        __syncthreads();
        for (int tt = 0; tt < chunk; tt++) {
            int s[R];
#pragma unroll
            for (int r = 0; r < R; r++) s[r] = symbuf[tt * R + r];
            unsigned char *row = frow + (size_t)(t0 + tt) * (N / 8);
            int mine[S];
#pragma unroll
            for (int i = 0; i < S; i++) {
                const int n = tid + i * THREADS;
                const int j = n >> 1;
                const int t = C::bm(s, cls[i]);
                const int tc = C::bm_tc(t);
                const bool odd = n & 1;
                // new[2j]: m0 = old[j]+t, m1 = old[j+H]+t' ; new[2j+1]: m2 = old[j]+t', m3 = old[j+H]+t
                const int lower = metric_add<C>(oldm[j], odd ? tc : t);
                const int upper = metric_add<C>(oldm[j + H], odd ? t : tc);
                int surv, d;
                acs_select<C>(lower, upper, surv, d);
                mine[i] = surv;
                newm[n] = (int16_t)surv;
                const unsigned long long mask = __ballot(d);
                if ((tid & 63) == 0) *reinterpret_cast<unsigned long long *>(row + (n >> 3)) = mask;
            }
            __syncthreads();
            if constexpr (C::renorm) {
                const int v0 = newm[0];
                const bool fire = (C::metric == I16SAT) ? (v0 >= C::renorm_thr) : (v0 > C::renorm_thr);
                if (fire) {  // workgroup-uniform
                    int mn = mine[0];
#pragma unroll
                    for (int i = 1; i < S; i++) mn = min(mn, mine[i]);
                    mn = wave_min(mn);  // six DPP steps (kernels.h); __shfl_xor would be six ds_bpermute round trips
                    if constexpr (THREADS > 64) {
                        if ((tid & 63) == 0) red[tid >> 6] = mn;
                        __syncthreads();
                        mn = red[0];
#pragma unroll
                        for (int w = 1; w < THREADS / 64; w++) mn = min(mn, red[w]);
                    }
#pragma unroll
                    for (int i = 0; i < S; i++) {
                        const int n = tid + i * THREADS;
                        if constexpr (C::metric == I16SAT)
                            newm[n] = (int16_t)(uint16_t)(mine[i] - (mn + 32768));  // sub_epi16 wraps  viterbi615_sse2.cpp:166-182
                        else
                            newm[n] = (int16_t)max(mine[i] - mn, 0);  // subs_epu8                    spiral47.cpp:327-330
                    }
                    __syncthreads();
                }
            }
            int16_t *tmp = oldm;
            oldm = newm;
            newm = tmp;
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += THREADS) gm[i] = oldm[i];
}

template <class C, int THREADS>
static hipError_t launch_one(const AcsLdsArgs &a, hipStream_t stream) {
    constexpr int N = 1 << (C::K - 1);
    const size_t smem = 2 * N * sizeof(int16_t) + SYM_CHUNK_STEPS * C::R + 16 + 16 * sizeof(int) + 64;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&acs_lds_kernel<C, THREADS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((acs_lds_kernel<C, THREADS>), dim3(a.nframes), dim3(THREADS), smem, stream, a);
    return hipGetLastError();
}

hipError_t launch_acs_lds(int code, const AcsLdsArgs &a, hipStream_t stream) {
    switch (code) {
    case VHIP_KA9Q27: return launch_one<Code27, 64>(a, stream);
    case VHIP_SPIRAL47: return launch_one<Code47, 64>(a, stream);
    case VHIP_KA9Q29: return launch_one<Code29, 256>(a, stream);
    case VHIP_SPIRAL49: return launch_one<Code49, 256>(a, stream);
    case VHIP_KA9Q615: return launch_one<Code615, 1024>(a, stream);
    case VHIP_SPIRAL27: return launch_one<CodeS27, 64>(a, stream);
    case VHIP_SPIRAL29: return launch_one<CodeS29, 256>(a, stream);
    case VHIP_SPIRAL615: return launch_one<CodeS615, 1024>(a, stream);
    }
    return hipErrorInvalidValue;
}

// ---- init: fill path metrics (init_viterbi27_sse2 viterbi27_sse2.cpp:42-54 and siblings) -------------
// n_per_frame is 2^(K-1) >= 64: a thread writes eight consecutive metrics (16 bytes) of one frame, the state index is a mask,
// and only the vector that holds the start state differs from the fill value.  (The first version took a 64-bit modulo per
// 2-byte store; on a 4096-frame K=15 handle that was 1.7-3.9 ms per decode of VALU work beside the update kernel --
// profiles/r03_viterbi615_kernel_stats.csv -- for a 134 MB fill.)
__global__ __launch_bounds__(256) void init_metrics_kernel(int16_t *metrics, unsigned n_mask, size_t total8, int init_all, int init_start,
                                                           unsigned start_state) {
    const unsigned short a = (unsigned short)init_all, s = (unsigned short)init_start;
    const unsigned aa = a | ((unsigned)a << 16);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned st0 = (unsigned)(i * 8) & n_mask;  // state of the first of the eight
        uint4 v = make_uint4(aa, aa, aa, aa);
        if ((start_state & ~7u) == st0) {
            unsigned short e[8] = {a, a, a, a, a, a, a, a};
#pragma unroll
            for (int k = 0; k < 8; k++)
                if ((start_state & 7u) == (unsigned)k) e[k] = s;
            v = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
        }
        reinterpret_cast<uint4 *>(metrics)[i] = v;
    }
}

hipError_t launch_init_metrics(int16_t *metrics, size_t n_per_frame, int nframes, int init_all, int init_start,
                               unsigned start_state, hipStream_t stream) {
    if (n_per_frame < 8 || (n_per_frame & (n_per_frame - 1)) != 0) return hipErrorInvalidValue;  // 2^(K-1) states
    const size_t total8 = n_per_frame * (size_t)nframes / 8;
    const int blocks = (int)min((size_t)4096, (total8 + 255) / 256);
    hipLaunchKernelGGL(init_metrics_kernel, dim3(blocks), dim3(256), 0, stream, metrics, (unsigned)(n_per_frame - 1), total8, init_all,
                       init_start, start_state);
    return hipGetLastError();
}

}  // namespace vh
