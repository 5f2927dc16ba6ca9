// acs_k24.hip -- K=24 r=1/2 ACS update with the 8 388 608 path metrics tiled through HBM.
//
// Replaces update_viterbi224_blk_sse2 (ka9q_libfec_port/viterbi224_sse2.cpp:135-258).
//
// One trellis step = one grid-wide launch over the 4 194 304 butterflies.  A thread owns 8 consecutive
// butterflies j0..j0+7: it reads old[j0..j0+7] and old[H+j0..H+j0+7] as two 16-byte loads (both streams
// are unit-stride, so every wave access is a whole number of 128-byte lines even though the partner sits
// 8 MiB away), and writes new[2*j0 .. 2*j0+15] as two adjacent 16-byte stores plus 16 decision bits (one
// u16 of the 1 MiB bitmap row; a wave covers 128 contiguous bytes).  The branch table the CPU code keeps
// in 2 x 8 MiB (viterbi224_sse2.cpp:16-20,68-73) is recomputed from popcount parity: 0 bytes of traffic.
// Algorithmic HBM bytes per step: 16 MiB read + 16 MiB written + 1 MiB decisions + 2 symbol bytes.
//
// Renormalisation (viterbi224_sse2.cpp:226-246) depends on new[0] only and fires about once per several
// hundred steps, so steps are launched speculatively: the thread that owns state 0 raises a sticky flag
// when new[0] >= 25000, later step launches see the flag and return at once (leaving both metric buffers
// untouched), and the host replays from that row after running the min-reduce + wrapping-subtract kernels.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {

constexpr int K24 = 24;
constexpr unsigned N24 = 1u << (K24 - 1);
constexpr unsigned H24 = N24 / 2;

__device__ __forceinline__ int sat16(int x) { return max(-32768, min(x, 32767)); }

__global__ __launch_bounds__(256) void acs_k24_step_kernel(const int16_t *__restrict__ oldm, int16_t *__restrict__ newm,
                                                           unsigned char *__restrict__ row,
                                                           const unsigned char *__restrict__ syms, int step, int poly0,
                                                           int poly1, int *__restrict__ flags) {
    // K24F_PENDING = 1 + index of the step after which a renormalisation is due (0 = none).  A pending EARLIER
    // step means this launch is speculative garbage-in: return without touching either buffer; the host replays.
    // (A single word, so blocks of the launch that raises it can never mistake it for an earlier one.)
    const int pending = flags[K24F_PENDING];
    if (pending != 0 && pending != step + 1) return;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned j0 = gid * 8u;
    const int s[2] = {syms[2 * step], syms[2 * step + 1]};
    int T[4];
#pragma unroll
    for (unsigned c = 0; c < 4; c++) T[c] = Code224::bm(s, c);  // xor + add, viterbi224_sse2.cpp:159

    const uint4 lo = *reinterpret_cast<const uint4 *>(oldm + j0);
    const uint4 hi = *reinterpret_cast<const uint4 *>(oldm + H24 + j0);
    const unsigned lw[4] = {lo.x, lo.y, lo.z, lo.w};
    const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w};
    // class of butterfly j0 (j0 is a multiple of 8); the three low bits are folded in per element
    const unsigned base0 = parity_u32((2u * j0) & (unsigned)poly0), base1 = parity_u32((2u * j0) & (unsigned)poly1);
    unsigned outw[8];
    unsigned dbits = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int a = (int)(int16_t)(lw[e >> 1] >> (16 * (e & 1)));
        const int b = (int)(int16_t)(hw[e >> 1] >> (16 * (e & 1)));
        const unsigned c0 = base0 ^ parity_u32((2u * (unsigned)e) & (unsigned)poly0);
        const unsigned c1 = base1 ^ parity_u32((2u * (unsigned)e) & (unsigned)poly1);
        const int t = T[c0 | (c1 << 1)];
        const int tc = Code224::bm_comp - t;
        const int m0 = sat16(a + t), m1 = sat16(b + tc), m2 = sat16(a + tc), m3 = sat16(b + t);  // adds_epi16 :163-166
        const unsigned d0 = m0 > m1, d1 = m2 > m3;                                                 // cmpgt_epi16 :190-191
        const int v0 = min(m0, m1), v1 = min(m2, m3);                                              // min_epi16   :193-194
        outw[e] = ((unsigned)v0 & 0xffffu) | ((unsigned)v1 << 16);
        dbits |= (d0 << (2 * e)) | (d1 << (2 * e + 1));
    }
    // non-temporal: the dirty lines of a plain store would be written back once more when the kernel ends (acs_k24t.hip)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 *dst = reinterpret_cast<u32x4 *>(newm + 2u * j0);
    const u32x4 o0 = {outw[0], outw[1], outw[2], outw[3]}, o1 = {outw[4], outw[5], outw[6], outw[7]};
    __builtin_nontemporal_store(o0, dst);
    __builtin_nontemporal_store(o1, dst + 1);
    reinterpret_cast<unsigned short *>(row)[gid] = (unsigned short)dbits;
    if (gid == 0) {
        const int new0 = (int)(int16_t)(outw[0] & 0xffffu);
        if (new0 >= Code224::renorm_thr) {  // viterbi224_sse2.cpp:226
            flags[K24F_PENDING] = step + 1;
        }
    }
}

__global__ __launch_bounds__(256) void k24_min_kernel(const int16_t *__restrict__ m, int *__restrict__ slot) {
    int mn = 32767;
    const uint4 *p = reinterpret_cast<const uint4 *>(m);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N24 / 8; i += gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            mn = min(mn, (int)(int16_t)(w[k] & 0xffffu));
            mn = min(mn, (int)(int16_t)(w[k] >> 16));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mn = min(mn, __shfl_xor(mn, off));
    // one atomic per workgroup (8192 contending wave-atomics on one word took 95 us)
    __shared__ int wmin[4];
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) atomicMin(slot, min(min(wmin[0], wmin[1]), min(wmin[2], wmin[3])));
}

__global__ __launch_bounds__(256) void k24_sub_kernel(int16_t *__restrict__ m, const int *__restrict__ slot) {
    const int adjust = *slot + 32768;  // min - SHRT_MIN              viterbi224_sse2.cpp:240
    uint4 *p = reinterpret_cast<uint4 *>(m);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < N24 / 8; i += gridDim.x * blockDim.x) {
        uint4 v = p[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned lo = ((w[k] & 0xffffu) - (unsigned)adjust) & 0xffffu;  // sub_epi16 wraps  :245-246
            const unsigned hi = ((w[k] >> 16) - (unsigned)adjust) & 0xffffu;
            w[k] = lo | (hi << 16);
        }
        p[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// reset_slot < 0: the pending flag and all three minimum slots (start of a decode); else the flag and that slot
__global__ void k24_flags_reset_kernel(int *flags, int reset_slot) {
    flags[K24F_PENDING] = 0;
    for (int s = 0; s < 3; s++)
        if (reset_slot < 0 || reset_slot == s) flags[K24F_MIN + s] = 0x7fffffff;
}

hipError_t launch_k24_step(const int16_t *oldm, int16_t *newm, unsigned char *row, const unsigned char *d_syms, int step,
                           const int *poly, int *flags, hipStream_t stream) {
    hipLaunchKernelGGL(acs_k24_step_kernel, dim3(H24 / 8 / 256), dim3(256), 0, stream, oldm, newm, row, d_syms, step,
                       poly[0], poly[1], flags);
    return hipGetLastError();
}

hipError_t launch_k24_renorm(int16_t *m, int *flags, hipStream_t stream, int min_slot, int reset_slot) {
    hipLaunchKernelGGL(k24_min_kernel, dim3(1024), dim3(256), 0, stream, m, flags + K24F_MIN + min_slot);
    hipLaunchKernelGGL(k24_sub_kernel, dim3(2048), dim3(256), 0, stream, m, flags + K24F_MIN + min_slot);
    hipLaunchKernelGGL(k24_flags_reset_kernel, dim3(1), dim3(1), 0, stream, flags, reset_slot);
    return hipGetLastError();
}

hipError_t launch_k24_flags_reset(int *flags, hipStream_t stream) {
    hipLaunchKernelGGL(k24_flags_reset_kernel, dim3(1), dim3(1), 0, stream, flags, -1);
    return hipGetLastError();
}

}  // namespace vh
