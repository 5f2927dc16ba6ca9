// framegen.hip -- device + host synthetic frame generation and BER reduction.
//
// Device-side analogue of the reference's input path (src/util.h:8-12 random payload, :14-62 encode_data,
// :64-73 get_total_bit_errors): at 10^5..10^6 frames per batch, generating on the host and copying over
// PCIe would dominate wall time, so payload, encoder and channel run on the GPU; the host version below is
// bit-identical (same integer arithmetic, framegen.h) and is what the CPU tests and the oracle consume.
#include <hip/hip_runtime.h>

#include "framegen.h"
#include "kernels.h"

namespace vh {

struct GenArgs {
    int K, R;
    int poly[8];
    uint64_t seed, frame0;
    int nframes, payload_bytes;
    int amp_q16, noise_q12;
    unsigned char *payload;  // [nframes][payload_bytes]
    unsigned char *syms;     // [nframes][(8*payload_bytes+K-1)*R]
};

__global__ __launch_bounds__(256) void gen_payload_kernel(GenArgs a) {
    const size_t total = (size_t)a.nframes * a.payload_bytes;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / a.payload_bytes;
        const uint32_t b = (uint32_t)(i % a.payload_bytes);
        a.payload[i] = (unsigned char)fg_payload_byte(fg_frame_key(a.seed, a.frame0 + f), b);
    }
}

// one thread per (frame, trellis step): rebuild the K-bit shift register from the payload window, emit R symbols
__global__ __launch_bounds__(256) void gen_symbols_kernel(GenArgs a) {
    const int steps = 8 * a.payload_bytes + a.K - 1;
    const size_t total = (size_t)a.nframes * steps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / steps;
        const int t = (int)(i % steps);
        const uint64_t key = fg_frame_key(a.seed, a.frame0 + f);
        uint32_t sr = 0;
        // payload bit index t-k feeds shift-register bit k (newest bit = LSB)
        for (int k = 0; k < a.K; k++) {
            const int bi = t - k;
            if (bi >= 0 && bi < 8 * a.payload_bytes) {
                const unsigned byte = fg_payload_byte(key, (uint32_t)(bi >> 3));
                sr |= ((byte >> (7 - (bi & 7))) & 1u) << k;
            }
        }
        unsigned char *out = a.syms + (f * steps + t) * (size_t)a.R;
        for (int r = 0; r < a.R; r++) {
            const unsigned c = parity_u32(sr & (uint32_t)a.poly[r]);
            out[r] = (unsigned char)fg_symbol(c, a.amp_q16, a.noise_q12, fg_noise_c2(key, (uint32_t)(t * a.R + r)));
        }
    }
}

hipError_t launch_gen_frames(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes,
                             int payload_bytes, int amp_q16, int noise_q12, unsigned char *d_payload,
                             unsigned char *d_syms, hipStream_t stream) {
    GenArgs a;
    a.K = K;
    a.R = R;
    for (int r = 0; r < 8; r++) a.poly[r] = r < R ? poly[r] : 0;
    a.seed = seed;
    a.frame0 = frame0;
    a.nframes = nframes;
    a.payload_bytes = payload_bytes;
    a.amp_q16 = amp_q16;
    a.noise_q12 = noise_q12;
    a.payload = d_payload;
    a.syms = d_syms;
    if (d_payload) {
        const size_t total = (size_t)nframes * payload_bytes;
        const int blocks = (int)min((size_t)8192, (total + 255) / 256);
        hipLaunchKernelGGL(gen_payload_kernel, dim3(blocks), dim3(256), 0, stream, a);
    }
    if (d_syms) {
        const size_t total = (size_t)nframes * (8 * payload_bytes + K - 1);
        const int blocks = (int)min((size_t)16384, (total + 255) / 256);
        hipLaunchKernelGGL(gen_symbols_kernel, dim3(blocks), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void count_bit_errors_kernel(const unsigned char *__restrict__ x, const unsigned char *__restrict__ y,
                                                               size_t nbytes, unsigned long long *count) {
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (size_t)gridDim.x * blockDim.x)
        local += __popc((unsigned)(x[i] ^ y[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

hipError_t launch_count_bit_errors(const unsigned char *a, const unsigned char *b, size_t nbytes,
                                   unsigned long long *d_count, hipStream_t stream) {
    const int blocks = (int)min((size_t)4096, (nbytes + 255) / 256);
    hipLaunchKernelGGL(count_bit_errors_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, a, b, nbytes, d_count);
    return hipGetLastError();
}

// host twin of the two kernels above (pure integer code from framegen.h)
void gen_frames_host(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes, int payload_bytes,
                     int amp_q16, int noise_q12, unsigned char *payload, unsigned char *syms) {
    const int steps = 8 * payload_bytes + K - 1;
    const uint32_t kmask = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
    for (int f = 0; f < nframes; f++) {
        const uint64_t key = fg_frame_key(seed, frame0 + (uint64_t)f);
        uint32_t sr = 0;
        for (int t = 0; t < steps; t++) {
            unsigned bit = 0;
            if (t < 8 * payload_bytes) {
                const unsigned byte = fg_payload_byte(key, (uint32_t)(t >> 3));
                if (payload && (t & 7) == 0) payload[(size_t)f * payload_bytes + (t >> 3)] = (unsigned char)byte;
                bit = (byte >> (7 - (t & 7))) & 1u;
            }
            sr = ((sr << 1) | bit) & kmask;
            if (syms) {
                unsigned char *out = syms + ((size_t)f * steps + t) * (size_t)R;
                for (int r = 0; r < R; r++) {
                    const unsigned c = parity_u32(sr & (uint32_t)poly[r]);
                    out[r] = (unsigned char)fg_symbol(c, amp_q16, noise_q12, fg_noise_c2(key, (uint32_t)(t * R + r)));
                }
            }
        }
    }
}

}  // namespace vh
