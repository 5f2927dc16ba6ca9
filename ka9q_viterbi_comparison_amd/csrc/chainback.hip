// chainback.hip -- traceback over natural decision rows: one thread per frame.
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

#include "kernels.h"

namespace vh {

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

hipError_t launch_chainback_rows(const ChainbackRowsArgs &a, hipStream_t stream) {
    const int blocks = (a.nframes + 63) / 64;
    hipLaunchKernelGGL(chainback_rows_kernel, dim3(blocks), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace vh
