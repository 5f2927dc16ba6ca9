// k24_copy_probe.hip -- what does a pass over the K=24 metric array cost with NO arithmetic?
// Same traffic as one acs_k24f 4-step pass: read 16 MiB (16 strided 8-byte vectors per thread), write 16 MiB the same
// way, plus ROWS MiB of decision words; and the same bytes as a plain linear copy.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef short i16x2 __attribute__((ext_vector_type(2)));
// WORK > 0: WORK rounds of 4 packed ops on every loaded register between the loads and the stores (32 registers x 4 x
// WORK wave instructions per thread; WORK = 4 is about the arithmetic of a 4-step ACS pass), VW2: 32 positions per thread
template <int BS, int ROWS, int WORK = 0>
__global__ __launch_bounds__(256) void strided_pass(const short *__restrict__ in, short *__restrict__ out, unsigned *__restrict__ rows) {
    const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned ulo = u & ((1u << (BS - 2)) - 1u), uhi = u >> (BS - 2);
    const unsigned pt = (uhi << (BS + 4)) | (ulo << 2);
    uint2 q[16];
#pragma unroll
    for (int v = 0; v < 16; v++) q[v] = *reinterpret_cast<const uint2 *>(in + (pt | ((unsigned)v << BS)));
    unsigned acc0 = 0, acc1 = 0;
#pragma unroll
    for (int w = 0; w < WORK; w++) {
#pragma unroll
        for (int v = 0; v < 16; v++) {
            i16x2 a = __builtin_bit_cast(i16x2, q[v].x), b = __builtin_bit_cast(i16x2, q[v].y);
            const i16x2 t = {(short)(w + 3), (short)(w + 5)};
            const i16x2 m0 = __builtin_elementwise_add_sat(a, t), m1 = __builtin_elementwise_add_sat(b, t);
            a = __builtin_elementwise_min(m0, m1);
            b = __builtin_elementwise_sub_sat(m0, m1);
            const i16x2 m2 = __builtin_elementwise_add_sat(a, b), m3 = __builtin_elementwise_sub_sat(b, t);
            a = __builtin_elementwise_min(m2, m3);
            b = __builtin_elementwise_add_sat(m2, t);
            q[v].x = __builtin_bit_cast(unsigned, a);
            q[v].y = __builtin_bit_cast(unsigned, b);
        }
    }
#pragma unroll
    for (int v = 0; v < 16; v++) { acc0 ^= q[v].x; acc1 ^= q[v].y; q[v].x += 1; }
#pragma unroll
    for (int s = 0; s < ROWS; s++) *reinterpret_cast<uint2 *>(rows + (size_t)s * (1u << 18) + (size_t)u * 2) = make_uint2(acc0 + s, acc1);
#pragma unroll
    for (int v = 0; v < 16; v++) *reinterpret_cast<uint2 *>(out + (pt | ((unsigned)v << BS))) = q[v];
}

template <int ROWS>
__global__ __launch_bounds__(256) void linear_pass(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint4 *__restrict__ rows, int per_thread) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int i = 0; i < per_thread; i++) {
        uint4 q = in[(size_t)i * nt + t];
        q.x += 1;
        out[(size_t)i * nt + t] = q;
    }
    // ROWS MiB of decisions
    for (unsigned i = t; i < (unsigned)ROWS * 65536u; i += nt) rows[i] = make_uint4(i, t, 0, 0);
}

int main() {
    short *a, *b; unsigned *rows;
    CK(hipMalloc(&a, 16u << 20)); CK(hipMalloc(&b, 16u << 20)); CK(hipMalloc(&rows, 8u << 20));
    CK(hipMemset(a, 1, 16u << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 200;
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 10; i++) launch(i);
        CK(hipEventRecord(e0));
        for (int i = 0; i < REP; i++) launch(i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.2f us per pass\n", name, ms * 1000 / REP);
    };
    timeit("strided BS=19, 64 pos/thread, 4 rows", [&](int i) { hipLaunchKernelGGL((strided_pass<19, 4>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    timeit("strided BS=11, 64 pos/thread, 4 rows", [&](int i) { hipLaunchKernelGGL((strided_pass<11, 4>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    timeit("strided BS=7,  64 pos/thread, 4 rows", [&](int i) { hipLaunchKernelGGL((strided_pass<7, 4>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    timeit("strided BS=19, 4 rows + 8 ops/reg x 4 rounds", [&](int i) { hipLaunchKernelGGL((strided_pass<19, 4, 4>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    timeit("strided BS=19, 4 rows + 8 ops/reg x 8 rounds", [&](int i) { hipLaunchKernelGGL((strided_pass<19, 4, 8>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    timeit("strided BS=19, 4 rows + 8 ops/reg x 16 rounds", [&](int i) { hipLaunchKernelGGL((strided_pass<19, 4, 16>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    timeit("strided BS=19, no rows", [&](int i) { hipLaunchKernelGGL((strided_pass<19, 0>), dim3(512), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, rows); });
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        char nm[64]; snprintf(nm, sizeof nm, "linear uint4, %d blocks, 4 rows", blocks);
        const int per = (int)((16u << 20) / 16 / (blocks * 256));
        timeit(nm, [&](int i) { hipLaunchKernelGGL((linear_pass<4>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)((i & 1) ? b : a), (uint4 *)((i & 1) ? a : b), (uint4 *)rows, per); });
    }
    timeit("empty-ish kernel (1 block)", [&](int i) { hipLaunchKernelGGL((linear_pass<0>), dim3(1), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, (uint4 *)rows, 0); });
    return 0;
}
