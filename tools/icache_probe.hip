// icache_probe.hip -- does straight-line code that a wave executes ONCE (the fully unrolled K=24 passes: 20-56 KB per
// kernel) issue as fast as a loop whose body stays in the instruction cache?  The same 64 Ki packed-VALU instructions per
// wave, as one straight line (512 KiB of code), as a 16 x 4096 loop (32 KiB body), as a 128 x 512 loop (4 KiB body);
// 1 and 2 waves per SIMD, every CU busy.  Build: hipcc --offload-arch=gfx950 -O2 tools/icache_probe.hip -o tools/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY8                                   \
    "v_pk_add_u16 %0, %0, %4\n"                 \
    "v_pk_sub_i16 %1, %1, %5\n"                 \
    "v_pk_max_i16 %2, %2, %4\n"                 \
    "v_pk_min_u16 %3, %3, %5\n"                 \
    "v_pk_add_u16 %0, %0, %5\n"                 \
    "v_pk_sub_i16 %1, %1, %4\n"                 \
    "v_pk_max_i16 %2, %2, %5\n"                 \
    "v_pk_min_u16 %3, %3, %4\n"

template <int REPT, int LOOPS>
__global__ __launch_bounds__(256) void k_probe(unsigned *out, unsigned seed) {
    unsigned r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, b = seed * 2654435761u, c = b ^ 0x5555aaaau;
    for (int i = 0; i < LOOPS; i++) {
        if constexpr (REPT == 8192)
            asm volatile(".rept 8192\n" BODY8 ".endr\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));
        else if constexpr (REPT == 512)
            asm volatile(".rept 512\n" BODY8 ".endr\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));
        else
            asm volatile(".rept 64\n" BODY8 ".endr\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c));
    }
    if ((r0 ^ r1 ^ r2 ^ r3) == 0x12345678u) out[0] = r0;
}

template <class F>
static void run(const char *name, F kern, int blocks, unsigned *d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int t = 0; t < 3; t++) {  // every launch starts on cold or at most launch-to-launch warm instruction caches
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s blocks %5d launch %d: %8.3f ms  %6.2f ns per wave64 instruction per SIMD\n", name, blocks, t, ms,
               ms * 1e6 / 65536.0 / (blocks / 256.0));
        if (ms < best) best = ms;
    }
}

int main() {
    unsigned *d;
    (void)hipMalloc(&d, 4096);
    for (int wps : {1, 2}) {
        run("straight line, 512 KiB of code", k_probe<8192, 1>, 256 * wps, d);
        run("16 x 32 KiB loop body", k_probe<512, 16>, 256 * wps, d);
        run("128 x 4 KiB loop body", k_probe<64, 128>, 256 * wps, d);
    }
    return 0;
}
