// sstore_rate.hip -- can wave-uniform decision ballots (SGPR pairs from v_cmp) leave through the scalar store path?
// Each wave writes NB bytes of its own region with s_store_dwordx4 (16 B per instruction), optionally beside a VALU
// stream of `valu_per_store` v_add_u16 per store (the ratio an unpacked ACS would have: 10 VALU per 2 states = 16 B).
// Build: hipcc --offload-arch=gfx950 -O2 tools/sstore_rate.hip -o /tmp/sstore_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VALU>
__global__ __launch_bounds__(256) void k_sstore(unsigned *out, int iters, unsigned seed) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long base = (unsigned long long)out + (unsigned long long)wave * (unsigned long long)iters * 64ull;
    unsigned r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    for (int i = 0; i < iters; i++) {
        // 4 stores of 16 B = one 64-byte line per iteration
        asm volatile(
            "s_mov_b32 s20, %4\n s_mov_b32 s21, %5\n s_mov_b32 s24, %6\n s_mov_b32 s25, %6\n s_mov_b32 s26, %6\n s_mov_b32 s27, %6\n"
            "s_store_dwordx4 s[24:27], s[20:21], 0x0\n"
            "s_store_dwordx4 s[24:27], s[20:21], 0x10\n"
            "s_store_dwordx4 s[24:27], s[20:21], 0x20\n"
            "s_store_dwordx4 s[24:27], s[20:21], 0x30\n"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)
            : "s"(lo), "s"(hi), "s"(i)
            : "s20", "s21", "s24", "s25", "s26", "s27", "memory");
#pragma unroll
        for (int v = 0; v < VALU; v++) {
            asm volatile("v_add_u16 %0, %0, %1\nv_add_u16 %1, %1, %2\nv_add_u16 %2, %2, %3\nv_add_u16 %3, %3, %0\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        }
        lo += 64;  // no carry: regions are 64-byte aligned and < 4 GiB apart in the low word for the sizes used here
    }
    asm volatile("s_waitcnt lgkmcnt(0)\ns_dcache_wb\ns_waitcnt lgkmcnt(0)" ::: "memory");
    if ((r0 ^ r1 ^ r2 ^ r3) == 0x12345678u) out[0] = r0;
}

template <int VALU>
__global__ __launch_bounds__(256) void k_vstore(unsigned *out, int iters, unsigned seed) {
    // the same bytes through the vector path: one 64-lane dword store = 256 B per 4 iterations' worth of ballots
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    unsigned *dst = out + (size_t)wave * iters * 16 + lane;
    unsigned r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7;
    for (int i = 0; i < iters; i += 4) {
        __builtin_nontemporal_store(r0, dst + (size_t)i * 16);
#pragma unroll
        for (int v = 0; v < 4 * VALU; v++) {
            asm volatile("v_add_u16 %0, %0, %1\nv_add_u16 %1, %1, %2\nv_add_u16 %2, %2, %3\nv_add_u16 %3, %3, %0\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        }
    }
    if ((r0 ^ r1 ^ r2 ^ r3) == 0x12345678u) out[0] = r0;
}

template <class F>
static void run(const char *name, F kern, int valu, int blocks, int iters, unsigned *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 64, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 64.0;
    printf("%-10s valu/16B=%2d blocks=%5d: %8.3f ms  %8.1f GB/s  %6.2f ns per 64-B line per wave  err=%s\n", name, valu * 4 / 4, blocks, ms,
           bytes / ms / 1e6, ms * 1e6 / iters, hipGetErrorString(hipGetLastError()));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 2048;
    unsigned *d;
    hipMalloc(&d, (size_t)cus * 8 * 4 * iters * 64 + 4096);
    for (int wps : {1, 2, 4}) {
        printf("-- %d wave(s) per SIMD\n", wps);
        run("s_store", k_sstore<0>, 0, cus * wps, iters, d);
        run("s_store", k_sstore<1>, 4, cus * wps, iters, d);
        run("s_store", k_sstore<4>, 16, cus * wps, iters, d);
        run("s_store", k_sstore<10>, 40, cus * wps, iters, d);
        run("v_store", k_vstore<0>, 0, cus * wps, iters, d);
        run("v_store", k_vstore<4>, 16, cus * wps, iters, d);
        run("v_store", k_vstore<10>, 40, cus * wps, iters, d);
    }
    return 0;
}
