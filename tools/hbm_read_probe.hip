// hbm_read_probe.hip -- how fast can ~1 GB that does NOT fit the 256 MB Infinity Cache be read?  (floor for the chainback
// kernels, which read the whole decision history once).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every block streams its own contiguous region (like one chainback wave walking its 1 MiB history), `per` uint4 per thread
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void read_regions(const u32x4 *__restrict__ in, unsigned *__restrict__ sink, long per) {
    const u32x4 *p = in + (long)blockIdx.x * per * blockDim.x + threadIdx.x;
    unsigned acc = 0;
#pragma unroll 8
    for (long i = 0; i < per; i++) {
        const u32x4 q = NT ? __builtin_nontemporal_load(p + i * blockDim.x) : p[i * blockDim.x];
        acc ^= q.x ^ q.y ^ q.z ^ q.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// one wave per block, global->LDS DMA in 12 KiB blocks, two in flight (the chainback_k7 pattern without the walk)
template <bool BACK, bool STORES, int AUX = 0>
__global__ __launch_bounds__(64) void read_dma(const unsigned char *__restrict__ in, unsigned *__restrict__ sink, long bytes_per_wave,
                                               unsigned char *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[3 * 12288];
    const unsigned char *src = in + (long)blockIdx.x * bytes_per_wave + threadIdx.x * 16;
    const long nblk = bytes_per_wave / 12288;
    auto issue = [&](int b, long n) {
#pragma unroll
        for (int c = 0; c < 12; c++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (BACK ? nblk - 1 - n : n) * 12288 + c * 1024),
                                             (__attribute__((address_space(3))) void *)(ring + b * 12288 + c * 1024), 16, 0, AUX);
    };
    issue(0, 0);
    if (nblk > 1) issue(1, 1);
    unsigned acc = 0;
    int buf = 0;
    for (long n = 0; n < nblk; n++) {
        if (n + 1 < nblk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (n + 2 < nblk) issue(buf == 0 ? 2 : buf - 1, n + 2);
        acc ^= reinterpret_cast<const unsigned *>(ring + buf * 12288)[threadIdx.x];
        if (STORES) {  // three bytes per block per lane, lanes 256 bytes apart (the decoded-byte stores of the chainback)
            unsigned char *o = out + ((long)blockIdx.x * 64 + threadIdx.x) * 256 + (n * 3) % 253;
            o[0] = (unsigned char)acc; o[1] = (unsigned char)(acc >> 8); o[2] = (unsigned char)(acc >> 16);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = 1024ull << 20;  // 1 GiB
    unsigned char *a; unsigned *sink;
    unsigned char *outb;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&outb, 4096 * 64 * 256));
    CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipEventRecord(e0));
        const int REP = 10;
        for (int i = 0; i < REP; i++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s %7.3f ms  %6.2f TB/s\n", name, ms / REP, bytes / (ms / REP * 1e-3) / 1e12);
    };
    for (int blocks : {1024, 2048, 4096, 16384}) {
        char nm[80]; snprintf(nm, sizeof nm, "vector loads, %d blocks x 256 thr, own region", blocks);
        const long per = (long)(bytes / 16 / ((size_t)blocks * 256));
        timeit(nm, [&] { hipLaunchKernelGGL((read_regions<false>), dim3(blocks), dim3(256), 0, 0, (const u32x4 *)a, sink, per); });
        snprintf(nm, sizeof nm, "  same, non-temporal loads");
        timeit(nm, [&] { hipLaunchKernelGGL((read_regions<true>), dim3(blocks), dim3(256), 0, 0, (const u32x4 *)a, sink, per); });
    }
    for (int waves : {1024, 2048, 4096}) {
        char nm[80]; snprintf(nm, sizeof nm, "LDS DMA, %d single-wave blocks, 12 KiB x 2 in flight", waves);
        const long bpw = (long)(bytes / waves) / 12288 * 12288;
        timeit(nm, [&] { hipLaunchKernelGGL((read_dma<false, false>), dim3(waves), dim3(64), 0, 0, a, sink, bpw, outb); });
        snprintf(nm, sizeof nm, "  same, non-temporal (aux = nt)");
        timeit(nm, [&] { hipLaunchKernelGGL((read_dma<false, false, 2>), dim3(waves), dim3(64), 0, 0, a, sink, bpw, outb); });
        snprintf(nm, sizeof nm, "  same, blocks walked from the top down");
        timeit(nm, [&] { hipLaunchKernelGGL((read_dma<true, false>), dim3(waves), dim3(64), 0, 0, a, sink, bpw, outb); });
        snprintf(nm, sizeof nm, "  same, top down + 3 byte stores per block per lane");
        timeit(nm, [&] { hipLaunchKernelGGL((read_dma<true, true>), dim3(waves), dim3(64), 0, 0, a, sink, bpw, outb); });
    }
    return 0;
}
