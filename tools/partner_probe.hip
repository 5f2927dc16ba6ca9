// partner_probe.hip -- checks the six cross-lane partner fetches of acs_wave.hip (lane ^ 1, 2, 4, 8, 16, 32: DPP quad_perm,
// bank-masked DPP row shifts, v_permlane16_swap, v_permlane32_swap) against their definition on this GPU.
//   hipcc --offload-arch=gfx950 -O2 tools/partner_probe.hip -o tools/partner_probe && ./tools/partner_probe
#include <hip/hip_runtime.h>

#include <cstdio>

template <int B>
__device__ unsigned partner(unsigned x, unsigned lane) {
    if constexpr (B == 0) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);
    else if constexpr (B == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);
    else if constexpr (B == 2) {
        int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);
    } else if constexpr (B == 3) {
        int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x108, 0xf, 0x3, false);
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x118, 0xf, 0xc, false);
    } else if constexpr (B == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16u) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32u) ? r[0] : r[1];
    }
}

__global__ void probe(unsigned *out) {
    const unsigned lane = threadIdx.x, x = lane * 7u + 3u;
    out[0 * 64 + lane] = partner<0>(x, lane);
    out[1 * 64 + lane] = partner<1>(x, lane);
    out[2 * 64 + lane] = partner<2>(x, lane);
    out[3 * 64 + lane] = partner<3>(x, lane);
    out[4 * 64 + lane] = partner<4>(x, lane);
    out[5 * 64 + lane] = partner<5>(x, lane);
    const auto a = __builtin_amdgcn_permlane16_swap(x, x + 1000u, false, false);
    out[6 * 64 + lane] = a[0];
    out[7 * 64 + lane] = a[1];
    const auto b = __builtin_amdgcn_permlane32_swap(x, x + 1000u, false, false);
    out[8 * 64 + lane] = b[0];
    out[9 * 64 + lane] = b[1];
}

int main() {
    unsigned *d, h[10 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 6; b++) {
        int wrong = 0;
        for (unsigned l = 0; l < 64; l++) wrong += h[b * 64 + l] != ((l ^ (1u << b)) * 7u + 3u);
        printf("lane ^ %2d: %s (%d lanes wrong)\n", 1 << b, wrong ? "WRONG" : "ok", wrong);
        bad += wrong;
    }
    for (int r = 6; r < 10; r++) {
        printf("%s[%d]:", r < 8 ? "permlane16_swap(x, x+1000)" : "permlane32_swap(x, x+1000)", r & 1);
        for (unsigned l = 0; l < 64; l += 8) printf(" %u", h[r * 64 + l]);
        printf("\n");
    }
    return bad ? 1 : 0;
}
