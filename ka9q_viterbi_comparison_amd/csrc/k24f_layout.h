// k24f_layout.h -- geometry of the fused K=24 passes (acs_k24f.hip), shared by the kernels, the chainback kernels and
// the host-side row conversion used by the parity tests.
//
// Phase phi = row mod 23 belongs to group g = phi/4 (phi < 16) or 4 (phi >= 16).  A thread of group g holds 16 vectors of
// VW positions: position p = thread part | (v << BS) | in-vector index, register rho = (v << KB) | k (KB = log2(VW)-1)
// holds in-vector elements 2k (low field) and 2k+1 (high field).  A decision row is [thread u][NR/16 words] with the
// decision of position p at bit k24f_decision_bit(rho, half) of word rho >> 4.
#pragma once
#if defined(__HIPCC__)
#define K24F_HD __host__ __device__ __forceinline__
#else
#define K24F_HD inline
#endif

namespace vh {

K24F_HD constexpr int k24f_group_of_phase(int phi) { return phi < 16 ? phi / 4 : 4; }
K24F_HD constexpr int k24f_bshift(int g) { return g < 4 ? 19 - 4 * g : 3; }  // lowest vector-index position bit
// log2(positions per vector): the strided groups keep K24F_LWS (1 -> 32 positions, 16 packed registers per thread,
// four waves per SIMD; 2 -> 64 positions, two waves per SIMD), group 4 keeps 8 contiguous positions per vector
constexpr int K24F_LWS = 1;
K24F_HD constexpr int k24f_lw(int g) { return g < 4 ? K24F_LWS : 3; }
K24F_HD constexpr int k24f_vw(int g) { return 1 << k24f_lw(g); }             // positions per vector

// bit of register rho's low (half = 0) / high (half = 1) field inside the 32-bit word of its 16-register group: the
// kernels gather the four sign bytes of a register pair (2i, 2i+1) with v_perm_b32 and shift them down by i
K24F_HD constexpr unsigned k24f_decision_bit(unsigned rho, unsigned half) {
    return 8u * (2u * (rho & 1u) + half) + 7u - ((rho & 15u) >> 1);
}

// thread part of the position for thread u of group g (vector-index and in-vector bits zero)
K24F_HD unsigned k24f_thread_base(int g, unsigned u) {
    const int BS = k24f_bshift(g), LW = k24f_lw(g);
    const unsigned ulo = u & ((1u << (BS - LW)) - 1u), uhi = u >> (BS - LW);
    return (uhi << (BS + 4)) | (ulo << LW);
}

// position p at phase phi -> 32-bit word index inside the 1 MiB row, bit inside the word
K24F_HD void k24f_locate(unsigned p, int phi, unsigned &word, unsigned &bit) {
    const int g = k24f_group_of_phase(phi);
    const int BS = k24f_bshift(g), LW = k24f_lw(g), KB = LW - 1;
    const unsigned h = p & 1u, k = (p >> 1) & ((1u << KB) - 1u), v = (p >> BS) & 15u;
    const unsigned ulo = (p >> LW) & ((1u << (BS - LW)) - 1u), uhi = p >> (BS + 4);
    const unsigned u = (uhi << (BS - LW)) | ulo;
    const unsigned rho = (v << KB) | k, wpt = (16u << LW) / 32u;  // words per thread = NR/16
    word = u * wpt + (rho >> 4);
    bit = k24f_decision_bit(rho, h);
}

}  // namespace vh
