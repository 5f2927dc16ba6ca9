// k24t_layout.h -- geometry of the two-pass K=24 kernels (acs_k24t.hip), shared by the kernels, the chainback kernels and
// the host-side row conversion used by the parity tests.
//
// Rotating in-place trellis (as acs_k24f.hip): position p holds state rotl^phi(p) before the step of phase phi = row mod
// 23, and that step pairs position bit 22 - phi.  The 23 phases of a period are covered by TWO passes over the 16 MiB
// metric array (natural position order in HBM):
//      pass H  phases 0..8   pairs position bits 22..14.  Tile = all values of bits 22..14 x bits 5..0 (bits 13..6 = tile
//              id, 256 tiles of 2^15 positions): 512 runs of 128 contiguous bytes, 64 KiB.  512 threads x 64 positions.
//      pass L  phases 9..22  pairs position bits 13..0.   Tile = 2^14 contiguous positions (bits 22..14 = tile id, 512 tiles
//              of 32 KiB).  256 threads x 64 positions.
// Inside a pass the phases run in register groups; a thread of group G holds the positions
//      thread part | spos(G, rho) | half            (rho = packed register index, half = position bit 0 in every group)
// and groups are joined by a transpose through the tile's LDS image (32- and 64-bit accesses only):
//      H1 phases 0..4   register bits <-> position bits 22..18          thread = bits 17..14 (4) : bits 5..1 (5)
//      H2 phases 5..8   register bits <-> position bits 17..14, 1       thread = bits 22..18 (5) : bits 5..2 (4)
//      L1 phases 9..13  register bits <-> position bits 13..9           thread = bits 8..1
//      L2 phases 14..18 register bits <-> position bits 8..4            thread = bits 13..9 (5) : bits 3..1 (3)
//      L3 phases 19..22 register bits <-> position bits 13,12, 3..1     thread = bits 11..4       (phase 22 = half stage)
// (every pass keeps two waves per SIMD busy: 256 x 512 threads in pass H, 512 x 256 threads in pass L)
// Decision row r (1 MiB = 2^18 words) is stored as the kernels produce it: [tile][thread][NR/16 words], 32 bits per 16
// registers, bit k24t_decision_bit(rho, half) (sign bytes gathered by v_perm_b32, as k24f).  The decision of new state n
// at row r is at position rotr^((r+1) mod 23)(n).
#pragma once
#if defined(__HIPCC__)
#define K24T_HD __host__ __device__ __forceinline__
#else
#define K24T_HD inline
#endif

namespace vh {

enum { K24T_H1 = 0, K24T_H2 = 1, K24T_L1 = 2, K24T_L2 = 3, K24T_L3 = 4 };

K24T_HD constexpr int k24t_group_of_phase(int phi) { return phi < 5 ? K24T_H1 : phi < 9 ? K24T_H2 : phi < 14 ? K24T_L1 : phi < 19 ? K24T_L2 : K24T_L3; }
K24T_HD constexpr int k24t_first_phase(int g) { return g == K24T_H1 ? 0 : g == K24T_H2 ? 5 : g == K24T_L1 ? 9 : g == K24T_L2 ? 14 : 19; }
K24T_HD constexpr int k24t_nphases(int g) { return (g == K24T_H2 || g == K24T_L3) ? 4 : 5; }
K24T_HD constexpr int k24t_nr(int g) { return (void)g, 32; }                     // packed registers per thread
K24T_HD constexpr int k24t_threads(int g) { return g <= K24T_H2 ? 512 : 256; }   // threads per tile
// passes: 0 = H (phases 0..8), 1 = L (phases 9..22)
K24T_HD constexpr int k24t_pass_of_phase(int phi) { return phi < 9 ? 0 : 1; }
K24T_HD constexpr int k24t_pass_first(int pass) { return pass == 0 ? 0 : 9; }
K24T_HD constexpr int k24t_pass_nphases(int pass) { return pass == 0 ? 9 : 14; }

// position of register rho's low field, thread part zero
K24T_HD constexpr unsigned k24t_spos(int g, unsigned rho) {
    return g == K24T_H1 ? rho << 18
         : g == K24T_H2 ? ((rho >> 1) << 14) | ((rho & 1u) << 1)
         : g == K24T_L1 ? rho << 9
         : g == K24T_L2 ? rho << 4
                        : ((rho >> 3) << 12) | ((rho & 7u) << 1);
}
// register-index bit that holds position bit b in group g (b must be one of the group's register bits)
K24T_HD constexpr int k24t_regbit(int g, int b) {
    return g == K24T_H1 ? b - 18 : g == K24T_H2 ? (b == 1 ? 0 : b - 13) : g == K24T_L1 ? b - 9 : g == K24T_L2 ? b - 4 : (b <= 3 ? b - 1 : b - 9);
}

K24T_HD constexpr unsigned k24t_decision_bit(unsigned rho, unsigned half) {
    return 8u * (2u * (rho & 1u) + half) + 7u - ((rho & 15u) >> 1);
}

// thread part of the position (register and half bits zero) for thread `tid` of tile `tile` in group g
K24T_HD unsigned k24t_thread_base(int g, unsigned tile, unsigned tid) {
    switch (g) {
    case K24T_H1: return ((tid >> 5) << 14) | (tile << 6) | ((tid & 31u) << 1);
    case K24T_H2: return ((tid >> 4) << 18) | (tile << 6) | ((tid & 15u) << 2);
    case K24T_L1: return (tile << 14) | (tid << 1);
    case K24T_L2: return (tile << 14) | ((tid >> 3) << 9) | ((tid & 7u) << 1);
    default: return (tile << 14) | (tid << 4);
    }
}

// position p at phase phi -> 32-bit word index inside the 1 MiB row, bit inside the word
K24T_HD void k24t_locate(unsigned p, int phi, unsigned &word, unsigned &bit) {
    const int g = k24t_group_of_phase(phi);
    unsigned tile, tid, rho;
    switch (g) {
    case K24T_H1:
        tile = (p >> 6) & 255u; tid = (((p >> 14) & 15u) << 5) | ((p >> 1) & 31u); rho = p >> 18;
        break;
    case K24T_H2:
        tile = (p >> 6) & 255u; tid = ((p >> 18) << 4) | ((p >> 2) & 15u); rho = (((p >> 14) & 15u) << 1) | ((p >> 1) & 1u);
        break;
    case K24T_L1:
        tile = p >> 14; tid = (p >> 1) & 255u; rho = (p >> 9) & 31u;
        break;
    case K24T_L2:
        tile = p >> 14; tid = (((p >> 9) & 31u) << 3) | ((p >> 1) & 7u); rho = (p >> 4) & 31u;
        break;
    default:
        tile = p >> 14; tid = (p >> 4) & 255u; rho = (((p >> 12) & 3u) << 3) | ((p >> 1) & 7u);
        break;
    }
    const unsigned wpt = (unsigned)k24t_nr(g) / 16u;
    word = (tile * (unsigned)k24t_threads(g) + tid) * wpt + (rho >> 4);
    bit = k24t_decision_bit(rho, p & 1u);
}

}  // namespace vh
