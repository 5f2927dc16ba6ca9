// k15_layout.h -- where acs_k15.hip puts a decision inside its 32-bit row words; shared by the K=15 kernels, the
// chainback kernels and the host-side row conversion used by the parity tests.
//
// A decision row of a frame is [4 words][128 threads]; thread t holds 64 packed registers rho = 0..63 (two positions
// each: low / high 16-bit field), word rho >> 4 covers 16 registers.  Two bit orders exist inside a word:
//   spiral615 (min trick of acs_regs.hip):   bit (rho & 15) + 16 * half
//   ka9q615  (sign bytes gathered by v_perm): bit 8 * (2 * (rho & 1) + half) + 7 - ((rho & 15) >> 1)
#pragma once
#if defined(__HIPCC__)
#define K15_HD __host__ __device__ __forceinline__
#else
#define K15_HD inline
#endif

namespace vh {

K15_HD unsigned k15_decision_bit(bool sign_bytes, unsigned rho, unsigned half) {
    return sign_bytes ? 8u * (2u * (rho & 1u) + half) + 7u - ((rho & 15u) >> 1) : (rho & 15u) + 16u * half;
}

}  // namespace vh
