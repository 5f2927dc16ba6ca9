// kernels.h -- launch interfaces between the C-ABI layer (viterbi_hip_api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace vh {

#if defined(__HIPCC__)
// A constant the compiler must keep in an SGPR instead of re-materialising it as a 32-bit literal at every use: VOP3
// instructions (v_and_or_b32, v_bitop3_b32, ...) cannot take literals on gfx950, so "(x & LITERAL) | acc" otherwise becomes
// v_and_b32 + v_or_b32; with the mask in an SGPR it is one v_and_or_b32.
template <unsigned V>
__device__ __forceinline__ unsigned sgpr_const() {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "i"(V));
    return r;
#else
    return V;
#endif
}
// Minimum over the 64 lanes of a wave with six DPP steps (quad_perm x2, row_half_mirror, row_mirror, row_bcast:15, row_bcast:31;
// the result lands in lane 63 and is read back as a scalar).  __shfl_xor compiles to ds_bpermute_b32: six dependent LDS round
// trips, which the spiral615 kernel paid on EVERY trellis step.
template <class T>
__device__ __forceinline__ T wave_min(T v) {
    static_assert(sizeof(T) == 4, "32-bit values");
#define VH_DPP_MIN(ctrl, rmask) v = min(v, (T)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    VH_DPP_MIN(0xB1, 0xf);   // quad_perm [1,0,3,2]
    VH_DPP_MIN(0x4E, 0xf);   // quad_perm [2,3,0,1]
    VH_DPP_MIN(0x141, 0xf);  // row_half_mirror
    VH_DPP_MIN(0x140, 0xf);  // row_mirror: every lane of a row of 16 holds the row's minimum
    VH_DPP_MIN(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    VH_DPP_MIN(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef VH_DPP_MIN
    return (T)__builtin_amdgcn_readlane((int)v, 63);
}
// the eight sign-byte masks 0x80808080 >> i used by the K=15 / K=24 decision gather (put_signs)
struct SignMasks {
    unsigned m[8];
    __device__ __forceinline__ SignMasks() {
        m[0] = sgpr_const<0x80808080u>(); m[1] = sgpr_const<0x40404040u>(); m[2] = sgpr_const<0x20202020u>(); m[3] = sgpr_const<0x10101010u>();
        m[4] = sgpr_const<0x08080808u>(); m[5] = sgpr_const<0x04040404u>(); m[6] = sgpr_const<0x02020202u>(); m[7] = sgpr_const<0x01010101u>();
    }
};
#endif

// ---------------------------------------------------------------- acs_lds.hip (K <= 15, any polynomial)
struct AcsLdsArgs {
    const unsigned char *syms;  // [nframes][sym_stride] offset-binary u8, step-major
    size_t sym_stride;          // bytes between frames
    int nsteps;                 // trellis steps to run
    int row0;                   // first decision row to write
    int cap_rows;               // rows allocated per frame
    int nframes;
    unsigned char *dec;         // natural rows: [nframes][cap_rows][N/8]
    int16_t *metrics;           // [nframes][N] persistent path metrics (natural units)
    int poly[8];
};
hipError_t launch_acs_lds(int code, const AcsLdsArgs &a, hipStream_t stream);
hipError_t launch_init_metrics(int16_t *metrics, size_t n_per_frame, int nframes, int init_all, int init_start,
                               unsigned start_state, hipStream_t stream);

// ---------------------------------------------------------------- acs_wave.hip (K = 7, one wave per frame: the latency geometry)
struct ChainbackRowsArgs;
bool wave_code_supported(int code);
hipError_t launch_acs_wave(int code, const AcsLdsArgs &a, hipStream_t stream);  // a.dec: [nframes][cap_rows] 64-bit position-ordered rows
hipError_t launch_chainback_wave(const ChainbackRowsArgs &a, hipStream_t stream);

// ---------------------------------------------------------------- acs_regs.hip (K <= 9, harness polynomials)
struct RegsLayout {  // decision layout [group][row][word][lane] produced by acs_regs_kernel
    int lb;      // log2(lanes per frame)
    int fpw;     // frames per wave (= per group)
    int nr;      // packed metric registers per lane
    int nrw;     // registers per decision word (16 -> 32-bit words, 8 -> 16-bit words)
    int dw;      // decision words per lane per row
    int wbytes;  // bytes per decision word
};
struct AcsRegsArgs {
    const unsigned char *syms;
    size_t sym_stride;
    int nsteps, row0, cap_rows, nframes;
    unsigned char *dec;   // [groups][cap_rows][dw][64] words
    int16_t *metrics;     // [nframes][N] canonical path metrics (natural units)
};
bool regs_poly_supported(int code, const int *poly);
bool regs_lanes_supported(int code, int lb);
RegsLayout regs_layout(int code, int lb);
hipError_t launch_acs_regs(int code, int lb, const AcsRegsArgs &a, hipStream_t stream);
// fused sliding-window decode (K <= 9, harness polynomials): depth / block / lanes-per-frame log2 per code
void windowed_params(int code, int *depth, int *block, int *lb);
hipError_t launch_decode_windowed(int code, const unsigned char *syms, size_t sym_stride, int nsteps, int nframes, unsigned char *data,
                                  size_t data_stride, unsigned nbits, hipStream_t stream);
struct ChainbackRegsArgs {
    const unsigned char *dec;
    RegsLayout lay;
    int cap_rows, rows_written, nframes;
    unsigned char *data;
    size_t data_stride;
    unsigned nbits, endstate;
    int K;
};
hipError_t launch_chainback_regs(const ChainbackRegsArgs &a, hipStream_t stream);

// ---------------------------------------------------------------- acs_k15.hip (K = 15, harness polynomials)
struct AcsK15Args {
    const unsigned char *syms;
    size_t sym_stride;
    int nsteps, row0, cap_rows, nframes;
    unsigned char *dec;   // [nframes][cap_rows][4][128] words
    int16_t *metrics;     // [nframes][16384] canonical path metrics
};
struct ChainbackRowsArgs;
bool k15_poly_supported(const int *poly);
// fused sliding-window decode for K=15 (harness polynomials): decision ring of the resident workgroups in global memory
size_t windowed_k15_ring_bytes(int nframes);
void windowed_k15_params(int *depth, int *block);
hipError_t launch_decode_windowed_k15(bool spiral, const unsigned char *syms, size_t sym_stride, int nsteps, int nframes, unsigned char *data,
                                      size_t data_stride, unsigned nbits, unsigned *ring, hipStream_t stream);
hipError_t launch_acs_k15(const AcsK15Args &a, bool spiral, hipStream_t stream);  // spiral: the spiral615 arithmetic

// ---------------------------------------------------------------- chainback.hip (natural rows)
struct ChainbackRowsArgs {
    const unsigned char *dec;  // natural rows [nframes][cap_rows][N/8]
    int cap_rows;
    int rows_written;          // rows beyond this read as zero
    int nframes;
    unsigned char *data;       // [nframes][data_stride]
    size_t data_stride;
    unsigned nbits;
    unsigned endstate;
    int K;
    int k224;                  // chainback_viterbi224_sse2 semantics (no tail skip, emits state&1)
    int k15_sign_bytes = 0;    // acs_k15 rows in the ka9q615 bit order (k15_layout.h)
    // chainback_spec.hip, segment-parallel form (few frames): nseg > 1 segments of seg_bits decoded bits (a multiple of 8) per
    // frame, each warmed up over seg_ovl rows; seg_scratch: (2 + 2 * nseg) * nframes words, the first 2 * nframes zero before the first launch
    int nseg = 0, seg_bits = 0, seg_ovl = 0;
    unsigned *seg_scratch = nullptr;
};
hipError_t launch_chainback_rows(const ChainbackRowsArgs &a, hipStream_t stream);

// ---------------------------------------------------------------- acs_k24.hip (K = 24, metrics in HBM)
enum { K24F_PENDING = 0, K24F_MIN = 1, K24F_COUNT = 4 };  // device flag words: pending row + 1, three rotating minimum slots
// Progress word of the multi-step K=24 passes, in pinned host memory: every pass stores (seq << 32) | pending flag when its
// flag-owning thread is done, so the host follows the stream by polling one word instead of waiting on events (an event
// record / wait per batch of passes put a ~8 us bubble into the stream each time).
struct K24Report {
    unsigned long long *word;
    unsigned seq;
};
#if defined(__HIPCC__)
__device__ __forceinline__ void k24_report(const K24Report &r, int pending) {
    if (r.word) __hip_atomic_store(r.word, ((unsigned long long)r.seq << 32) | (unsigned)pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
hipError_t launch_k24_step(const int16_t *oldm, int16_t *newm, unsigned char *row, const unsigned char *d_syms, int step,
                           const int *poly, int *flags, hipStream_t stream);
// min-reduce into slot min_slot, subtract, clear the pending flag and slot reset_slot
hipError_t launch_k24_renorm(int16_t *m, int *flags, hipStream_t stream, int min_slot = 0, int reset_slot = 0);
hipError_t launch_k24_flags_reset(int *flags, hipStream_t stream);

// ---------------------------------------------------------------- acs_k24f.hip (K = 24, 4/7 steps per pass)
bool k24f_poly_supported(const int *poly);
hipError_t launch_k24f_pass(int g, const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                            int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, hipStream_t stream);

// ---------------------------------------------------------------- acs_k24t.hip (K = 24, two passes per 23 steps)
bool k24t_poly_supported(const int *poly);
hipError_t launch_k24t_pass(int pass, const int16_t *oldm, int16_t *newm, unsigned char *rows, const unsigned char *syms,
                            int rel_row0, int s_lo, int s_hi, int *flags, K24Report mirror, hipStream_t stream, bool nt_stores,
                            int ctl);  // ctl: renormalisation duties of this pass (acs_k24t.hip), 0 for an ordinary pass

// ---------------------------------------------------------------- jit.hip (fast kernels for other polynomials)
bool jit_enabled();  // VHIP_JIT=0 turns the runtime specialisation off
int jit_sources_match(std::string *why);  // 1 / 0 / -1 (unreadable): the sources next to the library vs the fingerprint baked into it
// defs: one compiler argument each (no shell is involved)
bool jit_function(const char *src, const std::vector<std::string> &defs, const char *kname, hipFunction_t *fn, std::string *err);
std::string jit_poly_define(const int *poly, int n);

// ---------------------------------------------------------------- chainback_spec.hip (K = 15 / 24, one wave per frame)
enum { CB_LAY_NATURAL = 0, CB_LAY_K15 = 1, CB_LAY_K24F = 2, CB_LAY_K15_SIGN_BYTES = 3, CB_LAY_K24T = 4 };
hipError_t launch_chainback_spec(int layout, const ChainbackRowsArgs &a, hipStream_t stream);

// ---------------------------------------------------------------- framegen.hip
hipError_t launch_gen_frames(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes,
                             int payload_bytes, int amp_q16, int noise_q12, unsigned char *d_payload,
                             unsigned char *d_syms, hipStream_t stream);
hipError_t launch_count_bit_errors(const unsigned char *a, const unsigned char *b, size_t nbytes,
                                   unsigned long long *d_count, hipStream_t stream);

}  // namespace vh
