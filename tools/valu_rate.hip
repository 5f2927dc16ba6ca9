// valu_rate.hip -- issue-rate microbenchmark for the integer VALU instructions the ACS kernels are built from.
// One line per instruction: shader cycles per wave64 instruction per SIMD at 1, 2 and 4 waves per SIMD
// (every CU busy).  Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFINE_KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned long long *out, int iters, unsigned seed) {      \
        unsigned r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13,   \
                 r6 = r0 * 17, r7 = r0 * 19;                                                                   \
        unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5555aaaau;                                    \
        asm volatile("s_mov_b64 vcc, 0x5555\ns_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");\
        unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();                                             \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int i = 0; i < iters; i++) {                                                                      \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3)   \
                             ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6)      \
                                 ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                         : "v"(b), "v"(c)                                                                      \
                         : "vcc", "s20", "s21");                                                                             \
        }                                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();                                             \
        unsigned s = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                    \
        if (s == 0x12345678u) out[1] = s;                                                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = rt1 - rt0; }                     \
    }

// operand numbering: %0..%7 accumulators, %8 = b, %9 = c
#define A_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_SUB_U32(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define A_MIN_U32(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define A_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3(i) "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define A_PK_ADD_U16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_PK_ADD_U16_CL(i) "v_pk_add_u16 %" #i ", %" #i ", %8 clamp\n"
#define A_PK_SUB_I16(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define A_PK_MAX_I16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_PK_MIN_U16(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_PK_ADD_F16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define A_PK_FMA_F16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define A_PK_MAD_U16(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define A_PK_LSHL(i) "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
#define A_ADD_U16(i) "v_add_u16 %" #i ", %" #i ", %8\n"
#define A_MAX_I16(i) "v_max_i16 %" #i ", %" #i ", %8\n"
#define A_ADD_SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0\n"
#define A_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_ADD_DPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define A_SUB_U16(i) "v_sub_u16 %" #i ", %" #i ", %8\n"
#define A_MIN_U16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define A_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_PK_SUB_U16_CL(i) "v_pk_sub_u16 %" #i ", %" #i ", %8 clamp\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define A_CMP(i) "v_cmp_gt_i32 vcc, %" #i ", %8\n"
#define A_SAD_U8(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PK_ADD_I16_CL(i) "v_pk_add_i16 %" #i ", %" #i ", %8 clamp\n"
#define A_MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_DOT4(i) "v_dot4_i32_i8 %" #i ", %8, %9, %" #i "\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 16\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_MIN3_U32(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD_LSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 1\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
// VOPC compares (write VCC) and the compare + select pair an unpacked ACS would use; v_writelane (ballot -> VGPR lane)
#define A_CMP_I16(i) "v_cmp_gt_i16 vcc, %" #i ", %8\n"
#define A_CMP_U16(i) "v_cmp_gt_u16 vcc, %" #i ", %8\n"
#define A_CMP_E64(i) "v_cmp_gt_i16_e64 s[20:21], %" #i ", %8\n"
#define A_CMP_SEL(i) "v_cmp_gt_i16 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define A_CMP_SEL_MOV(i) "v_cmp_gt_i16 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\ns_mov_b64 s[20:21], vcc\n"
#define A_WRITELANE(i) "v_writelane_b32 %" #i ", s20, 5\n"
#define A_SUBREV_U16(i) "v_subrev_u16 %" #i ", %" #i ", %8\n"
#define A_PK_SUB_U16(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n"
// one unpacked (VOP2 / VOPC, one state per VGPR) add-compare-select per new state: add, add, sub, cmp, cndmask
#define A_ACS_VOP2(i) "v_add_u16 %" #i ", %" #i ", %8\nv_add_u16 %" #i ", %" #i ", %9\nv_sub_u16 %" #i ", %" #i ", %8\nv_cmp_gt_i16 vcc, %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\ns_mov_b64 s[20:21], vcc\n"
// one packed ACS per two new states as acs_regs.hip issues it (modular family): add, sub, max, min, or, sub (+ shared ops)
#define A_ACS_PK(i) "v_pk_add_u16 %" #i ", %" #i ", %8\nv_pk_sub_i16 %" #i ", %" #i ", %9\nv_pk_max_i16 %" #i ", %" #i ", %8\nv_pk_min_u16 %" #i ", %" #i ", %9\nv_or_b32 %" #i ", %" #i ", %8\nv_pk_sub_i16 %" #i ", %" #i ", %9\n"
// dependent chains: every instruction reads the previous result
#define A_DEP_ADD(i) "v_add_u32 %0, %0, %8\n"
#define A_DEP_PK(i) "v_pk_add_u16 %0, %0, %8\n"
#define A_DEP2_PK(i) "v_pk_add_u16 %0, %0, %8\nv_pk_sub_i16 %1, %1, %9\n"
#define A_DEP4_PK(i) "v_pk_add_u16 %0, %0, %8\nv_pk_sub_i16 %1, %1, %9\nv_pk_max_i16 %2, %2, %8\nv_pk_min_u16 %3, %3, %9\n"

#define LIST(X)                                                                                                   \
    X(add_u32, A_ADD_U32) X(sub_u32, A_SUB_U32) X(max_i32, A_MAX_I32) X(min_u32, A_MIN_U32) X(xor_b32, A_XOR)     \
    X(or3_b32, A_OR3) X(add3_u32, A_ADD3) X(perm_b32, A_PERM) X(and_or_b32, A_ANDOR) X(lshl_or_b32, A_LSHLOR)     \
    X(bfi_b32, A_BFI) X(med3_i32, A_MED3) X(max3_i32, A_MAX3) X(min3_u32, A_MIN3_U32) X(add_lshl_u32, A_ADD_LSHL) \
    X(pk_add_u16, A_PK_ADD_U16) X(pk_add_u16_clamp, A_PK_ADD_U16_CL) X(pk_add_i16_clamp, A_PK_ADD_I16_CL)         \
    X(pk_sub_i16, A_PK_SUB_I16) X(pk_max_i16, A_PK_MAX_I16) X(pk_min_u16, A_PK_MIN_U16) X(pk_lshlrev_b16, A_PK_LSHL) \
    X(pk_mad_u16, A_PK_MAD_U16) X(pk_add_f16, A_PK_ADD_F16) X(pk_fma_f16, A_PK_FMA_F16)                           \
    X(add_u16, A_ADD_U16) X(max_i16, A_MAX_I16) X(add_u32_sdwa, A_ADD_SDWA) X(mov_dpp, A_DPP) X(add_u32_dpp, A_ADD_DPP) \
    X(cndmask, A_CNDMASK) X(cndmask_e64_sgpr, A_CNDMASK64) X(sub_u16, A_SUB_U16) X(min_u16, A_MIN_U16) X(or_b32, A_OR) X(and_b32, A_AND) X(lshlrev_b32, A_LSHL) X(pk_sub_u16_clamp, A_PK_SUB_U16_CL) X(bfe_u32, A_BFE) X(xad_u32, A_XAD) X(cmp_gt_i32, A_CMP) X(sad_u8, A_SAD_U8) X(fma_f32, A_FMA_F32) X(mad_u32_u24, A_MAD_U24) \
    X(dot4_i32_i8, A_DOT4) X(alignbit, A_ALIGNBIT) X(mov_b32, A_MOV) X(cmp_gt_i16, A_CMP_I16) X(cmp_gt_u16, A_CMP_U16) X(cmp_gt_i16_e64, A_CMP_E64) X(cmp_sel_pair, A_CMP_SEL) X(cmp_sel_smov, A_CMP_SEL_MOV) X(writelane, A_WRITELANE) X(subrev_u16, A_SUBREV_U16) X(pk_sub_u16, A_PK_SUB_U16) X(acs_vop2_5, A_ACS_VOP2) X(acs_pk_6, A_ACS_PK) X(dep1_add_u32, A_DEP_ADD) X(dep1_pk_add, A_DEP_PK) X(dep2_pk, A_DEP2_PK) X(dep4_pk, A_DEP4_PK)

#define X(NAME, ASM) DEFINE_KERNEL(NAME, ASM)
LIST(X)
#undef X

typedef void (*kern_t)(unsigned long long *, int, unsigned);
struct Entry { const char *name; kern_t fn; };

int main() {
    std::vector<Entry> es;
#define X(NAME, ASM) es.push_back({#NAME, k_##NAME});
    LIST(X)
#undef X
    unsigned long long *d_out;
    hipMalloc(&d_out, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 4000;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    printf("%-20s %14s %14s %14s %14s  (s_memtime ticks / wall ns per wave64 instruction per SIMD)\n", "instr", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
    for (auto &e : es) {
        printf("%-20s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            // 256-thread blocks = 4 waves = one per SIMD; wps blocks per CU
            hipLaunchKernelGGL(e.fn, dim3(cus * wps), dim3(256), 0, 0, d_out, 100, 1u);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(cus * wps), dim3(256), 0, 0, d_out, iters, 1u);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long dt3[3] = {0, 0, 0};
            hipMemcpy(dt3, d_out, 24, hipMemcpyDeviceToHost);
            const unsigned long long dt = dt3[0];
            const double mhz = dt3[2] ? (double)dt3[0] / ((double)dt3[2] / 100.0) : 0.0;  // s_memrealtime ticks at 100 MHz
            int mult = 1;
            std::string nm = e.name;
            if (nm == "dep2_pk") mult = 2;
            if (nm == "dep4_pk") mult = 4;
            if (nm == "cmp_sel_pair" || nm == "cmp_sel_smov") mult = 2;  // VALU instructions per macro
            if (nm == "acs_vop2_5") mult = 5;
            if (nm == "acs_pk_6") mult = 6;
            double ninstr = (double)iters * 32.0 * mult;
            double per = (double)dt / ninstr / wps;
            double ns_per = (double)ms * 1e6 / ninstr / wps;  // wall ns per instruction per SIMD
            printf(" %7.2f/%5.2fns", per, ns_per);
            if (wps == 8) printf(" [%4.0f MHz s_memtime, %.2f cyc@2.4GHz]", mhz, ns_per * 2.4);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
