// VALU issue-rate micro-benchmark for gfx950 (MI355X): how many shader cycles does one SIMD need per wave64
// instruction, per opcode, with 1 / 2 / 4 / 8 waves resident on the SIMD?
//
// Why: VERDICT r01 weak #2.  Round 1 priced the kernels against "4 cycles per wave64 VALU instruction" (taken from
// SQ_ACTIVE_INST_VALU*4 / SQ_INSTS_VALU); MI355X_MICROARCH.md says CDNA4 SIMDs are 32 lanes wide (v_fma_f32 wave64 =
// 2 cycles) and that the SQ_ACTIVE_INST_* counters tick in quad-cycles.  This program measures the rates directly.
//
// Method: every kernel issues ITERS x 128 instructions of ONE opcode per wave from a loop whose body is 16 blocks of 8
// instructions on 8 different destination registers (independent chains, dependency distance 8) -- or, for the *_dep
// variants, on one register (dependent chain = latency).  The wave brackets the loop with s_memtime (shader-clock
// ticks per the guide) and s_memrealtime (100 MHz constant clock) and records HW_ID / XCC_ID, so the host can check
// where the waves really ran.  Workgroups are 256 threads (4 waves, one per SIMD) and carry a static LDS allocation of
// 160 KiB / W so that exactly W workgroups fit on a CU; the grid is CUs x W workgroups, i.e. the chip is exactly full
// with W waves on every SIMD.  Reported per (opcode, W):
//     cyc/inst/SIMD = median over waves of (s_memtime delta) / (instructions per wave x waves on that SIMD)
// and the same figure from the host-side hipEvent wall time x the measured shader clock.
//
// Build:  hipcc --offload-arch=gfx950 -O2 -o profiles/valu_issue_ubench profiles/valu_issue_ubench.hip
// Run  :  profiles/valu_issue_ubench [iters] > profiles/r02_valu_issue_ubench.json     (on the MI355X box)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

struct Rec {
    uint64_t t0, t1;    // s_memtime
    uint64_t r0, r1;    // s_memrealtime (100 MHz)
    uint32_t hw_id, xcc_id;
    uint32_t sink, pad;
};

__device__ __forceinline__ uint64_t memtime() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint64_t memrealtime() {
    uint64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// ---- opcode table: S(n) expands to ONE instruction writing destination operand %n; %8, %9 are two source VGPRs,
//      %10 is a third.  The *_dep forms always write %0.
#define X8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define D8(S) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0)
#define X16(B) B B B B B B B B B B B B B B B B

// every opcode is its own function-like macro (n = destination operand index):
#define I_v_mov_b32(n) "v_mov_b32 %" #n ", %8\n"
#define I_v_pk_mov_b32(n) "v_pk_mov_b32 %[p" #n "], %[q], %[q]\n"
#define I_v_add_u32(n) "v_add_u32 %" #n ", %8, %" #n "\n"
#define I_v_sub_u32_clamp(n) "v_sub_u32_e64 %" #n ", %" #n ", %8 clamp\n"
#define I_v_add3_u32(n) "v_add3_u32 %" #n ", %8, %9, %" #n "\n"
#define I_v_lshl_add_u32(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define I_v_and_b32(n) "v_and_b32 %" #n ", %8, %" #n "\n"
#define I_v_and_or_b32(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_v_lshlrev_b32(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_v_lshrrev_b32(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define I_v_ashrrev_i32(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define I_v_bfe_u32(n) "v_bfe_u32 %" #n ", %" #n ", 3, 9\n"
#define I_v_bfi_b32(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n"
#define I_v_alignbit_b32(n) "v_alignbit_b32 %" #n ", %8, %" #n ", 16\n"
#define I_v_perm_b32(n) "v_perm_b32 %" #n ", %8, %" #n ", %9\n"
#define I_v_max_u32(n) "v_max_u32 %" #n ", %8, %" #n "\n"
#define I_v_min_i32(n) "v_min_i32 %" #n ", %8, %" #n "\n"
#define I_v_med3_i32(n) "v_med3_i32 %" #n ", %8, %9, %" #n "\n"
#define I_v_cndmask_b32(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define I_v_cmp_lt_u32(n) "v_cmp_lt_u32 vcc, %8, %" #n "\n"
#define I_v_cmp_lt_u32_sgpr(n) "v_cmp_lt_u32 s[20:21], %8, %" #n "\n"
#define I_v_mul_lo_u32(n) "v_mul_lo_u32 %" #n ", %8, %" #n "\n"
#define I_v_mul_hi_u32(n) "v_mul_hi_u32 %" #n ", %8, %" #n "\n"
#define I_v_mul_u32_u24(n) "v_mul_u32_u24 %" #n ", %8, %" #n "\n"
#define I_v_mul_i32_i24(n) "v_mul_i32_i24 %" #n ", %8, %" #n "\n"
#define I_v_mad_u32_u24(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n"
#define I_v_mad_i32_i24(n) "v_mad_i32_i24 %" #n ", %8, %9, %" #n "\n"
#define I_v_mad_u64_u32(n) "v_mad_u64_u32 %[p" #n "], vcc, %8, %9, %[p" #n "]\n"
#define I_v_dot2_i32_i16_acc(n) "v_dot2_i32_i16 %" #n ", %8, %9, %" #n "\n"
#define I_v_dot2_i32_i16_c0(n) "v_dot2_i32_i16 %" #n ", %8, %" #n ", 0\n"
#define I_v_dot4_i32_i8(n) "v_dot4_i32_i8 %" #n ", %8, %9, %" #n "\n"
#define I_v_pk_add_u16(n) "v_pk_add_u16 %" #n ", %8, %" #n "\n"
#define I_v_pk_sub_i16(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define I_v_pk_mad_u16(n) "v_pk_mad_u16 %" #n ", %8, %9, %" #n "\n"
#define I_v_pk_mul_lo_u16(n) "v_pk_mul_lo_u16 %" #n ", %8, %" #n "\n"
#define I_v_pk_lshrrev_b16(n) "v_pk_lshrrev_b16 %" #n ", 1, %" #n " op_sel_hi:[0,1]\n"
#define I_v_pk_ashrrev_i16(n) "v_pk_ashrrev_i16 %" #n ", 1, %" #n " op_sel_hi:[0,1]\n"
#define I_v_pk_max_i16(n) "v_pk_max_i16 %" #n ", %8, %" #n "\n"
#define I_v_add_u16_sdwa(n) "v_add_u32_sdwa %" #n ", %8, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define I_v_add_u32_dpp(n) "v_add_u32_dpp %" #n ", %8, %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_v_mov_b32_dpp(n) "v_mov_b32_dpp %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_v_fma_f32(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I_v_add_f32(n) "v_add_f32 %" #n ", %8, %" #n "\n"
#define I_v_mul_f32(n) "v_mul_f32 %" #n ", %8, %" #n "\n"
#define I_v_pk_fma_f32(n) "v_pk_fma_f32 %[p" #n "], %[q], %[q], %[p" #n "]\n"
#define I_v_pk_mul_f32(n) "v_pk_mul_f32 %[p" #n "], %[q], %[p" #n "]\n"
#define I_v_sqrt_f32(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_v_rsq_f32(n) "v_rsq_f32 %" #n ", %" #n "\n"
#define I_v_rcp_f32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_v_log_f32(n) "v_log_f32 %" #n ", %" #n "\n"
#define I_v_exp_f32(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_v_cvt_f32_u32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_v_cvt_u32_f32(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I_v_cvt_f32_i32(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_v_cvt_pk_u16_u32(n) "v_cvt_pk_u16_u32 %" #n ", %8, %" #n "\n"
#define I_v_sad_u16(n) "v_sad_u16 %" #n ", %8, %9, %" #n "\n"
#define I_v_sad_u32(n) "v_sad_u32 %" #n ", %8, %9, %" #n "\n"
#define I_v_readlane(n) "v_readlane_b32 s20, %" #n ", 3\n"
#define I_v_readfirstlane(n) "v_readfirstlane_b32 s20, %" #n "\n"
#define I_v_ffbh_u32(n) "v_ffbh_u32 %" #n ", %" #n "\n"
#define I_v_xad_u32(n) "v_xad_u32 %" #n ", %8, %9, %" #n "\n"
#define I_v_add_co_u32(n) "v_add_co_u32 %" #n ", vcc, %8, %" #n "\n"
#define I_v_addc_co_u32(n) "v_addc_co_u32 %" #n ", vcc, %8, %" #n ", vcc\n"

// ---- second batch (r02b): more candidates for cheap replacements, and the v_cndmask_b32 anomaly ----
#define I_v_or_b32(n) "v_or_b32 %" #n ", %8, %" #n "\n"
#define I_v_xor_b32(n) "v_xor_b32 %" #n ", %8, %" #n "\n"
#define I_v_not_b32(n) "v_not_b32 %" #n ", %" #n "\n"
#define I_v_sub_u32(n) "v_sub_u32 %" #n ", %8, %" #n "\n"
#define I_v_subrev_u32(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
#define I_v_add_u32_e64(n) "v_add_u32_e64 %" #n ", %8, %" #n "\n"
#define I_v_add_u32_lit(n) "v_add_u32 %" #n ", 0x12345, %" #n "\n"
#define I_v_add_u32_sgpr(n) "v_add_u32 %" #n ", s22, %" #n "\n"
#define I_v_lshlrev_b32_reg(n) "v_lshlrev_b32 %" #n ", %8, %" #n "\n"
#define I_v_lshrrev_b32_reg(n) "v_lshrrev_b32 %" #n ", %8, %" #n "\n"
#define I_v_lshrrev_b32_16(n) "v_lshrrev_b32 %" #n ", 16, %" #n "\n"
#define I_v_ashrrev_i32_16(n) "v_ashrrev_i32 %" #n ", 16, %" #n "\n"
#define I_v_min_f32(n) "v_min_f32 %" #n ", %8, %" #n "\n"
#define I_v_max_f32(n) "v_max_f32 %" #n ", %8, %" #n "\n"
#define I_v_sub_f32(n) "v_sub_f32 %" #n ", %8, %" #n "\n"
#define I_v_fmac_f32(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_v_fma_f32_neg(n) "v_fma_f32 %" #n ", -%8, %9, %" #n "\n"
#define I_v_mul_f32_lit(n) "v_mul_f32 %" #n ", 0x41200000, %" #n "\n"
#define I_v_cmp_lt_f32(n) "v_cmp_lt_f32 vcc, %8, %" #n "\n"
#define I_v_cmp_eq_u32(n) "v_cmp_eq_u32 vcc, %8, %" #n "\n"
#define I_v_cmp_lt_i32(n) "v_cmp_lt_i32 vcc, %8, %" #n "\n"
#define I_v_cmp_class_f32(n) "v_cmp_class_f32 vcc, %" #n ", %8\n"
#define I_v_floor_f32(n) "v_floor_f32 %" #n ", %" #n "\n"
#define I_v_trunc_f32(n) "v_trunc_f32 %" #n ", %" #n "\n"
#define I_v_rndne_f32(n) "v_rndne_f32 %" #n ", %" #n "\n"
#define I_v_fract_f32(n) "v_fract_f32 %" #n ", %" #n "\n"
#define I_v_ldexp_f32(n) "v_ldexp_f32 %" #n ", %" #n ", %8\n"
#define I_v_cvt_i32_f32(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_v_cvt_f32_ubyte0(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define I_v_cvt_f16_f32(n) "v_cvt_f16_f32 %" #n ", %" #n "\n"
#define I_v_add_u16(n) "v_add_u16 %" #n ", %8, %" #n "\n"
#define I_v_sub_u16(n) "v_sub_u16 %" #n ", %8, %" #n "\n"
#define I_v_mul_lo_u16(n) "v_mul_lo_u16 %" #n ", %8, %" #n "\n"
#define I_v_lshrrev_b16(n) "v_lshrrev_b16 %" #n ", 1, %" #n "\n"
#define I_v_ashrrev_i16(n) "v_ashrrev_i16 %" #n ", 1, %" #n "\n"
#define I_v_max_i16(n) "v_max_i16 %" #n ", %8, %" #n "\n"
#define I_v_mad_u16(n) "v_mad_u16 %" #n ", %8, %9, %" #n "\n"
#define I_v_add_f16(n) "v_add_f16 %" #n ", %8, %" #n "\n"
#define I_v_fma_f16(n) "v_fma_f16 %" #n ", %8, %9, %" #n "\n"
#define I_v_pk_add_f16(n) "v_pk_add_f16 %" #n ", %8, %" #n "\n"
#define I_v_pk_fma_f16(n) "v_pk_fma_f16 %" #n ", %8, %9, %" #n "\n"
#define I_v_pk_add_i16(n) "v_pk_add_i16 %" #n ", %8, %" #n "\n"
#define I_v_pk_lshlrev_b16(n) "v_pk_lshlrev_b16 %" #n ", 1, %" #n " op_sel_hi:[0,1]\n"
#define I_v_pk_min_u16(n) "v_pk_min_u16 %" #n ", %8, %" #n "\n"
#define I_v_dot2_u32_u16(n) "v_dot2_u32_u16 %" #n ", %8, %9, %" #n "\n"
#define I_v_dot2c_i32_i16(n) "v_dot2c_i32_i16 %" #n ", %8, %9\n"
#define I_v_dot2_f32_f16(n) "v_dot2_f32_f16 %" #n ", %8, %9, %" #n "\n"
#define I_v_dot2c_f32_bf16(n) "v_dot2c_f32_bf16 %" #n ", %8, %9\n"
#define I_v_mad_u32_u16(n) "v_mad_u32_u16 %" #n ", %8, %9, %" #n "\n"
#define I_v_mad_i32_i16(n) "v_mad_i32_i16 %" #n ", %8, %9, %" #n "\n"
#define I_v_mul_hi_u32_u24(n) "v_mul_hi_u32_u24 %" #n ", %8, %" #n "\n"
#define I_v_lshl_or_b32(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, %8\n"
#define I_v_or3_b32(n) "v_or3_b32 %" #n ", %8, %9, %" #n "\n"
#define I_v_add_lshl_u32(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 1\n"
#define I_v_sub_co_u32(n) "v_sub_co_u32 %" #n ", vcc, %8, %" #n "\n"
#define I_v_bcnt_u32_b32(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n"
#define I_v_mbcnt_lo(n) "v_mbcnt_lo_u32_b32 %" #n ", %8, %" #n "\n"
#define I_v_bfrev_b32(n) "v_bfrev_b32 %" #n ", %" #n "\n"
#define I_v_swap_b32(n) "v_swap_b32 %" #n ", %8\n"
#define I_v_accvgpr_write(n) "v_accvgpr_write_b32 a" #n ", %" #n "\n"
#define I_v_accvgpr_read(n) "v_accvgpr_read_b32 %" #n ", a" #n "\n"
#define I_v_mov_b32_sdwa(n) "v_mov_b32_sdwa %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define I_v_mov_b32_lit(n) "v_mov_b32 %" #n ", 0x12345\n"
#define I_v_mov_b64(n) "v_mov_b64 %[p" #n "], %[q]\n"
#define I_v_lshlrev_b64(n) "v_lshlrev_b64 %[p" #n "], 1, %[p" #n "]\n"
#define I_v_lshl_add_u64(n) "v_lshl_add_u64 %[p" #n "], %[p" #n "], 1, %[q]\n"
#define I_v_pk_add_f32(n) "v_pk_add_f32 %[p" #n "], %[q], %[p" #n "]\n"
#define I_v_pk_add_u32_na(n) "v_pk_mul_f32 %[p" #n "], %[q], %[q]\n"
// v_cndmask_b32 forms.  vcc0/vcc1: vcc set to 0 / all ones before the loop; sg: mask in s[20:21] (VOP3);
// cmp: one v_cmp writing vcc at the head of each block of 8 (the usual pattern); alt: vcc alternates lanes
#define I_v_cndmask_vcc(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define I_v_cndmask_sgpr(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", s[20:21]\n"
#define I_v_cndmask_lit(n) "v_cndmask_b32_e64 %" #n ", -1, %" #n ", s[20:21]\n"
#define I_v_cndmask_dep0(n) "v_cndmask_b32 %0, %8, %0, vcc\n"

// ---- third batch (r02c): which v_cndmask_b32 forms are slow, operand-source effects ----
#define I_v_cndmask_e64_vcc(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", vcc\n"
#define I_v_cndmask_e32_3reg(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_v_cndmask_e32_const(n) "v_cndmask_b32 %" #n ", 0, %" #n ", vcc\n"
#define I_v_cndmask_sdwa(n) "v_cndmask_b32_sdwa %" #n ", %8, %" #n ", vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define I_v_cndmask_dpp(n) "v_cndmask_b32_dpp %" #n ", %8, %" #n ", vcc quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"
#define I_v_cndmask_e64_s22(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", s[22:23]\n"
#define I_mix_cmp_cnd_sgpr(n) "v_cmp_lt_u32 s[20:21], %8, %" #n "\ns_nop 1\nv_cndmask_b32_e64 %" #n ", %8, %" #n ", s[20:21]\n"
#define I_mix_cmp_cnd_vcc(n) "v_cmp_lt_u32 vcc, %8, %" #n "\ns_nop 1\nv_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define I_mix_cmp_cnd_vcc64(n) "v_cmp_lt_u32 vcc, %8, %" #n "\ns_nop 1\nv_cndmask_b32_e64 %" #n ", %8, %" #n ", vcc\n"
#define I_mix_cmp_addc(n) "v_cmp_lt_u32 vcc, %8, %" #n "\ns_nop 1\nv_addc_co_u32 %" #n ", vcc, 0, %" #n ", vcc\n"
#define I_v_add_u32_3reg(n) "v_add_u32 %" #n ", %8, %9\n"
#define I_v_add_u32_inl(n) "v_add_u32 %" #n ", 1, %" #n "\n"
#define I_v_and_b32_sgpr(n) "v_and_b32 %" #n ", s22, %" #n "\n"
#define I_v_mov_b32_sgpr(n) "v_mov_b32 %" #n ", s22\n"
#define I_v_mul_f32_sgpr(n) "v_mul_f32 %" #n ", s22, %" #n "\n"
#define I_v_fma_f32_3reg(n) "v_fma_f32 %" #n ", %8, %9, %10\n"
#define I_v_bfe_i32(n) "v_bfe_i32 %" #n ", %" #n ", 0, 16\n"
#define I_v_max_u16(n) "v_max_u16 %" #n ", %8, %" #n "\n"
#define I_v_min_i16(n) "v_min_i16 %" #n ", %8, %" #n "\n"
#define I_v_lshlrev_b16(n) "v_lshlrev_b16 %" #n ", 1, %" #n "\n"
#define I_v_mul_lo_u16_reg(n) "v_mul_lo_u16 %" #n ", %8, %9\n"
#define I_v_sub_u32_e64p(n) "v_sub_u32_e64 %" #n ", %" #n ", %8\n"
#define I_v_xnor_b32(n) "v_xnor_b32 %" #n ", %8, %" #n "\n"
#define I_v_sat_pk(n) "v_sat_pk_u8_i16 %" #n ", %" #n "\n"
#define I_v_cvt_f32_f16(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
#define I_v_add_i16_e64(n) "v_add_i16 %" #n ", %8, %" #n "\n"
#define I_v_mul_legacy(n) "v_mul_legacy_f32 %" #n ", %8, %" #n "\n"
#define I_v_exp_legacy(n) "v_exp_legacy_f32 %" #n ", %" #n "\n"
#define I_v_cvt_u16_f16(n) "v_cvt_u16_f16 %" #n ", %" #n "\n"
#define I_v_cvt_f16_u16(n) "v_cvt_f16_u16 %" #n ", %" #n "\n"
#define I_v_mul_f16(n) "v_mul_f16 %" #n ", %8, %" #n "\n"
#define I_v_max_f16(n) "v_max_f16 %" #n ", %8, %" #n "\n"

// ---- fourth batch (r02d): one v_cmp followed by k VOP2 v_cndmask on the same vcc ----
#define CND32(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define CND64(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", vcc\n"
#define I_mix3_cmp_cnd2_vcc(n) "v_cmp_lt_u32 vcc, %8, %" #n "\ns_nop 1\n" CND32(n) CND32(n)
#define I_mix5_cmp_cnd4_vcc(n) "v_cmp_lt_u32 vcc, %8, %" #n "\ns_nop 1\n" CND32(n) CND32(n) CND32(n) CND32(n)
#define I_mix5_cmp_cnd4_vcc64(n) "v_cmp_lt_u32 vcc, %8, %" #n "\ns_nop 1\n" CND64(n) CND64(n) CND64(n) CND64(n)
#define I_mix3_cmp_add_cnd(n) "v_cmp_lt_u32 vcc, %8, %" #n "\nv_add_u32 %" #n ", %8, %" #n "\n" CND32(n)
#define I_mix4_cmp_add2_cnd(n) "v_cmp_lt_u32 vcc, %8, %" #n "\nv_add_u32 %" #n ", %8, %" #n "\nv_add_u32 %" #n ", %8, %" #n "\n" CND32(n)
#define I_mix2_cmps_cnd(n) "v_cmp_lt_u32 s[20:21], %8, %" #n "\ns_nop 1\n" CND32(n)
#define I_mix2_smov_cnd(n) "s_mov_b64 vcc, s[22:23]\n" CND32(n)
#define I_mix2_cmpx(n) "v_cmp_lt_u32 vcc, %8, %" #n "\nv_cmp_lt_u32 vcc, %9, %" #n "\n"

// ---- fifth batch (r02e): do full-rate ops keep their rate when interleaved with 4-cycle ops? ----
#define ADDn(n) "v_add_u32 %" #n ", %8, %" #n "\n"
#define ANDn(n) "v_and_b32 %" #n ", %9, %" #n "\n"
#define LSRn(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define PRMn(n) "v_perm_b32 %" #n ", %8, %" #n ", %9\n"
#define DOTn(n) "v_dot2_i32_i16 %" #n ", %8, %9, %" #n "\n"
#define FMAn(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I_mix2_add_and(n) ADDn(n) ANDn(n)
#define I_mix3_add_and_lsr(n) ADDn(n) ANDn(n) LSRn(n)
#define I_mix2_add_perm(n) ADDn(n) PRMn(n)
#define I_mix4_add2_perm2(n) ADDn(n) ADDn(n) PRMn(n) PRMn(n)
#define I_mix8_add4_perm4(n) ADDn(n) ADDn(n) ADDn(n) ADDn(n) PRMn(n) PRMn(n) PRMn(n) PRMn(n)
#define I_mix4_add3_perm1(n) ADDn(n) ADDn(n) ADDn(n) PRMn(n)
#define I_mix4_add1_perm3(n) ADDn(n) PRMn(n) PRMn(n) PRMn(n)
#define I_mix2_add_dot(n) ADDn(n) DOTn(n)
#define I_mix2_fma_dot(n) FMAn(n) DOTn(n)
#define I_mix2_add_fma(n) ADDn(n) FMAn(n)
#define I_s_nop0(n) "s_nop 0\n"
#define I_s_add_u32(n) "s_add_u32 s20, s20, 1\n"
// mixed pairs: does a transcendental / a packed op pair with a plain op in the same issue window?
#define I_mix_sqrt_add(n) "v_sqrt_f32 %" #n ", %" #n "\n"                                                     \
                          "v_add_u32 %" #n ", %8, %" #n "\n"
#define I_mix_valu_salu(n) "v_add_u32 %" #n ", %8, %" #n "\n"                                                 \
                           "s_add_u32 s20, s20, 1\n"

// 32-bit destination kernels
#define DEFK32(NAME, LIST, IM, PRE)                                                                                      \
    template <int LDSB>                                                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(Rec* out, uint32_t iters, uint32_t a, uint32_t b) {        \
        __shared__ uint32_t pad[LDSB / 4];                                                                      \
        uint32_t r0 = threadIdx.x + a, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 77, r4 = r0 * 5, r5 = ~r0,    \
                 r6 = r0 + b, r7 = r0 - b;                                                                      \
        uint32_t va = a * 7 + (threadIdx.x & 3), vb = b | 0x01020304u;                                          \
        if (iters == 0xFFFFFFFFu) pad[threadIdx.x] = a; /* never true: keeps the LDS allocation alive */        \
        __syncthreads();                                                                                        \
        asm volatile(PRE ::: "vcc", "s20", "s21", "s22", "s23", "memory");                                             \
        uint64_t R0 = memrealtime();                                                                            \
        uint64_t T0 = memtime();                                                                                \
        for (uint32_t i = 0; i < iters; ++i) {                                                                  \
            asm volatile(X16(LIST(IM))                                                                    \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)       \
                         : "v"(va), "v"(vb)                                                                     \
                         : "vcc", "s20", "s21", "s22", "s23", "scc", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");                                                         \
        }                                                                                                       \
        uint64_t T1 = memtime();                                                                                \
        uint64_t R1 = memrealtime();                                                                            \
        uint32_t sink = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                  \
        if (iters == 0xFFFFFFFFu) sink ^= pad[(threadIdx.x * 7) & 255];                                         \
        uint32_t hw, xc;                                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));                                       \
        uint32_t lane_sink = __builtin_amdgcn_readfirstlane(sink);                                              \
        if ((threadIdx.x & 63) == 0) {                                                                          \
            Rec r;                                                                                              \
            r.t0 = T0; r.t1 = T1; r.r0 = R0; r.r1 = R1; r.hw_id = hw; r.xcc_id = xc; r.sink = lane_sink; r.pad = 0; \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;                                                       \
        }                                                                                                       \
    }

// 64-bit destination kernels (register pairs): operands p0..p7 are 64-bit, q is a 64-bit source
#define DEFK64(NAME, LIST, IM, PRE)                                                                                      \
    template <int LDSB>                                                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(Rec* out, uint32_t iters, uint32_t a, uint32_t b) {        \
        __shared__ uint32_t pad[LDSB / 4];                                                                      \
        uint64_t r0 = threadIdx.x + a, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 77, r4 = r0 * 5, r5 = ~r0,    \
                 r6 = r0 + b, r7 = r0 - b;                                                                      \
        uint64_t vq = ((uint64_t)(a * 7 + 1) << 32) | (b | 3u);                                                 \
        uint32_t va = a * 7 + (threadIdx.x & 3), vb = b | 0x01020304u;                                          \
        if (iters == 0xFFFFFFFFu) pad[threadIdx.x] = a;                                                         \
        __syncthreads();                                                                                        \
        asm volatile(PRE ::: "vcc", "s20", "s21", "s22", "s23", "memory");                                             \
        uint64_t R0 = memrealtime();                                                                            \
        uint64_t T0 = memtime();                                                                                \
        for (uint32_t i = 0; i < iters; ++i) {                                                                  \
            asm volatile(X16(LIST(IM))                                                                    \
                         : [p0] "+v"(r0), [p1] "+v"(r1), [p2] "+v"(r2), [p3] "+v"(r3), [p4] "+v"(r4),           \
                           [p5] "+v"(r5), [p6] "+v"(r6), [p7] "+v"(r7)                                          \
                         : "v"(va), "v"(vb), [q] "v"(vq)                                                        \
                         : "vcc", "s20", "s21", "s22", "s23", "scc", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");                                                         \
        }                                                                                                       \
        uint64_t T1 = memtime();                                                                                \
        uint64_t R1 = memrealtime();                                                                            \
        uint64_t s64 = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                   \
        uint32_t sink = (uint32_t)s64 ^ (uint32_t)(s64 >> 32);                                                  \
        if (iters == 0xFFFFFFFFu) sink ^= pad[(threadIdx.x * 7) & 255];                                         \
        uint32_t hw, xc;                                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));                                       \
        uint32_t lane_sink = __builtin_amdgcn_readfirstlane(sink);                                              \
        if ((threadIdx.x & 63) == 0) {                                                                          \
            Rec r;                                                                                              \
            r.t0 = T0; r.t1 = T1; r.r0 = R0; r.r1 = R1; r.hw_id = hw; r.xcc_id = xc; r.sink = lane_sink; r.pad = 0; \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;                                                       \
        }                                                                                                       \
    }

#define K32(NAME) DEFK32(NAME, X8, I_##NAME, "s_mov_b32 s22, 5\n")
#define K64(NAME) DEFK64(NAME, X8, I_##NAME, "s_mov_b32 s22, 5\n")

// dependent-chain variants get their own names
#define I_dep_v_add_u32(n) I_v_add_u32(0)
#define I_dep_v_perm_b32(n) I_v_perm_b32(0)
#define I_dep_v_dot2_i32_i16_acc(n) I_v_dot2_i32_i16_acc(0)
#define I_dep_v_pk_add_u16(n) I_v_pk_add_u16(0)
#define I_dep_v_mul_lo_u32(n) I_v_mul_lo_u32(0)
#define I_dep_v_sqrt_f32(n) I_v_sqrt_f32(0)
#define I_dep_v_fma_f32(n) I_v_fma_f32(0)
#define I_dep_v_mad_i32_i24(n) I_v_mad_i32_i24(0)

#define ALL32(X)                                                                                                \
    X(v_mov_b32) X(v_add_u32) X(v_sub_u32_clamp) X(v_add3_u32) X(v_lshl_add_u32) X(v_and_b32) X(v_and_or_b32)   \
    X(v_lshlrev_b32) X(v_lshrrev_b32) X(v_ashrrev_i32) X(v_bfe_u32) X(v_bfi_b32) X(v_alignbit_b32) X(v_perm_b32)\
    X(v_max_u32) X(v_min_i32) X(v_med3_i32) X(v_cndmask_b32) X(v_cmp_lt_u32) X(v_cmp_lt_u32_sgpr)               \
    X(v_mul_lo_u32) X(v_mul_hi_u32) X(v_mul_u32_u24) X(v_mul_i32_i24) X(v_mad_u32_u24) X(v_mad_i32_i24)         \
    X(v_dot2_i32_i16_acc) X(v_dot2_i32_i16_c0) X(v_dot4_i32_i8) X(v_pk_add_u16) X(v_pk_sub_i16) X(v_pk_mad_u16) \
    X(v_pk_mul_lo_u16) X(v_pk_lshrrev_b16) X(v_pk_ashrrev_i16) X(v_pk_max_i16) X(v_add_u16_sdwa)                \
    X(v_add_u32_dpp) X(v_mov_b32_dpp) X(v_fma_f32) X(v_add_f32) X(v_mul_f32) X(v_sqrt_f32) X(v_rsq_f32)         \
    X(v_rcp_f32) X(v_log_f32) X(v_exp_f32) X(v_cvt_f32_u32) X(v_cvt_u32_f32) X(v_cvt_f32_i32)                   \
    X(v_cvt_pk_u16_u32) X(v_sad_u16) X(v_sad_u32) X(v_readlane) X(v_readfirstlane) X(v_ffbh_u32) X(v_xad_u32)   \
    X(v_add_co_u32) X(v_addc_co_u32) X(s_nop0) X(s_add_u32) X(mix_sqrt_add) X(mix_valu_salu)                    \
    X(dep_v_add_u32) X(dep_v_perm_b32) X(dep_v_dot2_i32_i16_acc) X(dep_v_pk_add_u16) X(dep_v_mul_lo_u32)        \
    X(dep_v_sqrt_f32) X(dep_v_fma_f32) X(dep_v_mad_i32_i24)
#define ALL64(X) X(v_pk_mov_b32) X(v_mad_u64_u32) X(v_pk_fma_f32) X(v_pk_mul_f32)

#define NEW32(X)                                                                                                \
    X(v_or_b32) X(v_xor_b32) X(v_not_b32) X(v_sub_u32) X(v_subrev_u32) X(v_add_u32_e64) X(v_add_u32_lit)        \
    X(v_add_u32_sgpr) X(v_lshlrev_b32_reg) X(v_lshrrev_b32_reg) X(v_lshrrev_b32_16) X(v_ashrrev_i32_16)         \
    X(v_min_f32) X(v_max_f32) X(v_sub_f32) X(v_fmac_f32) X(v_fma_f32_neg) X(v_mul_f32_lit) X(v_cmp_lt_f32)      \
    X(v_cmp_eq_u32) X(v_cmp_lt_i32) X(v_cmp_class_f32) X(v_floor_f32) X(v_trunc_f32) X(v_rndne_f32)             \
    X(v_fract_f32) X(v_ldexp_f32) X(v_cvt_i32_f32) X(v_cvt_f32_ubyte0) X(v_cvt_f16_f32) X(v_add_u16)            \
    X(v_sub_u16) X(v_mul_lo_u16) X(v_lshrrev_b16) X(v_ashrrev_i16) X(v_max_i16) X(v_mad_u16) X(v_add_f16)       \
    X(v_fma_f16) X(v_pk_add_f16) X(v_pk_fma_f16) X(v_pk_add_i16) X(v_pk_lshlrev_b16) X(v_pk_min_u16)            \
    X(v_dot2_u32_u16) X(v_dot2c_i32_i16) X(v_dot2_f32_f16) X(v_dot2c_f32_bf16) X(v_mad_u32_u16)                 \
    X(v_mad_i32_i16) X(v_mul_hi_u32_u24) X(v_lshl_or_b32) X(v_or3_b32) X(v_add_lshl_u32) X(v_sub_co_u32)        \
    X(v_bcnt_u32_b32) X(v_mbcnt_lo) X(v_bfrev_b32) X(v_swap_b32) X(v_accvgpr_write) X(v_accvgpr_read)           \
    X(v_mov_b32_sdwa) X(v_mov_b32_lit)
#define NEW32C(X)                                                                                               \
    X(v_cndmask_e32_3reg) X(v_cndmask_e32_const) X(v_cndmask_sdwa) X(v_cndmask_dpp) X(v_add_u32_3reg)           \
    X(v_add_u32_inl) X(v_and_b32_sgpr) X(v_mov_b32_sgpr) X(v_mul_f32_sgpr) X(v_bfe_i32) X(v_max_u16)            \
    X(v_min_i16) X(v_lshlrev_b16) X(v_mul_lo_u16_reg) X(v_sub_u32_e64p) X(v_xnor_b32) X(v_sat_pk)               \
    X(v_cvt_f32_f16) X(v_add_i16_e64) X(v_mul_legacy) X(v_exp_legacy) X(v_cvt_u16_f16) X(v_cvt_f16_u16)         \
    X(v_mul_f16) X(v_max_f16) X(mix_cmp_cnd_sgpr) X(mix_cmp_cnd_vcc) X(mix_cmp_cnd_vcc64) X(mix_cmp_addc)
#define NEW32D(X) X(mix3_cmp_cnd2_vcc) X(mix5_cmp_cnd4_vcc) X(mix5_cmp_cnd4_vcc64) X(mix3_cmp_add_cnd) X(mix4_cmp_add2_cnd) \
    X(mix2_cmps_cnd) X(mix2_smov_cnd) X(mix2_cmpx)
#define NEW32E(X) X(mix2_add_and) X(mix3_add_and_lsr) X(mix2_add_perm) X(mix4_add2_perm2) X(mix8_add4_perm4) \
    X(mix4_add3_perm1) X(mix4_add1_perm3) X(mix2_add_dot) X(mix2_fma_dot) X(mix2_add_fma)
#define NEW64(X) X(v_mov_b64) X(v_lshlrev_b64) X(v_lshl_add_u64) X(v_pk_add_f32)

ALL32(K32)
ALL64(K64)
NEW32(K32)
NEW64(K64)
NEW32C(K32)
NEW32D(K32)
NEW32E(K32)
// ---- round 3: run-length experiment.  Does the 2-cycle issue mode of the plain ops engage when the
// 4-cycle ops are clustered, i.e. is there a run length N above which "add x N, perm x N" costs less than 4 cycles per
// instruction?  Bodies of 128 / 512 / 2048 instructions; every wave runs the same code, so on a SIMD the W waves drift
// through the runs independently (the hardware arbitrates between them every cycle).
#define ADD8 X8(I_v_add_u32)
#define PRM8 X8(I_v_perm_b32)
#define DOT8 X8(I_v_dot2_i32_i16_acc)
#define REP2(B) B B
#define REP4(B) B B B B
#define REP8(B) B B B B B B B B
#define REP16(B) REP4(REP4(B))
#define LIST_ID(S) S
#define BODY_run8 REP8(ADD8 PRM8)
#define BODY_run16 REP4(REP2(ADD8) REP2(PRM8))
#define BODY_run32 REP2(REP4(ADD8) REP4(PRM8))
#define BODY_run64 REP8(ADD8) REP8(PRM8)
#define BODY_run256 REP4(REP8(ADD8)) REP4(REP8(PRM8))
#define BODY_run1024 REP16(REP8(ADD8)) REP16(REP8(PRM8))
#define BODY_run64_dot REP8(ADD8) REP8(DOT8)
#define DEFKBODY(NAME) DEFK32B(NAME, BODY_##NAME)
// same frame as DEFK32 with an explicit loop body
#define DEFK32B(NAME, BODY)                                                                                     \
    template <int LDSB>                                                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(Rec* out, uint32_t iters, uint32_t a, uint32_t b) {        \
        __shared__ uint32_t pad[LDSB / 4];                                                                      \
        uint32_t r0 = threadIdx.x + a, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 77, r4 = r0 * 5, r5 = ~r0,    \
                 r6 = r0 + b, r7 = r0 - b;                                                                      \
        uint32_t va = a * 7 + (threadIdx.x & 3), vb = b | 0x01020304u;                                          \
        if (iters == 0xFFFFFFFFu) pad[threadIdx.x] = a;                                                         \
        __syncthreads();                                                                                        \
        uint64_t R0 = memrealtime();                                                                            \
        uint64_t T0 = memtime();                                                                                \
        for (uint32_t i = 0; i < iters; ++i) {                                                                  \
            asm volatile(BODY                                                                                   \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)       \
                         : "v"(va), "v"(vb)                                                                     \
                         : "vcc", "scc");                                                                       \
        }                                                                                                       \
        uint64_t T1 = memtime();                                                                                \
        uint64_t R1 = memrealtime();                                                                            \
        uint32_t sink = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                  \
        if (iters == 0xFFFFFFFFu) sink ^= pad[(threadIdx.x * 7) & 255];                                         \
        uint32_t hw, xc;                                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));                                       \
        uint32_t lane_sink = __builtin_amdgcn_readfirstlane(sink);                                              \
        if ((threadIdx.x & 63) == 0) {                                                                          \
            Rec r;                                                                                              \
            r.t0 = T0; r.t1 = T1; r.r0 = R0; r.r1 = R1; r.hw_id = hw; r.xcc_id = xc; r.sink = lane_sink; r.pad = 0; \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;                                                       \
        }                                                                                                       \
    }
#define RUNS(X) X(run8) X(run16) X(run32) X(run64) X(run256) X(run1024) X(run64_dot)
RUNS(DEFKBODY)
// "wave A pure adds / wave B pure perms on the same SIMD": the wave slot (HW_ID[3:0]) picks the stream, so that with
// W >= 2 waves on a SIMD half of them issue only plain adds and the other half only permutes (Rec.pad = 1 / 2)
#define DEFKSPLIT(NAME, BODYA, BODYB)                                                                           \
    template <int LDSB>                                                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(Rec* out, uint32_t iters, uint32_t a, uint32_t b) {        \
        __shared__ uint32_t pad[LDSB / 4];                                                                      \
        uint32_t r0 = threadIdx.x + a, r1 = r0 * 3 + 1, r2 = r0 ^ 0x55, r3 = r0 + 77, r4 = r0 * 5, r5 = ~r0,    \
                 r6 = r0 + b, r7 = r0 - b;                                                                      \
        uint32_t va = a * 7 + (threadIdx.x & 3), vb = b | 0x01020304u;                                          \
        if (iters == 0xFFFFFFFFu) pad[threadIdx.x] = a;                                                         \
        __syncthreads();                                                                                        \
        uint32_t hw, xc;                                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));                                       \
        const bool odd = __builtin_amdgcn_readfirstlane(hw) & 1;                                               \
        uint64_t R0 = memrealtime();                                                                            \
        uint64_t T0 = memtime();                                                                                \
        if (odd) {                                                                                              \
            for (uint32_t i = 0; i < iters; ++i)                                                                \
                asm volatile(BODYB : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(va), "v"(vb) : "vcc", "scc");                                                \
        } else {                                                                                                \
            for (uint32_t i = 0; i < iters; ++i)                                                                \
                asm volatile(BODYA : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(va), "v"(vb) : "vcc", "scc");                                                \
        }                                                                                                       \
        uint64_t T1 = memtime();                                                                                \
        uint64_t R1 = memrealtime();                                                                            \
        uint32_t sink = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                  \
        if (iters == 0xFFFFFFFFu) sink ^= pad[(threadIdx.x * 7) & 255];                                         \
        uint32_t lane_sink = __builtin_amdgcn_readfirstlane(sink);                                              \
        if ((threadIdx.x & 63) == 0) {                                                                          \
            Rec r;                                                                                              \
            r.t0 = T0; r.t1 = T1; r.r0 = R0; r.r1 = R1; r.hw_id = hw; r.xcc_id = xc; r.sink = lane_sink; r.pad = odd ? 2 : 1; \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;                                                       \
        }                                                                                                       \
    }
DEFKSPLIT(split_add_perm, REP16(ADD8), REP16(PRM8))
DEFKSPLIT(split_add_add, REP16(ADD8), REP16(ADD8))
DEFKSPLIT(split_perm_perm, REP16(PRM8), REP16(PRM8))
#define SPLITS(X) X(split_add_perm) X(split_add_add) X(split_perm_perm)
DEFK32(v_cndmask_e64_vcc, X8, I_v_cndmask_e64_vcc, "s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x0f0f0f0f\n")
DEFK32(v_cndmask_e64_s22, X8, I_v_cndmask_e64_s22, "s_mov_b32 s22, 0x55555555\ns_mov_b32 s23, 0x0f0f0f0f\n")
// v_cndmask_b32 under different mask sources
DEFK32(v_cndmask_vcc0, X8, I_v_cndmask_vcc, "s_mov_b64 vcc, 0\n")
DEFK32(v_cndmask_vcc1, X8, I_v_cndmask_vcc, "s_mov_b64 vcc, -1\n")
DEFK32(v_cndmask_vccalt, X8, I_v_cndmask_vcc, "s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x0f0f0f0f\n")
DEFK32(v_cndmask_vcc_valu, X8, I_v_cndmask_vcc, "v_cmp_lt_u32 vcc, 77, v0\ns_nop 4\n")
DEFK32(v_cndmask_sgpr0, X8, I_v_cndmask_sgpr, "s_mov_b64 s[20:21], 0\n")
DEFK32(v_cndmask_sgpralt, X8, I_v_cndmask_sgpr, "s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x0f0f0f0f\n")
DEFK32(v_cndmask_lit, X8, I_v_cndmask_lit, "s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x0f0f0f0f\n")
DEFK32(v_cndmask_dep, X8, I_v_cndmask_dep0, "s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x0f0f0f0f\n")
#define CND(X) X(v_cndmask_vcc0) X(v_cndmask_vcc1) X(v_cndmask_vccalt) X(v_cndmask_vcc_valu) X(v_cndmask_sgpr0) \
    X(v_cndmask_sgpralt) X(v_cndmask_lit) X(v_cndmask_dep) X(v_cndmask_e64_vcc) X(v_cndmask_e64_s22)


// ---- FETCH_SIZE calibration (run under rocprofv3 --pmc FETCH_SIZE): known byte counts, k_mfcc's access pattern ----
// k_calib_stream16: every lane reads 16 B, fully coalesced, each byte of the buffer once.
// k_calib_frames  : the frame kernel's pattern -- a wave reads 160-sample frames at a hop of 80 samples (every sample
//                   is requested by two frames) as 3 dword loads per lane at 2-byte-aligned addresses.
__global__ __launch_bounds__(256) void k_calib_stream16(const uint4* p, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
typedef uint32_t u32_a2 __attribute__((aligned(2)));
__global__ __launch_bounds__(256) void k_calib_frames(const uint16_t* p, size_t rows, size_t row_len, uint32_t frames,
                                                      uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (size_t item = wave; item < rows * (frames / 16); item += nw) {
        const size_t r = item / (frames / 16), t = item % (frames / 16);
        const uint16_t* x0 = p + r * row_len + 2400 + 80 * 16 * t;
        for (int f = 0; f < 16; ++f) {
            const uint16_t* x = x0 + 80 * f;
            for (int k = 0; k < 3; ++k) {
                const int i = lane + 64 * k;
                if (i < 160) acc ^= *(const u32_a2*)(x + i - 1);
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
static int calib_main() {
    const size_t rows = 20000, row_len = 25360, frames = 256;   // 1.01 GB, > the 256 MiB Infinity Cache
    const size_t bytes = rows * row_len * 2;
    uint16_t* d;
    uint32_t* sink;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(d, 1, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_calib_stream16, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)d, bytes / 16, sink);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_calib_frames, dim3(256 * 8), dim3(256), 0, 0, d, rows, row_len, (uint32_t)frames, sink);
        CHECK(hipDeviceSynchronize());
    }
    const size_t span = rows * ((80 * (frames - 1) + 160 + 1) * 2);
    printf("{\"calib\": true, \"stream16_bytes\": %zu, \"frames_unique_bytes\": %zu, \"frames_requested_bytes\": %zu}\n", bytes,
           span, rows * frames * 160 * 4);
    return 0;
}

typedef void (*kfn)(Rec*, uint32_t, uint32_t, uint32_t);
struct Entry {
    const char* name;
    int per_slot;   // instructions per S(n) expansion (2 for the mixed pairs)
    kfn f[4];       // W = 1, 2, 4, 8
};
// LDS per workgroup so that exactly W workgroups fit in the CU's 160 KiB
#define L1 (100 * 1024)
#define L2 (80 * 1024)
#define L4 (40 * 1024)
#define L8 (20 * 1024)
#define ENT(NAME) {#NAME, 1, {k_##NAME<L1>, k_##NAME<L2>, k_##NAME<L4>, k_##NAME<L8>}},
static Entry table[] = {ALL32(ENT) ALL64(ENT) NEW32(ENT) NEW64(ENT) NEW32C(ENT) NEW32D(ENT) NEW32E(ENT) CND(ENT) RUNS(ENT) SPLITS(ENT)};

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "calib") return calib_main();
    uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    const char* only = argc > 2 ? argv[2] : nullptr;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    Rec* d;
    const int maxw = cus * 8 * 4;
    CHECK(hipMalloc(&d, sizeof(Rec) * maxw));
    std::vector<Rec> h(maxw);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"gcnArch\": \"%s\", \"cus\": %d, \"clock_khz_reported\": %d, \"iters\": %u,\n"
           " \"method\": \"ITERS x 128 instructions of one opcode per wave (8 independent destination registers; dep_* = one "
           "register), 256-thread workgroups = 1 wave per SIMD, LDS-limited to W workgroups per CU, grid = CUs x W; "
           "cyc = median s_memtime delta / (instructions per wave x W)\",\n \"ops\": [\n",
           p.name, p.gcnArchName, cus, p.clockRate, iters);
    const int Ws[4] = {1, 2, 4, 8};
    bool first = true;
    for (auto& e : table) {
        if (only && std::string(e.name).find(only) == std::string::npos) continue;
        std::string nm = e.name;
        int per_slot = (nm.rfind("mix_", 0) == 0) ? 2 : (nm.rfind("mix", 0) == 0 ? nm[3] - '0' : 1);   // s_nop in the cmp pairs is not counted
        if (nm == "run256") per_slot = 4;      // bodies longer than 128 instructions
        if (nm == "run1024") per_slot = 16;
        uint64_t ninst = (uint64_t)iters * 128 * per_slot;
        printf("%s  {\"op\": \"%s\", \"insts_per_wave\": %llu", first ? "" : ",\n", e.name, (unsigned long long)ninst);
        first = false;
        for (int wi = 0; wi < 4; ++wi) {
            int W = Ws[wi];
            int grid = cus * W;
            hipLaunchKernelGGL(e.f[wi], dim3(grid), dim3(256), 0, 0, d, 16u, 3u, 5u);   // warm-up (code fetch, clocks)
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.f[wi], dim3(grid), dim3(256), 0, 0, d, iters, 3u, 5u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            int nw = grid * 4;
            CHECK(hipMemcpy(h.data(), d, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
            std::vector<double> cyc(nw), clk(nw);
            std::map<uint64_t, int> per_simd;
            uint64_t tmin = ~0ull, tmax = 0;
            for (int i = 0; i < nw; ++i) {
                cyc[i] = (double)(h[i].t1 - h[i].t0);
                double real_ns = (double)(h[i].r1 - h[i].r0) * 10.0;   // 100 MHz
                clk[i] = real_ns > 0 ? cyc[i] / real_ns : 0;            // GHz if s_memtime ticks at the shader clock
                // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
                uint64_t key = ((uint64_t)(h[i].xcc_id & 0xF) << 16) | (h[i].hw_id & 0xFF30u);
                per_simd[key]++;
                tmin = std::min(tmin, h[i].r0);
                tmax = std::max(tmax, h[i].r1);
            }
            int wmin = 1 << 30, wmax = 0;
            for (auto& kv : per_simd) {
                wmin = std::min(wmin, kv.second);
                wmax = std::max(wmax, kv.second);
            }
            std::vector<double> cycA, cycB;   // split kernels: waves of stream A (pad 1) / B (pad 2)
            for (int i = 0; i < nw; ++i) {
                if (h[i].pad == 1) cycA.push_back(cyc[i]);
                if (h[i].pad == 2) cycB.push_back(cyc[i]);
            }
            std::sort(cycA.begin(), cycA.end());
            std::sort(cycB.begin(), cycB.end());
            std::sort(cyc.begin(), cyc.end());
            std::sort(clk.begin(), clk.end());
            double med = cyc[nw / 2], mx = cyc[nw - 1], mn = cyc[0];
            double ghz = clk[nw / 2];
            double span_ns = (double)(tmax - tmin) * 10.0;
            printf(",\n   \"W%d\": {\"cyc_per_inst_per_simd\": %.3f, \"cyc_min\": %.3f, \"cyc_max\": %.3f, "
                   "\"memtime_ghz\": %.3f, \"simds_used\": %zu, \"waves_per_simd_min\": %d, \"waves_per_simd_max\": %d, "
                   "\"kernel_ms_event\": %.4f, \"span_ms_realtime\": %.4f, \"ns_per_inst_per_simd_wall\": %.4f}",
                   W, med / ((double)ninst * W), mn / ((double)ninst * W), mx / ((double)ninst * W), ghz,
                   per_simd.size(), wmin, wmax, ms, span_ns * 1e-6, span_ns / ((double)ninst * W));
            if (!cycA.empty() || !cycB.empty()) {
                // per-stream: median wave time / instructions of ONE wave (a wave alone would show ~4.3)
                printf(",\n   \"W%d_split\": {\"waves_A\": %zu, \"waves_B\": %zu, \"cyc_per_inst_per_wave_A\": %.3f, \"cyc_per_inst_per_wave_B\": %.3f}",
                       W, cycA.size(), cycB.size(), cycA.empty() ? 0.0 : cycA[cycA.size() / 2] / (double)ninst,
                       cycB.empty() ? 0.0 : cycB[cycB.size() / 2] / (double)ninst);
            }
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n]}\n");
    return 0;
}
