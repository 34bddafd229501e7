// sr_dev.h -- device helpers shared by every kernel translation unit (packed 16-bit arithmetic, exact sqrtf / log, DPP scans).
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#pragma once
#include <cstdlib>

#include "sr_device.h"
#include "sr_tables.h"

namespace sr {

typedef short short2v __attribute__((ext_vector_type(2)));
// native vector types: one load instruction of exactly this width (HIP's uint2/uint4 structs get re-split)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    // v_dot2_i32_i16: a.lo*b.lo + a.hi*b.hi + c, signed 16-bit halves, 32-bit wrap
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
}
__device__ __forceinline__ int sext_lo(uint32_t w) { return (int)(short)(w & 0xFFFFu); }
__device__ __forceinline__ int sext_hi(uint32_t w) { return (int)w >> 16; }
__device__ __forceinline__ uint32_t pack16(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }

// 16-bit dot product without accumulator (VOP3P form with inline 0: no v_mov to clear a destination)
__device__ __forceinline__ int sdot2z(uint32_t a, uint32_t b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a.lo*b.lo + a.hi*b.hi + c with a separate destination (VOP3P): C+D and C-D of the butterfly come straight out
// of the multiplier when the third leg's coefficient is also kept negated
__device__ __forceinline__ int sdot2a(uint32_t a, uint32_t b, int c)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Pre-emphasis term of MFCC.C:119, (s32)p * hp_ratio with hp_ratio = 95/100 in integer arithmetic: p*95/100 truncated toward
// zero, for |p| <= 65535 (u16 sample minus a u16 mid value; sr_mfcc_batch rejects a larger mid).  As three IEEE operations
// -- convert, multiply by 0.95000005f (the float above 0.95), convert with truncation -- instead of a multiply and a
// four-instruction signed division: p*95/100 is a multiple of 0.05, the relative error of the product stays below 2e-7,
// i.e. below 0.012 in absolute terms, so truncation lands on the same integer; checked for every p of the domain with
// the same IEEE operations in tests/test_oracle.py::test_preemphasis_float_form_is_exact.
__device__ __forceinline__ int preemph95(int p) { return (int)((float)p * 0.95000005f); }
// the same term negated, -(p*95/100): IEEE multiplication and the truncating conversion are symmetric in the sign
__device__ __forceinline__ int neg_preemph95(int p) { return (int)((float)p * -0.95000005f); }
// Pre-emphasis + Hamming weight of one sample (MFCC.C:119, 122) from the dword that holds x[i-1] (low half) and x[i] (high
// half), as the 16-bit pattern of the pass-1 output (s16)(temp*hamm/1000) >> 2: t = (x[i] - mid) - (x[i-1] - mid)*95/100 is
// formed already shifted by kWinShift (v_add_lshl_u32), the division by 1000 is folded into hm = hamm_fused_multiplier(hamm[i])
// (sr_tables.h: one v_mad_i64_i32, exact on the whole domain), and the s16 wrap + ">> 2" is one v_bfe_i32.
// the quotient trunc(t * hamm / 1000) of t = c + nq (c = x[i] - mid, nq = -((x[i-1] - mid)*95/100)), before the s16 wrap
__device__ __forceinline__ int window_quotient(int c, int nq, int hm)
{
    int u;  // t << 5 in one instruction
    asm("v_add_lshl_u32 %0, %1, %2, %3" : "=v"(u) : "v"(c), "v"(nq), "n"(kWinShift));
    const long long P = (long long)u * (long long)hm + (long long)(unsigned long long)(uint32_t)(u >> 31);
    return (int)(P >> 32);
}
// SH = 2: the pass-1 output A >> 2 of the sample; SH = 4: that shifted once more -- a sample that is only ever the A leg of a
// pass-2 butterfly (rows 0..63 of the zero-padded frame, sr_fft_dev.h fft_front_real160) is stored as pass 2 consumes it,
// (x >> 2) >> 2 = x >> 4 for arithmetic shifts, and pass 2 drops its own shift (round 6)
template <int SH = 2>
__device__ __forceinline__ uint32_t window_sample(uint32_t prev_cur, int mid, int hm)
{
    const int nq = neg_preemph95((int)(prev_cur & 0xFFFFu) - mid);  // -((x[i-1] - mid)*95/100)
    const int c = (int)(prev_cur >> 16) - mid;                      // x[i] - mid
    int u;                                                          // t << 5, t = c + nq, in one instruction
    asm("v_add_lshl_u32 %0, %1, %2, %3" : "=v"(u) : "v"(c), "v"(nq), "n"(kWinShift));
    const long long P = (long long)u * (long long)hm + (long long)(unsigned long long)(uint32_t)(u >> 31);
    // (s16)quotient >> 2 = bits 2..15 of the high dword, sign-extended: one v_bfe_i32 (stated as the instruction: left to the
    // compiler the 64-bit shift became v_alignbit_b32 + v_ashrrev_i32)
    int r;
    asm("v_bfe_i32 %0, %1, %2, %3" : "=v"(r) : "v"((int)(P >> 32)), "n"(SH), "n"(16 - SH));
    return (uint32_t)r;
}
// full-rate 24-bit multiplies where the operands provably fit (quarter-rate v_mul_lo_u32 otherwise)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ uint32_t umul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
// a*b + c on the signed low 24 bits of a and b, one instruction
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// One Mel filterbank term floor(E * tri / 100) (MFCC.C:139-161) for E <= kMelFusedMaxE: a single v_mul_hi_u32 of E << 4 with
// m = mel_fused_multiplier(tri); and the weight back out of its multiplier for the literal form (sr_tables.h has the proof,
// sr_mel_term_sweep the exhaustive check of exactly these two functions).
__device__ __forceinline__ uint32_t mel_term_fused(uint32_t e_shl4, uint32_t m) { return __umulhi(e_shl4, m); }
__device__ __forceinline__ uint32_t mel_tri_of_multiplier(uint32_t m) { return __umulhi(m, 1600u); }

// sqrtf of an integer-valued float, correctly rounded -- bit-identical to IEEE sqrtf, hence to the reference's sqrtf calls
// (MFCC.C:58, DTW.C:59).  Round 3: Markstein's fused correction of a reciprocal-root seed,
//     y = v_rsq_f32(f)   s0 = f*y   h = 0.5*y   r = fma(-s0, s0, f)   s = fma(r, h, s0)
// 6 issue slots (v_rsq counts twice) instead of 9 for v_sqrt_f32 + the residual test against both neighbouring floats, and
// the two multiplies / two fmas of TWO roots pack into v_pk_mul_f32 / v_pk_fma_f32 (sqrt_rn_int2).  The seed f*y is off by
// up to 2 ulp (wrong for 33 % of the inputs), the corrected value is the correctly rounded root for EVERY u32 input on
// gfx950: proven by exhaustion, tests/exhaustive_math_sweep.py sweeps all 2^32 values through sr_math_diag against the
// host's sqrtf (profiles/r03_exhaustive_math_sweep.txt), and every -m gpu run repeats a 2 M-value subset.  f = 0 gives
// y = inf and s = NaN: every caller converts with v_cvt_u32_f32, for which NaN is 0 = (u32)sqrtf(0).
__device__ __forceinline__ float sqrt_rn_int(float f)
{
    const float y = __builtin_amdgcn_rsqf(f);
    const float s0 = f * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-s0, s0, f);
    return __builtin_fmaf(r, h, s0);
}

// (u32)(sqrtf(f) * 10) for SMALL integer-valued f (MFCC.C:56-58 on quiet bins): v_sqrt_f32 (within 1 ulp) instead of the
// corrected reciprocal-root seed.  Only valid where sr_mag_fast_sweep has shown it equal to the exact form.
__device__ __forceinline__ uint32_t mag10_small(float f)
{
    uint32_t r;
    const float m = __builtin_amdgcn_sqrtf(f) * 10.0f;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(m));
    return r;
}

// float -> u32 with the HARDWARE's semantics (v_cvt_u32_f32: truncation, NaN -> 0, saturation), stated as the instruction:
// a C++ cast of NaN is undefined behaviour, and sqrt_rn_int(0) is NaN by design
__device__ __forceinline__ uint32_t cvt_u32(float x)
{
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two roots at once: the multiplies and fused corrections are packed f32 operations (one issue slot for both roots)
__device__ __forceinline__ f32x2 sqrt_rn_int2(f32x2 f)
{
    const f32x2 y = {__builtin_amdgcn_rsqf(f.x), __builtin_amdgcn_rsqf(f.y)};
    const f32x2 s0 = f * y, h = f32x2{0.5f, 0.5f} * y;
    const f32x2 r = __builtin_elementwise_fma(-s0, s0, f);
    return __builtin_elementwise_fma(r, h, s0);
}
typedef uint32_t u32_align2 __attribute__((aligned(2)));  // dword load at a 16-bit sample boundary
typedef uint32_t u32x2_raw __attribute__((ext_vector_type(2)));
typedef u32x2_raw u32_pair_align4 __attribute__((aligned(4)));  // two consecutive table words at any word boundary (one 8-byte load)
typedef uint32_t u32x2_align2 __attribute__((ext_vector_type(2), aligned(2)));  // 8 bytes at a 16-bit sample boundary

// ---- packed 16+16-bit helpers (VOP3P): one instruction works on the real and imaginary halves ----
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_ashr(uint32_t a, int n)
{
    return __builtin_bit_cast(uint32_t, (short2v)(__builtin_bit_cast(short2v, a) >> (short2v){(short)n, (short)n}));
}
__device__ __forceinline__ uint32_t pk_lshr(uint32_t a, int n)
{
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) >> (u16x2){(unsigned short)n, (unsigned short)n}));
}
// a*c + b per half (v_pk_mad_u16); c is a packed constant such as (+1, -1)
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t c, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, c) +
                                                __builtin_bit_cast(u16x2, b)));
}
// swap(a)*c + b per half, swap = the two halves of a exchanged: the exchange rides the instruction's operand selects
// (op_sel / op_sel_hi of VOP3P: the low result takes a's HIGH half, the high result a's LOW half) instead of a v_alignbit_b32
__device__ __forceinline__ uint32_t pk_mad_swap(uint32_t a, uint32_t c, uint32_t b)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(c), "v"(b));
    return r;
}
// (hi16(lo_src) , hi16(hi_src)) -> packed word: the ">>16" of a complex 32-bit pair in ONE v_perm_b32
__device__ __forceinline__ uint32_t pk_hi16(int lo_src, int hi_src)
{
    return __builtin_amdgcn_perm((uint32_t)hi_src, (uint32_t)lo_src, 0x07060302u);
}
// ((lo_src >> 15) & 0xFFFF , (hi_src >> 15) & 0xFFFF) -> packed word: bits 15..30 of both sources.  One plain shift
// for the low half (its upper bits are overwritten next) and one SDWA shift that writes only word 1.
__device__ __forceinline__ uint32_t pk_s15(int lo_src, int hi_src)
{
    uint32_t r = (uint32_t)lo_src >> 15;
    // trailing s_nop 0: a VALU read of a register right after a partial (dst_sel != DWORD) SDWA write of it needs one wait
    // state on gfx940-class chips (LLVM's "dst_sel forwarding hazard"); the compiler inserts it for its own SDWA code but
    // cannot see into the asm
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\ts_nop 0"
        : "+v"(r)
        : "v"(15), "v"(hi_src));
    return r;
}

// Orders LDS traffic between lanes of ONE wave: DS instructions of a wave execute in issue order, so
// only the compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// DPP row shifts: lane i takes the value of lane i-N inside its row of 16, 0 when there is none.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// inclusive prefix sum over the 64 lanes of a wave (u32 wrap)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    v += dpp_take<0x111, 0xF>(v);  // row_shr:1
    v += dpp_take<0x112, 0xF>(v);  // row_shr:2
    v += dpp_take<0x114, 0xF>(v);  // row_shr:4
    v += dpp_take<0x118, 0xF>(v);  // row_shr:8
    // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3, accumulated IN PLACE: lanes of the rows the mask
    // disables keep their value, which is what the sum needs there (through the intrinsic the compiler spends a v_mov and a
    // v_mov_dpp on the "old" value of the disabled rows)
    // (s_nop 1: a DPP read needs two wait states after the VALU write of its source; the compiler cannot see into the asm)
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa" : "+v"(v));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(v));
    return v;
}

// (u32)(log((double)n)*100), MFCC.C:168, as a step function (see sr_tables.cpp gen_log_thr).
// v_log_f32 puts the estimate within one step of the answer for every u32 input (all 2^32 checked by
// tests/exhaustive_math_sweep.py), so one look at the two neighbouring thresholds settles it.  Branch-free, one pair of
// loads: the estimate is clamped to [1, kLogMax - 1] (still within one step of an answer in [0, kLogMax]), which also
// takes care of both ends -- n = 0 (log2 = -inf -> clamped to 1, n < thr[1] = 2 -> 0 as on ARM softfp / x86-64, where the
// UB cast of -inf gives 0) and the last step (its upper neighbour is the sentinel) -- and the two corrections are
// compare + add/subtract-with-carry.  (The branching form the compiler made of the two-sided if cost two dependent
// global loads with a wait each inside divergent control flow.)
__device__ __forceinline__ uint32_t log100_est(uint32_t n)  // the clamped estimate m: thr[m], thr[m + 1] are the thresholds to look at
{
    return (uint32_t)(int)__builtin_amdgcn_fmed3f(__log2f((float)n) * 69.31471806f, 1.0f, (float)(kLogMax - 1));
}
__device__ __forceinline__ uint32_t log100_fix(uint32_t n, uint32_t m, uint32_t t0, uint32_t t1)
{
    return (uint32_t)((int)m - (int)(n < t0) + (int)(n >= t1));
}
__device__ __forceinline__ uint32_t log100_u32(uint32_t n, const uint32_t *__restrict__ thr)
{
    const uint32_t m = log100_est(n);
    return log100_fix(n, m, thr[m], thr[m + 1]);
}

}  // namespace sr
