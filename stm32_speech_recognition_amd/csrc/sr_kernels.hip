// HIP kernels for gfx950 (MI355X, CDNA4).  Wave = 64 lanes everywhere; no MFMA (the path has no
// dense contraction), integer VALU + LDS.  Every kernel reproduces the reference's integer
// arithmetic bit for bit; the cited lines are relative to the reference tree.
//
//   k_vad     noise_atap (VAD.C:22-71) + VAD (VAD.C:97-218) + frame count of get_mfcc (MFCC.C:102-107)
//   k_mfcc    get_mfcc (MFCC.C:86-191) incl. fft (MFCC.C:27-62) and cr4_fft_1024_stm32 (.s:95-281)
//   k_dtw     dtw / get_dis / dtw_limit (DTW.C:45-192) for every (utterance, template) pair
//   k_argmin  the template scan of spch_recg (main.c:276-295)
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

// ------------------------------------------------------------------------------------------------
// Q15 radix-4 butterfly of the ST FFT (cr4_fft_1024_stm32.s)
// ------------------------------------------------------------------------------------------------
// CXMUL_V7 (.s:95-102): Y*conj(K), Q14.  The asm's 3-multiply form equals, in the ring of 32-bit
// integers, re = Yr*Kc + Yi*Ks, im = Yi*Kc - Yr*Ks with Kc = Kr'+Ki, Ks = Ki (no term overflows).
__device__ __forceinline__ void cxmul(uint32_t y, uint32_t ka, uint32_t kb, int &re, int &im)
{
    re = sdot2z(y, ka);
    im = sdot2z(y, kb);
}

// CXADDA4 (.s:105-129, S = 14) and the combine of BUTFLY4ZERO_OPT (.s:147-168, S = 0).
// In: A (sign-extended sample), B, C, D (products or samples).  Out as stored by the asm:
//   x[j] = (ar, ai)   x[j+q] = (br, bi)   x[j+2q] = (cr, ci)   x[j+3q] = (di, dr)   <- "inversion"
template <int S>
__device__ __forceinline__ void r4_combine(int &ar, int &ai, int &br, int &bi, int &cr, int &ci, int &dr, int &di)
{
    int tr = cr + dr, ti = ci + di;  // (C,D) = (C+D, C-D)
    dr = cr - dr;
    di = ci - di;
    cr = tr;
    ci = ti;
    ar >>= 2;
    ai >>= 2;
    ar += br >> (2 + S);
    ai += bi >> (2 + S);
    br = ar - (br >> (1 + S));
    bi = ai - (bi >> (1 + S));
    ar += cr >> (2 + S);
    ai += ci >> (2 + S);
    cr = ar - (cr >> (1 + S));
    ci = ai - (ci >> (1 + S));
    br += di >> (2 + S);
    bi -= dr >> (2 + S);
    di = br - (di >> (1 + S));
    dr = bi + (dr >> (1 + S));
}

// One twiddled butterfly on packed words; k* = packed coefficient pairs for the legs j+q, j+2q, j+3q.
// CXADDA4 (.s:105-129) in packed 16-bit arithmetic.  Every value the asm stores is the low 16 bits of a
// 32-bit sum of terms (A>>2), (X>>16), (X>>15); sums mod 2^16 may be taken in any order.  With h(X) = X>>16 and
// p(X) = X>>15 (both halves at once, mod 2^16) and (X>>15) = 2*(X>>16) + bit15(X):
//   A1 = a + h(B)                B1 = A1 - (B>>15)                    = A1 - p(B)
//   A2 = A1 + h(C')   = x[j]     C2 = A2 - (C'>>15)                   = x[j] - p(C')      = x[j+2q]
//   B2 = B1 + S*h(D'~) = x[j+q]  D2 = B2 - S*(D'>>15)~                = x[j+q] - S*p(D'~) = x[j+3q]
// S = (+1,-1), ~ = halves swapped (.s:125-128: Br += Di>>16, Bi -= Dr>>16, Di = Br - Di>>15, Dr = Bi + Dr>>15, stored
// as (Di, Dr)).  Each ">>15" output is therefore ONE packed op on the matching ">>16" output; h() is one v_perm_b32,
// p() a plain shift + one SDWA shift: 16 VALU for the combine of a full butterfly (19 with the bit-15 form of round 1).
constexpr uint32_t kPkPlusMinus = 0xFFFF0001u;  // (+1, -1)
constexpr uint32_t kPkMinusPlus = 0x0001FFFFu;  // (-1, +1)

// combine step on 32-bit products B, C' = C+D, D' = C-D and the packed sample A
// CD_SAME: the D leg is zero, so C' = D' = C and the swapped D' halves are the C' halves rotated by 16 bits
template <bool HALF, bool HAS_B, bool CD_SAME = false>
__device__ __forceinline__ void r4_packed(uint32_t x0_in, int br, int bi, int sr, int si, int tr, int ti, uint32_t &x0,
                                          uint32_t &x1, uint32_t &x2, uint32_t &x3)
{
    const uint32_t a = pk_ashr(x0_in, 2);
    uint32_t A1 = a, B1 = a;
    if (HAS_B) {
        A1 = pk_add(a, pk_hi16(br, bi));
        B1 = pk_sub(A1, pk_s15(br, bi));
    }
    const uint32_t hC = pk_hi16(sr, si);
    const uint32_t hD = CD_SAME ? __builtin_amdgcn_alignbit(hC, hC, 16) : pk_hi16(ti, tr);  // swapped: (D'i>>16, D'r>>16)
    x0 = pk_add(A1, hC);
    x1 = pk_mad(hD, kPkPlusMinus, B1);
    if (!HALF) {
        const uint32_t pC = pk_s15(sr, si), pD = CD_SAME ? __builtin_amdgcn_alignbit(pC, pC, 16) : pk_s15(ti, tr);
        x2 = pk_sub(x0, pC);
        x3 = pk_mad(pD, kPkMinusPlus, x1);
    }
}

template <bool HALF>  // HALF: only x0, x1 are produced (last pass: bins >= 512 are never read, MFCC.C:49)
__device__ __forceinline__ void bfly_pk(uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t &x3, uint32_t k1a,
                                        uint32_t k1b, uint32_t k2a, uint32_t k2b, uint32_t k3a, uint32_t k3b,
                                        uint32_t k3na, uint32_t k3nb)
{
    int br, bi, cr, ci;
    cxmul(x2, k2a, k2b, cr, ci);
    cxmul(x1, k1a, k1b, br, bi);
    // C' = C + D and D' = C - D without ever forming D: D = x3*conj(K3) is accumulated onto C with K3 and -K3
    const int sr = sdot2a(x3, k3a, cr), si = sdot2a(x3, k3b, ci);
    const int tr = sdot2a(x3, k3na, cr), ti = sdot2a(x3, k3nb, ci);
    r4_packed<HALF, true>(x0, br, bi, sr, si, tr, ti, x0, x1, x2, x3);
}

__device__ __forceinline__ uint32_t pk_neg(uint32_t w) { return pk_sub(0u, w); }

__device__ __forceinline__ void bfly(uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t &x3, uint32_t k1a, uint32_t k1b,
                                     uint32_t k2a, uint32_t k2b, uint32_t k3a, uint32_t k3b)
{
    bfly_pk<false>(x0, x1, x2, x3, k1a, k1b, k2a, k2b, k3a, k3b, pk_neg(k3a), pk_neg(k3b));  // generic paths: negate on the fly
}

__device__ __forceinline__ int rev2(int d) { return ((d & 1) << 1) | (d >> 1); }

// LDS image of the 1024-point work array between passes 3 and 4: word j lives at j + 4*(j>>6).
// With lane = d0 + 4*d3 + 16*d4 writing j = d0 + 4*d1 + 16*d2 + 64*d3 + 256*d4 the 32 lanes of a
// ds_write_b32 group hit 32 distinct banks; reads by lane = j & 63 are consecutive words.
__device__ __forceinline__ int xaddr(int j) { return j + ((j >> 6) << 2); }
constexpr int kXchgWords = 1024 + 4 * 16;  // 1088

// Coefficients a lane needs, all lane-invariant across frames -> loaded once per wave into VGPRs.
struct LaneTw {
    uint32_t s2[2][2];     // pass 2 (q=4):   legs j+q, j+2q (j+3q is always zero-padding)
    uint32_t s3[4][4][2];  // pass 3 (q=16):  per d1, legs j+q, j+2q, j+3q, and the j+3q pair negated
    uint32_t s4[4][2];     // pass 4 (q=64)
    uint32_t s5[4][4][2];  // pass 5 (q=256): per d3
};

// Table entry order per butterfly is (leg j+3q, leg j+2q, leg j+q)  (.s:182-191).
__device__ __forceinline__ void load_tw3(const DevTables &t, int base, int b, uint32_t (&k)[3][2])
{
    const int e = base + 3 * b;
    k[0][0] = t.tw_a[e + 2];
    k[0][1] = t.tw_b[e + 2];  // leg j+q
    k[1][0] = t.tw_a[e + 1];
    k[1][1] = t.tw_b[e + 1];  // leg j+2q
    k[2][0] = t.tw_a[e + 0];
    k[2][1] = t.tw_b[e + 0];  // leg j+3q
}

// legs j+q, j+2q, j+3q as above + entry [3] = the j+3q pair negated (for C+D / C-D by accumulation)
__device__ __forceinline__ void load_tw4(const DevTables &t, int base, int b, uint32_t (&k)[4][2])
{
    uint32_t k3[3][2];
    load_tw3(t, base, b, k3);
#pragma unroll
    for (int e = 0; e < 3; e++) {
        k[e][0] = k3[e][0];
        k[e][1] = k3[e][1];
    }
    k[3][0] = pk_sub(0u, k3[2][0]);
    k[3][1] = pk_sub(0u, k3[2][1]);
}

__device__ __forceinline__ void load_lane_tw(const DevTables &t, int lane, LaneTw &tw)
{
    const int d0 = lane & 3;
    {
        uint32_t k[3][2];
        load_tw3(t, 0, d0, k);
        tw.s2[0][0] = k[0][0];
        tw.s2[0][1] = k[0][1];
        tw.s2[1][0] = k[1][0];
        tw.s2[1][1] = k[1][1];
    }
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++) load_tw4(t, 12, d0 + 4 * d1, tw.s3[d1]);
    load_tw4(t, 60, lane, tw.s4);
#pragma unroll
    for (int d3 = 0; d3 < 4; d3++) load_tw4(t, 252, lane + 64 * d3, tw.s5[d3]);
}

// 32 lane-invariant coefficient words (4 entries x 2 words per butterfly, 4 butterflies) parked in LDS as
// 8 x 16-byte chunks, chunk c of lane l at [c*64 + l] (consecutive lanes -> consecutive 16-byte slots:
// conflict-free ds_read_b128 / ds_write_b128)
__device__ __forceinline__ void store_tw32(u32x4 *lds, int lane, const uint32_t (&k)[4][4][2])
{
    const uint32_t *f = &k[0][0][0];
#pragma unroll
    for (int c = 0; c < 8; c++) lds[c * 64 + lane] = u32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
}
__device__ __forceinline__ void load_tw32(const u32x4 *lds, int lane, uint32_t (&k)[4][4][2])
{
    uint32_t *f = &k[0][0][0];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const u32x4 q = lds[c * 64 + lane];
        f[4 * c] = q.x;
        f[4 * c + 1] = q.y;
        f[4 * c + 2] = q.z;
        f[4 * c + 3] = q.w;
    }
}
// pass-3 coefficients depend on (d0, d1) only: 4 x 32 words, read with lane-broadcast by d0 = lane & 3
__device__ __forceinline__ void load_tw32_d0(const u32x4 *lds, int d0, uint32_t (&k)[4][4][2])
{
    uint32_t *f = &k[0][0][0];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const u32x4 q = lds[d0 * 8 + c];
        f[4 * c] = q.x;
        f[4 * c + 1] = q.y;
        f[4 * c + 2] = q.z;
        f[4 * c + 3] = q.w;
    }
}

// Passes 1-3 for the zero-padded real frame that get_mfcc feeds (MFCC.C:37-47): only x[0..159] are
// non-zero and every imaginary part is 0.  Pass 1 (.s:226-232) then degenerates exactly to
// out[4*idx+k] = x[bitrev8(idx)] >> 2 (k = 0..3), so it is folded into the gather.
// lane = d0 + 4*d3 + 16*d4 ; v[d1][d2] <-> j = d0 + 4*d1 + 16*d2 + 64*d3 + 256*d4.
__device__ __forceinline__ void fft_front_real160(const uint16_t *xw, int lane, const LaneTw &tw, const u32x4 *tw3_lds,
                                                  uint32_t (&v)[4][4])
{
    const int d3 = (lane >> 2) & 3, d4 = lane >> 4;
    const int base = rev2(d4) + 4 * rev2(d3);
    // bitrev8(j>>2) = base + 16*rev2(d2) + 64*rev2(d1); >= 160 <=> zero padding
    uint32_t y[10];
#pragma unroll
    for (int m = 0; m < 10; m++) y[m] = xw[base + 16 * m];  // already A >> 2 as a 16-bit pattern (pass 1, .s:147-148)
#pragma unroll
    for (int d2 = 0; d2 < 4; d2++) {
        const int r2 = ((d2 & 1) << 1) | (d2 >> 1);
        uint32_t x0 = y[r2];                                      // d1 = 0 -> rows   0..63
        uint32_t x2 = y[4 + r2];                                  // d1 = 2 -> rows  64..127
        uint32_t x1 = (d2 == 0) ? y[8] : (d2 == 2) ? y[9] : 0u;   // d1 = 1 -> rows 128..159, else padding
        uint32_t x3 = 0u;                                         // d1 = 3 -> rows >= 192: padding
        // real samples (imaginary half of the packed word is 0): the general Y*conj(K) dot products give
        // (Yr*Kc, -Yr*Ks) directly, no sign extension needed; D = 0 => C' = D' = C
        int cr, ci;
        cxmul(x2, tw.s2[1][0], tw.s2[1][1], cr, ci);
        (void)x3;
        if (d2 == 0 || d2 == 2) {
            int br, bi;
            cxmul(x1, tw.s2[0][0], tw.s2[0][1], br, bi);
            r4_packed<false, true, true>(x0, br, bi, cr, ci, cr, ci, v[0][d2], v[1][d2], v[2][d2], v[3][d2]);
        } else {
            r4_packed<false, false, true>(x0, 0, 0, cr, ci, cr, ci, v[0][d2], v[1][d2], v[2][d2], v[3][d2]);
        }
    }
    // pass-3 coefficients (24 words per lane) are parked in LDS, shared by the workgroup's waves
    uint32_t k3[4][4][2];
    load_tw32_d0(tw3_lds, lane & 3, k3);
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
        bfly_pk<false>(v[d1][0], v[d1][1], v[d1][2], v[d1][3], k3[d1][0][0], k3[d1][0][1], k3[d1][1][0], k3[d1][1][1],
                       k3[d1][2][0], k3[d1][2][1], k3[d1][3][0], k3[d1][3][1]);
}

// lane (d0,d3,d4) -> LDS -> lane' = j & 63 holding u[d3][d4]
__device__ __forceinline__ void fft_exchange(uint32_t *buf, int lane, const uint32_t (&v)[4][4], uint32_t (&u)[4][4])
{
    const int d0 = lane & 3, d3 = (lane >> 2) & 3, d4 = lane >> 4;
    const int jw = d0 + 64 * d3 + 256 * d4;
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
#pragma unroll
        for (int d2 = 0; d2 < 4; d2++) buf[xaddr(jw + 4 * d1 + 16 * d2)] = v[d1][d2];
    wave_sync();
#pragma unroll
    for (int e3 = 0; e3 < 4; e3++)
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) u[e3][e4] = buf[xaddr(lane + 64 * e3 + 256 * e4)];
}

// ------------------------------------------------------------------------------------------------
// k_mfcc
// ------------------------------------------------------------------------------------------------
constexpr int kMfccWaves = 4;       // waves per workgroup
constexpr int kFramesPerWave = 16;   // consecutive frames one wave turns into MFCCs per work item
constexpr int kFramesPerTile = kMfccWaves * kFramesPerWave;
// per-wave LDS: exchange/scratch words + windowed frame + filterbank outputs of the wave's frames
// rows of the filterbank outputs and of the DCT tables are kMelPad = 25 words apart: in the DCT the lanes of a wave read
// 6 different frames x 12 different coefficients rows at the same column, and a stride of 24 folds those onto 4 banks
constexpr int kMelPad = kMel + 1;
constexpr int kWaveLdsWords = kXchgWords + kFramesPerWave * kMelPad + 64;  // the windowed frame aliases the exchange area;
                                                                        // the last 64 words hold the odd filters' lane offsets

// (u32)(log((double)n)*100), MFCC.C:168, as a step function (see sr_tables.cpp gen_log_thr).
// v_log_f32 puts the estimate within one step of the answer for every u32 input (all 2^32 checked by
// tests/exhaustive_math_sweep.py), so one look at the two neighbouring thresholds settles it.  Branch-free, one pair of
// loads: the estimate is clamped to [1, kLogMax - 1] (still within one step of an answer in [0, kLogMax]), which also
// takes care of both ends -- n = 0 (log2 = -inf -> clamped to 1, n < thr[1] = 2 -> 0 as on ARM softfp / x86-64, where the
// UB cast of -inf gives 0) and the last step (its upper neighbour is the sentinel) -- and the two corrections are
// compare + add/subtract-with-carry.  (The branching form the compiler made of the two-sided if cost two dependent
// global loads with a wait each inside divergent control flow.)
__device__ __forceinline__ uint32_t log100_u32(uint32_t n, const uint32_t *__restrict__ thr)
{
    const float e = __builtin_amdgcn_fmed3f(__log2f((float)n) * 69.31471806f, 1.0f, (float)(kLogMax - 1));
    const int m = (int)e;
    const uint32_t t0 = thr[m], t1 = thr[m + 1];
    return (uint32_t)(m - (int)(n < t0) + (int)(n >= t1));
}

__global__ void __launch_bounds__(64 * kMfccWaves, 4) k_mfcc(const MfccArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_dctM[kCoef * kMelPad];
    __shared__ int s_dctS[kCoef * kMelPad];  // 32-bit: read with the wide LDS loads, no byte extraction
    __shared__ u32x4 s_tw3[8 * 4], s_tw5[8 * 64];  // pass-3 (per d0) / pass-5 (per lane) coefficients, shared by the waves
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *buf = smem + w * kWaveLdsWords;
    uint16_t *xw = (uint16_t *)buf;  // windowed frame: consumed by the pass-1 gather before the exchange overwrites it
    uint32_t *powb = buf + kXchgWords, *moff = powb + kFramesPerWave * kMelPad;

    // DCT term (MFCC.C:179): (s32)pow * dct / 100, truncated toward zero, with 0 <= pow <= 2218 (= (u32)(ln(2^32)*100))
    // and |dct| <= 128.  floor(pow*|c|/100) == (pow * M_c) >> 18 with M_c = ceil(|c| * 2^18 / 100) for every such pair
    // (the rounding excess pow*eps/2^18 stays below 1/100 because 100*pow < 2^18); the log stage stores pow << 14 so
    // the quotient is one v_mul_hi_u32, and the sign of c is applied by the accumulating 24-bit multiply.
    for (int i = threadIdx.x; i < kCoef * kMel; i += blockDim.x) {
        const int c = a.t.dct[i], o = (i / kMel) * kMelPad + i % kMel;
        s_dctM[o] = (uint32_t)(((c < 0 ? -c : c) * 262144 + 99) / 100);
        s_dctS[o] = (c > 0) - (c < 0);
    }
    __syncthreads();

    // ---- lane-invariant constants --------------------------------------------------------------
    LaneTw tw;
    load_lane_tw(a.t, lane, tw);
    if (w == 0 && lane < 4) {
        const uint32_t *f = &tw.s3[0][0][0];
#pragma unroll
        for (int c = 0; c < 8; c++) s_tw3[lane * 8 + c] = u32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
    }
    if (w == 1 % kMfccWaves) store_tw32(s_tw5, lane, tw.s5);
    __syncthreads();
    int hamm_r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) hamm_r[k] = (lane + 64 * k < kFrameLen) ? (int)a.t.hamm[lane + 64 * k] : 0;
    // triangle weights of bins 8*lane .. 8*lane+7 of both poly-lines: 16 registers for the whole kernel
    uint32_t tri_e[8], tri_o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        tri_e[k] = a.t.tri_even32[8 * lane + k];
        tri_o[k] = a.t.tri_odd32[8 * lane + k];
    }
    // filter h < 24 owned by lane h: bins [lo, hi) of poly-line (h & 1)  (MFCC.C:136-162)
    int f_lo = 0, f_hi = 0;
    if (lane < kMel) {
        const int h = lane;
        f_lo = (h == 0) ? 0 : (int)a.t.tri_cen[h - 1];
        f_hi = (h == kMel - 1) ? kBins : (int)a.t.tri_cen[h + 1];
    }

    // the per-utterance record of the NEXT work item is fetched while the current one is processed
    uint32_t item = blockIdx.x;
    uint32_t nx_nfrm = 0;
    int nx_mid = 0, nx_seg0 = 0;
    if (item < a.n_items) {
        const sr_vad_rec *rec = a.vad + item / a.tiles;
        nx_nfrm = rec->frm_num;
        nx_mid = (int)rec->atap.mid_val;
        nx_seg0 = rec->seg[0];
    }
    for (; item < a.n_items; item += gridDim.x) {
        const uint32_t b = item / a.tiles, tile = item - b * a.tiles;
        const uint32_t nfrm = nx_nfrm;
        const int mid = nx_mid, seg0 = nx_seg0;
        if (item + gridDim.x < a.n_items) {
            const sr_vad_rec *rec = a.vad + (item + gridDim.x) / a.tiles;
            nx_nfrm = rec->frm_num;
            nx_mid = (int)rec->atap.mid_val;
            nx_seg0 = rec->seg[0];
        }
        const uint16_t *row = a.pcm + (uint64_t)b * a.pcm_stride;
        int16_t *out = a.mfcc + (uint64_t)b * a.max_frames * kCoef;
        const uint32_t f0 = tile * kFramesPerTile + w * kFramesPerWave;
        uint32_t nf = 0;  // frames this wave really has
        if (f0 < nfrm) nf = (nfrm - f0 < (uint32_t)kFramesPerWave) ? nfrm - f0 : (uint32_t)kFramesPerWave;

        // samples of frame fi+1 are requested while frame fi is transformed
        // one 2-byte-aligned dword per sample: x[i-1] in the low half, x[i] in the high half
        uint32_t s_pp[3] = {0, 0, 0};
        if (nf) {
            const uint16_t *x = row + seg0 + (int)kHop * (int)f0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int i = lane + 64 * k;
                if (i < kFrameLen) s_pp[k] = *(const u32_align2 *)(x + i - 1);
            }
        }
        for (uint32_t fi = 0; fi < nf; fi++) {
            // ---- pre-emphasis + Hamming (MFCC.C:115-124); x[-1] is the sample before the frame
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int i = lane + 64 * k;
                if (i < kFrameLen) {
                    const int cur = (int)(s_pp[k] >> 16) - mid, prv = (int)(s_pp[k] & 0xFFFFu) - mid;
                    const int t = cur - preemph95(prv);
                    // stored as the pass-1 output A >> 2 of the s16 sample (16-bit LDS store; the gather zero-extends)
                    xw[i] = (uint16_t)((int)(short)(mul24(t, hamm_r[k]) / 1000) >> 2);
                }
            }
            if (fi + 1 < nf) {
                const uint16_t *x = row + seg0 + (int)kHop * (int)(f0 + fi + 1);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int i = lane + 64 * k;
                    if (i < kFrameLen) s_pp[k] = *(const u32_align2 *)(x + i - 1);
                }
            }
            wave_sync();
            // ---- FFT passes 1-3 in registers, exchange, passes 4-5 in registers
            uint32_t v[4][4], u[4][4];
            fft_front_real160(xw, lane, tw, s_tw3, v);
            fft_exchange(buf, lane, v, u);
#pragma unroll
            for (int e4 = 0; e4 < 4; e4++)
                bfly_pk<false>(u[0][e4], u[1][e4], u[2][e4], u[3][e4], tw.s4[0][0], tw.s4[0][1], tw.s4[1][0], tw.s4[1][1],
                               tw.s4[2][0], tw.s4[2][1], tw.s4[3][0], tw.s4[3][1]);
            // pass 5: only x[j] and x[j+q] (bins < 512) are consumed (MFCC.C:49)
            wave_sync();
            uint32_t k5[4][4][2];
            load_tw32(s_tw5, lane, k5);
#pragma unroll
            for (int e3 = 0; e3 < 4; e3++) {
                bfly_pk<true>(u[e3][0], u[e3][1], u[e3][2], u[e3][3], k5[e3][0][0], k5[e3][0][1], k5[e3][1][0],
                              k5[e3][1][1], k5[e3][2][0], k5[e3][2][1], k5[e3][3][0], k5[e3][3][1]);
                // ---- |X|*10 and energy (MFCC.C:49-60, 128-133) on the stored 16-bit halves
                {
                    // re*re + im*im from the packed word; both bins' roots and the x10 in packed f32 operations (plain IEEE
                    // multiplies and fused multiply-adds, see sqrt_rn_int)
                    const f32x2 m = sqrt_rn_int2(f32x2{(float)sdot2z(u[e3][0], u[e3][0]), (float)sdot2z(u[e3][1], u[e3][1])}) *
                                    f32x2{10.0f, 10.0f};
                    const uint32_t m0 = cvt_u32(m.x), m1 = cvt_u32(m.y);  // < 2^19
                    buf[lane + 64 * e3] = umul24(m0, m0);
                    buf[lane + 64 * e3 + 256] = umul24(m1, m1);
                }
            }
            wave_sync();
            // ---- Mel filterbank as prefix sums over bins (each term /100 before summing, u32 wrap)
            uint32_t pe[8], po[8], xe, xo;
            {
                const uint4 q0 = *(const uint4 *)(buf + 8 * lane), q1 = *(const uint4 *)(buf + 8 * lane + 4);
                const uint32_t e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                uint32_t se = 0, so = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    se += e[k] * tri_e[k] / 100u;
                    so += e[k] * tri_o[k] / 100u;
                    pe[k] = se;
                    po[k] = so;
                }
                xe = wave_scan_incl(se) - se;  // sum over the bins of the lanes below
                xo = wave_scan_incl(so) - so;
            }
            wave_sync();
            // in-lane prefixes and the per-lane offsets are stored separately: the 24 filter lanes add them on lookup
            // (2 adds) instead of every lane adding its offset to 16 prefixes
            *(uint4 *)(buf + 8 * lane) = make_uint4(pe[0], pe[1], pe[2], pe[3]);
            *(uint4 *)(buf + 8 * lane + 4) = make_uint4(pe[4], pe[5], pe[6], pe[7]);
            *(uint4 *)(buf + kBins + 8 * lane) = make_uint4(po[0], po[1], po[2], po[3]);
            *(uint4 *)(buf + kBins + 8 * lane + 4) = make_uint4(po[4], po[5], po[6], po[7]);
            buf[2 * kBins + lane] = xe;
            moff[lane] = xo;
            wave_sync();
            if (lane < kMel) {
                const uint32_t *P = buf + ((lane & 1) ? kBins : 0), *X = (lane & 1) ? moff : buf + 2 * kBins;
                const int ih = f_hi - 1, il = f_lo - 1;
                const uint32_t hi = P[ih] + X[ih >> 3], lo = f_lo ? P[il] + X[il >> 3] : 0u;
                powb[fi * kMelPad + lane] = hi - lo;
            }
            wave_sync();
        }

        // ---- log (MFCC.C:165-170) and DCT (MFCC.C:173-183) for the wave's nf frames, all lanes busy
        // (the pad word of each row goes through the log as well: harmless, never read)
        for (uint32_t t = lane; t < nf * kMelPad; t += 64) powb[t] = log100_u32(powb[t], a.t.log_thr) << 14;
        wave_sync();
        // output t = fi*12 + h of the wave's tile goes to out[(f0 + fi)*12 + h] = out_w[t]: consecutive lanes store
        // consecutive s16; fi = t / 12 by a 24-bit multiply (exact for t < 2^13), all index arithmetic in 32 bits
        // (left to the compiler the 64-bit subscript became eight v_mad_u64_u32 per round)
        {
            int16_t *out_w = out + (size_t)f0 * kCoef;
#pragma unroll
            for (uint32_t t = lane; t < (uint32_t)(kFramesPerWave * kCoef); t += 64) {
                if (t < nf * kCoef) {
                    const uint32_t fi = umul24(t, 10923u) >> 17, h = t - umul24(fi, (uint32_t)kCoef);
                    const uint32_t *pw = powb + umul24(fi, (uint32_t)kMelPad), *dm = s_dctM + umul24(h, (uint32_t)kMelPad);
                    const int *ds = s_dctS + umul24(h, (uint32_t)kMelPad);
                    int acc = 0;
#pragma unroll
                    for (int i = 0; i < kMel; i++) acc = mad24((int)__umulhi(pw[i], dm[i]), ds[i], acc);
                    out_w[t] = (int16_t)acc;
                }
            }
        }
        wave_sync();
        // rows >= frm_num of this tile are zeroed so that every row of the output is defined
        {
            const uint32_t r0 = f0 + nf, r1 = (f0 + kFramesPerWave < a.max_frames) ? f0 + kFramesPerWave : a.max_frames;
            for (uint32_t t = r0 * kCoef + lane; t < r1 * kCoef && r0 < r1; t += 64) out[t] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_mfcc_ext: EXTENSION front end (BASELINE.json configs[4]: 16 kHz, 20/10 ms framing = 320/160 samples,
// 512-point transform, 40 Mel filters, 12 coefficients).  No reference counterpart: the reference's FFT only
// converts 1024 points (.s:214-215).  Same arithmetic rules as get_mfcc (MFCC.C:86-191) with the tables generated
// from the same Matlab formulas at fs = 16000; the 512-point transform is two ST-style 256-point radix-4
// transforms (even / odd samples) + one truncating radix-2 pass, as defined in oracle/q15_fft.c.
//
// Register-resident like k_mfcc: a wave works on FOUR frames at once, 16 lanes per frame, each lane holding 16 points
// of both 256-point sub-transforms.  With j = d0 + 4*d1 + 16*d2 + 64*d3 the index of a point after pass 1:
//   layout A  lane = (d2, d3)  holds v[d0][d1]: passes 1 and 2 (digits d0 / d1) in registers.  Pass 1 reads
//             src[bitrev6(j >> 2) + 64 m] = the samples i == 2*base (+1) (mod 32), base = rev2(d3) + 4*rev2(d2): exactly
//             the 20 samples this lane windows itself, so there is no LDS gather at all; legs >= 160 are zero padding.
//   exchange  one pass through LDS (element e = d0 + 4*d1 of lane l at e*65 + l: writes and reads conflict-free)
//   layout B  lane = (d0, d1)  holds u[d2][d3]: passes 3 and 4 (digits d2 / d3) and the radix-2 pass E[k] +- O[k]W[k]
//             (both sub-transforms of a bin live in the same lane) in registers, then |X|*10 and the energy.
// The energies are transposed through LDS to 16 contiguous bins per lane for the filterbank prefix sums (row scans
// over the frame's 16 lanes), log and DCT are batched over the wave's frames as in k_mfcc.
// ------------------------------------------------------------------------------------------------
namespace ext {
constexpr int kFL = 320, kHopE = 160, kBinsE = 256, kMelE = 40;
constexpr int kWaves = 4, kGrp = 4, kFpw = 8, kTile = kWaves * kFpw;  // kGrp frames in flight per wave
constexpr int kXStride = 65;                  // exchange image: element e of lane l at e*65 + l
constexpr int kXSub = 16 * kXStride;          // one sub-transform
constexpr int kXWords = 2 * kXSub;            // 2080; later reused for energies and prefix sums
constexpr int kEStride = 336;                 // energies of one frame: bin k at k + 4*(k >> 4), frames 336 words apart
constexpr int kMelEPad = kMelE + 1;             // row stride of the filterbank outputs / DCT tables (see kMelPad)
constexpr int kWaveWords = kXWords + 2 * 16 * kGrp + kFpw * kMelEPad;  // + prefix lane offsets + filterbank outputs
static_assert(kFL == 2 * 16 * 10, "a 16-lane frame group windows 10 sample pairs per lane");
static_assert(kGrp * kEStride <= kXWords && kGrp * 2 * kBinsE <= kXWords, "energies / prefix sums reuse the exchange image");
}  // namespace ext

// inclusive prefix sum inside each row of 16 lanes
__device__ __forceinline__ uint32_t row_scan_incl(uint32_t v)
{
    v += dpp_take<0x111, 0xF>(v);  // row_shr:1
    v += dpp_take<0x112, 0xF>(v);  // row_shr:2
    v += dpp_take<0x114, 0xF>(v);  // row_shr:4
    v += dpp_take<0x118, 0xF>(v);  // row_shr:8
    return v;
}

// Occupancy: 3 waves per SIMD (156 VGPRs, 50 KB of LDS per 4-wave workgroup).  Round 3 built the 4-waves-per-SIMD form the
// round-2 review asked for (8-wave workgroups, one batch per item, lane constants re-read per batch: 126 VGPRs, 2 x 80 KB of
// LDS per CU): mean waves per SIMD 2.74 -> 3.43, but the kernel alone stayed at 14.5 ms (its waits are the texture
// addresser's, not latency that more waves would hide) and the pipelined step got SLOWER (51.5 vs 50.6 ms) because the
// two workgroups took the whole LDS of a CU and the DTW workgroups of the other streams could no longer co-reside.
__global__ void __launch_bounds__(64 * ext::kWaves, 3) k_mfcc_ext(const MfccArgs a)
{
    using namespace ext;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_dctM[kCoef * kMelEPad];  // same exact-division-by-100 device as k_mfcc (see there)
    __shared__ int s_dctS[kCoef * kMelEPad];
    __shared__ u32x4 s_tw4[8 * 16], s_w512[8 * 16], s_tri[8 * 16];  // per-lane constants of layout B, chunk c of lane l at [c*16 + l]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, gl = lane & 15;
    uint32_t *xb = smem + w * kWaveWords;
    uint32_t *moff = xb + kXWords, *powb = moff + 2 * 16 * kGrp;
    for (int i = threadIdx.x; i < kCoef * kMelE; i += blockDim.x) {
        const int c = a.t.dct[i], o = (i / kMelE) * kMelEPad + i % kMelE;
        s_dctM[o] = (uint32_t)(((c < 0 ? -c : c) * 262144 + 99) / 100);
        s_dctS[o] = (c > 0) - (c < 0);
    }
    // ---- constants of layout A: lane = (d2, d3) --------------------------------------------------
    const int base = rev2(gl >> 2) + 4 * rev2(gl & 3);
    uint32_t hp[10];  // Hamming weights of the lane's sample pairs (2*base + 32 t, +1)
#pragma unroll
    for (int t = 0; t < 10; t++) hp[t] = a.t.hamm_pk[base + 16 * t];
    // ---- constants of layout B: lane = (d0, d1), j = gl + 16*d2 + 64*d3 ----------------------------
    uint32_t k3[4][2];  // pass 3 (q = 16, coefficient block N = 64): index j & 15 = gl
    load_tw4(a.t, 12, gl, k3);
    if (w == 0 && lane < 16) {  // pass 4 (q = 64, block N = 256): index j & 63 = gl + 16*d2
#pragma unroll
        for (int d2 = 0; d2 < 4; d2++) {
            uint32_t k[4][2];
            load_tw4(a.t, 60, gl + 16 * d2, k);
            s_tw4[(2 * d2) * 16 + gl] = u32x4{k[0][0], k[0][1], k[1][0], k[1][1]};
            s_tw4[(2 * d2 + 1) * 16 + gl] = u32x4{k[2][0], k[2][1], k[3][0], k[3][1]};
        }
    }
    if (w == 1 % kWaves && lane < 16) {  // radix-2 coefficients of the lane's bins k = gl + 16 m
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int k0 = gl + 16 * (2 * c), k1 = gl + 16 * (2 * c + 1);
            s_w512[c * 16 + gl] = u32x4{a.t.w512_a[k0], a.t.w512_b[k0], a.t.w512_a[k1], a.t.w512_b[k1]};
        }
    }
    if (w == 2 % kWaves && lane < 16) {  // triangle weights of the bins 16*gl .. 16*gl + 15 (filterbank layout)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int b0 = 16 * gl + 2 * c;
            s_tri[c * 16 + gl] = u32x4{a.t.tri_even[b0], a.t.tri_odd[b0], a.t.tri_even[b0 + 1], a.t.tri_odd[b0 + 1]};
        }
    }
    // filters h = gl, gl + 16, gl + 32 (< 40) of the lane's frame: bins [lo, hi) of poly-line h & 1 (MFCC.C:136-162)
    uint32_t f_lohi[3];  // f_lo << 16 | f_hi (both <= 256)
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int h = gl + 16 * q;
        const int lo = (h == 0 || h >= kMelE) ? 0 : (int)a.t.tri_cen[h - 1];
        const int hi = (h >= kMelE) ? 1 : (h == kMelE - 1) ? kBinsE : (int)a.t.tri_cen[h + 1];
        f_lohi[q] = ((uint32_t)lo << 16) | (uint32_t)hi;
    }
    __syncthreads();

    // Work items are (utterance, tile of kTile frames); the records of the next two items are read ahead, and the
    // samples of the NEXT batch of four frames (same item, or the first batch of the next item that has frames) are
    // requested before the current batch is transformed, so the loads have a whole batch of arithmetic to land.
    struct Item {
        const uint16_t *row;  // the capture buffer the item belongs to
        int s0;               // sample index (in that buffer) of the wave's first frame
        int mid;
        uint32_t nf;         // frames this wave has in the item
    };
    auto item_info = [&](uint32_t it) {
        Item r{nullptr, 0, 0, 0u};
        if (it < a.n_items) {
            const uint32_t bb = it / a.tiles, tl = it - bb * a.tiles;
            const sr_vad_rec *rec = a.vad + bb;
            const uint32_t nfrm = rec->frm_num, ff = tl * kTile + w * kFpw;
            r.mid = (int)rec->atap.mid_val;
            r.row = a.pcm + (uint64_t)bb * a.pcm_stride;
            r.s0 = rec->seg[0] + kHopE * (int)ff;
            if (ff < nfrm) r.nf = (nfrm - ff < (uint32_t)kFpw) ? nfrm - ff : (uint32_t)kFpw;
        }
        // wave-uniform (they depend on the wave's index only), but loaded through the vector memory path: moved to
        // SGPRs so that the three records in flight do not occupy 12 VGPRs
        const uint64_t xp = (uint64_t)(uintptr_t)r.row;
        r.row = (const uint16_t *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xp >> 32)) << 32) |
                                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xp));
        r.s0 = __builtin_amdgcn_readfirstlane(r.s0);
        r.mid = __builtin_amdgcn_readfirstlane(r.mid);
        r.nf = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.nf);
        return r;
    };
    // pending samples x[i-2], x[i-1] (qa) and x[i], x[i+1] (qb) for i = 2*base + 32 t: one 8-byte load per pair, see fetch
    uint32_t qa[10], qb[10];
    uint32_t q_item = 0xFFFFFFFFu, q_fb = 0;
    auto fetch = [&](const Item &it, uint32_t it_id, uint32_t fb) {
        const uint32_t fi = fb + (uint32_t)g;
        // Buffer loads: the (wave-uniform) capture buffer is a raw buffer resource in SGPRs, the lane's sample index one
        // VGPR byte offset, the pair index t the instruction's immediate offset; the compiler emits buffer_load_dwordx2 for
        // an 8-byte access of unknown alignment (for a global pointer it would split it into dwords).  A lane windows the
        // samples i and i + 1, i = 2*base + 32 t, and needs x[i-1] for the pre-emphasis (MFCC.C:119): the 8 bytes fetched
        // are x[i-2 .. i+1], which start on a 4-byte boundary whenever the segment starts on an even sample (segments from
        // the VAD start on frame boundaries: always) -- the 2-byte-aligned form x[i-1 .. i+2] kept the texture addresser
        // busy 55 % of the kernel.  x[i-2] of the very first pair may lie before the buffer (segment at sample 1): the
        // offset is then negative = out of range for the resource, the load returns 0, and the value is never used.
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void *)it.row, 0, (int)(2 * (uint32_t)a.pcm_stride), 0x00027000);
        const int off = 2 * (it.s0 - 2 + kHopE * (int)(fi < it.nf ? fi : it.nf - 1) + 2 * base);  // groups past the last frame redo it
#pragma unroll
        for (int t = 0; t < 10; t++) {
            const u32x2 q2 = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 64 * t, 0, 0);
            qa[t] = q2.x;  // x[i-2] | x[i-1] << 16
            qb[t] = q2.y;  // x[i]   | x[i+1] << 16
        }
        q_item = it_id;
        q_fb = fb;
    };
    Item cur = item_info(blockIdx.x), nx1 = item_info(blockIdx.x + gridDim.x), nx2 = item_info(blockIdx.x + 2 * gridDim.x);
    if (cur.nf) fetch(cur, blockIdx.x, 0);
    for (uint32_t item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const uint32_t b = item / a.tiles, tile = item - b * a.tiles;
        int16_t *out = a.mfcc + (uint64_t)b * a.max_frames * kCoef;
        const uint32_t f0 = tile * kTile + w * kFpw;
        const uint32_t nf = cur.nf;
        const int mid = cur.mid;

        for (uint32_t fb = 0; fb < nf; fb += kGrp) {
            const uint32_t fi = fb + (uint32_t)g;
            const bool live = fi < nf;
            if (!(q_item == item && q_fb == fb)) fetch(cur, item, fb);  // not read ahead (first batch after a run of empty items)
            uint32_t pa[10], pb[10];
#pragma unroll
            for (int t = 0; t < 10; t++) {
                pa[t] = qa[t];
                pb[t] = qb[t];
            }
            if (fb + kGrp < nf) fetch(cur, item, fb + kGrp);
            else if (nx1.nf) fetch(nx1, item + gridDim.x, 0);
            else if (nx2.nf) fetch(nx2, item + 2 * gridDim.x, 0);
            // ---- pre-emphasis + Hamming (MFCC.C:115-124) of the lane's 20 samples: pairs (2*base + 32 t, +1), t < 10;
            //      the even one belongs to sub-transform 0 (slot base + 16 t), the odd one to sub-transform 1
            uint32_t ws[2][10];
#pragma unroll
            for (int t = 0; t < 10; t++) {
                const int p0 = (int)(pa[t] >> 16) - mid, c0 = (int)(pb[t] & 0xFFFFu) - mid;  // x[i-1], x[i]
                const int c1 = (int)(pb[t] >> 16) - mid, p1 = c0;                              // x[i+1], x[i]
                const int t0 = c0 - preemph95(p0), t1 = c1 - preemph95(p1);
                ws[0][t] = (uint32_t)(mul24(t0, (int)(hp[t] & 0xFFFFu)) / 1000) & 0xFFFFu;
                ws[1][t] = (uint32_t)(mul24(t1, (int)(hp[t] >> 16)) / 1000) & 0xFFFFu;
            }
            // ---- passes 1 and 2 of both sub-transforms in registers, then the exchange image
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                uint32_t v[4][4];  // [d0][d1]
#pragma unroll
                for (int d1 = 0; d1 < 4; d1++) {
                    // butterfly idx = d1 + 4*d2 + 16*d3 reads src[r], src[r+64], src[r+128], src[r+192] (A, C, B, D as in
                    // .s:134-145) with r = bitrev6(idx) = base + 16*rev2(d1): slots t = rev2(d1), +4, +8; D (and B when
                    // r + 128 >= 160) is zero padding.  Real samples: the S = 0 combine of BUTFLY4ZERO_OPT (.s:147-168)
                    // is the packed S = 14 combine on the samples scaled by 2^14 ((x << 14) >> 16 = x >> 2).
                    const int rd = ((d1 & 1) << 1) | (d1 >> 1);
                    const uint32_t wa = ws[sub][rd], wc = ws[sub][4 + rd];
                    const int cr = (int)(wc << 16) >> 2;
                    if (rd < 2) {
                        const int br = (int)(ws[sub][8 + rd] << 16) >> 2;
                        r4_packed<false, true, true>(wa, br, 0, cr, 0, cr, 0, v[0][d1], v[1][d1], v[2][d1], v[3][d1]);
                    } else {
                        r4_packed<false, false, true>(wa, 0, 0, cr, 0, cr, 0, v[0][d1], v[1][d1], v[2][d1], v[3][d1]);
                    }
                }
#pragma unroll
                for (int d0 = 0; d0 < 4; d0++) {  // pass 2 (q = 4, block N = 16): coefficient index j & 3 = d0, lane-invariant
                    uint32_t k2[4][2];
                    load_tw4(a.t, 0, d0, k2);
                    bfly_pk<false>(v[d0][0], v[d0][1], v[d0][2], v[d0][3], k2[0][0], k2[0][1], k2[1][0], k2[1][1], k2[2][0],
                                   k2[2][1], k2[3][0], k2[3][1]);
                }
#pragma unroll
                for (int d1 = 0; d1 < 4; d1++)
#pragma unroll
                    for (int d0 = 0; d0 < 4; d0++) xb[sub * kXSub + (d0 + 4 * d1) * kXStride + lane] = v[d0][d1];
            }
            wave_sync();
            uint32_t u[2][4][4];  // [sub][d2][d3], lane = (d0, d1) = gl
#pragma unroll
            for (int sub = 0; sub < 2; sub++)
#pragma unroll
                for (int d3 = 0; d3 < 4; d3++)
#pragma unroll
                    for (int d2 = 0; d2 < 4; d2++) u[sub][d2][d3] = xb[sub * kXSub + gl * kXStride + d2 + 4 * d3 + 16 * g];
            // ---- passes 3 (q = 16) and 4 (q = 64)
#pragma unroll
            for (int sub = 0; sub < 2; sub++)
#pragma unroll
                for (int d3 = 0; d3 < 4; d3++)
                    bfly_pk<false>(u[sub][0][d3], u[sub][1][d3], u[sub][2][d3], u[sub][3][d3], k3[0][0], k3[0][1], k3[1][0],
                                   k3[1][1], k3[2][0], k3[2][1], k3[3][0], k3[3][1]);
#pragma unroll
            for (int d2 = 0; d2 < 4; d2++) {
                const u32x4 ka = s_tw4[(2 * d2) * 16 + gl], kb = s_tw4[(2 * d2 + 1) * 16 + gl];
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
                    bfly_pk<false>(u[sub][d2][0], u[sub][d2][1], u[sub][d2][2], u[sub][d2][3], ka.x, ka.y, ka.z, ka.w, kb.x, kb.y,
                                   kb.z, kb.w);
            }
            wave_sync();  // the exchange image has been consumed by every lane: reuse it for the energies
            // ---- radix-2 pass for bins k = gl + 16 m < 256, |X|*10, energy
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const u32x4 wq = s_w512[c * 16 + gl];
                float nrm[2];  // re^2 + im^2 of the two bins of this chunk; their roots are taken together (sqrt_rn_int2)
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    const int m = 2 * c + h2, d2 = m & 3, d3 = m >> 2;
                    const uint32_t e = u[0][d2][d3], o = u[1][d2][d3];
                    int pr, pi;
                    cxmul(o, h2 ? wq.z : wq.x, h2 ? wq.w : wq.y, pr, pi);
                    // (E + (P >> 14)) >> 1 == ((E << 14) + P) >> 15 (the dropped low bits of P are < 1/2); doubled once more
                    // so that the wanted 16 bits are the high halves, packed by one v_perm and squared by one dot product
                    const int t_re = (int)(((uint32_t)((int)(e << 16) >> 1)) + ((uint32_t)pr << 1));
                    const int t_im = (int)(((uint32_t)((int)(e & 0xFFFF0000u) >> 1)) + ((uint32_t)pi << 1));
                    const uint32_t xk = pk_hi16(t_re, t_im);  // (re, im) of X[k] as stored 16-bit values
                    nrm[h2] = (float)sdot2z(xk, xk);
                }
                const f32x2 mg = sqrt_rn_int2(f32x2{nrm[0], nrm[1]}) * f32x2{10.0f, 10.0f};
                const uint32_t mag0 = cvt_u32(mg.x), mag1 = cvt_u32(mg.y);
                xb[g * kEStride + gl + 20 * (2 * c)] = mag0 * mag0;  // bin k = gl + 16 m at k + 4*(k >> 4)
                xb[g * kEStride + gl + 20 * (2 * c + 1)] = mag1 * mag1;
            }
            wave_sync();
            // ---- Mel filterbank via prefix sums (MFCC.C:136-162 at 40 filters / 256 bins): this lane owns the 16
            //      contiguous bins 16*gl .. of its frame
            uint32_t pe[16], po[16], xe, xo;
            {
                uint32_t se = 0, so = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const u32x4 q = *(const u32x4 *)(xb + g * kEStride + 20 * gl + 4 * c);
                    const uint32_t e4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {
                        const u32x4 tq = s_tri[(2 * c + h2) * 16 + gl];  // (even, odd) weights of two bins
                        se += e4[2 * h2] * tq.x / 100u;
                        so += e4[2 * h2] * tq.y / 100u;
                        pe[4 * c + 2 * h2] = se;
                        po[4 * c + 2 * h2] = so;
                        se += e4[2 * h2 + 1] * tq.z / 100u;
                        so += e4[2 * h2 + 1] * tq.w / 100u;
                        pe[4 * c + 2 * h2 + 1] = se;
                        po[4 * c + 2 * h2 + 1] = so;
                    }
                }
                xe = row_scan_incl(se) - se;  // bins of the frame's lower lanes
                xo = row_scan_incl(so) - so;
            }
            wave_sync();
#pragma unroll
            for (int c = 0; c < 4; c++) {
                *(u32x4 *)(xb + g * 512 + 16 * gl + 4 * c) = u32x4{pe[4 * c], pe[4 * c + 1], pe[4 * c + 2], pe[4 * c + 3]};
                *(u32x4 *)(xb + g * 512 + 256 + 16 * gl + 4 * c) = u32x4{po[4 * c], po[4 * c + 1], po[4 * c + 2], po[4 * c + 3]};
            }
            moff[g * 32 + gl] = xe;
            moff[g * 32 + 16 + gl] = xo;
            wave_sync();
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int h = gl + 16 * q;
                if (h < kMelE) {
                    const uint32_t *P = xb + g * 512 + ((h & 1) ? 256 : 0), *X = moff + g * 32 + ((h & 1) ? 16 : 0);
                    const int f_lo = (int)(f_lohi[q] >> 16), ih = (int)(f_lohi[q] & 0xFFFFu) - 1, il = f_lo - 1;
                    const uint32_t hi = P[ih] + X[ih >> 4], lo = f_lo ? P[il] + X[il >> 4] : 0u;
                    if (live) powb[fi * kMelEPad + h] = hi - lo;
                }
            }
            wave_sync();
        }
        for (uint32_t t = lane; t < nf * kMelEPad; t += 64) powb[t] = log100_u32(powb[t], a.t.log_thr) << 14;
        wave_sync();
        {   // see k_mfcc: out[(f0 + fi)*12 + h] = out_w[t], 32-bit index arithmetic
            int16_t *out_w = out + (size_t)f0 * kCoef;
#pragma unroll
            for (uint32_t t = lane; t < (uint32_t)(kFpw * kCoef); t += 64) {
                if (t < nf * kCoef) {
                    const uint32_t fi = umul24(t, 10923u) >> 17, h = t - umul24(fi, (uint32_t)kCoef);
                    const uint32_t *pw = powb + umul24(fi, (uint32_t)kMelEPad), *dm = s_dctM + umul24(h, (uint32_t)kMelEPad);
                    const int *ds = s_dctS + umul24(h, (uint32_t)kMelEPad);
                    int acc = 0;
#pragma unroll
                    for (int i = 0; i < kMelE; i++) acc = mad24((int)__umulhi(pw[i], dm[i]), ds[i], acc);
                    out_w[t] = (int16_t)acc;
                }
            }
        }
        wave_sync();
        {
            const uint32_t r0 = f0 + nf, r1 = (f0 + kFpw < a.max_frames) ? f0 + kFpw : a.max_frames;
            for (uint32_t t = r0 * kCoef + lane; t < r1 * kCoef && r0 < r1; t += 64) out[t] = 0;
        }
        cur = nx1;
        nx1 = nx2;
        nx2 = item_info(item + 3 * gridDim.x);
    }
}

uint32_t mfcc_frames_per_tile(uint32_t frame_len) { return frame_len == 320 ? (uint32_t)ext::kTile : (uint32_t)kFramesPerTile; }

// workgroups of the frame kernel that fit on the current device at once (occupancy query x CU count)
uint32_t mfcc_resident_workgroups(uint32_t frame_len)
{
    int dev = 0, n_cu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) return 0;
    hipError_t e;
    if (frame_len == 320)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mfcc_ext, 64 * ext::kWaves,
                                                         (size_t)ext::kWaves * ext::kWaveWords * sizeof(uint32_t));
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mfcc, 64 * kMfccWaves,
                                                         (size_t)kMfccWaves * kWaveLdsWords * sizeof(uint32_t));
    if (e != hipSuccess || per_cu < 1) return 0;
    return (uint32_t)(per_cu * n_cu);
}

void launch_mfcc(const MfccArgs &a, hipStream_t s)
{
    if (a.n_items == 0) return;
    // persistent-style grid: a few times the workgroups that are resident at once (see sr_create), work items strided
    const uint32_t cap = a.grid_cap ? a.grid_cap : 4096u;
    const uint32_t grid = a.n_items < cap ? a.n_items : cap;
    if (a.frame_len == 320) {
        const size_t lds = (size_t)ext::kWaves * ext::kWaveWords * sizeof(uint32_t);
        hipLaunchKernelGGL(k_mfcc_ext, dim3(grid), dim3(64 * ext::kWaves), lds, s, a);
        return;
    }
    const size_t lds = (size_t)kMfccWaves * kWaveLdsWords * sizeof(uint32_t);
    hipLaunchKernelGGL(k_mfcc, dim3(grid), dim3(64 * kMfccWaves), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// generic complex FFT (cr4_fft_1024_stm32 symbol) and fft() magnitudes: one wave per array
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bitrev8(int v) { return (int)(__brev((uint32_t)v) >> 24); }

// Full-complex passes 1-5, result in natural order in `out` (global).  in/out may not alias in LDS terms.
__device__ void fft_full_wave(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *buf, int lane,
                              const DevTables &t)
{
    // pass 1 (.s:226-232): 256 butterflies, 4 per lane, bit-reversed gather, legs 256 words apart
    // loaded in the order A, C, B, D (.s:134-145); outputs to buf[4*idx + k]
    for (int m = 0; m < 4; m++) {
        const int idx = lane + 64 * m, r = bitrev8(idx);
        const uint32_t wa = in[r], wc = in[r + 256], wb = in[r + 512], wd = in[r + 768];
        int ar = sext_lo(wa), ai = sext_hi(wa), br = sext_lo(wb), bi = sext_hi(wb);
        int cr = sext_lo(wc), ci = sext_hi(wc), dr = sext_lo(wd), di = sext_hi(wd);
        r4_combine<0>(ar, ai, br, bi, cr, ci, dr, di);
        buf[xaddr(4 * idx + 0)] = pack16(ar, ai);
        buf[xaddr(4 * idx + 1)] = pack16(br, bi);
        buf[xaddr(4 * idx + 2)] = pack16(cr, ci);
        buf[xaddr(4 * idx + 3)] = pack16(di, dr);
    }
    wave_sync();
    LaneTw tw;
    load_lane_tw(t, lane, tw);
    uint32_t k2[3][2];
    load_tw3(t, 0, lane & 3, k2);
    const int d0 = lane & 3, d3 = (lane >> 2) & 3, d4 = lane >> 4;
    uint32_t v[4][4], u[4][4];
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
#pragma unroll
        for (int d2 = 0; d2 < 4; d2++) v[d1][d2] = buf[xaddr(d0 + 4 * d1 + 16 * d2 + 64 * d3 + 256 * d4)];
    wave_sync();
#pragma unroll
    for (int d2 = 0; d2 < 4; d2++)
        bfly(v[0][d2], v[1][d2], v[2][d2], v[3][d2], k2[0][0], k2[0][1], k2[1][0], k2[1][1], k2[2][0], k2[2][1]);
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
        bfly(v[d1][0], v[d1][1], v[d1][2], v[d1][3], tw.s3[d1][0][0], tw.s3[d1][0][1], tw.s3[d1][1][0],
             tw.s3[d1][1][1], tw.s3[d1][2][0], tw.s3[d1][2][1]);
    fft_exchange(buf, lane, v, u);
#pragma unroll
    for (int e4 = 0; e4 < 4; e4++)
        bfly(u[0][e4], u[1][e4], u[2][e4], u[3][e4], tw.s4[0][0], tw.s4[0][1], tw.s4[1][0], tw.s4[1][1], tw.s4[2][0],
             tw.s4[2][1]);
#pragma unroll
    for (int e3 = 0; e3 < 4; e3++) {
        bfly(u[e3][0], u[e3][1], u[e3][2], u[e3][3], tw.s5[e3][0][0], tw.s5[e3][0][1], tw.s5[e3][1][0],
             tw.s5[e3][1][1], tw.s5[e3][2][0], tw.s5[e3][2][1]);
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) out[lane + 64 * e3 + 256 * e4] = u[e3][e4];
    }
}

__global__ void __launch_bounds__(64) k_fft_q15(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables t)
{
    __shared__ uint32_t buf[kXchgWords];
    const int lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        fft_full_wave(in + (size_t)i * kNfft, out + (size_t)i * kNfft, buf, lane, t);
        wave_sync();
    }
}

void launch_fft_q15(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables &t, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_fft_q15, dim3(n < 4096 ? n : 4096), dim3(64), 0, s, in, out, n, t);
}

// fft() of MFCC.C:27-62 on its own: zero-extend len samples, FFT, magnitudes of bins 0..511.
__global__ void __launch_bounds__(64) k_fft_mag(const int16_t *frames, uint32_t len, uint32_t *mag, uint32_t *raw_hi,
                                                uint32_t n, const DevTables t)
{
    __shared__ uint32_t buf[kXchgWords];
    __shared__ uint32_t fin[kNfft], fout[kNfft];
    const int lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        for (int k = lane; k < kNfft; k += 64) fin[k] = (k < (int)len) ? (uint32_t)(uint16_t)frames[(size_t)i * len + k] : 0u;
        wave_sync();
        fft_full_wave(fin, fout, buf, lane, t);
        wave_sync();
        for (int k = lane; k < kBins; k += 64) {
            const int re = sext_lo(fout[k]), im = sext_hi(fout[k]);
            const int r = re * re + im * im;
            mag[(size_t)i * kBins + k] = (uint32_t)(sqrtf((float)r) * 10.0f);
            if (raw_hi) raw_hi[(size_t)i * kBins + k] = fout[kBins + k];
        }
        wave_sync();
    }
}

void launch_fft_mag(const int16_t *frames, uint32_t len, uint32_t *mag, uint32_t *raw_hi, uint32_t n, const DevTables &t,
                    hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_fft_mag, dim3(n < 4096 ? n : 4096), dim3(64), 0, s, frames, len, mag, raw_hi, n, t);
}

// ------------------------------------------------------------------------------------------------
// k_vad: one wave per capture buffer
// ------------------------------------------------------------------------------------------------
constexpr int kVadWaves = 4;

__device__ __forceinline__ uint32_t absdiff(uint32_t v, uint32_t mid) { return v > mid ? v - mid : mid - v; }

// kSad: |x - mid| sums of two samples per instruction (v_sad_u16).  Only valid for 16-bit mid values, which is what
// noise_atap produces (a mean of u16 samples, VAD.C:41-47); the variant without it serves callers that hand their own
// thresholds in (atap_in), which may hold anything.
template <int kFrameLen, int kHop, bool kSad>  // 160/80 = the reference (VAD.H:5-8); 320/160 = the 16 kHz extension
__global__ void __launch_bounds__(64 * kVadWaves) k_vad(const VadArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * kVadWaves + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const uint4 *row = (const uint4 *)(a.pcm + (uint64_t)b * a.pcm_stride);
    const uint32_t S = a.buf_len;

    // ---- noise_atap (VAD.C:22-71) over the first noise_len samples --------------------------
    uint32_t mid, n_thl, z_thl, s_thl;
    if (a.atap_in) {
        mid = a.atap_in[b].mid_val;
        n_thl = a.atap_in[b].n_thl;
        z_thl = a.atap_in[b].z_thl;
        s_thl = a.atap_in[b].s_thl;
    } else {
        const uint32_t nvec = a.noise_len / 8;
        uint32_t part = 0;
        for (uint32_t v = lane; v < nvec; v += 64) {
            const uint4 q = row[v];
            part += (q.x & 0xFFFF) + (q.x >> 16) + (q.y & 0xFFFF) + (q.y >> 16) + (q.z & 0xFFFF) + (q.z >> 16) +
                    (q.w & 0xFFFF) + (q.w >> 16);
        }
        mid = wave_sum(part) / a.noise_len;  // VAD.C:41-45
        const uint32_t nblk = a.noise_len / a.atap_frm, vpb = a.atap_frm / 8;
        uint32_t max_sum = 0, abs_part = 0;
        for (uint32_t blk = 0; blk < nblk; blk++) {  // VAD.C:48-63
            uint32_t nmax = 0;
            for (uint32_t v = lane; v < vpb; v += 64) {
                const uint4 q = row[blk * vpb + v];
                const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const uint32_t ad = absdiff((wds[s >> 1] >> (16 * (s & 1))) & 0xFFFF, mid);
                    nmax = ad > nmax ? ad : nmax;
                    abs_part += ad;
                }
            }
            max_sum += wave_max(nmax);
        }
        uint32_t abs_sum = wave_sum(abs_part);
        abs_sum /= (a.noise_len / (uint32_t)kFrameLen);  // VAD.C:65 (divides by n_len/frame_len)
        max_sum /= nblk;                                 // VAD.C:66
        n_thl = max_sum & 0xFFFF;                        // u16 field, n_thl_ratio = 1
        s_thl = abs_sum * 11 / 10;                       // s_thl_ratio
        z_thl = (uint32_t)kFrameLen * 2 / 160 / 1;       // VAD.C:70
    }
    const uint32_t a_thl = mid + n_thl, b_thl = mid - n_thl;  // VAD.C:112-113 (u32, may wrap)
    const uint32_t mid2 = (mid & 0xFFFFu) * 0x10001u;          // mid in both halves (kSad)

    // ---- per-frame short-time magnitude and band-crossing count (VAD.C:121-157) ----------------
    // Frames start every 80 samples, so both quantities are assembled from per-80-sample block
    // summaries.  Class of a sample: 2 above the band, 1 below, 0 inside.  last_sig is never
    // reset (VAD.C:99): on entry to frame f it is the class of the last out-of-band sample at
    // position <= 80f+78, because the previous frame already scanned up to there.
    const uint32_t F = (S > (uint32_t)kFrameLen) ? (S - kFrameLen + kHop - 1) / kHop : 0;  // frames, VAD.C:121
    uint32_t cur = 0, front = 0, back = 0, vcon = 0;  // VAD.C:100-102,109
    // segment bounds go straight to the record as they are found (rare events, lane 0 only);
    // segment 0 is also kept in registers for the frame count below
    int seg0_start = -1, seg0_end = -1;
    sr_vad_rec *rec_out = a.vad + b;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 2 * SR_MAX_SEG; i++) rec_out->seg[i] = -1;
    }
    uint32_t carry = 0;  // class of the last out-of-band sample before the current round's first block
    bool done = false;
    const uint32_t v_durmin = 8, s_durmax = 11;  // VAD.C:72-75 at 20 ms / 10 ms framing

    for (uint32_t jb = 0; jb < F && !done; jb += 63) {
        const uint32_t j = jb + lane;  // block index; frame f = j uses blocks j and j+1
        uint32_t A = 0, internal = 0, last = 0, cf = 0, c78 = 0;
        int pfo = -1;
        if (j <= F) {
#pragma unroll
            for (int t = 0; t < kHop / 8; t++) {
                const uint4 q = row[(uint64_t)j * (kHop / 8) + t];
                const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
                if (kSad) {  // v_sad_u16: |a.lo-b.lo| + |a.hi-b.hi| + c
#pragma unroll
                    for (int wdi = 0; wdi < 4; wdi++) A = __builtin_amdgcn_sad_u16(wds[wdi], mid2, A);
                } else {
#pragma unroll
                    for (int s = 0; s < 8; s++) A += absdiff((wds[s >> 1] >> (16 * (s & 1))) & 0xFFFF, mid);
                }
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int off = t * 8 + s;
                    const uint32_t x = (wds[s >> 1] >> (16 * (s & 1))) & 0xFFFF;
                    const uint32_t c = (x >= a_thl) ? 2u : (x < b_thl ? 1u : 0u);
                    if (off == kHop - 1) c78 = last;
                    const bool nz = c != 0;
                    internal += (nz && last != 0 && last != c) ? 1u : 0u;
                    const bool first = nz && last == 0;
                    cf = first ? c : cf;
                    pfo = first ? off : pfo;
                    last = nz ? c : last;
                }
            }
        }
        const uint32_t c80 = last;
        // R(j) = class of the last out-of-band sample in blocks <= j
        uint32_t R = c80;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(R, d, 64);
            if (lane >= d) R = R ? R : o;
        }
        R = R ? R : carry;
        uint32_t Rprev = __shfl_up(R, 1, 64);
        if (lane == 0) Rprev = carry;
        carry = __shfl(R, 62, 64);
        const uint32_t ff = (cf != 0 && Rprev != 0 && Rprev != cf) ? 1u : 0u;  // flip at the block's first out-of-band sample
        const uint32_t fl = internal + ff;
        const uint32_t fl_next = __shfl_down(fl, 1, 64), A_next = __shfl_down(A, 1, 64);
        uint32_t Z = internal + fl_next;
        if (pfo < 0 || pfo == kHop - 1)
            Z += ff;  // entry state = history before the frame: natural count
        else if (pfo > 0 && j > 0)
            Z += (c78 != cf) ? 1u : 0u;  // entry state comes from inside the frame (positions <= 78);
                                         // frame 0 starts with last_sig = 0 (VAD.C:99)
        const uint32_t frm_sum = A + A_next;
        const bool loud = (lane < 63) && (j < F) && (frm_sum > s_thl || Z > z_thl);  // VAD.C:164
        const uint64_t mask = __ballot(loud);
        if (a.dbg_masks && lane == 0) a.dbg_masks[(uint64_t)b * 16 + (jb / 63 < 16 ? jb / 63 : 15)] = mask;
        const uint32_t nfr = (F - jb < 63u) ? F - jb : 63u;

        // ---- endpoint state machine (VAD.C:164-216), wave-uniform, advanced one RUN of equal frames at a time
        // (count-trailing-zeros on the ballot) instead of frame by frame: the scalar unit is the busiest resource
        // of this kernel.  State and counters carry across rounds exactly as cur/front/back do in the reference.
        //   silence(0): quiet frames do nothing; the first loud frame starts an onset with front = 1
        //   onset(1):   each loud frame front++, the frame that makes front == v_durmin opens the segment
        //               (start = i - (v_durmin-1)*hop, VAD.C:175-180); a quiet frame falls back to silence
        //   speech(2):  loud frames do nothing; the first quiet frame starts a tail with back = 1
        //   tail(3):    each quiet frame back++, the frame that makes back == s_durmax closes the segment
        //               (end = i - s_durmax*hop + frame_len, VAD.C:198-207); a loud frame returns to speech
        uint32_t t = 0;
        while (t < nfr) {
            const uint64_t rem = mask >> t, stop = 1ull << (nfr - t);  // sentinel: runs end at the round's last frame
            const uint32_t ones = (uint32_t)__builtin_ctzll(~rem | stop), zeros = (uint32_t)__builtin_ctzll(rem | stop);
            if (cur == 0) {
                t += zeros;
                if (t < nfr) {
                    cur = 1;
                    front = 1;
                    t++;
                }
            } else if (cur == 1) {
                const uint32_t need = v_durmin - front;
                if (ones >= need) {
                    t += need;
                    const int i = (int)((jb + t - 1) * kHop);  // the frame that completed the run
                    const int st = i - (int)((v_durmin - 1) * kHop);
                    if (vcon == 0) seg0_start = st;
                    if (lane == 0) rec_out->seg[2 * vcon] = st;
                    cur = 2;
                    front = 0;
                } else if (t + ones < nfr) {  // a quiet frame ends the onset
                    t += ones + 1;
                    front = 0;
                    cur = 0;
                } else {
                    front += ones;
                    t = nfr;
                }
            } else if (cur == 2) {
                t += ones;
                if (t < nfr) {
                    cur = 3;
                    back = 1;
                    t++;
                }
            } else {
                const uint32_t need = s_durmax - back;
                if (zeros >= need) {
                    t += need;
                    const int i = (int)((jb + t - 1) * kHop);
                    const int en = i - (int)(s_durmax * kHop) + kFrameLen;
                    if (vcon == 0) seg0_end = en;
                    if (lane == 0) rec_out->seg[2 * vcon + 1] = en;
                    vcon++;
                    cur = 0;
                    back = 0;
                    if (vcon == a.max_seg) {  // VAD.C:203-206
                        done = true;
                        break;
                    }
                } else if (t + zeros < nfr) {  // a loud frame returns to speech
                    t += zeros + 1;
                    back = 0;
                    cur = 2;
                } else {
                    back += zeros;
                    t = nfr;
                }
            }
        }
    }

    if (lane == 0) {
        sr_atap at;
        at.mid_val = mid;
        at.n_thl = (uint16_t)n_thl;
        at.z_thl = (uint16_t)z_thl;
        at.s_thl = s_thl;
        rec_out->atap = at;
        uint32_t frm = 0, status;
        if (seg0_end < 0) {
            status = SR_ST_VAD_FAIL;
        } else if (seg0_start < 1) {
            status = SR_ST_SEG_OOB;
        } else {
            // MFCC.C:102: u32 arithmetic, result truncated to u16
            const uint32_t n = ((((uint32_t)(seg0_end - seg0_start) - kFrameLen) / kHop) + 1) & 0xFFFF;
            if (n > a.max_frames) {
                status = SR_ST_MFCC_FAIL;
            } else {
                status = SR_ST_OK;
                frm = n;
            }
        }
        rec_out->frm_num = frm;
        rec_out->status = status;
        rec_out->_pad = 0;
    }
}

// Re-targets the per-utterance records at VAD segment `seg_idx` (0..2): the frame and DTW kernels always work
// on "segment 0" of the record they are given.  Frame count and status follow MFCC.C:102-107 / main.c:261-274.
__global__ void k_select_segment(const sr_vad_rec *in, sr_vad_rec *out, uint32_t B, uint32_t seg_idx, uint32_t max_frames,
                                 uint32_t frame_len, uint32_t hop)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    sr_vad_rec r = in[b];
    const int st = r.seg[2 * seg_idx], en = r.seg[2 * seg_idx + 1];
    r.seg[0] = st;
    r.seg[1] = en;
    r.frm_num = 0;
    if (en < 0) {
        r.status = SR_ST_VAD_FAIL;
    } else if (st < 1) {
        r.status = SR_ST_SEG_OOB;
    } else {
        const uint32_t n = ((((uint32_t)(en - st) - frame_len) / hop) + 1) & 0xFFFF;
        r.status = n > max_frames ? SR_ST_MFCC_FAIL : SR_ST_OK;
        r.frm_num = n > max_frames ? 0 : n;
    }
    out[b] = r;
}
void launch_select_segment(const sr_vad_rec *in, sr_vad_rec *out, uint32_t B, uint32_t seg_idx, uint32_t max_frames,
                           uint32_t frame_len, uint32_t hop, hipStream_t s)
{
    if (!B) return;
    hipLaunchKernelGGL(k_select_segment, dim3((B + 255) / 256), dim3(256), 0, s, in, out, B, seg_idx, max_frames,
                       frame_len, hop);
}

void launch_vad(const VadArgs &a, hipStream_t s)
{
    if (!a.B) return;
    const dim3 grid((a.B + kVadWaves - 1) / kVadWaves), block(64 * kVadWaves);
    const bool own_thresholds = a.atap_in == nullptr;  // noise_atap runs in the kernel: mid is a 16-bit quantity
    if (a.frame_len == 320) {
        if (own_thresholds) hipLaunchKernelGGL((k_vad<320, 160, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_vad<320, 160, false>), grid, block, 0, s, a);
    } else {
        if (own_thresholds) hipLaunchKernelGGL((k_vad<160, 80, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_vad<160, 80, false>), grid, block, 0, s, a);
    }
}

// ------------------------------------------------------------------------------------------------
// k_dtw: one lane per (utterance, template) pair, greedy local walk of DTW.C:120-192
// ------------------------------------------------------------------------------------------------
struct Frame12 {
    uint32_t w[6];
};
__device__ __forceinline__ Frame12 load_frame(const int16_t *p)
{
    const uint2 *q = (const uint2 *)p;  // rows are 24 bytes, 8-byte aligned
    const uint2 a = q[0], b = q[1], c = q[2];
    Frame12 f;
    f.w[0] = a.x;
    f.w[1] = a.y;
    f.w[2] = b.x;
    f.w[3] = b.y;
    f.w[4] = c.x;
    f.w[5] = c.y;
    return f;
}
__device__ __forceinline__ uint32_t norm2(const Frame12 &f)
{
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) acc = sdot2(f.w[i], f.w[i], acc);
    return (uint32_t)acc;
}
// get_dis (DTW.C:45-62): sum (a-b)^2 in u32 wrap = |a|^2 + |b|^2 - 2 a.b in the same ring
__device__ __forceinline__ uint32_t get_dis_dev(const Frame12 &fa, uint32_t na, const Frame12 &fb, uint32_t nb)
{
    int dot = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) dot = sdot2(fa.w[i], fb.w[i], dot);
    const uint32_t d = na + nb - 2u * (uint32_t)dot;
    return cvt_u32(sqrt_rn_int((float)d));
}
// dtw_limit (DTW.C:76-109); returns true when (x, y) is OUTSIDE the relaxed parallelogram
__device__ __forceinline__ bool dtw_out(int x, int y, int X1, int X2, int in_n, int mdl_n)
{
    const bool o1 = (x < X1) ? (y >= 2 * x + 2) : (2 * y + in_n - 2 * mdl_n >= x + 4);
    const bool o2 = (x < X2) ? (2 * y + 2 <= x) : (y + 4 <= 2 * x + mdl_n - 2 * in_n);
    return o1 || o2;
}

__device__ uint32_t dtw_pair(const int16_t *in, uint32_t in_n, uint32_t in_rows, const int16_t *mdl, uint32_t mdl_n,
                             uint32_t mdl_rows)
{
    if (in_n > mdl_n * 2 || 2 * in_n < mdl_n) return SR_DIS_ERR;  // DTW.C:133-137
    const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142 (u16 statics)
    const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
    uint32_t px = 0, py = 0;  // 0-based rows under the in / mdl pointers (x = px+1, y = py+1)
    Frame12 ci = load_frame(in), cm = load_frame(mdl);
    uint32_t nci = norm2(ci), ncm = norm2(cm);
    uint32_t dis = get_dis_dev(ci, nci, cm, ncm);
    uint32_t step = 1;
    do {
        // rows px+1 / py+1 are read even when they lie past the sequence end (do-while, DTW.C:150-154);
        // clamped to the allocated rows so the access stays inside the buffer
        const uint32_t rx = (px + 1 < in_rows) ? px + 1 : in_rows - 1, ry = (py + 1 < mdl_rows) ? py + 1 : mdl_rows - 1;
        const Frame12 ni = load_frame(in + (size_t)rx * kCoef), nm = load_frame(mdl + (size_t)ry * kCoef);
        const uint32_t nni = norm2(ni), nnm = norm2(nm);
        const int x = (int)px + 1, y = (int)py + 1;
        const uint32_t up = dtw_out(x, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ci, nci);
        const uint32_t right = dtw_out(x + 1, y, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(cm, ncm, ni, nni);
        const uint32_t diag =
            dtw_out(x + 1, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ni, nni);
        uint32_t mn = diag;  // DTW.C:156-164
        if (mn > right) mn = right;
        if (mn > up) mn = up;
        dis += mn;
        const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:168-184
        const bool adv_x = mv_diag || !mv_up, adv_y = mv_diag || mv_up;
        if (adv_x) {
            ci = ni;
            nci = nni;
            px++;
        }
        if (adv_y) {
            cm = nm;
            ncm = nnm;
            py++;
        }
        step = (step + 1) & 0xFFFF;  // u16 step
    } while (px + 1 < in_n && py + 1 < mdl_n);  // DTW.C:188
    return dis / step;
}

// ---- get_mdl (DTW.C:217-296) + get_mean (DTW.C:195-205): template averaging, one lane per pair --------------
// The same greedy walk as dtw_pair with in1 in the "in" role and in2 in the "mdl" role; the start point and every
// point the walk moves to contribute one merged frame = per-coefficient (a + b) / 2 in int arithmetic (truncation
// toward zero).  The merged template has `step` frames; frames >= out_rows are dropped (the reference would write
// past its 119-frame record there).
__device__ __forceinline__ uint32_t mean_word(uint32_t a, uint32_t b)
{
    const int lo = (sext_lo(a) + sext_lo(b)) / 2, hi = (sext_hi(a) + sext_hi(b)) / 2;
    return pack16(lo, hi);
}
__device__ __forceinline__ void store_mean(int16_t *row, const Frame12 &a, const Frame12 &b)
{
    uint2 *q = (uint2 *)row;
    q[0] = make_uint2(mean_word(a.w[0], b.w[0]), mean_word(a.w[1], b.w[1]));
    q[1] = make_uint2(mean_word(a.w[2], b.w[2]), mean_word(a.w[3], b.w[3]));
    q[2] = make_uint2(mean_word(a.w[4], b.w[4]), mean_word(a.w[5], b.w[5]));
}

__global__ void __launch_bounds__(64) k_get_mdl(const GetMdlArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.P) return;
    const uint32_t in_n = a.n1[p], mdl_n = a.n2[p];
    const int16_t *in = a.in1 + (size_t)p * a.rows1 * kCoef, *mdl = a.in2 + (size_t)p * a.rows2 * kCoef;
    int16_t *out = a.mdl + (size_t)p * a.mdl_rows * kCoef;
    if (in_n == 0 || mdl_n == 0 || in_n > mdl_n * 2 || 2 * in_n < mdl_n) {  // DTW.C:236-239
        a.dis[p] = SR_DIS_ERR;
        a.mdl_frames[p] = 0;
        return;
    }
    const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);
    const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
    uint32_t px = 0, py = 0;
    Frame12 ci = load_frame(in), cm = load_frame(mdl);
    uint32_t nci = norm2(ci), ncm = norm2(cm);
    uint32_t dis = get_dis_dev(ci, nci, cm, ncm);
    if (a.mdl_rows) store_mean(out, ci, cm);  // DTW.C:250-251
    uint32_t step = 1;
    do {
        const uint32_t rx = (px + 1 < a.rows1) ? px + 1 : a.rows1 - 1, ry = (py + 1 < a.rows2) ? py + 1 : a.rows2 - 1;
        const Frame12 ni = load_frame(in + (size_t)rx * kCoef), nm = load_frame(mdl + (size_t)ry * kCoef);
        const uint32_t nni = norm2(ni), nnm = norm2(nm);
        const int x = (int)px + 1, y = (int)py + 1;
        const uint32_t up = dtw_out(x, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ci, nci);
        const uint32_t right = dtw_out(x + 1, y, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(cm, ncm, ni, nni);
        const uint32_t diag =
            dtw_out(x + 1, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ni, nni);
        uint32_t mn = diag;  // DTW.C:260-268
        if (mn > right) mn = right;
        if (mn > up) mn = up;
        dis += mn;
        const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:272-288
        if (mv_diag || !mv_up) {
            ci = ni;
            nci = nni;
            px++;
        }
        if (mv_diag || mv_up) {
            cm = nm;
            ncm = nnm;
            py++;
        }
        if (step < a.mdl_rows) store_mean(out + (size_t)step * kCoef, ci, cm);  // DTW.C:286-287 (row = step before ++)
        step = (step + 1) & 0xFFFF;
    } while (px + 1 < in_n && py + 1 < mdl_n);  // DTW.C:291
    a.mdl_frames[p] = step;  // DTW.C:293
    a.dis[p] = dis / step;
}
void launch_get_mdl(const GetMdlArgs &a, hipStream_t s)
{
    if (!a.P) return;
    hipLaunchKernelGGL(k_get_mdl, dim3((a.P + 63) / 64), dim3(64), 0, s, a);
}

__global__ void __launch_bounds__(128) k_dtw(const DtwArgs a)
{
    const uint64_t pid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= (uint64_t)a.B * a.K) return;
    const uint32_t b = (uint32_t)(pid / a.K), k = (uint32_t)(pid - (uint64_t)b * a.K);
    uint32_t in_n, ok;
    if (a.in_frames) {
        in_n = a.in_frames[b];
        ok = in_n != 0;
    } else {
        in_n = a.vad[b].frm_num;
        ok = a.vad[b].status == SR_ST_OK && in_n != 0;
    }
    uint32_t d = SR_DIS_ERR;
    if (ok && a.tpl_valid[k])  // main.c:283
        d = dtw_pair(a.mfcc + (size_t)b * a.max_frames * kCoef, in_n, a.max_frames, a.tpl + (size_t)k * a.tpl_stride,
                     a.tpl_frames[k], a.tpl_rows);
    a.scores[pid] = d;
}

// ---- k_dtw_lds: the production DTW kernel ---------------------------------------------------------
// A workgroup owns U utterances whose MFCC rows (+ squared norms) are staged in LDS once and reused by
// all K templates; lanes are the U*K pairs ordered (template-sorted-by-length major, utterance minor),
// so the lanes of a wave walk sequences of nearly equal length (little trip-count divergence) and
// touch at most ceil(64/U) templates.  Templates live in HBM/L2 in a row-interleaved layout
// tplR[row][ks] (ks = rank of the template by length) of 32-byte rows: 12 x s16 holding -2*coefficient, the
// u32 squared norm of the row, pad -- so lanes that advance in step read neighbouring addresses and a squared
// distance |m|^2 + |in|^2 - 2 m.in is the norm sum fed through six accumulating dot products.
// One root per step: the root of the smallest admissible squared candidate; the reference's tie order is decided on
// the squared values against (root+1)^2 -/+ a proven margin, with the literal three-root form as wave-uniform
// fallback (see the loop).  The result is bit-identical to dtw_pair above.

__device__ __forceinline__ uint32_t dis_from(uint32_t na, uint32_t nb, int dot)
{
    const uint32_t d = na + nb - 2u * (uint32_t)dot;
    return cvt_u32(sqrt_rn_int((float)d));
}

// floor(sqrtf(f)) by bracketing: v_sqrt_f32 is within 1 ulp, so the correctly rounded root is s0 or one of its two
// neighbours; when floor() of both neighbours agree the answer is known without the correction step.  `unsafe`
// collects the (rare: ~1e-4 per value) cases that need sqrt_rn_int.  Validated for all 2^32 inputs by
// tests/exhaustive_math_sweep.py through sr_math_diag.
__device__ __forceinline__ uint32_t sqrt_floor_bracket(uint32_t d, bool &unsafe)
{
    const float s0 = __builtin_amdgcn_sqrtf((float)d);
    const uint32_t hi = (uint32_t)__int_as_float(__float_as_int(s0) + 1);  // floor(succ(s0))
    unsafe |= !(s0 > (float)hi);  // s0 > hi  <=>  pred(s0) >= hi  <=>  floor(pred(s0)) == hi as well
    return hi;
}
// a - b, saturating at 0
__device__ __forceinline__ uint32_t sub_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}


struct DtwLdsArgs {
    DtwArgs d;
    const u32x4 *tplR;          // [tpl_rows][K] 32-byte rows: 12 x s16 holding -2*coef | u32 squared norm | pad ; length order
    const uint32_t *tpl_frames_s;  // [K] frames, sorted order; 0 for invalid slots
    const uint32_t *tpl_orig;   // [K] original slot of sorted position
    uint32_t U;                 // utterances per workgroup
    const int8_t *tie_delta;    // [tie_g] tie-threshold table (sr_tables.h), staged at the start of the dynamic LDS
    uint32_t tie_g;             // entries staged: roots >= tie_g take the literal path
    uint32_t Kc;                // templates (ranks) per workgroup: blockIdx.y selects the chunk [y*Kc, (y+1)*Kc) of the K ranks
};

constexpr int kDtwMaxU = 16;
// words between utterances in the LDS image: rows are 6 words; the stride is the next value == 22 (mod 64)
// so that equal rows of different utterances do not alias (64 banks for 8-byte reads); norms: == 11 (mod 32)
__host__ __device__ inline uint32_t dtw_lds_row_stride(uint32_t R)
{
    uint32_t s = R * 6;
    return s + ((22 + 64 - (s & 63)) & 63);
}
__host__ __device__ inline uint32_t dtw_lds_nrm_stride(uint32_t R) { return R + ((11 + 32 - (R & 31)) & 31); }

// 32-byte feature row: w[0..5] = 12 x s16, w[6] = squared norm (u32 wrap), w[7] unused
struct Row32 {
    uint32_t w[8];
};
__device__ __forceinline__ Row32 row_from(const u32x4 lo, const u32x4 hi)
{
    Row32 r;
    r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
    r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
    return r;
}
__device__ __forceinline__ Row32 row_from2(const u32x2 a, const u32x2 b, const u32x2 c, uint32_t nrm)
{
    Row32 r;
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = b.x; r.w[3] = b.y; r.w[4] = c.x; r.w[5] = c.y;
    r.w[6] = nrm; r.w[7] = 0;
    return r;
}
// dst = src as four 64-bit moves (v_pk_mov_b32 moves a register pair per issue slot)
__device__ __forceinline__ void copy_row(Row32 &dst, const Row32 &src)
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        u32x2 d;
        const u32x2 v = {src.w[i], src.w[i + 1]};
        asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(d) : "v"(v));
        dst.w[i] = d.x;
        dst.w[i + 1] = d.y;
    }
}
__device__ __forceinline__ int dot_rows(const Row32 &a, const Row32 &b)
{
    int acc = sdot2z(a.w[0], b.w[0]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc = sdot2(a.w[i], b.w[i], acc);
    return acc;
}

// c + a.b over the 12 coefficients: the accumulator input carries the norm sum
__device__ __forceinline__ int dot_rows_acc(const Row32 &a, const Row32 &b, int c)
{
    int acc = sdot2a(a.w[0], b.w[0], c);
#pragma unroll
    for (int i = 1; i < 6; i++) acc = sdot2(a.w[i], b.w[i], acc);
    return acc;
}

// rows r and r+1 of an utterance's LDS image (24-byte rows, squared norms in a separate array) into registers.
// The six 8-byte reads are volatile so that they stay ds_read_b64 (2 LDS cycles each): merged into ds_read2_b64 they
// cost 8 cycles a pair (MI355X_MICROARCH.md, LDS table).
typedef __attribute__((address_space(3))) const volatile u32x2 lds_cv_u32x2;
typedef __attribute__((address_space(3))) const uint32_t lds_c_u32;
typedef __attribute__((address_space(3))) const int8_t lds_c_i8;
// LDS byte offset of a pointer into the workgroup's shared memory
__device__ __forceinline__ uint32_t lds_offset(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// rows r and r+1 at byte offset row_off (24-byte rows) and their squared norms at nrm_off
__device__ __forceinline__ void lds_rows2(uint32_t row_off, uint32_t nrm_off, Row32 &r0, Row32 &r1)
{
    lds_cv_u32x2 *q = (lds_cv_u32x2 *)(uintptr_t)row_off;
    lds_c_u32 *np = (lds_c_u32 *)(uintptr_t)nrm_off;
    const u32x2 a0 = q[0], a1 = q[1], a2 = q[2], b0 = q[3], b1 = q[4], b2 = q[5];
    r0 = row_from2(a0, a1, a2, np[0]);
    r1 = row_from2(b0, b1, b2, np[1]);
}
#ifdef SR_DTW_STATS
// development build only: wave-steps in total / on the literal path, by reason (bracket, table range, lost lane)
__device__ unsigned long long g_dtw_stats[8];
extern "C" void sr_debug_dtw_stats(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dtw_stats), sizeof(g_dtw_stats));
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dtw_stats), z, sizeof(z));
    }
}
#endif
__global__ void __launch_bounds__(1024) k_dtw_lds(const DtwLdsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32x2 smem2[];  // 8-byte typed: rows are read as ds_read_b64
    const uint32_t U = a.U, R = a.d.max_frames, K = a.d.K;
    // LDS (all of it dynamic): [tie-threshold table, tie_g bytes][rows][norms][frame counts].  The table comes FIRST, at
    // LDS address 0 (the kernel has no static LDS), so that a look-up is one ds_read_i8 whose address register is the
    // root itself -- no VALU address arithmetic.
    // Image of the utterances: 24-byte rows (3 x 8 bytes) + a separate array of squared norms.  Strides are padded so
    // that lanes sitting on the same row of different utterances fall on different banks.
    u32x2 *s_rows = smem2 + a.tie_g / 8;
    const uint32_t row_stride = dtw_lds_row_stride(R), nrm_stride = dtw_lds_nrm_stride(R);  // words
    uint32_t *s_nrm = (uint32_t *)(s_rows + (size_t)U * (row_stride / 2));
    uint32_t *s_n = s_nrm + (size_t)U * nrm_stride + 8;  // frames of the workgroup's utterances, 0 = skip (+8 words: reload slack)
    const uint32_t tid = threadIdx.x, b0 = blockIdx.x * U;
    for (uint32_t i = tid; i < a.tie_g / 16; i += blockDim.x) ((u32x4 *)smem2)[i] = ((const u32x4 *)a.tie_delta)[i];

    if (tid < U) {
        const uint32_t b = b0 + tid;
        uint32_t n = 0;
        if (b < a.d.B) {
            if (a.d.in_frames) n = a.d.in_frames[b];
            else n = (a.d.vad[b].status == SR_ST_OK) ? a.d.vad[b].frm_num : 0u;
        }
        s_n[tid] = n;
    }
    __syncthreads();
    // ---- stage rows [0, min(n+1, R)) of every utterance and their squared norms ----
    for (uint32_t u = 0; u < U; u++) {
        const uint32_t n = s_n[u];
        if (!n) continue;
        const uint32_t rows = (n + 1 < R) ? n + 1 : R;
        const uint2 *src = (const uint2 *)(a.d.mfcc + (size_t)(b0 + u) * R * kCoef);
        u32x2 *dst = s_rows + (size_t)u * (row_stride / 2);
        for (uint32_t r = tid; r < rows; r += blockDim.x) {
            const uint2 q0 = src[3 * r], q1 = src[3 * r + 1], q2 = src[3 * r + 2];
            int nr = sdot2z(q0.x, q0.x);
            nr = sdot2(q0.y, q0.y, nr);
            nr = sdot2(q1.x, q1.x, nr);
            nr = sdot2(q1.y, q1.y, nr);
            nr = sdot2(q2.x, q2.x, nr);
            nr = sdot2(q2.y, q2.y, nr);
            dst[3 * r] = u32x2{q0.x, q0.y};
            dst[3 * r + 1] = u32x2{q1.x, q1.y};
            dst[3 * r + 2] = u32x2{q2.x, q2.y};
            s_nrm[u * nrm_stride + r] = (uint32_t)nr;
        }
    }
    __syncthreads();

    // a store of more than 1024 / U templates is walked in chunks of Kc ranks (grid y): every chunk stages the same U
    // utterances again (6 KB each from L2) and scores them against its slice of the length-sorted store
    if (tid >= U * a.Kc) return;
    // (rank major, utterance minor; the other order -- a wave = consecutive ranks of one utterance -- is slower: 6.52 vs 6.31 ms)
    const uint32_t ks = blockIdx.y * a.Kc + tid / U, u = tid % U, b = b0 + u;
    if (b >= a.d.B || ks >= K) return;
    const uint32_t in_n = s_n[u], mdl_n = a.tpl_frames_s[ks];
    uint32_t score = SR_DIS_ERR;
    if (in_n && mdl_n && !(in_n > mdl_n * 2 || 2 * in_n < mdl_n)) {  // main.c:283, DTW.C:133-137
        const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142
        const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
        const int c1s = 3 - ((int)in_n - 2 * (int)mdl_n), c2s = ((int)mdl_n - 2 * (int)in_n) - 3;
        // cursors: input rows x-1 / x (0-based) live in registers and are re-read from LDS only when x advances;
        // the NEXT template row comes from HBM/L2 when y advances.  Rows x / y always exist inside the loop (x < in_n <= R
        // and y < mdl_n < tpl_rows; for 1-frame sequences row 1 is the slack row the reference's do-while reads,
        // DTW.C:150-154); the reload after the last advance may touch one row past the utterance's image, which the
        // launch pads for.
        uint32_t in_off = lds_offset(s_rows + (size_t)u * (row_stride / 2));  // LDS byte offsets of row x-1 and its norm
        uint32_t nrm_off = lds_offset(s_nrm + (size_t)u * nrm_stride);
        // template rows through a 32-bit byte offset from the (uniform) table base: advancing it is ONE add, and the
        // loads take the base from SGPRs (a 64-bit per-lane pointer costs an add-with-carry pair per advance)
        const char *tbase = (const char *)a.tplR;
        uint32_t t_off = ks * 32u;
        const uint32_t t_stride = K * 32u;  // bytes per template row level (K * 32 * rows < 2^32: checked at upload)
        auto tpl_row = [&](uint32_t off) {
            const u32x4 *q = (const u32x4 *)(tbase + off);
            return row_from(q[0], q[1]);  // (a 16 + 12 byte pair of loads, skipping the pad word, is slower: 6.44 -> 6.67 ms)
        };
        Row32 cm = tpl_row(t_off);
        t_off += t_stride;
        Row32 nm = tpl_row(t_off);
        Row32 ci, ni;
        lds_rows2(in_off, nrm_off, ci, ni);
        uint32_t dis = cvt_u32(sqrt_rn_int((float)(uint32_t)dot_rows_acc(cm, ci, (int)(cm.w[6] + ci.w[6]))));  // DTW.C:146
        // dtw_limit (DTW.C:76-109) as an interval test per column: (x', y') is inside  <=>  lb(x') <= y' <= ub(x')
        //   ub(x') = x' < X1 ? 2x'+1 : (x'+3-c1) >> 1      (negation of DTW.C:78-91; >> floors)
        //   lb(x') = x' < X2 ? x' >> 1 : 2x'+c2-3           (negation of DTW.C:93-106)
        // Carried: ub of column x (A), lb and ub of column x+1 (B); one new column is evaluated when x advances.
        // lb(x) is not needed: the walk only ever moves to admissible points, so lb(x) <= y holds for the current point
        // and (x, y+1) can only leave through ub(x).  The one exception -- all three candidates outside, DTW.C:156-184
        // then moves diagonally to an outside point -- makes the lane `lost`: from then on it takes the literal path.
        // What is carried is y1 = y + 1 and the upper bounds PLUS ONE, so that every test is one compare of carried values:
        //   (x, y+1) inside    <=>  y1 <  ubA1                 (ubA1 = ub(x) + 1)
        //   (x+1, y) inside    <=>  lbB <  y1  &&  y1 <= ubB1  (ubB1 = ub(x+1) + 1)
        //   (x+1, y+1) inside  <=>  lbB <= y1  &&  y1 <  ubB1
        const int c1s2 = c1s + 2;
        auto ub1_of = [&](int xx) { return (xx < X1) ? 2 * xx + 2 : ((xx + c1s2) >> 1); };
        auto lb_of = [&](int xx) { return (xx < X2) ? (xx >> 1) : 2 * xx + c2s; };
        int xB = 2, y1 = 2;  // xB = x + 1, y1 = y + 1; DTW.C:147-148
        int ubA1 = ub1_of(1), lbB = lb_of(2), ubB1 = ub1_of(2);
        uint64_t lost = 0;  // lane mask (kept scalar: OR-ed into the wave-uniform branch condition without touching the VALU)
        uint32_t step = 1;  // u16 in the reference; cannot wrap here (steps < in_n + mdl_n <= 2R, R bounded by LDS)
        do {
            // all three candidate squared distances, unconditionally: |m|^2 + |i|^2 + (-2m).i, the norm sum seeds
            // the dot2 accumulator (template rows are stored as -2m, see upload_templates)
            // (x+1, y) first: it needs neither the template row that may still be in flight from the previous step's y advance
            const uint32_t d_rt = (uint32_t)dot_rows_acc(cm, ni, (int)(cm.w[6] + ni.w[6]));  // (x+1, y):   get_dis(mdl, in+12)
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t d_up = (uint32_t)dot_rows_acc(nm, ci, (int)(nm.w[6] + ci.w[6]));  // (x, y+1):   get_dis(mdl+12, in)
            const uint32_t d_dg = (uint32_t)dot_rows_acc(nm, ni, (int)(nm.w[6] + ni.w[6]));  // (x+1, y+1)
            bool in_up = (y1 < ubA1), in_rt = (lbB < y1) & (y1 <= ubB1), in_dg = (lbB <= y1) & (y1 < ubB1);
            // DTW.C:152-184 on the SQUARED candidates.  g(d) = (u32)sqrtf((float)d) is monotone, so the step cost is
            // g(min of the admissible candidates) -- one root instead of three -- and "min == right_up" / "min == up"
            // (the tie order of DTW.C:168-184) become g(q) == g(min)  <=>  q < T, T = first d with g(d) = g(min)+1.
            // T is (g+1)^2 up to the rounding of (float)d; the exact value comes from a byte table in LDS (sr_tables.cpp:
            // T(g) = g*(g+2) + tie_delta[g]), so there is no uncertain band around it.  A root outside the staged part of
            // the table (which includes "all three outside"), an unsafe bracket or a lost lane sends the wave down the
            // literal three-root path.
            // minimum over the admissible candidates: one select and two v_min_u32 under the admissibility masks
            // (the masked-out candidates are never materialised; 0xFFFFFFFF when none is admissible)
            const uint64_t m_up = __builtin_amdgcn_ballot_w64(y1 < ubA1),
                           m_rt = __builtin_amdgcn_ballot_w64(lbB < y1) & __builtin_amdgcn_ballot_w64(y1 <= ubB1),
                           m_dg = __builtin_amdgcn_ballot_w64(lbB <= y1) & __builtin_amdgcn_ballot_w64(y1 < ubB1);
            uint32_t m2;
            {
                uint64_t ex;
                // m_dg is the result of an s_and_b64, i.e. SALU-written: no VALU-write -> VALU-read SGPR hazard on the select
                asm volatile("v_cndmask_b32_e64 %0, -1, %5, %2\n\t"
                             "s_mov_b64 %1, exec\n\t"
                             "s_and_b64 exec, %1, %3\n\t"
                             "v_min_u32 %0, %0, %6\n\t"
                             "s_and_b64 exec, %1, %4\n\t"
                             "v_min_u32 %0, %0, %7\n\t"
                             "s_mov_b64 exec, %1"
                             : "=&v"(m2), "=&s"(ex)
                             : "s"(m_dg), "s"(m_rt), "s"(m_up), "v"(d_dg), "v"(d_rt), "v"(d_up)
                             : "scc");
            }
            // the conditions that send the wave down the literal path are collected as LANE MASKS (ballots of the plain
            // compares, combined on the scalar unit): a bool OR-ed together and balloted afterwards costs two extra VALU ops
            const float s0 = __builtin_amdgcn_sqrtf((float)m2);
            uint32_t mn = (uint32_t)__int_as_float(__float_as_int(s0) + 1);  // floor(succ(s0)), see sqrt_floor_bracket
            // T = first squared distance whose root is mn + 1 = mn*(mn + 2) + tie_delta[mn] (exact, sr_tables.cpp): a
            // candidate q >= m2 has the root mn  <=>  q < T.  One byte from the LDS table (address register = the root,
            // base = immediate offset), one add, one 24-bit multiply-add.  Roots >= tie_g -- which includes m2 = 0xFFFFFFFF,
            // "all three outside" -- read past the table (out-of-range LDS reads return 0) and go down the literal path.
            uint32_t T;
            {
                const int dl = *(lds_c_i8 *)(uintptr_t)mn;  // s_tie[mn]: the table sits at LDS address 0
                const uint32_t mp2 = mn + 2;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(T) : "v"(mn), "v"(mp2), "v"(dl));
            }
            const bool tie_dg = in_dg & (d_dg < T), tie_up = in_up & (d_up < T);
            // hard = the lanes that need the literal form; a failed bracket alone (about 5e-4 of the lane-steps, i.e. 3 % of the
            // wave-steps) is settled exactly and cheaply inside the branch
            const uint64_t hard = __builtin_amdgcn_ballot_w64(mn >= a.tie_g) | lost;
            const uint64_t unsafe = __builtin_amdgcn_ballot_w64(!(s0 > (float)mn)) | hard;
            bool mv_diag = tie_dg, mv_up = tie_up & !tie_dg;
#ifdef SR_DTW_STATS
            {
                const uint64_t mb = __builtin_amdgcn_ballot_w64(!(s0 > (float)mn)), mr = __builtin_amdgcn_ballot_w64(mn >= a.tie_g),
                               act = __builtin_amdgcn_ballot_w64(true);
                if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)) == 0) {  // first active lane
                    atomicAdd(&g_dtw_stats[0], 1ull);
                    atomicAdd(&g_dtw_stats[5], (unsigned long long)__builtin_popcountll(act));
                    if (unsafe) atomicAdd(&g_dtw_stats[1], 1ull);
                    if (mb) atomicAdd(&g_dtw_stats[2], 1ull);
                    if (mr) atomicAdd(&g_dtw_stats[3], 1ull);
                    if (lost) atomicAdd(&g_dtw_stats[4], 1ull);
                }
            }
#endif
            if (unsafe != 0ull && hard == 0ull) {
                // Only the bracket is in doubt: mn = floor(succ(s0)) is the root or one too many.  With k = mn the exact test
                // g(d) < k  <=>  fmaf(-k, pred(k), (float)d) <= 0 settles it (k - h, h = half an ulp below k, is where sqrt
                // rounds up to k; (k - h)^2 = k*pred(k) + h^2 and (float)d - k*pred(k) is a multiple of 4h^2, so the sign
                // of the fused residual decides; checked against the C expression for every k and d around k^2 and on
                // 2e8 random d, tests/test_oracle.py).  Lanes whose bracket was safe keep their mn.  Then the threshold and
                // the two tie tests once more.
                const float kf = (float)mn, f2 = (float)m2;
                if (__builtin_fmaf(-kf, __int_as_float(__float_as_int(kf) - 1), f2) <= 0.0f) mn -= 1;
                const int dl = *(lds_c_i8 *)(uintptr_t)mn;
                const uint32_t T2 = mn * (mn + 2) + (uint32_t)dl;
                mv_diag = in_dg & (d_dg < T2);
                mv_up = in_up & (d_up < T2) & !mv_diag;
            } else if (unsafe != 0ull) {  // wave-uniform; the literal form: dtw_limit on the three points, three roots, min, equality tests
                const int x = xB - 1, y = y1 - 1;
                in_up = !dtw_out(x, y1, X1, X2, (int)in_n, (int)mdl_n);
                in_rt = !dtw_out(xB, y, X1, X2, (int)in_n, (int)mdl_n);
                in_dg = !dtw_out(xB, y1, X1, X2, (int)in_n, (int)mdl_n);
                const uint32_t up = in_up ? cvt_u32(sqrt_rn_int((float)d_up)) : SR_DIS_ERR,
                               right = in_rt ? cvt_u32(sqrt_rn_int((float)d_rt)) : SR_DIS_ERR,
                               diag = in_dg ? cvt_u32(sqrt_rn_int((float)d_dg)) : SR_DIS_ERR;
                mn = diag;  // DTW.C:156-164
                if (mn > right) mn = right;
                if (mn > up) mn = up;
                mv_diag = (mn == diag);  // DTW.C:168-184
                mv_up = !mv_diag && (mn == up);
                lost |= __builtin_amdgcn_ballot_w64(!(in_up | in_rt | in_dg));
            }
            dis += mn;
            const bool adv_y = mv_diag || mv_up, adv_x = mv_diag || !mv_up;
            // the y advance goes first: its template-row loads come from L2 and have the longest way to go before the next
            // step's distances need them (the kernel runs close to where a wave's serial latency, not the issue port, sets
            // the pace: 6 waves per SIMD, LDS-limited; this order alone is worth 6 % of the kernel's time)
            if (adv_y) {
                y1++;
                copy_row(cm, nm);
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(t_off) : "s"(t_stride));
                nm = tpl_row(t_off);
            }
            if (adv_x) {
                // in-place updates (tied asm operands): without them the compiler builds the new values in fresh
                // registers and copies them into the loop-carried ones at the end of the block (three v_mov per step)
                asm volatile("v_add_u32 %0, 1, %0" : "+v"(xB));
                asm volatile("v_add_u32 %0, 24, %0" : "+v"(in_off));
                asm volatile("v_add_u32 %0, 4, %0" : "+v"(nrm_off));
                lds_rows2(in_off, nrm_off, ci, ni);
                asm volatile("v_mov_b32 %0, %1" : "+v"(ubA1) : "v"(ubB1));
                {
                    int ua, lb;  // (xB << 1) + constant as ONE v_lshl_add_u32 each (the compiler shares 2*xB and spends two adds)
                    asm("v_lshl_add_u32 %0, %1, 1, 2" : "=v"(ua) : "v"(xB));
                    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(lb) : "v"(xB), "v"(c2s));
                    const int ub = (xB + c1s2) >> 1, la = xB >> 1;
                    const uint64_t m1 = __builtin_amdgcn_ballot_w64(xB < X1), m2x = __builtin_amdgcn_ballot_w64(xB < X2);
                    // s_nop 1: the masks come straight from v_cmp, and a VALU read of an SGPR written by the VALU needs two
                    // wait states (the compiler pads its own v_cmp -> v_cndmask pairs the same way)
                    asm volatile("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(ubB1) : "v"(ub), "v"(ua), "s"(m1));
                    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(lbB) : "v"(lb), "v"(la), "s"(m2x));
                }
            }
            step++;
        } while (xB <= (int)in_n && y1 <= (int)mdl_n);  // DTW.C:188 (x < in && y < mdl)
        score = dis / step;
    }
    a.d.scores[(size_t)b * K + a.tpl_orig[ks]] = score;
}

// pick U: maximise resident lanes doing useful work (LDS 160 KiB/CU, 32 waves/CU, 1024 threads/workgroup), then give
// what is left of the workgroup's LDS share to the tie-threshold table (tie_g entries of one byte, a power of two).
// gfx950 hands out LDS in granules of 1280 bytes (160 KiB / 128): three workgroups fit a CU only if each stays within
// 42 granules = 53 760 bytes -- 20 bytes more and the third one silently does not (measured: mean waves per SIMD 5.0 -> 3.3),
// although hipOccupancyMaxActiveBlocksPerMultiprocessor still reports 3.
constexpr size_t kLdsGranule = 1280, kCuLds = 160 * 1024;
__host__ __device__ inline size_t dtw_lds_fixed(uint32_t U, uint32_t max_frames)
{
    const size_t per_u = (size_t)(dtw_lds_row_stride(max_frames) + dtw_lds_nrm_stride(max_frames)) * 4;
    // + 32: the reload after the last advance may read one row / two norms past the last utterance's image;
    // + the frame counts of the U utterances
    return U * per_u + 32 + ((4 * (size_t)U + 15) & ~(size_t)15);
}
// Geometry of a k_dtw_lds launch for a store of K templates and max_frames rows per utterance: U utterances and Kc
// templates (ranks) per workgroup (U * Kc <= 1024 lanes; the store is walked in ceil(K / Kc) equal chunks over grid.y), the
// LDS bytes, and the tie-table entries that fit beside the utterances.  What counts: the fraction of lanes that carry a pair,
// enough resident waves to cover the LDS / L2 latency of the walk, and -- measured at K = 500 -- how many lanes share a
// template row: the texture addresser is the limit there, and U = 6 x 167 templates runs 12 % faster than U = 2 x 500
// (30.2 vs 34.3 ms per 65 536 utterances), while U = 10 x 100 (one workgroup per CU) loses 10 %.
uint32_t dtw_lds_pick_u(uint32_t K, uint32_t max_frames, size_t *lds_bytes, uint32_t *tie_g, uint32_t *kc_out)
{
    const uint32_t kMinTie = 4096;  // below 4096 every threshold is the exact square: the least useful table
    auto blocks_for = [](size_t lds) { return (uint32_t)(kCuLds / ((lds + kLdsGranule - 1) / kLdsGranule * kLdsGranule)); };
    uint32_t best_u = 0, best_g = 0, best_kc = 0;
    double best = 0;
    const char *force = getenv("SR_DTW_U");  // tuning overrides
    const char *force_g = getenv("SR_DTW_TIE_G");
    const char *force_kc = getenv("SR_DTW_KC");
    for (uint32_t U = 1; U <= (uint32_t)kDtwMaxU; U++) {
        uint32_t kc = K < 1024u / U ? K : 1024u / U;  // lanes of a workgroup
        if (force_kc && atoi(force_kc) >= 1 && (uint32_t)atoi(force_kc) < kc) kc = (uint32_t)atoi(force_kc);
        if (!kc) break;
        const uint32_t chunks = (K + kc - 1) / kc;
        kc = (K + chunks - 1) / chunks;             // equal chunks
        const uint64_t pairs = (uint64_t)U * kc;
        const size_t lds = dtw_lds_fixed(U, max_frames);
        if (lds + kMinTie > 150 * 1024) break;
        const uint32_t waves = (uint32_t)((pairs + 63) / 64);
        uint32_t blocks = blocks_for(lds + kMinTie);
        if (blocks > 32 / waves) blocks = 32 / waves;
        if (blocks > 8) blocks = 8;
        if (blocks < 1) continue;
        uint32_t g = kMinTie;  // the largest table that does not cost a resident workgroup
        while (g < (uint32_t)kTieMax && blocks_for(lds + 2 * g) >= blocks) g *= 2;
        if (force_g && atoi(force_g) >= (int)kMinTie && atoi(force_g) <= kTieMax) g = (uint32_t)atoi(force_g) & ~1023u;
        const double eff = (double)K / ((double)chunks * 64.0 * waves / U);  // lanes that carry a pair, over all chunks
        const double resident = (double)(blocks * waves);
        // lanes per template row: worth more the larger the store (at K = 100 five utterances x 100 templates in three
        // workgroups per CU beat eight x 100 in two, 6.4 vs 6.6 ms; at K = 500 eight x 125 beat two x 500, 31 vs 34 ms);
        // every further chunk stages the utterances once more
        const double w = 0.5 * (K >= 400 ? 1.0 : K / 400.0);
        // a table that ends at a root of 8192 (4096) sends squared distances above 6.7e7 (1.7e7) down the literal path
        const double cover = g >= 16384 ? 1.0 : g >= 8192 ? 0.98 : 0.9;
        const double share = (1.0 - w / U) * (1.0 - 0.005 * (chunks - 1)) * cover;
        double score = eff * (resident >= 24 ? 1.0 : resident / 24.0) * share;
        if (force && (uint32_t)atoi(force) == U) score = 100.0;
        if (score > best + 1e-9) {
            best = score;
            best_u = U;
            best_g = g;
            best_kc = kc;
        }
    }
    if (best_u && lds_bytes) *lds_bytes = dtw_lds_fixed(best_u, max_frames) + best_g;
    if (tie_g) *tie_g = best_g;
    if (kc_out) *kc_out = best_kc;
    return best_u;
}

void launch_dtw(const DtwArgs &a, hipStream_t s)
{
    const uint64_t n = (uint64_t)a.B * a.K;
    if (!n) return;
    // U (utterances per workgroup) and the LDS size were chosen once, when the template store was set
    const uint32_t U = a.tplR ? a.lds_u : 0;
    const size_t lds = a.lds_bytes;
    if (U) {
        const uint32_t Kc = (a.lds_kc && a.lds_kc < a.K) ? a.lds_kc : a.K, chunks = (a.K + Kc - 1) / Kc;
        DtwLdsArgs la{a, (const u32x4 *)a.tplR, a.tpl_frames_s, a.tpl_orig, U, a.tie_delta, a.tie_g, Kc};
        const uint32_t threads = (uint32_t)(((uint64_t)U * Kc + 63) / 64 * 64);
        hipLaunchKernelGGL(k_dtw_lds, dim3((a.B + U - 1) / U, chunks), dim3(threads), lds, s, la);
    } else {  // very long sequences / very many templates: generic global-memory walk
        hipLaunchKernelGGL(k_dtw, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, s, a);
    }
}

// argmin with strict '<' in slot order (main.c:276-291): first minimum wins; all dis_err -> slot 0
__global__ void __launch_bounds__(256) k_argmin(const DtwArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const uint32_t *sc = a.scores + (size_t)b * a.K;
    uint32_t best = SR_DIS_ERR, idx = 0xFFFFFFFFu;
    for (uint32_t k = lane; k < a.K; k += 64) {
        const uint32_t d = sc[k];
        if (d < best) {
            best = d;
            idx = k;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t ob = __shfl_xor(best, d, 64), oi = __shfl_xor(idx, d, 64);
        if (ob < best || (ob == best && oi < idx)) {
            best = ob;
            idx = oi;
        }
    }
    if (lane == 0) {
        sr_result r;
        r.best_tpl = (best == SR_DIS_ERR) ? 0u : idx;
        r.min_dis = best;
        if (a.in_frames) {
            r.frm_num = a.in_frames[b];
            r.status = SR_ST_OK;
        } else {
            r.frm_num = a.vad[b].frm_num;
            r.status = a.vad[b].status;
        }
        a.results[b] = r;
    }
}

void launch_argmin(const DtwArgs &a, hipStream_t s)
{
    if (!a.B) return;
    hipLaunchKernelGGL(k_argmin, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// k_dtw_dp: OPT-IN, NON-REFERENCE scorer.  The reference's dtw() is a greedy local walk;
// this kernel is the classic dynamic-programming DTW the project brief describes: the anti-diagonal
// wavefront lives in the 64 lanes of a wave (lane = one utterance frame / column, rows advance skewed by one
// per lane), the left/diagonal neighbours arrive by __shfl_up, the template is staged in LDS, and the same
// relaxed parallelogram (dtw_limit, DTW.C:76-109) and local distance (get_dis, DTW.C:45-62) are used:
//   D(1,1) = d(1,1);  D(x,y) = d(x,y) + min(D(x-1,y-1), D(x-1,y), D(x,y-1)) over cells inside the parallelogram;
//   score = D(in,mdl) / (in + mdl)   (dis_err if the lengths fail the 1/2..2x gate or the end cell is unreachable).
// It never backs the dtw() symbol or the recognition path; it has its own oracle (sr_oracle_dtw_dp).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kDpInf = 0xFFFFFFFFu;

__global__ void __launch_bounds__(256) k_dtw_dp(const DtwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32x4 dp_smem[];  // template rows: [tpl_rows][2] u32x4
    const uint32_t k = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.y * 4 + w;
    const uint32_t mdl_n = a.tpl_valid[k] ? a.tpl_frames[k] : 0u;
    uint32_t *s_col = (uint32_t *)(dp_smem + (size_t)a.tpl_rows * 2) + (size_t)w * a.tpl_rows;  // boundary column per wave
    // stage the template (24-byte rows + squared norm) once per workgroup
    for (uint32_t r = threadIdx.x; r < a.tpl_rows; r += blockDim.x) {
        const uint2 *src = (const uint2 *)(a.tpl + (size_t)k * a.tpl_stride + (size_t)r * kCoef);
        const uint2 q0 = src[0], q1 = src[1], q2 = src[2];
        int nr = sdot2z(q0.x, q0.x);
        nr = sdot2(q0.y, q0.y, nr);
        nr = sdot2(q1.x, q1.x, nr);
        nr = sdot2(q1.y, q1.y, nr);
        nr = sdot2(q2.x, q2.x, nr);
        nr = sdot2(q2.y, q2.y, nr);
        dp_smem[2 * r] = u32x4{q0.x, q0.y, q1.x, q1.y};
        dp_smem[2 * r + 1] = u32x4{q2.x, q2.y, (uint32_t)nr, 0u};
    }
    __syncthreads();
    if (b >= a.B) return;
    uint32_t in_n;
    if (a.in_frames) in_n = a.in_frames[b];
    else in_n = (a.vad[b].status == SR_ST_OK) ? a.vad[b].frm_num : 0u;
    uint32_t score = SR_DIS_ERR;
    if (in_n && mdl_n && !(in_n > mdl_n * 2 || 2 * in_n < mdl_n)) {
        const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF), X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
        const int16_t *in = a.mfcc + (size_t)b * a.max_frames * kCoef;
        uint32_t d_end = kDpInf;
        for (uint32_t x0 = 0; x0 < in_n; x0 += 64) {  // 64 columns at a time, boundary column kept in LDS
            const uint32_t col = x0 + lane;           // 0-based utterance frame
            const bool live = col < in_n;
            Row32 fi;
            {
                const uint2 *src = (const uint2 *)(in + (size_t)(live ? col : 0) * kCoef);
                const uint2 q0 = src[0], q1 = src[1], q2 = src[2];
                fi = row_from2(u32x2{q0.x, q0.y}, u32x2{q1.x, q1.y}, u32x2{q2.x, q2.y}, 0u);
                fi.w[6] = (uint32_t)dot_rows(fi, fi);
            }
            uint32_t up = kDpInf;    // D(col, row-1), own previous step
            uint32_t left = kDpInf;  // D(col-1, row) as delivered last step = this step's diagonal
            const uint32_t steps = mdl_n + 63;
            for (uint32_t t = 0; t < steps; t++) {
                const int row = (int)t - (int)lane;  // 0-based template frame of this lane at this step
                // value of the lane to the left at the SAME row was produced one step ago
                uint32_t from_left = __shfl_up(up, 1, 64);
                if (lane == 0) from_left = (x0 == 0 || row < 0 || row >= (int)mdl_n) ? kDpInf : s_col[row];
                const uint32_t diag = left;  // D(col-1, row-1)
                uint32_t cur = kDpInf;
                const bool in_range = live && row >= 0 && row < (int)mdl_n;
                if (in_range) {
                    const int x = (int)col + 1, y = row + 1;
                    if (!dtw_out(x, y, X1, X2, (int)in_n, (int)mdl_n)) {
                        const Row32 fm = row_from(dp_smem[2 * row], dp_smem[2 * row + 1]);
                        const uint32_t d = dis_from(fi.w[6], fm.w[6], dot_rows(fi, fm));
                        uint32_t best = diag < from_left ? diag : from_left;
                        best = up < best ? up : best;
                        if (col == 0 && row == 0) best = 0;  // D(1,1) = d(1,1)
                        if (best != kDpInf) {
                            const uint32_t sum = best + d;
                            cur = sum < best ? 0xFFFFFFFEu : (sum == kDpInf ? 0xFFFFFFFEu : sum);  // saturate below INF
                        }
                    }
                }
                left = from_left;
                if (in_range) up = cur;
                // last column of the chunk publishes its values for the next chunk
                if (lane == 63 && in_range) s_col[row] = cur;
                if (in_range && col == in_n - 1 && row == (int)mdl_n - 1) d_end = cur;
            }
            wave_sync();
        }
        // the end cell lives in exactly one lane
        d_end = wave_min_u32(d_end);
        if (d_end != kDpInf) score = d_end / (in_n + mdl_n);
    }
    if (lane == 0) a.scores[(size_t)b * a.K + k] = score;
}

void launch_dtw_dp(const DtwArgs &a, hipStream_t s)
{
    if (!a.B || !a.K) return;
    const size_t lds = (size_t)a.tpl_rows * 32 + (size_t)4 * a.tpl_rows * 4;
    hipLaunchKernelGGL(k_dtw_dp, dim3(a.K, (a.B + 3) / 4), dim3(256), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// k_delta_mfcc: EXTENSION, no reference counterpart (the accompanying thesis, p.32, lists difference cepstra as future
// work).  Two-frame regression over the s16 MFCC rows of a record, d[t] = ((m[t+1]-m[t-1]) + 2(m[t+2]-m[t-2])) / 10 with
// rows clamped to [0, n-1], s32 arithmetic, division truncating toward zero (as every division of MFCC.C); rows >= n
// are zero.  Defined in oracle/sr_oracle.c (sr_oracle_delta_mfcc); pure streaming: one thread per output element.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_delta_mfcc(const int16_t *mfcc, const sr_vad_rec *vad, const uint32_t *frames,
                                                    uint32_t B, uint32_t max_frames, int16_t *delta)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per = max_frames * kCoef;
    if (i >= (uint64_t)B * per) return;
    const uint32_t b = (uint32_t)(i / per), r = (uint32_t)(i - (uint64_t)b * per), t = r / kCoef, c = r - t * kCoef;
    uint32_t n = frames ? frames[b] : ((vad[b].status == SR_ST_OK) ? vad[b].frm_num : 0u);
    if (n > max_frames) n = max_frames;
    int16_t out = 0;
    if (t < n) {
        const int16_t *m = mfcc + (uint64_t)b * per + c;
        const uint32_t p1 = t + 1 < n ? t + 1 : n - 1, p2 = t + 2 < n ? t + 2 : n - 1;
        const uint32_t m1 = t >= 1 ? t - 1 : 0, m2 = t >= 2 ? t - 2 : 0;
        const int num = ((int)m[p1 * kCoef] - (int)m[m1 * kCoef]) + 2 * ((int)m[p2 * kCoef] - (int)m[m2 * kCoef]);
        out = (int16_t)(num / 10);
    }
    delta[i] = out;
}
void launch_delta_mfcc(const int16_t *mfcc, const sr_vad_rec *vad, const uint32_t *frames, uint32_t B, uint32_t max_frames,
                       int16_t *delta, hipStream_t s)
{
    const uint64_t n = (uint64_t)B * max_frames * kCoef;
    if (!n) return;
    hipLaunchKernelGGL(k_delta_mfcc, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, mfcc, vad, frames, B, max_frames, delta);
}

// ------------------------------------------------------------------------------------------------
// diagnostics: the three non-integer device functions on their own, so tests can sweep them directly
//   out[3i+0] = (u32)(log((double)x)*100)                       MFCC.C:168   (step-function evaluation)
//   out[3i+1] = (u32)sqrtf((float)x)                            DTW.C:59     (v_rsq_f32 seed + fused correction, sqrt_rn_int)
//   out[3i+2] = (u32)(sqrtf((float)(s32)x)*10), x < 2^31        MFCC.C:56-58
// ------------------------------------------------------------------------------------------------
__global__ void k_math_diag(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *log_thr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = in[i];
    out[3 * i + 0] = log100_u32(x, log_thr);
    // the exact routine and the bracketed fast path of the DTW kernel are reported through one word: the exact value,
    // or a poison value if the fast path claims "safe" and disagrees (tests/exhaustive_math_sweep.py: all 2^32 inputs)
    {
        bool unsafe = false;
        const uint32_t q = sqrt_floor_bracket(x, unsafe), e = cvt_u32(sqrt_rn_int((float)x));
        out[3 * i + 1] = (unsafe || q == e) ? e : 0xDEAD0001u;
    }
    out[3 * i + 2] = cvt_u32(sqrt_rn_int((float)(int)(x & 0x7FFFFFFFu)) * 10.0f);
}
void launch_math_diag(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables &t, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_math_diag, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, t.log_thr);
}

// ------------------------------------------------------------------------------------------------
// scalar helpers of DTW.C exposed by the reference-compatible symbols
// ------------------------------------------------------------------------------------------------
__global__ void k_get_dis(const int16_t *pa, const int16_t *pb, uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Frame12 fa = load_frame(pa + (size_t)i * kCoef), fb = load_frame(pb + (size_t)i * kCoef);
    out[i] = get_dis_dev(fa, norm2(fa), fb, norm2(fb));
}
void launch_get_dis(const int16_t *pa, const int16_t *pb, uint32_t *out, uint32_t n, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_get_dis, dim3((n + 63) / 64), dim3(64), 0, s, pa, pb, out, n);
}

__global__ void k_dtw_limit(const uint16_t *xy, uint8_t *out, uint32_t n, int X1, int X2, int in_n, int mdl_n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dtw_out((int)xy[2 * i], (int)xy[2 * i + 1], X1, X2, in_n, mdl_n) ? 1 : 0;
}
void launch_dtw_limit(const uint16_t *xy, uint8_t *out, uint32_t n, int X1, int X2, int in_n, int mdl_n, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_dtw_limit, dim3((n + 63) / 64), dim3(64), 0, s, xy, out, n, X1, X2, in_n, mdl_n);
}

}  // namespace sr
