// sr_fft_dev.h -- the Q15 radix-4 butterfly of cr4_fft_1024_stm32.s in packed 16-bit arithmetic, coefficient loading, and the register/LDS layouts shared by k_mfcc, k_mfcc_ext and the generic FFT kernels.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#pragma once
#include "sr_dev.h"

namespace sr {

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
// A_SHIFTED: x0_in already holds A >> 2 (k_mfcc stores the samples that are pass-2 A legs that way)
template <bool HALF, bool HAS_B, bool CD_SAME = false, bool A_SHIFTED = false>
__device__ __forceinline__ void r4_packed(uint32_t x0_in, int br, int bi, int sr, int si, int tr, int ti, uint32_t &x0,
                                          uint32_t &x1, uint32_t &x2, uint32_t &x3)
{
    const uint32_t a = A_SHIFTED ? x0_in : pk_ashr(x0_in, 2);
    uint32_t A1 = a, B1 = a;
    if (HAS_B) {
        A1 = pk_add(a, pk_hi16(br, bi));
        B1 = pk_sub(A1, pk_s15(br, bi));
    }
    const uint32_t hC = pk_hi16(sr, si);
    x0 = pk_add(A1, hC);
    // D' with its halves swapped: (D'i>>16, D'r>>16).  CD_SAME: they are C''s own halves, exchanged by the multiply-add's operand
    // selects (round 6; a v_alignbit_b32 each before: 8 instructions per frame in pass 2)
    x1 = CD_SAME ? pk_mad_swap(hC, kPkPlusMinus, B1) : pk_mad(pk_hi16(ti, tr), kPkPlusMinus, B1);
    if (!HALF) {
        const uint32_t pC = pk_s15(sr, si);
        x2 = pk_sub(x0, pC);
        x3 = CD_SAME ? pk_mad_swap(pC, kPkMinusPlus, x1) : pk_mad(pk_s15(ti, tr), kPkMinusPlus, x1);
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
// pass-3 coefficients depend on (d0, d1) only: 4 x 32 words, read with lane-broadcast by d0 = lane & 3.  The four rows are
// kTw3Row = 9 chunks apart, not 8: the lanes of a ds_read_b128 service group hold all four d0, and with rows of 32 words
// d0 = 0 / 2 (and 1 / 3) fall on the same banks with different addresses -- every one of the loads was a 2-way conflict
// (round 6: 32 of k_mfcc's 53 conflict cycles per frame, profiles/experiments/RESULTS.md)
constexpr int kTw3Row = 9;
__device__ __forceinline__ void load_tw32_d0(const u32x4 *lds, int d0, uint32_t (&k)[4][4][2])
{
    uint32_t *f = &k[0][0][0];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const u32x4 q = lds[d0 * kTw3Row + c];
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
// xw: the windowed frame as one 32-bit word per sample (16-bit pattern in the low half, the high half is never read): 4-byte
// stores of consecutive samples and the 2-byte reads below at word base + 16 m are both conflict-free; with two samples per
// word the stores of lanes 2 m / 2 m + 1 went to one word
__device__ __forceinline__ void fft_front_real160(const uint32_t *xw, int lane, const LaneTw &tw, const u32x4 *tw3_lds,
                                                  uint32_t (&v)[4][4])
{
    const int d3 = (lane >> 2) & 3, d4 = lane >> 4;
    const int base = rev2(d4) + 4 * rev2(d3);
    // bitrev8(j>>2) = base + 16*rev2(d2) + 64*rev2(d1); >= 160 <=> zero padding
    uint32_t y[10];
#pragma unroll
    // already A >> 2 as a 16-bit pattern (pass 1, .s:147-148); m < 4 = samples 0..63 = the A legs of pass 2, stored >> 2 once more
    for (int m = 0; m < 10; m++) y[m] = *(const uint16_t *)(xw + base + 16 * m);
    // (the A legs go into packed adds as they are: opaque, or the compiler re-zeroes halves that ds_read_u16 has just zero-extended)
#pragma unroll
    for (int m = 0; m < 4; m++) asm("" : "+v"(y[m]));
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
            r4_packed<false, true, true, true>(x0, br, bi, cr, ci, cr, ci, v[0][d2], v[1][d2], v[2][d2], v[3][d2]);
        } else {
            r4_packed<false, false, true, true>(x0, 0, 0, cr, ci, cr, ci, v[0][d2], v[1][d2], v[2][d2], v[3][d2]);
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

// ---- the full complex transform, one wave per array (used by the generic kernels: k_fft.hip, k_mfcc_gen.hip) ----
__device__ __forceinline__ int bitrev8(int v) { return (int)(__brev((uint32_t)v) >> 24); }

// Full-complex passes 1-5, result in natural order in `out` (global).  in/out may not alias in LDS terms.
__device__ inline void fft_full_wave(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *buf, int lane,
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

}  // namespace sr
