// k_mfcc_gen.hip -- get_mfcc (MFCC.C:86-191) for the GENERIC front end: any sampling rate and framing the VAD kernel is
// built for (frame_len <= 1024), the reference's 1024-point Q15 transform, 4..64 Mel filters (even), 1..16 coefficients.
// The reference's constants are compile-time (#defines in MFCC.H:7-16, VAD.H:4-8, pasted tables in MFCC_Arg.h); here they
// are run-time values with the tables generated from the same Matlab formulas (csrc/sr_tables.cpp), and the arithmetic is
// the reference's rule for rule, checked bit for bit against the parametrised oracle.  No reference counterpart for the
// constants themselves.  One wave per frame, the transform through LDS (fft_real_half below), the filterbank as prefix
// sums like k_mfcc: 2.3x the time of k_mfcc per frame; the two specialised kernels stay untouched.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; integer VALU + LDS.
#include "sr_fft_dev.h"

namespace sr {

constexpr int kGenWaves = 4;
constexpr int kGenMaxMel = 64;  // sr_create: n_mel <= 64, n_coef <= 16 (one lane per filter / coefficient)
// per-wave LDS words: transform input, output, exchange scratch, bin energies, filterbank outputs
constexpr int kGenWaveWords = kNfft + kNfft + kXchgWords + kBins + kGenMaxMel;

// The 1024-point transform of a zero-padded REAL frame, bins 0..511 only (MFCC.C:49 consumes no more): fft_full_wave of
// sr_fft_dev.h with (a) the lane's coefficients handed in (loaded once per wave, not once per frame), (b) pass 1
// (.s:226-232) on real samples -- every imaginary part is 0, so half of the S = 0 combine folds away --, (c) the last pass
// producing x[j], x[j+q] only.  Same arithmetic, checked bit for bit against the oracle like everything else.
struct GenTw {
    LaneTw tw;
    uint32_t k2[3][2];  // pass 2 (q = 4), all three legs (LaneTw::s2 is the reference kernel's two-leg form)
};
__device__ __forceinline__ void load_gen_tw(const DevTables &t, int lane, GenTw &g)
{
    load_lane_tw(t, lane, g.tw);
    load_tw3(t, 0, lane & 3, g.k2);
}
__device__ __forceinline__ void fft_real_half(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *buf, int lane,
                                              const GenTw &g)
{
    for (int m = 0; m < 4; m++) {
        const int idx = lane + 64 * m, r = bitrev8(idx);
        int ar = sext_lo(in[r]), br = sext_lo(in[r + 512]), cr = sext_lo(in[r + 256]), dr = sext_lo(in[r + 768]);
        int ai = 0, bi = 0, ci = 0, di = 0;
        r4_combine<0>(ar, ai, br, bi, cr, ci, dr, di);
        buf[xaddr(4 * idx + 0)] = pack16(ar, ai);
        buf[xaddr(4 * idx + 1)] = pack16(br, bi);
        buf[xaddr(4 * idx + 2)] = pack16(cr, ci);
        buf[xaddr(4 * idx + 3)] = pack16(di, dr);
    }
    wave_sync();
    const LaneTw &tw = g.tw;
    const int d0 = lane & 3, d3 = (lane >> 2) & 3, d4 = lane >> 4;
    uint32_t v[4][4], u[4][4];
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
#pragma unroll
        for (int d2 = 0; d2 < 4; d2++) v[d1][d2] = buf[xaddr(d0 + 4 * d1 + 16 * d2 + 64 * d3 + 256 * d4)];
    wave_sync();
#pragma unroll
    for (int d2 = 0; d2 < 4; d2++)
        bfly(v[0][d2], v[1][d2], v[2][d2], v[3][d2], g.k2[0][0], g.k2[0][1], g.k2[1][0], g.k2[1][1], g.k2[2][0], g.k2[2][1]);
#pragma unroll
    for (int d1 = 0; d1 < 4; d1++)
        bfly_pk<false>(v[d1][0], v[d1][1], v[d1][2], v[d1][3], tw.s3[d1][0][0], tw.s3[d1][0][1], tw.s3[d1][1][0], tw.s3[d1][1][1],
                       tw.s3[d1][2][0], tw.s3[d1][2][1], tw.s3[d1][3][0], tw.s3[d1][3][1]);
    fft_exchange(buf, lane, v, u);
#pragma unroll
    for (int e4 = 0; e4 < 4; e4++)
        bfly_pk<false>(u[0][e4], u[1][e4], u[2][e4], u[3][e4], tw.s4[0][0], tw.s4[0][1], tw.s4[1][0], tw.s4[1][1], tw.s4[2][0],
                       tw.s4[2][1], tw.s4[3][0], tw.s4[3][1]);
#pragma unroll
    for (int e3 = 0; e3 < 4; e3++) {
        bfly_pk<true>(u[e3][0], u[e3][1], u[e3][2], u[e3][3], tw.s5[e3][0][0], tw.s5[e3][0][1], tw.s5[e3][1][0], tw.s5[e3][1][1],
                      tw.s5[e3][2][0], tw.s5[e3][2][1], tw.s5[e3][3][0], tw.s5[e3][3][1]);
        out[lane + 64 * e3] = u[e3][0];        // x[j]
        out[lane + 64 * e3 + 256] = u[e3][1];  // x[j + q]
    }
}

__global__ void __launch_bounds__(64 * kGenWaves) k_mfcc_gen(const MfccArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *fin = smem + (size_t)w * kGenWaveWords, *fout = fin + kNfft, *buf = fout + kNfft, *en = buf + kXchgWords,
             *pw = en + kBins;
    const uint32_t FL = a.frame_len, hop = a.hop, nm = a.n_mel, nc = a.n_coef;
    const uint64_t n_items = (uint64_t)a.B * a.max_frames;
    // lane-invariant: triangle weights of the lane's 8 bins on both poly-lines, the bin range of the lane's filter, and the
    // lane's share of the DCT (coefficient c = lane / P, filters p, p + P, ... with P = 64 / n_coef lanes per coefficient)
    uint32_t tri_e[8], tri_o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        tri_e[k] = a.t.tri_even32[8 * lane + k];
        tri_o[k] = a.t.tri_odd32[8 * lane + k];
    }
    int f_lo = 0, f_hi = 0;
    if ((uint32_t)lane < nm) {
        f_lo = (lane == 0) ? 0 : (int)a.t.tri_cen[lane - 1];
        f_hi = ((uint32_t)lane == nm - 1) ? kBins : (int)a.t.tri_cen[lane + 1];
    }
    const uint32_t P = 64u / nc, dc = (uint32_t)lane / P, dp = (uint32_t)lane - dc * P;
    GenTw gtw;
    load_gen_tw(a.t, lane, gtw);
    // the zero padding of the transform's input (MFCC.C:37-47) is written once: fft_full_wave only reads `fin`
    for (uint32_t i = FL + lane; i < (uint32_t)kNfft; i += 64) fin[i] = 0;
    for (uint64_t item = (uint64_t)blockIdx.x * kGenWaves + w; item < n_items; item += (uint64_t)gridDim.x * kGenWaves) {
        const uint32_t b = (uint32_t)(item / a.max_frames), f = (uint32_t)(item - (uint64_t)b * a.max_frames);
        const sr_vad_rec *rec = a.vad + b;
        int16_t *out = a.mfcc + ((uint64_t)b * a.max_frames + f) * nc;
        if (f >= rec->frm_num) {  // rows >= frm_num are zero so that every row of the output is defined
            if ((uint32_t)lane < nc) out[lane] = 0;
            continue;
        }
        const int mid = (int)rec->atap.mid_val;
        const uint16_t *x = a.pcm + (uint64_t)b * a.pcm_stride + rec->seg[0] + (uint64_t)hop * f;
        // ---- pre-emphasis + Hamming (MFCC.C:115-124); x[-1] is the sample before the frame
        for (uint32_t i = lane; i < FL; i += 64) {
            const int cur = (int)x[i] - mid, prv = (int)x[(int)i - 1] - mid;
            const int t = cur - preemph95(prv);
            fin[i] = (uint32_t)(uint16_t)(int16_t)(mul24(t, (int)a.t.hamm[i]) / 1000);  // real part, imaginary half 0
        }
        wave_sync();
        fft_real_half(fin, fout, buf, lane, gtw);  // cr4_fft_1024_stm32 (.s:95-281), bins 0..511
        wave_sync();
        // ---- |X| * 10 and energy (MFCC.C:49-60, 128-133), u32 wrap: bins 8*lane .. 8*lane + 7 of this lane
        uint32_t e[8];
        {
            const u32x4 q0 = *(const u32x4 *)(fout + 8 * lane), q1 = *(const u32x4 *)(fout + 8 * lane + 4);
            const uint32_t wd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const f32x2 m = sqrt_rn_int2(f32x2{(float)sdot2z(wd[k], wd[k]), (float)sdot2z(wd[k + 1], wd[k + 1])}) * f32x2{10.0f, 10.0f};
                const uint32_t m0 = cvt_u32(m.x), m1 = cvt_u32(m.y);
                e[k] = m0 * m0;
                e[k + 1] = m1 * m1;
            }
        }
        // ---- Mel filterbank (MFCC.C:136-162) as prefix sums over bins, as in k_mfcc: every term is E*tri/100 BEFORE the
        //      summation and sums wrap mod 2^32, so the order is free; filter h = P[hi-1] - P[lo-1] on its poly-line
        uint32_t pe[8], po[8], se = 0, so = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            se += e[k] * tri_e[k] / 100u;
            so += e[k] * tri_o[k] / 100u;
            pe[k] = se;
            po[k] = so;
        }
        const uint32_t xe = wave_scan_incl(se) - se, xo = wave_scan_incl(so) - so;  // bins of the lanes below
        wave_sync();  // fout has been consumed by every lane: reuse it for the prefixes
#pragma unroll
        for (int k = 0; k < 8; k++) {
            fout[8 * lane + k] = pe[k] + xe;
            fout[kBins + 8 * lane + k] = po[k] + xo;
        }
        wave_sync();
        if ((uint32_t)lane < nm) {
            const uint32_t *Pp = fout + ((lane & 1) ? kBins : 0);
            const uint32_t hi = Pp[f_hi - 1], lo = f_lo ? Pp[f_lo - 1] : 0u;
            pw[lane] = log100_u32(hi - lo, a.t.log_thr);  // MFCC.C:165-170
        }
        wave_sync();
        // ---- DCT (MFCC.C:173-183): per-term truncating / 100, s16 accumulator (= the low 16 bits of the integer sum, so
        //      the terms of a coefficient may be summed by P lanes and combined)
        {
            int acc = 0;
            if (dc < nc) {
                const int8_t *d = a.t.dct + dc * nm;
                for (uint32_t h = dp; h < nm; h += P) acc += (int)(int16_t)((int)pw[h] * (int)d[h] / 100);
            }
            en[lane] = (uint32_t)acc;
        }
        wave_sync();
        if ((uint32_t)lane < nc) {
            int acc = 0;
            for (uint32_t p = 0; p < P; p++) acc += (int)en[(uint32_t)lane * P + p];
            out[lane] = (int16_t)acc;
        }
        wave_sync();
    }
}

void launch_mfcc_gen(const MfccArgs &a, hipStream_t s)
{
    const uint64_t n = (uint64_t)a.B * a.max_frames;
    if (!n) return;
    const uint64_t wgs = (n + kGenWaves - 1) / kGenWaves;
    const uint32_t grid = (uint32_t)(wgs < 8192 ? wgs : 8192);
    hipLaunchKernelGGL(k_mfcc_gen, dim3(grid), dim3(64 * kGenWaves), (size_t)kGenWaves * kGenWaveWords * sizeof(uint32_t), s, a);
}

}  // namespace sr
