// k_mfcc_ext.hip -- EXTENSION front end (BASELINE.json configs[4]; no reference counterpart).
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_fft_dev.h"

namespace sr {

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
    // window weights of the lane's 20 samples as fused multipliers (sr_tables.h hamm_fused_multiplier), layout A: chunk c of
    // gl at [c*16 + gl] = (even, odd) sample of pair 2c, (even, odd) sample of pair 2c + 1
    __shared__ u32x4 s_hm[5 * 16];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar wave index, see k_mfcc
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
    if (w == 3 % kWaves && lane < 16) {  // Hamming weights of the lane's sample pairs (2*base + 32 t, +1)
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const uint32_t h0 = a.t.hamm_pk[base + 16 * (2 * c)], h1 = a.t.hamm_pk[base + 16 * (2 * c + 1)];
            s_hm[c * 16 + gl] = u32x4{(uint32_t)hamm_fused_multiplier(h0 & 0xFFFFu), (uint32_t)hamm_fused_multiplier(h0 >> 16),
                                      (uint32_t)hamm_fused_multiplier(h1 & 0xFFFFu), (uint32_t)hamm_fused_multiplier(h1 >> 16)};
        }
    }
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
#ifndef SR_INJECT_LDS_RACE  // (fault injection for the suite's race-class tests, see k_mfcc.hip: the fill moves behind the barrier)
    if (w == 2 % kWaves && lane < 16) {  // triangle weights of the bins 16*gl .. 16*gl + 15 (filterbank layout)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int b0 = 16 * gl + 2 * c;
            s_tri[c * 16 + gl] = u32x4{a.t.tri_even[b0], a.t.tri_odd[b0], a.t.tri_even[b0 + 1], a.t.tri_odd[b0 + 1]};
        }
    }
#endif
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
#ifdef SR_INJECT_LDS_RACE
    if (w == 2 % kWaves && lane < 16) {
#if SR_INJECT_LDS_RACE >= 2  // level 2: the late writer is also held back ~3.4 us (what a slow table load does to it once in thousands of calls)
        __builtin_amdgcn_s_sleep(127);
#endif
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int b0 = 16 * gl + 2 * c;
            s_tri[c * 16 + gl] = u32x4{a.t.tri_even[b0], a.t.tri_odd[b0], a.t.tri_even[b0 + 1], a.t.tri_odd[b0 + 1]};
        }
    }
#endif

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
        // A segment that starts at sample 1 (only sr_mfcc_batch / sr_recognize_segments callers can produce one: VAD
        // segments start on frame boundaries) puts the first pair of frame 0 at byte offset -2: out of range for the
        // resource, so the load (at least its first dword: x[i-2] AND x[i-1]) came back as 0 -- but x[i-1] = sample 0 is
        // the pre-emphasis predecessor of the segment's first sample (MFCC.C:119).  Wave-uniform and rare: that lane
        // fetches its first pair again with plain loads (x[i-2] does not exist and is never used).
        if (it.s0 < 2) {
            if (off < 0) {
                qa[0] = (uint32_t)it.row[0] << 16;
                qb[0] = (uint32_t)it.row[1] | ((uint32_t)it.row[2] << 16);
            }
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
            for (int c = 0; c < 5; c++) {
                const u32x4 hm = s_hm[c * 16 + gl];
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    const int t = 2 * c + h2;
                    const int p0 = (int)(pa[t] >> 16) - mid, c0 = (int)(pb[t] & 0xFFFFu) - mid;  // x[i-1], x[i]
                    const int c1 = (int)(pb[t] >> 16) - mid, p1 = c0;                              // x[i+1], x[i]
                    // (s16)(temp*hamm/1000): the division folded into the weight (sr_dev.h window_quotient)
                    ws[0][t] = (uint32_t)window_quotient(c0, neg_preemph95(p0), (int)(h2 ? hm.z : hm.x)) & 0xFFFFu;
                    ws[1][t] = (uint32_t)window_quotient(c1, neg_preemph95(p1), (int)(h2 ? hm.w : hm.y)) & 0xFFFFu;
                }
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
        if (nf) {  // (round 6, as in k_mfcc: estimates of all rounds, all threshold pairs in flight at once, then the corrections)
            constexpr int kLogRounds = (kFpw * kMelEPad + 63) / 64;
            const uint32_t n_log = nf * kMelEPad;
            uint32_t nv[kLogRounds], mv[kLogRounds];
            u32x2 tv[kLogRounds];
#pragma unroll
            for (int r = 0; r < kLogRounds; r++) {
                const uint32_t t = lane + 64u * r;
                nv[r] = powb[t < n_log ? t : n_log - 1];  // (the filterbank outputs end the wave's scratch: stay inside it)
                mv[r] = log100_est(nv[r]);
                tv[r] = *(const u32_pair_align4 *)((const char *)a.t.log_thr + 4u * mv[r]);
            }
#pragma unroll
            for (int r = 0; r < kLogRounds; r++) {
                const uint32_t t = lane + 64u * r;
                if (t < n_log) powb[t] = log100_fix(nv[r], mv[r], tv[r].x, tv[r].y) << 14;
            }
        }
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

uint32_t mfcc_ext_frames_per_tile() { return (uint32_t)ext::kTile; }
int mfcc_ext_occupancy(int *per_cu)
{
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_mfcc_ext, 64 * ext::kWaves,
                                                             (size_t)ext::kWaves * ext::kWaveWords * sizeof(uint32_t));
}
void launch_mfcc_ext(const MfccArgs &a, uint32_t grid, hipStream_t s)
{
    const size_t lds = (size_t)ext::kWaves * ext::kWaveWords * sizeof(uint32_t);
    hipLaunchKernelGGL(k_mfcc_ext, dim3(grid), dim3(64 * ext::kWaves), lds, s, a);
}

}  // namespace sr
