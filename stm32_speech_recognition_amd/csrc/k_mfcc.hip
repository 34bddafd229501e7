// k_mfcc.hip -- get_mfcc (MFCC.C:86-191) incl. fft (MFCC.C:27-62) and cr4_fft_1024_stm32 (.s:95-281): the reference front end, one wave per frame.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_fft_dev.h"

#ifndef SR_MEL_CHUNKED
#define SR_MEL_CHUNKED 1  // experiment switch (profiles/experiments/RESULTS.md, round 6): 0 = bin-major scratch order of round 5
#endif

namespace sr {

// ------------------------------------------------------------------------------------------------
// k_mfcc
// ------------------------------------------------------------------------------------------------
constexpr int kMfccWaves = 4;       // waves per workgroup
// consecutive frames one wave turns into MFCCs per work item: 16 in the batch form (the lane constants and tables a workgroup
// sets up are amortised over 64 frames); 4 and 1 for launches that would leave the chip underfilled (a wave's frames are a
// serial chain of ~1.5 us each: one capture = 110 frames is 28 workgroups of 4 frames instead of 2 of 64 -- 7 us instead of
// 28-33 -- and 256 captures are 2 048 workgroups of 16 frames instead of 512 of 64)
constexpr int kFramesPerWave = 16, kFramesPerWaveMid = 4, kFramesPerWaveSmall = 1;
// per-wave LDS: exchange/scratch words + windowed frame + filterbank outputs of the wave's frames
// rows of the filterbank outputs and of the DCT tables are kMelPad = 25 words apart: in the DCT the lanes of a wave read
// 6 different frames x 12 different coefficients rows at the same column, and a stride of 24 folds those onto 4 banks
constexpr int kMelPad = kMel + 1;
// the windowed frame aliases the exchange area; the last 64 words hold the odd filters' lane offsets
// Scratch order of the filterbank stage (energies, then the in-lane prefixes): bin j = 8 l + k sits in chunk k >> 2 (256 words
// each) at 16-byte slot l ^ (4 (k >> 2)), word k & 3.  Lane l's two 16-byte reads and its prefix stores are then lane-contiguous
// chunks (conflict-free; bin-major order put two lanes of every service group on the same banks), and the slot swizzle of
// chunk 1 makes the magnitude stage's 4-byte stores -- 32 consecutive bins per service group = 4 lanes' slots x both chunks --
// hit 32 different banks as well.
__device__ __forceinline__ constexpr int mel_chunk_word(int j) { return 256 * ((j >> 2) & 1) + 4 * ((j >> 3) ^ (4 * ((j >> 2) & 1))) + (j & 3); }
constexpr int mfcc_wave_lds_words(int fpw) { return kXchgWords + fpw * kMelPad + 64; }

template <int kFPW>
__global__ void __launch_bounds__(64 * kMfccWaves, 4) k_mfcc(const MfccArgs a)
{
    constexpr int kWaveLdsWords = mfcc_wave_lds_words(kFPW);
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_dctM[kCoef * kMelPad];
    __shared__ int s_dctS[kCoef * kMelPad];  // 32-bit: read with the wide LDS loads, no byte extraction
    __shared__ u32x4 s_tw3[kTw3Row * 4], s_tw5[8 * 64];  // pass-3 (per d0) / pass-5 (per lane) coefficients, shared by the waves
    __shared__ u32x4 s_tm[4 * 64];                 // filterbank multipliers of the lane's eight bins (both poly-lines)
    // (the wave index as a SCALAR: frame counts, loop bounds and the frame's sample pointer then live on the scalar unit --
    // left in a VGPR they cost three vector instructions per frame for the next frame's load address and loop test)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *buf = smem + w * kWaveLdsWords;
    uint32_t *xw = buf;  // windowed frame, one word per sample: consumed by the pass-1 gather before the exchange overwrites it
    uint32_t *powb = buf + kXchgWords, *moff = powb + kFPW * kMelPad;

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
        for (int c = 0; c < 8; c++) s_tw3[lane * kTw3Row + c] = u32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
    }
    if (w == 1 % kMfccWaves) store_tw32(s_tw5, lane, tw.s5);
    // triangle weights of bins 8*lane .. 8*lane+7 of both poly-lines as fused multipliers ceil(tri * 2^28 / 100)
    // (sr_tables.h mel_fused_multiplier), parked in LDS as lane-contiguous 16-byte chunks like the pass-5 coefficients
    // (chunk c of lane l at s_tm[64 c + l]: even 0-3, even 4-7, odd 0-3, odd 4-7; conflict-free ds_read_b128, and LDS
    // instructions do not take VALU issue slots): held in 16 VGPRs for the whole kernel they pushed it past 128
    // (-DSR_INJECT_LDS_RACE=1 / =2, fault injection for the suite's race-class tests -- profiles/experiments/ab_build.sh race2 . -DSR_INJECT_LDS_RACE=2:
    // the fill is moved BEHIND the barrier, which is the defect round 5's soak found; level 2 also delays the late writer, which
    // opens the window on every workgroup instead of once in thousands of calls; what the tests make of both: RESULTS.md, round 6)
#ifndef SR_INJECT_LDS_RACE
    if (w == 2 % kMfccWaves) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            s_tm[64 * c + lane] = *(const u32x4 *)(a.t.tri_even_m + 8 * lane + 4 * c);
            s_tm[64 * (c + 2) + lane] = *(const u32x4 *)(a.t.tri_odd_m + 8 * lane + 4 * c);
        }
    }
#endif
    __syncthreads();  // every table above is written by ONE wave and read by all of them: nothing shared is written below this line
#ifdef SR_INJECT_LDS_RACE
    if (w == 2 % kMfccWaves) {
#if SR_INJECT_LDS_RACE >= 2  // level 2: the late writer is also held back ~3.4 us (what a slow table load does to it once in thousands of calls)
        __builtin_amdgcn_s_sleep(127);
#endif
#pragma unroll
        for (int c = 0; c < 2; c++) {
            s_tm[64 * c + lane] = *(const u32x4 *)(a.t.tri_even_m + 8 * lane + 4 * c);
            s_tm[64 * (c + 2) + lane] = *(const u32x4 *)(a.t.tri_odd_m + 8 * lane + 4 * c);
        }
    }
#endif
    int hamm_m[3];  // window weights of the lane's three samples as fused multipliers (sr_tables.h hamm_fused_multiplier)
#pragma unroll
    for (int k = 0; k < 3; k++) hamm_m[k] = (lane + 64 * k < kFrameLen) ? hamm_fused_multiplier(a.t.hamm[lane + 64 * k]) : 0;
    // filter h < 24 owned by lane h: bins [lo, hi) of poly-line (h & 1)  (MFCC.C:136-162)
    // prefix P[j] of bin j = 8 l + k sits at word 256 (k >> 2) + 4 l + (k & 3) of its poly-line's half (chunked order, see the
    // magnitude stage), the lane offsets X[l] behind them: the addresses of P[hi - 1] / P[lo - 1] are lane constants
    int f_lo = 0, p_hi = 0, p_lo = 0, x_hi = 0, x_lo = 0;
    if (lane < kMel) {
        const int h = lane;
        f_lo = (h == 0) ? 0 : (int)a.t.tri_cen[h - 1];
        const int f_hi = (h == kMel - 1) ? kBins : (int)a.t.tri_cen[h + 1];
        const int ih = f_hi - 1, il = f_lo ? f_lo - 1 : 0, half = (h & 1) ? kBins : 0;
        // X holds INCLUSIVE lane sums: the bins below lane l's first bin sum to X[l - 1]; every edge looked up lies in lane >= 1
        // (tri_cen[0] - 1 = 10), clamped so that an unused entry stays inside the array
#if SR_MEL_CHUNKED
        p_hi = half + mel_chunk_word(ih), x_hi = (ih >> 3) > 0 ? (ih >> 3) - 1 : 0;
        p_lo = half + mel_chunk_word(il), x_lo = (il >> 3) > 0 ? (il >> 3) - 1 : 0;
#else
        p_hi = half + ih, x_hi = (ih >> 3) > 0 ? (ih >> 3) - 1 : 0;
        p_lo = half + il, x_lo = (il >> 3) > 0 ? (il >> 3) - 1 : 0;
#endif
    }
    const uint32_t tw_off = 32u * (uint32_t)lane;  // byte offset of the lane's eight raw filterbank weights (LOUD / MID tiers)
    // bin lane + 64 e3 (+ 256) of the magnitude stage in the same order: word e_base + 32 e3 (+ 128)
#if SR_MEL_CHUNKED
    const int e_base = mel_chunk_word(lane);  // bins lane + 64 e3 (+ 256): 8 e3 (+ 32) slots further on, bit 2 of the slot untouched
    constexpr int kEs = 32, kEh = 128;
    const int c0 = mel_chunk_word(8 * lane), c1 = mel_chunk_word(8 * lane + 4);
#else
    const int e_base = lane;
    constexpr int kEs = 64, kEh = 256;
    const int c0 = 8 * lane, c1 = 8 * lane + 4;
#endif

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
        const uint32_t f0 = tile * (kMfccWaves * kFPW) + w * kFPW;
        uint32_t nf = 0;  // frames this wave really has
        if (f0 < nfrm) nf = (nfrm - f0 < (uint32_t)kFPW) ? nfrm - f0 : (uint32_t)kFPW;

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
                // stored as the pass-1 output A >> 2 of the s16 sample (the gather reads the low 16 bits, zero-extended); samples
                // 0..63 (k = 0) are only ever A legs of pass 2 and are stored as that pass consumes them, >> 2 once more
                if (i < kFrameLen) xw[i] = k == 0 ? window_sample<4>(s_pp[k], mid, hamm_m[k]) : window_sample<2>(s_pp[k], mid, hamm_m[k]);
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
            uint32_t nn[8];  // re^2 + im^2 of the lane's eight bins (one v_dot2_i32_i16 each, from the stored 16-bit halves)
#pragma unroll
            for (int e3 = 0; e3 < 4; e3++) {
                bfly_pk<true>(u[e3][0], u[e3][1], u[e3][2], u[e3][3], k5[e3][0][0], k5[e3][0][1], k5[e3][1][0],
                              k5[e3][1][1], k5[e3][2][0], k5[e3][2][1], k5[e3][3][0], k5[e3][3][1]);
                nn[2 * e3] = (uint32_t)sdot2z(u[e3][0], u[e3][0]);
                nn[2 * e3 + 1] = (uint32_t)sdot2z(u[e3][1], u[e3][1]);
            }
            // ---- |X|*10 and energy (MFCC.C:49-60, 128-133), three tiers by the largest re^2 + im^2 of the frame, decided for the whole
            // wave so that every branch is uniform:
            //   QUIET  (<= kMagSmallMax = 26 843, i.e. |X|*10 <= 1638 and E <= kMelFusedMaxE): cheap magnitude + fused filterbank term
            //   MID    (<= a.mag_cheap_max = 70 171 on gfx950, |X|*10 <= 2648):              cheap magnitude + literal filterbank term
            //   LOUD   (anything else):                                                     exact magnitude + literal filterbank term
            // cheap magnitude = (u32)(v_sqrt_f32 * 10): equal to the exact form for every n <= 70 171 on gfx950 -- a property of this
            // chip's v_sqrt_f32, so sr_create sweeps the whole range on the device it runs on and passes 0 if it does not hold
            // (sr_engine.cpp, sr_mag_fast_sweep); 4.5 issue slots per bin against 6.5 for the exactly corrected root (all 2^32 inputs
            // certified, sr_dev.h sqrt_rn_int).  At the benchmark's amplitudes 97-99 % of the frames are QUIET; at SURVEY 8(d)'s
            // (gain 2.4) 29 % QUIET / 47 % MID / 24 % LOUD; near-clipping captures are LOUD (profiles/experiments/RESULTS.md).
            const uint32_t nmax = max(max(max(max(nn[0], nn[1]), nn[2]), max(max(nn[3], nn[4]), nn[5])), max(nn[6], nn[7]));
            const bool quiet = __builtin_expect(__builtin_amdgcn_ballot_w64(nmax > kMagSmallMax) == 0, 1);
            // the literal filterbank form multiplies by the weights themselves: requested from the (cache-resident) table now, they
            // arrive behind the magnitude stage.  Kept out of LDS (the workgroup's share is full at four workgroups per CU) and out
            // of the quiet path's registers.
            u32x4 tw_e[2], tw_o[2];
            bool cheap = true;
            if (!quiet) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    tw_e[c] = *(const u32x4 *)((const char *)a.t.tri_even32 + tw_off + 16 * c);
                    tw_o[c] = *(const u32x4 *)((const char *)a.t.tri_odd32 + tw_off + 16 * c);
                }
                cheap = __builtin_amdgcn_ballot_w64(nmax > a.mag_cheap_max) == 0;
            } else {
                // never read on this path: "defined" by an empty asm so that the quiet frame does not pay 16 register clears
                asm volatile("" : "=v"(tw_e[0]), "=v"(tw_e[1]), "=v"(tw_o[0]), "=v"(tw_o[1]));
            }
            // energies go to the wave's scratch in the chunked order the filterbank reads them in (mel_chunk_word)
            uint32_t *eb = buf + e_base;
            if (cheap) {
#pragma unroll
                for (int e3 = 0; e3 < 4; e3++) {
                    const f32x2 m = f32x2{__builtin_amdgcn_sqrtf((float)(int)nn[2 * e3]), __builtin_amdgcn_sqrtf((float)(int)nn[2 * e3 + 1])} *
                                    f32x2{10.0f, 10.0f};
                    const uint32_t m0 = cvt_u32(m.x), m1 = cvt_u32(m.y);
                    eb[kEs * e3] = umul24(m0, m0);
                    eb[kEs * e3 + kEh] = umul24(m1, m1);
                }
            } else {
#pragma unroll
                for (int e3 = 0; e3 < 4; e3++) {
                    // both bins' roots and the x10 in packed f32 operations (plain IEEE multiplies and fused multiply-adds, see sqrt_rn_int)
                    const f32x2 m = sqrt_rn_int2(f32x2{(float)(int)nn[2 * e3], (float)(int)nn[2 * e3 + 1]}) * f32x2{10.0f, 10.0f};
                    const uint32_t m0 = cvt_u32(m.x), m1 = cvt_u32(m.y);  // < 2^19
                    eb[kEs * e3] = umul24(m0, m0);
                    eb[kEs * e3 + kEh] = umul24(m1, m1);
                }
            }
            wave_sync();
            // ---- Mel filterbank as prefix sums over bins (each term /100 before summing, u32 wrap)
            uint32_t pe[8], po[8], xe, xo;
            {
                const uint4 q0 = *(const uint4 *)(buf + c0), q1 = *(const uint4 *)(buf + c1);
                const uint32_t e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                uint32_t se = 0, so = 0;
                // E*tri/100 (MFCC.C:139-161): in a quiet frame (every E <= kMelFusedMaxE, see above) a term is ONE v_mul_hi_u32 of
                // E << 4 with the per-bin multiplier (the shift shared by both poly-lines): 5 instructions per bin instead of 8
                // (mul_lo, mul_hi, shift, add per term).  A louder frame takes the literal u32-wrapping form.
                if (quiet) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const u32x4 me = s_tm[64 * c + lane], mo = s_tm[64 * (c + 2) + lane];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t es = e[4 * c + k] << 4;
                            se += mel_term_fused(es, me[k]);
                            so += mel_term_fused(es, mo[k]);
                            pe[4 * c + k] = se;
                            po[4 * c + k] = so;
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            se += e[4 * c + k] * tw_e[c][k] / 100u;
                            so += e[4 * c + k] * tw_o[c][k] / 100u;
                            pe[4 * c + k] = se;
                            po[4 * c + k] = so;
                        }
                    }
                }
                xe = wave_scan_incl(se);  // sum over the bins of the lanes up to and including this one: a filter lane reads the
                xo = wave_scan_incl(so);  // entry of the lane BELOW its bin's (never lane 0: the first filter edge is bin 10)
            }
            // (no ordering point is needed here: a lane overwrites only the eight energies it has read itself, every other
            // store below goes to words nobody reads before the next wave_sync)
            // in-lane prefixes and the per-lane offsets are stored separately: the 24 filter lanes add them on lookup
            // (2 adds) instead of every lane adding its offset to 16 prefixes
            *(uint4 *)(buf + c0) = make_uint4(pe[0], pe[1], pe[2], pe[3]);
            *(uint4 *)(buf + c1) = make_uint4(pe[4], pe[5], pe[6], pe[7]);
            *(uint4 *)(buf + kBins + c0) = make_uint4(po[0], po[1], po[2], po[3]);
            *(uint4 *)(buf + kBins + c1) = make_uint4(po[4], po[5], po[6], po[7]);
            buf[2 * kBins + lane] = xe;
            moff[lane] = xo;
            wave_sync();
            if (lane < kMel) {
                const uint32_t *X = (lane & 1) ? moff : buf + 2 * kBins;
                const uint32_t hi = buf[p_hi] + X[x_hi], lo = f_lo ? buf[p_lo] + X[x_lo] : 0u;
                powb[fi * kMelPad + lane] = hi - lo;
            }
            wave_sync();
        }

        // ---- log (MFCC.C:165-170) and DCT (MFCC.C:173-183) for the wave's nf frames, all lanes busy
        // (the pad word of each row goes through the log as well: harmless, never read)
        // (round 6: all rounds' estimates first, then all threshold pairs in flight at once, then the corrections -- as a loop
        // each round waited for its own pair of thresholds from the table in L2: seven serial round trips per 16 frames)
        {
            constexpr int kLogRounds = (kFPW * kMelPad + 63) / 64;
            const uint32_t n_log = nf * kMelPad;
            uint32_t nv[kLogRounds], mv[kLogRounds];
            u32x2 tv[kLogRounds];
#pragma unroll
            for (int r = 0; r < kLogRounds; r++) {
                const uint32_t t = lane + 64u * r;
                nv[r] = powb[t];  // (rounds past n_log read the wave's own scratch behind the filterbank outputs: in bounds, unused)
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
        // output t = fi*12 + h of the wave's tile goes to out[(f0 + fi)*12 + h] = out_w[t]: consecutive lanes store
        // consecutive s16; fi = t / 12 by a 24-bit multiply (exact for t < 2^13), all index arithmetic in 32 bits
        // (left to the compiler the 64-bit subscript became eight v_mad_u64_u32 per round)
        {
            int16_t *out_w = out + (size_t)f0 * kCoef;
#pragma unroll
            for (uint32_t t = lane; t < (uint32_t)(kFPW * kCoef); t += 64) {
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
            const uint32_t r0 = f0 + nf, r1 = (f0 + kFPW < a.max_frames) ? f0 + kFPW : a.max_frames;
            for (uint32_t t = r0 * kCoef + lane; t < r1 * kCoef && r0 < r1; t += 64) out[t] = 0;
        }
    }
}

// the 16 kHz / 512-point EXTENSION front end lives in k_mfcc_ext.hip
uint32_t mfcc_ext_frames_per_tile();
int mfcc_ext_occupancy(int *per_cu);
void launch_mfcc_ext(const MfccArgs &a, uint32_t grid, hipStream_t s);

uint32_t mfcc_frames_per_tile(uint32_t frame_len) { return frame_len == 320 ? mfcc_ext_frames_per_tile() : (uint32_t)(kMfccWaves * kFramesPerWave); }
// frames per work item of the underfilled-launch forms, 0 = mid, 1 = small (the extension kernel has one form)
uint32_t mfcc_frames_per_tile_small(uint32_t frame_len, uint32_t which)
{
    return frame_len == 320 ? mfcc_ext_frames_per_tile() : (uint32_t)(kMfccWaves * (which ? kFramesPerWaveSmall : kFramesPerWaveMid));
}

// workgroups of the frame kernel that fit on the current device at once (occupancy query x CU count)
uint32_t mfcc_resident_workgroups(uint32_t frame_len)
{
    int dev = 0, n_cu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) return 0;
    int e;
    if (frame_len == 320)
        e = mfcc_ext_occupancy(&per_cu);
    else
        e = (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mfcc<kFramesPerWave>, 64 * kMfccWaves,
                                                              (size_t)kMfccWaves * mfcc_wave_lds_words(kFramesPerWave) * sizeof(uint32_t));
    if (e != (int)hipSuccess || per_cu < 1) return 0;
    return (uint32_t)(per_cu * n_cu);
}

void launch_mfcc(const MfccArgs &a, hipStream_t s)
{
    if (a.generic) {
        launch_mfcc_gen(a, s);
        return;
    }
    if (a.n_items == 0) return;
    // persistent-style grid: a few times the workgroups that are resident at once (see sr_create), work items strided
    const uint32_t cap = a.grid_cap ? a.grid_cap : 4096u;
    const uint32_t grid = a.n_items < cap ? a.n_items : cap;
    if (a.frame_len == 320) {
        launch_mfcc_ext(a, grid, s);
        return;
    }
    if (a.small_tiles == 2) {  // a.tiles counts tiles of mfcc_frames_per_tile_small(frame_len, 1) frames
        const size_t lds = (size_t)kMfccWaves * mfcc_wave_lds_words(kFramesPerWaveSmall) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_mfcc<kFramesPerWaveSmall>, dim3(grid), dim3(64 * kMfccWaves), lds, s, a);
        return;
    }
    if (a.small_tiles == 1) {  // ... of mfcc_frames_per_tile_small(frame_len, 0) frames
        const size_t lds = (size_t)kMfccWaves * mfcc_wave_lds_words(kFramesPerWaveMid) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_mfcc<kFramesPerWaveMid>, dim3(grid), dim3(64 * kMfccWaves), lds, s, a);
        return;
    }
    const size_t lds = (size_t)kMfccWaves * mfcc_wave_lds_words(kFramesPerWave) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_mfcc<kFramesPerWave>, dim3(grid), dim3(64 * kMfccWaves), lds, s, a);
}

}  // namespace sr
