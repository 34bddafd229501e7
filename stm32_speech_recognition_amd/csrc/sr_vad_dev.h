// sr_vad_dev.h -- the VAD kernel template: noise_atap (VAD.C:22-71) + VAD (VAD.C:97-218) + frame count of get_mfcc
// (MFCC.C:102-107), one wave per capture buffer.  Instantiated for the reference / extension framings in k_vad.hip and
// for the other accepted framings (frame_len = 2 * hop, hop a multiple of 8) in k_vad_gen.hip.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; integer VALU + the scalar unit.
#pragma once
#include "sr_dev.h"

namespace sr {

// ------------------------------------------------------------------------------------------------
// k_vad: one wave per capture buffer
// ------------------------------------------------------------------------------------------------
constexpr int kVadWaves = 4;

__device__ __forceinline__ uint32_t absdiff(uint32_t v, uint32_t mid) { return v > mid ? v - mid : mid - v; }


// ---- summary of one block of kHop samples (VAD.C:121-157), with no state from earlier blocks --------------------------
//   A        sum |x - mid|                                        (short-time magnitude, half a frame)
//   internal band crossings among the block's own out-of-band samples (class changes 2 <-> 1, in-band samples skipped)
//   last     class of the block's last out-of-band sample (0: none); c78: the same without the block's last sample
//   cf, pfo  class and offset of its first out-of-band sample (0, -1: none)
// Class of a sample: 2 above the band (x >= a_thl), 1 below (x < b_thl), 0 inside; "above" wins when both hold (u32
// thresholds that wrapped, VAD.C:112-113).
// Round 5: BIT-PARALLEL.  Per sample only two compares, each shifted into a mask with its carry (m = m + m + cc: v_cmp +
// v_addc_co_u32) -- samples are taken from the block's end to its start, so that sample i lands on bit i.  Everything else
// is arithmetic on the two masks (kHop bits each):
//   * crossings 2 -> 1: a "below" sample whose nearest earlier out-of-band sample is "above".  That is exactly the carry
//     of an adder with generate = above, propagate = in-band (kill = below): carries = ((G | P) + G) ^ (G | P) ^ G; the count
//     is popcount(below & carries).  Likewise 1 -> 2 with the roles swapped.
//   * first / last out-of-band sample: lowest / highest set bit of the two masks, compared with each other for the class.
// (Round 4 walked the samples one by one with five carried values: 11.6 vector instructions per sample.)
__device__ __forceinline__ uint32_t mask_lowest(const uint32_t *m, int nw)  // position of the lowest set bit, 0xFFFFFFFF: none
{
    uint32_t p = 0xFFFFFFFFu;
#pragma unroll
    for (int w = nw - 1; w >= 0; w--) p = m[w] ? (uint32_t)(32 * w + __builtin_ctz(m[w])) : p;
    return p;
}
__device__ __forceinline__ uint32_t mask_highest1(const uint32_t *m, int nw)  // position of the highest set bit + 1, 0: none
{
    uint32_t p = 0;
#pragma unroll
    for (int w = 0; w < nw; w++) p = m[w] ? (uint32_t)(32 * w + 32 - __builtin_clz(m[w])) : p;
    return p;
}
// one sample into the two masks: m = m + m + (compare), the compare on one 16-bit half of the word that holds two samples
// (SDWA operand select), its result handed to the add as the carry: four instructions per sample.  (Left to the compiler:
// v_cmp + v_cndmask per sample and a shift + v_or3 per pair, six per sample.)
template <int kHalf>
__device__ __forceinline__ void vad_sample_bits(uint32_t word, uint32_t a_thl, uint32_t b_thl, uint32_t &ma, uint32_t &mb)
{
    if (kHalf)
        asm("v_cmp_ge_u32_sdwa vcc, %2, %3 src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_lt_u32_sdwa vcc, %2, %4 src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
            : "+v"(ma), "+v"(mb)
            : "v"(word), "v"(a_thl), "v"(b_thl)
            : "vcc");
    else
        asm("v_cmp_ge_u32_sdwa vcc, %2, %3 src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
            "v_cmp_lt_u32_sdwa vcc, %2, %4 src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
            : "+v"(ma), "+v"(mb)
            : "v"(word), "v"(a_thl), "v"(b_thl)
            : "vcc");
}
template <int kHop, bool kSad>
__device__ __forceinline__ void vad_block_summary(const uint4 *blk, uint32_t mid, uint32_t mid2, uint32_t a_thl, uint32_t b_thl,
                                                  uint32_t &A_out, uint32_t &internal, uint32_t &last, uint32_t &cf, uint32_t &c78,
                                                  int &pfo)
{
    static_assert(kHop % 8 == 0, "blocks are read as 16-byte vectors");
    constexpr int NW = (kHop + 31) / 32, NV = kHop / 8;
    uint32_t wd[NV * 4];  // two samples per word
    uint32_t A = 0;
#pragma unroll
    for (int t = 0; t < NV; t++) {
        const uint4 q = blk[t];
        wd[4 * t] = q.x, wd[4 * t + 1] = q.y, wd[4 * t + 2] = q.z, wd[4 * t + 3] = q.w;
        if (kSad) {  // v_sad_u16: |a.lo-b.lo| + |a.hi-b.hi| + c
#pragma unroll
            for (int i = 0; i < 4; i++) A = __builtin_amdgcn_sad_u16(wd[4 * t + i], mid2, A);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) A += absdiff(wd[4 * t + i] & 0xFFFF, mid) + absdiff(wd[4 * t + i] >> 16, mid);
        }
    }
    uint32_t ma[NW], mb[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int s = 31; s >= 0; s--) {
            const int i = 32 * w + s;
            if (i < kHop) {
                if (i & 1) vad_sample_bits<1>(wd[i >> 1], a_thl, b_thl, a, b);
                else vad_sample_bits<0>(wd[i >> 1], a_thl, b_thl, a, b);
            }
        }
        ma[w] = a;
        mb[w] = b & ~a;  // "above" wins
    }
    // band crossings inside the block: carries of the two (generate, propagate = in-band) additions, see above
    uint32_t cnt = 0, ca = 0, cb = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t p = ~(ma[w] | mb[w]);
        const uint32_t xa = ma[w] | p, xb = mb[w] | p;
        const uint64_t sa = (uint64_t)xa + ma[w] + ca, sb = (uint64_t)xb + mb[w] + cb;
        ca = (uint32_t)(sa >> 32);
        cb = (uint32_t)(sb >> 32);
        cnt += (uint32_t)__builtin_popcount(mb[w] & ((uint32_t)sa ^ xa ^ ma[w]));  // 2 -> 1
        cnt += (uint32_t)__builtin_popcount(ma[w] & ((uint32_t)sb ^ xb ^ mb[w]));  // 1 -> 2
    }
    const uint32_t fa = mask_lowest(ma, NW), fb = mask_lowest(mb, NW);
    pfo = (int)(fa < fb ? fa : fb);                       // -1: none
    cf = fa < fb ? 2u : (fb < fa ? 1u : 0u);              // equal only when both are "none"
    const uint32_t la = mask_highest1(ma, NW), lb = mask_highest1(mb, NW);
    last = la > lb ? 2u : (lb > la ? 1u : 0u);
    // the same without the block's last sample (VAD.C:131-157 evaluate the frame's first sample against the state BEFORE it)
    ma[NW - 1] &= ~(1u << ((kHop - 1) & 31));
    mb[NW - 1] &= ~(1u << ((kHop - 1) & 31));
    const uint32_t la8 = mask_highest1(ma, NW), lb8 = mask_highest1(mb, NW);
    c78 = la8 > lb8 ? 2u : (lb8 > la8 ? 1u : 0u);
    A_out = A;
    internal = cnt;
}

// kSad: |x - mid| sums of two samples per instruction (v_sad_u16).  Only valid for 16-bit mid values, which is what
// noise_atap produces (a mean of u16 samples, VAD.C:41-47); the variant without it serves callers that hand their own
// thresholds in (atap_in), which may hold anything.
template <int kFrameLen, int kHop, bool kSad>  // 160/80 = the reference (VAD.H:5-8); 320/160 = the 16 kHz extension
__global__ void __launch_bounds__(64 * kVadWaves) __attribute__((amdgpu_waves_per_eu(1, kHop > 80 ? 4 : 8))) k_vad(const VadArgs a)
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
    // VAD.C:72-75: 80 ms / 110 ms in frames of (frame_time - frame_mov_t) ms: 8 and 11 at the reference's 20 / 10 ms framing
    const uint32_t v_durmin = a.v_durmin, s_durmax = a.s_durmax;

    for (uint32_t jb = 0; jb < F && !done; jb += 63) {
        const uint32_t j = jb + lane;  // block index; frame f = j uses blocks j and j+1
        uint32_t A = 0, internal = 0, last = 0, cf = 0, c78 = 0;
        int pfo = -1;
        if (j <= F) vad_block_summary<kHop, kSad>(row + (uint64_t)j * (kHop / 8), mid, mid2, a_thl, b_thl, A, internal, last, cf, c78, pfo);
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
                // the count is checked on a LOUD frame met in this state, after front++ (VAD.C:173-181): at least one more
                // loud frame is needed even when v_durmin is 1 (hops above 40 ms)
                const uint32_t need = (v_durmin > front) ? v_durmin - front : 1u;
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
                const uint32_t need = (s_durmax > back) ? s_durmax - back : 1u;  // likewise (VAD.C:196-207)
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

}  // namespace sr
