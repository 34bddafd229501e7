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
    // VAD.C:72-75: 80 ms / 110 ms in frames of (frame_time - frame_mov_t) ms: 8 and 11 at the reference's 20 / 10 ms framing
    const uint32_t v_durmin = a.v_durmin, s_durmax = a.s_durmax;

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
