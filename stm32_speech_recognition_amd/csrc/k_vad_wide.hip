// k_vad_wide.hip -- noise_atap (VAD.C:22-71) + VAD (VAD.C:97-218) + frame count of get_mfcc (MFCC.C:102-107) for SMALL launches:
// one WORKGROUP of four waves per capture buffer (k_vad: one wave per capture).
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; integer VALU + the scalar unit.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
//
// k_vad's wave scans a capture in rounds of 63 frames, ~1 000 dependent-ish instructions per lane and round (80 samples each):
// 22 us for the firmware's 16 000-sample capture when nothing else runs on the SIMD.  With a handful of captures (spch_recg:
// one) the chip is idle, so here the rounds of a capture are spread over four waves: the per-block summaries (magnitude sum,
// band crossings, first / last out-of-band class -- everything that does not depend on earlier blocks) are computed by all
// four waves at once, and only the short part that carries state across rounds -- the class of the last out-of-band sample
// (last_sig is never reset, VAD.C:99) and the endpoint state machine (VAD.C:164-216) -- goes round by round, one wave after
// the other, through LDS.  noise_atap's sums and block maxima are split over the waves the same way (integer sums: any order).
// Same arithmetic per block, per frame and per run as k_vad (sr_vad_dev.h), whose comments explain the block algebra.
#include "sr_vad_dev.h"

namespace sr {
namespace wide {
constexpr int kWaves = 4;
struct RunState {  // what k_vad carries across rounds in registers
    uint32_t carry, cur, front, back, vcon, done;
    int seg0_start, seg0_end;
};
}  // namespace wide

template <int kFrameLen, int kHop, bool kSad>
__global__ void __launch_bounds__(64 * wide::kWaves) k_vad_wide(const VadArgs a)
{
    using namespace wide;
    __shared__ uint32_t s_part[kWaves][2];
    __shared__ RunState s_run;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x;
    const uint4 *row = (const uint4 *)(a.pcm + (uint64_t)b * a.pcm_stride);
    const uint32_t S = a.buf_len;
    sr_vad_rec *rec_out = a.vad + b;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 2 * SR_MAX_SEG; i++) rec_out->seg[i] = -1;
        s_run = RunState{0, 0, 0, 0, 0, 0, -1, -1};  // VAD.C:100-102,109
    }

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
        for (uint32_t v = threadIdx.x; v < nvec; v += 64 * kWaves) {
            const uint4 q = row[v];
            part += (q.x & 0xFFFF) + (q.x >> 16) + (q.y & 0xFFFF) + (q.y >> 16) + (q.z & 0xFFFF) + (q.z >> 16) +
                    (q.w & 0xFFFF) + (q.w >> 16);
        }
        part = wave_sum(part);
        if (lane == 0) s_part[w][0] = part;
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < kWaves; i++) total += s_part[i][0];
        mid = total / a.noise_len;  // VAD.C:41-45
        const uint32_t nblk = a.noise_len / a.atap_frm, vpb = a.atap_frm / 8;
        uint32_t max_part = 0, abs_part = 0;
        for (uint32_t blk = w; blk < nblk; blk += kWaves) {  // VAD.C:48-63
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
            max_part += wave_max(nmax);
        }
        abs_part = wave_sum(abs_part);
        __syncthreads();  // everyone has read the first pass's partial sums
        if (lane == 0) {
            s_part[w][0] = max_part;
            s_part[w][1] = abs_part;
        }
        __syncthreads();
        uint32_t max_sum = 0, abs_sum = 0;
#pragma unroll
        for (int i = 0; i < kWaves; i++) {
            max_sum += s_part[i][0];
            abs_sum += s_part[i][1];
        }
        abs_sum /= (a.noise_len / (uint32_t)kFrameLen);  // VAD.C:65 (divides by n_len/frame_len)
        max_sum /= nblk;                                 // VAD.C:66
        n_thl = max_sum & 0xFFFF;                        // u16 field, n_thl_ratio = 1
        s_thl = abs_sum * 11 / 10;                       // s_thl_ratio
        z_thl = (uint32_t)kFrameLen * 2 / 160 / 1;       // VAD.C:70
    }
    const uint32_t a_thl = mid + n_thl, b_thl = mid - n_thl;  // VAD.C:112-113 (u32, may wrap)
    const uint32_t mid2 = (mid & 0xFFFFu) * 0x10001u;          // mid in both halves (kSad)
    const uint32_t F = (S > (uint32_t)kFrameLen) ? (S - kFrameLen + kHop - 1) / kHop : 0;  // frames, VAD.C:121
    const uint32_t v_durmin = a.v_durmin, s_durmax = a.s_durmax;
    __syncthreads();  // s_run is set

    for (uint32_t jb0 = 0; jb0 < F; jb0 += 63 * kWaves) {
        // workgroup-uniform exit: every wave reads the flag the previous pass left, and only after ALL of them have read it
        // may wave 0 rewrite s_run in round 0 of this pass (without the barrier a late wave could see this pass's "done" and
        // leave while the others still wait at the round loop's barriers)
        const bool was_done = s_run.done;
        __syncthreads();
        if (was_done) break;
        const uint32_t jb = jb0 + 63 * (uint32_t)w;
        const bool mine = jb < F;  // this wave has a round in this pass
        // ---- per-block summaries of the wave's round: no state from earlier blocks (see k_vad)
        const uint32_t j = jb + lane;  // block index; frame f = j uses blocks j and j+1
        uint32_t A = 0, internal = 0, last = 0, cf = 0, c78 = 0;
        int pfo = -1;
        if (mine && j <= F) vad_block_summary<kHop, kSad>(row + (uint64_t)j * (kHop / 8), mid, mid2, a_thl, b_thl, A, internal, last, cf, c78, pfo);
        const uint32_t c80 = last;
        // the part without the carry: class of the last out-of-band sample in the wave's blocks <= j
        uint32_t R = c80;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(R, d, 64);
            if (lane >= d) R = R ? R : o;
        }
        const uint32_t A_next = __shfl_down(A, 1, 64);
        const uint32_t frm_sum = A + A_next;

        // ---- the rounds in order: carry, band-crossing counts, loudness, endpoint state machine
        for (int r = 0; r < kWaves; r++) {
            if (w == r && mine && !s_run.done) {
                const uint32_t carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_run.carry);
                uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_run.cur);
                uint32_t front = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_run.front);
                uint32_t back = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_run.back);
                uint32_t vcon = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_run.vcon);
                int seg0_start = __builtin_amdgcn_readfirstlane(s_run.seg0_start);
                int seg0_end = __builtin_amdgcn_readfirstlane(s_run.seg0_end);
                bool done = false;
                // R(j) = class of the last out-of-band sample in blocks <= j
                const uint32_t Rc = R ? R : carry;
                uint32_t Rprev = __shfl_up(Rc, 1, 64);
                if (lane == 0) Rprev = carry;
                const uint32_t carry_out = __shfl(Rc, 62, 64);
                const uint32_t ff = (cf != 0 && Rprev != 0 && Rprev != cf) ? 1u : 0u;  // flip at the block's first out-of-band sample
                const uint32_t fl = internal + ff;
                const uint32_t fl_next = __shfl_down(fl, 1, 64);
                uint32_t Z = internal + fl_next;
                if (pfo < 0 || pfo == kHop - 1)
                    Z += ff;  // entry state = history before the frame: natural count
                else if (pfo > 0 && j > 0)
                    Z += (c78 != cf) ? 1u : 0u;  // entry state comes from inside the frame (positions <= 78);
                                                 // frame 0 starts with last_sig = 0 (VAD.C:99)
                const bool loud = (lane < 63) && (j < F) && (frm_sum > s_thl || Z > z_thl);  // VAD.C:164
                const uint64_t mask = __ballot(loud);
                if (a.dbg_masks && lane == 0) a.dbg_masks[(uint64_t)b * 16 + (jb / 63 < 16 ? jb / 63 : 15)] = mask;
                const uint32_t nfr = (F - jb < 63u) ? F - jb : 63u;

                // endpoint state machine (VAD.C:164-216), one RUN of equal frames at a time: see k_vad
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
                        const uint32_t need = (s_durmax > back) ? s_durmax - back : 1u;
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
                if (lane == 0) s_run = RunState{carry_out, cur, front, back, vcon, done ? 1u : 0u, seg0_start, seg0_end};
            }
            __syncthreads();
        }
    }

    if (threadIdx.x == 0) {
        const int seg0_start = s_run.seg0_start, seg0_end = s_run.seg0_end;
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

template <int FL>
static void launch_wide(const VadArgs &a, hipStream_t s)
{
    const dim3 grid(a.B), block(64 * wide::kWaves);
    if (a.atap_in == nullptr) hipLaunchKernelGGL((k_vad_wide<FL, FL / 2, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_vad_wide<FL, FL / 2, false>), grid, block, 0, s, a);
}

// every framing vad_framing_supported accepts
void launch_vad_wide(const VadArgs &a, hipStream_t s)
{
    if (!a.B) return;
    switch (a.frame_len) {
    case 160: launch_wide<160>(a, s); break;
    case 240: launch_wide<240>(a, s); break;
    case 256: launch_wide<256>(a, s); break;
    case 320: launch_wide<320>(a, s); break;
    case 400: launch_wide<400>(a, s); break;
    case 512: launch_wide<512>(a, s); break;
    default: break;
    }
}

}  // namespace sr
