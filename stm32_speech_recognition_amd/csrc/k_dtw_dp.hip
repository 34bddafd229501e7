// k_dtw_dp.hip -- OPT-IN, NON-REFERENCE full dynamic-programming DTW scorer.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_dtw_dev.h"

namespace sr {

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

}  // namespace sr
