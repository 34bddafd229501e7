// k_dtw_cells.hip -- dtw (DTW.C:120-192) for SMALL launches: one capture against the store (spch_recg, main.c:276-295), one dtw() call.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
//
// The greedy walk is a serial chain of up to in + mdl steps.  In k_dtw_lds a step costs ~1 400 shader cycles when the wave
// has its SIMD to itself (three distances, a root, the tie order, the bound tests, two row fetches, all dependent): with a
// handful of pairs the GPU idles while 80 lanes crawl, 140 us for one 110-frame capture against 80 slots.  Here the chip's
// idle width is spent instead: ONE WORKGROUP PER PAIR first evaluates, for EVERY point (x, y) of the rectangle at once, what
// the walk would do if it stood there -- the three candidates of DTW.C:152-154 (dtw_limit + get_dis), their minimum
// (DTW.C:156-164), the move (DTW.C:168-184) and whether the loop ends after it (DTW.C:188) -- packed into one 32-bit word per
// point in LDS; the walk itself is then one lane chasing those words: one LDS read, two adds and a shift per step.
// Same arithmetic as k_dtw_gen (any feature width up to 16 coefficients, plain template store), so the scores are identical.
#include "sr_dtw_cells.h"
#include "sr_dtw_dev.h"

namespace sr {
namespace cells {
constexpr uint32_t kThreads = 1024;
constexpr uint32_t kRowWords = 9;  // 8 packed coefficient pairs + the squared norm; odd stride: lanes on consecutive rows, distinct banks
// word of a point: bits 0-15 the step cost (root of the smallest admissible candidate), bit 16 "all three outside" (the cost is
// dis_err = 2^32 - 1, DTW.C:152-164), bit 17 the walk ends after this step, bits 18-31 the distance to the next point in words
constexpr uint32_t kErrBit = 1u << 16, kStopBit = 1u << 17;

struct Row16 {
    uint32_t w[8];
    uint32_t n;
};
// a feature row of nc <= 16 s16 (2-byte aligned when nc is odd) as packed pairs, zero-padded: the pad adds nothing to
// get_dis' sum of squares (DTW.C:45-62)
__device__ __forceinline__ void stage_row(uint32_t *dst, const int16_t *p, uint32_t nc)
{
    int acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t lo = (2 * i < nc) ? (uint32_t)(uint16_t)p[2 * i] : 0u, hi = (2 * i + 1 < nc) ? (uint32_t)(uint16_t)p[2 * i + 1] : 0u;
        const uint32_t w = lo | (hi << 16);
        dst[i] = w;
        acc = sdot2(w, w, acc);
    }
    dst[8] = (uint32_t)acc;
}
__device__ __forceinline__ Row16 lds_row(const uint32_t *p)
{
    Row16 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = p[i];
    r.n = p[8];
    return r;
}
// get_dis (DTW.C:45-62): sum (a-b)^2 in u32 wrap = |a|^2 + |b|^2 - 2 a.b in the same ring, then (u32)sqrtf
__device__ __forceinline__ uint32_t dis_rows(const Row16 &a, const Row16 &b)
{
    int dot = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) dot = sdot2(a.w[i], b.w[i], dot);
    return cvt_u32(sqrt_rn_int((float)(a.n + b.n - 2u * (uint32_t)dot)));
}
}  // namespace cells

size_t dtw_cells_lds(uint32_t max_frames, uint32_t tpl_rows)
{
    // rows 0 .. n of both sequences + one word per point (px, py), px < in_n - 1, py < mdl_n - 1 (at least one point)
    return ((size_t)(max_frames + tpl_rows) * cells::kRowWords + (size_t)max_frames * tpl_rows) * sizeof(uint32_t);
}
bool dtw_cells_fits(const DtwArgs &a)
{
    // two rows per sequence at least (the do-while of DTW.C:150-188 reads row 1 even of 1-frame sequences); the jump to the
    // next point must fit 14 bits; one workgroup's LDS
    return a.max_frames >= 2 && a.tpl_rows >= 2 && a.tpl_rows + 1 < (1u << 14) && a.n_coef >= 1 && a.n_coef <= 16 &&
           dtw_cells_lds(a.max_frames, a.tpl_rows) <= 150 * 1024;
}

__global__ void __launch_bounds__(cells::kThreads) k_dtw_cells(const DtwArgs a, uint32_t b0)
{
    using namespace cells;
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    const uint32_t k = blockIdx.x, b = b0 + blockIdx.y, nc = a.n_coef, tid = threadIdx.x;
    uint32_t in_n, ok;
    if (a.in_frames) {
        in_n = a.in_frames[b];
        ok = in_n != 0;
    } else {
        in_n = a.vad[b].frm_num;
        ok = a.vad[b].status == SR_ST_OK && in_n != 0;
    }
    const uint32_t mdl_n = a.tpl_frames[k];
    uint32_t *out = a.scores + (size_t)b * a.K + k;
    // main.c:283, DTW.C:133-137; counts beyond the allocation (never produced by this library) are not walked
    if (!(ok && a.tpl_valid[k]) || in_n > mdl_n * 2 || 2 * in_n < mdl_n || in_n > a.max_frames || mdl_n >= a.tpl_rows) {
        if (tid == 0) *out = SR_DIS_ERR;
        return;
    }
    // rows the walk can touch: x + 1 <= max(in_n, 2), y + 1 <= max(mdl_n, 2) (1-based; row 1 of a 1-frame sequence is the
    // slack row the reference's do-while reads, DTW.C:150-154)
    const uint32_t NX = in_n > 1 ? in_n : 2, NY = mdl_n > 1 ? mdl_n : 2;
    const uint32_t MX = NX - 1, MY = NY - 1;  // points the walk can stand on: px < MX, py < MY (0-based)
    uint32_t *s_in = sm, *s_md = s_in + NX * kRowWords, *s_pt = s_md + NY * kRowWords;
    const int16_t *in = a.mfcc + (size_t)b * a.max_frames * nc, *mdl = a.tpl + (size_t)k * a.tpl_stride;
    for (uint32_t r = tid; r < NX + NY; r += kThreads) {
        if (r < NX) stage_row(s_in + r * kRowWords, in + (size_t)r * nc, nc);
        else stage_row(s_md + (r - NX) * kRowWords, mdl + (size_t)(r - NX) * nc, nc);
    }
    __syncthreads();
    const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142 (u16 statics)
    const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
    // dtw_limit (DTW.C:76-109) as an interval per column, lb(x) <= y < ub1(x) (see k_dtw_lds; every length pair checked point
    // by point against the oracle's dtw_limit in tests/test_oracle.py)
    const int c1s2 = 5 - ((int)in_n - 2 * (int)mdl_n), c2s = ((int)mdl_n - 2 * (int)in_n) - 3;
    auto ub1_of = [&](int xx) { return (xx < X1) ? 2 * xx + 2 : ((xx + c1s2) >> 1); };
    auto lb_of = [&](int xx) { return (xx < X2) ? (xx >> 1) : 2 * xx + c2s; };
    // ---- every point at once.  With E[ix][iy] = the candidate "rows ix / iy" (dtw_limit of the point (ix+1, iy+1), then get_dis;
    // dis_err outside), the point (px, py) needs up = E[px][py+1], right = E[px+1][py], diag = E[px+1][py+1]: three entries of E
    // per point, but ONE NEW entry per point when a lane keeps a column c = py + 1 and walks down the rows -- its previous
    // entry is the next point's `up`, the new one its `diag`, and `right` is the new entry of the lane to its left (one DPP
    // move across the wave).  A wave owns a block of 64 columns (63 points wide: lane 0 only feeds lane 1) and a contiguous
    // range of rows; the template row of a lane's column stays in registers, the input row is one broadcast LDS read.
    const uint32_t lane = tid & 63, wv = tid >> 6;
    const uint32_t n_cb = (MY + 62) / 63;                              // column blocks
    const uint32_t n_rg = (kThreads / 64) / n_cb ? (kThreads / 64) / n_cb : 1;  // row groups sharing the workgroup's 16 waves
    const uint32_t rows_per = (MX + n_rg - 1) / n_rg;
    for (uint32_t unit = wv; unit < n_cb * n_rg; unit += kThreads / 64) {  // (more than 16 column blocks: a wave takes several)
        const uint32_t cb = unit % n_cb, rg = unit / n_cb;
        const uint32_t p0 = rg * rows_per, p1 = (p0 + rows_per < MX) ? p0 + rows_per : MX;
        const uint32_t c = cb * 63 + lane;  // column of E = template row; the lane's points are (ix - 1, c - 1)
        if (p0 >= p1) continue;
        const bool col = c <= MY;
        const Row16 md = lds_row(s_md + (col ? c : 0u) * kRowWords);
        const int y = (int)c + 1;  // 1-based y of the candidates in column c
        auto entry = [&](uint32_t ix) {
            const Row16 ir = lds_row(s_in + ix * kRowWords);  // same address in every lane: broadcast
            const int x = (int)ix + 1;
            // (the start point (1, 1) is never a candidate: every candidate has x + 1 >= 2 or y + 1 >= 2)
            const bool inside = (lb_of(x) <= y) & (y < ub1_of(x));
            return inside ? dis_rows(md, ir) : SR_DIS_ERR;
        };
        uint32_t e_prev = entry(p0);
        for (uint32_t ix = p0 + 1; ix <= p1; ix++) {
            const uint32_t diag = entry(ix), up = e_prev;
            const uint32_t right = dpp_take<0x138, 0xF>(diag);  // wave_shr:1: E[ix][c - 1] from the lane to the left
            e_prev = diag;
            uint32_t mn = diag;  // DTW.C:156-164
            if (mn > right) mn = right;
            if (mn > up) mn = up;
            const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:168-184
            const uint32_t px = ix - 1, py = c - 1;
            const uint32_t qx = px + ((mv_diag || !mv_up) ? 1u : 0u), qy = py + ((mv_diag || mv_up) ? 1u : 0u);
            const bool stop = !(qx + 1 < in_n && qy + 1 < mdl_n);  // DTW.C:188
            const uint32_t jump = (qx - px) * MY + (qy - py);
            // a root is at most 65 535; dis_err is kept as a flag (cost field 0)
            if (lane != 0 && col) s_pt[px * MY + py] = (mn == SR_DIS_ERR ? kErrBit : mn) | (stop ? kStopBit : 0u) | (jump << 18);
        }
    }
    __syncthreads();
    if (tid != 0) return;
    // ---- the walk: DTW.C:146-191 as a chase through the points ----
    uint32_t dis = dis_rows(lds_row(s_in), lds_row(s_md));  // DTW.C:146
    uint32_t step = 1, at = 0, w;
    do {
        w = s_pt[at];
        dis += (w & 0xFFFFu) - ((w >> 16) & 1u);  // + dis_err = - 1 in the u32 ring (DTW.C:186)
        at += w >> 18;
        step = (step + 1) & 0xFFFF;  // u16 step
    } while (!(w & kStopBit));
    *out = dis / step;  // DTW.C:191
}

void launch_dtw_cells(const DtwArgs &a, hipStream_t s)
{
    if (!a.B || !a.K) return;
    const size_t lds = dtw_cells_lds(a.max_frames, a.tpl_rows);
    for (uint32_t b0 = 0; b0 < a.B; b0 += 65535) {  // utterances are the grid's second dimension
        const uint32_t nb = a.B - b0 < 65535 ? a.B - b0 : 65535;
        hipLaunchKernelGGL(k_dtw_cells, dim3(a.K, nb), dim3(cells::kThreads), lds, s, a, b0);
    }
}

}  // namespace sr
