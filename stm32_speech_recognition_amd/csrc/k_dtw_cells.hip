// k_dtw_cells.hip -- dtw (DTW.C:120-192) for SMALL launches: one capture against the store (spch_recg, main.c:276-295), one dtw() call.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
//
// The greedy walk is a serial chain of up to in + mdl steps.  In k_dtw_lds a step costs ~1 400 shader cycles when the wave
// has its SIMD to itself (three distances, a root, the tie order, the bound tests, two row fetches, all dependent): with a
// handful of pairs the GPU idles while 80 lanes crawl, 140 us for one 110-frame capture against 80 slots.  Here the chip's
// idle width is spent instead: ONE WORKGROUP PER PAIR first evaluates, for EVERY point (x, y) of dtw_limit's band at once, what
// the walk would do if it stood there -- the three candidates of DTW.C:152-154 (dtw_limit + get_dis), their minimum
// (DTW.C:156-164), the move (DTW.C:168-184) and whether the loop ends after it (DTW.C:188) -- packed into one 32-bit word per
// point in LDS; the walk itself is then one lane chasing those words (two steps per word after one more pass): an LDS read, a few adds.
// Same arithmetic as k_dtw_gen (any feature width up to 16 coefficients, plain template store), so the scores are identical.
#include <vector>

#include "sr_dtw_cells.h"
#include "sr_dtw_dev.h"

namespace sr {
namespace cells {
constexpr uint32_t kThreads = 1024;
constexpr uint32_t kRowWords = 9;  // template rows: 8 packed coefficient pairs + the squared norm; odd stride: lanes on consecutive rows, distinct banks
constexpr uint32_t kInWords = 12;  // input rows: the same nine words at a 48-byte stride (every lane reads the SAME row: three wide broadcast reads)
// word of a point = what one or two steps of the walk from there add up to:
//   bits 0-16  the cost (roots of the smallest admissible candidates, each at most 65 535)
//   bits 17-18 how many of the steps had all three candidates outside (each costs dis_err = 2^32 - 1, DTW.C:152-164)
//   bit 19     the walk ends after these steps (DTW.C:188)      bit 20  two steps (0: one)
//   bits 21-31 the distance to the point reached, in words (one step: at most MY + 1 <= 1023)
constexpr uint32_t kCostMask = 0x1FFFFu, kErrShift = 17, kStopBit = 1u << 19, kTwoBit = 1u << 20, kJumpShift = 21;
constexpr uint32_t kPairPoints = 28;  // points per thread the two-step pass keeps in registers

struct Row16 {
    uint32_t w[8];
    uint32_t n;
};
// a feature row of nc <= 16 s16 as packed pairs, zero-padded (the pad adds nothing to get_dis' sum of squares, DTW.C:45-62),
// + its squared norm.  12-coefficient rows are 24 bytes and 8-byte aligned (three wide loads); other widths are fetched
// coefficient by coefficient (rows of an odd number of s16 are only 2-byte aligned).
__device__ __forceinline__ void stage_row(uint32_t *dst, const int16_t *p, uint32_t nc)
{
    uint32_t w[8];
    if (nc == (uint32_t)kCoef) {
        const u32x2 *q = (const u32x2 *)p;
        const u32x2 q0 = q[0], q1 = q[1], q2 = q[2];
        w[0] = q0.x, w[1] = q0.y, w[2] = q1.x, w[3] = q1.y, w[4] = q2.x, w[5] = q2.y, w[6] = 0, w[7] = 0;
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t lo = (2 * i < nc) ? (uint32_t)(uint16_t)p[2 * i] : 0u, hi = (2 * i + 1 < nc) ? (uint32_t)(uint16_t)p[2 * i + 1] : 0u;
            w[i] = lo | (hi << 16);
        }
    }
    int acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        dst[i] = w[i];
        acc = sdot2(w[i], w[i], acc);
    }
    dst[8] = (uint32_t)acc;
}
template <int kWords>
__device__ __forceinline__ Row16 lds_row(const uint32_t *p)
{
    Row16 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = i < kWords ? p[i] : 0u;
    r.n = p[8];
    return r;
}
// an input row (48-byte stride, 16-byte aligned): wide reads
template <int kWords>
__device__ __forceinline__ Row16 lds_in_row(const uint32_t *p)
{
    const u32x4 *q = (const u32x4 *)p;
    const u32x4 a = q[0], b = q[1];
    Row16 r;
    r.w[0] = a.x, r.w[1] = a.y, r.w[2] = a.z, r.w[3] = a.w, r.w[4] = b.x, r.w[5] = b.y;
    r.w[6] = kWords > 6 ? b.z : 0u, r.w[7] = kWords > 6 ? b.w : 0u;
    r.n = p[8];
    return r;
}
// get_dis (DTW.C:45-62): sum (a-b)^2 in u32 wrap = |a|^2 + |b|^2 - 2 a.b in the same ring, then (u32)sqrtf.
// kWords = 6: feature rows of at most 12 coefficients (words 6 and 7 are zero padding)
template <int kWords>
__device__ __forceinline__ uint32_t dis_rows(const Row16 &a, const Row16 &b)
{
    int dot = 0;
#pragma unroll
    for (int i = 0; i < kWords; i++) dot = sdot2(a.w[i], b.w[i], dot);
    return cvt_u32(sqrt_rn_int((float)(a.n + b.n - 2u * (uint32_t)dot)));
}
}  // namespace cells

// ---- dtw_limit's band (DTW.C:76-109) as stored ranges: row px (x = px + 1) keeps the points py in [lo, hi), i.e. the
// admissible y = py + 1 of column x, lb(x) <= y < ub1(x), clipped to the points the walk can stand on (py < MY); row 0 always
// starts at 0 (the start point is never tested, DTW.C:146-148).  A walk that never meets a step with all three candidates
// outside only visits admissible points (every move goes to an admissible candidate), so these are all the points it can
// read; the other kind of walk is recognised by its first such step and done literally (see the kernel).
struct CellsBand {
    int X1, X2, c1s2, c2s;
    uint32_t MX, MY;
    __host__ __device__ CellsBand(uint32_t in_n, uint32_t mdl_n)
    {
        X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142 (u16 statics)
        X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
        c1s2 = 5 - ((int)in_n - 2 * (int)mdl_n);
        c2s = ((int)mdl_n - 2 * (int)in_n) - 3;
        MX = (in_n > 1 ? in_n : 2) - 1;
        MY = (mdl_n > 1 ? mdl_n : 2) - 1;
    }
    // dtw_limit as an interval per column, lb(x) <= y < ub1(x) (see k_dtw_lds; every length pair checked point by point
    // against the oracle's dtw_limit in tests/test_oracle.py)
    __host__ __device__ int ub1_of(int xx) const { return (xx < X1) ? 2 * xx + 2 : ((xx + c1s2) >> 1); }
    __host__ __device__ int lb_of(int xx) const { return (xx < X2) ? (xx >> 1) : 2 * xx + c2s; }
    __host__ __device__ void row(uint32_t px, uint32_t &lo, uint32_t &hi) const
    {
        const int x = (int)px + 1;
        int l = lb_of(x) - 1, h = ub1_of(x) - 1;  // py = y - 1
        if (l < 0 || px == 0) l = 0;
        if (h > (int)MY) h = (int)MY;
        lo = (uint32_t)l;
        hi = h > l ? (uint32_t)h : (uint32_t)l;
    }
    __host__ __device__ uint32_t points() const
    {
        uint32_t n = 0;
        for (uint32_t px = 0; px < MX; px++) {
            uint32_t lo, hi;
            row(px, lo, hi);
            n += hi - lo;
        }
        return n;
    }
};

// the most band points any (utterance, template) pair of this store can have: utterances of 1..max_frames frames against the
// lengths in the store, pairs that pass the gate of DTW.C:133-137.  0 = the small-launch kernel is not worth setting up
// (more than 400 rows on either side: the band alone would not fit the LDS).
uint32_t dtw_cells_max_points(uint32_t max_frames, const uint32_t *frames, const uint8_t *valid, uint32_t K, std::vector<uint32_t> &by_len)
{
    if (max_frames > 400) return 0;
    if (by_len.size() != 402) by_len.assign(402, 0u);  // by_len[m]: the most points over the admissible utterance lengths (0 = not yet known)
    uint32_t best = 1;
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t m = frames[k];
        if ((valid && !valid[k]) || m == 0) continue;
        if (m > 400) return 0;
        if (!by_len[m]) {
            uint32_t bm = 1;
            for (uint32_t n = (m + 1) / 2; n <= 2 * m && n <= max_frames; n++) {
                if (n == 0) continue;
                const uint32_t p = CellsBand(n, m).points();
                if (p > bm) bm = p;
            }
            by_len[m] = bm;
        }
        if (by_len[m] > best) best = by_len[m];
    }
    return best;
}

size_t dtw_cells_lds(uint32_t max_frames, uint32_t tpl_rows, uint32_t max_points)
{
    // every allocated row of both sequences (staged before the frame counts are known), three words per input row for the
    // stored ranges (offset, first, end) and one word per band point
    return ((size_t)max_frames * cells::kInWords + (size_t)tpl_rows * cells::kRowWords + 3 * ((size_t)max_frames + 1) + max_points) *
           sizeof(uint32_t);
}
bool dtw_cells_fits(const DtwArgs &a)
{
    // two rows per sequence at least (the do-while of DTW.C:150-188 reads row 1 even of 1-frame sequences); the jump over two
    // steps must fit 11 bits; one workgroup's LDS
    return a.cells_points != 0 && a.max_frames >= 2 && a.tpl_rows >= 2 && a.tpl_rows <= 1023 && a.n_coef >= 1 && a.n_coef <= 16 &&
           dtw_cells_lds(a.max_frames, a.tpl_rows, a.cells_points) <= 150 * 1024;
}

#ifdef SR_CELLS_TIMING
// development build only (-DSR_CELLS_TIMING): s_memtime at the phase boundaries of the workgroup of pair (0, 0)
__device__ unsigned long long g_cells_t[8];
extern "C" void sr_debug_cells_timing(unsigned long long *out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cells_t), sizeof(g_cells_t));
}
#define CELLS_T(i)                                                                        \
    do {                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_cells_t[i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define CELLS_T(i) ((void)0)
#endif
template <int kWords>
__global__ void __launch_bounds__(cells::kThreads) k_dtw_cells(const DtwArgs a, uint32_t b0)
{
    CELLS_T(0);
    using namespace cells;
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    const uint32_t k = blockIdx.x, b = b0 + blockIdx.y, nc = a.n_coef, tid = threadIdx.x;
    // every allocated row of the pair goes to LDS first -- before the frame counts have arrived, so that the two round trips
    // to memory (the records, the rows) overlap
    uint32_t *s_in = sm, *s_md = s_in + a.max_frames * kInWords, *s_off = s_md + a.tpl_rows * kRowWords;
    uint32_t *s_lo = s_off + (a.max_frames + 1), *s_hi = s_lo + (a.max_frames + 1), *s_pt = s_hi + (a.max_frames + 1);
    {
        const int16_t *in = a.mfcc + (size_t)b * a.max_frames * nc, *mdl = a.tpl + (size_t)k * a.tpl_stride;
        for (uint32_t r = tid; r < a.max_frames + a.tpl_rows; r += kThreads) {
            if (r < a.max_frames) stage_row(s_in + r * kInWords, in + (size_t)r * nc, nc);
            else stage_row(s_md + (r - a.max_frames) * kRowWords, mdl + (size_t)(r - a.max_frames) * nc, nc);
        }
    }
    uint32_t in_n, ok;
    if (a.in_frames) {
        in_n = a.in_frames[b];
        ok = in_n != 0;
    } else {
        in_n = a.vad[b].frm_num;
        ok = a.vad[b].status == SR_ST_OK && in_n != 0;
    }
    const uint32_t mdl_n = a.tpl_frames[k];
    // (the wave index as a scalar: row ranges, loop counters and the column bounds of dtw_limit stay on the scalar unit)
    const uint32_t lane = tid & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint32_t score = SR_DIS_ERR;
    // main.c:283, DTW.C:133-137; counts beyond the allocation (never produced by this library) are not walked
    if (ok && a.tpl_valid[k] && !(in_n > mdl_n * 2 || 2 * in_n < mdl_n || in_n > a.max_frames || mdl_n >= a.tpl_rows)) {  // workgroup-uniform
        const CellsBand band(in_n, mdl_n);
        const uint32_t MX = band.MX, MY = band.MY;  // points the walk can stand on: px < MX, py < MY (0-based)
        // ---- stored range of every row and its offset in the point array
        for (uint32_t px = tid; px < MX; px += kThreads) {
            uint32_t lo, hi;
            band.row(px, lo, hi);
            s_lo[px] = lo;
            s_hi[px] = hi;
            s_off[px] = hi - lo;
        }
        if (tid == 0) s_lo[MX] = s_hi[MX] = 0;
        __syncthreads();
        if (wv == 0) {  // exclusive prefix sum of the row widths, 64 rows at a time
            uint32_t carry = 0;
            for (uint32_t base = 0; base < MX; base += 64) {
                const uint32_t w = base + lane < MX ? s_off[base + lane] : 0u;
                uint32_t incl = w;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d, 64);
                    if ((int)lane >= d) incl += o;
                }
                if (base + lane < MX) s_off[base + lane] = carry + incl - w;
                carry += __shfl(incl, 63, 64);
            }
            if (lane == 0) s_off[MX] = carry;
        }
        __syncthreads();
        CELLS_T(1);  // rows staged, ranges known
        const uint32_t npts = s_off[MX];
        // ---- every point of the band at once.  With E[ix][iy] = the candidate "rows ix / iy" (dtw_limit of the point
        // (ix+1, iy+1), then get_dis; dis_err outside), the point (px, py) needs up = E[px][py+1], right = E[px+1][py],
        // diag = E[px+1][py+1]: three entries of E per point, but ONE NEW entry per point when a lane keeps a column c = py + 1
        // and walks down the rows -- its previous entry is the next point's `up`, the new one its `diag`, and `right` is the new
        // entry of the lane to its left (one DPP move across the wave).  A wave owns a contiguous range of rows and, block by
        // block of 64 columns (63 points wide: lane 0 only feeds lane 1), the columns their stored ranges span; the template
        // row of a lane's column stays in registers, the input row is one broadcast LDS read.
        bool literal = a.cells_literal != 0 || npts > a.cells_points;  // (the second: never, dtw_cells_max_points bounds it)
        if (!literal) {
            const uint32_t rows_per = (MX + kThreads / 64 - 1) / (kThreads / 64);
            const uint32_t p0 = wv * rows_per, p1 = (p0 + rows_per < MX) ? p0 + rows_per : MX;
            if (p0 < p1) {
                uint32_t c_lo = 0xFFFFFFFFu, c_hi = 0;  // columns of E the group's stored points need: [c_lo, c_hi]
                for (uint32_t px = p0; px < p1; px++) {
                    const uint32_t lo = s_lo[px], hi = s_hi[px];
                    if (hi > lo) {
                        c_lo = lo < c_lo ? lo : c_lo;  // feeder column = first point's py
                        c_hi = hi > c_hi ? hi : c_hi;  // last point's py + 1
                    }
                }
                c_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_lo);
                c_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_hi);
                for (uint32_t cbase = c_lo; cbase < c_hi; cbase += 63) {
                    const uint32_t c = cbase + lane;  // column of E = template row; the lane's points are (ix - 1, c - 1)
                    const bool col = c <= MY;
                    const Row16 md = lds_row<kWords>(s_md + (col ? c : 0u) * kRowWords);
                    const int y = (int)c + 1;  // 1-based y of the candidates in column c
                    auto entry = [&](uint32_t ix) {
                        const Row16 ir = lds_in_row<kWords>(s_in + ix * kInWords);  // same address in every lane: broadcast
                        const int x = (int)ix + 1;
                        // (the start point (1, 1) is never a candidate: every candidate has x + 1 >= 2 or y + 1 >= 2)
                        const bool inside = band.lb_of(x) <= y && y < band.ub1_of(x);
                        return inside ? dis_rows<kWords>(md, ir) : SR_DIS_ERR;
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
                        const uint32_t dx = (mv_diag || !mv_up) ? 1u : 0u, dy = (mv_diag || mv_up) ? 1u : 0u;
                        const bool stop = !(px + dx + 1 < in_n && py + dy + 1 < mdl_n);  // DTW.C:188
                        // rows px, px + 1: offset of the row in the point array minus its first column
                        const int base0 = (int)s_off[px] - (int)s_lo[px], base1 = (int)s_off[ix] - (int)s_lo[ix];
                        const uint32_t jump = stop ? 0u : (uint32_t)((dx ? base1 - base0 : 0) + (int)dy);
                        // a root is at most 65 535; dis_err is kept as a flag (cost field 0)
                        const uint32_t word = (mn == SR_DIS_ERR ? (1u << kErrShift) : mn) | (stop ? kStopBit : 0u) | (jump << kJumpShift);
                        if (lane != 0 && py >= s_lo[px] && py < s_hi[px]) s_pt[(uint32_t)(base0 + (int)py)] = word;
                    }
                }
            }
        }
        __syncthreads();
        CELLS_T(2);  // points
        // ---- two steps per word: every point absorbs the point its step leads to (all reads, a barrier, all writes: in place).
        // The walk below is a chain of dependent LDS reads, ~110 cycles each; this halves it for one more pass over the points.
        if (!literal && npts <= kPairPoints * kThreads) {
            uint32_t keep[kPairPoints];
#pragma unroll
            for (uint32_t i = 0; i < kPairPoints; i++) {
                const uint32_t c = tid + i * kThreads;
                uint32_t w = 0;
                if (i * kThreads < npts && c < npts) {  // (the first test is uniform: whole rounds past the last point are skipped)
                    w = s_pt[c];
                    if (!(w & kStopBit) && !((w >> kErrShift) & 3u)) {
                        const uint32_t w2 = s_pt[c + (w >> kJumpShift)];
                        // costs, outside counts and jumps add up field by field (no carry: 2 x 65 535 < 2^17, 1 + 1 < 4, the jump
                        // bound is checked by dtw_cells_fits); the second step decides whether the walk ends
                        w = (w + (w2 & ~(kStopBit | kTwoBit))) | (w2 & kStopBit) | kTwoBit;
                    }
                }
                keep[i] = w;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t i = 0; i < kPairPoints; i++) {
                const uint32_t c = tid + i * kThreads;
                if (i * kThreads < npts && c < npts) s_pt[c] = keep[i];
            }
            __syncthreads();
        }
        CELLS_T(3);  // pair pass
        if (tid == 0) {
            uint32_t dis = dis_rows<kWords>(lds_in_row<kWords>(s_in), lds_row<kWords>(s_md));  // DTW.C:146
            uint32_t step = 1;
            if (!literal) {
                // ---- the walk: DTW.C:146-191 as a chase through the points ----
                uint32_t at = 0, w;
                do {
                    w = s_pt[at];
                    if ((w >> kErrShift) & 3u) {  // a step with all three candidates outside leaves the band: the literal walk takes over
                        literal = true;
                        break;
                    }
                    dis += w & kCostMask;
                    at += w >> kJumpShift;
                    step += 1 + ((w >> 20) & 1u);
                } while (!(w & kStopBit));
            }
            if (literal) {
                // ---- the literal walk (k_dtw_gen's loop on the staged rows): dtw_limit on the three points, three roots, the
                // minimum, the equality tests.  Never taken on any data seen so far (a walk that runs out of admissible
                // candidates, DTW.C:152-164 with three dis_err); the development hook "cells_literal" forces it for the tests.
                uint32_t px = 0, py = 0;
                dis = dis_rows<kWords>(lds_in_row<kWords>(s_in), lds_row<kWords>(s_md));
                step = 1;
                do {
                    const Row16 ci = lds_in_row<kWords>(s_in + px * kInWords), ni = lds_in_row<kWords>(s_in + (px + 1) * kInWords);
                    const Row16 cm = lds_row<kWords>(s_md + py * kRowWords), nm = lds_row<kWords>(s_md + (py + 1) * kRowWords);
                    const int x = (int)px + 1, y = (int)py + 1;
                    const uint32_t up = dtw_out(x, y + 1, band.X1, band.X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : dis_rows<kWords>(nm, ci);
                    const uint32_t right = dtw_out(x + 1, y, band.X1, band.X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : dis_rows<kWords>(cm, ni);
                    const uint32_t diag = dtw_out(x + 1, y + 1, band.X1, band.X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : dis_rows<kWords>(nm, ni);
                    uint32_t mn = diag;  // DTW.C:156-164
                    if (mn > right) mn = right;
                    if (mn > up) mn = up;
                    dis += mn;
                    const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:168-184
                    if (mv_diag || !mv_up) px++;
                    if (mv_diag || mv_up) py++;
                    step++;
                } while (px + 1 < in_n && py + 1 < mdl_n);  // DTW.C:188
            }
            step &= 0xFFFF;  // u16 step (DTW.C:126)
            score = dis / step;  // DTW.C:191
        }
    }
    CELLS_T(4);  // walk
    if (wv != 0) return;
    uint32_t *sc = a.scores + (size_t)b * a.K;
    if (lane == 0) __hip_atomic_store(sc + k, score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!a.results || !a.pair_count) return;
    // ---- the slot scan of spch_recg (main.c:276-295) by whichever workgroup of the utterance finishes last: the count of
    // finished pairs is taken after the score is out (release / acquire at device scope), the scores are read back coherently,
    // and the counter is left at zero for the next launch.  Strict '<' in slot order: the first minimum wins; all dis_err ->
    // slot 0 (see k_argmin).
    uint32_t last = 0;
    if (lane == 0) last = __hip_atomic_fetch_add(a.pair_count + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == a.K - 1 ? 1u : 0u;
    if (!__builtin_amdgcn_readfirstlane((int)last)) return;
    uint32_t best = SR_DIS_ERR, idx = 0xFFFFFFFFu;
    for (uint32_t kk = lane; kk < a.K; kk += 64) {
        const uint32_t d = __hip_atomic_load(sc + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d < best) {
            best = d;
            idx = kk;
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
        __hip_atomic_store(a.pair_count + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    CELLS_T(5);  // slot scan (only if this pair was the last one)
}

void launch_dtw_cells(const DtwArgs &a, hipStream_t s)
{
    if (!a.B || !a.K) return;
    const size_t lds = dtw_cells_lds(a.max_frames, a.tpl_rows, a.cells_points);
    for (uint32_t b0 = 0; b0 < a.B; b0 += 65535) {  // utterances are the grid's second dimension
        const uint32_t nb = a.B - b0 < 65535 ? a.B - b0 : 65535;
        if (a.n_coef <= (uint32_t)kCoef) hipLaunchKernelGGL(k_dtw_cells<6>, dim3(a.K, nb), dim3(cells::kThreads), lds, s, a, b0);
        else hipLaunchKernelGGL(k_dtw_cells<8>, dim3(a.K, nb), dim3(cells::kThreads), lds, s, a, b0);
    }
}

}  // namespace sr
