// k_dtw_quad.hip -- dtw (DTW.C:120-192) for MID-SIZED launches (a few thousand to ~100 000 pairs): FOUR LANES PER PAIR.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
//
// Between the launches k_dtw_cells serves (one workgroup per pair: a few hundred pairs) and the ones that fill the chip with
// k_dtw_lds' one-lane-per-pair waves (hundreds of thousands of pairs) the batch kernel's time does not depend on the size at
// all: 126 us for 5 000 as for 160 000 pairs of 110 x 119 frames, because a wave that has its SIMD to itself needs ~1 400
// shader cycles for one step of the walk -- ~100 vector + ~37 scalar instructions, all of one dependent chain (three distances
// one after the other, the minimum under three admissibility masks, one root, the tie thresholds, two exec-masked row
// advances with their copies).  The chip is mostly idle at these sizes, so the step is made SHORT instead of narrow:
//
//   * the three candidates of a step (DTW.C:152-154: diag (x+1, y+1), up (x, y+1), right (x+1, y)) are evaluated by three
//     lanes of a quad AT THE SAME TIME -- one dtw_limit (as the interval of its column), one get_dis incl. its exactly
//     rounded root (sqrt_rn_int) per lane, no admissibility masks, no tie table, no bracket, no literal fallback: every
//     lane simply has the reference's value of its candidate (dis_err outside the band).  The fourth lane repeats the
//     diagonal one.
//   * minimum AND move in one reduction: each lane forms key = (~cost << 2) | move with move = 3 (diag: x and y advance),
//     2 (up: y), 1 (right: x); two v_max_u32 over the quad (DPP quad_perm) leave the largest key in all four lanes: the
//     smallest cost, and among equal costs diag before up before right -- the order of DTW.C:168-184.  cost = ~(key >> 2)
//     (arithmetic shift: the all-outside key (0 << 2 | move) gives dis_err = 2^32 - 1, which DTW.C:156-164 then adds, wrapping),
//     and bits 0 / 1 of the key are the increments of x / y.
//   * BOTH sequences are staged in LDS (24-byte coefficient rows + an array of squared norms: 28 bytes per row; 32 + 4 up to
//     16 coefficients; template rows as -2 * coefficient when the store allows it) and every lane reads the two rows of its
//     own candidate point afresh in every step -- three 8-byte reads + one 4-byte read each, no register copies, no
//     exec-masked advance blocks, no template rows on their way from L2.
//
// A step is 43 vector instructions on a chain of one LDS round trip, six dot products, one root and two DPP moves.  A
// workgroup of 256 lanes walks PU utterances x PK templates (PU * PK <= 64 pairs), the shape that keeps the most pairs
// resident per CU (8 x 8 = 53 536 bytes = three workgroups per CU at the firmware's 119 / 120 rows).
// Same arithmetic as k_dtw / k_dtw_gen (rows as packed pairs + norm, |a|^2 + |b|^2 - 2 a.b in the u32 ring): identical scores.
#include <algorithm>

#include "sr_dtw_dev.h"
#include "sr_dtw_quad.h"

namespace sr {
namespace quad {
constexpr uint32_t kThreads = 256;  // 64 quads
// A staged record: the rows' coefficients (kWords packed pairs each: 24 or 32 bytes, 8-byte aligned) back to back, then their
// squared norms (4 bytes each).  28 bytes per 12-coefficient row: at the firmware's shapes (119 / 120 rows) eight utterances
// and eight templates take 53 536 bytes, i.e. THREE workgroups of 64 pairs per CU (gfx950 hands out LDS in 1 280-byte granules).
__host__ __device__ constexpr uint32_t coef_bytes(int words) { return 4u * (uint32_t)words; }
__host__ __device__ constexpr uint32_t rec_bytes(uint32_t rows, int words) { return (rows * (coef_bytes(words) + 4u) + 7u) & ~7u; }

// row r of a record from global memory (nc s16 per row) -> packed pairs (x -2 for template rows when kScale) + squared norm
template <int kWords>
__device__ __forceinline__ void stage_row(uint8_t *rec, uint32_t rows, uint32_t row, const int16_t *p, uint32_t nc, bool neg2)
{
    uint32_t w[8];
    if (nc == (uint32_t)kCoef) {  // 24-byte rows, 8-byte aligned: three wide loads
        const u32x2 *q = (const u32x2 *)p;
        const u32x2 q0 = q[0], q1 = q[1], q2 = q[2];
        w[0] = q0.x, w[1] = q0.y, w[2] = q1.x, w[3] = q1.y, w[4] = q2.x, w[5] = q2.y, w[6] = 0, w[7] = 0;
    } else {  // any other width: rows of an odd number of s16 are only 2-byte aligned
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t lo = (2 * i < nc) ? (uint32_t)(uint16_t)p[2 * i] : 0u, hi = (2 * i + 1 < nc) ? (uint32_t)(uint16_t)p[2 * i + 1] : 0u;
            w[i] = lo | (hi << 16);
        }
    }
    int acc = 0;
#pragma unroll
    for (int i = 0; i < kWords; i++) acc = sdot2(w[i], w[i], acc);
    u32x2 *d = (u32x2 *)(rec + (size_t)row * coef_bytes(kWords));
#pragma unroll
    for (int i = 0; i < kWords; i += 2) {
        // template rows as -2 * coefficient (both halves at once, mod 2^16: exact because the store's coefficients were checked
        // to lie in [-16383, 16384] when it was set): a candidate is then |a|^2 + |b|^2 + six accumulating dot products
        const uint32_t w0 = neg2 ? pk_mad(w[i], 0xFFFEFFFEu, 0u) : w[i], w1 = neg2 ? pk_mad(w[i + 1], 0xFFFEFFFEu, 0u) : w[i + 1];
        d[i / 2] = u32x2{w0, w1};
    }
    ((uint32_t *)(rec + (size_t)rows * coef_bytes(kWords)))[row] = (uint32_t)acc;
}

template <int kWords>
struct QRow {
    uint32_t w[kWords];
    uint32_t n;
};
// the row whose coefficients start at LDS address `pc` and whose norm lies at `pn`: kWords / 2 8-byte reads + one 4-byte read
template <int kWords>
__device__ __forceinline__ QRow<kWords> lds_row(const uint8_t *pc, const uint8_t *pn)
{
    const u32x2 *q = (const u32x2 *)pc;
    QRow<kWords> r;
#pragma unroll
    for (int i = 0; i < kWords; i += 2) {
        const u32x2 v = q[i / 2];
        r.w[i] = v.x;
        r.w[i + 1] = v.y;
    }
    r.n = *(const uint32_t *)pn;
    return r;
}
// squared distance of get_dis (DTW.C:45-62): sum (a-b)^2 in u32 wrap = |a|^2 + |b|^2 - 2 a.b in the same ring.
// kNeg2: the b row holds -2 * coefficient, so the whole sum is the norms + kWords accumulating dot products
template <int kWords, bool kNeg2>
__device__ __forceinline__ uint32_t dist2_rows(const QRow<kWords> &a, const QRow<kWords> &b)
{
    if (kNeg2) {
        int acc = sdot2a(a.w[0], b.w[0], (int)(a.n + b.n));
#pragma unroll
        for (int i = 1; i < kWords; i++) acc = sdot2(a.w[i], b.w[i], acc);
        return (uint32_t)acc;
    }
    int dot = sdot2z(a.w[0], b.w[0]);
#pragma unroll
    for (int i = 1; i < kWords; i++) dot = sdot2(a.w[i], b.w[i], dot);
    return a.n + b.n - 2u * (uint32_t)dot;
}
// the largest key of the quad in all four of its lanes
__device__ __forceinline__ uint32_t quad_max(uint32_t v)
{
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
    v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
    return o > v ? o : v;
}
}  // namespace quad

template <int kWords, bool kNeg2>
__global__ void __launch_bounds__(quad::kThreads) k_dtw_quad(const DtwArgs a, uint32_t pu, uint32_t pk, uint32_t b_base)
{
    using namespace quad;
    extern __shared__ __attribute__((aligned(16))) uint8_t qsm[];
    const uint32_t tid = threadIdx.x, nc = a.n_coef;
    const uint32_t in_rec = rec_bytes(a.max_frames, kWords), tp_rec = rec_bytes(a.tpl_rows, kWords);
    uint8_t *s_in = qsm, *s_tp = qsm + (size_t)pu * in_rec;
    const uint32_t b0 = b_base + blockIdx.y * pu, k0 = blockIdx.x * pk;

    // ---- stage the rows the walks can read: rows 0 .. n of a sequence of n frames (the do-while of DTW.C:150-188 reads row 1 even
    //      of a 1-frame sequence), never more than are allocated
    for (uint32_t r = 0; r < pu + pk; r++) {  // workgroup-uniform loop over the records
        uint32_t n = 0, rows;
        const int16_t *src;
        uint8_t *dst;
        if (r < pu) {
            const uint32_t b = b0 + r;
            rows = a.max_frames;
            src = a.mfcc + (size_t)(b < a.B ? b : 0) * a.max_frames * nc;
            dst = s_in + (size_t)r * in_rec;
            if (b < a.B) n = a.in_frames ? a.in_frames[b] : (a.vad[b].status == SR_ST_OK ? a.vad[b].frm_num : 0u);
        } else {
            const uint32_t k = k0 + (r - pu);
            rows = a.tpl_rows;
            src = a.tpl + (size_t)(k < a.K ? k : 0) * a.tpl_stride;
            dst = s_tp + (size_t)(r - pu) * tp_rec;
            if (k < a.K && a.tpl_valid[k]) n = a.tpl_frames[k];
        }
        const uint32_t n_stage = n ? (n + 1 < rows ? n + 1 : rows) : 0u;
        for (uint32_t row = tid; row < n_stage; row += kThreads) stage_row<kWords>(dst, rows, row, src + (size_t)row * nc, nc, kNeg2 && r >= pu);
    }
    __syncthreads();

    // ---- one quad per pair
    const uint32_t q = tid >> 2, role = tid & 3;
    const uint32_t ul = q / pk, kl = q - ul * pk;
    const uint32_t b = b0 + ul, k = k0 + kl;
    const bool have = ul < pu && b < a.B && k < a.K;
    uint32_t in_n = 0, ok = 0, mdl_n = 0, valid = 0;
    if (have) {
        if (a.in_frames) {
            in_n = a.in_frames[b];
            ok = in_n != 0;
        } else {
            in_n = a.vad[b].frm_num;
            ok = a.vad[b].status == SR_ST_OK && in_n != 0;
        }
        mdl_n = a.tpl_frames[k];
        valid = a.tpl_valid[k];
    }
    // main.c:283, DTW.C:133-137; counts beyond the allocation (never produced by this library) are not walked
    const bool walk = have && ok && valid && !(in_n > mdl_n * 2 || 2 * in_n < mdl_n || in_n > a.max_frames || mdl_n >= a.tpl_rows);
    uint32_t score = SR_DIS_ERR;
    if (walk) {  // whole quads: the four lanes of a pair share every condition
        const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142 (u16 statics)
        const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
        // LDS byte offsets (32-bit address arithmetic) of the pair's two records: coefficient rows, then norms
        const uint32_t ra = ul * in_rec, rb = pu * in_rec + kl * tp_rec;
        const uint32_t na = ra + a.max_frames * coef_bytes(kWords), nb = rb + a.tpl_rows * coef_bytes(kWords);
        // the lane's candidate: move bits (bit 0: x advances, bit 1: y advances) and its point (px, py), 1-based
        const uint32_t mv = role == 1 ? 2u : role == 2 ? 1u : 3u;
        const uint32_t mv4 = mv - 4u;  // see the key below
        const int dx = (int)(mv & 1u), dy = (int)(mv >> 1);
        int px = 1 + dx, py = 1 + dy;
        // The walk goes on while x < in && y < mdl (DTW.C:188), i.e. px < in + dx && py < mdl + dy.  Rows read: px - 1 <= in and
        // py - 1 <= mdl (0-based) -- row `in` only by the first trip of a 1-frame sequence -- all of them staged above
        // (in + 1 <= max_frames except for a full record, whose row `in` is never reached: its first trip reads row 1 <= in - 1)
        const int x_end = (int)in_n + dx, y_end = (int)mdl_n + dy;
        const uint32_t ra1 = ra - coef_bytes(kWords), rb1 = rb - coef_bytes(kWords), na1 = na - 4u, nb1 = nb - 4u;  // indexed by px / py
        uint32_t dis;
        {
            const uint32_t d0 = dist2_rows<kWords, kNeg2>(lds_row<kWords>(qsm + ra, qsm + na), lds_row<kWords>(qsm + rb, qsm + nb));
            dis = cvt_u32(sqrt_rn_int((float)d0));  // DTW.C:146-148
        }
        uint32_t step = 1;
        const int c1s2 = 5 - ((int)in_n - 2 * (int)mdl_n), c2s = ((int)mdl_n - 2 * (int)in_n) - 3;  // dtw_limit's column bounds, below
        do {
            // all reads of the step are issued first, the band test runs while they are on their way
            const QRow<kWords> fa = lds_row<kWords>(qsm + (ra1 + umul24((uint32_t)px, coef_bytes(kWords))), qsm + (na1 + 4u * (uint32_t)px));
            const QRow<kWords> fb = lds_row<kWords>(qsm + (rb1 + umul24((uint32_t)py, coef_bytes(kWords))), qsm + (nb1 + 4u * (uint32_t)py));
            // dtw_limit as the interval of its column (an algebraic identity over the integers, the form k_dtw_lds / k_dtw_cells /
            // k_dtw_dp use): (px, py) is inside <=> lb(px) <= py < ub1(px),
            //   ub1(x) = x < X1 ? 2x + 2 : (x + 5 - in + 2 mdl) >> 1      (DTW.C:80-92:  y >= 2x+2  /  2y + in - 2mdl >= x + 4)
            //   lb(x)  = x < X2 ? x >> 1 : 2x + mdl - 2 in - 3            (DTW.C:94-106: 2y+2 <= x  /  y + 4 <= 2x + mdl - 2in)
            const int ub1 = px < X1 ? 2 * px + 2 : (px + c1s2) >> 1;
            const int lb = px < X2 ? px >> 1 : 2 * px + c2s;
            const bool out = !((lb <= py) & (py < ub1));
            // the candidate's key (~g << 2) | mv from the NEGATED root: Markstein's last fused step with both addends negated gives
            // -sqrtf(d) (IEEE negation is exact), its signed truncation is -g, and (~g << 2) | mv = (-g << 2) + (mv - 4): one
            // v_lshl_add_u32 instead of v_not + v_lshl_or.  (d = 0: the seed is inf, the result NaN, both conversions give 0.)
            int ng;
            {
                const float f = (float)dist2_rows<kWords, kNeg2>(fa, fb);
                const float y = __builtin_amdgcn_rsqf(f), s0 = f * y, h = 0.5f * y;
                const float r = __builtin_fmaf(-s0, s0, f);
                const float ns = __builtin_fmaf(-r, h, -s0);
                asm("v_cvt_i32_f32 %0, %1" : "=v"(ng) : "v"(ns));
            }
            const uint32_t kv = ((uint32_t)ng << 2) + mv4;
            const uint32_t key = quad_max(out ? mv : kv);                 // DTW.C:152-184 in one reduction
            dis += (uint32_t)~((int)key >> 2);                            // + min (dis_err when all three are outside)
            px += (int)(key & 1u);
            {
                uint32_t b1;
                asm("v_bfe_u32 %0, %1, 1, 1" : "=v"(b1) : "v"(key));
                py += (int)b1;
            }
            step++;  // (u16 in the reference: a walk has fewer than in + mdl <= 32 766 steps)
        } while (px < x_end && py < y_end);
        score = dis / step;  // DTW.C:191
    }
    if (have && role == 0) a.scores[(size_t)b * a.K + k] = score;
}

// workgroup shape for a store: PU utterances x PK templates per workgroup, the most pairs (<= 64) whose records fit the LDS
// budget; false = not even 1 x 4 fits (very long sequences)
bool dtw_quad_pick(const DtwArgs &a, uint32_t *pu, uint32_t *pk, size_t *lds)
{
    if (a.max_frames < 2 || a.tpl_rows < 2 || a.n_coef < 1 || a.n_coef > 16 || !a.K) return false;
    const int words = a.n_coef <= (uint32_t)kCoef ? 6 : 8;
    const size_t in_rec = quad::rec_bytes(a.max_frames, words), tp_rec = quad::rec_bytes(a.tpl_rows, words);
    static const uint32_t shapes[][2] = {{8, 8}, {4, 16}, {4, 8}, {2, 16}, {4, 4}, {2, 8}, {1, 16}, {2, 4}, {1, 8}, {1, 4}};
    uint64_t best = 0;
    // LDS of the device the engine runs on (MI355X: 128 granules of 1 280 bytes per CU, all of which one workgroup may take)
    const size_t cu_gran = (a.dev_lds_cu ? a.dev_lds_cu : 160u * 1024u) / 1280, wg_max = a.dev_lds_wg ? a.dev_lds_wg : 160u * 1024u;
    for (const auto &sh : shapes) {  // the most pairs resident per CU (workgroups by LDS granules, at most 8 of four waves)
        const size_t need = sh[0] * in_rec + sh[1] * tp_rec, gran = (need + 1279) / 1280;
        if (gran + 8 > cu_gran || need > wg_max) continue;
        const uint64_t per_cu = std::min<uint64_t>(8, cu_gran / gran) * sh[0] * sh[1];
        if (per_cu <= best) continue;
        best = per_cu;
        *pu = sh[0];
        *pk = sh[1];
        *lds = need;
    }
    return best != 0;
}
bool dtw_quad_fits(const DtwArgs &a)
{
    uint32_t pu, pk;
    size_t lds;
    return dtw_quad_pick(a, &pu, &pk, &lds);
}

void launch_dtw_quad(const DtwArgs &a, hipStream_t s)
{
    if (!a.B || !a.K) return;
    uint32_t pu = 0, pk = 0;
    size_t lds = 0;
    if (!dtw_quad_pick(a, &pu, &pk, &lds)) return;  // callers check dtw_quad_fits first
    if (lds > 64 * 1024) {  // above the default limit the kernel's dynamic LDS has to be allowed explicitly (once per instance)
        const void *f = a.n_coef <= (uint32_t)kCoef ? (a.tpl_neg2_ok ? (const void *)k_dtw_quad<6, true> : (const void *)k_dtw_quad<6, false>)
                                                    : (a.tpl_neg2_ok ? (const void *)k_dtw_quad<8, true> : (const void *)k_dtw_quad<8, false>);
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(a.dev_lds_wg ? a.dev_lds_wg : 160u * 1024u));
    }
    const uint32_t gx = (a.K + pk - 1) / pk;
    for (uint32_t b0 = 0; b0 < a.B; b0 += 65535u * pu) {  // utterance groups are the grid's second dimension
        const uint32_t nb = a.B - b0 < 65535u * pu ? a.B - b0 : 65535u * pu;
        const dim3 grid(gx, (nb + pu - 1) / pu);
        const dim3 blk(quad::kThreads);
        if (a.n_coef <= (uint32_t)kCoef) {
            if (a.tpl_neg2_ok) hipLaunchKernelGGL((k_dtw_quad<6, true>), grid, blk, lds, s, a, pu, pk, b0);
            else hipLaunchKernelGGL((k_dtw_quad<6, false>), grid, blk, lds, s, a, pu, pk, b0);
        } else {
            if (a.tpl_neg2_ok) hipLaunchKernelGGL((k_dtw_quad<8, true>), grid, blk, lds, s, a, pu, pk, b0);
            else hipLaunchKernelGGL((k_dtw_quad<8, false>), grid, blk, lds, s, a, pu, pk, b0);
        }
    }
}

}  // namespace sr
