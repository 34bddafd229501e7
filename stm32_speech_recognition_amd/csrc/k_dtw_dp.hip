// k_dtw_dp.hip -- OPT-IN, NON-REFERENCE full dynamic-programming DTW scorer.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include <type_traits>

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

__global__ void __launch_bounds__(256) k_dtw_dp_wave64(const DtwArgs a)
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

// ------------------------------------------------------------------------------------------------
// k_dtw_dp_band (round 4): the same recurrence, band-limited and with G lanes per pair.
//
// Why: k_dtw_dp_wave64 above gives one (utterance, template) pair a whole wave and sweeps the full in x mdl rectangle
// 64 columns at a time (mdl + 63 steps per chunk, dtw_limit evaluated per cell, utterance rows from HBM per pair):
// at 256 x 256 frames 1 276 steps of which 27 % of the lane-steps carry a cell of the parallelogram.  Here
//   * a pair is walked in STRIPS of G columns (G = 4 / 8 / 16 lanes, 64 / G pairs per wave), lane j of the group owns
//     column x = G*s + j + 1 of strip s and meets row r = t - j at step t (the systolic skew), and the step counter t only
//     runs over the rows the strip's columns admit: [min_j(lb_j + j), max_j(ub_j + j)] with lb / ub the interval form of
//     dtw_limit (DTW.C:76-109; evaluated once per column, not per cell).  Narrow strips waste little of the
//     parallelogram: at 256 x 256 frames G = 8 needs 32 strips x ~108 steps for 8 pairs per wave = 432 wave-steps per
//     pair (79 % of the lane-steps carry a cell), G = 16 about 528, the 64-wide sweep 1 276;
//   * the value from the left, D(x-1, r), is the neighbour lane's result of the previous step: one DPP row_shr:1; lane
//     0 of a group takes it from the group's boundary column in LDS (the last lane of the previous strip wrote it there),
//     the diagonal D(x-1, r-1) is last step's value from the left, D(x, r-1) the lane's own last result;
//   * the template is staged once per workgroup from the length-sorted copy of the store (rows of 12 x s16 holding
//     -2*coef | squared norm, as k_dtw_lds uses them), so a squared distance is the norm sum fed through six accumulating
//     v_dot2_i32_i16; all groups of the workgroup score against the SAME template, so at equal utterance lengths the
//     lanes j of different groups read the same row: LDS broadcast, no bank conflicts;
//   * the utterance frame of a lane's column stays in registers for the whole strip (7 VGPRs), fetched a strip ahead;
//   * min(D(x-1,y-1), D(x-1,y), D(x,y-1)) is one v_min3_u32, "+ d" one saturating add (0xFFFFFFFF = unreachable stays
//     unreachable), "inside the band" one unsigned compare of a per-lane counter against the band height + one select.
// The boundary column of a strip must also be right where the NEXT strip's first column looks outside the rows the
// strip itself covered (the bounds of dtw_limit are not monotone at the switch columns X1 / X2): the step range of a
// strip is widened to the band of the next strip's first column, so stale entries of an older strip are overwritten
// with "unreachable" before they are read.
// Sums cannot reach the oracle's saturation value: d < 2^16 and a path has at most in + mdl <= 32 766 points.
// Stores whose coefficients do not fit -2*coef (|coef| > 16383) or whose rows do not fit the LDS budget go to
// k_dtw_dp_wave64.
// ------------------------------------------------------------------------------------------------
struct DpBandArgs {
    DtwArgs d;
    const u32x4 *tplR;             // [tpl_rows][K] 32-byte rows, length order (see k_dtw_lds)
    const uint32_t *tpl_frames_s;  // [K] frames in length order, 0 = invalid slot
    const uint32_t *tpl_orig;      // [K] slot of each rank
    uint32_t rows_pad;             // LDS rows of the template image / words of one boundary column: G + (tpl_rows - 1) + G - 1
};

__device__ __forceinline__ uint32_t add_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// wave-wide minimum / maximum of a signed value, result uniform (SGPR): a prefix scan inside each row of 16 lanes by DPP
// row shifts (a lane without a source keeps its own value), then the four row results (lanes 15, 31, 47, 63) are read
// back to the scalar unit.  Four DPP operations + four v_readlane per reduction; the ds_bpermute butterflies of
// __shfl_xor cost six LDS round trips each, which is what a strip's set-up waited for.
template <bool MAX>
__device__ __forceinline__ int wave_reduce_i32(int v)
{
#define SR_DP_STEP(ctrl)                                                                  \
    {                                                                                     \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false);           \
        v = MAX ? (o > v ? o : v) : (o < v ? o : v);                                      \
    }
    SR_DP_STEP(0x111)  // row_shr:1
    SR_DP_STEP(0x112)  // row_shr:2
    SR_DP_STEP(0x114)  // row_shr:4
    SR_DP_STEP(0x118)  // row_shr:8
#undef SR_DP_STEP
    const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31), c = __builtin_amdgcn_readlane(v, 47),
              d = __builtin_amdgcn_readlane(v, 63);
    const int ab = MAX ? (a > b ? a : b) : (a < b ? a : b), cd = MAX ? (c > d ? c : d) : (c < d ? c : d);
    return MAX ? (ab > cd ? ab : cd) : (ab < cd ? ab : cd);
}
__device__ __forceinline__ int wave_min_i32(int v) { return wave_reduce_i32<false>(v); }
__device__ __forceinline__ int wave_max_i32(int v) { return wave_reduce_i32<true>(v); }

template <int G, int kDpWaves>  // lanes per pair, waves per workgroup
__global__ void __launch_bounds__(64 * kDpWaves) k_dtw_dp_band(const DpBandArgs a)
{
    constexpr int NG = 64 / G;  // pairs per wave
    extern __shared__ __attribute__((aligned(16))) u32x4 dp_smem[];
    const uint32_t K = a.d.K, ks = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = (int)(lane % G);
    const uint32_t grp = w * NG + lane / G;
    const uint32_t RP = a.rows_pad;
    // LDS: template image [RP] rows of 32 bytes, row r at index r + G (G pad rows in front: lanes j > 0 start above the
    // band), then one boundary column of RP words per group, entry r at index r + G
    u32x4 *s_tpl = dp_smem;
    uint32_t *s_col = (uint32_t *)(dp_smem + 2 * (size_t)RP) + (size_t)grp * RP;
    const int mdl_n = (int)a.tpl_frames_s[ks];
    {
        const uint32_t rows = a.d.tpl_rows;
        for (uint32_t r = tid; r < rows; r += blockDim.x) {
            const u32x4 *q = a.tplR + ((size_t)r * K + ks) * 2;
            s_tpl[2 * (r + G)] = q[0];
            s_tpl[2 * (r + G) + 1] = q[1];
        }
        // D(0,0) = 0 is the virtual predecessor of cell (1,1): entry -1 of the boundary column; everything else unreachable
        for (uint32_t i = (uint32_t)j; i < RP; i += G) s_col[i] = (i == (uint32_t)(G - 1)) ? 0u : kDpInf;
    }
    __syncthreads();
    const uint32_t b = (blockIdx.y * kDpWaves + w) * NG + lane / G;
    int in_n = 0;
    if (b < a.d.B) {
        if (a.d.in_frames) in_n = (int)a.d.in_frames[b];
        else in_n = (a.d.vad[b].status == SR_ST_OK) ? (int)a.d.vad[b].frm_num : 0;
    }
    const bool pair_ok = in_n > 0 && mdl_n > 0 && !(in_n > 2 * mdl_n || 2 * in_n < mdl_n);  // main.c:283, DTW.C:133-137
    const int in_eff = pair_ok ? in_n : 0;
    const int X1 = ((2 * mdl_n - in_n) / 3) & 0xFFFF, X2 = ((4 * in_n - 2 * mdl_n) / 3) & 0xFFFF;  // DTW.C:141-142
    const int c1s2 = 5 - (in_n - 2 * mdl_n), c2s = (mdl_n - 2 * in_n) - 3;
    // interval form of dtw_limit for column x (1-based): rows lb..ub (1-based) are inside, clamped to the template
    auto band = [&](int x, int &lo0, int &hi0) {  // 0-based row range, empty when lo0 > hi0
        const int ub1 = (x < X1) ? 2 * x + 2 : ((x + c1s2) >> 1);
        const int lb = (x < X2) ? (x >> 1) : 2 * x + c2s;
        lo0 = (lb < 1 ? 1 : lb) - 1;
        hi0 = (ub1 - 1 > mdl_n ? mdl_n : ub1 - 1) - 1;
        if (x > in_eff) {
            lo0 = 1;
            hi0 = 0;
        }
    };
    const int n_strips = __builtin_amdgcn_readfirstlane(wave_max_i32((in_eff + G - 1) / G));
    const int16_t *in = a.d.mfcc + (size_t)(b < a.d.B ? b : 0) * a.d.max_frames * kCoef;
    const int s_fin = in_eff ? (in_eff - 1) / G : -1, j_fin = in_eff ? (in_eff - 1) % G : -1;
    uint32_t fin = kDpInf;
    bool end_in_band = false;

    auto load_col = [&](int x) {  // the lane's utterance frame for column x (clamped to the record)
        const int row = x <= in_eff ? x - 1 : 0;
        const uint2 *src = (const uint2 *)(in + (size_t)row * kCoef);
        const uint2 q0 = src[0], q1 = src[1], q2 = src[2];
        Row32 f = row_from2(u32x2{q0.x, q0.y}, u32x2{q1.x, q1.y}, u32x2{q2.x, q2.y}, 0u);
        return f;
    };
    Row32 nxt = load_col(j + 1);
    for (int s = 0; s < n_strips; s++) {
        const int x = s * G + j + 1;
        Row32 fi = nxt;
        fi.w[6] = (uint32_t)dot_rows(fi, fi);
        if (s + 1 < n_strips) nxt = load_col(x + G);
        int rlo, rhi, nlo, nhi;
        band(x, rlo, rhi);
        band((s + 1) * G + 1, nlo, nhi);  // first column of the next strip (group-uniform)
        const bool has = rlo <= rhi;
        int tlo_l = has ? rlo + j : 0x7FFFFFFF, thi_l = has ? rhi + j : -1;
        if (nlo <= nhi) {  // the boundary column this strip leaves behind must be right on rows nlo-1 .. nhi
            const int e0 = nlo - 1 + (G - 1), e1 = nhi + (G - 1);
            tlo_l = e0 < tlo_l ? e0 : tlo_l;
            thi_l = e1 > thi_l ? e1 : thi_l;
        }
        const int t_lo = __builtin_amdgcn_readfirstlane(wave_min_i32(tlo_l));
        const int t_hi = __builtin_amdgcn_readfirstlane(wave_max_i32(thi_l));
        if (t_lo > t_hi) continue;
        if (s == s_fin && j == j_fin) end_in_band = has && rhi == mdl_n - 1;
        const uint32_t span = has ? (uint32_t)(rhi - rlo + 1) : 0u;
        uint32_t rel = (uint32_t)(t_lo - j - rlo);
        uint32_t up = kDpInf;
        uint32_t diag = (j == 0) ? s_col[t_lo - 1 + G] : kDpInf;
        const u32x4 *tp = s_tpl + 2 * (t_lo - j + G);
        const uint32_t *crd = s_col + (t_lo + G);
        uint32_t *cwr = s_col + (t_lo - (G - 1) + G);
        const bool fin_strip = __builtin_amdgcn_ballot_w64(s == s_fin) != 0ull;
        // Two steps per trip, the template rows / boundary entries of the NEXT step requested before the current one is
        // evaluated (two register sets, no copies).  An odd step count is rounded up: the extra step lies above every
        // lane's band (t_hi + 1), produces "unreachable" everywhere and touches only the pad rows of the LDS images.
        const int t_end = t_hi + ((t_hi - t_lo + 1) & 1);
        auto run = [&](auto fin_tag) {
            constexpr bool FIN = decltype(fin_tag)::value;
            // squared distance of the lane's utterance frame to a template row: |m|^2 + |i|^2 + (-2m).i
            auto dist2 = [&](const u32x4 &lo, const u32x4 &hi) {
                const Row32 fm = row_from(lo, hi);
                return (uint32_t)dot_rows_acc(fm, fi, (int)(fm.w[6] + fi.w[6]));
            };
            auto cell = [&](uint32_t d, uint32_t scv, uint32_t rel_now, uint32_t *wr) {
                // D(x-1, r): the neighbour lane's result of the previous step; lane 0 of the group: the boundary column
                uint32_t fl;
                if (G == 16) {  // a DPP row IS a group: the lane without a source keeps `old` = the boundary entry
                    fl = (uint32_t)__builtin_amdgcn_update_dpp((int)scv, (int)up, 0x111, 0xF, 0xF, false);  // row_shr:1
                } else {
                    fl = (uint32_t)__builtin_amdgcn_mov_dpp((int)up, 0x111, 0xF, 0xF, true);
                    fl = (j == 0) ? scv : fl;
                }
                uint32_t best = diag < fl ? diag : fl;
                best = up < best ? up : best;
                const uint32_t sum = add_sat(best, d);
                const bool inb = rel_now < span;
                const uint32_t cur = inb ? sum : kDpInf;
                if (FIN) fin = inb ? sum : fin;
                diag = fl;
                up = cur;
                if (j == G - 1) *wr = cur;
            };
            u32x4 a0 = tp[0], a1 = tp[1], b0 = tp[2], b1 = tp[3];
            uint32_t sa = crd[0], sb = crd[1];
            for (int t = t_lo; t <= t_end; t += 2) {
                // the local distances do not depend on the recurrence: both steps' roots are taken together, so that
                // the multiplies and fused corrections of get_dis' sqrtf (DTW.C:59) are packed f32 operations
                const f32x2 rt = sqrt_rn_int2(f32x2{(float)dist2(a0, a1), (float)dist2(b0, b1)});
                const uint32_t d_a = cvt_u32(rt.x), d_b = cvt_u32(rt.y);
                const uint32_t s0 = sa, s1 = sb;
                a0 = tp[4];
                a1 = tp[5];
                b0 = tp[6];
                b1 = tp[7];
                sa = crd[2];
                sb = crd[3];
                cell(d_a, s0, rel, cwr);
                cell(d_b, s1, rel + 1, cwr + 1);
                tp += 4;
                crd += 2;
                cwr += 2;
                rel += 2;
            }
        };
        if (fin_strip) run(std::true_type{});
        else run(std::false_type{});
        wave_sync();
    }
    if (b < a.d.B) {
        uint32_t *out = a.d.scores + (size_t)b * K + a.tpl_orig[ks];
        if (!pair_ok) {
            if (j == 0) *out = SR_DIS_ERR;
        } else if (j == j_fin) {
            *out = (end_in_band && fin != kDpInf) ? fin / (uint32_t)(in_n + mdl_n) : SR_DIS_ERR;
        }
    }
}

// LDS of a band-kernel workgroup: the template image + one boundary column per group, both rows_pad long.  Reads may
// run up to five rows / entries past the last step (the rounded-up step and the rows requested ahead): they land in the
// next region of the image (values never used) and, for the last group, in the 32 spare bytes at the end.
static uint32_t dp_band_lds(uint32_t tpl_rows, int G, int waves, uint32_t *rows_pad)
{
    const uint32_t rp = (uint32_t)G + (tpl_rows - 1) + (uint32_t)G - 1;
    if (rows_pad) *rows_pad = rp;
    return rp * 32u + (uint32_t)(waves * (64 / G)) * rp * 4u + 32u;
}
// waves per workgroup.  Two-wave workgroups were measured and dropped (G = 4: 177 ms against 120 with four waves although
// three of them fit a CU instead of one: their waves pile up on two of the four SIMDs)
static int dp_waves_for(int G) { (void)G; return 4; }

void launch_dtw_dp(const DtwArgs &a, hipStream_t s)
{
    if (!a.B || !a.K) return;
    int G = (int)a.dp_lanes;  // 0 = choose
    uint32_t rp = 0;
    if (G != 1 && G != 4 && G != 8 && G != 16) {
        // 8 lanes per pair while three of its workgroups fit a CU's 160 KiB (handed out in granules of 1280 bytes): 111 ms
        // per 65 536 x 100 pairs at a 320-frame cap; one granule more and only two fit (153 ms) -- then 16 lanes per pair
        // (half the boundary columns, 120 ms with three or four workgroups per CU) is the better shape
        auto wgs = [&](int g) { return (160u * 1024u) / ((dp_band_lds(a.tpl_rows, g, dp_waves_for(g), nullptr) + 1279u) / 1280u * 1280u); };
        G = (wgs(8) >= 3 || wgs(16) < 3) ? 8 : 16;
    }
    if (G != 1 && (!a.tplR || dp_band_lds(a.tpl_rows, G, dp_waves_for(G), &rp) > 150u * 1024u)) {
        G = 16;  // fewer pairs per wave: fewer boundary columns
        if (!a.tplR || dp_band_lds(a.tpl_rows, G, dp_waves_for(G), &rp) > 150u * 1024u) G = 1;
    }
    if (G == 1) {
        const size_t lds = (size_t)a.tpl_rows * 32 + (size_t)4 * a.tpl_rows * 4;
        for (uint32_t b0 = 0; b0 < a.B; b0 += 65535u * 4u) {  // grid.y <= 65 535
            DtwArgs sa = a;
            sa.B = (a.B - b0 < 65535u * 4u) ? a.B - b0 : 65535u * 4u;
            sa.mfcc = a.mfcc + (size_t)b0 * a.max_frames * kCoef;
            sa.vad = a.vad ? a.vad + b0 : nullptr;
            sa.in_frames = a.in_frames ? a.in_frames + b0 : nullptr;
            sa.scores = a.scores + (size_t)b0 * a.K;
            hipLaunchKernelGGL(k_dtw_dp_wave64, dim3(a.K, (sa.B + 3) / 4), dim3(256), lds, s, sa);
        }
        return;
    }
    const int W = dp_waves_for(G);
    const size_t lds = dp_band_lds(a.tpl_rows, G, W, &rp);
    const uint32_t per_wg = (uint32_t)(W * (64 / G));
    // the utterance blocks are the grid's second dimension (<= 65 535): larger batches go out in slices
    const uint32_t slice = 65535u * per_wg;
    for (uint32_t b0 = 0; b0 < a.B; b0 += slice) {
        DtwArgs sa = a;
        sa.B = (a.B - b0 < slice) ? a.B - b0 : slice;
        sa.mfcc = a.mfcc + (size_t)b0 * a.max_frames * kCoef;
        sa.vad = a.vad ? a.vad + b0 : nullptr;
        sa.in_frames = a.in_frames ? a.in_frames + b0 : nullptr;
        sa.scores = a.scores + (size_t)b0 * a.K;
        DpBandArgs ba{sa, (const u32x4 *)a.tplR, a.tpl_frames_s, a.tpl_orig, rp};
        const dim3 grid(a.K, (sa.B + per_wg - 1) / per_wg), block(64 * W);
        if (G == 4) hipLaunchKernelGGL((k_dtw_dp_band<4, 4>), grid, block, lds, s, ba);
        else if (G == 8) hipLaunchKernelGGL((k_dtw_dp_band<8, 4>), grid, block, lds, s, ba);
        else hipLaunchKernelGGL((k_dtw_dp_band<16, 4>), grid, block, lds, s, ba);
    }
}

}  // namespace sr
