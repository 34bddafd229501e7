// k_dtw.hip -- dtw / get_dis / dtw_limit (DTW.C:45-192) for every (utterance, template) pair: the generic k_dtw, the LDS-staged production kernel k_dtw_lds, get_mdl (DTW.C:217-296) and the template scan of spch_recg (main.c:276-295).
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#include "sr_dtw_dev.h"

namespace sr {

// ------------------------------------------------------------------------------------------------
// k_dtw: one lane per (utterance, template) pair, greedy local walk of DTW.C:120-192
// ------------------------------------------------------------------------------------------------
__device__ uint32_t dtw_pair(const int16_t *in, uint32_t in_n, uint32_t in_rows, const int16_t *mdl, uint32_t mdl_n,
                             uint32_t mdl_rows)
{
    if (in_n > mdl_n * 2 || 2 * in_n < mdl_n) return SR_DIS_ERR;  // DTW.C:133-137
    const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142 (u16 statics)
    const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
    uint32_t px = 0, py = 0;  // 0-based rows under the in / mdl pointers (x = px+1, y = py+1)
    Frame12 ci = load_frame(in), cm = load_frame(mdl);
    uint32_t nci = norm2(ci), ncm = norm2(cm);
    uint32_t dis = get_dis_dev(ci, nci, cm, ncm);
    uint32_t step = 1;
    do {
        // rows px+1 / py+1 are read even when they lie past the sequence end (do-while, DTW.C:150-154);
        // clamped to the allocated rows so the access stays inside the buffer
        const uint32_t rx = (px + 1 < in_rows) ? px + 1 : in_rows - 1, ry = (py + 1 < mdl_rows) ? py + 1 : mdl_rows - 1;
        const Frame12 ni = load_frame(in + (size_t)rx * kCoef), nm = load_frame(mdl + (size_t)ry * kCoef);
        const uint32_t nni = norm2(ni), nnm = norm2(nm);
        const int x = (int)px + 1, y = (int)py + 1;
        const uint32_t up = dtw_out(x, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ci, nci);
        const uint32_t right = dtw_out(x + 1, y, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(cm, ncm, ni, nni);
        const uint32_t diag =
            dtw_out(x + 1, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ni, nni);
        uint32_t mn = diag;  // DTW.C:156-164
        if (mn > right) mn = right;
        if (mn > up) mn = up;
        dis += mn;
        const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:168-184
        const bool adv_x = mv_diag || !mv_up, adv_y = mv_diag || mv_up;
        if (adv_x) {
            ci = ni;
            nci = nni;
            px++;
        }
        if (adv_y) {
            cm = nm;
            ncm = nnm;
            py++;
        }
        step = (step + 1) & 0xFFFF;  // u16 step
    } while (px + 1 < in_n && py + 1 < mdl_n);  // DTW.C:188
    return dis / step;
}

// ---- get_mdl (DTW.C:217-296) + get_mean (DTW.C:195-205): template averaging, one lane per pair --------------
// The same greedy walk as dtw_pair with in1 in the "in" role and in2 in the "mdl" role; the start point and every
// point the walk moves to contribute one merged frame = per-coefficient (a + b) / 2 in int arithmetic (truncation
// toward zero).  The merged template has `step` frames; frames >= out_rows are dropped (the reference would write
// past its 119-frame record there).
__device__ __forceinline__ uint32_t mean_word(uint32_t a, uint32_t b)
{
    const int lo = (sext_lo(a) + sext_lo(b)) / 2, hi = (sext_hi(a) + sext_hi(b)) / 2;
    return pack16(lo, hi);
}
__device__ __forceinline__ void store_mean(int16_t *row, const Frame12 &a, const Frame12 &b)
{
    uint2 *q = (uint2 *)row;
    q[0] = make_uint2(mean_word(a.w[0], b.w[0]), mean_word(a.w[1], b.w[1]));
    q[1] = make_uint2(mean_word(a.w[2], b.w[2]), mean_word(a.w[3], b.w[3]));
    q[2] = make_uint2(mean_word(a.w[4], b.w[4]), mean_word(a.w[5], b.w[5]));
}

__global__ void __launch_bounds__(64) k_get_mdl(const GetMdlArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.P) return;
    const uint32_t in_n = a.n1[p], mdl_n = a.n2[p];
    const int16_t *in = a.in1 + (size_t)p * a.rows1 * kCoef, *mdl = a.in2 + (size_t)p * a.rows2 * kCoef;
    int16_t *out = a.mdl + (size_t)p * a.mdl_rows * kCoef;
    if (in_n == 0 || mdl_n == 0 || in_n > mdl_n * 2 || 2 * in_n < mdl_n) {  // DTW.C:236-239
        a.dis[p] = SR_DIS_ERR;
        a.mdl_frames[p] = 0;
        return;
    }
    const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);
    const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
    uint32_t px = 0, py = 0;
    Frame12 ci = load_frame(in), cm = load_frame(mdl);
    uint32_t nci = norm2(ci), ncm = norm2(cm);
    uint32_t dis = get_dis_dev(ci, nci, cm, ncm);
    if (a.mdl_rows) store_mean(out, ci, cm);  // DTW.C:250-251
    uint32_t step = 1;
    do {
        const uint32_t rx = (px + 1 < a.rows1) ? px + 1 : a.rows1 - 1, ry = (py + 1 < a.rows2) ? py + 1 : a.rows2 - 1;
        const Frame12 ni = load_frame(in + (size_t)rx * kCoef), nm = load_frame(mdl + (size_t)ry * kCoef);
        const uint32_t nni = norm2(ni), nnm = norm2(nm);
        const int x = (int)px + 1, y = (int)py + 1;
        const uint32_t up = dtw_out(x, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ci, nci);
        const uint32_t right = dtw_out(x + 1, y, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(cm, ncm, ni, nni);
        const uint32_t diag =
            dtw_out(x + 1, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_dev(nm, nnm, ni, nni);
        uint32_t mn = diag;  // DTW.C:260-268
        if (mn > right) mn = right;
        if (mn > up) mn = up;
        dis += mn;
        const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:272-288
        if (mv_diag || !mv_up) {
            ci = ni;
            nci = nni;
            px++;
        }
        if (mv_diag || mv_up) {
            cm = nm;
            ncm = nnm;
            py++;
        }
        if (step < a.mdl_rows) store_mean(out + (size_t)step * kCoef, ci, cm);  // DTW.C:286-287 (row = step before ++)
        step = (step + 1) & 0xFFFF;
    } while (px + 1 < in_n && py + 1 < mdl_n);  // DTW.C:291
    a.mdl_frames[p] = step;  // DTW.C:293
    a.dis[p] = dis / step;
}
void launch_get_mdl(const GetMdlArgs &a, hipStream_t s)
{
    if (!a.P) return;
    hipLaunchKernelGGL(k_get_mdl, dim3((a.P + 63) / 64), dim3(64), 0, s, a);
}

__global__ void __launch_bounds__(128) k_dtw(const DtwArgs a)
{
    const uint64_t pid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= (uint64_t)a.B * a.K) return;
    const uint32_t b = (uint32_t)(pid / a.K), k = (uint32_t)(pid - (uint64_t)b * a.K);
    uint32_t in_n, ok;
    if (a.in_frames) {
        in_n = a.in_frames[b];
        ok = in_n != 0;
    } else {
        in_n = a.vad[b].frm_num;
        ok = a.vad[b].status == SR_ST_OK && in_n != 0;
    }
    uint32_t d = SR_DIS_ERR;
    if (ok && a.tpl_valid[k])  // main.c:283
        d = dtw_pair(a.mfcc + (size_t)b * a.max_frames * kCoef, in_n, a.max_frames, a.tpl + (size_t)k * a.tpl_stride,
                     a.tpl_frames[k], a.tpl_rows);
    a.scores[pid] = d;
}

// ---- k_dtw_gen: the same walk for feature rows of any width (GENERIC front end, n_coef != 12, stores that cannot be staged) ----
// One lane per pair.  A row of n_coef <= 16 coefficients lives in eight registers as packed pairs (zero-padded: the pad
// contributes nothing to get_dis' sum of squares, DTW.C:45-62) with its squared norm; rows are fetched coefficient by
// coefficient (rows of an odd number of s16 are only 2-byte aligned) when the walk advances, and a distance is the norm
// sum minus twice eight v_dot2_i32_i16 -- the arithmetic of dtw_pair above in the same u32 ring.  Stores with 12
// coefficients never come here (k_dtw_lds / k_dtw); other widths only when k_dtw_lds cannot stage them (full-scale
// coefficients, frame caps beyond the LDS).
struct Frame16 {
    uint32_t w[8];
    uint32_t n;
};
__device__ __forceinline__ Frame16 load_frame_n(const int16_t *p, uint32_t nc)
{
    Frame16 f;
    int acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t lo = (2 * i < nc) ? (uint32_t)(uint16_t)p[2 * i] : 0u, hi = (2 * i + 1 < nc) ? (uint32_t)(uint16_t)p[2 * i + 1] : 0u;
        f.w[i] = lo | (hi << 16);
        acc = sdot2(f.w[i], f.w[i], acc);
    }
    f.n = (uint32_t)acc;
    return f;
}
__device__ __forceinline__ uint32_t get_dis_n(const Frame16 &a, const Frame16 &b)
{
    int dot = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) dot = sdot2(a.w[i], b.w[i], dot);
    return cvt_u32(sqrt_rn_int((float)(a.n + b.n - 2u * (uint32_t)dot)));
}
__global__ void __launch_bounds__(128) k_dtw_gen(const DtwArgs a)
{
    const uint64_t pid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= (uint64_t)a.B * a.K) return;
    // (template major, utterance minor: the lanes of a wave walk one template, so its rows are fetched once per wave)
    const uint32_t k = (uint32_t)(pid / a.B), b = (uint32_t)(pid - (uint64_t)k * a.B), nc = a.n_coef;
    uint32_t in_n, ok;
    if (a.in_frames) {
        in_n = a.in_frames[b];
        ok = in_n != 0;
    } else {
        in_n = a.vad[b].frm_num;
        ok = a.vad[b].status == SR_ST_OK && in_n != 0;
    }
    uint32_t score = SR_DIS_ERR;
    const uint32_t mdl_n = a.tpl_frames[k];
    if (ok && a.tpl_valid[k] && !(in_n > mdl_n * 2 || 2 * in_n < mdl_n)) {  // main.c:283, DTW.C:133-137
        const int16_t *in = a.mfcc + (size_t)b * a.max_frames * nc, *mdl = a.tpl + (size_t)k * a.tpl_stride;
        const uint32_t in_rows = a.max_frames, mdl_rows = a.tpl_rows;
        const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142
        const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
        uint32_t px = 0, py = 0, step = 1;
        Frame16 ci = load_frame_n(in, nc), cm = load_frame_n(mdl, nc);
        uint32_t dis = get_dis_n(ci, cm);
        // rows px+1 / py+1 are read even past the sequence end (do-while, DTW.C:150-154), clamped to the allocation
        Frame16 ni = load_frame_n(in + (size_t)(1 < in_rows ? 1 : in_rows - 1) * nc, nc);
        Frame16 nm = load_frame_n(mdl + (size_t)(1 < mdl_rows ? 1 : mdl_rows - 1) * nc, nc);
        do {
            const int x = (int)px + 1, y = (int)py + 1;
            const uint32_t up = dtw_out(x, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_n(nm, ci);
            const uint32_t right = dtw_out(x + 1, y, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_n(cm, ni);
            const uint32_t diag = dtw_out(x + 1, y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_DIS_ERR : get_dis_n(nm, ni);
            uint32_t mn = diag;  // DTW.C:156-164
            if (mn > right) mn = right;
            if (mn > up) mn = up;
            dis += mn;
            const bool mv_diag = (mn == diag), mv_up = !mv_diag && (mn == up);  // DTW.C:168-184
            if (mv_diag || !mv_up) {
                px++;
                ci = ni;
                const uint32_t rx = (px + 1 < in_rows) ? px + 1 : in_rows - 1;
                ni = load_frame_n(in + (size_t)rx * nc, nc);
            }
            if (mv_diag || mv_up) {
                py++;
                cm = nm;
                const uint32_t ry = (py + 1 < mdl_rows) ? py + 1 : mdl_rows - 1;
                nm = load_frame_n(mdl + (size_t)ry * nc, nc);
            }
            step = (step + 1) & 0xFFFF;
        } while (px + 1 < in_n && py + 1 < mdl_n);  // DTW.C:188
        score = dis / step;
    }
    a.scores[(size_t)b * a.K + k] = score;
}

// ---- k_dtw_lds: the production DTW kernel ---------------------------------------------------------
// A workgroup owns U utterances whose MFCC rows (+ squared norms) are staged in LDS once and reused by
// all K templates; lanes are the U*K pairs ordered (template-sorted-by-length major, utterance minor),
// so the lanes of a wave walk sequences of nearly equal length (little trip-count divergence) and
// touch at most ceil(64/U) templates.  Templates live in HBM/L2 in a row-interleaved layout
// tplR[row][ks] (ks = rank of the template by length) of 32-byte rows: 12 x s16 holding -2*coefficient, the
// u32 squared norm of the row, pad -- so lanes that advance in step read neighbouring addresses and a squared
// distance |m|^2 + |in|^2 - 2 m.in is the norm sum fed through six accumulating dot products.
// One root per step: the root of the smallest admissible squared candidate; the reference's tie order is decided on
// the squared values against (root+1)^2 -/+ a proven margin, with the literal three-root form as wave-uniform
// fallback (see the loop).  The result is bit-identical to dtw_pair above.

struct DtwLdsArgs {
    DtwArgs d;
    const u32x4 *tplR;          // [tpl_rows][K] 32-byte rows: 12 x s16 holding -2*coef | u32 squared norm | pad ; length order
    const uint32_t *tpl_frames_s;  // [K] frames, sorted order; 0 for invalid slots
    const uint32_t *tpl_orig;   // [K] original slot of sorted position
    uint32_t U;                 // utterances per workgroup
    const int8_t *tie_delta;    // [tie_g] tie-threshold table (sr_tables.h), staged at the start of the dynamic LDS
    uint32_t tie_g;             // entries staged: roots >= tie_g take the literal path
    uint32_t Kc;                // templates (ranks) per workgroup: blockIdx.y selects the chunk [y*Kc, (y+1)*Kc) of the K ranks
};

constexpr int kDtwMaxU = 16;
// words between utterances in the LDS image: rows are 6 words; the stride is the next value == 22 (mod 64)
// so that equal rows of different utterances do not alias (64 banks for 8-byte reads); norms: == 11 (mod 32)
__host__ __device__ inline uint32_t dtw_lds_row_stride(uint32_t R, uint32_t row_words = 6)
{
    uint32_t s = R * row_words;
    return s + ((22 + 64 - (s & 63)) & 63);
}
__host__ __device__ inline uint32_t dtw_lds_nrm_stride(uint32_t R) { return R + ((11 + 32 - (R & 31)) & 31); }

// rows r and r+1 of an utterance's LDS image (24-byte rows, squared norms in a separate array) into registers.
// The six 8-byte reads are volatile so that they stay ds_read_b64 (2 LDS cycles each): merged into ds_read2_b64 they
// cost 8 cycles a pair (MI355X_MICROARCH.md, LDS table).
typedef __attribute__((address_space(3))) const volatile u32x2 lds_cv_u32x2;
typedef __attribute__((address_space(3))) const uint32_t lds_c_u32;
typedef __attribute__((address_space(3))) const int8_t lds_c_i8;
// LDS byte offset of a pointer into the workgroup's shared memory
__device__ __forceinline__ uint32_t lds_offset(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// rows r and r+1 at byte offset row_off (24-byte rows) and their squared norms at nrm_off
__device__ __forceinline__ void lds_rows2(uint32_t row_off, uint32_t nrm_off, Row32 &r0, Row32 &r1)
{
    lds_cv_u32x2 *q = (lds_cv_u32x2 *)(uintptr_t)row_off;
    lds_c_u32 *np = (lds_c_u32 *)(uintptr_t)nrm_off;
    const u32x2 a0 = q[0], a1 = q[1], a2 = q[2], b0 = q[3], b1 = q[4], b2 = q[5];
    r0 = row_from2(a0, a1, a2, np[0]);
    r1 = row_from2(b0, b1, b2, np[1]);
}
// ---- the row formats of k_dtw_lds: 12 coefficients (the reference, and narrower GENERIC rows zero-padded) or 16 (GENERIC
// front end, n_coef 13..16, zero-padded).  LDS image of an utterance: kWords packed pairs per row + a separate array of norms;
// store rows in HBM / L2: -2*coef in kWords words, the squared norm in the next word, padded to kTplBytes.
struct Dtw12 {
    typedef Row32 Row;  // w[0..5] coefficients, w[6] norm
    static constexpr uint32_t kWords = 6, kLdsRowBytes = 24, kTplBytes = 32;
    static __device__ __forceinline__ uint32_t nrm(const Row &r) { return r.w[6]; }
    static __device__ __forceinline__ Row tpl_row(const char *p)
    {
        const u32x4 *q = (const u32x4 *)p;
        return row_from(q[0], q[1]);  // (a 16 + 12 byte pair of loads, skipping the pad word, is slower: 6.44 -> 6.67 ms)
    }
    static __device__ __forceinline__ void rows2(uint32_t row_off, uint32_t nrm_off, Row &r0, Row &r1) { lds_rows2(row_off, nrm_off, r0, r1); }
    static __device__ __forceinline__ int dot_acc(const Row &a, const Row &b, int c) { return dot_rows_acc(a, b, c); }
    static __device__ __forceinline__ void copy(Row &d, const Row &s) { copy_row(d, s); }
};
struct Row48 {
    uint32_t w[10];  // w[0..7] coefficients, w[8] norm, w[9] unused
};
struct Dtw16 {
    typedef Row48 Row;
    static constexpr uint32_t kWords = 8, kLdsRowBytes = 32, kTplBytes = 48;
    static __device__ __forceinline__ uint32_t nrm(const Row &r) { return r.w[8]; }
    static __device__ __forceinline__ Row tpl_row(const char *p)
    {
        const u32x4 *q = (const u32x4 *)p;
        const u32x4 a = q[0], b = q[1], c = q[2];
        Row r;
        r.w[0] = a.x, r.w[1] = a.y, r.w[2] = a.z, r.w[3] = a.w, r.w[4] = b.x, r.w[5] = b.y, r.w[6] = b.z, r.w[7] = b.w;
        r.w[8] = c.x, r.w[9] = c.y;
        return r;
    }
    static __device__ __forceinline__ void rows2(uint32_t row_off, uint32_t nrm_off, Row &r0, Row &r1)
    {
        lds_cv_u32x2 *q = (lds_cv_u32x2 *)(uintptr_t)row_off;
        lds_c_u32 *np = (lds_c_u32 *)(uintptr_t)nrm_off;
        const u32x2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], b0 = q[4], b1 = q[5], b2 = q[6], b3 = q[7];
        r0.w[0] = a0.x, r0.w[1] = a0.y, r0.w[2] = a1.x, r0.w[3] = a1.y, r0.w[4] = a2.x, r0.w[5] = a2.y, r0.w[6] = a3.x, r0.w[7] = a3.y;
        r1.w[0] = b0.x, r1.w[1] = b0.y, r1.w[2] = b1.x, r1.w[3] = b1.y, r1.w[4] = b2.x, r1.w[5] = b2.y, r1.w[6] = b3.x, r1.w[7] = b3.y;
        r0.w[8] = np[0], r0.w[9] = 0;
        r1.w[8] = np[1], r1.w[9] = 0;
    }
    static __device__ __forceinline__ int dot_acc(const Row &a, const Row &b, int c)
    {
        int acc = sdot2a(a.w[0], b.w[0], c);
#pragma unroll
        for (int i = 1; i < 8; i++) acc = sdot2(a.w[i], b.w[i], acc);
        return acc;
    }
    static __device__ __forceinline__ void copy(Row &d, const Row &s)
    {
#pragma unroll
        for (int i = 0; i < 10; i += 2) {
            u32x2 t;
            const u32x2 v = {s.w[i], s.w[i + 1]};
            asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(t) : "v"(v));
            d.w[i] = t.x;
            d.w[i + 1] = t.y;
        }
    }
};
#ifdef SR_DTW_STATS
// development build only: wave-steps in total / on the literal path, by reason (bracket, table range, lost lane)
__device__ unsigned long long g_dtw_stats[8];
extern "C" void sr_debug_dtw_stats(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dtw_stats), sizeof(g_dtw_stats));
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dtw_stats), z, sizeof(z));
    }
}
#endif
template <class F>  // F = Dtw12 / Dtw16: the row format
__global__ void __launch_bounds__(1024) k_dtw_lds(const DtwLdsArgs a)
{
    typedef typename F::Row Row;
    extern __shared__ __attribute__((aligned(16))) u32x2 smem2[];  // 8-byte typed: rows are read as ds_read_b64
    const uint32_t U = a.U, R = a.d.max_frames, K = a.d.K;
    // LDS (all of it dynamic): [tie-threshold table, tie_g bytes][rows][norms][frame counts].  The table comes FIRST, at
    // LDS address 0 (the kernel has no static LDS), so that a look-up is one ds_read_i8 whose address register is the
    // root itself -- no VALU address arithmetic.
    // Image of the utterances: 24-byte rows (3 x 8 bytes) + a separate array of squared norms.  Strides are padded so
    // that lanes sitting on the same row of different utterances fall on different banks.
    u32x2 *s_rows = smem2 + a.tie_g / 8;
    const uint32_t row_stride = dtw_lds_row_stride(R, F::kWords), nrm_stride = dtw_lds_nrm_stride(R);  // words
    uint32_t *s_nrm = (uint32_t *)(s_rows + (size_t)U * (row_stride / 2));
    uint32_t *s_n = s_nrm + (size_t)U * nrm_stride + 8;  // frames of the workgroup's utterances, 0 = skip (+8 words: reload slack)
    const uint32_t tid = threadIdx.x, b0 = blockIdx.x * U;
    for (uint32_t i = tid; i < a.tie_g / 16; i += blockDim.x) ((u32x4 *)smem2)[i] = ((const u32x4 *)a.tie_delta)[i];

    if (tid < U) {
        const uint32_t b = b0 + tid;
        uint32_t n = 0;
        if (b < a.d.B) {
            if (a.d.in_frames) n = a.d.in_frames[b];
            else n = (a.d.vad[b].status == SR_ST_OK) ? a.d.vad[b].frm_num : 0u;
        }
        s_n[tid] = n;
    }
    __syncthreads();
    // ---- stage rows [0, min(n+1, R)) of every utterance and their squared norms ----
    for (uint32_t u = 0; u < U; u++) {
        const uint32_t n = s_n[u];
        if (!n) continue;
        const uint32_t rows = (n + 1 < R) ? n + 1 : R;
        const uint32_t nc = a.d.n_coef;  // F::kWords * 2, or fewer on the GENERIC front end: the LDS image is zero-padded
        const uint2 *src = (const uint2 *)(a.d.mfcc + (size_t)(b0 + u) * R * kCoef);
        u32x2 *dst = s_rows + (size_t)u * (row_stride / 2);
        for (uint32_t r = tid; r < rows; r += blockDim.x) {
            uint32_t w[F::kWords];
            if (F::kWords == 6 && nc == (uint32_t)kCoef) {
                const uint2 q0 = src[3 * r], q1 = src[3 * r + 1], q2 = src[3 * r + 2];
                w[0] = q0.x, w[1] = q0.y, w[2] = q1.x, w[3] = q1.y, w[4] = q2.x, w[5] = q2.y;
            } else {  // other widths (2-byte aligned when nc is odd): coefficient by coefficient; zero padding adds nothing
                      // to get_dis' sum of squares (DTW.C:45-62)
                const int16_t *p = a.d.mfcc + ((size_t)(b0 + u) * R + r) * nc;
#pragma unroll
                for (uint32_t i = 0; i < F::kWords; i++) {
                    const uint32_t lo = (2 * i < nc) ? (uint32_t)(uint16_t)p[2 * i] : 0u, hi = (2 * i + 1 < nc) ? (uint32_t)(uint16_t)p[2 * i + 1] : 0u;
                    w[i] = lo | (hi << 16);
                }
            }
            int nr = sdot2z(w[0], w[0]);
#pragma unroll
            for (uint32_t i = 1; i < F::kWords; i++) nr = sdot2(w[i], w[i], nr);
#pragma unroll
            for (uint32_t i = 0; i < F::kWords / 2; i++) dst[(F::kWords / 2) * r + i] = u32x2{w[2 * i], w[2 * i + 1]};
            s_nrm[u * nrm_stride + r] = (uint32_t)nr;
        }
    }
    __syncthreads();

    // a store of more than 1024 / U templates is walked in chunks of Kc ranks (grid y): every chunk stages the same U
    // utterances again (6 KB each from L2) and scores them against its slice of the length-sorted store
    if (tid >= U * a.Kc) return;
    // (rank major, utterance minor; the other order -- a wave = consecutive ranks of one utterance -- is slower: 6.52 vs 6.31 ms)
    const uint32_t ks = blockIdx.y * a.Kc + tid / U, u = tid % U, b = b0 + u;
    if (b >= a.d.B || ks >= K) return;
    const uint32_t in_n = s_n[u], mdl_n = a.tpl_frames_s[ks];
    uint32_t score = SR_DIS_ERR;
    if (in_n && mdl_n && !(in_n > mdl_n * 2 || 2 * in_n < mdl_n)) {  // main.c:283, DTW.C:133-137
        const int X1 = (int)(((2 * (int)mdl_n - (int)in_n) / 3) & 0xFFFF);  // DTW.C:141-142
        const int X2 = (int)(((4 * (int)in_n - 2 * (int)mdl_n) / 3) & 0xFFFF);
        const int c1s = 3 - ((int)in_n - 2 * (int)mdl_n), c2s = ((int)mdl_n - 2 * (int)in_n) - 3;
        // cursors: input rows x-1 / x (0-based) live in registers and are re-read from LDS only when x advances;
        // the NEXT template row comes from HBM/L2 when y advances.  Rows x / y always exist inside the loop (x < in_n <= R
        // and y < mdl_n < tpl_rows; for 1-frame sequences row 1 is the slack row the reference's do-while reads,
        // DTW.C:150-154); the reload after the last advance may touch one row past the utterance's image, which the
        // launch pads for.
        uint32_t in_off = lds_offset(s_rows + (size_t)u * (row_stride / 2));  // LDS byte offsets of row x-1 and its norm
        uint32_t nrm_off = lds_offset(s_nrm + (size_t)u * nrm_stride);
        // template rows through a 32-bit byte offset from the (uniform) table base: advancing it is ONE add, and the
        // loads take the base from SGPRs (a 64-bit per-lane pointer costs an add-with-carry pair per advance)
        const char *tbase = (const char *)a.tplR;
        uint32_t t_off = ks * F::kTplBytes;
        const uint32_t t_stride = K * F::kTplBytes;  // bytes per template row level (K * kTplBytes * rows < 2^32: checked at upload)
        auto tpl_row = [&](uint32_t off) { return F::tpl_row(tbase + off); };
        Row cm = tpl_row(t_off);
        t_off += t_stride;
        Row nm = tpl_row(t_off);
        Row ci, ni;
        F::rows2(in_off, nrm_off, ci, ni);
        uint32_t dis = cvt_u32(sqrt_rn_int((float)(uint32_t)F::dot_acc(cm, ci, (int)(F::nrm(cm) + F::nrm(ci)))));  // DTW.C:146
        // dtw_limit (DTW.C:76-109) as an interval test per column: (x', y') is inside  <=>  lb(x') <= y' <= ub(x')
        //   ub(x') = x' < X1 ? 2x'+1 : (x'+3-c1) >> 1      (negation of DTW.C:78-91; >> floors)
        //   lb(x') = x' < X2 ? x' >> 1 : 2x'+c2-3           (negation of DTW.C:93-106)
        // Carried: ub of column x (A), lb and ub of column x+1 (B); one new column is evaluated when x advances.
        // lb(x) is not needed: the walk only ever moves to admissible points, so lb(x) <= y holds for the current point
        // and (x, y+1) can only leave through ub(x).  The one exception -- all three candidates outside, DTW.C:156-184
        // then moves diagonally to an outside point -- makes the lane `lost`: from then on it takes the literal path.
        // What is carried is y1 = y + 1 and the upper bounds PLUS ONE, so that every test is one compare of carried values:
        //   (x, y+1) inside    <=>  y1 <  ubA1                 (ubA1 = ub(x) + 1)
        //   (x+1, y) inside    <=>  lbB <  y1  &&  y1 <= ubB1  (ubB1 = ub(x+1) + 1)
        //   (x+1, y+1) inside  <=>  lbB <= y1  &&  y1 <  ubB1
        const int c1s2 = c1s + 2;
        const uint32_t tie_g1 = a.tie_g - 1;  // roots up to tie_g - 2 may be stepped by one and still find their threshold staged
        auto ub1_of = [&](int xx) { return (xx < X1) ? 2 * xx + 2 : ((xx + c1s2) >> 1); };
        auto lb_of = [&](int xx) { return (xx < X2) ? (xx >> 1) : 2 * xx + c2s; };
        int xB = 2, y1 = 2;  // xB = x + 1, y1 = y + 1; DTW.C:147-148
        int ubA1 = ub1_of(1), lbB = lb_of(2), ubB1 = ub1_of(2);
        uint64_t lost = 0;  // lane mask (kept scalar: OR-ed into the wave-uniform branch condition without touching the VALU)
        // step (DTW.C:149, 186): u16 in the reference; cannot wrap here (steps < in_n + mdl_n <= 2R, R bounded by LDS).  Every lane
        // of the wave enters the loop in the same trip, so a lane's step count is the wave's TRIP count when the lane leaves: the
        // count lives on the scalar unit and a lane picks it up once, in the trip it leaves in (round 6; a per-lane v_add per
        // step before)
        uint32_t trips = 1, step = 1;
        bool cont;
        do {
            // all three candidate squared distances, unconditionally: |m|^2 + |i|^2 + (-2m).i, the norm sum seeds
            // the dot2 accumulator (template rows are stored as -2m, see upload_templates)
            // (x+1, y) first: it needs neither the template row that may still be in flight from the previous step's y advance
            const uint32_t d_rt = (uint32_t)F::dot_acc(cm, ni, (int)(F::nrm(cm) + F::nrm(ni)));  // (x+1, y):   get_dis(mdl, in+12)
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t d_up = (uint32_t)F::dot_acc(nm, ci, (int)(F::nrm(nm) + F::nrm(ci)));  // (x, y+1):   get_dis(mdl+12, in)
            const uint32_t d_dg = (uint32_t)F::dot_acc(nm, ni, (int)(F::nrm(nm) + F::nrm(ni)));  // (x+1, y+1)
            // (x+1, y) can only leave through the LOWER bound: the current point is inside, y1 <= ub(x) + 1 = ubA1, and ub is
            // non-decreasing in x (within each piece, and across the junction: ((X1 + c1s2) >> 1) >= 2 X1 + 2 because
            // 3 X1 <= 2 mdl - in), so y1 <= ubB1 always holds for a lane that is not `lost` (round 6: one compare less)
            bool in_up = (y1 < ubA1), in_rt = (lbB < y1), in_dg = (lbB <= y1) & (y1 < ubB1);
            // DTW.C:152-184 on the SQUARED candidates.  g(d) = (u32)sqrtf((float)d) is monotone, so the step cost is
            // g(min of the admissible candidates) -- one root instead of three -- and "min == right_up" / "min == up"
            // (the tie order of DTW.C:168-184) become g(q) == g(min)  <=>  q < T, T = first d with g(d) = g(min)+1.
            // T is (g+1)^2 up to the rounding of (float)d; the exact value comes from a byte table in LDS (sr_tables.cpp:
            // T(g) = g*(g+2) + tie_delta[g]), so there is no uncertain band around it.  A root outside the staged part of
            // the table (which includes "all three outside"), an unsafe bracket or a lost lane sends the wave down the
            // literal three-root path.
            // minimum over the admissible candidates: one select and two v_min_u32 under the admissibility masks
            // (the masked-out candidates are never materialised; 0xFFFFFFFF when none is admissible)
            const uint64_t m_up = __builtin_amdgcn_ballot_w64(y1 < ubA1),
                           m_rt = __builtin_amdgcn_ballot_w64(lbB < y1),
                           m_dg = __builtin_amdgcn_ballot_w64(lbB <= y1) & __builtin_amdgcn_ballot_w64(y1 < ubB1);
            uint32_t m2;
            {
                uint64_t ex;
                // m_dg is the result of an s_and_b64, i.e. SALU-written: no VALU-write -> VALU-read SGPR hazard on the select
                asm volatile("v_cndmask_b32_e64 %0, -1, %5, %2\n\t"
                             "s_mov_b64 %1, exec\n\t"
                             "s_and_b64 exec, %1, %3\n\t"
                             "v_min_u32 %0, %0, %6\n\t"
                             "s_and_b64 exec, %1, %4\n\t"
                             "v_min_u32 %0, %0, %7\n\t"
                             "s_mov_b64 exec, %1"
                             : "=&v"(m2), "=&s"(ex)
                             : "s"(m_dg), "s"(m_rt), "s"(m_up), "v"(d_dg), "v"(d_rt), "v"(d_up)
                             : "scc");
            }
            // the conditions that send the wave down the literal path are collected as LANE MASKS (ballots of the plain
            // compares, combined on the scalar unit): a bool OR-ed together and balloted afterwards costs two extra VALU ops
            // Root of the step: v_sqrt_f32 is within one ulp of the correctly rounded root r of (float)m2, so floor(pred(s0)) is
            // g = floor(r) or g - 1 (never above: pred(s0) <= r), and it is g - 1 exactly when m2 >= T(mn) -- the threshold the
            // tie tests need anyway.  Round 6: replaces the two-sided bracket floor(succ(s0)) + "s0 > (float)mn" (a convert and a
            // float compare per step, and a fused residual in the rare branch); swept over all 2^32 inputs through sr_math_diag.
            uint32_t mn = sqrt_floor_low(m2);  // floor(pred(v_sqrt_f32)), sr_dtw_dev.h
            // T = first squared distance whose root is mn + 1 = mn*(mn + 2) + tie_delta[mn] (exact, sr_tables.cpp): a
            // candidate q >= m2 has the root mn  <=>  q < T.  One byte from the LDS table (address register = the root,
            // base = immediate offset), one add, one 24-bit multiply-add.  Roots >= tie_g -- which includes m2 = 0xFFFFFFFF,
            // "all three outside" -- read past the table (out-of-range LDS reads return 0) and go down the literal path.
            uint32_t T;
            {
                const int dl = *(lds_c_i8 *)(uintptr_t)mn;  // s_tie[mn]: the table sits at LDS address 0
                const uint32_t mp2 = mn + 2;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(T) : "v"(mn), "v"(mp2), "v"(dl));
            }
            const bool tie_dg = in_dg & (d_dg < T), tie_up = in_up & (d_up < T);
            // hard = the lanes that need the literal form (the root, or the root plus one, beyond the staged table); a root that
            // came out one short (m2 >= T: squares and their close neighbours, about 5e-4 of the lane-steps) is settled inside the branch
            const uint64_t hard = __builtin_amdgcn_ballot_w64(mn >= tie_g1) | lost;
            const uint64_t unsafe = __builtin_amdgcn_ballot_w64(m2 >= T) | hard;
            bool mv_diag = tie_dg, mv_up = tie_up & !tie_dg;
#ifdef SR_DTW_STATS
            {
                const uint64_t mb = __builtin_amdgcn_ballot_w64(m2 >= T), mr = __builtin_amdgcn_ballot_w64(mn >= tie_g1),
                               act = __builtin_amdgcn_ballot_w64(true);
                if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)) == 0) {  // first active lane
                    atomicAdd(&g_dtw_stats[0], 1ull);
                    atomicAdd(&g_dtw_stats[5], (unsigned long long)__builtin_popcountll(act));
                    if (unsafe) atomicAdd(&g_dtw_stats[1], 1ull);
                    if (mb) atomicAdd(&g_dtw_stats[2], 1ull);
                    if (mr) atomicAdd(&g_dtw_stats[3], 1ull);
                    if (lost) atomicAdd(&g_dtw_stats[4], 1ull);
                }
            }
#endif
            if (unsafe != 0ull && hard == 0ull) {
                // the root came out one short on some lanes (m2 >= T(mn) <=> g = mn + 1, exact: T is the first squared distance
                // whose root is mn + 1): step it, look the next threshold up (mn + 1 < tie_g by the test above), the two tie tests
                // once more.  Lanes with m2 < T keep everything.
                const bool shrt = m2 >= T;
                mn += shrt ? 1u : 0u;
                const int dl = *(lds_c_i8 *)(uintptr_t)mn;
                const uint32_t T2 = mn * (mn + 2) + (uint32_t)dl;
                mv_diag = in_dg & (d_dg < T2);
                mv_up = in_up & (d_up < T2) & !mv_diag;
            } else if (unsafe != 0ull) {  // wave-uniform; the literal form: dtw_limit on the three points, three roots, min, equality tests
                const int x = xB - 1, y = y1 - 1;
                in_up = !dtw_out(x, y1, X1, X2, (int)in_n, (int)mdl_n);
                in_rt = !dtw_out(xB, y, X1, X2, (int)in_n, (int)mdl_n);
                in_dg = !dtw_out(xB, y1, X1, X2, (int)in_n, (int)mdl_n);
                const uint32_t up = in_up ? cvt_u32(sqrt_rn_int((float)d_up)) : SR_DIS_ERR,
                               right = in_rt ? cvt_u32(sqrt_rn_int((float)d_rt)) : SR_DIS_ERR,
                               diag = in_dg ? cvt_u32(sqrt_rn_int((float)d_dg)) : SR_DIS_ERR;
                mn = diag;  // DTW.C:156-164
                if (mn > right) mn = right;
                if (mn > up) mn = up;
                mv_diag = (mn == diag);  // DTW.C:168-184
                mv_up = !mv_diag && (mn == up);
                lost |= __builtin_amdgcn_ballot_w64(!(in_up | in_rt | in_dg));
            }
            dis += mn;
            const bool adv_y = mv_diag || mv_up, adv_x = mv_diag || !mv_up;
            // the y advance goes first: its template-row loads come from L2 and have the longest way to go before the next
            // step's distances need them (the kernel runs close to where a wave's serial latency, not the issue port, sets
            // the pace: 6 waves per SIMD, LDS-limited; this order alone is worth 6 % of the kernel's time)
            if (adv_y) {
                y1++;
                F::copy(cm, nm);
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(t_off) : "s"(t_stride));
                nm = tpl_row(t_off);
            }
            if (adv_x) {
                // in-place updates (tied asm operands): without them the compiler builds the new values in fresh
                // registers and copies them into the loop-carried ones at the end of the block (three v_mov per step)
                asm volatile("v_add_u32 %0, 1, %0" : "+v"(xB));
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(in_off) : "n"(F::kLdsRowBytes));
                asm volatile("v_add_u32 %0, 4, %0" : "+v"(nrm_off));
                F::rows2(in_off, nrm_off, ci, ni);
                asm volatile("v_mov_b32 %0, %1" : "+v"(ubA1) : "v"(ubB1));
                {
                    int ua, lb;  // (xB << 1) + constant as ONE v_lshl_add_u32 each (the compiler shares 2*xB and spends two adds)
                    asm("v_lshl_add_u32 %0, %1, 1, 2" : "=v"(ua) : "v"(xB));
                    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(lb) : "v"(xB), "v"(c2s));
                    const int ub = (xB + c1s2) >> 1, la = xB >> 1;
                    const uint64_t m1 = __builtin_amdgcn_ballot_w64(xB < X1), m2x = __builtin_amdgcn_ballot_w64(xB < X2);
                    // s_nop 1: the masks come straight from v_cmp, and a VALU read of an SGPR written by the VALU needs two
                    // wait states (the compiler pads its own v_cmp -> v_cndmask pairs the same way)
                    asm volatile("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(ubB1) : "v"(ub), "v"(ua), "s"(m1));
                    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(lbB) : "v"(lb), "v"(la), "s"(m2x));
                }
            }
            trips++;
            const bool cx = xB <= (int)in_n, cy = y1 <= (int)mdl_n;  // DTW.C:188 (x < in && y < mdl)
            cont = cx && cy;
            // (the ballots of the two plain compares, combined on the scalar unit: a ballot of `cont` costs a select and a compare)
            if ((__builtin_amdgcn_ballot_w64(cx) & __builtin_amdgcn_ballot_w64(cy)) != __builtin_amdgcn_ballot_w64(true)) {  // wave-uniform: only trips in which a lane leaves
                asm volatile("" : "+v"(step));  // (keeps the block a branch target: if-converted, it is three selects in EVERY trip)
                if (!cont) step = trips;
            }
        } while (cont);
        score = dis / step;
    }
    a.d.scores[(size_t)b * K + a.tpl_orig[ks]] = score;
}

// pick U: maximise resident lanes doing useful work (LDS 160 KiB/CU, 32 waves/CU, 1024 threads/workgroup), then give
// what is left of the workgroup's LDS share to the tie-threshold table (tie_g entries of one byte, a power of two).
// gfx950 hands out LDS in granules of 1280 bytes (160 KiB / 128): three workgroups fit a CU only if each stays within
// 42 granules = 53 760 bytes -- 20 bytes more and the third one silently does not (measured: mean waves per SIMD 5.0 -> 3.3),
// although hipOccupancyMaxActiveBlocksPerMultiprocessor still reports 3.
constexpr size_t kLdsGranule = 1280, kCuLds = 160 * 1024;
__host__ __device__ inline size_t dtw_lds_fixed(uint32_t U, uint32_t max_frames, uint32_t row_words = 6)
{
    const size_t per_u = (size_t)(dtw_lds_row_stride(max_frames, row_words) + dtw_lds_nrm_stride(max_frames)) * 4;
    // + 32: the reload after the last advance may read one row / two norms past the last utterance's image;
    // + the frame counts of the U utterances
    return U * per_u + 32 + ((4 * (size_t)U + 15) & ~(size_t)15);
}
// Geometry of a k_dtw_lds launch for a store of K templates and max_frames rows per utterance: U utterances and Kc
// templates (ranks) per workgroup (U * Kc <= 1024 lanes; the store is walked in ceil(K / Kc) equal chunks over grid.y), the
// LDS bytes, and the tie-table entries that fit beside the utterances.  What counts: the fraction of lanes that carry a pair,
// enough resident waves to cover the LDS / L2 latency of the walk, and -- measured at K = 500 -- how many lanes share a
// template row: the texture addresser is the limit there, and U = 6 x 167 templates runs 12 % faster than U = 2 x 500
// (30.2 vs 34.3 ms per 65 536 utterances), while U = 10 x 100 (one workgroup per CU) loses 10 %.
uint32_t dtw_lds_pick_u(uint32_t K, uint32_t max_frames, size_t *lds_bytes, uint32_t *tie_g, uint32_t *kc_out, uint32_t row_words)
{
    const uint32_t kMinTie = 4096;  // below 4096 every threshold is the exact square: the least useful table
    auto blocks_for = [](size_t lds) { return (uint32_t)(kCuLds / ((lds + kLdsGranule - 1) / kLdsGranule * kLdsGranule)); };
    uint32_t best_u = 0, best_g = 0, best_kc = 0;
    double best = 0;
    // development hooks (sr_dev_hook): force U / the table size / cap the templates per workgroup.  A forced combination
    // that does not fit the CU's LDS or the grid is skipped like any other candidate (the generic kernel serves the store
    // if nothing fits).
    const uint32_t force = (uint32_t)dev_hook(kHookDtwU), force_g = (uint32_t)dev_hook(kHookDtwTieG),
                   force_kc = (uint32_t)dev_hook(kHookDtwKc);
    for (uint32_t U = 1; U <= (uint32_t)kDtwMaxU; U++) {
        uint32_t kc = K < 1024u / U ? K : 1024u / U;  // lanes of a workgroup
        if (force_kc >= 1 && force_kc < kc) kc = force_kc;
        if (!kc) break;
        const uint32_t chunks = (K + kc - 1) / kc;
        if (chunks > 65535) continue;               // the chunks are the grid's second dimension
        kc = (K + chunks - 1) / chunks;             // equal chunks
        const uint64_t pairs = (uint64_t)U * kc;
        const size_t lds = dtw_lds_fixed(U, max_frames, row_words);
        if (lds + kMinTie > 150 * 1024) break;
        const uint32_t waves = (uint32_t)((pairs + 63) / 64);
        uint32_t blocks = blocks_for(lds + kMinTie);
        if (blocks > 32 / waves) blocks = 32 / waves;
        if (blocks > 8) blocks = 8;
        if (blocks < 1) continue;
        uint32_t g = kMinTie;  // the largest table that does not cost a resident workgroup
        while (g < (uint32_t)kTieMax && blocks_for(lds + 2 * g) >= blocks) g *= 2;
        if (force_g >= kMinTie && force_g <= (uint32_t)kTieMax) {
            g = force_g & ~1023u;
            if (blocks_for(lds + g) < 1) continue;  // a forced table that does not fit beside the utterances
        }
        const double eff = (double)K / ((double)chunks * 64.0 * waves / U);  // lanes that carry a pair, over all chunks
        const double resident = (double)(blocks * waves);
        // lanes per template row: worth more the larger the store (at K = 100 five utterances x 100 templates in three
        // workgroups per CU beat eight x 100 in two, 6.4 vs 6.6 ms; at K = 500 eight x 125 beat two x 500, 31 vs 34 ms);
        // every further chunk stages the utterances once more
        const double w = 0.5 * (K >= 400 ? 1.0 : K / 400.0);
        // a table that ends at a root of 8192 (4096) sends squared distances above 6.7e7 (1.7e7) down the literal path
        const double cover = g >= 16384 ? 1.0 : g >= 8192 ? 0.98 : 0.9;
        const double share = (1.0 - w / U) * (1.0 - 0.005 * (chunks - 1)) * cover;
        double score = eff * (resident >= 24 ? 1.0 : resident / 24.0) * share;
        if (force && force == U) score = 100.0;
        if (score > best + 1e-9) {
            best = score;
            best_u = U;
            best_g = g;
            best_kc = kc;
        }
    }
    if (best_u && lds_bytes) *lds_bytes = dtw_lds_fixed(best_u, max_frames, row_words) + best_g;
    if (tie_g) *tie_g = best_g;
    if (kc_out) *kc_out = best_kc;
    return best_u;
}

void launch_dtw(const DtwArgs &a, hipStream_t s)
{
    const uint64_t n = (uint64_t)a.B * a.K;
    if (!n) return;
    // U (utterances per workgroup) and the LDS size were chosen once, when the template store was set
    const uint32_t U = a.tplR ? a.lds_u : 0;
    // GENERIC front end with another feature width: up to 12 coefficients ride the staged kernel's 12-wide form (rows zero-padded
    // to 12 in its LDS image and in the length-sorted store), 13..16 its 16-wide form; stores that cannot be staged take k_dtw_gen
    if (a.n_coef != (uint32_t)kCoef && !U) {
        hipLaunchKernelGGL(k_dtw_gen, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, s, a);
        return;
    }
    const size_t lds = a.lds_bytes;
    if (U) {
        const uint32_t Kc = (a.lds_kc && a.lds_kc < a.K) ? a.lds_kc : a.K, chunks = (a.K + Kc - 1) / Kc;
        DtwLdsArgs la{a, (const u32x4 *)a.tplR, a.tpl_frames_s, a.tpl_orig, U, a.tie_delta, a.tie_g, Kc};
        const uint32_t threads = (uint32_t)(((uint64_t)U * Kc + 63) / 64 * 64);
        if (a.n_coef > (uint32_t)kCoef) hipLaunchKernelGGL(k_dtw_lds<Dtw16>, dim3((a.B + U - 1) / U, chunks), dim3(threads), lds, s, la);
        else hipLaunchKernelGGL(k_dtw_lds<Dtw12>, dim3((a.B + U - 1) / U, chunks), dim3(threads), lds, s, la);
    } else {  // very long sequences / very many templates: generic global-memory walk
        hipLaunchKernelGGL(k_dtw, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, s, a);
    }
}

// argmin with strict '<' in slot order (main.c:276-291): first minimum wins; all dis_err -> slot 0
__global__ void __launch_bounds__(256) k_argmin(const DtwArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const uint32_t *sc = a.scores + (size_t)b * a.K;
    uint32_t best = SR_DIS_ERR, idx = 0xFFFFFFFFu;
    for (uint32_t k = lane; k < a.K; k += 64) {
        const uint32_t d = sc[k];
        if (d < best) {
            best = d;
            idx = k;
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
    }
}

void launch_argmin(const DtwArgs &a, hipStream_t s)
{
    if (!a.B) return;
    hipLaunchKernelGGL(k_argmin, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace sr
