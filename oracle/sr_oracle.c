/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle ("tier ii").  See sr_oracle.h.
 * Each function cites the reference lines it restates.  Integer widths and
 * wrap/truncation behaviour are the reference's (u32 wrap, s16 wrap, C99
 * truncating division, float->u32 truncation); compile with -fwrapv.
 */
#include "sr_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

void cr4_fft_1024_stm32(void *pssOUT, void *pssIN, uint16_t Nbin); /* q15_fft.c */
void sr_oracle_q15_fft512(uint32_t *out, const uint32_t *in);        /* q15_fft.c, EXTENSION */

struct sr_oracle {
    sr_oracle_cfg cfg;
    uint32_t frame_len;    /* VAD.H:7   frame_time*fs/1000 */
    uint32_t frame_mov;    /* VAD.H:8   frame_mov_t*fs/1000 */
    uint32_t hop;          /* frame_len-frame_mov, the stride used everywhere */
    uint32_t noise_len;    /* ADC.H:11  (fs/1000)*atap_len_t */
    uint32_t atap_frm_len; /* VAD.C:13-14 (fs/1000)*30 */
    uint32_t v_durmin_f;   /* VAD.C:72-73 80/(frame_time-frame_mov_t) */
    uint32_t s_durmax_f;   /* VAD.C:74-75 110/(frame_time-frame_mov_t) */
    uint32_t frq_max;      /* MFCC.H:9  nfft/2 */
    uint16_t *hamm, *tri_cen, *tri_odd, *tri_even;
    int8_t *dct;
};

/* Matlab int32(): round half away from zero */
static double m_round(double v) { return v >= 0.0 ? floor(v + 0.5) : -floor(-v + 0.5); }

void sr_oracle_default_cfg(sr_oracle_cfg *c)
{
    c->fs = 8000;
    c->frame_time = 20;
    c->frame_mov_t = 10;
    c->nfft = 1024;
    c->n_mel = 24;
    c->n_coef = 12;
    c->max_frames = 119;
    c->noise_len_t = 300;
    c->max_seg = 3;
}

/* speech_recog.m:217-222 */
static void gen_hamm(sr_oracle *o)
{
    uint32_t n = o->frame_len;
    for (uint32_t i = 0; i < n; i++) {
        double w = 0.54 - 0.46 * cos(2 * M_PI * (double)i / (double)(n - 1));
        o->hamm[i] = (uint16_t)m_round(w * 10000.0);
    }
}

/* speech_recog.m:240-310.  Matlab's arrays are 1-based: element j lands at C index j-1,
   and Matlab's tri_odd is the C table tri_even and vice versa (MFCC_Arg.h:17-27). */
static void gen_tri(sr_oracle *o)
{
    const double top = 1000.0;
    uint32_t n = o->cfg.n_mel, nb = o->frq_max;
    double f_max = (double)o->cfg.fs / 2;
    double mel_max = 2595 * log10(1 + f_max / 700);
    double mel_step = mel_max / (n + 1);
    double *cen = calloc(n + 2, sizeof(double));
    double *m_odd = calloc(nb + 2, sizeof(double));
    double *m_even = calloc(nb + 2, sizeof(double));
    for (uint32_t i = 1; i <= n; i++) {
        double c;
        if ((double)i < 1000.0 / mel_step)
            c = mel_step * i;
        else
            c = (exp(log(10) * (mel_step * i) / 2595) - 1) * 700;
        cen[i] = m_round(c / (f_max / (double)nb));
        o->tri_cen[i - 1] = (uint16_t)cen[i];
    }
#define RISE(arr, lo, hi)                                      \
    for (long j = (long)cen[lo]; j <= (long)cen[hi]; j++)      \
        if (j >= 1 && j <= (long)nb) arr[j] = top * (j - cen[lo]) / (cen[hi] - cen[lo]);
#define FALL(arr, lo, hi)                                      \
    for (long j = (long)cen[lo] + 1; j <= (long)cen[hi]; j++)  \
        if (j >= 1 && j <= (long)nb) arr[j] = top * (cen[hi] - j) / (cen[hi] - cen[lo]);
    for (long j = 1; j <= (long)cen[1]; j++) m_odd[j] = top * j / cen[1];
    FALL(m_odd, 1, 2)
    for (uint32_t h = 3; h <= n; h += 2) {
        RISE(m_odd, h - 1, h)
        FALL(m_odd, h, h + 1)
    }
    for (uint32_t h = 2; h + 2 <= n; h += 2) {
        RISE(m_even, h - 1, h)
        FALL(m_even, h, h + 1)
    }
    RISE(m_even, n - 1, n)
    for (long j = (long)cen[n] + 1; j <= (long)nb; j++) m_even[j] = top * ((double)nb - j) / ((double)nb - cen[n]);
#undef RISE
#undef FALL
    for (uint32_t j = 1; j <= nb; j++) {
        o->tri_even[j - 1] = (uint16_t)m_round(m_odd[j]);
        o->tri_odd[j - 1] = (uint16_t)m_round(m_even[j]);
    }
    free(cen);
    free(m_odd);
    free(m_even);
}

/* teat.m:19-26 */
static void gen_dct(sr_oracle *o)
{
    uint32_t nm = o->cfg.n_mel, nc = o->cfg.n_coef;
    for (uint32_t h = 1; h <= nc; h++)
        for (uint32_t j = 1; j <= nm; j++)
            o->dct[(h - 1) * nm + (j - 1)] = (int8_t)m_round(cos(h * M_PI * (j - 0.5) / nm) * 100);
}

sr_oracle *sr_oracle_create(const sr_oracle_cfg *cfg)
{
    sr_oracle *o;
    /* nfft 512 = the EXTENSION front end (no reference counterpart) */
    if ((cfg->nfft != 1024 && cfg->nfft != 512) || (cfg->n_mel & 1) || cfg->n_mel < 4 || cfg->n_mel > 64 ||
        cfg->max_frames == 0 || cfg->max_frames > 16383)
        return NULL;
    o = calloc(1, sizeof(*o));
    o->cfg = *cfg;
    o->frame_len = cfg->frame_time * cfg->fs / 1000;
    o->frame_mov = cfg->frame_mov_t * cfg->fs / 1000;
    o->hop = o->frame_len - o->frame_mov;
    o->noise_len = (cfg->fs / 1000) * cfg->noise_len_t;
    o->atap_frm_len = (cfg->fs / 1000) * 30;
    o->v_durmin_f = 80 / (cfg->frame_time - cfg->frame_mov_t);
    o->s_durmax_f = 110 / (cfg->frame_time - cfg->frame_mov_t);
    o->frq_max = cfg->nfft / 2;
    if (o->frame_len > cfg->nfft || o->frame_len < 2) {
        free(o);
        return NULL;
    }
    o->hamm = calloc(o->frame_len, sizeof(uint16_t));
    o->tri_cen = calloc(cfg->n_mel, sizeof(uint16_t));
    o->tri_odd = calloc(o->frq_max, sizeof(uint16_t));
    o->tri_even = calloc(o->frq_max, sizeof(uint16_t));
    o->dct = calloc(cfg->n_coef * cfg->n_mel, 1);
    gen_hamm(o);
    gen_tri(o);
    gen_dct(o);
    return o;
}

void sr_oracle_destroy(sr_oracle *o)
{
    if (!o)
        return;
    free(o->hamm);
    free(o->tri_cen);
    free(o->tri_odd);
    free(o->tri_even);
    free(o->dct);
    free(o);
}

uint32_t sr_oracle_frame_len(const sr_oracle *o) { return o->frame_len; }
uint32_t sr_oracle_hop(const sr_oracle *o) { return o->hop; }
uint32_t sr_oracle_noise_len(const sr_oracle *o) { return o->noise_len; }
const uint16_t *sr_oracle_hamm(const sr_oracle *o) { return o->hamm; }
const uint16_t *sr_oracle_tri_cen(const sr_oracle *o) { return o->tri_cen; }
const uint16_t *sr_oracle_tri_odd(const sr_oracle *o) { return o->tri_odd; }
const uint16_t *sr_oracle_tri_even(const sr_oracle *o) { return o->tri_even; }
const int8_t *sr_oracle_dct(const sr_oracle *o) { return o->dct; }

/* ---- VAD.C:22-71 -------------------------------------------------------- */
int sr_oracle_noise_atap(const sr_oracle *o, const uint16_t *noise, uint32_t n_len, sr_oracle_atap *atap)
{
    uint32_t afl = o->atap_frm_len;
    uint32_t n_sum = 0, max_sum = 0, abs_sum = 0, mid;
    if (n_len == 0 || (n_len % afl) != 0)
        return 1; /* VAD.C:33-36: silent return, atap left as it was */
    for (uint32_t i = 0; i < n_len; i++) n_sum += noise[i];
    mid = n_sum / n_len;
    for (uint32_t i = 0; i < n_len; i += afl) {
        uint32_t n_max = 0;
        for (uint32_t h = 0; h < afl; h++) {
            uint32_t v = noise[i + h];
            uint32_t a = (v > mid) ? (v - mid) : (mid - v);
            if (a > n_max)
                n_max = a;
            abs_sum += a;
        }
        max_sum += n_max;
    }
    abs_sum /= (n_len / o->frame_len); /* VAD.C:65: divisor counts frame_len blocks, not atap frames */
    max_sum /= (n_len / afl);
    atap->mid_val = mid;
    atap->n_thl = (uint16_t)(max_sum * 1);      /* n_thl_ratio 1 */
    atap->s_thl = abs_sum * 11 / 10;            /* s_thl_ratio 11/10 */
    atap->z_thl = (uint16_t)(o->frame_len * 2 / 160 / 1); /* VAD.C:70 with z_thl_ratio 2/160 */
    return 0;
}

/* ---- VAD.C:97-218 ------------------------------------------------------- */
void sr_oracle_vad(const sr_oracle *o, const uint16_t *vc, uint32_t buf_len, const sr_oracle_atap *atap, int32_t *seg)
{
    uint8_t last_sig = 0; /* never reset: carries across samples AND frames (VAD.C:99) */
    uint8_t cur_stus = 0;
    uint32_t front = 0, back = 0, valid_con = 0;
    uint32_t fl = o->frame_len, hop = o->hop;
    uint32_t mid = atap->mid_val;
    uint32_t a_thl = mid + atap->n_thl;
    uint32_t b_thl = mid - atap->n_thl; /* wraps if n_thl > mid, as in the reference */

    for (uint32_t s = 0; s < o->cfg.max_seg; s++) seg[2 * s] = seg[2 * s + 1] = -1;
    if (buf_len < fl)
        return;
    for (uint32_t i = 0; i < buf_len - fl; i += hop) {
        uint32_t frm_sum = 0, frm_zero = 0;
        for (uint32_t h = 0; h < fl; h++) {
            uint32_t v = vc[i + h];
            frm_sum += (v > mid) ? (v - mid) : (mid - v);
        }
        for (uint32_t h = 0; h < fl - 1; h++) {
            uint32_t v0 = vc[i + h], v1 = vc[i + h + 1];
            if (v0 >= a_thl)
                last_sig = 2;
            else if (v0 < b_thl)
                last_sig = 1;
            if (v1 >= a_thl) {
                if (last_sig == 1)
                    frm_zero++;
            } else if (v1 < b_thl) {
                if (last_sig == 2)
                    frm_zero++;
            }
        }
        if (frm_sum > atap->s_thl || frm_zero > atap->z_thl) {
            if (cur_stus == 0) {
                cur_stus = 1;
                front = 1;
            } else if (cur_stus == 1) {
                front++;
                if (front >= o->v_durmin_f) {
                    cur_stus = 2;
                    seg[2 * valid_con] = (int32_t)i - (int32_t)((o->v_durmin_f - 1) * hop);
                    front = 0;
                }
            } else if (cur_stus == 3) {
                back = 0;
                cur_stus = 2;
            }
        } else {
            if (cur_stus == 2) {
                cur_stus = 3;
                back = 1;
            } else if (cur_stus == 3) {
                back++;
                if (back >= o->s_durmax_f) {
                    cur_stus = 0;
                    seg[2 * valid_con + 1] = (int32_t)i - (int32_t)(o->s_durmax_f * hop) + (int32_t)fl;
                    valid_con++;
                    if (valid_con == o->cfg.max_seg)
                        return;
                    back = 0;
                }
            } else if (cur_stus == 1) {
                front = 0;
                cur_stus = 0;
            }
        }
    }
}

/* ---- MFCC.C:27-62 ------------------------------------------------------- */
int sr_oracle_fft_mag(const sr_oracle *o, const int16_t *frame, uint32_t len, uint32_t *mag)
{
    uint32_t in[1024], out[1024];
    if (len > o->cfg.nfft)
        return 1;
    for (uint32_t i = 0; i < len; i++) in[i] = (uint16_t)frame[i]; /* imag = 0 in the high half */
    for (uint32_t i = len; i < o->cfg.nfft; i++) in[i] = 0;
    if (o->cfg.nfft == 1024)
        cr4_fft_1024_stm32(out, in, 1024);
    else
        sr_oracle_q15_fft512(out, in);
    for (uint32_t i = 0; i < o->frq_max; i++) {
        int32_t re = (int16_t)out[i], im = (int16_t)(out[i] >> 16);
        int32_t r = re * re + im * im;
        mag[i] = (uint32_t)(sqrtf((float)r) * 10);
    }
    return 0;
}

/* one frame of MFCC.C:113-187; x points at the frame's first sample, x[-1] is read */
static void mfcc_frame(const sr_oracle *o, const uint16_t *x, int32_t mid, int16_t *out)
{
    int16_t vc_temp[1024];
    uint32_t spct[512];
    uint32_t pw[64];
    const uint16_t *cen = o->tri_cen;
    uint32_t fl = o->frame_len, nb = o->frq_max, nm = o->cfg.n_mel, nc = o->cfg.n_coef;

    for (uint32_t i = 0; i < fl; i++) {
        /* MFCC.C:119: (x[i]-mid) - (x[i-1]-mid)*95/100, * then / left to right, truncating */
        int32_t t = ((int32_t)x[i] - mid) - ((int32_t)x[(int32_t)i - 1] - mid) * 95 / 100;
        vc_temp[i] = (int16_t)(t * (int32_t)o->hamm[i] / (10000 / 10)); /* MFCC.C:122 */
    }
    sr_oracle_fft_mag(o, vc_temp, fl, spct);
    for (uint32_t i = 0; i < nb; i++) spct[i] *= spct[i]; /* MFCC.C:131, u32 wrap */

    /* MFCC.C:136-162: each term divided by tri_top/10 = 100 before accumulation, u32 wrap */
    pw[0] = 0;
    for (uint32_t i = 0; i < cen[1]; i++) pw[0] += spct[i] * o->tri_even[i] / 100;
    for (uint32_t h = 2; h < nm; h += 2) {
        pw[h] = 0;
        for (uint32_t i = cen[h - 1]; i < cen[h + 1]; i++) pw[h] += spct[i] * o->tri_even[i] / 100;
    }
    for (uint32_t h = 1; h < nm - 2; h += 2) {
        pw[h] = 0;
        for (uint32_t i = cen[h - 1]; i < cen[h + 1]; i++) pw[h] += spct[i] * o->tri_odd[i] / 100;
    }
    pw[nm - 1] = 0;
    for (uint32_t i = cen[nm - 2]; i < nb; i++) pw[nm - 1] += spct[i] * o->tri_odd[i] / 100;

    /* MFCC.C:165-170.  log(0) = -inf; (u32)(-inf) is UB in C but yields 0 on ARM softfp
       and on x86-64 gcc; made explicit here. */
    for (uint32_t h = 0; h < nm; h++) pw[h] = pw[h] ? (uint32_t)(log((double)pw[h]) * 100) : 0u;

    /* MFCC.C:173-183: per-term truncating /100, accumulated in s16 */
    for (uint32_t h = 0; h < nc; h++) {
        int16_t acc = 0;
        for (uint32_t i = 0; i < nm; i++) acc += (int16_t)((int32_t)pw[i] * (int32_t)o->dct[h * nm + i] / 100);
        out[h] = acc;
    }
}

/* ---- MFCC.C:86-191 ------------------------------------------------------ */
uint32_t sr_oracle_mfcc(const sr_oracle *o, const uint16_t *buf, int32_t start, int32_t end,
                        const sr_oracle_atap *atap, int16_t *mfcc)
{
    uint32_t fl = o->frame_len, hop = o->hop;
    /* MFCC.C:102, u32 arithmetic then truncated to u16 */
    uint16_t n = (uint16_t)(((uint32_t)(end - start) - fl) / hop + 1);
    uint32_t cnt = 0;
    if (n > o->cfg.max_frames)
        return 0; /* MFCC.C:103-107 */
    for (int32_t p = start; p <= end - (int32_t)fl; p += (int32_t)hop) {
        mfcc_frame(o, buf + p, (int32_t)atap->mid_val, mfcc + (size_t)cnt * o->cfg.n_coef);
        cnt++;
    }
    return cnt;
}

/* Diagnostic (describes an input, computes nothing of the path): the largest re^2 + im^2 (MFCC.C:56-57) over the first
   nfft/2 bins of every frame of the segment [start, end), frames windowed as in MFCC.C:115-124.  bench.py reports from it
   which magnitude / filterbank tier of the device's frame kernel a workload's frames fall into.  Returns the frame count. */
uint32_t sr_oracle_frame_peaks(const sr_oracle *o, const uint16_t *buf, int32_t start, int32_t end,
                               const sr_oracle_atap *atap, uint32_t *peaks, uint32_t max_out)
{
    uint32_t fl = o->frame_len, hop = o->hop, cnt = 0;
    int32_t mid = (int32_t)atap->mid_val;
    for (int32_t p = start; p <= end - (int32_t)fl && cnt < max_out; p += (int32_t)hop) {
        uint32_t in[1024], out[1024], mx = 0;
        const uint16_t *x = buf + p;
        for (uint32_t i = 0; i < fl; i++) {
            int32_t t = ((int32_t)x[i] - mid) - ((int32_t)x[(int32_t)i - 1] - mid) * 95 / 100;
            in[i] = (uint16_t)(int16_t)(t * (int32_t)o->hamm[i] / (10000 / 10));
        }
        for (uint32_t i = fl; i < o->cfg.nfft; i++) in[i] = 0;
        if (o->cfg.nfft == 1024)
            cr4_fft_1024_stm32(out, in, 1024);
        else
            sr_oracle_q15_fft512(out, in);
        for (uint32_t i = 0; i < o->frq_max; i++) {
            int32_t re = (int16_t)out[i], im = (int16_t)(out[i] >> 16);
            uint32_t r = (uint32_t)(re * re + im * im);
            if (r > mx) mx = r;
        }
        peaks[cnt++] = mx;
    }
    return cnt;
}

/* ---- DTW.C:45-62 -------------------------------------------------------- */
uint32_t sr_oracle_get_dis(const int16_t *a, const int16_t *b, uint32_t n_coef)
{
    uint32_t dis = 0;
    for (uint32_t i = 0; i < n_coef; i++) {
        int32_t d = (int32_t)a[i] - (int32_t)b[i];
        dis += (uint32_t)d * (uint32_t)d;
    }
    return (uint32_t)sqrtf((float)dis);
}

/* ---- DTW.C:76-109 (file statics become arguments) ----------------------- */
static int dtw_outside(int x, int y, int X1, int X2, int in_n, int mdl_n)
{
    if (x < X1) {
        if (y >= 2 * x + 2)
            return 1;
    } else {
        if (2 * y + in_n - 2 * mdl_n >= x + 4)
            return 1;
    }
    if (x < X2) {
        if (2 * y + 2 <= x)
            return 1;
    } else {
        if (y + 4 <= 2 * x + mdl_n - 2 * in_n)
            return 1;
    }
    return 0;
}

/* ---- DTW.C:120-192: greedy local walk, tie order diagonal > up > right -- */
uint32_t sr_oracle_dtw(const int16_t *in, uint32_t in_n, const int16_t *mdl, uint32_t mdl_n, uint32_t nc)
{
    uint32_t dis, step, x, y;
    int X1, X2;
    if (in_n > mdl_n * 2 || 2 * in_n < mdl_n)
        return SR_ORACLE_DIS_ERR;
    X1 = (int)(uint16_t)((2 * (int)mdl_n - (int)in_n) / 3);
    X2 = (int)(uint16_t)((4 * (int)in_n - 2 * (int)mdl_n) / 3);
    dis = sr_oracle_get_dis(in, mdl, nc);
    x = y = step = 1;
    do {
        uint32_t up, right, diag, mn;
        up = dtw_outside((int)x, (int)y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_ORACLE_DIS_ERR
                                                                             : sr_oracle_get_dis(mdl + nc, in, nc);
        right = dtw_outside((int)x + 1, (int)y, X1, X2, (int)in_n, (int)mdl_n) ? SR_ORACLE_DIS_ERR
                                                                                : sr_oracle_get_dis(mdl, in + nc, nc);
        diag = dtw_outside((int)x + 1, (int)y + 1, X1, X2, (int)in_n, (int)mdl_n)
                   ? SR_ORACLE_DIS_ERR
                   : sr_oracle_get_dis(mdl + nc, in + nc, nc);
        mn = diag;
        if (mn > right)
            mn = right;
        if (mn > up)
            mn = up;
        dis += mn; /* u32 wrap when all three are dis_err */
        if (mn == diag) {
            in += nc;
            x++;
            mdl += nc;
            y++;
        } else if (mn == up) {
            mdl += nc;
            y++;
        } else {
            in += nc;
            x++;
        }
        step = (uint16_t)(step + 1);
    } while (x < in_n && y < mdl_n);
    return dis / step;
}

/* ---- DTW.C:195-205 get_mean and DTW.C:217-296 get_mdl (not called by the reference's main.c) -------------
 * get_mdl walks exactly the dtw() path of (in1 as "in", in2 as "mdl") and emits, for the start point and after
 * every move, the per-coefficient mean (a+b)/2 (int arithmetic, truncation toward zero) of the two frames the
 * cursors are on; the merged template has `step` frames (DTW.C:291).  Rows beyond out_rows are not written
 * (the reference would run past its 119-frame record there).  Returns dis/step, or dis_err for length ratios
 * outside 1/2..2 (then nothing is written and *out_frames = 0). */
void sr_oracle_get_mean(const int16_t *a, const int16_t *b, int16_t *mean, uint32_t nc)
{
    for (uint32_t i = 0; i < nc; i++)
        mean[i] = (int16_t)(((int)a[i] + (int)b[i]) / 2);
}

uint32_t sr_oracle_get_mdl(const int16_t *in, uint32_t in_n, const int16_t *mdl, uint32_t mdl_n, uint32_t nc,
                           int16_t *out, uint32_t out_rows, uint32_t *out_frames)
{
    uint32_t dis, step, x, y;
    int X1, X2;
    *out_frames = 0;
    if (in_n > mdl_n * 2 || 2 * in_n < mdl_n)
        return SR_ORACLE_DIS_ERR;
    X1 = (int)(uint16_t)((2 * (int)mdl_n - (int)in_n) / 3);
    X2 = (int)(uint16_t)((4 * (int)in_n - 2 * (int)mdl_n) / 3);
    dis = sr_oracle_get_dis(in, mdl, nc);
    if (out_rows > 0)
        sr_oracle_get_mean(in, mdl, out, nc);
    x = y = step = 1;
    do {
        uint32_t up, right, diag, mn;
        up = dtw_outside((int)x, (int)y + 1, X1, X2, (int)in_n, (int)mdl_n) ? SR_ORACLE_DIS_ERR
                                                                             : sr_oracle_get_dis(mdl + nc, in, nc);
        right = dtw_outside((int)x + 1, (int)y, X1, X2, (int)in_n, (int)mdl_n) ? SR_ORACLE_DIS_ERR
                                                                                : sr_oracle_get_dis(mdl, in + nc, nc);
        diag = dtw_outside((int)x + 1, (int)y + 1, X1, X2, (int)in_n, (int)mdl_n)
                   ? SR_ORACLE_DIS_ERR
                   : sr_oracle_get_dis(mdl + nc, in + nc, nc);
        mn = diag;
        if (mn > right)
            mn = right;
        if (mn > up)
            mn = up;
        dis += mn;
        if (mn == diag) {
            in += nc;
            x++;
            mdl += nc;
            y++;
        } else if (mn == up) {
            mdl += nc;
            y++;
        } else {
            in += nc;
            x++;
        }
        if (step < out_rows) /* row index = step before the increment (DTW.C:286-287) */
            sr_oracle_get_mean(in, mdl, out + (size_t)step * nc, nc);
        step = (uint16_t)(step + 1);
    } while (x < in_n && y < mdl_n);
    *out_frames = step;
    return dis / step;
}

/* the three non-integer expressions of the path on their own (same C expressions as MFCC.C:168, DTW.C:59, MFCC.C:56-58) */
void sr_oracle_math_diag(const uint32_t *in, uint32_t *out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        uint32_t x = in[i];
        int32_t r = (int32_t)(x & 0x7FFFFFFFu);
        out[3 * i + 0] = x ? (uint32_t)(log((double)x) * 100) : 0u;
        out[3 * i + 1] = (uint32_t)sqrtf((float)x);
        out[3 * i + 2] = (uint32_t)(sqrtf((float)r) * 10);
    }
}

typedef struct {
    const uint32_t *in;
    uint32_t *out;
    uint32_t n;
} diag_job;
static void *diag_worker(void *arg)
{
    diag_job *j = arg;
    sr_oracle_math_diag(j->in, j->out, j->n);
    return NULL;
}
/* same, split over n_threads host threads (used by the exhaustive 2^32 sweep) */
void sr_oracle_math_diag_mt(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t n_threads)
{
    pthread_t th[512];
    diag_job jobs[512];
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 512) n_threads = 512;
    for (uint32_t t = 0; t < n_threads; t++) {
        uint32_t lo = (uint32_t)((uint64_t)n * t / n_threads), hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
        jobs[t].in = in + lo;
        jobs[t].out = out + (size_t)3 * lo;
        jobs[t].n = hi - lo;
        pthread_create(&th[t], NULL, diag_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}

/* dtw_limit (DTW.C:76-109) for one point of one length pair: 1 = outside the parallelogram (test helper) */
int sr_oracle_dtw_outside(int x, int y, uint32_t in_n, uint32_t mdl_n)
{
    int X1 = (int)(uint16_t)((2 * (int)mdl_n - (int)in_n) / 3), X2 = (int)(uint16_t)((4 * (int)in_n - 2 * (int)mdl_n) / 3);
    return dtw_outside(x, y, X1, X2, (int)in_n, (int)mdl_n);
}

/* ---- NON-REFERENCE extension: full dynamic-programming DTW ---------------
 * Own definition (no reference counterpart; the reference's dtw() is the greedy walk above):
 *   cells (x,y), 1-based, allowed iff dtw_limit(x,y) == ins (DTW.C:76-109) with the pair's X1/X2;
 *   d = get_dis (DTW.C:45-62); D(1,1) = d(1,1); D(x,y) = d(x,y) + min over allowed, reachable
 *   predecessors (x-1,y-1), (x-1,y), (x,y-1); sums saturate at 0xFFFFFFFE;
 *   result = D(in,mdl) / (in+mdl), dis_err when the length gate (DTW.C:133) fails or the end cell is unreachable. */
uint32_t sr_oracle_dtw_dp(const int16_t *in, uint32_t in_n, const int16_t *mdl, uint32_t mdl_n, uint32_t nc)
{
    const uint32_t INF = 0xFFFFFFFFu;
    uint32_t *prev, *cur, res;
    int X1, X2;
    if (!in_n || !mdl_n || in_n > mdl_n * 2 || 2 * in_n < mdl_n)
        return SR_ORACLE_DIS_ERR;
    X1 = (int)(uint16_t)((2 * (int)mdl_n - (int)in_n) / 3);
    X2 = (int)(uint16_t)((4 * (int)in_n - 2 * (int)mdl_n) / 3);
    prev = malloc(sizeof(uint32_t) * (mdl_n + 1));
    cur = malloc(sizeof(uint32_t) * (mdl_n + 1));
    for (uint32_t y = 0; y <= mdl_n; y++) prev[y] = INF;
    for (uint32_t x = 1; x <= in_n; x++) { /* column by column; prev = column x-1 */
        cur[0] = INF;
        for (uint32_t y = 1; y <= mdl_n; y++) {
            uint32_t best, d, sum;
            cur[y] = INF;
            if (dtw_outside((int)x, (int)y, X1, X2, (int)in_n, (int)mdl_n))
                continue;
            best = prev[y - 1];
            if (prev[y] < best) best = prev[y];
            if (cur[y - 1] < best) best = cur[y - 1];
            if (x == 1 && y == 1) best = 0;
            if (best == INF)
                continue;
            d = sr_oracle_get_dis(in + (size_t)(x - 1) * nc, mdl + (size_t)(y - 1) * nc, nc);
            sum = best + d;
            cur[y] = (sum < best || sum == INF) ? 0xFFFFFFFEu : sum;
        }
        { uint32_t *t = prev; prev = cur; cur = t; }
    }
    res = prev[mdl_n] == INF ? SR_ORACLE_DIS_ERR : prev[mdl_n] / (in_n + mdl_n);
    free(prev);
    free(cur);
    return res;
}

/* all pairs of B records x K templates through sr_oracle_dtw_dp, split over n_threads host threads (test helper):
 * in [B][in_rows][nc], in_n[B]; mdl [K][mdl_rows][nc], mdl_n[K] (0 = invalid slot -> dis_err); out [B][K] */
typedef struct {
    const int16_t *in, *mdl;
    const uint32_t *in_n, *mdl_n;
    uint32_t in_rows, mdl_rows, K, nc, b0, b1;
    uint32_t *out;
} dp_job;
static void *dp_worker(void *arg)
{
    dp_job *j = arg;
    for (uint32_t b = j->b0; b < j->b1; b++)
        for (uint32_t k = 0; k < j->K; k++)
            j->out[(size_t)b * j->K + k] = sr_oracle_dtw_dp(j->in + (size_t)b * j->in_rows * j->nc, j->in_n[b],
                                                            j->mdl + (size_t)k * j->mdl_rows * j->nc, j->mdl_n[k], j->nc);
    return NULL;
}
void sr_oracle_dtw_dp_batch(const int16_t *in, const uint32_t *in_n, uint32_t in_rows, uint32_t B, const int16_t *mdl,
                            const uint32_t *mdl_n, uint32_t mdl_rows, uint32_t K, uint32_t nc, uint32_t *out,
                            uint32_t n_threads)
{
    pthread_t th[256];
    dp_job jobs[256];
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (n_threads > B) n_threads = B ? B : 1;
    for (uint32_t t = 0; t < n_threads; t++) {
        dp_job j = {in, mdl, in_n, mdl_n, in_rows, mdl_rows, K, nc, (uint32_t)((uint64_t)B * t / n_threads),
                    (uint32_t)((uint64_t)B * (t + 1) / n_threads), out};
        jobs[t] = j;
        pthread_create(&th[t], NULL, dp_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}

/* ---- main.c:249-296 ----------------------------------------------------- */
static void recognize_segment(const sr_oracle *o, const uint16_t *pcm, uint32_t buf_len, const sr_oracle_templates *tpl,
                              uint32_t seg_idx, sr_oracle_result *res, int16_t *mfcc_out, uint32_t *scores);

void sr_oracle_recognize(const sr_oracle *o, const uint16_t *pcm, uint32_t buf_len, const sr_oracle_templates *tpl,
                         sr_oracle_result *res, int16_t *mfcc_out, uint32_t *scores)
{
    recognize_segment(o, pcm, buf_len, tpl, 0, res, mfcc_out, scores);
}

/* Extension of main.c:249-296 to every segment the VAD returns (the firmware only matches segment 0,
   main.c:268): res[s], scores[s*n + k] for s < max_seg. */
void sr_oracle_recognize_segments(const sr_oracle *o, const uint16_t *pcm, uint32_t buf_len,
                                  const sr_oracle_templates *tpl, sr_oracle_result *res, uint32_t *scores)
{
    for (uint32_t s = 0; s < o->cfg.max_seg; s++)
        recognize_segment(o, pcm, buf_len, tpl, s, &res[s], NULL, scores ? scores + (size_t)s * tpl->n : NULL);
}

static void recognize_segment(const sr_oracle *o, const uint16_t *pcm, uint32_t buf_len, const sr_oracle_templates *tpl,
                              uint32_t seg_idx, sr_oracle_result *res, int16_t *mfcc_out, uint32_t *scores)
{
    sr_oracle_atap atap;
    int32_t seg[16];
    int16_t *mfcc = mfcc_out;
    uint32_t nfrm, min_dis = SR_ORACLE_DIS_ERR, min_idx = 0;
    uint32_t nc = o->cfg.n_coef;

    memset(&atap, 0, sizeof atap);
    res->best_tpl = 0;
    res->min_dis = SR_ORACLE_DIS_ERR;
    res->frm_num = 0;
    if (scores)
        for (uint32_t k = 0; k < tpl->n; k++) scores[k] = SR_ORACLE_DIS_ERR;
    sr_oracle_noise_atap(o, pcm, o->noise_len, &atap);
    sr_oracle_vad(o, pcm, buf_len, &atap, seg);
    if (seg[2 * seg_idx + 1] < 0) {
        res->status = SR_ORACLE_VAD_FAIL;
        return;
    }
    if (seg[2 * seg_idx] < 1) {
        res->status = SR_ORACLE_SEG_OOB;
        return;
    }
    if (!mfcc)
        mfcc = calloc((size_t)(o->cfg.max_frames + 1) * nc, sizeof(int16_t));
    nfrm = sr_oracle_mfcc(o, pcm, seg[2 * seg_idx], seg[2 * seg_idx + 1], &atap, mfcc);
    res->frm_num = nfrm;
    if (nfrm == 0) {
        res->status = SR_ORACLE_MFCC_FAIL;
    } else {
        for (uint32_t k = 0; k < tpl->n; k++) {
            uint32_t cur = tpl->valid[k] ? sr_oracle_dtw(mfcc, nfrm, tpl->mfcc + (size_t)k * tpl->stride, tpl->frames[k], nc)
                                         : SR_ORACLE_DIS_ERR;
            if (scores)
                scores[k] = cur;
            if (cur < min_dis) {
                min_dis = cur;
                min_idx = k;
            }
        }
        res->best_tpl = min_idx;
        res->min_dis = min_dis;
        res->status = SR_ORACLE_OK;
    }
    if (!mfcc_out)
        free(mfcc);
}

typedef struct {
    const sr_oracle *o;
    const uint16_t *pcm;
    uint64_t pcm_stride;
    uint32_t buf_len, b0, b1;
    const sr_oracle_templates *tpl;
    sr_oracle_result *res;
    int16_t *mfcc_out;
    uint32_t *scores;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *j = arg;
    size_t msz = (size_t)j->o->cfg.max_frames * j->o->cfg.n_coef;
    for (uint32_t b = j->b0; b < j->b1; b++)
        sr_oracle_recognize(j->o, j->pcm + (size_t)b * j->pcm_stride, j->buf_len, j->tpl, &j->res[b],
                            j->mfcc_out ? j->mfcc_out + (size_t)b * msz : NULL,
                            j->scores ? j->scores + (size_t)b * j->tpl->n : NULL);
    return NULL;
}

void sr_oracle_recognize_batch(const sr_oracle *o, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len,
                               uint32_t B, const sr_oracle_templates *tpl, sr_oracle_result *res, int16_t *mfcc_out,
                               uint32_t *scores, uint32_t n_threads)
{
    pthread_t th[256];
    batch_job jobs[256];
    if (n_threads < 1)
        n_threads = 1;
    if (n_threads > 256)
        n_threads = 256;
    if (n_threads > B)
        n_threads = B ? B : 1;
    for (uint32_t t = 0; t < n_threads; t++) {
        batch_job *j = &jobs[t];
        j->o = o;
        j->pcm = pcm;
        j->pcm_stride = pcm_stride;
        j->buf_len = buf_len;
        j->b0 = (uint32_t)((uint64_t)B * t / n_threads);
        j->b1 = (uint32_t)((uint64_t)B * (t + 1) / n_threads);
        j->tpl = tpl;
        j->res = res;
        j->mfcc_out = mfcc_out;
        j->scores = scores;
        if (n_threads == 1)
            batch_worker(j);
        else
            pthread_create(&th[t], NULL, batch_worker, j);
    }
    if (n_threads > 1)
        for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}

/* ---- EXTENSION, no reference counterpart: first-order difference cepstra ("delta MFCC") ---------------------------
 * The thesis that accompanies the reference (p.32) names difference cepstra as future work; the firmware has none.
 * Definition used by this repository (the product kernel k_delta_mfcc mirrors it):
 *   the standard two-frame regression over the s16 MFCC rows of one record,
 *     d[t][c] = ( (m[t+1][c] - m[t-1][c]) + 2*(m[t+2][c] - m[t-2][c]) ) / 10
 *   with row indices clamped to [0, n-1] (edge replication), all arithmetic in s32 and the division truncating toward
 *   zero like every division of MFCC.C (C99); |numerator| <= 6*65535, so the quotient always fits s16.
 * out[n][nc]; rows are independent of anything past row n-1. */
void sr_oracle_delta_mfcc(const int16_t *m, uint32_t n, uint32_t nc, int16_t *out)
{
    for (uint32_t t = 0; t < n; t++) {
        const uint32_t p1 = t + 1 < n ? t + 1 : n - 1, p2 = t + 2 < n ? t + 2 : n - 1;
        const uint32_t m1 = t >= 1 ? t - 1 : 0, m2 = t >= 2 ? t - 2 : 0;
        for (uint32_t c = 0; c < nc; c++) {
            const int32_t num = ((int32_t)m[p1 * nc + c] - (int32_t)m[m1 * nc + c]) +
                                2 * ((int32_t)m[p2 * nc + c] - (int32_t)m[m2 * nc + c]);
            out[t * nc + c] = (int16_t)(num / 10);
        }
    }
}
