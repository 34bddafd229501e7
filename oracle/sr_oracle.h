/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle ("tier ii"), never part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parametrised C restatement of the reference hot path
 *   noise_atap -> VAD -> get_mfcc (fft -> cr4_fft_1024_stm32) -> dtw -> spch_recg
 * (reference Src/Speech_Recog/{VAD,MFCC,DTW}.C, Src/BSP/cr4_fft_1024_stm32.s,
 * Src/APP/main.c:249-296).  At the reference's compile-time constants it is
 * required (tests/test_oracle.py, tests/test_real_audio.py) to be bit-identical to tier (i) =
 * oracle/_ref/libsr_ref.so, the reference's own .C files compiled verbatim.
 * It exists because the reference's #defines are unguarded, so the verbatim
 * build cannot run the 256-frame / many-template benchmark shapes.
 *
 * Parity pinning: the reference ships NO golden vectors or tests (SURVEY.md
 * section 4).  Parity is pinned by tier (i) run in the build container -- at the
 * reference's constants (119 frames) and, through oracle/_ref/libsr_ref320.so
 * (the same .C files with vv_tim_max raised, see oracle/Makefile), at the
 * benchmark's 256-frame / 100-template shape -- and by fixtures under
 * tests/golden/ generated from tier (i) by tests/golden/make_golden.py.
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sr_oracle_cfg {
    uint32_t fs;           /* ADC.H:7        8000 */
    uint32_t frame_time;   /* VAD.H:5        20 ms */
    uint32_t frame_mov_t;  /* VAD.H:6        10 ms */
    uint32_t nfft;         /* MFCC.H:8       1024 (only value the Q15 FFT converts) */
    uint32_t n_mel;        /* MFCC.H:12      24, must be even */
    uint32_t n_coef;       /* MFCC.H:13      12 */
    uint32_t max_frames;   /* MFCC.H:15-16   119 */
    uint32_t noise_len_t;  /* ADC.H:10       300 ms */
    uint32_t max_seg;      /* VAD.H:4        3 */
} sr_oracle_cfg;

typedef struct sr_oracle_atap { /* VAD.H:10-16 */
    uint32_t mid_val;
    uint16_t n_thl;
    uint16_t z_thl;
    uint32_t s_thl;
} sr_oracle_atap;

typedef struct sr_oracle sr_oracle;

#define SR_ORACLE_DIS_ERR 0xFFFFFFFFu /* DTW.H:4 */

/* status codes of sr_oracle_recognize (mirrors main.c:38-41 + one new case) */
#define SR_ORACLE_OK 0
#define SR_ORACLE_VAD_FAIL 1  /* main.c:261-266 */
#define SR_ORACLE_MFCC_FAIL 2 /* main.c:269-274 */
#define SR_ORACLE_SEG_OOB 3   /* segment start < 1: the reference would read outside VcBuf */

void sr_oracle_default_cfg(sr_oracle_cfg *cfg);
sr_oracle *sr_oracle_create(const sr_oracle_cfg *cfg);
void sr_oracle_destroy(sr_oracle *o);

/* derived constants */
uint32_t sr_oracle_frame_len(const sr_oracle *o);
uint32_t sr_oracle_hop(const sr_oracle *o);
uint32_t sr_oracle_noise_len(const sr_oracle *o);

/* constant tables (MFCC_Arg.h:6-44), regenerated from the Matlab formulas */
const uint16_t *sr_oracle_hamm(const sr_oracle *o);     /* [frame_len] */
const uint16_t *sr_oracle_tri_cen(const sr_oracle *o);  /* [n_mel] */
const uint16_t *sr_oracle_tri_odd(const sr_oracle *o);  /* [nfft/2] */
const uint16_t *sr_oracle_tri_even(const sr_oracle *o); /* [nfft/2] */
const int8_t *sr_oracle_dct(const sr_oracle *o);        /* [n_coef*n_mel] */

/* VAD.C:22-71.  Returns 1 (atap untouched) when n_len % atap_frm_len != 0. */
int sr_oracle_noise_atap(const sr_oracle *o, const uint16_t *noise, uint32_t n_len, sr_oracle_atap *atap);
/* VAD.C:97-218.  seg[2*s], seg[2*s+1] = start/end sample offsets, -1 = NULL. */
void sr_oracle_vad(const sr_oracle *o, const uint16_t *vc, uint32_t buf_len, const sr_oracle_atap *atap,
                   int32_t *seg);
/* MFCC.C:27-62 (magnitude*10 of the first nfft/2 bins of one frame). */
int sr_oracle_fft_mag(const sr_oracle *o, const int16_t *frame, uint32_t len, uint32_t *mag);
/* MFCC.C:86-191.  buf[start-1] is read (MFCC.C:119).  Returns frm_num (0 if > max_frames). */
uint32_t sr_oracle_mfcc(const sr_oracle *o, const uint16_t *buf, int32_t start, int32_t end,
                        const sr_oracle_atap *atap, int16_t *mfcc);
/* diagnostic: largest re^2 + im^2 (MFCC.C:56-57) of each frame of the segment; returns the number of frames written */
uint32_t sr_oracle_frame_peaks(const sr_oracle *o, const uint16_t *buf, int32_t start, int32_t end,
                               const sr_oracle_atap *atap, uint32_t *peaks, uint32_t max_out);
/* DTW.C:45-62 */
uint32_t sr_oracle_get_dis(const int16_t *a, const int16_t *b, uint32_t n_coef);
/* DTW.C:120-192.  Frames past *_frames may be read (do-while), exactly as the reference does. */
uint32_t sr_oracle_dtw(const int16_t *in, uint32_t in_frames, const int16_t *mdl, uint32_t mdl_frames,
                       uint32_t n_coef);

/* out[3i] = (u32)(log((double)x)*100), out[3i+1] = (u32)sqrtf((float)x), out[3i+2] = (u32)(sqrtf((float)(s32)(x&0x7fffffff))*10) */
void sr_oracle_get_mean(const int16_t *a, const int16_t *b, int16_t *mean, uint32_t nc);           /* DTW.C:195-205 */
uint32_t sr_oracle_get_mdl(const int16_t *in, uint32_t in_n, const int16_t *mdl, uint32_t mdl_n, uint32_t nc,
                           int16_t *out, uint32_t out_rows, uint32_t *out_frames);                  /* DTW.C:217-296 */
void sr_oracle_math_diag(const uint32_t *in, uint32_t *out, uint32_t n);

/* NON-REFERENCE extension (own definition, see sr_oracle.c): full-DP DTW with the same parallelogram and distance. */
uint32_t sr_oracle_dtw_dp(const int16_t *in, uint32_t in_frames, const int16_t *mdl, uint32_t mdl_frames,
                          uint32_t n_coef);
/* dtw_limit (DTW.C:76-109) for one point of one length pair: 1 = outside (test helper) */
int sr_oracle_dtw_outside(int x, int y, uint32_t in_frames, uint32_t mdl_frames);
/* all B x K pairs, threaded (test helper); mdl_n[k] == 0 marks an invalid slot */
void sr_oracle_dtw_dp_batch(const int16_t *in, const uint32_t *in_n, uint32_t in_rows, uint32_t B, const int16_t *mdl,
                            const uint32_t *mdl_n, uint32_t mdl_rows, uint32_t K, uint32_t nc, uint32_t *out,
                            uint32_t n_threads);

/* EXTENSION (no reference counterpart): two-frame regression delta cepstra of one record, see sr_oracle.c */
void sr_oracle_delta_mfcc(const int16_t *m, uint32_t n, uint32_t nc, int16_t *out);

/*
 * Template store in the batched layout: tpl_mfcc[k] starts at k*tpl_stride int16s,
 * frame-major; tpl_frames[k] = frm_num; tpl_valid[k] != 0 <=> save_sign == 12345.
 */
typedef struct sr_oracle_templates {
    const int16_t *mfcc;
    const uint32_t *frames;
    const uint8_t *valid;
    uint32_t n;
    uint32_t stride; /* in int16 elements */
} sr_oracle_templates;

typedef struct sr_oracle_result { /* same record the product writes */
    uint32_t best_tpl; /* argmin slot, first minimum wins (main.c:285-289) */
    uint32_t min_dis;  /* *mtch_dis */
    uint32_t frm_num;  /* frames of segment 0 */
    uint32_t status;
} sr_oracle_result;

/* main.c:249-296 for one capture buffer.  mfcc_out (optional) [max_frames*n_coef],
   scores (optional) [tpl->n]. */
void sr_oracle_recognize(const sr_oracle *o, const uint16_t *pcm, uint32_t buf_len, const sr_oracle_templates *tpl,
                         sr_oracle_result *res, int16_t *mfcc_out, uint32_t *scores);

/* Every VAD segment (up to max_seg) matched like segment 0; extension, the firmware stops at segment 0
   (main.c:268).  res[max_seg], scores[max_seg * tpl->n] (optional). */
void sr_oracle_recognize_segments(const sr_oracle *o, const uint16_t *pcm, uint32_t buf_len,
                                  const sr_oracle_templates *tpl, sr_oracle_result *res, uint32_t *scores);

/* B independent buffers (pcm + b*pcm_stride), utterances split over n_threads host threads.
   Used for parity sweeps and for bench.py's cpu_baseline ("port") timing. */
void sr_oracle_recognize_batch(const sr_oracle *o, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len,
                               uint32_t B, const sr_oracle_templates *tpl, sr_oracle_result *res,
                               int16_t *mfcc_out /* B*max_frames*n_coef or NULL */,
                               uint32_t *scores /* B*n or NULL */, uint32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
