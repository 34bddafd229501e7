/*
 * sr_engine.h -- C ABI of the MI355X-native isolated-word recognition engine.
 *
 * Drop-in boundary for the hot path of gk969/stm32-speech-recognition
 * (noise_atap -> VAD -> get_mfcc/fft/cr4_fft_1024_stm32 -> dtw -> spch_recg).
 * The reference has no plugin API; its boundary is the set of C functions that
 * Src/APP/main.c calls (main.c:121-138, 249-296).  This header declares
 *
 *   1. the batched engine API the benchmark drives (no reference counterpart:
 *      the firmware recognises one utterance at a time), and
 *   2. the reference's own scalar entry points, same names / argument meaning /
 *      sentinel error behaviour, each executed as a 1-item dispatch of the same
 *      HIP kernels (there is NO CPU fallback anywhere in this library: every
 *      entry point fails with SR_ERR_NO_DEVICE if no gfx950 device is usable).
 *
 * Plain C types only; every pointer is caller-owned.  Thread-safety: distinct
 * sr_engine handles are independent; one handle must not be used concurrently.
 * The scalar reference-compatible symbols share one implicit handle and are,
 * like the reference (file-scope statics in MFCC.C:14-15, DTW.C:65-68,
 * main.c:22-25), not re-entrant.
 */
#ifndef SR_ENGINE_H
#define SR_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status / errors */
#define SR_OK 0
#define SR_ERR_NO_DEVICE 1   /* no HIP device / not gfx950 / HIP runtime error */
#define SR_ERR_BAD_CONFIG 2  /* configuration not supported by the kernels */
#define SR_ERR_BAD_ARG 3     /* null pointer, bad alignment, size out of range */
#define SR_ERR_NO_TEMPLATES 4
#define SR_ERR_HIP 5

/* per-utterance status in sr_result.status (the reference's sentinel returns) */
#define SR_ST_OK 0
#define SR_ST_VAD_FAIL 1  /* valid_voice[0].end == NULL            main.c:261-266 */
#define SR_ST_MFCC_FAIL 2 /* frm_num == 0 (segment > max_frames)   main.c:269-274, MFCC.C:103-107 */
#define SR_ST_SEG_OOB 3   /* segment starts at sample < 1: the reference would read before VcBuf (MFCC.C:119) */

#define SR_DIS_ERR 0xFFFFFFFFu /* dis_err / dis_max, DTW.H:4-5 */
#define SR_SAVE_MASK 12345u    /* save_mask, Flash.H:11 */

/* ------------------------------------------------------------------ configuration
 * The reference's compile-time constants (ADC.H:7-11, VAD.H:4-8, MFCC.H:7-16) as a run-time configuration.  Two front ends
 * have specialised kernels:
 *   reference   fs 8000, 20/10 ms framing (160/80 samples), nfft 1024, 24 Mel, 12 MFCC   (the firmware's constants)
 *   extension   fs 16000, 20/10 ms framing (320/160 samples), nfft 512, 40 Mel, 12 MFCC   (no reference counterpart)
 * Every other accepted configuration runs the GENERIC front end (same arithmetic rules, tables from the same formulas, about
 * 2.3x slower per frame; no reference counterpart for the constants):
 *   nfft 1024; fs a multiple of 4000 Hz; frame_time_ms = 2 * frame_mov_ms with a frame of 160, 240, 256, 320, 400 or 512
 *   samples; n_mel even, 4..64; n_coef 1..16 (feature records are n_coef wide everywhere: mfcc[B][max_frames][n_coef],
 *   template rows, slot images; sr_get_mdl_batch and the full-DP scorer require 12).
 * Anything else: SR_ERR_BAD_CONFIG.  Free in every configuration: max_frames (2..16383), noise_len_ms (a multiple of 30 ms
 * that holds whole frames), max_seg (1..3), device. */
typedef struct sr_config {
    uint32_t fs;            /* ADC.H:7       8000 */
    uint32_t frame_time_ms; /* VAD.H:5       20  -> frame_len 160 */
    uint32_t frame_mov_ms;  /* VAD.H:6       10  -> hop 80 */
    uint32_t nfft;          /* MFCC.H:8      1024 */
    uint32_t n_mel;         /* MFCC.H:12     24 */
    uint32_t n_coef;        /* MFCC.H:13     12 */
    uint32_t max_frames;    /* MFCC.H:15-16  119 in the firmware; runtime cap here (<= 16383) */
    uint32_t noise_len_ms;  /* ADC.H:10      300 -> atap_len 2400 */
    uint32_t max_seg;       /* VAD.H:4       3 */
    int32_t device;         /* HIP device ordinal; -1 = current device */
} sr_config;

/* adaptive thresholds, VAD.H:10-16 (same field order and widths) */
typedef struct sr_atap {
    uint32_t mid_val;
    uint16_t n_thl;
    uint16_t z_thl;
    uint32_t s_thl;
} sr_atap;

#define SR_MAX_SEG 3
/* what the VAD stage leaves per utterance (device- or host-resident), 48 bytes */
typedef struct sr_vad_rec {
    sr_atap atap;                /* noise_atap output */
    int32_t seg[2 * SR_MAX_SEG]; /* start/end sample offsets of up to 3 segments, -1 = NULL (VAD.H:18-22) */
    uint32_t frm_num;            /* frames of segment 0 per MFCC.C:102-107 (0 on any failure) */
    uint32_t status;             /* SR_ST_* */
    uint32_t _pad;
} sr_vad_rec;

/* recognition record, 16 bytes: the argmin of main.c:276-295 */
typedef struct sr_result {
    uint32_t best_tpl; /* template slot of the first minimum (strict <, main.c:285) */
    uint32_t min_dis;  /* *mtch_dis; SR_DIS_ERR if nothing matched */
    uint32_t frm_num;  /* frames of segment 0 */
    uint32_t status;   /* SR_ST_* */
} sr_result;

typedef struct sr_engine sr_engine;

/* ------------------------------------------------------------------ lifecycle */
void sr_default_config(sr_config *cfg); /* the reference's compile-time constants */
int sr_create(const sr_config *cfg, sr_engine **out);
void sr_destroy(sr_engine *h);
const char *sr_last_error(void); /* thread-local text of the last failure */

/* Host-only (touches no device): the constant tables sr_create generates for cfg, for inspection and for
 * diffing against the reference's pasted tables (MFCC_Arg.h:6-44: hamm, tri_cen, tri_odd, tri_even, dct_arg;
 * cr4_fft_1024_stm32.s:285-629: the (Kr', Ki) coefficient columns).  Every pointer may be NULL. */
typedef struct sr_tables {
    uint16_t *hamm;     /* [frame_len]       hamm[],    MFCC_Arg.h:6-9 */
    uint16_t *tri_cen;  /* [n_mel]           tri_cen[], MFCC_Arg.h:11-15 */
    uint16_t *tri_even; /* [nfft / 2]        tri_even[] */
    uint16_t *tri_odd;  /* [nfft / 2]        tri_odd[] */
    int8_t *dct;        /* [n_coef * n_mel]  dct_arg[] */
    int16_t *tw_kr;     /* [1020]            first  DCW column of the ST coefficient table, in table order */
    int16_t *tw_ki;     /* [1020]            second DCW column */
    uint32_t *log_thr;  /* [2220]            log_thr[m] = min{n : (u32)(log((double)n)*100) >= m} (host libm), [2219] = sentinel */
} sr_tables; /* layout frozen (eight pointers): further tables get their own entry point, like sr_build_tie_table below, so a
              * caller compiled against an older header never hands over a shorter struct than the library reads */
int sr_build_tables(const sr_config *cfg, const sr_tables *out);
/* Host-only: the tie-threshold table of the staged DTW kernel, out[32768]:
 * DTW.C:59,156-184: T(g) = g*(g+2) + out[g] = min{d : (u32)sqrtf((float)d) >= g + 1}  (independent of the front end) */
int sr_build_tie_table(int8_t *out);
/* The log step table is built with the run-time host's libm `log` -- the expression MFCC.C:168 evaluates -- and compared with
 * the positions the library ships (csrc/sr_log_thr_ref.inc: the libm the golden fixtures were generated with).  If they
 * differ (a libm whose log is off in the last bit at an integer crossing of log(n)*100) the SHIPPED table is used, sr_create
 * still succeeds, sr_last_error() holds a warning, and this returns the number of differing entries of the last
 * sr_create / sr_build_tables of the process (0 = this host agrees). */
int sr_log_table_mismatches(void);
/* The magnitude stage of the reference front end's frame kernel, (u32)(sqrtf(re^2+im^2)*10) (MFCC.C:56-58), takes the
 * uncorrected v_sqrt_f32 on frames whose largest re^2+im^2 is at most 70 171: equal to the exact form there on gfx950, a
 * property of the chip that sr_create re-checks on the device it runs on by sweeping the whole range.  Returns the bound
 * in use for this engine: 70171 after a clean sweep, 0 (every frame takes the exactly corrected root; sr_last_error()
 * holds a warning from sr_create) otherwise, and for front ends whose kernels do not use the form. */
uint32_t sr_mag_cheap_bound(const sr_engine *h);

/* ------------------------------------------------------------------ template store
 * The firmware keeps templates as v_ftr_tag images in MCU flash at a 4 KiB stride
 * (Flash.H:11-20, MFCC.H:18-25: u16 save_sign | u16 frm_num | s16 mfcc[]), and
 * spch_recg scans every slot in address order (main.c:279-291). */
int sr_set_templates(sr_engine *h, const void *store, uint32_t n_slots, uint32_t stride_bytes);
/* dense batched layout: mfcc[k*tpl_stride + frame*n_coef + c]; valid[k]!=0 <=> save_sign==12345 */
int sr_set_templates_dense(sr_engine *h, const int16_t *mfcc, const uint32_t *frames, const uint8_t *valid,
                           uint32_t n_templates, uint32_t tpl_stride);
uint32_t sr_num_templates(const sr_engine *h);
/* Template training = save_mdl (main.c:121-138): capture i -> noise_atap/VAD/get_mfcc -> slot[i] of the
 * host store image exactly as save_ftr_mdl programs it (Flash.C:17-67: slot erased to 0xFF, then
 * save_mask | frm_num | frm_num*12 coefficients).  status[i] (optional): 0 save_ok, 1 VAD_fail,
 * 2 MFCC_fail (main.c:38-40), 3 SR_ST_SEG_OOB; failed captures leave their slot untouched. */
int sr_train_store(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t n,
                   const uint32_t *slot, void *store, uint32_t n_slots, uint32_t stride_bytes, uint32_t *status);

/* ------------------------------------------------------------------ batched recognition
 * B capture buffers of buf_len samples, buffer b at pcm + b*pcm_stride (in samples).
 * Outputs (each may be NULL except results):
 *   results[B], scores[B*K] (cur_dis of every slot, main.c:283), mfcc[B*max_frames*n_coef]
 *   (frame-major, rows >= frm_num zeroed), vad[B].
 *
 * sr_recognize_batch:      HOST buffers; stages them through HBM (PCIe-inclusive).
 * sr_recognize_batch_dev:  DEVICE buffers already resident in HBM, enqueued on `stream`
 *                          (a hipStream_t; NULL = default stream), asynchronous.
 *                          pcm must be 16-byte aligned and pcm_stride a multiple of 8.
 */
int sr_recognize_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                       sr_result *results, uint32_t *scores, int16_t *mfcc, sr_vad_rec *vad);
int sr_recognize_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                           sr_result *d_results, uint32_t *d_scores, int16_t *d_mfcc, sr_vad_rec *d_vad,
                           void *stream);
/* sr_recognize_batch for HOST callers whose captures are 12-bit ADC codes (ADC.H:7-11: the firmware's converter), packed
 * two samples in three bytes: sample 2i = b[3i] | (b[3i+1] & 0x0F) << 8, sample 2i+1 = b[3i+1] >> 4 | b[3i+2] << 4; row b
 * starts at packed + b*row_stride_bytes and holds ceil(buf_len / 2) * 3 bytes.  The host-buffer path is PCIe-bound
 * (INTEGRATION.md 2); this moves 25 % fewer bytes and unpacks on the device.  Same results as the u16 call on the same codes. */
int sr_recognize_batch_packed12(sr_engine *h, const uint8_t *packed, uint64_t row_stride_bytes, uint32_t buf_len, uint32_t B,
                                sr_result *results, uint32_t *scores, int16_t *mfcc, sr_vad_rec *vad);

/* Multi-segment recognition: every segment the VAD returns (up to max_seg, VAD.H:4) is matched like segment 0.
 * The firmware stops at segment 0 (main.c:268); this is an extension with segment-major outputs:
 * results[s*B + b], scores[(s*B + b)*K + k]; a segment that does not exist has status SR_ST_VAD_FAIL. */
int sr_recognize_segments_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                                sr_result *results, uint32_t *scores, sr_vad_rec *vad);
int sr_recognize_segments_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t buf_len,
                                    uint32_t B, sr_result *d_results, uint32_t *d_scores, sr_vad_rec *d_vad,
                                    void *stream);

/* stage-level entry points on DEVICE buffers (same kernels the full path launches) */
int sr_vad_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                     sr_vad_rec *d_vad, void *stream);
int sr_mfcc_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t B, const sr_vad_rec *d_vad,
                      int16_t *d_mfcc, void *stream);
int sr_dtw_batch_dev(sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad, uint32_t B, uint32_t *d_scores,
                     sr_result *d_results, void *stream);

/* stage-level entry points on HOST buffers (copy in, launch, copy out) */
int sr_vad_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                 sr_vad_rec *vad);
/* MFCC of segment [start[b], end[b]) of buffer b with mid value mid[b]; frm_num[b] receives the frame count */
int sr_mfcc_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                  const int32_t *start, const int32_t *end, const uint32_t *mid, int16_t *mfcc, uint32_t *frm_num);
/* The same with PER-RECORD failure, as get_mfcc has it (MFCC.C:102-107: a segment shorter than a frame underflows the u32
 * frame count, which then exceeds vv_frm_max -> frm_num = 0): a bad record yields frm_num[b] = 0, an all-zero MFCC record
 * and status[b] = SR_ST_SEG_OOB (start < 1, end beyond the buffer, end < start) or SR_ST_MFCC_FAIL (shorter than a frame,
 * more than max_frames frames); the other records of the batch are processed.  sr_mfcc_batch is this call with
 * status == NULL.  (One deviation from the literal u16 arithmetic of MFCC.C:102: a segment shorter than a frame always
 * fails; the wrapped count 13107 would pass a cap of 13107 frames or more and read far beyond the segment.) */
int sr_mfcc_batch_status(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                         const int32_t *start, const int32_t *end, const uint32_t *mid, int16_t *mfcc, uint32_t *frm_num,
                         uint32_t *status);
/* all-pairs greedy DTW of B feature sequences (in_mfcc[b*max_frames*n_coef], in_frames[b]) against the store */
int sr_dtw_batch(sr_engine *h, const int16_t *in_mfcc, const uint32_t *in_frames, uint32_t B, uint32_t *scores,
                 sr_result *results);
/* get_mdl (DTW.C:217-296; present in the reference but not called by main.c): template averaging.  For each of
 * the P pairs the greedy dtw() path of (in1 as input, in2 as model) is walked and the start point plus every point
 * the walk moves to contributes one merged frame, the per-coefficient get_mean (DTW.C:195-205) (a + b) / 2 in int
 * arithmetic.  in1 [P][rows1][12], in2 [P][rows2][12] (host); rows past n are read as by dtw() (one slack row).
 * mdl [P][mdl_rows][12] receives the merged templates (rows past the merged length are zero);
 * mdl_frames[p] = number of merged frames = step (DTW.C:293) -- when it exceeds mdl_rows only the first mdl_rows
 * frames were stored (the reference would overrun its 119-frame record); dis[p] = dis/step.  Pairs whose length ratio
 * is outside 1/2..2 give dis = 0xFFFFFFFF, mdl_frames = 0 and an all-zero mdl (DTW.C:236-239). */
int sr_get_mdl_batch(sr_engine *h, const int16_t *in1, const uint32_t *n1, uint32_t rows1, const int16_t *in2,
                     const uint32_t *n2, uint32_t rows2, uint32_t P, int16_t *mdl, uint32_t mdl_rows,
                     uint32_t *mdl_frames, uint32_t *dis);

/* OPT-IN, NON-REFERENCE scorer: full dynamic-programming DTW (anti-diagonal wavefront across the 64-lane wave,
 * template staged in LDS) with the reference's parallelogram (dtw_limit) and local distance (get_dis):
 *   D(1,1)=d(1,1); D(x,y)=d(x,y)+min(D(x-1,y-1),D(x-1,y),D(x,y-1)); score = D(in,mdl)/(in+mdl), dis_err if gated/unreachable.
 * The reference's dtw() is a greedy walk (DTW.C:150-188), so these scores differ from dtw()'s by design and are
 * never used by sr_recognize_* or the dtw symbol. */
/* Kernel: a band-limited anti-diagonal wavefront, `lanes` lanes of a wave per (utterance, template) pair (strips of that
 * many utterance frames; the value from the left moves by DPP, strip boundaries through LDS, the template staged in
 * LDS); sr_set_dp_lanes chooses 4, 8 or 16 (0 = automatic: 8 while three of its workgroups fit a CU's LDS, else 16),
 * or 1 = the first version (one wave per pair, 64-column sweeps of the whole rectangle), which also serves stores the
 * band kernel cannot stage.  All variants give identical scores. */
int sr_set_dp_lanes(sr_engine *h, uint32_t lanes);
int sr_dtw_dp_batch(sr_engine *h, const int16_t *in_mfcc, const uint32_t *in_frames, uint32_t B, uint32_t *scores);
int sr_dtw_dp_batch_dev(sr_engine *h, const int16_t *d_mfcc, const uint32_t *d_in_frames, const sr_vad_rec *d_vad,
                        uint32_t B, uint32_t *d_scores, void *stream);
/* EXTENSION, NO REFERENCE COUNTERPART (the thesis that accompanies the reference, p.32, lists difference cepstra as
 * future work; the firmware computes none): delta MFCC by the standard two-frame regression over the s16 rows,
 *   delta[t][c] = ((m[t+1][c] - m[t-1][c]) + 2*(m[t+2][c] - m[t-2][c])) / 10,
 * row indices clamped to [0, frames-1], s32 arithmetic, division truncating toward zero; rows >= frames are zero.
 * mfcc / delta: [B][max_frames][n_coef].  Never used by the recognition path or the reference-compatible symbols. */
int sr_delta_mfcc_batch(sr_engine *h, const int16_t *mfcc, const uint32_t *frames, uint32_t B, int16_t *delta);
int sr_delta_mfcc_batch_dev(sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad /* or */, const uint32_t *d_frames,
                            uint32_t B, int16_t *d_delta, void *stream);
/* generic 1024-point Q15 FFT of n independent packed-complex arrays (re = low half, im = high half) */
int sr_fft_q15_batch(sr_engine *h, const uint32_t *in, uint32_t *out, uint32_t n);
/* batched forms of the two small scalar symbols below (the same kernels get_dis() / dtw_limit() launch with n = 1):
 * get_dis (DTW.C:45-62) on n pairs of 12-coefficient rows a[i*12..], b[i*12..];
 * dtw_limit (DTW.C:76-109) on n points xy[2i] = x, xy[2i+1] = y for the file statics a dtw() call of in_frames against
 * mdl_frames leaves behind (DTW.C:129-130, 141-142); out[i] = 1: the point lies outside the parallelogram. */
int sr_get_dis_batch(sr_engine *h, const int16_t *a, const int16_t *b, uint32_t n, uint32_t *out);
int sr_dtw_limit_batch(sr_engine *h, const uint16_t *xy, uint32_t n, uint32_t in_frames, uint32_t mdl_frames, uint8_t *out);

/* ------------------------------------------------------------------ multi-GPU (one process, several MI355X)
 * Utterances are sharded over the devices (B_per_dev each), templates are replicated, the argmin is local, and the
 * path's single exchange step is ONE RCCL all-gather of the per-template score matrix over xGMI: after the call every
 * device holds u32 scores[n_dev*B_per_dev][K] in global utterance order -- what the firmware's slot scan
 * (main.c:279-291: cur_dis of every slot) produces, for every utterance of the job.  RCCL (librccl.so.1) is bound at
 * run time; without it sr_multi_create fails with SR_ERR_NO_DEVICE.  Processes that run one rank per GPU (MPI,
 * torchrun) keep one sr_engine each and use sr_allgather_scores on their own communicator. */
typedef struct sr_multi sr_multi;
int sr_multi_create(const sr_config *cfg, const int *devices, uint32_t n_dev, sr_multi **out); /* cfg->device ignored */
void sr_multi_destroy(sr_multi *m);
uint32_t sr_multi_num_devices(const sr_multi *m);
sr_engine *sr_multi_engine(sr_multi *m, uint32_t i); /* the engine of devices[i] */
int sr_multi_set_templates(sr_multi *m, const void *store, uint32_t n_slots, uint32_t stride_bytes);
int sr_multi_set_templates_dense(sr_multi *m, const int16_t *mfcc, const uint32_t *frames, const uint8_t *valid,
                                 uint32_t n_templates, uint32_t tpl_stride);
/* device-resident shards: d_pcm[i], d_results[i] (B_per_dev records) and d_scores_all[i] (n_dev*B_per_dev*K words) live
 * on devices[i]; asynchronous on streams[i] (hipStream_t; streams == NULL: internal streams, returns when drained) */
int sr_multi_recognize_dev(sr_multi *m, const uint16_t *const *d_pcm, uint64_t pcm_stride, uint32_t buf_len,
                           uint32_t B_per_dev, sr_result *const *d_results, uint32_t *const *d_scores_all,
                           void *const *streams);
/* host buffers: shards, uploads, recognises, gathers; results[B], scores[B*K] (optional, read back from devices[0]) */
int sr_multi_recognize(sr_multi *m, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                       sr_result *results, uint32_t *scores);
/* the exchange step alone on a caller-owned ncclComm_t: d_all[n_ranks*count] <- all ranks' d_scores[count] */
int sr_allgather_scores(void *nccl_comm, const uint32_t *d_scores, uint32_t *d_all, uint64_t count, void *stream);

/* ------------------------------------------------------------------ measurement hooks (bench.py)
 * sr_recognize_batch_dev cuts a large batch into chunks (at least min_chunk = 4096 utterances each, at most
 * max_chunks = 12; one chunk per stream once the store holds 256 templates or more) and runs them on streams = 3 (max 4)
 * internal streams forked from / joined to the
 * caller's stream, so each kernel is launched once per chunk and kernels of different chunks overlap
 * (sr_set_pipeline changes the three numbers; streams = 1 keeps everything on the caller's stream).  The library reads
 * no tuning knob from the environment (the only environment variable it honours is SR_RCCL_LIBRARY, the path of the
 * collective library).
 * With profiling on, every kernel launch is bracketed with hipEvents on the stream it is launched on;
 * sr_get_stage_ms synchronises and returns, averaged over everything recorded since sr_set_profiling(h, 1):
 * ms[0] VAD, ms[1] MFCC (frame kernel), ms[2] DTW, ms[3] argmin = duration of ONE launch of that kernel (under
 * overlap with the other chunks' kernels), ms[4] = one whole call on the caller's stream (fork -> join).
 * sr_get_stage_launches: launches of each kernel per call (= chunks). */
int sr_set_pipeline(sr_engine *h, uint32_t streams, uint32_t min_chunk, uint32_t max_chunks); /* same knobs at run time */
/* Small launches -- one capture against the store (spch_recg, main.c:276-295), one dtw() / get_mfcc() call, a handful of
 * captures -- are latency-bound: every stage of the path is a serial chain per capture (VAD's state across frames, a wave's
 * frames, dtw's walk), and a few of them leave the GPU idle.  The engine then spends the idle width instead:
 *   VAD   fewer than 1024 captures: a workgroup of four waves per capture (k_vad_wide) instead of one wave;
 *   MFCC  fewer than 1024 work items of 64 frames: 16 or 4 frames per workgroup instead (reference front end);
 *   DTW   up to 320 000 / max_frames pairs per launch (2 689 at the firmware's 119 frames, 1 000 at 320): every pair gets
 *         its own workgroup (k_dtw_cells: all points of dtw_limit's band evaluated at once, then one lane follows the
 *         precomputed moves, and the last pair of an utterance does the slot scan), provided the band fits a workgroup's
 *         LDS beside the rows (frame cap and templates up to 400 frames; a pair whose band is larger than the LDS is walked
 *         literally by its workgroup);
 *         beyond that, up to two "rounds" of what the chip holds at once (65 536 pairs at the firmware's shapes): FOUR LANES
 *         PER PAIR (k_dtw_quad: the three candidates of a step evaluated by three lanes of a quad at the same time, minimum
 *         and move in one cross-lane reduction, both sequences staged in LDS), because the batch kernel's one-lane-per-pair
 *         walk takes the same ~126 us for 5 000 pairs as for 160 000;
 *   host  sr_recognize_batch with at most 256 KB of captures: pinned staging, results written to pinned host memory.
 * Same results bit for bit (tests run the DTW / VAD / recognition cases in every mode).  One 16 000-sample capture against
 * 80 slots: 242 us -> 64 us per spch_recg call on an otherwise idle MI355X (profiles/, latency block of bench.py).
 * The in-kernel slot scan of the one-workgroup-per-pair form counts finished pairs in per-engine counters, so it is available to
 * ONE caller stream per engine -- the first that launches it (the internal stream of the host-buffer calls counts as one);
 * small launches on any other stream of the same engine run the separate slot-scan kernel instead (same results, one more
 * launch).  Calls on one engine from several host threads at once remain forbidden as everywhere in this API.
 * mode 0 = automatic (default), 1 = never (always the batch kernels), 2 = always (the one-workgroup-per-pair DTW form whenever
 * the rectangle fits), 3 = the four-lanes-per-pair DTW form whenever the sequences fit (VAD / MFCC / host side as in mode 0),
 * whatever the launch size: for tests and measurements. */
int sr_set_small_launch(sr_engine *h, int mode);
int sr_set_profiling(sr_engine *h, int on);
int sr_get_stage_ms(sr_engine *h, float ms[5]);
int sr_get_stage_launches(sr_engine *h, uint32_t *launches_per_call);

/* Development and test hooks -- NOT part of the production surface and NOT in the product library: libsr_engine.so is
 * built without them (every name is refused with SR_ERR_BAD_ARG, sr_testing_build() == 0); the test suite and the tuning
 * sweeps load libsr_engine_testing.so, the same sources compiled with -DSR_TESTING (csrc/Makefile).
 * Process-global integer knobs, 0 = default:
 *   "dtw_u", "dtw_tie_g", "dtw_kc"   force the staged DTW kernel's workgroup geometry (read when a template store is set;
 *                                    a forced combination that does not fit the LDS / the grid is ignored)
 *   "mfcc_grid"                      workgroups of the frame kernel (read by sr_create)
 *   "dtw_debug"                      print the DTW geometry when a store is set
 *   "perturb_log_thr", "log_thr_from_host"   exercise / bypass the shipped log-step-table check (sr_log_table_mismatches)
 *   "mag_cheap_off"                  sr_create behaves as if its device sweep of the cheap magnitude form had failed
 *   "multi_allow_dup"                sr_multi_create accepts one device several times; honoured only when SR_RCCL_LIBRARY
 *                                    names the collective library explicitly (1-GPU tests over the in-process RCCL double)
 * Unknown names return SR_ERR_BAD_ARG. */
int sr_dev_hook(const char *name, int64_t value);
int sr_testing_build(void);

/* diagnostics, host-only (touches no device): the launch geometry the staged DTW kernel would use for a store of n_templates
 * and a frame cap of max_frames: out[0] = utterances per workgroup (0 = generic kernel), out[1] = templates per workgroup (the
 * store is walked in ceil(n_templates / out[1]) chunks), out[2] = tie-threshold table entries staged in LDS, out[3] = LDS bytes
 * per workgroup, out[4] = workgroups that fit one CU's 160 KiB at gfx950's allocation granule of 1280 bytes. */
int sr_dtw_geometry(uint32_t n_templates, uint32_t max_frames, uint32_t out[5]);

/* diagnostics: the path's non-integer device functions swept directly:
 * out[3i] = (u32)(log((double)x)*100) (MFCC.C:168), out[3i+1] = (u32)sqrtf((float)x) (DTW.C:59),
 * out[3i+2] = (u32)(sqrtf((float)(s32)(x & 0x7fffffff))*10) (MFCC.C:56-58) */
int sr_math_diag(sr_engine *h, const uint32_t *in, uint32_t *out, uint32_t n);
/* diagnostics: the fused Mel filterbank term of the frame kernel -- one v_mul_hi_u32 of E << 4 with ceil(tri * 2^28 / 100) --
 * against the reference's u32 expression frq_spct[i]*tri[i]/(tri_top/10) (MFCC.C:139-161) for every weight tri in
 * [tri_lo, tri_hi) and every energy E in [0, e_max]: mismatches[tri - tri_lo] = number of E where they differ (plus 2^40 if
 * the weight recovered from the multiplier for the literal form is wrong).  The kernel takes the fused form while every E
 * of a frame is <= 2 684 354 = floor(2^28 / 100). */
int sr_mel_term_sweep(sr_engine *h, uint32_t tri_lo, uint32_t tri_hi, uint32_t e_max, uint64_t *mismatches);
/* diagnostics: the cheap magnitude form the frame kernel may use on quiet frames -- (u32)(v_sqrt_f32((float)n) * 10) -- against the
 * exact one for every n in [0, n_max]: out[0] = number of n where they differ, out[1] = the smallest such n (0xFFFFFFFF: none).
 * With bit 31 of n_max set the sweep compares what the DTW kernel's small-root form uses, floor(v_sqrt_f32((float)d)), with the
 * exact (u32)sqrtf((float)d) of DTW.C:59 for every d in [0, n_max & 0x7fffffff]. */
int sr_mag_fast_sweep(sr_engine *h, uint32_t n_max, uint64_t out[2]);
/* diagnostics: fill the whole local data share of every compute unit with a seeded pattern (asynchronous on `stream`; one
 * workgroup per CU at a time, each taking the device's full per-CU LDS).  *bytes_per_cu (optional) receives the bytes
 * each workgroup filled (163840 on MI355X).  The test suite launches it between calls: a kernel that reads LDS it has not
 * written itself gets the same values only as long as the CU's previous tenant was a workgroup of the same kernel. */
int sr_lds_poison(sr_engine *h, uint32_t seed, void *stream, uint32_t *bytes_per_cu);
/* diagnostics: per-utterance ballots of the VAD "loud" decision (VAD.C:164), 63 frames per 64-bit word, 16 words */
int sr_vad_debug_masks(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                       sr_vad_rec *vad, uint64_t *masks);

/* ------------------------------------------------------------------ reference-compatible scalar symbols
 * Exact reference signatures (u8/u16/u32/s16 = stdint fixed widths, stm32f10x.h:421-439).
 * Each one replaces the reference function cited; all run on the GPU through an implicit
 * default engine (reference constants, max_frames 119). */
typedef struct {
    uint32_t mid_val;
    uint16_t n_thl;
    uint16_t z_thl;
    uint32_t s_thl;
} atap_tag; /* VAD.H:10-16 */
typedef struct {
    uint16_t *start;
    uint16_t *end;
} valid_tag; /* VAD.H:18-22 */
#define SR_VV_FRM_MAX 119 /* MFCC.H:15-16 */
#pragma pack(push, 1)
typedef struct {
    uint16_t save_sign;
    uint16_t frm_num;
    int16_t mfcc_dat[SR_VV_FRM_MAX * 12];
} v_ftr_tag; /* MFCC.H:18-25, 2860 bytes */
#pragma pack(pop)

void noise_atap(const uint16_t *noise, uint16_t n_len, atap_tag *atap);                          /* VAD.H:24, VAD.C:22 */
void VAD(const uint16_t *vc, uint16_t buf_len, valid_tag *valid_voice, atap_tag *atap_arg);      /* VAD.H:25, VAD.C:97 */
void get_mfcc(valid_tag *valid, v_ftr_tag *v_ftr, atap_tag *atap_arg);                           /* MFCC.H:27, MFCC.C:86 */
uint32_t *fft(int16_t *dat_buf, uint16_t buf_len);                                               /* MFCC.C:27 */
void cr4_fft_1024_stm32(void *pssOUT, void *pssIN, uint16_t Nbin);                               /* MFCC.C:12, .s:219 */
uint32_t get_dis(int16_t *frm_ftr1, int16_t *frm_ftr2);                                          /* DTW.C:45 */
uint8_t dtw_limit(uint16_t x, uint16_t y);                                                       /* DTW.C:76 */
/* dtw(): models are cached by content on the device (up to 128 records, least recently used replaced); a model that is
 * new to the cache costs one store upload, an input record that is new one launch against every cached model.  Callers
 * whose model changes on every call (random pair sweeps) should use sr_dtw_batch.  Like every symbol of this section:
 * one implicit engine, file-scope state, NOT thread-safe (as the reference: DTW.C:65-68). */
uint32_t dtw(v_ftr_tag *ftr_in, v_ftr_tag *frt_mdl);                                             /* DTW.H:7, DTW.C:120 */
uint8_t *spch_recg(uint16_t *v_dat, uint32_t *mtch_dis);                                         /* main.c:249 */
void get_mean(int16_t *frm_ftr1, int16_t *frm_ftr2, int16_t *mean);                              /* DTW.C:195 */
uint32_t get_mdl(v_ftr_tag *ftr_in1, v_ftr_tag *ftr_in2, v_ftr_tag *ftr_mdl);                    /* DTW.C:217 */
/* BASELINE.json's north-star spellings; they do not exist in the reference -> aliases of get_mfcc */
void GetMfcc(valid_tag *valid, v_ftr_tag *v_ftr, atap_tag *atap_arg);
void MFCC_Comp(valid_tag *valid, v_ftr_tag *v_ftr, atap_tag *atap_arg);

/* The firmware reads its template store at a fixed flash address (Flash.H:19-20) and returns a
 * pointer into commstr[] (main.c:31,295).  The scalar spch_recg needs both handed over: */
int sr_compat_set_templates(const void *store, uint32_t n_slots, uint32_t stride_bytes);
int sr_compat_set_labels(const uint8_t *labels, uint32_t n_labels, uint32_t label_stride, uint32_t ftr_per_comm);
sr_engine *sr_compat_engine(void);
/* diagnostics: out[0] = uploads of dtw()'s model store so far, out[1] = DTW launches, out[2] = models cached */
void sr_compat_dtw_stats(uint32_t out[3]);

#ifdef __cplusplus
}
#endif
#endif /* SR_ENGINE_H */
