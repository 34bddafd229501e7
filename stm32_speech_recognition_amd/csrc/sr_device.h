// Kernel argument blocks and launch entry points shared by the host engine and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/sr_engine.h"

namespace sr {

// Development / test hooks (C ABI: sr_dev_hook): process-global integer knobs, 0 = default behaviour.  They exist ONLY in
// the -DSR_TESTING build (libsr_engine_testing.so, used by the test suite and the tuning sweeps): in the product library
// dev_hook() is the constant 0, every branch on it folds away at compile time, and sr_dev_hook refuses every name.
// Nothing in the product depends on the environment, except SR_RCCL_LIBRARY (the path of the collective library, a
// deployment setting).
enum DevHook {
    kHookDtwU,            // "dtw_u":       force U utterances per k_dtw_lds workgroup (read when a template store is set)
    kHookDtwTieG,         // "dtw_tie_g":   force the staged tie-table size
    kHookDtwKc,           // "dtw_kc":      cap the templates per k_dtw_lds workgroup
    kHookMfccGrid,        // "mfcc_grid":   workgroups of the frame kernel (read by sr_create)
    kHookPerturbLogThr,   // "perturb_log_thr": move host-built log step m by one (exercises the shipped-table check)
    kHookLogThrFromHost,  // "log_thr_from_host": keep the host's log step table even where it differs from the shipped one
    kHookMultiAllowDup,   // "multi_allow_dup": sr_multi_create accepts one device several times (1-GPU tests over the RCCL double)
    kHookDtwDebug,        // "dtw_debug":   print the k_dtw_lds geometry when a store is set
    kHookCellsLiteral,    // "cells_literal": k_dtw_cells walks every pair literally (the fallback of walks that leave the band)
    kHookMagCheapOff,     // "mag_cheap_off": sr_create behaves as if the device sweep of the cheap magnitude form had failed (bound 0)
    kHookCount
};
#ifdef SR_TESTING
int64_t dev_hook(DevHook h);
#else
constexpr int64_t dev_hook(DevHook) { return 0; }
#endif

// device-resident constant tables (see sr_tables.h)
struct DevTables {
    const uint16_t *hamm;      // [160]
    const uint16_t *tri_even;  // [512]
    const uint16_t *tri_odd;   // [512]
    const uint16_t *tri_cen;   // [24]
    const int8_t *dct;         // [288]
    const uint32_t *tw_a;      // [1020]
    const uint32_t *tw_b;      // [1020]
    const uint32_t *log_thr;   // [2220]
    const uint32_t *tri_even32;  // [bins] the same weights widened to 32 bits (16-byte vector loads in k_mfcc)
    const uint32_t *tri_odd32;
    const uint32_t *tri_even_m;  // [bins] mel_fused_multiplier(tri_even[i]) (sr_tables.h): k_mfcc's filterbank terms are one v_mul_hi_u32
    const uint32_t *tri_odd_m;
    const uint32_t *w512_a;    // [256]  EXTENSION front end only
    const uint32_t *w512_b;    // [256]
    const int8_t *tie_delta;   // [kTieMax] DTW tie thresholds: T(g) = g*(g+2) + tie_delta[g] (sr_tables.h)
    const uint32_t *hamm_pk;   // [frame_len / 2] hamm[2p] | hamm[2p+1] << 16 (k_mfcc_ext reads its window weights as pairs)
};

struct VadArgs {
    const uint16_t *pcm;  // [B][pcm_stride], 16-byte aligned rows
    uint64_t pcm_stride;  // samples
    uint32_t buf_len;     // samples scanned by VAD (VcBuf_Len)
    uint32_t noise_len;   // atap_len
    uint32_t atap_frm;    // atap_frm_len (240)
    uint32_t max_frames;  // vv_frm_max
    uint32_t max_seg;     // max_vc_con
    uint32_t B;
    sr_vad_rec *vad;      // [B]
    const sr_atap *atap_in;  // optional [B]: use these thresholds instead of running noise_atap
    uint64_t *dbg_masks;     // optional [B][16]: per-round ballot of "loud" frames (diagnostics)
    uint32_t frame_len;      // 160 (reference), 320 (extension) or one of the generic front end's framings (k_vad_gen.hip)
    uint32_t v_durmin;       // VAD.C:72-73: 80 ms / (frame_time - frame_mov_t) frames (8 at 20 / 10 ms)
    uint32_t s_durmax;       // VAD.C:74-75: 110 ms / (frame_time - frame_mov_t) frames (11)
    uint32_t wide;           // 1: a workgroup of four waves per capture (k_vad_wide.hip; small launches), 0: one wave per capture
};

struct MfccArgs {
    const uint16_t *pcm;
    uint64_t pcm_stride;
    uint32_t B;
    uint32_t max_frames;
    const sr_vad_rec *vad;  // segment 0 + mid_val + frm_num per utterance
    int16_t *mfcc;          // [B][max_frames][12]
    uint32_t tiles;         // frame tiles per utterance
    uint32_t small_tiles;   // 0: 64-frame work items; 1 / 2: the 16- / 4-frame forms of k_mfcc for underfilled launches (tiles counts those)
    uint32_t n_items;       // B * tiles
    uint32_t grid_cap;      // resident workgroups of k_mfcc on this device (0 = default)
    uint32_t mag_cheap_max; // k_mfcc: largest re^2 + im^2 whose magnitude may take the uncorrected v_sqrt_f32 (kMagCheapMax where sr_create's sweep of this device confirmed it, else 0)
    uint32_t frame_len;     // 160 -> k_mfcc (reference front end), 320 -> k_mfcc_ext (extension)
    uint32_t generic;       // 1 -> k_mfcc_gen (GENERIC front end): the four fields below are only read there
    uint32_t hop, n_mel, n_coef;
    DevTables t;
};

struct DtwArgs {
    const int16_t *mfcc;      // [B][max_frames][12]
    const sr_vad_rec *vad;    // frm_num / status per utterance (or NULL with in_frames)
    const uint32_t *in_frames;  // optional [B] explicit frame counts (stage-level API)
    uint32_t B;
    uint32_t max_frames;
    const int16_t *tpl;       // [K][tpl_stride]
    const uint32_t *tpl_frames;
    const uint8_t *tpl_valid;
    uint32_t K;
    uint32_t tpl_stride;      // int16 elements per template
    uint32_t tpl_rows;        // rows allocated per template
    uint32_t *scores;         // [B][K]
    sr_result *results;       // [B] (argmin kernel)
    // length-sorted, row-interleaved copy of the store for k_dtw_lds (NULL -> generic kernel)
    const void *tplR;             // [tpl_rows][K] 32-byte rows: 12 x s16 | u32 squared norm | pad  (48-byte rows of 16 x s16 when n_coef > 12)
    const uint32_t *tpl_frames_s; // [K]
    const uint32_t *tpl_orig;     // [K]
    uint32_t lds_u;               // utterances per k_dtw_lds workgroup (0 -> generic kernel), see dtw_lds_pick_u
    uint32_t lds_bytes;           // dynamic LDS of k_dtw_lds for that choice
    const int8_t *tie_delta;      // DevTables::tie_delta
    uint32_t tie_g;               // entries of it the workgroup stages in LDS (a multiple of 1024, <= kTieMax)
    uint32_t lds_kc;              // templates per k_dtw_lds workgroup (dtw_lds_pick_u); the store is walked in K / lds_kc chunks
    uint32_t n_coef;              // s16 per feature row: 12 everywhere except the GENERIC front end (1..16)
    uint32_t dp_lanes;            // k_dtw_dp only: lanes per pair of the band kernel (4 / 8 / 16; 0 = default 8; 1 = k_dtw_dp_wave64)
    uint32_t *pair_count;         // k_dtw_cells only: [B] zeroed counters of finished pairs (the last one does the slot scan); may be NULL
    uint32_t cells_points;        // k_dtw_cells only: most band points of any pair of this store (dtw_cells_max_points; 0 = kernel not usable)
    uint32_t tpl_neg2_ok;         // k_dtw_quad only: every coefficient of the store lies in [-16383, 16384], so -2 * coefficient fits s16 (checked when the store is set)
    uint32_t cells_literal;       // k_dtw_cells only: development hook "cells_literal" -- every pair takes the literal fallback walk
    uint32_t dev_cus, dev_lds_cu, dev_lds_wg;  // compute units / LDS bytes per CU / LDS bytes one workgroup may take on this device (sr_create); 0 = MI355X's 256 / 160 KiB / 160 KiB
};

// get_mdl (DTW.C:217-296): P independent pairs
struct GetMdlArgs {
    const int16_t *in1;     // [P][rows1][12]
    const uint32_t *n1;     // [P] frames of in1 ("in" role)
    uint32_t rows1;
    const int16_t *in2;     // [P][rows2][12]
    const uint32_t *n2;     // [P] frames of in2 ("mdl" role)
    uint32_t rows2;
    uint32_t P;
    int16_t *mdl;           // [P][mdl_rows][12] merged templates
    uint32_t mdl_rows;
    uint32_t *mdl_frames;   // [P] step count = frames of the merged template (may exceed mdl_rows), 0 on dis_err
    uint32_t *dis;          // [P] dis/step or dis_err
};
void launch_get_mdl(const GetMdlArgs &a, hipStream_t s);

void launch_vad(const VadArgs &a, hipStream_t s);
void launch_vad_wide(const VadArgs &a, hipStream_t s);  // k_vad_wide.hip
bool vad_framing_supported(uint32_t frame_len, uint32_t hop);  // the VAD kernel is instantiated per framing
void launch_select_segment(const sr_vad_rec *in, sr_vad_rec *out, uint32_t B, uint32_t seg_idx, uint32_t max_frames,
                           uint32_t frame_len, uint32_t hop, hipStream_t s);
void launch_mfcc(const MfccArgs &a, hipStream_t s);
void launch_mfcc_gen(const MfccArgs &a, hipStream_t s);  // GENERIC front end (k_mfcc_gen.hip)
uint32_t mfcc_frames_per_tile(uint32_t frame_len);        // frames one work item of the frame kernel covers
uint32_t mfcc_frames_per_tile_small(uint32_t frame_len, uint32_t which);  // ... of its forms for underfilled launches (MfccArgs::small_tiles - 1)
uint32_t mfcc_resident_workgroups(uint32_t frame_len);  // occupancy x CUs on the current device
void launch_dtw(const DtwArgs &a, hipStream_t s);
// utterances per k_dtw_lds workgroup for K templates / max_frames rows (0 = use the generic kernel); tuning
// override: development hooks dtw_u / dtw_kc / dtw_tie_g (sr_dev_hook), read when the template store is set
// (row_words: packed coefficient pairs per feature row of the staged kernel's form: 6 = up to 12 coefficients, 8 = 13..16)
uint32_t dtw_lds_pick_u(uint32_t K, uint32_t max_frames, size_t *lds_bytes, uint32_t *tie_g, uint32_t *kc, uint32_t row_words = 6);
void launch_argmin(const DtwArgs &a, hipStream_t s);
void launch_dtw_dp(const DtwArgs &a, hipStream_t s);  // opt-in non-reference full-DP scorer
// generic complex 1024-point Q15 FFT, n arrays (cr4_fft_1024_stm32 semantics)
void launch_fft_q15(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables &t, hipStream_t s);
// magnitude*10 of bins 0..511 of zero-padded real frames: fft() of MFCC.C:27-62.
// frames: [n][160] int16; mag: [n][512]; also writes raw FFT words of bins 512..1023 to raw_hi if non-null
void launch_fft_mag(const int16_t *frames, uint32_t len, uint32_t *mag, uint32_t *raw_hi, uint32_t n,
                    const DevTables &t, hipStream_t s);
// EXTENSION: delta cepstra of B records (frame counts from vad[] or, when non-null, frames[])
void launch_delta_mfcc(const int16_t *mfcc, const sr_vad_rec *vad, const uint32_t *frames, uint32_t B, uint32_t max_frames,
                       uint32_t n_coef, int16_t *delta, hipStream_t s);
// host transport: rows of 12-bit codes packed 2 samples / 3 bytes (row_bytes a multiple of 12) -> u16 rows (out_stride samples)
void launch_unpack12(const void *packed, uint64_t row_bytes, uint16_t *out, uint64_t out_stride, uint32_t buf_len, uint32_t B,
                     hipStream_t s);
// diagnostics: log / sqrt device functions swept directly (see k_math_diag)
void launch_math_diag(const uint32_t *in, uint32_t *out, uint32_t n, const DevTables &t, hipStream_t s);
// diagnostics: fused Mel filterbank term vs the reference's u32 expression, weights [tri_lo, tri_lo + n_tri), E <= e_max
void launch_mel_term_sweep(uint32_t tri_lo, uint32_t n_tri, uint32_t e_max, unsigned long long *bad, hipStream_t s);
int launch_lds_poison(uint32_t seed, hipStream_t s);  // k_misc.hip; returns the bytes of LDS each workgroup filled, < 0 on failure
void launch_mag_fast_sweep(uint32_t n_max, unsigned long long *bad, uint32_t *first_bad, hipStream_t s);
// get_dis (DTW.C:45-62) for n frame pairs
void launch_get_dis(const int16_t *a, const int16_t *b, uint32_t *out, uint32_t n, hipStream_t s);
// dtw_limit (DTW.C:76-109) for n points with explicit statics
void launch_dtw_limit(const uint16_t *xy, uint8_t *out, uint32_t n, int X1, int X2, int in_n, int mdl_n, hipStream_t s);

}  // namespace sr
