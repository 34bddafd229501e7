// Host side of the C ABI declared in include/sr_engine.h: device tables, template store, staging
// buffers, kernel sequencing on a HIP stream.  No CPU implementation of the recognition path exists
// in this library; every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sr_device.h"
#include "sr_dtw_cells.h"
#include "sr_tables.h"

namespace sr {

static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int set_error(int code, const std::string &msg) { return fail(code, msg); }  // for the other translation units

#ifdef SR_TESTING
static std::atomic<int64_t> g_hooks[kHookCount];
int64_t dev_hook(DevHook h) { return g_hooks[h].load(std::memory_order_relaxed); }
static const char *const kHookNames[kHookCount] = {"dtw_u", "dtw_tie_g", "dtw_kc", "mfcc_grid", "perturb_log_thr",
                                                   "log_thr_from_host", "multi_allow_dup", "dtw_debug", "cells_literal"};
#endif
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(SR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)


// Every entry point runs on the engine's device and puts the caller's current device back afterwards (a
// single-process multi-GPU caller -- or PyTorch on another ordinal -- keeps its own current device).
struct DeviceGuard {
    int prev = -1;
    bool restore = false;
    int enter(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        hipError_t e = hipSetDevice(dev);
        if (e != hipSuccess) return fail(SR_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
        restore = prev >= 0 && prev != dev;
        return SR_OK;
    }
    ~DeviceGuard()
    {
        if (restore) (void)hipSetDevice(prev);
    }
};
#define ENTER_DEVICE(h)                      \
    DeviceGuard dev_guard_;                  \
    do {                                     \
        int rc_dev_ = dev_guard_.enter((h)->device); \
        if (rc_dev_) return rc_dev_;         \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int reserve(size_t count)
    {
        if (count <= n) return SR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) return fail(SR_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        n = count;
        return SR_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace sr

using namespace sr;

// launch sizes below which VAD / the frame kernel take their small-launch forms (captures; work items of 64 frames)
static constexpr uint32_t kVadWideBelow = 1024, kMfccFill = 1024;  // measured crossover ~2 000 captures; work items that fill 256 CUs x 4 (RESULTS.md)
// utterances of one call whose slot scan k_dtw_cells can do itself (one counter each); beyond that k_argmin runs as usual
static constexpr uint32_t kPairCounters = 65536;

struct sr_engine {
    sr_config cfg;
    int device = 0;
    uint32_t noise_len = 0, atap_frm = 0;
    uint32_t mfcc_tile = 64, mfcc_tile_mid = 64, mfcc_tile_small = 64, mfcc_grid_cap = 0;  // frames per k_mfcc work item (batch form / the two forms for underfilled launches), resident workgroups
    uint32_t frame_len = 160, hop = 80;          // 160/80 reference, 320/160 extension, or the generic front end's framing
    uint32_t nc = 12, n_mel = 24;                // s16 per feature row (n_coef), Mel filters
    bool generic = false;                        // GENERIC front end (k_mfcc_gen; k_dtw_lds's 16-wide form when nc > 12)
    uint32_t v_durmin = 8, s_durmax = 11;        // VAD.C:72-75 in frames
    HostTables host;
    DevTables dev{};
    void *table_blob = nullptr;
    // template store, dense layout in HBM
    DevBuf<int16_t> tpl;
    DevBuf<uint32_t> tpl_frames;
    DevBuf<uint8_t> tpl_valid;
    bool tpl_staged_ok = true;     // every coefficient of the store fits the -2*coef rows of tplR
    DevBuf<uint32_t> tplR;         // [rows][K] 32-byte rows (12 x s16 | norm | pad), templates ordered by length
    DevBuf<uint32_t> tpl_frames_s, tpl_orig;
    uint32_t K = 0, tpl_rows = 0, tpl_stride = 0;
    uint32_t dtw_u = 0, dtw_lds = 0, dtw_tie_g = 0, dtw_kc = 0;  // k_dtw_lds geometry for this store (0 = generic kernel)
    uint32_t dp_lanes = 0;         // sr_set_dp_lanes: lanes per pair of the opt-in full-DP scorer (0 = default)
    uint32_t cells_points = 0;     // most band points of any pair of this store (k_dtw_cells' LDS; 0 = not usable)
    std::vector<uint32_t> cells_by_len;  // ... per template length, computed once (dtw_cells_max_points)
    int small_launch = 0;          // sr_set_small_launch: 0 = k_dtw_cells for launches of a few hundred pairs, 1 = never, 2 = whenever it fits
    // scratch used when the caller does not ask for an intermediate (or passes host buffers)
    DevBuf<uint16_t> s_pcm;
    DevBuf<uint8_t> s_pack;   // sr_recognize_batch_packed12: the packed rows as uploaded, before k_unpack12
    DevBuf<sr_vad_rec> s_vad;
    DevBuf<int16_t> s_mfcc;
    DevBuf<uint32_t> s_scores;
    DevBuf<sr_result> s_results;
    DevBuf<uint32_t> s_u32a, s_u32b;
    DevBuf<sr_atap> s_atap;
    DevBuf<sr_vad_rec> s_vad2;
    DevBuf<uint32_t> s_pcnt;  // k_dtw_cells: finished-pair counters per utterance of a call, zero between launches (kPairCounters)
    // host-buffer pipeline (sr_recognize_batch): upload of chunk c+1 overlaps the kernels of chunk c
    hipStream_t st_copy = nullptr, st_comp = nullptr;
    // small host-buffer calls (spch_recg: one capture): pinned staging area for the upload, results written by the kernel
    // straight into pinned host memory -- one stream synchronisation per call instead of a blocking copy each way
    void *pin_buf = nullptr;
    size_t pin_cap = 0;
    bool pin_failed = false;
    std::vector<hipEvent_t> ev_chunk;
    // device-resident pipeline (sr_recognize_batch_dev): the batch is cut into chunks that run on a few internal
    // streams, forked from and joined back to the caller's stream, so that the kernels of different chunks overlap
    // (k_vad / k_dtw_lds waves fill the issue slots k_mfcc leaves idle: 32.0 -> 28.2 ms per 65 536 utterances)
    static constexpr uint32_t kPipeStreams = 4;
    hipStream_t st_pipe[kPipeStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kPipeStreams] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t pipe_streams = 3;             // sr_set_pipeline streams (1 = one chunk on the caller's stream); measured: 2 -> 28.8,
                                           // 3 -> 28.2, 4 -> 30.0 ms per 65 536 utterances (1 -> 32.0)
    uint32_t pipe_min_chunk = 4096;        // sr_set_pipeline min_chunk: utterances per chunk at least (smaller chunks lose more than they gain:
                                           // 4 096 x 10 as two chunks of 2 048: 1.93 ms per step, as one chunk 1.63)
    uint32_t pipe_max_chunks = 12;         // chunks per call at most (sr_set_pipeline); 6 for large stores, see upload_templates
    bool pipe_user_set = false;            // sr_set_pipeline was called: the engine no longer adapts the chunk count to the store
    // profiling (sr_set_profiling / sr_get_stage_ms): events recorded since profiling was switched on
    bool profiling = false;
    std::vector<hipEvent_t> ev;  // 5 per kernel group (chunk): before VAD, MFCC, DTW, argmin, after argmin
    size_t ev_used = 0;          // groups recorded
    std::vector<hipEvent_t> ev_call;  // 2 per call on the caller's stream: before the fork, after the join
    size_t calls_used = 0;
};

static int check_device(int want, int *out_dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SR_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count 0"));
    int dev = want;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= n) return fail(SR_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(SR_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SR_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    *out_dev = dev;
    return SR_OK;
}

extern "C" {

const char *sr_last_error(void) { return g_err.c_str(); }

void sr_default_config(sr_config *c)
{
    c->fs = 8000;
    c->frame_time_ms = 20;
    c->frame_mov_ms = 10;
    c->nfft = 1024;
    c->n_mel = 24;
    c->n_coef = 12;
    c->max_frames = 119;
    c->noise_len_ms = 300;
    c->max_seg = 3;
    c->device = -1;
}

// Two front ends have specialised kernels: the reference's (8 kHz, 160/80 framing, 1024-point FFT, 24 Mel, 12 MFCC) and the
// 16 kHz / 512-point / 40-Mel EXTENSION of BASELINE.json configs[4] (no reference counterpart).  Every other accepted
// configuration runs the GENERIC front end (k_mfcc_gen + the VAD instance of its framing; round 4): the reference's
// compile-time constants (MFCC.H:7-16, VAD.H:4-8, ADC.H:7-11) as run-time values -- any fs that is a multiple of 4000 Hz
// (the VAD kernel reads the 30 ms blocks of noise_atap, VAD.C:48-63, eight samples at a time),
// 1024-point transform, frame_time = 2 * frame_mov with a framing the VAD kernel is instantiated for (frame_len 160, 240,
// 256, 320, 400, 512 samples), an even number of 4..64 Mel filters, 1..16 coefficients.
static int front_end_of(const sr_config *cfg, FrontEnd *fe)
{
    const bool is_ref = cfg->fs == 8000 && cfg->nfft == 1024 && cfg->n_mel == 24;
    const bool is_ext = cfg->fs == 16000 && cfg->nfft == 512 && cfg->n_mel == 40;
    if ((is_ref || is_ext) && cfg->frame_time_ms == 20 && cfg->frame_mov_ms == 10 && cfg->n_coef == 12) {
        *fe = is_ext ? kFrontExt : kFrontRef;
        return SR_OK;
    }
    const char *what = "supported: fs=8000/nfft=1024/24 Mel/12 MFCC (reference), fs=16000/nfft=512/40 Mel/12 MFCC (extension), or the generic "
                       "front end: nfft=1024, fs a multiple of 4000, frame_time_ms = 2*frame_mov_ms with frame_len in {160,240,256,320,400,512}, "
                       "n_mel even 4..64, n_coef 1..16";
    if (cfg->nfft != 1024 || cfg->fs == 0 || cfg->fs % 4000 || cfg->fs > 1000000) return fail(SR_ERR_BAD_CONFIG, what);
    const uint32_t fl = cfg->fs / 1000 * cfg->frame_time_ms, mov = cfg->fs / 1000 * cfg->frame_mov_ms;
    if (cfg->frame_time_ms != 2 * cfg->frame_mov_ms || fl < 2 || fl > 1024 || !vad_framing_supported(fl, fl - mov))
        return fail(SR_ERR_BAD_CONFIG, what);
    if ((cfg->n_mel & 1) || cfg->n_mel < 4 || cfg->n_mel > 64 || cfg->n_coef < 1 || cfg->n_coef > 16) return fail(SR_ERR_BAD_CONFIG, what);
    *fe = FrontEnd{(int)cfg->fs, (int)fl, (int)(fl - mov), 1024, 512, (int)cfg->n_mel, (int)cfg->n_coef, true};
    return SR_OK;
}

int sr_log_table_mismatches(void) { return log_table_mismatches(); }

// 1 in the -DSR_TESTING build (development hooks compiled in), 0 in the product library
int sr_testing_build(void)
{
#ifdef SR_TESTING
    return 1;
#else
    return 0;
#endif
}

int sr_dev_hook(const char *name, int64_t value)
{
    if (!name) return fail(SR_ERR_BAD_ARG, "null hook name");
#ifdef SR_TESTING
    for (int i = 0; i < kHookCount; i++)
        if (std::strcmp(name, kHookNames[i]) == 0) {
            g_hooks[i].store(value, std::memory_order_relaxed);
            return SR_OK;
        }
    return fail(SR_ERR_BAD_ARG, std::string("unknown development hook: ") + name);
#else
    (void)value;
    return fail(SR_ERR_BAD_ARG, std::string("development hook \"") + name + "\": hooks are not compiled into the product library "
                                "(the -DSR_TESTING build, libsr_engine_testing.so, has them)");
#endif
}

int sr_dtw_geometry(uint32_t n_templates, uint32_t max_frames, uint32_t out[5])
{
    if (!out || !n_templates || max_frames < 2 || max_frames > 16383) return fail(SR_ERR_BAD_ARG, "null / zero argument");
    size_t lds = 0;
    uint32_t tie_g = 0, kc = 0;
    const uint32_t U = dtw_lds_pick_u(n_templates, max_frames, &lds, &tie_g, &kc);
    out[0] = U;
    out[1] = U ? kc : 0;
    out[2] = U ? tie_g : 0;
    out[3] = U ? (uint32_t)lds : 0;
    out[4] = U ? (uint32_t)((160u * 1024u) / ((lds + 1279) / 1280 * 1280)) : 0;
    return SR_OK;
}

static void warn_log_table()
{
    if (const int bad = log_table_mismatches()) {
        const std::string msg = "warning: this host's libm log() moves " + std::to_string(bad) +
                                " of the 2219 steps of (u32)(log(n)*100) (MFCC.C:168) relative to the shipped table; the shipped "
                                "positions are used, so results equal the golden fixtures', not this host's C path";
        (void)fail(SR_OK, msg);
        static bool once = false;
        if (!once) std::fprintf(stderr, "sr_engine: %s\n", msg.c_str());
        once = true;
    }
}

// Host-only: the tables sr_create would upload for cfg, copied out for inspection (tests diff them against
// MFCC_Arg.h:6-44 and cr4_fft_1024_stm32.s:285-629).  No device is touched.
int sr_build_tables(const sr_config *cfg, const sr_tables *out)
{
    if (!cfg || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    FrontEnd fe;
    int rc = front_end_of(cfg, &fe);
    if (rc) return rc;
    HostTables t;
    build_tables(t, fe);
    warn_log_table();
    if (out->hamm) std::memcpy(out->hamm, t.hamm.data(), t.hamm.size() * 2);
    if (out->tri_cen) std::memcpy(out->tri_cen, t.tri_cen.data(), t.tri_cen.size() * 2);
    if (out->tri_even) std::memcpy(out->tri_even, t.tri_even.data(), t.tri_even.size() * 2);
    if (out->tri_odd) std::memcpy(out->tri_odd, t.tri_odd.data(), t.tri_odd.size() * 2);
    if (out->dct) std::memcpy(out->dct, t.dct.data(), t.dct.size());
    if (out->tw_kr) std::memcpy(out->tw_kr, t.tw_kr.data(), t.tw_kr.size() * 2);
    if (out->tw_ki) std::memcpy(out->tw_ki, t.tw_ki.data(), t.tw_ki.size() * 2);
    if (out->log_thr) std::memcpy(out->log_thr, t.log_thr.data(), t.log_thr.size() * 4);
    return SR_OK;
}

int sr_build_tie_table(int8_t *out)
{
    if (!out) return fail(SR_ERR_BAD_ARG, "null argument");
    HostTables t;
    build_tables(t, kFrontRef);  // the tie thresholds do not depend on the front end
    if (t.tie_delta.size() != (size_t)kTieMax) return fail(SR_ERR_BAD_CONFIG, "internal: DTW tie-threshold table does not fit 8 bits");
    std::memcpy(out, t.tie_delta.data(), t.tie_delta.size());
    return SR_OK;
}

int sr_create(const sr_config *cfg, sr_engine **out)
{
    if (!cfg || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    FrontEnd fe;
    if (int rcf = front_end_of(cfg, &fe)) return rcf;
    if (cfg->max_frames < 2 || cfg->max_frames > 16383) return fail(SR_ERR_BAD_CONFIG, "max_frames must be 2..16383");
    if (cfg->max_seg < 1 || cfg->max_seg > SR_MAX_SEG) return fail(SR_ERR_BAD_CONFIG, "max_seg must be 1..3");
    const uint32_t noise_len = (cfg->fs / 1000) * cfg->noise_len_ms, atap_frm = (cfg->fs / 1000) * 30;
    if (noise_len == 0 || noise_len % atap_frm != 0 || noise_len % (uint32_t)fe.frame_len != 0)
        return fail(SR_ERR_BAD_CONFIG, "noise_len_ms must be a non-zero multiple of 60 ms");
    int dev = 0;
    int rc = check_device(cfg->device, &dev);
    if (rc) return rc;
    DeviceGuard dev_guard_;
    if ((rc = dev_guard_.enter(dev))) return rc;

    sr_engine *h = new sr_engine();
    h->cfg = *cfg;
    h->device = dev;
    h->noise_len = noise_len;
    h->atap_frm = atap_frm;
    h->frame_len = (uint32_t)fe.frame_len;
    h->hop = (uint32_t)fe.hop;
    h->nc = (uint32_t)fe.n_coef;
    h->n_mel = (uint32_t)fe.n_mel;
    h->generic = fe.generic;
    h->v_durmin = 80 / (cfg->frame_time_ms - cfg->frame_mov_ms);   // VAD.C:72-75
    h->s_durmax = 110 / (cfg->frame_time_ms - cfg->frame_mov_ms);
    if (h->v_durmin < 1 || h->s_durmax < 1) {
        delete h;
        return fail(SR_ERR_BAD_CONFIG, "frame_time_ms - frame_mov_ms must not exceed 80 ms (VAD.C:72-75)");
    }
    h->mfcc_tile = h->generic ? 1u : mfcc_frames_per_tile(h->frame_len);
    h->mfcc_tile_mid = h->generic ? 1u : mfcc_frames_per_tile_small(h->frame_len, 0);
    h->mfcc_tile_small = h->generic ? 1u : mfcc_frames_per_tile_small(h->frame_len, 1);
    // Grid of the frame kernel: FOUR times the workgroups that are resident at once, work items strided.  Exactly the
    // resident set (one persistent wave of workgroups) left ~15 % of the kernel's own time to stragglers: the workgroups
    // do not finish together, and with more, shorter ones the dispatcher back-fills the CUs that are done (measured alone
    // on the chip, 65 536 x 256 frames: 1024 workgroups 19.0 ms, 2048 17.8, 4096 17.1, 16 384 16.6; the per-workgroup
    // set-up -- coefficient and DCT tables -- is amortised over 80 items at 4096).
    h->mfcc_grid_cap = h->generic ? 0u : 4 * mfcc_resident_workgroups(h->frame_len);
    if (const int64_t gv = dev_hook(kHookMfccGrid)) {  // development hook: workgroups of the frame kernel
        if (gv > 0) h->mfcc_grid_cap = (uint32_t)gv;
    }
    build_tables(h->host, fe);
    warn_log_table();
    // one blob, 16-byte aligned sub-tables
    const HostTables &t = h->host;
    std::vector<uint32_t> te32(t.tri_even.begin(), t.tri_even.end()), to32(t.tri_odd.begin(), t.tri_odd.end());
    std::vector<uint32_t> tem(t.tri_even.size()), tom(t.tri_odd.size());
    for (size_t i = 0; i < tem.size(); i++) {
        if (t.tri_even[i] > kMelTriMax || t.tri_odd[i] > kMelTriMax) {
            delete h;
            return fail(SR_ERR_BAD_CONFIG, "internal: Mel triangle weight above the fused multiplier's range");
        }
        tem[i] = mel_fused_multiplier(t.tri_even[i]);
        tom[i] = mel_fused_multiplier(t.tri_odd[i]);
    }
    std::vector<uint32_t> hpk(t.hamm.size() / 2);
    for (size_t i = 0; i < hpk.size(); i++) hpk[i] = (uint32_t)t.hamm[2 * i] | ((uint32_t)t.hamm[2 * i + 1] << 16);
    struct Part {
        const void *src;
        size_t bytes;
        size_t off;
    } parts[16] = {{t.hamm.data(), t.hamm.size() * 2, 0},       {t.tri_even.data(), t.tri_even.size() * 2, 0},
                  {t.tri_odd.data(), t.tri_odd.size() * 2, 0}, {t.tri_cen.data(), t.tri_cen.size() * 2, 0},
                  {t.dct.data(), t.dct.size(), 0},             {t.tw_a.data(), t.tw_a.size() * 4, 0},
                  {t.tw_b.data(), t.tw_b.size() * 4, 0},       {t.log_thr.data(), t.log_thr.size() * 4, 0},
                  {t.w512_a.data(), t.w512_a.size() * 4, 0},   {t.w512_b.data(), t.w512_b.size() * 4, 0},
                  {te32.data(), te32.size() * 4, 0},           {to32.data(), to32.size() * 4, 0},
                  {t.tie_delta.data(), t.tie_delta.size(), 0}, {hpk.data(), hpk.size() * 4, 0},
                  {tem.data(), tem.size() * 4, 0},             {tom.data(), tom.size() * 4, 0}};
    if (t.tie_delta.size() != (size_t)kTieMax) {
        delete h;
        return fail(SR_ERR_BAD_CONFIG, "internal: DTW tie-threshold table does not fit 8 bits");
    }
    size_t total = 0;
    for (auto &p : parts) {
        p.off = total;
        total += (p.bytes + 255) & ~(size_t)255;
    }
    std::vector<uint8_t> blob(total, 0);
    for (auto &p : parts) std::memcpy(blob.data() + p.off, p.src, p.bytes);
    hipError_t e = hipMalloc(&h->table_blob, total);
    if (e == hipSuccess) e = hipMemcpy(h->table_blob, blob.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        delete h;
        return fail(SR_ERR_HIP, std::string("table upload: ") + hipGetErrorString(e));
    }
    uint8_t *base = (uint8_t *)h->table_blob;
    h->dev.hamm = (const uint16_t *)(base + parts[0].off);
    h->dev.tri_even = (const uint16_t *)(base + parts[1].off);
    h->dev.tri_odd = (const uint16_t *)(base + parts[2].off);
    h->dev.tri_cen = (const uint16_t *)(base + parts[3].off);
    h->dev.dct = (const int8_t *)(base + parts[4].off);
    h->dev.tw_a = (const uint32_t *)(base + parts[5].off);
    h->dev.tw_b = (const uint32_t *)(base + parts[6].off);
    h->dev.log_thr = (const uint32_t *)(base + parts[7].off);
    h->dev.w512_a = (const uint32_t *)(base + parts[8].off);
    h->dev.w512_b = (const uint32_t *)(base + parts[9].off);
    h->dev.tri_even32 = (const uint32_t *)(base + parts[10].off);
    h->dev.tri_odd32 = (const uint32_t *)(base + parts[11].off);
    h->dev.tie_delta = (const int8_t *)(base + parts[12].off);
    h->dev.hamm_pk = (const uint32_t *)(base + parts[13].off);
    h->dev.tri_even_m = (const uint32_t *)(base + parts[14].off);
    h->dev.tri_odd_m = (const uint32_t *)(base + parts[15].off);
    if (h->s_pcnt.reserve(kPairCounters) != SR_OK || hipMemset(h->s_pcnt.p, 0, kPairCounters * sizeof(uint32_t)) != hipSuccess) {
        sr_destroy(h);
        return fail(SR_ERR_HIP, "pair counters");
    }
    *out = h;
    return SR_OK;
}

void sr_destroy(sr_engine *h)
{
    if (!h) return;
    DeviceGuard dev_guard_;
    (void)dev_guard_.enter(h->device);
    (void)hipDeviceSynchronize();
    if (h->table_blob) (void)hipFree(h->table_blob);
    h->tpl.release();
    h->tpl_frames.release();
    h->tpl_valid.release();
    h->tplR.release();
    h->tpl_frames_s.release();
    h->tpl_orig.release();
    h->s_pcm.release();
    h->s_pack.release();
    h->s_vad.release();
    h->s_mfcc.release();
    h->s_scores.release();
    h->s_results.release();
    h->s_u32a.release();
    h->s_u32b.release();
    h->s_atap.release();
    h->s_vad2.release();
    h->s_pcnt.release();
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    for (auto &e : h->ev_call) (void)hipEventDestroy(e);
    for (auto &e : h->ev_chunk) (void)hipEventDestroy(e);
    for (uint32_t i = 0; i < sr_engine::kPipeStreams; i++) {
        if (h->st_pipe[i]) (void)hipStreamDestroy(h->st_pipe[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->pin_buf) (void)hipHostFree(h->pin_buf);
    if (h->st_copy) (void)hipStreamDestroy(h->st_copy);
    if (h->st_comp) (void)hipStreamDestroy(h->st_comp);
    delete h;
}

uint32_t sr_num_templates(const sr_engine *h) { return h ? h->K : 0; }

// ---- template store -------------------------------------------------------------------------------
// Builds the new store in fresh device buffers and publishes it (pointers, K, rows, kernel geometry) only after every
// upload has succeeded, so a failure leaves the previous store intact.  The call first waits for all work on the device:
// the internal pipeline streams and user streams are non-blocking, and a kernel of an earlier asynchronous
// sr_recognize_batch_dev may still be reading the old rows.
static int upload_templates(sr_engine *h, const std::vector<int16_t> &m, const std::vector<uint32_t> &f,
                            const std::vector<uint8_t> &v, uint32_t K, uint32_t rows)
{
    ENTER_DEVICE(h);
    const uint32_t nc = h->nc;  // s16 per feature row; the staged kernels (k_dtw_lds, k_dtw_dp_band) are built for 12
    HIP_TRY(hipDeviceSynchronize());
    DevBuf<int16_t> n_tpl;
    DevBuf<uint32_t> n_frames, n_tplR, n_frames_s, n_orig;
    DevBuf<uint8_t> n_valid;
    bool fits = true;
    auto build = [&]() -> int {
        int rc;
        if ((rc = n_tpl.reserve(m.size()))) return rc;
        if ((rc = n_frames.reserve(K))) return rc;
        if ((rc = n_valid.reserve(K))) return rc;
        HIP_TRY(hipMemcpy(n_tpl.p, m.data(), m.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(n_frames.p, f.data(), K * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(n_valid.p, v.data(), K, hipMemcpyHostToDevice));
        // length-sorted, row-interleaved copy + squared norms for the LDS-staged DTW kernel
        std::vector<uint32_t> order(K);
        for (uint32_t k = 0; k < K; k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const uint32_t fx = v[x] ? f[x] : 0xFFFFFFFFu, fy = v[y] ? f[y] : 0xFFFFFFFFu;
            return fx < fy;
        });
        // Rows hold -2*coef so that get_dis's sum of squares (DTW.C:51-57) becomes |m|^2 + |in|^2 + (-2m).in with the
        // norm sum seeding the v_dot2 accumulator.  -2*coef must fit s16: coefficients outside [-16383, 16384]
        // (unreachable for log-Mel cepstra, reachable for arbitrary s16 records) disable the staged kernel for
        // this store and the generic k_dtw, which makes no such assumption, scores it.
        // Row format: up to 12 coefficients -> 32 bytes (12 x s16, zero-padded | norm | pad); 13..16 -> 48 bytes (16 x s16 | norm | pad)
        const uint32_t cw = nc > (uint32_t)kCoef ? 16u : (uint32_t)kCoef, rw = nc > (uint32_t)kCoef ? 12u : 8u;  // coefficients / words per row
        std::vector<uint32_t> rt((size_t)rows * K * rw, 0u), fs(K);
        for (uint32_t ks = 0; ks < K; ks++) {
            const uint32_t k = order[ks];
            fs[ks] = v[k] ? f[k] : 0u;
            for (uint32_t r = 0; r < rows; r++) {
                // narrower rows (GENERIC front end) are zero-padded: nothing is added to get_dis' sum
                int16_t src[16] = {0}, neg2[16];
                std::memcpy(src, &m[((size_t)k * rows + r) * nc], (size_t)nc * 2);
                uint32_t *dst = &rt[((size_t)r * K + ks) * rw];
                uint32_t nrm = 0;
                for (uint32_t c = 0; c < cw; c++) {
                    nrm += (uint32_t)((int32_t)src[c] * (int32_t)src[c]);
                    if (src[c] < -16383 || src[c] > 16384) fits = false;
                    neg2[c] = (int16_t)(-2 * (int32_t)src[c]);
                }
                std::memcpy(dst, neg2, (size_t)cw * 2);
                dst[cw / 2] = nrm;
            }
        }
        if ((uint64_t)rows * K * rw * 4 >= (1ull << 32)) fits = false;  // k_dtw_lds addresses the rows through a 32-bit byte offset
        if ((rc = n_tplR.reserve(rt.size()))) return rc;
        if ((rc = n_frames_s.reserve(K))) return rc;
        if ((rc = n_orig.reserve(K))) return rc;
        HIP_TRY(hipMemcpy(n_tplR.p, rt.data(), rt.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(n_frames_s.p, fs.data(), K * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(n_orig.p, order.data(), K * 4, hipMemcpyHostToDevice));
        return SR_OK;
    };
    const int rc = build();
    if (rc) {  // previous store untouched
        n_tpl.release();
        n_frames.release();
        n_valid.release();
        n_tplR.release();
        n_frames_s.release();
        n_orig.release();
        return rc;
    }
    std::swap(h->tpl, n_tpl);
    std::swap(h->tpl_frames, n_frames);
    std::swap(h->tpl_valid, n_valid);
    std::swap(h->tplR, n_tplR);
    std::swap(h->tpl_frames_s, n_frames_s);
    std::swap(h->tpl_orig, n_orig);
    n_tpl.release();  // the old store (nothing in flight: synchronised above)
    n_frames.release();
    n_valid.release();
    n_tplR.release();
    n_frames_s.release();
    n_orig.release();
    h->tpl_staged_ok = fits;
    {
        size_t lds = 0;
        uint32_t tie_g = 0, kc = 0;
        h->dtw_u = h->tpl_staged_ok ? dtw_lds_pick_u(K, h->cfg.max_frames, &lds, &tie_g, &kc, nc > (uint32_t)kCoef ? 8u : 6u) : 0;
        h->dtw_lds = (uint32_t)lds;
        h->dtw_tie_g = tie_g;
        h->dtw_kc = kc;
        if (dev_hook(kHookDtwDebug))
                std::fprintf(stderr, "sr_engine: k_dtw_lds geometry for K = %u, %u rows: U = %u, Kc = %u, tie table %u, LDS %zu bytes\n", K,
                             h->cfg.max_frames, h->dtw_u, kc, tie_g, lds);
    }
    // Chunk count of the device-resident pipeline by store size (round-4 sweeps, profiles/experiments/RESULTS.md): with
    // 100 templates 3 streams x 6..15 chunks are equivalent (22.3 ms per 65 536 utterances); with 500 templates the DTW
    // launches dominate and fewer, longer chunks win by 1 % (3 x 6: 45.0-46.2 ms, 3 x 12: 45.2-46.7).
    // small launches: the most band points any pair of this store can have (k_dtw_cells keeps one word per point in LDS)
    // -- capped at what a workgroup's LDS holds beside the rows: a pair with more points than that (utterances near the frame
    // cap against the longest templates) is walked literally by its workgroup, which costs what the batch kernel costs
    h->cells_points = dtw_cells_max_points(h->cfg.max_frames, f.data(), v.data(), K, h->cells_by_len);
    {
        const size_t fixed = dtw_cells_lds(h->cfg.max_frames, rows, 0), budget = 150 * 1024;
        const size_t room = fixed < budget ? (budget - fixed) / sizeof(uint32_t) : 0;
        if (h->cells_points > room) h->cells_points = room >= 4096 ? (uint32_t)room : 0u;
    }
    if (!h->pipe_user_set) h->pipe_max_chunks = K >= 256 ? 6 : 12;
    h->K = K;
    h->tpl_rows = rows;
    h->tpl_stride = rows * nc;
    return SR_OK;
}

int sr_set_templates_dense(sr_engine *h, const int16_t *mfcc, const uint32_t *frames, const uint8_t *valid,
                           uint32_t K, uint32_t tpl_stride)
{
    if (!h || !mfcc || !frames || K == 0) return fail(SR_ERR_BAD_ARG, "null/empty template set");
    const uint32_t nc = h->nc;
    if (tpl_stride % nc) return fail(SR_ERR_BAD_ARG, "tpl_stride must be a multiple of n_coef");
    const uint32_t src_rows = tpl_stride / nc;
    uint32_t maxf = 1;
    for (uint32_t k = 0; k < K; k++) {
        if (frames[k] > src_rows) return fail(SR_ERR_BAD_ARG, "template frame count exceeds its stride");
        if (frames[k] > 16383) return fail(SR_ERR_BAD_ARG, "template longer than 16383 frames");
        maxf = frames[k] > maxf ? frames[k] : maxf;
    }
    // one row of slack: the do-while of DTW.C:150-154 reads row 1 of a 1-frame template
    const uint32_t rows = maxf + 1;
    std::vector<int16_t> m((size_t)K * rows * nc, 0);
    std::vector<uint32_t> f(frames, frames + K);
    std::vector<uint8_t> v(K, 1);
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t copy_rows = src_rows < rows ? src_rows : rows;
        std::memcpy(&m[(size_t)k * rows * nc], mfcc + (size_t)k * tpl_stride, (size_t)copy_rows * nc * 2);
        if (valid) v[k] = valid[k] ? 1 : 0;
    }
    return upload_templates(h, m, f, v, K, rows);
}

int sr_set_templates(sr_engine *h, const void *store, uint32_t n_slots, uint32_t stride_bytes)
{
    if (!h || !store || n_slots == 0) return fail(SR_ERR_BAD_ARG, "null/empty template store");
    const uint32_t nc = h->nc;
    if (stride_bytes < 4 + 2 * nc) return fail(SR_ERR_BAD_ARG, "slot stride too small for a v_ftr_tag");
    // v_ftr_tag image: u16 save_sign | u16 frm_num | s16 mfcc_dat[] (MFCC.H:18-25), slot stride Flash.H:13
    const uint8_t *s = (const uint8_t *)store;
    const uint32_t slot_rows = (stride_bytes - 4) / (2 * nc);
    std::vector<uint32_t> f(n_slots);
    std::vector<uint8_t> v(n_slots);
    uint32_t maxf = 1;
    for (uint32_t k = 0; k < n_slots; k++) {
        uint16_t sign, fr;
        std::memcpy(&sign, s + (size_t)k * stride_bytes, 2);
        std::memcpy(&fr, s + (size_t)k * stride_bytes + 2, 2);
        v[k] = sign == SR_SAVE_MASK;
        f[k] = fr;
        if (v[k]) {
            if (fr > slot_rows) return fail(SR_ERR_BAD_ARG, "slot frm_num exceeds the slot size");
            if (fr > 16383) return fail(SR_ERR_BAD_ARG, "template longer than 16383 frames");
            maxf = fr > maxf ? fr : maxf;
        }
    }
    uint32_t rows = maxf + 1;
    std::vector<int16_t> m((size_t)n_slots * rows * nc, 0);
    for (uint32_t k = 0; k < n_slots; k++) {
        if (!v[k]) continue;
        const uint32_t copy_rows = slot_rows < rows ? slot_rows : rows;  // keeps whatever follows frm_num rows
        std::memcpy(&m[(size_t)k * rows * nc], s + (size_t)k * stride_bytes + 4, (size_t)copy_rows * nc * 2);
    }
    return upload_templates(h, m, f, v, n_slots, rows);
}

// ---- profiling ------------------------------------------------------------------------------------
int sr_set_profiling(sr_engine *h, int on)
{
    if (!h) return fail(SR_ERR_BAD_ARG, "null engine");
    ENTER_DEVICE(h);
    h->profiling = on != 0;
    h->ev_used = 0;
    h->calls_used = 0;
    return SR_OK;
}

int sr_get_stage_ms(sr_engine *h, float ms[5])
{
    if (!h || !ms) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->ev_used || !h->calls_used) return fail(SR_ERR_BAD_ARG, "no profiled call recorded");
    // ms[0..3]: average duration of ONE launch of each kernel (a call launches each kernel once per chunk, see
    // sr_get_stage_launches); ms[4]: average duration of a whole call on the caller's stream (fork -> join)
    double acc[5] = {0, 0, 0, 0, 0};
    for (size_t c = 0; c < h->ev_used; c++) {
        hipEvent_t *e = &h->ev[5 * c];
        HIP_TRY(hipEventSynchronize(e[4]));
        float t;
        for (int i = 0; i < 4; i++) {
            HIP_TRY(hipEventElapsedTime(&t, e[i], e[i + 1]));
            acc[i] += t;
        }
    }
    for (size_t c = 0; c < h->calls_used; c++) {
        float t;
        HIP_TRY(hipEventSynchronize(h->ev_call[2 * c + 1]));
        HIP_TRY(hipEventElapsedTime(&t, h->ev_call[2 * c], h->ev_call[2 * c + 1]));
        acc[4] += t;
    }
    for (int i = 0; i < 4; i++) ms[i] = (float)(acc[i] / (double)h->ev_used);
    ms[4] = (float)(acc[4] / (double)h->calls_used);
    return SR_OK;
}

int sr_set_pipeline(sr_engine *h, uint32_t streams, uint32_t min_chunk, uint32_t max_chunks)
{
    if (!h) return fail(SR_ERR_BAD_ARG, "null engine");
    if (streams < 1 || streams > sr_engine::kPipeStreams || min_chunk < 1 || max_chunks < 1 || max_chunks > 64)
        return fail(SR_ERR_BAD_ARG, "streams 1..4, min_chunk >= 1, max_chunks 1..64");
    h->pipe_streams = streams;
    h->pipe_min_chunk = min_chunk;
    h->pipe_max_chunks = max_chunks;
    h->pipe_user_set = true;
    return SR_OK;
}

int sr_set_dp_lanes(sr_engine *h, uint32_t lanes)
{
    if (!h) return fail(SR_ERR_BAD_ARG, "null engine");
    if (lanes != 0 && lanes != 1 && lanes != 4 && lanes != 8 && lanes != 16) return fail(SR_ERR_BAD_ARG, "lanes per pair: 0 (default), 1 (one wave per pair), 4, 8 or 16");
    h->dp_lanes = lanes;
    return SR_OK;
}

int sr_set_small_launch(sr_engine *h, int mode)
{
    if (!h) return fail(SR_ERR_BAD_ARG, "null engine");
    if (mode < 0 || mode > 2) return fail(SR_ERR_BAD_ARG, "small-launch mode: 0 (automatic), 1 (never), 2 (whenever the store fits)");
    h->small_launch = mode;
    return SR_OK;
}

int sr_get_stage_launches(sr_engine *h, uint32_t *launches_per_call)
{
    if (!h || !launches_per_call) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->calls_used) return fail(SR_ERR_BAD_ARG, "no profiled call recorded");
    *launches_per_call = (uint32_t)(h->ev_used / h->calls_used);
    return SR_OK;
}

// ---- device-resident pipeline ---------------------------------------------------------------------
// the frame kernel indexes (utterance, tile) work items with 32 bits
static int check_batch(const sr_engine *h, uint32_t B)
{
    if ((uint64_t)B * ((h->cfg.max_frames + h->mfcc_tile - 1) / h->mfcc_tile) > 0xFFFFFFFFull) return fail(SR_ERR_BAD_ARG, "batch too large");
    return SR_OK;
}
static int check_pcm(const sr_engine *h, const uint16_t *pcm, uint64_t stride, uint32_t buf_len)
{
    if (!pcm) return fail(SR_ERR_BAD_ARG, "null pcm");
    if (((uintptr_t)pcm & 15) || (stride & 7)) return fail(SR_ERR_BAD_ARG, "pcm must be 16-byte aligned, stride % 8 == 0");
    if (buf_len > stride) return fail(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    if (buf_len < h->noise_len || buf_len <= h->frame_len) return fail(SR_ERR_BAD_ARG, "buf_len shorter than the noise head");
    if (buf_len > 0x7FFFFFF0u) return fail(SR_ERR_BAD_ARG, "buf_len too large");
    // the extension frame kernel addresses a capture row through a raw buffer resource of 2 * pcm_stride bytes (32 bits)
    if (stride >= (1ull << 31)) return fail(SR_ERR_BAD_ARG, "pcm_stride must be below 2^31 samples");
    return SR_OK;
}

static VadArgs vad_args(const sr_engine *h, const uint16_t *pcm, uint64_t stride, uint32_t buf_len, uint32_t noise_len, uint32_t B,
                        sr_vad_rec *vad, const sr_atap *atap_in = nullptr, uint64_t *dbg = nullptr)
{
    // fewer captures than CUs: a workgroup of four waves per capture instead of one wave (k_vad_wide; same records)
    const uint32_t wide = (h->small_launch == 2 || (h->small_launch == 0 && B < kVadWideBelow)) ? 1u : 0u;
    return VadArgs{pcm, stride, buf_len, noise_len, h->atap_frm, h->cfg.max_frames, h->cfg.max_seg, B, vad, atap_in, dbg,
                   h->frame_len, h->v_durmin, h->s_durmax, wide};
}

int sr_vad_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                     sr_vad_rec *d_vad, void *stream)
{
    if (!h || !d_vad) return fail(SR_ERR_BAD_ARG, "null argument");
    int rc = check_pcm(h, d_pcm, pcm_stride, buf_len);
    if (rc) return rc;
    ENTER_DEVICE(h);
    VadArgs a = vad_args(h, d_pcm, pcm_stride, buf_len, h->noise_len, B, d_vad);
    launch_vad(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

static MfccArgs mfcc_args(const sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t B,
                          const sr_vad_rec *d_vad, int16_t *d_mfcc)
{
    MfccArgs a;
    a.pcm = d_pcm;
    a.pcm_stride = pcm_stride;
    a.B = B;
    a.max_frames = h->cfg.max_frames;
    a.vad = d_vad;
    a.mfcc = d_mfcc;
    a.tiles = (h->cfg.max_frames + h->mfcc_tile - 1) / h->mfcc_tile;
    a.small_tiles = 0;
    // Too few 64-frame work items to fill the chip (a wave's frames are a serial chain, and nothing else would run): the frame
    // kernel's forms with 16 or 4 frames per workgroup -- the largest whose work items reach kMfccFill, else the smallest.
    // Same arithmetic.  (Mode 2 = always the smallest.)
    if (h->mfcc_tile_small < h->mfcc_tile && h->small_launch != 1) {
        const uint32_t t_mid = (h->cfg.max_frames + h->mfcc_tile_mid - 1) / h->mfcc_tile_mid;
        const uint32_t t_small = (h->cfg.max_frames + h->mfcc_tile_small - 1) / h->mfcc_tile_small;
        if (h->small_launch == 2 || (uint64_t)B * t_mid < kMfccFill) {
            a.tiles = t_small;
            a.small_tiles = 2;
        } else if ((uint64_t)B * a.tiles < kMfccFill) {
            a.tiles = t_mid;
            a.small_tiles = 1;
        }
    }
    a.grid_cap = h->mfcc_grid_cap;
    a.frame_len = h->frame_len;
    a.n_items = B * a.tiles;
    a.generic = h->generic ? 1u : 0u;
    a.hop = h->hop;
    a.n_mel = h->n_mel;
    a.n_coef = h->nc;
    a.t = h->dev;
    return a;
}

int sr_mfcc_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t B, const sr_vad_rec *d_vad,
                      int16_t *d_mfcc, void *stream)
{
    if (!h || !d_pcm || !d_vad || !d_mfcc) return fail(SR_ERR_BAD_ARG, "null argument");
    if (pcm_stride >= (1ull << 31)) return fail(SR_ERR_BAD_ARG, "pcm_stride must be below 2^31 samples");
    if (int rcb = check_batch(h, B)) return rcb;
    ENTER_DEVICE(h);
    launch_mfcc(mfcc_args(h, d_pcm, pcm_stride, B, d_vad, d_mfcc), (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

static DtwArgs dtw_args(const sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad, const uint32_t *d_in_frames,
                        uint32_t B, uint32_t *d_scores, sr_result *d_results)
{
    DtwArgs a;
    a.mfcc = d_mfcc;
    a.vad = d_vad;
    a.in_frames = d_in_frames;
    a.B = B;
    a.max_frames = h->cfg.max_frames;
    a.tpl = h->tpl.p;
    a.tpl_frames = h->tpl_frames.p;
    a.tpl_valid = h->tpl_valid.p;
    a.K = h->K;
    a.tpl_stride = h->tpl_stride;
    a.tpl_rows = h->tpl_rows;
    a.scores = d_scores;
    a.results = d_results;
    a.tplR = h->tplR.p;
    a.tpl_frames_s = h->tpl_frames_s.p;
    a.tpl_orig = h->tpl_orig.p;
    a.lds_u = h->dtw_u;
    a.lds_bytes = h->dtw_lds;
    a.tie_delta = h->dev.tie_delta;
    a.tie_g = h->dtw_tie_g;
    a.lds_kc = h->dtw_kc;
    a.n_coef = h->nc;
    a.dp_lanes = h->dp_lanes;
    a.pair_count = nullptr;
    a.cells_points = h->cells_points;
    a.cells_literal = dev_hook(kHookCellsLiteral) != 0 ? 1u : 0u;
    return a;
}

// dtw for every pair of the launch: the batch kernels (k_dtw_lds / k_dtw_gen / k_dtw), or -- a few hundred pairs, i.e. a GPU
// that would otherwise idle behind a handful of serial walks -- one workgroup per pair (k_dtw_cells).  Same scores.
// Measured (profiles/r04_small_launch_sweep.json, profiles/experiments/RESULTS.md): 110-frame captures against 80 slots of up
// to 119 frames: 80 / 320 / 640 / 1 280 / 2 560 / 5 120 pairs take 25 / 33 / 44 / 65 / 115 / 212 us with one workgroup per pair
// against 126 us for the batch kernel at any of these sizes; 256-frame captures against 100 templates of 192-320 frames (the
// benchmark's shapes): 100 / 400 pairs 60 / 107 us against 215.  A pair costs in proportion to its band (~ frames^2), the
// batch kernel's latency grows with the frames, so the automatic mode stops at 320 000 / max_frames pairs (2 689 / 1 000).
static uint64_t small_launch_pairs(const DtwArgs &a) { return 320000u / (a.max_frames > 64 ? a.max_frames : 64u); }
// returns true when the slot scan (argmin) has been done as well: k_dtw_cells with result records asked for and the utterances
// b0 .. b0 + B of the call within the counters
static bool launch_dtw_auto(const sr_engine *h, DtwArgs &a, uint32_t b0, hipStream_t s)
{
    if (h->small_launch != 1 && dtw_cells_fits(a) && (h->small_launch == 2 || (uint64_t)a.B * a.K <= small_launch_pairs(a))) {
        a.pair_count = (a.results && (uint64_t)b0 + a.B <= kPairCounters) ? h->s_pcnt.p + b0 : nullptr;
        launch_dtw_cells(a, s);
        return a.pair_count != nullptr;
    }
    launch_dtw(a, s);
    return false;
}

int sr_dtw_batch_dev(sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad, uint32_t B, uint32_t *d_scores,
                     sr_result *d_results, void *stream)
{
    if (!h || !d_mfcc || !d_vad || !d_scores) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    ENTER_DEVICE(h);
    DtwArgs a = dtw_args(h, d_mfcc, d_vad, nullptr, B, d_scores, d_results);
    if (!launch_dtw_auto(h, a, 0, (hipStream_t)stream) && d_results) launch_argmin(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

int sr_recognize_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                           sr_result *d_results, uint32_t *d_scores, int16_t *d_mfcc, sr_vad_rec *d_vad, void *stream)
{
    if (!h || !d_results) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    int rc = check_pcm(h, d_pcm, pcm_stride, buf_len);
    if (rc) return rc;
    if ((rc = check_batch(h, B))) return rc;
    ENTER_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    if (!d_vad) {
        if ((rc = h->s_vad.reserve(B))) return rc;
        d_vad = h->s_vad.p;
    }
    if (!d_mfcc) {
        if ((rc = h->s_mfcc.reserve((size_t)B * h->cfg.max_frames * h->nc))) return rc;
        d_mfcc = h->s_mfcc.p;
    }
    if (!d_scores) {
        if ((rc = h->s_scores.reserve((size_t)B * h->K))) return rc;
        d_scores = h->s_scores.p;
    }
    // ---- chunks over the internal streams ------------------------------------------------------------
    uint32_t n_chunks = std::min<uint32_t>(h->pipe_max_chunks, B / std::max<uint32_t>(1, h->pipe_min_chunk));
    if (n_chunks < 2 || h->pipe_streams < 2) n_chunks = 1;
    const uint32_t n_streams = (n_chunks == 1) ? 1 : std::min(h->pipe_streams, n_chunks);
    const bool prof = h->profiling;
    if (prof) {
        while (h->ev.size() < 5 * (h->ev_used + n_chunks)) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            h->ev.push_back(e);
        }
        while (h->ev_call.size() < 2 * (h->calls_used + 1)) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            h->ev_call.push_back(e);
        }
        HIP_TRY(hipEventRecord(h->ev_call[2 * h->calls_used], s));
    }
    if (n_chunks > 1) {
        if (!h->ev_fork) HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (uint32_t i = 0; i < n_streams; i++) {
            if (!h->st_pipe[i]) HIP_TRY(hipStreamCreateWithFlags(&h->st_pipe[i], hipStreamNonBlocking));
            if (!h->ev_join[i]) HIP_TRY(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(h->ev_fork, s));  // everything the caller queued before this call
        for (uint32_t i = 0; i < n_streams; i++) HIP_TRY(hipStreamWaitEvent(h->st_pipe[i], h->ev_fork, 0));
    }
    const uint32_t per = (B + n_chunks - 1) / n_chunks;
    uint32_t c = 0;
    for (uint32_t b0 = 0; b0 < B; b0 += per, c++) {
        const uint32_t n = std::min(per, B - b0);
        hipStream_t sc = (n_chunks == 1) ? s : h->st_pipe[c % n_streams];
        hipEvent_t *ev = prof ? &h->ev[5 * (h->ev_used + c)] : nullptr;
        const uint16_t *pc = d_pcm + (size_t)b0 * pcm_stride;
        sr_vad_rec *vc = d_vad + b0;
        int16_t *mc = d_mfcc + (size_t)b0 * h->cfg.max_frames * h->nc;
        VadArgs va = vad_args(h, pc, pcm_stride, buf_len, h->noise_len, n, vc);
        if (prof) HIP_TRY(hipEventRecord(ev[0], sc));
        launch_vad(va, sc);
        if (prof) HIP_TRY(hipEventRecord(ev[1], sc));
        launch_mfcc(mfcc_args(h, pc, pcm_stride, n, vc, mc), sc);
        if (prof) HIP_TRY(hipEventRecord(ev[2], sc));
        DtwArgs da = dtw_args(h, mc, vc, nullptr, n, d_scores + (size_t)b0 * h->K, d_results + b0);
        const bool scanned = launch_dtw_auto(h, da, b0, sc);
        if (prof) HIP_TRY(hipEventRecord(ev[3], sc));
        if (!scanned) launch_argmin(da, sc);
        if (prof) HIP_TRY(hipEventRecord(ev[4], sc));
    }
    if (n_chunks > 1) {
        for (uint32_t i = 0; i < n_streams; i++) {
            HIP_TRY(hipEventRecord(h->ev_join[i], h->st_pipe[i]));
            HIP_TRY(hipStreamWaitEvent(s, h->ev_join[i], 0));  // the caller's stream continues after every chunk
        }
    }
    if (prof) {
        HIP_TRY(hipEventRecord(h->ev_call[2 * h->calls_used + 1], s));
        h->ev_used += c;
        h->calls_used++;
    }
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

// Every segment the VAD finds (up to max_seg), each matched like segment 0.  The firmware's spch_recg stops at
// segment 0 (main.c:268); this is the "multi-segment" extension of SURVEY.md 8(f).  Segment-major outputs:
// d_results[s*B + b], d_scores[(s*B + b)*K + k].
int sr_recognize_segments_batch_dev(sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t buf_len,
                                    uint32_t B, sr_result *d_results, uint32_t *d_scores, sr_vad_rec *d_vad, void *stream)
{
    if (!h || !d_results) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    int rc = check_pcm(h, d_pcm, pcm_stride, buf_len);
    if (rc) return rc;
    if ((rc = check_batch(h, B))) return rc;
    ENTER_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    if (!d_vad) {
        if ((rc = h->s_vad.reserve(B))) return rc;
        d_vad = h->s_vad.p;
    }
    if ((rc = h->s_vad2.reserve(B))) return rc;
    if ((rc = h->s_mfcc.reserve((size_t)B * h->cfg.max_frames * h->nc))) return rc;
    if (!d_scores) {
        if ((rc = h->s_scores.reserve((size_t)B * h->K * h->cfg.max_seg))) return rc;
        d_scores = h->s_scores.p;
    }
    VadArgs va = vad_args(h, d_pcm, pcm_stride, buf_len, h->noise_len, B, d_vad);
    launch_vad(va, s);
    for (uint32_t sg = 0; sg < h->cfg.max_seg; sg++) {
        launch_select_segment(d_vad, h->s_vad2.p, B, sg, h->cfg.max_frames, h->frame_len, h->hop, s);
        launch_mfcc(mfcc_args(h, d_pcm, pcm_stride, B, h->s_vad2.p, h->s_mfcc.p), s);
        DtwArgs da = dtw_args(h, h->s_mfcc.p, h->s_vad2.p, nullptr, B, d_scores + (size_t)sg * B * h->K,
                              d_results + (size_t)sg * B);
        if (!launch_dtw_auto(h, da, 0, s)) launch_argmin(da, s);
    }
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

int sr_recognize_segments_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                                sr_result *results, uint32_t *scores, sr_vad_rec *vad);

// ---- host-buffer wrappers (stage through HBM) --------------------------------------------------------
// Pinned host area of the small host-buffer calls (spch_recg / get_mfcc / VAD / dtw: one capture, one record): what goes up
// is staged in its first part, what comes back lands in its second part (result records are written there by the kernel
// itself).  Everything is enqueued on one internal stream and the host waits ONCE per call, instead of one blocking copy per
// buffer and direction.
static constexpr size_t kPinUpload = 256 * 1024, kPinMaxB = 256;           // captures of one small call; utterances
static constexpr size_t kPinUpBytes = kPinUpload + 64 * 1024;             // + records / frame counts / thresholds
static constexpr size_t kPinDownBytes = 704 * 1024, kPinTotal = kPinUpBytes + kPinDownBytes;
static constexpr size_t kPinMfccBytes = 512 * 1024;
// false = no pinned area on this host (allocation refused: the callers keep their blocking copies)
static bool ensure_pin(sr_engine *h)
{
    if (h->pin_cap < kPinTotal) {
        if (h->pin_failed) return false;
        if (hipHostMalloc(&h->pin_buf, kPinTotal, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            h->pin_buf = nullptr;
            h->pin_failed = true;
            return false;
        }
        h->pin_cap = kPinTotal;
    }
    if (!h->st_comp && hipStreamCreateWithFlags(&h->st_comp, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        h->st_comp = nullptr;
        return false;
    }
    return true;
}
// one small call: bump allocation in the two parts of the pinned area, asynchronous copies on the internal stream
struct PinCall {
    sr_engine *h;
    uint8_t *base;
    size_t up = 0, down = kPinUpBytes;
    bool ok = true;
    explicit PinCall(sr_engine *e) : h(e), base((uint8_t *)e->pin_buf) {}
    hipStream_t stream() const { return h->st_comp; }
    uint8_t *stage(size_t bytes)  // room in the upload part (callers check the sizes beforehand with pin_fits)
    {
        uint8_t *p = base + up;
        up += (bytes + 63) & ~(size_t)63;
        return p;
    }
    void upload(void *dev, const void *src, size_t bytes)  // host buffer -> staging -> device
    {
        uint8_t *p = stage(bytes);
        std::memcpy(p, src, bytes);
        if (hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, h->st_comp) != hipSuccess) ok = false;
    }
    uint8_t *landing(size_t bytes)  // room in the download part
    {
        uint8_t *p = base + down;
        down += (bytes + 63) & ~(size_t)63;
        return p;
    }
    uint8_t *download(const void *dev, size_t bytes)
    {
        uint8_t *p = landing(bytes);
        if (hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, h->st_comp) != hipSuccess) ok = false;
        return p;
    }
    int finish()  // the one synchronisation of the call
    {
        const hipError_t e = hipStreamSynchronize(h->st_comp);
        if (e != hipSuccess || !ok) {
            (void)hipGetLastError();
            return fail(SR_ERR_HIP, "small host call: copy / synchronisation failed");
        }
        return SR_OK;
    }
};
static bool pin_fits(sr_engine *h, size_t up_bytes, size_t down_bytes, uint32_t n_up = 1, uint32_t n_down = 1)
{
    return h->small_launch != 1 && up_bytes + 64 * (size_t)n_up <= kPinUpBytes && down_bytes + 64 * (size_t)n_down <= kPinDownBytes &&
           ensure_pin(h);
}
// rows of buf_len samples into the staging area at the device pitch ds (samples), the pad zeroed
static void stage_rows(uint8_t *stage, const uint8_t *src, uint64_t src_pitch, uint64_t row_bytes, uint64_t ds, uint32_t B)
{
    for (uint32_t b = 0; b < B; b++) {
        std::memcpy(stage + (size_t)b * ds * 2, src + (size_t)b * src_pitch, (size_t)row_bytes);
        if (ds * 2 > row_bytes) std::memset(stage + (size_t)b * ds * 2 + row_bytes, 0, (size_t)(ds * 2 - row_bytes));
    }
}
// captures of a small call: staged at the device pitch and sent on their way; false = not a small call (caller: stage_pcm)
static bool pin_stage_pcm(sr_engine *h, PinCall &pc, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                          uint64_t *dev_stride, int *rc)
{
    const uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    if ((*rc = h->s_pcm.reserve((size_t)B * ds))) return false;
    uint8_t *st = pc.stage((size_t)B * ds * 2);
    stage_rows(st, (const uint8_t *)pcm, pcm_stride * 2, (uint64_t)buf_len * 2, ds, B);
    if (hipMemcpyAsync(h->s_pcm.p, st, (size_t)B * ds * 2, hipMemcpyHostToDevice, h->st_comp) != hipSuccess) pc.ok = false;
    *dev_stride = ds;
    return true;
}

static int stage_pcm(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                     uint64_t *dev_stride)
{
    const uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    int rc = h->s_pcm.reserve((size_t)B * ds);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2D(h->s_pcm.p, ds * 2, pcm, pcm_stride * 2, (size_t)buf_len * 2, B, hipMemcpyHostToDevice));
    *dev_stride = ds;
    return SR_OK;
}

// Host buffers -> results.  `packed` = false: u16 rows of pcm_stride SAMPLES; true: rows of 12-bit codes, two samples in
// three bytes, row stride in BYTES (sr_recognize_batch_packed12).
static int recognize_host(sr_engine *h, const void *pcm, uint64_t row_stride, bool packed, uint32_t buf_len, uint32_t B,
                          sr_result *results, uint32_t *scores, int16_t *mfcc, sr_vad_rec *vad)
{
    if (!h || !pcm || !results) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    const uint64_t src_row_bytes = packed ? ((uint64_t)(buf_len + 1) / 2) * 3 : (uint64_t)buf_len * 2;  // bytes that carry samples
    const uint64_t src_pitch = packed ? row_stride : row_stride * 2;
    if (src_row_bytes > src_pitch) return fail(SR_ERR_BAD_ARG, packed ? "row stride smaller than ceil(buf_len / 2) * 3 bytes" : "buf_len exceeds pcm_stride");
    ENTER_DEVICE(h);
    const uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    const uint64_t dpk = ds / 8 * 12;  // device pitch of a packed row: whole groups of 8 samples = 12 bytes
    int rc;
    if ((rc = h->s_pcm.reserve((size_t)B * ds))) return rc;
    if (packed && (rc = h->s_pack.reserve((size_t)B * dpk + 16))) return rc;
    if ((rc = h->s_results.reserve(B))) return rc;
    if ((rc = h->s_vad.reserve(B))) return rc;
    if ((rc = h->s_mfcc.reserve((size_t)B * h->cfg.max_frames * h->nc))) return rc;
    if ((rc = h->s_scores.reserve((size_t)B * h->K))) return rc;
    // The upload dominates (2*buf_len bytes per utterance over PCIe vs ~0.5 us of kernels): split the batch into
    // chunks and let the upload of chunk c+1 run on the copy stream while chunk c is processed on the compute
    // stream.  hipMemcpy2DAsync from pageable memory returns when the host buffer has been consumed, so the host
    // thread paces the copies; kernels are only enqueued.  Results come back once, after the last chunk.
    const uint32_t n_chunks = (B >= 2048) ? std::min<uint32_t>(16, B / 1024) : 1;
    const uint8_t *src = (const uint8_t *)pcm;
    // A few captures (spch_recg's one): two blocking copies cost more than the kernels.  The rows go through a pinned staging
    // area, the result records are written by the kernel into pinned host memory, and the host waits once.
    if (!packed && !h->profiling && B <= kPinMaxB && pin_fits(h, (size_t)B * ds * 2, (size_t)B * sizeof(sr_result))) {
        PinCall pc(h);
        uint64_t ds2 = 0;
        if (!pin_stage_pcm(h, pc, (const uint16_t *)pcm, row_stride, buf_len, B, &ds2, &rc)) return rc;
        uint8_t *res_host = pc.landing((size_t)B * sizeof(sr_result));
        void *d_res = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d_res, res_host, 0));
        rc = sr_recognize_batch_dev(h, h->s_pcm.p, ds, buf_len, B, (sr_result *)d_res, h->s_scores.p, h->s_mfcc.p, h->s_vad.p, pc.stream());
        const int rcs = pc.finish();
        if (rc) return rc;
        if (rcs) return rcs;
        std::memcpy(results, res_host, (size_t)B * sizeof(sr_result));
        if (scores) HIP_TRY(hipMemcpy(scores, h->s_scores.p, (size_t)B * h->K * 4, hipMemcpyDeviceToHost));
        if (mfcc) HIP_TRY(hipMemcpy(mfcc, h->s_mfcc.p, (size_t)B * h->cfg.max_frames * h->nc * 2, hipMemcpyDeviceToHost));
        if (vad) HIP_TRY(hipMemcpy(vad, h->s_vad.p, (size_t)B * sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
        return SR_OK;
    }
    if (n_chunks <= 1 || h->profiling) {
        if (packed) {
            HIP_TRY(hipMemcpy2D(h->s_pack.p, dpk, src, src_pitch, src_row_bytes, B, hipMemcpyHostToDevice));
            launch_unpack12(h->s_pack.p, dpk, h->s_pcm.p, ds, buf_len, B, nullptr);
        } else {
            HIP_TRY(hipMemcpy2D(h->s_pcm.p, ds * 2, src, src_pitch, src_row_bytes, B, hipMemcpyHostToDevice));
        }
        rc = sr_recognize_batch_dev(h, h->s_pcm.p, ds, buf_len, B, h->s_results.p, h->s_scores.p, h->s_mfcc.p, h->s_vad.p,
                                    nullptr);
        if (rc) return rc;
    } else {
        if (!h->st_copy) HIP_TRY(hipStreamCreateWithFlags(&h->st_copy, hipStreamNonBlocking));
        if (!h->st_comp) HIP_TRY(hipStreamCreateWithFlags(&h->st_comp, hipStreamNonBlocking));
        while (h->ev_chunk.size() < n_chunks) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->ev_chunk.push_back(e);
        }
        HIP_TRY(hipDeviceSynchronize());  // earlier null-stream work on the scratch buffers is finished
        const uint32_t per = (B + n_chunks - 1) / n_chunks;
        for (uint32_t c = 0, b0 = 0; b0 < B; c++, b0 += per) {
            const uint32_t n = std::min(per, B - b0);
            if (packed)
                HIP_TRY(hipMemcpy2DAsync(h->s_pack.p + (size_t)b0 * dpk, dpk, src + (size_t)b0 * src_pitch, src_pitch, src_row_bytes, n,
                                         hipMemcpyHostToDevice, h->st_copy));
            else
                HIP_TRY(hipMemcpy2DAsync(h->s_pcm.p + (size_t)b0 * ds, ds * 2, src + (size_t)b0 * src_pitch, src_pitch, src_row_bytes, n,
                                         hipMemcpyHostToDevice, h->st_copy));
            HIP_TRY(hipEventRecord(h->ev_chunk[c], h->st_copy));
            HIP_TRY(hipStreamWaitEvent(h->st_comp, h->ev_chunk[c], 0));
            if (packed) launch_unpack12(h->s_pack.p + (size_t)b0 * dpk, dpk, h->s_pcm.p + (size_t)b0 * ds, ds, buf_len, n, h->st_comp);
            rc = sr_recognize_batch_dev(h, h->s_pcm.p + (size_t)b0 * ds, ds, buf_len, n, h->s_results.p + b0,
                                        h->s_scores.p + (size_t)b0 * h->K,
                                        h->s_mfcc.p + (size_t)b0 * h->cfg.max_frames * h->nc, h->s_vad.p + b0, h->st_comp);
            if (rc) {
                (void)hipDeviceSynchronize();
                return rc;
            }
        }
        HIP_TRY(hipStreamSynchronize(h->st_comp));
    }
    HIP_TRY(hipMemcpy(results, h->s_results.p, (size_t)B * sizeof(sr_result), hipMemcpyDeviceToHost));
    if (scores) HIP_TRY(hipMemcpy(scores, h->s_scores.p, (size_t)B * h->K * 4, hipMemcpyDeviceToHost));
    if (mfcc)
        HIP_TRY(hipMemcpy(mfcc, h->s_mfcc.p, (size_t)B * h->cfg.max_frames * h->nc * 2, hipMemcpyDeviceToHost));
    if (vad) HIP_TRY(hipMemcpy(vad, h->s_vad.p, (size_t)B * sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_recognize_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                       sr_result *results, uint32_t *scores, int16_t *mfcc, sr_vad_rec *vad)
{
    return recognize_host(h, pcm, pcm_stride, false, buf_len, B, results, scores, mfcc, vad);
}

int sr_recognize_batch_packed12(sr_engine *h, const uint8_t *packed, uint64_t row_stride_bytes, uint32_t buf_len, uint32_t B,
                                sr_result *results, uint32_t *scores, int16_t *mfcc, sr_vad_rec *vad)
{
    return recognize_host(h, packed, row_stride_bytes, true, buf_len, B, results, scores, mfcc, vad);
}

int sr_recognize_segments_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                                sr_result *results, uint32_t *scores, sr_vad_rec *vad)
{
    if (!h || !pcm || !results) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    if (buf_len > pcm_stride) return fail(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    ENTER_DEVICE(h);
    const uint32_t ms = h->cfg.max_seg;
    uint64_t ds = 0;
    int rc = stage_pcm(h, pcm, pcm_stride, buf_len, B, &ds);
    if (rc) return rc;
    if ((rc = h->s_results.reserve((size_t)B * ms))) return rc;
    if ((rc = h->s_vad.reserve(B))) return rc;
    if ((rc = h->s_scores.reserve((size_t)B * h->K * ms))) return rc;
    rc = sr_recognize_segments_batch_dev(h, h->s_pcm.p, ds, buf_len, B, h->s_results.p, h->s_scores.p, h->s_vad.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(results, h->s_results.p, (size_t)B * ms * sizeof(sr_result), hipMemcpyDeviceToHost));
    if (scores) HIP_TRY(hipMemcpy(scores, h->s_scores.p, (size_t)B * h->K * ms * 4, hipMemcpyDeviceToHost));
    if (vad) HIP_TRY(hipMemcpy(vad, h->s_vad.p, (size_t)B * sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_vad_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B, sr_vad_rec *vad)
{
    if (!h || !pcm || !vad) return fail(SR_ERR_BAD_ARG, "null argument");
    if (B == 0) return SR_OK;
    if (buf_len > pcm_stride) return fail(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    ENTER_DEVICE(h);
    uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    int rc;
    if ((rc = h->s_vad.reserve(B))) return rc;
    if (B <= kPinMaxB && pin_fits(h, (size_t)B * ds * 2, (size_t)B * sizeof(sr_vad_rec))) {  // a few captures: see PinCall
        PinCall pc(h);
        if (!pin_stage_pcm(h, pc, pcm, pcm_stride, buf_len, B, &ds, &rc)) return rc;
        rc = sr_vad_batch_dev(h, h->s_pcm.p, ds, buf_len, B, h->s_vad.p, pc.stream());
        const uint8_t *back = rc ? nullptr : pc.download(h->s_vad.p, (size_t)B * sizeof(sr_vad_rec));
        const int rcs = pc.finish();
        if (rc) return rc;
        if (rcs) return rcs;
        std::memcpy(vad, back, (size_t)B * sizeof(sr_vad_rec));
        return SR_OK;
    }
    rc = stage_pcm(h, pcm, pcm_stride, buf_len, B, &ds);
    if (rc) return rc;
    if ((rc = sr_vad_batch_dev(h, h->s_pcm.p, ds, buf_len, B, h->s_vad.p, nullptr))) return rc;
    HIP_TRY(hipMemcpy(vad, h->s_vad.p, (size_t)B * sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    return SR_OK;
}

// diagnostics: per-utterance ballots of the "loud" decision (VAD.C:164), 63 frames per 64-bit word
int sr_vad_debug_masks(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                       sr_vad_rec *vad, uint64_t *masks /* [B][16] */)
{
    if (!h || !pcm || !vad || !masks) return fail(SR_ERR_BAD_ARG, "null argument");
    if (B == 0) return SR_OK;
    if (buf_len > pcm_stride) return fail(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    ENTER_DEVICE(h);
    uint64_t ds = 0;
    int rc = stage_pcm(h, pcm, pcm_stride, buf_len, B, &ds);
    if (rc) return rc;
    if ((rc = check_pcm(h, h->s_pcm.p, ds, buf_len))) return rc;
    if ((rc = h->s_vad.reserve(B))) return rc;
    DevBuf<uint64_t> dm;
    if ((rc = dm.reserve((size_t)B * 16))) return rc;
    HIP_TRY(hipMemset(dm.p, 0, (size_t)B * 16 * 8));
    VadArgs a = vad_args(h, h->s_pcm.p, ds, buf_len, h->noise_len, B, h->s_vad.p, nullptr, dm.p);
    launch_vad(a, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(vad, h->s_vad.p, (size_t)B * sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(masks, dm.p, (size_t)B * 16 * 8, hipMemcpyDeviceToHost));
    dm.release();
    return SR_OK;
}

// Per-item failure, as get_mfcc has it (MFCC.C:102-107: a segment shorter than a frame underflows the u32 frame count,
// which then exceeds vv_frm_max -> frm_num = 0): one bad record yields frm_num[b] = 0, an all-zero MFCC record and
// status[b] != 0; the other records of the batch are processed.
int sr_mfcc_batch_status(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                         const int32_t *start, const int32_t *end, const uint32_t *mid, int16_t *mfcc, uint32_t *frm_num,
                         uint32_t *status)
{
    if (!h || !pcm || !start || !end || !mid || !mfcc) return fail(SR_ERR_BAD_ARG, "null argument");
    if (B == 0) return SR_OK;
    if (buf_len > pcm_stride) return fail(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    ENTER_DEVICE(h);
    // build the per-utterance records the frame kernel consumes (what k_vad would have produced)
    std::vector<sr_vad_rec> recs(B);
    for (uint32_t b = 0; b < B; b++) {
        sr_vad_rec &r = recs[b];
        std::memset(&r, 0, sizeof r);
        // samples and mid are 16-bit quantities in the reference (u16 VcBuf, mid_val = a mean of u16 samples,
        // VAD.C:41-47); the frame kernel's 24-bit multiplies rely on |sample - mid| < 2^23
        if (mid[b] > 0xFFFFu) return fail(SR_ERR_BAD_ARG, "mid exceeds the u16 sample range");
        r.atap.mid_val = mid[b];
        for (int i = 0; i < 2 * SR_MAX_SEG; i++) r.seg[i] = -1;
        r.seg[0] = start[b];
        r.seg[1] = end[b];
        if (start[b] < 1 || end[b] > (int32_t)buf_len || end[b] < start[b]) {
            r.status = SR_ST_SEG_OOB;  // outside the buffer (start >= 1: MFCC.C:119 reads start[-1])
        } else {
            // MFCC.C:102: u32 arithmetic, u16 truncation -- a segment shorter than a frame wraps to a count above the cap
            const uint32_t n = ((((uint32_t)(end[b] - start[b]) - h->frame_len) / h->hop) + 1) & 0xFFFF;
            const bool shorter = (uint32_t)(end[b] - start[b]) < h->frame_len;  // the wrapped count may alias a small one
            r.status = (shorter || n > h->cfg.max_frames) ? SR_ST_MFCC_FAIL : SR_ST_OK;  // MFCC.C:103-107
            r.frm_num = r.status == SR_ST_OK ? n : 0;
        }
        if (r.status != SR_ST_OK) r.seg[0] = 1, r.seg[1] = 1;  // never dereferenced (no frames); keep the record harmless
        if (frm_num) frm_num[b] = r.frm_num;
        if (status) status[b] = r.status;
    }
    uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    const size_t mbytes = (size_t)B * h->cfg.max_frames * h->nc * 2;
    int rc;
    if (B <= kPinMaxB && mbytes <= kPinMfccBytes && pin_fits(h, (size_t)B * (ds * 2 + sizeof(sr_vad_rec)), mbytes, 2, 1)) {
        // a few segments (get_mfcc: one): through the pinned area, one synchronisation (see PinCall)
        if ((rc = h->s_vad.reserve(B))) return rc;
        if ((rc = h->s_mfcc.reserve(mbytes / 2))) return rc;
        PinCall pc(h);
        if (!pin_stage_pcm(h, pc, pcm, pcm_stride, buf_len, B, &ds, &rc)) return rc;
        pc.upload(h->s_vad.p, recs.data(), (size_t)B * sizeof(sr_vad_rec));
        rc = sr_mfcc_batch_dev(h, h->s_pcm.p, ds, B, h->s_vad.p, h->s_mfcc.p, pc.stream());
        const uint8_t *back = rc ? nullptr : pc.download(h->s_mfcc.p, mbytes);
        const int rcs = pc.finish();
        if (rc) return rc;
        if (rcs) return rcs;
        std::memcpy(mfcc, back, mbytes);
        return SR_OK;
    }
    rc = stage_pcm(h, pcm, pcm_stride, buf_len, B, &ds);
    if (rc) return rc;
    if ((rc = h->s_vad.reserve(B))) return rc;
    if ((rc = h->s_mfcc.reserve((size_t)B * h->cfg.max_frames * h->nc))) return rc;
    HIP_TRY(hipMemcpy(h->s_vad.p, recs.data(), (size_t)B * sizeof(sr_vad_rec), hipMemcpyHostToDevice));
    if ((rc = sr_mfcc_batch_dev(h, h->s_pcm.p, ds, B, h->s_vad.p, h->s_mfcc.p, nullptr))) return rc;
    HIP_TRY(hipMemcpy(mfcc, h->s_mfcc.p, (size_t)B * h->cfg.max_frames * h->nc * 2, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_mfcc_batch(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                  const int32_t *start, const int32_t *end, const uint32_t *mid, int16_t *mfcc, uint32_t *frm_num)
{
    return sr_mfcc_batch_status(h, pcm, pcm_stride, buf_len, B, start, end, mid, mfcc, frm_num, nullptr);
}

// Template training: save_mdl (main.c:121-138) for n captures + the slot image save_ftr_mdl programs
// (Flash.C:17-67): on success the slot is erased (0xFF) and u16 save_mask | u16 frm_num | frm_num*12 s16 are
// written; on VAD / MFCC failure the slot is left untouched (main.c:126-135).
int sr_train_store(sr_engine *h, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t n,
                   const uint32_t *slot, void *store, uint32_t n_slots, uint32_t stride_bytes, uint32_t *status)
{
    if (!h || !pcm || !slot || !store) return fail(SR_ERR_BAD_ARG, "null argument");
    if (n == 0) return SR_OK;
    if (buf_len > pcm_stride) return fail(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    if (stride_bytes < 4 + 2 * h->nc) return fail(SR_ERR_BAD_ARG, "slot stride too small for a v_ftr_tag");
    const uint32_t slot_rows = (stride_bytes - 4) / (2 * h->nc);
    for (uint32_t i = 0; i < n; i++)
        if (slot[i] >= n_slots) return fail(SR_ERR_BAD_ARG, "slot index outside the store");  // Flash.C:22-26
    ENTER_DEVICE(h);
    uint64_t ds = 0;
    int rc = stage_pcm(h, pcm, pcm_stride, buf_len, n, &ds);
    if (rc) return rc;
    if ((rc = h->s_vad.reserve(n))) return rc;
    const size_t msz = (size_t)n * h->cfg.max_frames * h->nc;
    if ((rc = h->s_mfcc.reserve(msz))) return rc;
    if ((rc = sr_vad_batch_dev(h, h->s_pcm.p, ds, buf_len, n, h->s_vad.p, nullptr))) return rc;
    if ((rc = sr_mfcc_batch_dev(h, h->s_pcm.p, ds, n, h->s_vad.p, h->s_mfcc.p, nullptr))) return rc;
    std::vector<sr_vad_rec> recs(n);
    std::vector<int16_t> mf(msz);
    HIP_TRY(hipMemcpy(recs.data(), h->s_vad.p, (size_t)n * sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mf.data(), h->s_mfcc.p, msz * 2, hipMemcpyDeviceToHost));
    uint8_t *st = (uint8_t *)store;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t code = recs[i].status;  // 0 save_ok, 1 VAD_fail, 2 MFCC_fail (main.c:38-40)
        if (code == SR_ST_OK && recs[i].frm_num > slot_rows) code = SR_ST_MFCC_FAIL;
        if (status) status[i] = code;
        if (code != SR_ST_OK) continue;
        uint8_t *dst = st + (size_t)slot[i] * stride_bytes;
        std::memset(dst, 0xFF, stride_bytes);  // FLASH_ErasePage, Flash.C:32-39
        const uint16_t sign = SR_SAVE_MASK, fr = (uint16_t)recs[i].frm_num;
        std::memcpy(dst, &sign, 2);
        std::memcpy(dst + 2, &fr, 2);
        std::memcpy(dst + 4, &mf[(size_t)i * h->cfg.max_frames * h->nc], (size_t)fr * h->nc * 2);
    }
    return SR_OK;
}

int sr_dtw_batch(sr_engine *h, const int16_t *in_mfcc, const uint32_t *in_frames, uint32_t B, uint32_t *scores,
                 sr_result *results)
{
    if (!h || !in_mfcc || !in_frames || !scores) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    for (uint32_t b = 0; b < B; b++)
        if (in_frames[b] > h->cfg.max_frames) return fail(SR_ERR_BAD_ARG, "in_frames exceeds max_frames");
    ENTER_DEVICE(h);
    int rc;
    const size_t msz = (size_t)B * h->cfg.max_frames * h->nc;
    if ((rc = h->s_mfcc.reserve(msz))) return rc;
    if ((rc = h->s_u32a.reserve(B))) return rc;
    if ((rc = h->s_scores.reserve((size_t)B * h->K))) return rc;
    if ((rc = h->s_results.reserve(B))) return rc;
    const size_t sc_bytes = (size_t)B * h->K * 4, res_bytes = results ? (size_t)B * sizeof(sr_result) : 0;
    if (pin_fits(h, msz * 2 + (size_t)B * 4, sc_bytes + res_bytes, 2, 2)) {  // a few records (dtw(): one): see PinCall
        PinCall pc(h);
        pc.upload(h->s_mfcc.p, in_mfcc, msz * 2);
        pc.upload(h->s_u32a.p, in_frames, (size_t)B * 4);
        DtwArgs a = dtw_args(h, h->s_mfcc.p, nullptr, h->s_u32a.p, B, h->s_scores.p, h->s_results.p);
        if (!launch_dtw_auto(h, a, 0, pc.stream())) launch_argmin(a, pc.stream());
        const hipError_t le = hipGetLastError();
        const uint8_t *sc_back = pc.download(h->s_scores.p, sc_bytes);
        const uint8_t *res_back = results ? pc.download(h->s_results.p, res_bytes) : nullptr;
        if ((rc = pc.finish())) return rc;
        if (le != hipSuccess) return fail(SR_ERR_HIP, hipGetErrorString(le));
        std::memcpy(scores, sc_back, sc_bytes);
        if (results) std::memcpy(results, res_back, res_bytes);
        return SR_OK;
    }
    HIP_TRY(hipMemcpy(h->s_mfcc.p, in_mfcc, msz * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_u32a.p, in_frames, (size_t)B * 4, hipMemcpyHostToDevice));
    DtwArgs a = dtw_args(h, h->s_mfcc.p, nullptr, h->s_u32a.p, B, h->s_scores.p, h->s_results.p);
    if (!launch_dtw_auto(h, a, 0, nullptr)) launch_argmin(a, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(scores, h->s_scores.p, (size_t)B * h->K * 4, hipMemcpyDeviceToHost));
    if (results) HIP_TRY(hipMemcpy(results, h->s_results.p, (size_t)B * sizeof(sr_result), hipMemcpyDeviceToHost));
    return SR_OK;
}

// get_mdl (DTW.C:217-296): merge pairs of feature records along their greedy DTW path
int sr_get_mdl_batch(sr_engine *h, const int16_t *in1, const uint32_t *n1, uint32_t rows1, const int16_t *in2,
                     const uint32_t *n2, uint32_t rows2, uint32_t P, int16_t *mdl, uint32_t mdl_rows,
                     uint32_t *mdl_frames, uint32_t *dis)
{
    if (!h || !in1 || !n1 || !in2 || !n2 || !mdl_frames || !dis || (mdl_rows && !mdl))
        return fail(SR_ERR_BAD_ARG, "null argument");
    if (P == 0) return SR_OK;
    if (h->nc != (uint32_t)kCoef) return fail(SR_ERR_BAD_CONFIG, "get_mdl is built for 12-coefficient records");
    if (rows1 == 0 || rows2 == 0) return fail(SR_ERR_BAD_ARG, "rows1 / rows2 must be at least 1");
    for (uint32_t p = 0; p < P; p++)
        if (n1[p] > rows1 || n2[p] > rows2 || n1[p] > 0xFFFF || n2[p] > 0xFFFF)
            return fail(SR_ERR_BAD_ARG, "frame count exceeds the rows of its record (or the u16 range)");
    ENTER_DEVICE(h);
    int rc;
    const size_t e1 = (size_t)P * rows1 * kCoef, e2 = (size_t)P * rows2 * kCoef, eo = (size_t)P * mdl_rows * kCoef;
    if ((rc = h->s_mfcc.reserve(e1 + e2 + eo + 16))) return rc;
    if ((rc = h->s_u32a.reserve((size_t)2 * P))) return rc;
    if ((rc = h->s_u32b.reserve((size_t)2 * P))) return rc;
    int16_t *d1 = h->s_mfcc.p, *d2 = d1 + ((e1 + 3) & ~(size_t)3), *dm = d2 + ((e2 + 3) & ~(size_t)3);  // 8-byte aligned rows
    HIP_TRY(hipMemcpy(d1, in1, e1 * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d2, in2, e2 * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_u32a.p, n1, (size_t)P * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_u32a.p + P, n2, (size_t)P * 4, hipMemcpyHostToDevice));
    if (eo) HIP_TRY(hipMemset(dm, 0, eo * 2));
    GetMdlArgs a{d1, h->s_u32a.p, rows1, d2, h->s_u32a.p + P, rows2, P, dm, mdl_rows, h->s_u32b.p, h->s_u32b.p + P};
    launch_get_mdl(a, nullptr);
    HIP_TRY(hipGetLastError());
    if (eo) HIP_TRY(hipMemcpy(mdl, dm, eo * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mdl_frames, h->s_u32b.p, (size_t)P * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dis, h->s_u32b.p + P, (size_t)P * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

// OPT-IN, NON-REFERENCE: full dynamic-programming DTW with the reference's parallelogram and local distance
// (see k_dtw_dp).  Never used by sr_recognize_* or the dtw() symbol.
int sr_dtw_dp_batch_dev(sr_engine *h, const int16_t *d_mfcc, const uint32_t *d_in_frames, const sr_vad_rec *d_vad,
                        uint32_t B, uint32_t *d_scores, void *stream)
{
    if (!h || !d_mfcc || !d_scores || (!d_in_frames && !d_vad)) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (h->nc != (uint32_t)kCoef) return fail(SR_ERR_BAD_CONFIG, "the full-DP scorer is built for 12-coefficient records");
    if ((size_t)h->tpl_rows * 48 > 150 * 1024) return fail(SR_ERR_BAD_ARG, "templates too long for the LDS-staged DP kernel");
    ENTER_DEVICE(h);
    DtwArgs a = dtw_args(h, d_mfcc, d_vad, d_in_frames, B, d_scores, nullptr);
    if (!h->tpl_staged_ok) a.tplR = nullptr;  // coefficients beyond +-16383: the band kernel's -2*coef rows do not hold them
    launch_dtw_dp(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

int sr_dtw_dp_batch(sr_engine *h, const int16_t *in_mfcc, const uint32_t *in_frames, uint32_t B, uint32_t *scores)
{
    if (!h || !in_mfcc || !in_frames || !scores) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    for (uint32_t b = 0; b < B; b++)
        if (in_frames[b] > h->cfg.max_frames) return fail(SR_ERR_BAD_ARG, "in_frames exceeds max_frames");
    ENTER_DEVICE(h);
    int rc;
    const size_t msz = (size_t)B * h->cfg.max_frames * h->nc;
    if ((rc = h->s_mfcc.reserve(msz))) return rc;
    if ((rc = h->s_u32a.reserve(B))) return rc;
    if ((rc = h->s_scores.reserve((size_t)B * h->K))) return rc;
    HIP_TRY(hipMemcpy(h->s_mfcc.p, in_mfcc, msz * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_u32a.p, in_frames, (size_t)B * 4, hipMemcpyHostToDevice));
    if ((rc = sr_dtw_dp_batch_dev(h, h->s_mfcc.p, h->s_u32a.p, nullptr, B, h->s_scores.p, nullptr))) return rc;
    HIP_TRY(hipMemcpy(scores, h->s_scores.p, (size_t)B * h->K * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

// EXTENSION (no reference counterpart): delta cepstra, see k_delta_mfcc
int sr_delta_mfcc_batch_dev(sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad, const uint32_t *d_frames,
                            uint32_t B, int16_t *d_delta, void *stream)
{
    if (!h || !d_mfcc || !d_delta || (!d_vad && !d_frames)) return fail(SR_ERR_BAD_ARG, "null argument");
    if ((uint64_t)B * h->cfg.max_frames * h->nc > 0xFFFFFFFFull * 256) return fail(SR_ERR_BAD_ARG, "batch too large");
    ENTER_DEVICE(h);
    launch_delta_mfcc(d_mfcc, d_vad, d_frames, B, h->cfg.max_frames, h->nc, d_delta, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SR_OK;
}

int sr_delta_mfcc_batch(sr_engine *h, const int16_t *mfcc, const uint32_t *frames, uint32_t B, int16_t *delta)
{
    if (!h || !mfcc || !frames || !delta) return fail(SR_ERR_BAD_ARG, "null argument");
    if (B == 0) return SR_OK;
    ENTER_DEVICE(h);
    int rc;
    const size_t msz = (size_t)B * h->cfg.max_frames * h->nc;
    if ((rc = h->s_mfcc.reserve(2 * msz))) return rc;
    if ((rc = h->s_u32a.reserve(B))) return rc;
    HIP_TRY(hipMemcpy(h->s_mfcc.p, mfcc, msz * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_u32a.p, frames, (size_t)B * 4, hipMemcpyHostToDevice));
    if ((rc = sr_delta_mfcc_batch_dev(h, h->s_mfcc.p, nullptr, h->s_u32a.p, B, h->s_mfcc.p + msz, nullptr))) return rc;
    HIP_TRY(hipMemcpy(delta, h->s_mfcc.p + msz, msz * 2, hipMemcpyDeviceToHost));
    return SR_OK;
}

// diagnostics: out[3*i + {0,1,2}] = (u32)(log(x)*100), (u32)sqrtf(x), (u32)(sqrtf((s32)x)*10) as the kernels compute them
int sr_math_diag(sr_engine *h, const uint32_t *in, uint32_t *out, uint32_t n)
{
    if (!h || !in || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    if (n == 0) return SR_OK;
    ENTER_DEVICE(h);
    int rc;
    if ((rc = h->s_u32a.reserve(n))) return rc;
    if ((rc = h->s_u32b.reserve((size_t)3 * n))) return rc;
    HIP_TRY(hipMemcpy(h->s_u32a.p, in, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_math_diag(h->s_u32a.p, h->s_u32b.p, n, h->dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32b.p, (size_t)3 * n * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

// diagnostics: k_mfcc's fused filterbank term against the reference's expression, see k_mel_term_sweep
int sr_mel_term_sweep(sr_engine *h, uint32_t tri_lo, uint32_t tri_hi, uint32_t e_max, uint64_t *mismatches)
{
    if (!h || !mismatches) return fail(SR_ERR_BAD_ARG, "null argument");
    if (tri_hi <= tri_lo || tri_hi - tri_lo > 65535u || tri_hi - 1 > kMelTriMax || e_max >= (1u << 28))
        return fail(SR_ERR_BAD_ARG, "sr_mel_term_sweep: weights must lie in [0, 1599], at most 65535 of them, and E below 2^28");
    ENTER_DEVICE(h);
    const uint32_t n = tri_hi - tri_lo;
    int rc;
    if ((rc = h->s_u32a.reserve((size_t)2 * n))) return rc;
    HIP_TRY(hipMemset(h->s_u32a.p, 0, (size_t)n * 8));
    launch_mel_term_sweep(tri_lo, n, e_max, (unsigned long long *)h->s_u32a.p, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(mismatches, h->s_u32a.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_fft_q15_batch(sr_engine *h, const uint32_t *in, uint32_t *out, uint32_t n)
{
    if (!h || !in || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    if (n == 0) return SR_OK;
    ENTER_DEVICE(h);
    int rc;
    if ((rc = h->s_u32a.reserve((size_t)n * kNfft))) return rc;
    if ((rc = h->s_u32b.reserve((size_t)n * kNfft))) return rc;
    HIP_TRY(hipMemcpy(h->s_u32a.p, in, (size_t)n * kNfft * 4, hipMemcpyHostToDevice));
    launch_fft_q15(h->s_u32a.p, h->s_u32b.p, n, h->dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32b.p, (size_t)n * kNfft * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

}  // extern "C"

// internal hooks for sr_compat.cpp
namespace sr {
int engine_fft_mag(sr_engine *h, const int16_t *frame, uint32_t len, uint32_t *mag, uint32_t *raw_hi)
{
    ENTER_DEVICE(h);
    int rc;
    if ((rc = h->s_mfcc.reserve(len > 0 ? len : 1))) return rc;
    if ((rc = h->s_u32a.reserve(kBins))) return rc;
    if ((rc = h->s_u32b.reserve(kBins))) return rc;
    if (len) HIP_TRY(hipMemcpy(h->s_mfcc.p, frame, (size_t)len * 2, hipMemcpyHostToDevice));
    launch_fft_mag(h->s_mfcc.p, len, h->s_u32a.p, h->s_u32b.p, 1, h->dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(mag, h->s_u32a.p, kBins * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(raw_hi, h->s_u32b.p, kBins * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int engine_get_dis(sr_engine *h, const int16_t *a, const int16_t *b, uint32_t *out)
{
    ENTER_DEVICE(h);
    int rc;
    if ((rc = h->s_mfcc.reserve(2 * kCoef))) return rc;
    if ((rc = h->s_u32a.reserve(1))) return rc;
    HIP_TRY(hipMemcpy(h->s_mfcc.p, a, kCoef * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_mfcc.p + kCoef, b, kCoef * 2, hipMemcpyHostToDevice));
    launch_get_dis(h->s_mfcc.p, h->s_mfcc.p + kCoef, h->s_u32a.p, 1, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32a.p, 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int engine_dtw_limit(sr_engine *h, uint16_t x, uint16_t y, int X1, int X2, int in_n, int mdl_n, uint8_t *out)
{
    ENTER_DEVICE(h);
    int rc;
    if ((rc = h->s_u32a.reserve(2))) return rc;
    const uint16_t xy[2] = {x, y};
    HIP_TRY(hipMemcpy(h->s_u32a.p, xy, 4, hipMemcpyHostToDevice));
    launch_dtw_limit((const uint16_t *)h->s_u32a.p, (uint8_t *)(h->s_u32a.p + 1), 1, X1, X2, in_n, mdl_n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32a.p + 1, 1, hipMemcpyDeviceToHost));
    return SR_OK;
}

int engine_vad_with_atap(sr_engine *h, const uint16_t *pcm, uint32_t buf_len, const sr_atap *atap, sr_vad_rec *rec)
{
    ENTER_DEVICE(h);
    uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    int rc;
    if ((rc = h->s_vad.reserve(1))) return rc;
    if ((rc = h->s_atap.reserve(1))) return rc;
    if (pin_fits(h, (size_t)ds * 2 + sizeof(sr_atap), sizeof(sr_vad_rec), 2, 1)) {  // VAD(): one capture, see PinCall
        PinCall pc(h);
        if (!pin_stage_pcm(h, pc, pcm, buf_len, buf_len, 1, &ds, &rc)) return rc;
        pc.upload(h->s_atap.p, atap, sizeof(sr_atap));
        VadArgs a = vad_args(h, h->s_pcm.p, ds, buf_len, h->noise_len, 1, h->s_vad.p, h->s_atap.p);
        launch_vad(a, pc.stream());
        const hipError_t le = hipGetLastError();
        const uint8_t *back = pc.download(h->s_vad.p, sizeof(sr_vad_rec));
        if ((rc = pc.finish())) return rc;
        if (le != hipSuccess) return fail(SR_ERR_HIP, hipGetErrorString(le));
        std::memcpy(rec, back, sizeof(sr_vad_rec));
        return SR_OK;
    }
    rc = stage_pcm(h, pcm, buf_len, buf_len, 1, &ds);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h->s_atap.p, atap, sizeof(sr_atap), hipMemcpyHostToDevice));
    VadArgs a = vad_args(h, h->s_pcm.p, ds, buf_len, h->noise_len, 1, h->s_vad.p, h->s_atap.p);
    launch_vad(a, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rec, h->s_vad.p, sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    return SR_OK;
}

// noise_atap alone: run the VAD kernel on the noise head only (buf_len = n_len gives F frames of no
// interest; only the atap part of the record is used)
int engine_noise_atap(sr_engine *h, const uint16_t *noise, uint32_t n_len, sr_atap *out)
{
    ENTER_DEVICE(h);
    uint64_t ds = ((uint64_t)n_len + 7) & ~7ull;
    int rc;
    if ((rc = h->s_vad.reserve(1))) return rc;
    if (pin_fits(h, (size_t)ds * 2, sizeof(sr_vad_rec))) {  // noise_atap(): one noise head, see PinCall
        PinCall pc(h);
        if (!pin_stage_pcm(h, pc, noise, n_len, n_len, 1, &ds, &rc)) return rc;
        VadArgs a = vad_args(h, h->s_pcm.p, ds, n_len, n_len, 1, h->s_vad.p);
        launch_vad(a, pc.stream());
        const hipError_t le = hipGetLastError();
        const uint8_t *back = pc.download(h->s_vad.p, sizeof(sr_vad_rec));
        if ((rc = pc.finish())) return rc;
        if (le != hipSuccess) return fail(SR_ERR_HIP, hipGetErrorString(le));
        *out = ((const sr_vad_rec *)back)->atap;
        return SR_OK;
    }
    rc = stage_pcm(h, noise, n_len, n_len, 1, &ds);
    if (rc) return rc;
    VadArgs a = vad_args(h, h->s_pcm.p, ds, n_len, n_len, 1, h->s_vad.p);
    launch_vad(a, nullptr);
    HIP_TRY(hipGetLastError());
    sr_vad_rec rec;
    HIP_TRY(hipMemcpy(&rec, h->s_vad.p, sizeof(sr_vad_rec), hipMemcpyDeviceToHost));
    *out = rec.atap;
    return SR_OK;
}
}  // namespace sr
