// Host side of the C ABI declared in include/sr_engine.h: device tables, template store, staging
// buffers, kernel sequencing on a HIP stream.  No CPU implementation of the recognition path exists
// in this library; every entry point needs a gfx950 device.
#include "sr_engine_internal.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace sr {

static thread_local std::string g_err;
int set_error(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#ifdef SR_TESTING
static std::atomic<int64_t> g_hooks[kHookCount];
int64_t dev_hook(DevHook h) { return g_hooks[h].load(std::memory_order_relaxed); }
static const char *const kHookNames[kHookCount] = {"dtw_u", "dtw_tie_g", "dtw_kc", "mfcc_grid", "perturb_log_thr",
                                                   "log_thr_from_host", "multi_allow_dup", "dtw_debug", "cells_literal",
                                                   "mag_cheap_off"};
#endif
}  // namespace sr

using namespace sr;

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
uint32_t sr_mag_cheap_bound(const sr_engine *h) { return h ? h->mag_cheap_max : 0u; }

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
        return fail(SR_ERR_BAD_CONFIG, "noise_len_ms: the noise head (" + std::to_string(noise_len) + " samples) must be a non-zero multiple of the 30 ms "
                                       "block of noise_atap (" + std::to_string(atap_frm) + " samples, VAD.C:48-63) and of the frame length (" +
                                       std::to_string(fe.frame_len) + " samples)");
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
    {  // what launch-shape decisions need to know about THIS device (a partitioned or CU-masked part is not 256 CUs)
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) h->n_cu = (uint32_t)v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0) h->lds_per_cu = (uint32_t)v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
            h->lds_per_wg = std::max<uint32_t>((uint32_t)v, 64u * 1024u) > h->lds_per_cu ? h->lds_per_cu : std::max<uint32_t>((uint32_t)v, 64u * 1024u);
        (void)hipGetLastError();
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
    // k_mfcc looks the sums of the lanes BELOW a filter edge's lane up at index lane - 1 (inclusive lane sums): every edge it
    // looks up must lie in lane >= 1, i.e. the first centre at bin 9 or later (11 for the reference's tables)
    if (!h->generic && h->frame_len == (uint32_t)kFrameLen && (t.tri_cen.empty() || t.tri_cen[0] < 9)) {
        delete h;
        return fail(SR_ERR_BAD_CONFIG, "internal: first Mel centre below bin 9");
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
    // The cheap magnitude form of k_mfcc's QUIET / MID tiers rests on a property of this chip's v_sqrt_f32, so it is checked
    // here, on the device the engine will run on, over the WHOLE range it is used on (70 172 values, microseconds): any
    // difference from the exactly corrected root and every frame takes the exact form (bound 0) -- parity never depends on a
    // stepping or a microcode revision the test suite has not seen.  sr_mag_cheap_bound() reports the outcome.
    if (!h->generic && h->frame_len == (uint32_t)kFrameLen) {
        uint32_t *sw = nullptr;
        const uint32_t init[4] = {0, 0, 0xFFFFFFFFu, 0};
        uint32_t back[4] = {1, 0, 0, 0};
        hipError_t se = hipMalloc(&sw, sizeof init);
        if (se == hipSuccess) se = hipMemcpy(sw, init, sizeof init, hipMemcpyHostToDevice);
        if (se == hipSuccess) {
            launch_mag_fast_sweep(kMagCheapMax, (unsigned long long *)sw, sw + 2, nullptr);
            se = hipGetLastError();
        }
        if (se == hipSuccess) se = hipMemcpy(back, sw, sizeof back, hipMemcpyDeviceToHost);
        if (sw) (void)hipFree(sw);
        if (se != hipSuccess) {
            sr_destroy(h);
            return fail(SR_ERR_HIP, std::string("magnitude sweep: ") + hipGetErrorString(se));
        }
        const bool ok = back[0] == 0 && back[1] == 0;
        h->mag_cheap_max = ok ? kMagCheapMax : 0u;
        if (dev_hook(kHookMagCheapOff)) h->mag_cheap_max = 0;  // development hook: exercise the fallback
        if (!ok) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "sr_create: v_sqrt_f32 magnitude differs from the exact form at n = %u on this device; "
                          "every frame takes the exact root", back[2]);
            std::fprintf(stderr, "%s\n", msg);
            (void)fail(SR_OK, msg);
        }
    }
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
    if (h->ev_scratch) (void)hipEventDestroy(h->ev_scratch);
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
    // Large stores: the DTW is most of a step (70 % at K = 500), every chunk adds one drain of its long workgroups, and there
    // is little left to overlap it with: one chunk per stream.  Measured at 65 536 x 500 (profiles/experiments/RESULTS.md):
    // 3 streams x 3 chunks 44.4-44.6 ms, x 6: 44.8-44.9, x 12: 45.2; x 4 (one chunk left over on one stream): 45.4.
    if (!h->pipe_user_set) h->pipe_max_chunks = K >= 256 ? h->pipe_streams : 12;
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
    if (mode < 0 || mode > 3)
        return fail(SR_ERR_BAD_ARG, "small-launch mode: 0 (automatic), 1 (never), 2 (one workgroup per pair whenever the store fits), "
                                    "3 (four lanes per pair whenever the store fits)");
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

}  // extern "C"
