// The reference's own scalar entry points (exact names and signatures), each a 1-item dispatch of
// the HIP kernels through an implicit default engine built with the firmware's constants
// (ADC.H:7-11, VAD.H:4-8, MFCC.H:7-16, Flash.H:11-20).  Like the firmware they are not re-entrant.
// There is no error channel in these signatures: if no gfx950 device can be used the process is
// stopped with a message rather than silently computing on the CPU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "sr_device.h"
#include "sr_tables.h"

struct sr_engine;
namespace sr {
int engine_fft_mag(sr_engine *h, const int16_t *frame, uint32_t len, uint32_t *mag, uint32_t *raw_hi);
int engine_get_dis(sr_engine *h, const int16_t *a, const int16_t *b, uint32_t *out);
int engine_dtw_limit(sr_engine *h, uint16_t x, uint16_t y, int X1, int X2, int in_n, int mdl_n, uint8_t *out);
int engine_vad_with_atap(sr_engine *h, const uint16_t *pcm, uint32_t buf_len, const sr_atap *atap, sr_vad_rec *rec);
int engine_noise_atap(sr_engine *h, const uint16_t *noise, uint32_t n_len, sr_atap *out);
}  // namespace sr

namespace {

constexpr uint32_t kVcBufLen = 16000;  // ADC.H:9
constexpr uint32_t kAtapFrm = 240;     // VAD.C:13-14

sr_engine *g_engine = nullptr;  // recognition engine (holds the compat template store)
sr_engine *g_pair = nullptr;    // engine whose store is the single model of a dtw() call

[[noreturn]] void die(const char *what)
{
    std::fprintf(stderr, "sr_engine (reference-compatible symbol %s): %s\n", what, sr_last_error());
    std::abort();
}

sr_engine *engine(const char *who)
{
    if (!g_engine) {
        sr_config c;
        sr_default_config(&c);
        if (sr_create(&c, &g_engine) != SR_OK) die(who);
    }
    return g_engine;
}
sr_engine *pair_engine(const char *who)
{
    if (!g_pair) {
        sr_config c;
        sr_default_config(&c);
        if (sr_create(&c, &g_pair) != SR_OK) die(who);
    }
    return g_pair;
}

// DTW.C:65-68 file statics, observable through dtw_limit()
uint16_t g_X1 = 0, g_X2 = 0, g_in_frm = 0, g_mdl_frm = 0;

uint32_t g_fft_out[1024];  // MFCC.C:14, fft() returns a pointer to it

// commstr[] of main.c:31 -- 3-byte labels, GB2312 for the eight direction/size words
const uint8_t kDefaultLabels[18][3] = {
    {'0', ' ', 0}, {'1', ' ', 0}, {'2', ' ', 0}, {'3', ' ', 0}, {'4', ' ', 0}, {'5', ' ', 0},
    {'6', ' ', 0}, {'7', ' ', 0}, {'8', ' ', 0}, {'9', ' ', 0}, {0xC9, 0xCF, 0}, {0xCF, 0xC2, 0},
    {0xC7, 0xB0, 0}, {0xBA, 0xF3, 0}, {0xD7, 0xF3, 0}, {0xD3, 0xD2, 0}, {0xB4, 0xF3, 0}, {0xD0, 0xA1, 0}};
std::vector<uint8_t> g_labels(&kDefaultLabels[0][0], &kDefaultLabels[0][0] + sizeof kDefaultLabels);
uint32_t g_n_labels = 18, g_label_stride = 3, g_ftr_per_comm = 4;  // Flash.H:15
uint8_t g_empty_label[1] = {0};

// dtw(): models seen so far, addressed by content (see dtw() below)
constexpr uint32_t kRecWords = SR_VV_FRM_MAX * 12;  // s16 per v_ftr_tag record, MFCC.H:23
constexpr size_t kMaxCachedModels = 128;            // comm_num * ftr_per_comm = 80 in the firmware (Flash.H:15-17)
struct ModelStore {
    std::vector<int16_t> rows;     // [n][kRecWords]
    std::vector<uint32_t> frames;  // [n]
    std::vector<uint64_t> hash;    // [n]
    std::vector<uint64_t> used;    // [n] tick of the last hit (least recently used record is replaced when full)
    uint64_t tick = 0;
    bool dirty = false;            // rows changed since the last upload
    uint64_t gen = 0;              // bumped whenever the store changes
    // scores of the last input record against every cached model
    std::vector<int16_t> in_rows;
    uint32_t in_frames = 0;
    uint64_t memo_gen = ~0ull;
    std::vector<uint32_t> scores;
    uint32_t uploads = 0, launches = 0;

    static uint64_t fnv(const int16_t *p, uint32_t frm)
    {
        uint64_t h = 1469598103934665603ull ^ frm;
        const uint8_t *b = (const uint8_t *)p;
        for (size_t i = 0; i < kRecWords * sizeof(int16_t); i++) h = (h ^ b[i]) * 1099511628211ull;
        return h;
    }
    // Where the caller's record at address `key` was last found.  The firmware's slot scan hands the SAME 80 flash
    // addresses to dtw() for every utterance (main.c:279-291): a pointer hit is confirmed by comparing the record (one
    // 2.8 KB memcmp, ~0.1 us) and skips the hash of the record (~3 us per call, 240 us per slot scan).
    std::unordered_map<const void *, size_t> by_addr;
    size_t find_or_add(const int16_t *p, uint32_t frm, const void *key)
    {
        auto it = by_addr.find(key);
        if (it != by_addr.end() && it->second < frames.size() && frames[it->second] == frm &&
            std::memcmp(&rows[it->second * kRecWords], p, kRecWords * sizeof(int16_t)) == 0) {
            used[it->second] = ++tick;
            return it->second;
        }
        const size_t slot = find_or_add_hashed(p, frm);
        if (by_addr.size() > 4096) by_addr.clear();  // callers that pass ever-new addresses: keep the map small
        by_addr[key] = slot;
        return slot;
    }
    size_t find_or_add_hashed(const int16_t *p, uint32_t frm)
    {
        const uint64_t hv = fnv(p, frm);
        for (size_t i = 0; i < frames.size(); i++)
            if (hash[i] == hv && frames[i] == frm && std::memcmp(&rows[i * kRecWords], p, kRecWords * sizeof(int16_t)) == 0) {
                used[i] = ++tick;
                return i;
            }
        dirty = true;
        gen++;
        if (frames.size() == kMaxCachedModels) {  // full: the least recently used record makes room (a cyclic scan over
            size_t v = 0;                         // more slots than the store holds still misses every time, but 127
            for (size_t i = 1; i < used.size(); i++)  // records stay valid instead of none)
                if (used[i] < used[v]) v = i;
            std::memcpy(&rows[v * kRecWords], p, kRecWords * sizeof(int16_t));
            frames[v] = frm;
            hash[v] = hv;
            used[v] = ++tick;
            return v;
        }
        rows.insert(rows.end(), p, p + kRecWords);
        frames.push_back(frm);
        hash.push_back(hv);
        used.push_back(++tick);
        return frames.size() - 1;
    }
} g_models;

}  // namespace

extern "C" {

sr_engine *sr_compat_engine(void) { return engine("sr_compat_engine"); }

int sr_compat_set_templates(const void *store, uint32_t n_slots, uint32_t stride_bytes)
{
    return sr_set_templates(engine("sr_compat_set_templates"), store, n_slots, stride_bytes);
}

int sr_compat_set_labels(const uint8_t *labels, uint32_t n_labels, uint32_t label_stride, uint32_t ftr_per_comm)
{
    if (!labels || !n_labels || !label_stride || !ftr_per_comm) return SR_ERR_BAD_ARG;
    g_labels.assign(labels, labels + (size_t)n_labels * label_stride);
    g_n_labels = n_labels;
    g_label_stride = label_stride;
    g_ftr_per_comm = ftr_per_comm;
    return SR_OK;
}

// VAD.C:22-71
void noise_atap(const uint16_t *noise, uint16_t n_len, atap_tag *atap)
{
    if (n_len == 0 || (n_len % kAtapFrm) != 0) return;  // VAD.C:33-36: silent return
    sr_atap a;
    if (sr::engine_noise_atap(engine("noise_atap"), noise, n_len, &a) != SR_OK) die("noise_atap");
    atap->mid_val = a.mid_val;
    atap->n_thl = a.n_thl;
    atap->z_thl = a.z_thl;
    atap->s_thl = a.s_thl;
}

// VAD.C:97-218
void VAD(const uint16_t *vc, uint16_t buf_len, valid_tag *valid_voice, atap_tag *atap_arg)
{
    for (int i = 0; i < SR_MAX_SEG; i++) valid_voice[i].start = valid_voice[i].end = nullptr;  // VAD.C:115-119
    if (buf_len <= sr::kFrameLen) return;                                                      // loop of VAD.C:121 is empty
    sr_atap a{atap_arg->mid_val, atap_arg->n_thl, atap_arg->z_thl, atap_arg->s_thl};
    sr_vad_rec rec;
    if (sr::engine_vad_with_atap(engine("VAD"), vc, buf_len, &a, &rec) != SR_OK) die("VAD");
    for (int i = 0; i < SR_MAX_SEG; i++) {
        if (rec.seg[2 * i] != -1) valid_voice[i].start = (uint16_t *)vc + rec.seg[2 * i];
        if (rec.seg[2 * i + 1] != -1) valid_voice[i].end = (uint16_t *)vc + rec.seg[2 * i + 1];
    }
}

// MFCC.C:86-191
void get_mfcc(valid_tag *valid, v_ftr_tag *v_ftr, atap_tag *atap_arg)
{
    const ptrdiff_t len = valid->end - valid->start;
    if (len < (ptrdiff_t)sr::kFrameLen) {  // MFCC.C:102-113: either the cap trips or the loop body never runs
        v_ftr->frm_num = 0;
        return;
    }
    sr_engine *h = engine("get_mfcc");
    const int32_t start = 1, end = (int32_t)len + 1;  // buffer handed over begins at start[-1] (MFCC.C:119)
    const uint32_t mid = atap_arg->mid_val;
    std::vector<int16_t> out((size_t)SR_VV_FRM_MAX * 12);
    uint32_t n = 0;
    if (sr_mfcc_batch(h, valid->start - 1, (uint64_t)len + 1, (uint32_t)len + 1, 1, &start, &end, &mid, out.data(), &n) !=
        SR_OK)
        die("get_mfcc");
    v_ftr->frm_num = (uint16_t)n;  // 0 when the segment exceeds vv_frm_max (MFCC.C:103-107)
    if (n) std::memcpy(v_ftr->mfcc_dat, out.data(), (size_t)n * 12 * sizeof(int16_t));
}
void GetMfcc(valid_tag *valid, v_ftr_tag *v_ftr, atap_tag *atap_arg) { get_mfcc(valid, v_ftr, atap_arg); }
void MFCC_Comp(valid_tag *valid, v_ftr_tag *v_ftr, atap_tag *atap_arg) { get_mfcc(valid, v_ftr, atap_arg); }

// MFCC.C:27-62
uint32_t *fft(int16_t *dat_buf, uint16_t buf_len)
{
    if (buf_len > sr::kNfft) return nullptr;  // MFCC.C:32-35
    if (sr::engine_fft_mag(engine("fft"), dat_buf, buf_len, g_fft_out, g_fft_out + sr::kBins) != SR_OK) die("fft");
    return g_fft_out;
}

// cr4_fft_1024_stm32.s:219-281
void cr4_fft_1024_stm32(void *pssOUT, void *pssIN, uint16_t Nbin)
{
    (void)Nbin;  // "this optimized FFT function can only convert 1024 points" (.s:214-215)
    if (sr_fft_q15_batch(engine("cr4_fft_1024_stm32"), (const uint32_t *)pssIN, (uint32_t *)pssOUT, 1) != SR_OK)
        die("cr4_fft_1024_stm32");
}

// DTW.C:45-62
uint32_t get_dis(int16_t *frm_ftr1, int16_t *frm_ftr2)
{
    uint32_t d = 0;
    if (sr::engine_get_dis(engine("get_dis"), frm_ftr1, frm_ftr2, &d) != SR_OK) die("get_dis");
    return d;
}

// DTW.C:76-109 (uses the statics left by the last dtw() call)
uint8_t dtw_limit(uint16_t x, uint16_t y)
{
    uint8_t o = 0;
    if (sr::engine_dtw_limit(engine("dtw_limit"), x, y, g_X1, g_X2, g_in_frm, g_mdl_frm, &o) != SR_OK) die("dtw_limit");
    return o;
}

// DTW.C:120-192
uint32_t dtw(v_ftr_tag *ftr_in, v_ftr_tag *frt_mdl)
{
    g_in_frm = ftr_in->frm_num;
    g_mdl_frm = frt_mdl->frm_num;
    if (g_in_frm > g_mdl_frm * 2 || 2 * g_in_frm < g_mdl_frm) return SR_DIS_ERR;  // DTW.C:133-137
    g_X1 = (uint16_t)((2 * g_mdl_frm - g_in_frm) / 3);
    g_X2 = (uint16_t)((4 * g_in_frm - 2 * g_mdl_frm) / 3);
    sr_engine *h = pair_engine("dtw");
    const uint32_t mf = frt_mdl->frm_num, inf = ftr_in->frm_num;
    if (mf > SR_VV_FRM_MAX || inf > SR_VV_FRM_MAX || inf == 0) {
        std::fprintf(stderr, "dtw: frm_num outside the v_ftr_tag capacity\n");
        std::abort();
    }
    // The firmware calls dtw() once per flash slot and utterance (main.c:279-291).  Models are kept in a small
    // content-addressed store on the device (all 119 rows of a record: rows past frm_num are read as in DTW.C:152-154),
    // so a slot is uploaded the first time it is seen, not on every call; and one launch scores an input record against
    // every cached model, so the other 79 calls of the slot scan are look-ups.  Hits are confirmed by comparing the
    // full records, never by the hash alone.
    const size_t slot = g_models.find_or_add(frt_mdl->mfcc_dat, mf, frt_mdl);
    if (g_models.dirty) {
        const uint32_t Kc = (uint32_t)g_models.frames.size();
        if (sr_set_templates_dense(h, g_models.rows.data(), g_models.frames.data(), nullptr, Kc, kRecWords) != SR_OK) die("dtw");
        g_models.dirty = false;
        g_models.uploads++;
    }
    if (!(g_models.memo_gen == g_models.gen && g_models.in_frames == inf &&
          std::memcmp(g_models.in_rows.data(), ftr_in->mfcc_dat, kRecWords * sizeof(int16_t)) == 0)) {
        g_models.scores.assign(g_models.frames.size(), 0u);
        if (sr_dtw_batch(h, ftr_in->mfcc_dat, &inf, 1, g_models.scores.data(), nullptr) != SR_OK) die("dtw");
        g_models.in_rows.assign(ftr_in->mfcc_dat, ftr_in->mfcc_dat + kRecWords);
        g_models.in_frames = inf;
        g_models.memo_gen = g_models.gen;
        g_models.launches++;
    }
    return g_models.scores[slot];
}

// diagnostics for tests: out[0] = uploads of the dtw() model store, out[1] = DTW launches, out[2] = cached models
void sr_compat_dtw_stats(uint32_t out[3])
{
    out[0] = g_models.uploads;
    out[1] = g_models.launches;
    out[2] = (uint32_t)g_models.frames.size();
}

// DTW.C:217-296.  ftr_mdl->save_sign is left alone, frm_num = step (DTW.C:293).  A merged template longer than
// the 119-frame record (the reference would write past it) is a fatal error here.
uint32_t get_mdl(v_ftr_tag *ftr_in1, v_ftr_tag *ftr_in2, v_ftr_tag *ftr_mdl)
{
    sr_engine *h = pair_engine("get_mdl");
    const uint32_t n1 = ftr_in1->frm_num, n2 = ftr_in2->frm_num;
    if (n1 > SR_VV_FRM_MAX || n2 > SR_VV_FRM_MAX) {
        std::fprintf(stderr, "get_mdl: frm_num outside the v_ftr_tag capacity\n");
        std::abort();
    }
    g_in_frm = (uint16_t)n1;  // DTW.C:229-230: the file statics dtw_limit() reads
    g_mdl_frm = (uint16_t)n2;
    if (n1 > n2 * 2 || 2 * n1 < n2) return SR_DIS_ERR;  // DTW.C:232-235, ftr_mdl untouched
    g_X1 = (uint16_t)((2 * g_mdl_frm - g_in_frm) / 3);
    g_X2 = (uint16_t)((4 * g_in_frm - 2 * g_mdl_frm) / 3);
    if (n1 == 0 || n2 == 0) {
        std::fprintf(stderr, "get_mdl: empty record\n");
        std::abort();
    }
    std::vector<int16_t> out((size_t)SR_VV_FRM_MAX * 12);
    uint32_t frames = 0, dis = 0;
    if (sr_get_mdl_batch(h, ftr_in1->mfcc_dat, &n1, SR_VV_FRM_MAX, ftr_in2->mfcc_dat, &n2, SR_VV_FRM_MAX, 1, out.data(),
                         SR_VV_FRM_MAX, &frames, &dis) != SR_OK)
        die("get_mdl");
    if (frames > SR_VV_FRM_MAX) {
        std::fprintf(stderr, "get_mdl: merged template has %u frames, the v_ftr_tag record holds %d\n", frames, SR_VV_FRM_MAX);
        std::abort();
    }
    std::memcpy(ftr_mdl->mfcc_dat, out.data(), (size_t)frames * 12 * sizeof(int16_t));
    ftr_mdl->frm_num = (uint16_t)frames;
    return dis;
}

// DTW.C:195-205: one frame pair = get_mdl of two 1-frame records, first merged frame
void get_mean(int16_t *frm_ftr1, int16_t *frm_ftr2, int16_t *mean)
{
    sr_engine *h = pair_engine("get_mean");
    const uint32_t one = 1;
    uint32_t frames = 0, dis = 0;
    int16_t out[12];
    if (sr_get_mdl_batch(h, frm_ftr1, &one, 1, frm_ftr2, &one, 1, 1, out, 1, &frames, &dis) != SR_OK) die("get_mean");
    std::memcpy(mean, out, sizeof(out));
}

// main.c:249-296
uint8_t *spch_recg(uint16_t *v_dat, uint32_t *mtch_dis)
{
    sr_engine *h = engine("spch_recg");
    if (sr_num_templates(h) == 0) {
        std::fprintf(stderr, "spch_recg: no template store (call sr_compat_set_templates first)\n");
        std::abort();
    }
    sr_result r;
    if (sr_recognize_batch(h, v_dat, kVcBufLen, kVcBufLen, 1, &r, nullptr, nullptr, nullptr) != SR_OK) die("spch_recg");
    *mtch_dis = r.min_dis;
    if (r.status != SR_ST_OK) return nullptr;  // main.c:261-274 (VAD fail / MFCC fail)
    const uint32_t comm = r.best_tpl / g_ftr_per_comm;  // main.c:292
    if (comm >= g_n_labels) return g_empty_label;       // the firmware would index past commstr[]
    return g_labels.data() + (size_t)comm * g_label_stride;
}

}  // extern "C"
