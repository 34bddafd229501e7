// Shared by the host translation units of the library (sr_engine.cpp: lifecycle, template store, settings;
// sr_launch.cpp: the device-resident entry points and kernel sequencing; sr_host.cpp: host-buffer entry points, staging,
// diagnostics): the engine handle, device-memory / device-selection helpers and the launch-argument builders.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "sr_device.h"
#include "sr_dtw_cells.h"
#include "sr_tables.h"

namespace sr {

int set_error(int code, const std::string &msg);  // records the message sr_last_error() returns (thread-local), returns code
static inline int fail(int code, const std::string &msg) { return set_error(code, msg); }

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

using namespace sr;  // internal header of three host units of one library: the handle below is a global C type built from sr:: parts

// launch sizes below which VAD / the frame kernel take their small-launch forms (captures; work items of 64 frames)
static constexpr uint32_t kVadWideBelow = 1024, kMfccFill = 1024;  // measured crossover ~2 000 captures; work items that fill 256 CUs x 4 (RESULTS.md)
// utterances of one call whose slot scan k_dtw_cells can do itself (one counter each); beyond that k_argmin runs as usual
static constexpr uint32_t kPairCounters = 65536;

struct sr_engine {
    sr_config cfg;
    int device = 0;
    uint32_t noise_len = 0, atap_frm = 0;
    uint32_t n_cu = 256, lds_per_cu = 160 * 1024, lds_per_wg = 160 * 1024;  // of the engine's device (sr_create): launch-shape decisions use these, not MI355X's figures
    uint32_t mag_cheap_max = 0;  // kMagCheapMax once the device sweep at sr_create has confirmed the cheap magnitude form on this chip, else 0
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
    int small_launch = 0;          // sr_set_small_launch: 0 = automatic (k_dtw_cells for a few hundred pairs, k_dtw_quad up to two rounds of the chip), 1 = never, 2 / 3 = k_dtw_cells / k_dtw_quad whenever it fits
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
    // The counters are hidden per-engine state shared by every launch: two small calls in flight on DIFFERENT caller streams
    // would both count in them.  They therefore belong to one caller stream at a time (internal chunk streams are forked from /
    // joined to it, so its order covers them); a call on any other stream leaves the slot scan to k_argmin -- unless the last
    // launch that used the counters has COMPLETED (ev_cells, recorded behind every such launch): then nothing is in flight in
    // them and the calling stream becomes the owner (round 6; before, the first stream kept them for the engine's lifetime,
    // e.g. the internal stream of a first host-buffer call, or a stream the caller had destroyed since).
    hipStream_t cells_owner = nullptr;
    bool cells_owner_set = false;
    hipEvent_t ev_cells = nullptr;
    // An asynchronous *_dev call that was handed no buffer for an intermediate uses the engine's scratch (s_vad, s_mfcc,
    // s_scores, s_vad2) on the CALLER's stream.  The event marks the end of the last such call; the host-buffer entry points,
    // which reuse the same scratch on internal or the null stream, order their stream behind it first (order_after_scratch_users).
    hipEvent_t ev_scratch = nullptr;
    bool scratch_pending = false;
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
    uint32_t pipe_max_chunks = 12;         // chunks per call at most (sr_set_pipeline); one per stream for large stores, see upload_templates
    bool pipe_user_set = false;            // sr_set_pipeline was called: the engine no longer adapts the chunk count to the store
    // profiling (sr_set_profiling / sr_get_stage_ms): events recorded since profiling was switched on
    bool profiling = false;
    std::vector<hipEvent_t> ev;  // 5 per kernel group (chunk): before VAD, MFCC, DTW, argmin, after argmin
    size_t ev_used = 0;          // groups recorded
    std::vector<hipEvent_t> ev_call;  // 2 per call on the caller's stream: before the fork, after the join
    size_t calls_used = 0;
};

// ---- helpers shared by the launch and the host-buffer units (sr_launch.cpp) --------------------------------------
int check_batch(const sr_engine *h, uint32_t B);
int check_pcm(const sr_engine *h, const uint16_t *pcm, uint64_t stride, uint32_t buf_len);
int mark_scratch_user(sr_engine *h, hipStream_t s);
int order_after_scratch_users(sr_engine *h, hipStream_t s);
VadArgs vad_args(const sr_engine *h, const uint16_t *pcm, uint64_t stride, uint32_t buf_len, uint32_t noise_len, uint32_t B,
                     sr_vad_rec *vad, const sr_atap *atap_in = nullptr, uint64_t *dbg = nullptr);
MfccArgs mfcc_args(const sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t B, const sr_vad_rec *d_vad, int16_t *d_mfcc);
DtwArgs dtw_args(const sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad, const uint32_t *d_in_frames, uint32_t B,
                     uint32_t *d_scores, sr_result *d_results);
bool launch_dtw_auto(sr_engine *h, DtwArgs &a, uint32_t b0, hipStream_t s, hipStream_t owner);
