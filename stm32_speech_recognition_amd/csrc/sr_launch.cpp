// Device-resident entry points of the C ABI (include/sr_engine.h): argument checks, the launch-argument blocks of the kernels,
// and the sequencing of VAD -> frame kernel -> DTW -> slot scan on the caller's stream and the internal chunk streams.
#include "sr_engine_internal.h"
#include "sr_dtw_quad.h"

using namespace sr;
// ---- device-resident pipeline ---------------------------------------------------------------------
// the frame kernel indexes (utterance, tile) work items with 32 bits; the bound uses the SMALLEST tile mfcc_args can pick
// (the 16- / 4-frame forms of underfilled launches have up to 16 x more items than the 64-frame batch form)
int check_batch(const sr_engine *h, uint32_t B)
{
    const uint32_t tile = std::min(h->mfcc_tile, std::min(h->mfcc_tile_mid, h->mfcc_tile_small));
    if ((uint64_t)B * ((h->cfg.max_frames + tile - 1) / tile) > 0xFFFFFFFFull) return fail(SR_ERR_BAD_ARG, "batch too large");
    return SR_OK;
}

// an asynchronous call on `s` has just used the engine's scratch buffers: remember where it ends
int mark_scratch_user(sr_engine *h, hipStream_t s)
{
    if (!h->ev_scratch) HIP_TRY(hipEventCreateWithFlags(&h->ev_scratch, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->ev_scratch, s));
    h->scratch_pending = true;
    return SR_OK;
}
// `s` is about to reuse the scratch buffers: it runs after the last asynchronous user (a no-op when there was none, or when
// that user ran on `s` itself)
int order_after_scratch_users(sr_engine *h, hipStream_t s)
{
    if (!h->scratch_pending) return SR_OK;
    if (hipEventQuery(h->ev_scratch) == hipSuccess) {  // that call has finished: nothing to wait for any more
        h->scratch_pending = false;
        return SR_OK;
    }
    (void)hipGetLastError();  // hipErrorNotReady
    HIP_TRY(hipStreamWaitEvent(s, h->ev_scratch, 0));
    return SR_OK;
}
int check_pcm(const sr_engine *h, const uint16_t *pcm, uint64_t stride, uint32_t buf_len)
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

VadArgs vad_args(const sr_engine *h, const uint16_t *pcm, uint64_t stride, uint32_t buf_len, uint32_t noise_len, uint32_t B,
                        sr_vad_rec *vad, const sr_atap *atap_in, uint64_t *dbg)
{
    // fewer captures than CUs: a workgroup of four waves per capture instead of one wave (k_vad_wide; same records)
    const uint32_t wide = (h->small_launch == 2 || (h->small_launch != 1 && B < kVadWideBelow)) ? 1u : 0u;  // (mode 3 only forces the DTW form)
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

MfccArgs mfcc_args(const sr_engine *h, const uint16_t *d_pcm, uint64_t pcm_stride, uint32_t B,
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
    a.mag_cheap_max = h->mag_cheap_max;
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

DtwArgs dtw_args(const sr_engine *h, const int16_t *d_mfcc, const sr_vad_rec *d_vad, const uint32_t *d_in_frames,
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
    a.tpl_neg2_ok = h->tpl_staged_ok ? 1u : 0u;
    a.cells_literal = dev_hook(kHookCellsLiteral) != 0 ? 1u : 0u;
    a.dev_cus = h->n_cu;
    a.dev_lds_cu = h->lds_per_cu;
    a.dev_lds_wg = h->lds_per_wg;
    return a;
}

// dtw for every pair of the launch: the batch kernels (k_dtw_lds / k_dtw_gen / k_dtw), or -- a few hundred pairs, i.e. a GPU
// that would otherwise idle behind a handful of serial walks -- one workgroup per pair (k_dtw_cells).  Same scores.
// Measured (profiles/r04_small_launch_sweep.json, profiles/experiments/RESULTS.md): 110-frame captures against 80 slots of up
// to 119 frames: 80 / 320 / 640 / 1 280 / 2 560 / 5 120 pairs take 25 / 33 / 44 / 65 / 115 / 212 us with one workgroup per pair
// against 126 us for the batch kernel at any of these sizes; 256-frame captures against 100 templates of 192-320 frames (the
// benchmark's shapes): 100 / 400 pairs 60 / 107 us against 215.  A pair costs in proportion to its band (~ frames^2), the
// batch kernel's latency grows with the frames, so the automatic mode stops at 320 000 / max_frames pairs (2 689 / 1 000).
// Round 5: where the four-lanes-per-pair form (k_dtw_quad, below) can take over, one workgroup per pair only pays up to
// 120 000 / max_frames pairs (1 008 / 375): 640 / 1 280 / 2 560 pairs take 44 / 65 / 114 us against a flat 49 us there.
static uint64_t small_launch_pairs(const DtwArgs &a) { return (dtw_quad_fits(a) ? 120000u : 320000u) / (a.max_frames > 64 ? a.max_frames : 64u); }
// returns true when the slot scan (argmin) has been done as well: k_dtw_cells with result records asked for and the utterances
// b0 .. b0 + B of the call within the counters
// Mid-sized launches: four lanes per pair (k_dtw_quad.hip).  Its workgroups hold PU x PK pairs with both sequences in LDS; a
// "round" is what the chip holds at once (workgroups per CU by LDS, at most 8, x 256 CUs).  The batch kernel's time is flat up
// to ~400 000 pairs (126 us at the firmware's shapes), a round of the quad kernel takes a third of that, so the automatic mode
// hands it launches of up to two rounds (profiles/r05_small_launch_sweep.json).
static uint64_t quad_launch_pairs(const sr_engine *h, const DtwArgs &a)
{
    uint32_t pu = 0, pk = 0;
    size_t lds = 0;
    if (!dtw_quad_pick(a, &pu, &pk, &lds)) return 0;
    const uint64_t granules = (uint64_t)h->lds_per_cu / 1280;  // gfx950 hands out LDS in granules of 1 280 bytes (128 per CU on MI355X)
    const uint64_t per_cu = std::min<uint64_t>(8, granules / ((lds + 1279) / 1280));
    return 2 * (uint64_t)h->n_cu * per_cu * pu * pk;
}
// `owner` = the caller-level stream of the call (the counters belong to one caller stream, see sr_engine::cells_owner)
bool launch_dtw_auto(sr_engine *h, DtwArgs &a, uint32_t b0, hipStream_t s, hipStream_t owner)
{
    const uint64_t pairs = (uint64_t)a.B * a.K;
    if (h->small_launch != 1 && h->small_launch != 3 && dtw_cells_fits(a) && (h->small_launch == 2 || pairs <= small_launch_pairs(a))) {
        bool counters = a.results && (uint64_t)b0 + a.B <= kPairCounters;
        if (counters) {
            if (!h->ev_cells && hipEventCreateWithFlags(&h->ev_cells, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                h->ev_cells = nullptr;
            }
            // another stream may take the counters over once the last launch that used them is done (nothing counts in them then)
            if (h->cells_owner_set && h->cells_owner != owner && h->ev_cells && hipEventQuery(h->ev_cells) == hipSuccess)
                h->cells_owner_set = false;
            (void)hipGetLastError();  // hipErrorNotReady of the query is not an error of this call
            if (!h->cells_owner_set) {
                h->cells_owner = owner;
                h->cells_owner_set = true;
            }
            counters = h->cells_owner == owner && h->ev_cells != nullptr;
        }
        a.pair_count = counters ? h->s_pcnt.p + b0 : nullptr;
        launch_dtw_cells(a, s);
        if (a.pair_count) (void)hipEventRecord(h->ev_cells, s);
        return a.pair_count != nullptr;
    }
    if ((h->small_launch == 3 && dtw_quad_fits(a)) || (h->small_launch == 0 && pairs <= quad_launch_pairs(h, a))) {
        launch_dtw_quad(a, s);
        return false;
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
    if (!launch_dtw_auto(h, a, 0, (hipStream_t)stream, (hipStream_t)stream) && d_results) launch_argmin(a, (hipStream_t)stream);
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
    const bool own_scratch = !d_vad || !d_mfcc || !d_scores;  // (the host-buffer entry points pass the scratch buffers explicitly)
    if (own_scratch && (rc = order_after_scratch_users(h, s))) return rc;
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
        const bool scanned = launch_dtw_auto(h, da, b0, sc, s);
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
    if (own_scratch && (rc = mark_scratch_user(h, s))) return rc;
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
    if ((rc = order_after_scratch_users(h, s))) return rc;  // s_vad2 / s_mfcc are always the engine's
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
        if (!launch_dtw_auto(h, da, 0, s, s)) launch_argmin(da, s);
    }
    HIP_TRY(hipGetLastError());
    return mark_scratch_user(h, s);
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
