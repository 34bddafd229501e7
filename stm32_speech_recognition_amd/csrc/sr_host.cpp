// Host-buffer entry points of the C ABI (include/sr_engine.h): captures / records staged through HBM (pinned staging area and
// one synchronisation for small calls, chunked upload overlapped with the kernels for large ones), the diagnostics, and the
// scalar helpers behind the reference-compatible symbols of sr_compat.cpp.
#include "sr_engine_internal.h"

using namespace sr;
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
    // the internal stream is non-blocking: it is ordered explicitly behind the last asynchronous call that used the scratch
    // buffers on a caller's stream (before round 5 the blocking null-stream copies of these paths did that implicitly for the
    // legacy default stream only)
    explicit PinCall(sr_engine *e) : h(e), base((uint8_t *)e->pin_buf)
    {
        if (order_after_scratch_users(e, e->st_comp) != SR_OK) ok = false;
    }
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
        if (!launch_dtw_auto(h, a, 0, pc.stream(), pc.stream())) launch_argmin(a, pc.stream());
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
    if (!launch_dtw_auto(h, a, 0, nullptr, nullptr)) launch_argmin(a, nullptr);
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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

int sr_dtw_dp_batch(sr_engine *h, const int16_t *in_mfcc, const uint32_t *in_frames, uint32_t B, uint32_t *scores)
{
    if (!h || !in_mfcc || !in_frames || !scores) return fail(SR_ERR_BAD_ARG, "null argument");
    if (!h->K) return fail(SR_ERR_NO_TEMPLATES, "no templates set");
    if (B == 0) return SR_OK;
    for (uint32_t b = 0; b < B; b++)
        if (in_frames[b] > h->cfg.max_frames) return fail(SR_ERR_BAD_ARG, "in_frames exceeds max_frames");
    ENTER_DEVICE(h);
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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

int sr_delta_mfcc_batch(sr_engine *h, const int16_t *mfcc, const uint32_t *frames, uint32_t B, int16_t *delta)
{
    if (!h || !mfcc || !frames || !delta) return fail(SR_ERR_BAD_ARG, "null argument");
    if (B == 0) return SR_OK;
    ENTER_DEVICE(h);
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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

// diagnostics: the cheap magnitude form against the exact one, n in [0, n_max]: out[0] = mismatches, out[1] = first mismatching n
int sr_mag_fast_sweep(sr_engine *h, uint32_t n_max, uint64_t out[2])
{
    if (!h || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    ENTER_DEVICE(h);
    int rc;
    if ((rc = h->s_u32a.reserve(4))) return rc;
    uint32_t init[4] = {0, 0, 0xFFFFFFFFu, 0};
    HIP_TRY(hipMemcpy(h->s_u32a.p, init, sizeof init, hipMemcpyHostToDevice));
    launch_mag_fast_sweep(n_max, (unsigned long long *)h->s_u32a.p, h->s_u32a.p + 2, nullptr);
    HIP_TRY(hipGetLastError());
    uint32_t back[4];
    HIP_TRY(hipMemcpy(back, h->s_u32a.p, sizeof back, hipMemcpyDeviceToHost));
    out[0] = (uint64_t)back[0] | ((uint64_t)back[1] << 32);
    out[1] = back[2];
    return SR_OK;
}

// diagnostics: overwrite the local data share of every compute unit with a seeded pattern (on `stream`, asynchronous); the
// suite launches it between calls so that no kernel can pass by reading what its own previous workgroups left in LDS
int sr_lds_poison(sr_engine *h, uint32_t seed, void *stream, uint32_t *bytes_per_cu)
{
    if (!h) return fail(SR_ERR_BAD_ARG, "null argument");
    ENTER_DEVICE(h);
    const int n = launch_lds_poison(seed, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    if (n < 0) return fail(SR_ERR_HIP, "sr_lds_poison: device attribute query failed");
    if (bytes_per_cu) *bytes_per_cu = (uint32_t)n;
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
    int rc;
    if ((rc = h->s_u32a.reserve((size_t)n * kNfft))) return rc;
    if ((rc = h->s_u32b.reserve((size_t)n * kNfft))) return rc;
    HIP_TRY(hipMemcpy(h->s_u32a.p, in, (size_t)n * kNfft * 4, hipMemcpyHostToDevice));
    launch_fft_q15(h->s_u32a.p, h->s_u32b.p, n, h->dev, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32b.p, (size_t)n * kNfft * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

namespace sr {
int engine_fft_mag(sr_engine *h, const int16_t *frame, uint32_t len, uint32_t *mag, uint32_t *raw_hi)
{
    ENTER_DEVICE(h);
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
    int rc;
    if ((rc = h->s_u32a.reserve(2))) return rc;
    const uint16_t xy[2] = {x, y};
    HIP_TRY(hipMemcpy(h->s_u32a.p, xy, 4, hipMemcpyHostToDevice));
    launch_dtw_limit((const uint16_t *)h->s_u32a.p, (uint8_t *)(h->s_u32a.p + 1), 1, X1, X2, in_n, mdl_n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32a.p + 1, 1, hipMemcpyDeviceToHost));
    return SR_OK;
}

}  // namespace sr

// Batched forms of the two small scalar symbols (the kernels get_dis() / dtw_limit() launch with n = 1), so that tests can
// compare them with the reference object over whole grids in one launch.
// get_dis (DTW.C:45-62) on n pairs of 12-coefficient rows
int sr_get_dis_batch(sr_engine *h, const int16_t *a, const int16_t *b, uint32_t n, uint32_t *out)
{
    if (!h || !a || !b || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    if (n == 0) return SR_OK;
    if (h->nc != (uint32_t)kCoef) return fail(SR_ERR_BAD_CONFIG, "sr_get_dis_batch: 12-coefficient front ends only");
    ENTER_DEVICE(h);
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;
    int rc;
    if ((rc = h->s_mfcc.reserve((size_t)2 * n * kCoef))) return rc;
    if ((rc = h->s_u32a.reserve(n))) return rc;
    HIP_TRY(hipMemcpy(h->s_mfcc.p, a, (size_t)n * kCoef * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->s_mfcc.p + (size_t)n * kCoef, b, (size_t)n * kCoef * 2, hipMemcpyHostToDevice));
    launch_get_dis(h->s_mfcc.p, h->s_mfcc.p + (size_t)n * kCoef, h->s_u32a.p, n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, h->s_u32a.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

// dtw_limit (DTW.C:76-109) on n points (x, y) for the statics a dtw() call of in_frames against mdl_frames leaves behind
// (DTW.C:129-130, 141-142: X1 = (2 mdl - in) / 3, X2 = (4 in - 2 mdl) / 3 as u16); out[i] = 1: outside
int sr_dtw_limit_batch(sr_engine *h, const uint16_t *xy, uint32_t n, uint32_t in_frames, uint32_t mdl_frames, uint8_t *out)
{
    if (!h || !xy || !out) return fail(SR_ERR_BAD_ARG, "null argument");
    if (n == 0) return SR_OK;
    if (in_frames > 65535u || mdl_frames > 65535u) return fail(SR_ERR_BAD_ARG, "sr_dtw_limit_batch: frame counts are u16 (DTW.C:65-68)");
    ENTER_DEVICE(h);
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;
    int rc;
    const size_t words = ((size_t)n * 4 + 3) / 4 + ((size_t)n + 3) / 4;
    if ((rc = h->s_u32a.reserve(words))) return rc;
    uint8_t *d_out = (uint8_t *)(h->s_u32a.p + n);
    HIP_TRY(hipMemcpy(h->s_u32a.p, xy, (size_t)n * 4, hipMemcpyHostToDevice));
    const int X1 = (int)(uint16_t)((2 * (int)mdl_frames - (int)in_frames) / 3), X2 = (int)(uint16_t)((4 * (int)in_frames - 2 * (int)mdl_frames) / 3);
    launch_dtw_limit((const uint16_t *)h->s_u32a.p, d_out, n, X1, X2, (int)in_frames, (int)mdl_frames, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost));
    return SR_OK;
}

namespace sr {

int engine_vad_with_atap(sr_engine *h, const uint16_t *pcm, uint32_t buf_len, const sr_atap *atap, sr_vad_rec *rec)
{
    ENTER_DEVICE(h);
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
    if (int rc_ord_ = order_after_scratch_users(h, nullptr)) return rc_ord_;  // the null stream reuses the scratch buffers
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
