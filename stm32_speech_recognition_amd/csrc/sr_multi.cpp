// Multi-GPU surface of the C ABI (include/sr_engine.h, "multi-GPU" section): ONE process drives several MI355X.
// Utterances are sharded over the devices, templates are replicated, and the path's single exchange step -- the
// all-gather of the per-template score matrix u32 [B][K] that replaces the firmware's slot scan result
// (main.c:279-291) on every device -- is one RCCL all-gather over xGMI, issued on the same streams as the kernels.
// RCCL is bound at run time (dlopen of librccl.so.1): libsr_engine.so itself has no link-time dependency on it, and
// a process that already carries a RCCL (PyTorch) shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sr_device.h"

namespace sr {
int set_error(int code, const std::string &msg);  // sr_engine.cpp: thread-local text behind sr_last_error()
}
using sr::set_error;

namespace {

// The handful of RCCL declarations this file needs, stated locally (values as in rccl.h of ROCm 7: ncclResult_t 0 =
// ncclSuccess, ncclDataType_t 3 = ncclUint32): the library is bound at run time, so a ROCm install without the rccl
// development headers still builds libsr_engine.so.
typedef int ncclResult_t;
typedef struct ncclComm *ncclComm_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint32 = 3;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // SR_RCCL_LIBRARY: explicit path of the collective library (a site-specific RCCL build; the in-process test
        // double tests/fake_rccl/librccl.so.1).  Bound RTLD_LOCAL so that its symbols never shadow a RCCL the process
        // already carries.
        const char *override_path = getenv("SR_RCCL_LIBRARY");
        if (override_path && *override_path) {
            r.lib = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.lib) break;
            }
        }
        if (!r.lib) {
            const char *why = dlerror();
            r.err = std::string("cannot load RCCL (") + (override_path && *override_path ? override_path : "librccl.so.1") +
                    "): " + (why ? why : "unknown dlopen failure");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

#define HIP_M(expr)                                                                                       \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return set_error(SR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCL_M(expr)                                                                                      \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) return set_error(SR_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

struct DevGuard {
    int prev = -1;
    DevGuard() { (void)hipGetDevice(&prev); }
    ~DevGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct sr_multi {
    std::vector<int> dev;
    std::vector<sr_engine *> eng;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st;
    // scratch of the host-buffer call, one set per device
    std::vector<uint16_t *> d_pcm;
    std::vector<sr_result *> d_res;
    std::vector<uint32_t *> d_all;
    std::vector<size_t> cap_pcm, cap_res, cap_all;
    // false after a template upload that reached only some of the devices: the engines then disagree about the store,
    // and every recognise call is refused until a later upload succeeds on all of them
    bool store_consistent = true;
};

// One upload per device.  The single-engine upload is failure-atomic (a failed call keeps the old store), so a failure
// on the FIRST device leaves every device on the old store and the handle stays usable; a failure on a later device
// leaves the devices before it on the new store: the handle is marked inconsistent and recognise calls are refused
// (mixed stores would be gathered as one mis-strided score matrix) until an upload succeeds everywhere.
template <class F>
static int replicate(sr_multi *m, F upload)
{
    for (size_t i = 0; i < m->eng.size(); i++)
        if (int rc = upload(m->eng[i])) {
            if (i > 0) m->store_consistent = false;
            return rc;
        }
    m->store_consistent = true;
    return SR_OK;
}

// every recognise entry: the store must be the same everywhere (K is the stride of the gathered matrix)
static int check_store(const sr_multi *m, uint32_t *K)
{
    if (!m->store_consistent)
        return set_error(SR_ERR_NO_TEMPLATES, "template stores differ between devices (an earlier sr_multi_set_templates* failed part-way): upload again");
    *K = sr_num_templates(m->eng[0]);
    if (!*K) return set_error(SR_ERR_NO_TEMPLATES, "no templates set");
    for (sr_engine *e : m->eng)
        if (sr_num_templates(e) != *K) return set_error(SR_ERR_NO_TEMPLATES, "template count differs between devices");
    return SR_OK;
}

extern "C" {

int sr_multi_create(const sr_config *cfg, const int *devices, uint32_t n_dev, sr_multi **out)
{
    if (!cfg || !devices || !out || n_dev == 0 || n_dev > 64) return set_error(SR_ERR_BAD_ARG, "null argument / device count 1..64");
    *out = nullptr;
    // development hook "multi_allow_dup" (tests only): several "ranks" on one device, so that the N > 1 bookkeeping can be
    // executed on a 1-GPU box against the in-process collective double.  Honoured only together with an explicitly named
    // collective library (SR_RCCL_LIBRARY): a real RCCL errors or hangs on duplicate devices.
    const bool allow_dup = sr::dev_hook(sr::kHookMultiAllowDup) != 0 && getenv("SR_RCCL_LIBRARY") != nullptr;
    if (!allow_dup)
        for (uint32_t i = 0; i < n_dev; i++)
            for (uint32_t j = 0; j < i; j++)
                if (devices[i] == devices[j]) return set_error(SR_ERR_BAD_ARG, "duplicate device ordinal");
    Rccl *R = rccl();
    if (!R->err.empty()) return set_error(SR_ERR_NO_DEVICE, R->err);
    DevGuard guard;
    sr_multi *m = new sr_multi();
    m->dev.assign(devices, devices + n_dev);
    m->eng.assign(n_dev, nullptr);
    m->st.assign(n_dev, nullptr);
    m->d_pcm.assign(n_dev, nullptr);
    m->d_res.assign(n_dev, nullptr);
    m->d_all.assign(n_dev, nullptr);
    m->cap_pcm.assign(n_dev, 0);
    m->cap_res.assign(n_dev, 0);
    m->cap_all.assign(n_dev, 0);
    int rc = SR_OK;
    for (uint32_t i = 0; i < n_dev && rc == SR_OK; i++) {
        sr_config c = *cfg;
        c.device = devices[i];
        rc = sr_create(&c, &m->eng[i]);
        if (rc == SR_OK && (hipSetDevice(devices[i]) != hipSuccess ||
                            hipStreamCreateWithFlags(&m->st[i], hipStreamNonBlocking) != hipSuccess))
            rc = set_error(SR_ERR_HIP, "stream creation failed");
    }
    if (rc == SR_OK) {
        m->comm.assign(n_dev, nullptr);
        ncclResult_t r = R->CommInitAll(m->comm.data(), (int)n_dev, devices);  // one communicator per device, this process owns all ranks
        if (r != ncclSuccess) {
            m->comm.clear();
            rc = set_error(SR_ERR_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(r));
        }
    }
    if (rc != SR_OK) {
        const std::string keep = sr_last_error();
        sr_multi_destroy(m);
        return set_error(rc, keep);
    }
    *out = m;
    return SR_OK;
}

void sr_multi_destroy(sr_multi *m)
{
    if (!m) return;
    DevGuard guard;
    for (size_t i = 0; i < m->dev.size(); i++) {
        (void)hipSetDevice(m->dev[i]);
        (void)hipDeviceSynchronize();
        if (i < m->comm.size() && m->comm[i]) (void)rccl()->CommDestroy(m->comm[i]);
        if (m->st[i]) (void)hipStreamDestroy(m->st[i]);
        if (m->d_pcm[i]) (void)hipFree(m->d_pcm[i]);
        if (m->d_res[i]) (void)hipFree(m->d_res[i]);
        if (m->d_all[i]) (void)hipFree(m->d_all[i]);
        if (m->eng[i]) sr_destroy(m->eng[i]);
    }
    delete m;
}

uint32_t sr_multi_num_devices(const sr_multi *m) { return m ? (uint32_t)m->dev.size() : 0; }
sr_engine *sr_multi_engine(sr_multi *m, uint32_t i) { return (m && i < m->eng.size()) ? m->eng[i] : nullptr; }

int sr_multi_set_templates(sr_multi *m, const void *store, uint32_t n_slots, uint32_t stride_bytes)
{
    if (!m) return set_error(SR_ERR_BAD_ARG, "null handle");
    return replicate(m, [&](sr_engine *e) { return sr_set_templates(e, store, n_slots, stride_bytes); });  // 0.8 MB at K = 100
}

int sr_multi_set_templates_dense(sr_multi *m, const int16_t *mfcc, const uint32_t *frames, const uint8_t *valid,
                                 uint32_t n_templates, uint32_t tpl_stride)
{
    if (!m) return set_error(SR_ERR_BAD_ARG, "null handle");
    return replicate(m, [&](sr_engine *e) { return sr_set_templates_dense(e, mfcc, frames, valid, n_templates, tpl_stride); });
}

// Device-resident shards.  Device i recognises its B_per_dev utterances (d_pcm[i]) and writes its block of the score
// matrix straight into d_scores_all[i] + i*B_per_dev*K; the in-place all-gather then completes d_scores_all[i]
// ([n_dev*B_per_dev][K], global utterance order) on EVERY device.  Asynchronous on streams[i] (NULL: the handle's own
// streams, and the call returns after they have drained).
int sr_multi_recognize_dev(sr_multi *m, const uint16_t *const *d_pcm, uint64_t pcm_stride, uint32_t buf_len,
                           uint32_t B_per_dev, sr_result *const *d_results, uint32_t *const *d_scores_all,
                           void *const *streams)
{
    if (!m || !d_pcm || !d_results || !d_scores_all) return set_error(SR_ERR_BAD_ARG, "null argument");
    const uint32_t n = (uint32_t)m->dev.size();
    uint32_t K = 0;
    if (int rc = check_store(m, &K)) return rc;
    if (B_per_dev == 0) return SR_OK;
    Rccl *R = rccl();
    DevGuard guard;
    const size_t block = (size_t)B_per_dev * K;
    for (uint32_t i = 0; i < n; i++) {
        if (!d_pcm[i] || !d_results[i] || !d_scores_all[i])
            return set_error(SR_ERR_BAD_ARG, "null per-device pointer (device index " + std::to_string(i) + ": " +
                                                 (!d_pcm[i] ? "d_pcm" : !d_results[i] ? "d_results" : "d_scores_all") + ")");
        hipStream_t s = streams ? (hipStream_t)streams[i] : m->st[i];
        if (int rc = sr_recognize_batch_dev(m->eng[i], d_pcm[i], pcm_stride, buf_len, B_per_dev, d_results[i],
                                            d_scores_all[i] + (size_t)i * block, nullptr, nullptr, s))
            return rc;
    }
    NCCL_M(R->GroupStart());
    for (uint32_t i = 0; i < n; i++) {
        hipStream_t s = streams ? (hipStream_t)streams[i] : m->st[i];
        ncclResult_t r = R->AllGather(d_scores_all[i] + (size_t)i * block, d_scores_all[i], block, ncclUint32, m->comm[i], s);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            return set_error(SR_ERR_HIP, std::string("ncclAllGather: ") + R->GetErrorString(r));
        }
    }
    NCCL_M(R->GroupEnd());
    if (!streams)
        for (uint32_t i = 0; i < n; i++) {
            HIP_M(hipSetDevice(m->dev[i]));
            HIP_M(hipStreamSynchronize(m->st[i]));
        }
    return SR_OK;
}

// Host buffers: B utterances are cut into n_dev equal shards (the last one padded with copies of its final utterance so
// that the all-gather blocks have one size), uploaded, recognised, gathered; results[B] come from the owning devices,
// scores[B*K] (optional) from device 0's copy of the gathered matrix.
int sr_multi_recognize(sr_multi *m, const uint16_t *pcm, uint64_t pcm_stride, uint32_t buf_len, uint32_t B,
                       sr_result *results, uint32_t *scores)
{
    if (!m || !pcm || !results) return set_error(SR_ERR_BAD_ARG, "null argument");
    if (B == 0) return SR_OK;
    if (buf_len > pcm_stride) return set_error(SR_ERR_BAD_ARG, "buf_len exceeds pcm_stride");
    const uint32_t n = (uint32_t)m->dev.size();
    uint32_t K = 0;
    if (int rc = check_store(m, &K)) return rc;
    const uint32_t per = (B + n - 1) / n;
    const uint64_t ds = ((uint64_t)buf_len + 7) & ~7ull;
    DevGuard guard;
    auto grow = [](auto **p, size_t *cap, size_t bytes) -> hipError_t {
        if (bytes <= *cap) return hipSuccess;
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *cap = 0;
        hipError_t e = hipMalloc((void **)p, bytes);
        if (e == hipSuccess) *cap = bytes;
        return e;
    };
    for (uint32_t i = 0; i < n; i++) {
        HIP_M(hipSetDevice(m->dev[i]));
        HIP_M(grow(&m->d_pcm[i], &m->cap_pcm[i], (size_t)per * ds * 2));
        HIP_M(grow(&m->d_res[i], &m->cap_res[i], (size_t)per * sizeof(sr_result)));
        HIP_M(grow(&m->d_all[i], &m->cap_all[i], (size_t)n * per * K * 4));
        const uint32_t b0 = i * per, have = b0 < B ? std::min(per, B - b0) : 0;
        if (have)
            HIP_M(hipMemcpy2DAsync(m->d_pcm[i], ds * 2, pcm + (size_t)b0 * pcm_stride, pcm_stride * 2, (size_t)buf_len * 2, have,
                                   hipMemcpyHostToDevice, m->st[i]));
        for (uint32_t b = have; b < per; b++) {  // padding rows: any valid capture (their results are never returned)
            const uint16_t *src = pcm + (size_t)(have ? b0 + have - 1 : B - 1) * pcm_stride;
            HIP_M(hipMemcpyAsync(m->d_pcm[i] + (size_t)b * ds, src, (size_t)buf_len * 2, hipMemcpyHostToDevice, m->st[i]));
        }
    }
    std::vector<void *> st(n);
    for (uint32_t i = 0; i < n; i++) st[i] = m->st[i];
    if (int rc = sr_multi_recognize_dev(m, m->d_pcm.data(), ds, buf_len, per, m->d_res.data(), m->d_all.data(), st.data()))
        return rc;
    for (uint32_t i = 0; i < n; i++) {
        HIP_M(hipSetDevice(m->dev[i]));
        const uint32_t b0 = i * per, have = b0 < B ? std::min(per, B - b0) : 0;
        if (have)
            HIP_M(hipMemcpyAsync(results + b0, m->d_res[i], (size_t)have * sizeof(sr_result), hipMemcpyDeviceToHost, m->st[i]));
    }
    if (scores) {
        HIP_M(hipSetDevice(m->dev[0]));
        HIP_M(hipMemcpyAsync(scores, m->d_all[0], (size_t)B * K * 4, hipMemcpyDeviceToHost, m->st[0]));
    }
    for (uint32_t i = 0; i < n; i++) {
        HIP_M(hipSetDevice(m->dev[i]));
        HIP_M(hipStreamSynchronize(m->st[i]));
    }
    return SR_OK;
}

// The exchange step on its own, for callers that run one process per GPU and own a communicator (ncclComm_t from
// ncclCommInitRank): all-gather of this rank's u32 [count] score block into d_all ([n_ranks*count], rank order) on
// `stream`.  d_scores may be d_all + rank*count (in place).
int sr_allgather_scores(void *nccl_comm, const uint32_t *d_scores, uint32_t *d_all, uint64_t count, void *stream)
{
    if (!nccl_comm || !d_scores || !d_all) return set_error(SR_ERR_BAD_ARG, "null argument");
    Rccl *R = rccl();
    if (!R->err.empty()) return set_error(SR_ERR_NO_DEVICE, R->err);
    NCCL_M(R->AllGather(d_scores, d_all, (size_t)count, ncclUint32, (ncclComm_t)nccl_comm, (hipStream_t)stream));
    return SR_OK;
}

}  // extern "C"
