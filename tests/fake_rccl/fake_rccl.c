/*
 * fake_rccl.c -- TEST DOUBLE, never part of the product: an in-process stand-in for the five RCCL entry points
 * csrc/sr_multi.cpp binds at run time (ncclCommInitAll, ncclCommDestroy, ncclGroupStart/End, ncclAllGather,
 * ncclGetErrorString).  Built as tests/fake_rccl/librccl.so.1 and selected with SR_RCCL_LIBRARY=<that path>.
 *
 * Purpose: execute the N > 1 bookkeeping of sr_multi_* (block offsets of the in-place all-gather, padded and empty
 * last shards, read-back from the owning devices) on a box with ONE GPU.  Together with SR_MULTI_TEST_ALLOW_DUP=1 the
 * "ranks" of a communicator may all sit on device 0; the all-gather is then n*(n-1) device-to-device copies between the
 * ranks' buffers, ordered against the ranks' streams with events exactly as a real collective orders them:
 *   - every rank's copies start after ALL ranks' earlier work on their streams (the send blocks are complete),
 *   - every rank's later work starts after all ranks have read its send block.
 * Semantics checked like RCCL does: one call per rank of the communicator inside one group, equal counts, a data type
 * of known size; anything else returns ncclInvalidUsage / ncclInvalidArgument.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
       ncclInvalidUsage = 5 };

#define MAX_RANKS 64

struct clique {
    int n, live;
};
struct ncclComm {
    struct clique *cl;
    int rank, dev;
};
typedef struct ncclComm *ncclComm_t;

struct op {
    const char *send;
    char *recv;
    size_t bytes;
    struct ncclComm *comm;
    hipStream_t st;
};
static struct op g_ops[MAX_RANKS];
static int g_nops, g_depth;
static ncclResult_t g_group_err;
/* counters a test can read: collectives executed, copies issued */
static uint64_t g_stats[2];

static size_t type_size(int t)
{
    switch (t) {
    case 0: case 1: return 1;          /* int8, uint8 */
    case 2: case 3: case 7: return 4;  /* int32, uint32, float32 */
    case 4: case 5: case 8: return 8;  /* int64, uint64, float64 */
    case 6: case 9: return 2;          /* float16, bfloat16 */
    default: return 0;
    }
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
    case ncclInvalidArgument: return "invalid argument (fake RCCL)";
    case ncclInvalidUsage: return "invalid usage (fake RCCL)";
    default: return "error (fake RCCL)";
    }
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1 || ndev > MAX_RANKS) return ncclInvalidArgument;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return ncclUnhandledCudaError;
    struct clique *cl = (struct clique *)calloc(1, sizeof *cl);
    cl->n = cl->live = ndev;
    for (int i = 0; i < ndev; i++) {
        int d = devlist ? devlist[i] : i;
        if (d < 0 || d >= have) {
            for (int j = 0; j < i; j++) free(comms[j]);
            free(cl);
            return ncclInvalidArgument;
        }
        comms[i] = (struct ncclComm *)calloc(1, sizeof(struct ncclComm));
        comms[i]->cl = cl;
        comms[i]->rank = i;
        comms[i]->dev = d;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    if (--c->cl->live == 0) free(c->cl);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    if (g_depth++ == 0) {
        g_nops = 0;
        g_group_err = ncclSuccess;
    }
    return ncclSuccess;
}

static ncclResult_t run_allgather(void)
{
    if (g_nops == 0) return ncclSuccess;
    struct clique *cl = g_ops[0].comm->cl;
    const int n = cl->n;
    if (g_nops != n) return ncclInvalidUsage; /* a real communicator would hang: one call per rank is the contract */
    struct op *by_rank[MAX_RANKS] = {0};
    for (int i = 0; i < n; i++) {
        struct op *o = &g_ops[i];
        if (o->comm->cl != cl || o->bytes != g_ops[0].bytes || by_rank[o->comm->rank]) return ncclInvalidUsage;
        by_rank[o->comm->rank] = o;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    hipEvent_t ready[MAX_RANKS], done[MAX_RANKS];
    ncclResult_t rc = ncclSuccess;
#define H(x)                                   \
    do {                                       \
        if ((x) != hipSuccess) {               \
            rc = ncclUnhandledCudaError;       \
            goto out;                          \
        }                                      \
    } while (0)
    for (int s = 0; s < n; s++) { /* send blocks are complete once every rank's stream reaches this point */
        H(hipSetDevice(by_rank[s]->comm->dev));
        H(hipEventCreateWithFlags(&ready[s], hipEventDisableTiming));
        H(hipEventCreateWithFlags(&done[s], hipEventDisableTiming));
        H(hipEventRecord(ready[s], by_rank[s]->st));
    }
    for (int r = 0; r < n; r++) {
        struct op *o = by_rank[r];
        H(hipSetDevice(o->comm->dev));
        for (int s = 0; s < n; s++)
            if (s != r) H(hipStreamWaitEvent(o->st, ready[s], 0));
        for (int s = 0; s < n; s++) {
            char *dst = o->recv + (size_t)s * o->bytes;
            if (dst == by_rank[s]->send) continue; /* in place: the rank's own block already sits where it belongs */
            H(hipMemcpyAsync(dst, by_rank[s]->send, o->bytes, hipMemcpyDefault, o->st));
            g_stats[1]++;
        }
        H(hipEventRecord(done[r], o->st));
    }
    for (int s = 0; s < n; s++) { /* nobody overwrites a send block before every rank has read it */
        H(hipSetDevice(by_rank[s]->comm->dev));
        for (int r = 0; r < n; r++)
            if (r != s) H(hipStreamWaitEvent(by_rank[s]->st, done[r], 0));
    }
    g_stats[0]++;
out:
    for (int s = 0; s < n; s++) { /* destruction of a recorded event is deferred by the runtime until it has fired */
        (void)hipEventDestroy(ready[s]);
        (void)hipEventDestroy(done[s]);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
#undef H
}

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = g_group_err != ncclSuccess ? g_group_err : run_allgather();
    g_nops = 0;
    return rc;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    const size_t ts = type_size(datatype);
    if (!sendbuff || !recvbuff || !comm || !ts) return ncclInvalidArgument;
    const int implicit_group = (g_depth == 0);
    if (implicit_group) {
        if (comm->cl->n != 1) return ncclInvalidUsage; /* several ranks in one thread need a group */
        ncclGroupStart();
    }
    if (g_nops >= MAX_RANKS) {
        g_group_err = ncclInvalidUsage;
    } else {
        struct op *o = &g_ops[g_nops++];
        o->send = (const char *)sendbuff;
        o->recv = (char *)recvbuff;
        o->bytes = sendcount * ts;
        o->comm = comm;
        o->st = stream;
    }
    return implicit_group ? ncclGroupEnd() : ncclSuccess;
}

/* test hook (not a RCCL symbol): out[0] = all-gathers executed, out[1] = device copies issued */
void fake_rccl_stats(uint64_t out[2]) { memcpy(out, g_stats, sizeof g_stats); }
