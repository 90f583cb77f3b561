// Multi-GPU: one process per GPU, RCCL over xGMI.  The hot path shards by tile
// with no data-path exchange; the only collective is the fold of the per-index
// statistics records: one ncclAllGather of the packed records
// followed by a local fold in rank order, so every rank gets bit-identical
// global statistics.
//
// librccl.so is loaded lazily with dlopen so that single-GPU users (the
// Streamlit host) do not need it on the loader path.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"

using namespace lars;

namespace {

typedef struct { char internal[LARS_COMM_ID_BYTES]; } nccl_uid;
typedef void *nccl_comm;
typedef int nccl_result;
enum { NCCL_UINT8 = 1, NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3 };

struct Rccl {
    void *lib = nullptr;
    nccl_result (*GetUniqueId)(nccl_uid *) = nullptr;
    nccl_result (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
    nccl_result (*CommDestroy)(nccl_comm) = nullptr;
    nccl_result (*CommCount)(const nccl_comm, int *) = nullptr;
    nccl_result (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    nccl_result (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(nccl_result) = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return LARS_OK;
    const char *env = getenv("LARS_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) return fail(LARS_ERR_RCCL, "cannot load librccl.so (set LARS_RCCL_LIB): %s", dlerror());
#define SYM(field, name)                                                                   \
    *(void **)(&g_rccl.field) = dlsym(lib, name);                                          \
    if (!g_rccl.field) { dlclose(lib); return fail(LARS_ERR_RCCL, "librccl.so lacks %s", name); }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.lib = lib;
    return LARS_OK;
}

#define RCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        nccl_result _r = (expr);                                                                             \
        if (_r != 0) return fail(LARS_ERR_RCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));           \
    } while (0)

struct Comm {
    nccl_comm comm = nullptr;
    int nranks = 1, rank = 0;
    void *dbuf = nullptr;          // device staging: [send | recv]
    size_t dbuf_bytes = 0;
};

int comm_buf(Comm *cm, size_t bytes)
{
    if (bytes <= cm->dbuf_bytes) return LARS_OK;
    if (cm->dbuf) LARS_HIP_TRY(hipFree(cm->dbuf));
    cm->dbuf = nullptr; cm->dbuf_bytes = 0;
    LARS_HIP_TRY(hipMalloc(&cm->dbuf, bytes));
    cm->dbuf_bytes = bytes;
    return LARS_OK;
}

}  // namespace

extern "C" {

// LARS_OK if librccl can be loaded and has every symbol the communicator needs: what every rank checks (and tells the
// others, dist.agree) BEFORE anybody enters ncclCommInitRank, which would wait forever for a rank that cannot join
int lars_comm_available(void) { return load_rccl(); }

int lars_comm_unique_id(uint8_t *id_out)
{
    if (!id_out) return fail(LARS_ERR_INVALID, "lars_comm_unique_id: NULL");
    LARS_TRY(load_rccl());
    nccl_uid id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, LARS_COMM_ID_BYTES);
    return LARS_OK;
}

int lars_comm_init(void **comm, int nranks, int rank, const uint8_t *unique_id)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));                      // binds the calling thread's device first
    if (!comm || !unique_id || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(LARS_ERR_INVALID, "lars_comm_init: bad arguments");
    LARS_TRY(load_rccl());
    nccl_uid id;
    memcpy(id.internal, unique_id, LARS_COMM_ID_BYTES);
    Comm *cm = new Comm;
    cm->nranks = nranks; cm->rank = rank;
    nccl_result r = g_rccl.CommInitRank(&cm->comm, nranks, id, rank);
    if (r != 0) {
        delete cm;
        return fail(LARS_ERR_RCCL, "ncclCommInitRank(nranks=%d, rank=%d) failed: %s", nranks, rank, g_rccl.GetErrorString(r));
    }
    *comm = cm;
    return LARS_OK;
}

int lars_comm_count(void *comm, int *nranks_out)
{
    Comm *cm = static_cast<Comm *>(comm);
    if (!cm || !cm->comm || !nranks_out) return fail(LARS_ERR_INVALID, "lars_comm_count: bad arguments");
    int n = 0;
    RCCL_TRY(g_rccl.CommCount(cm->comm, &n));          // what RCCL itself joined, not what the caller asked for
    *nranks_out = n;
    return LARS_OK;
}

int lars_comm_destroy(void *comm)
{
    if (!comm) return LARS_OK;
    Comm *cm = static_cast<Comm *>(comm);
    if (cm->dbuf) hipFree(cm->dbuf);
    if (cm->comm) g_rccl.CommDestroy(cm->comm);
    delete cm;
    return LARS_OK;
}

int lars_comm_allreduce_stats(void *comm, lars_stats *records, int64_t n, int is_device, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    Comm *cm = static_cast<Comm *>(comm);
    if (!cm || !records || n <= 0) return fail(LARS_ERR_INVALID, "lars_comm_allreduce_stats: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    const size_t bytes = (size_t)n * sizeof(lars_stats);
    LARS_TRY(comm_buf(cm, bytes * (1 + (size_t)cm->nranks)));
    char *send = static_cast<char *>(cm->dbuf);
    char *recv = send + bytes;
    LARS_HIP_TRY(hipMemcpyAsync(send, records, bytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    RCCL_TRY(g_rccl.AllGather(send, recv, bytes, NCCL_UINT8, cm->comm, s));
    std::vector<lars_stats> all((size_t)n * cm->nranks);
    LARS_HIP_TRY(hipMemcpyAsync(all.data(), recv, bytes * cm->nranks, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    std::vector<lars_stats> folded((size_t)n), column((size_t)cm->nranks);
    for (int64_t i = 0; i < n; ++i) {
        for (int r = 0; r < cm->nranks; ++r) column[r] = all[(size_t)r * n + i];
        LARS_TRY(lars_stats_merge(column.data(), cm->nranks, &folded[i]));
    }
    if (is_device) {
        LARS_HIP_TRY(hipMemcpyAsync(records, folded.data(), bytes, hipMemcpyHostToDevice, s));
        LARS_HIP_TRY(hipStreamSynchronize(s));
    } else {
        memcpy(records, folded.data(), bytes);
    }
    return LARS_OK;
}

int lars_comm_allreduce_f64(void *comm, double *values_host, int64_t n, int op)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    Comm *cm = static_cast<Comm *>(comm);
    if (!cm || !values_host || n <= 0 || op < 0 || op > 2) return fail(LARS_ERR_INVALID, "lars_comm_allreduce_f64: bad arguments");
    const size_t bytes = (size_t)n * sizeof(double);
    LARS_TRY(comm_buf(cm, 2 * bytes));
    double *d = static_cast<double *>(cm->dbuf);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(d, values_host, bytes, hipMemcpyHostToDevice, s));
    const int nop = op == 0 ? NCCL_SUM : (op == 1 ? NCCL_MAX : NCCL_MIN);
    RCCL_TRY(g_rccl.AllReduce(d, d + n, (size_t)n, NCCL_FLOAT64, nop, cm->comm, s));
    LARS_HIP_TRY(hipMemcpyAsync(values_host, d + n, bytes, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

int lars_comm_barrier(void *comm)
{
    double one = 1.0;
    return lars_comm_allreduce_f64(comm, &one, 1, 0);
}

}  // extern "C"
