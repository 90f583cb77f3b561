// Runtime plumbing of liblars_hip.so: per-thread context, errors, memory,
// streams, events.  (Errors, the ABI version and the host-side fold of statistics records: host_core.cpp.)
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include <cmath>
#include <mutex>

#include "common.h"

namespace lars {

static thread_local ThreadCtx g_ctx;

// A thread that used the library gives its stream, workspaces and FFT plan back when it ends (Streamlit sessions
// are threads that come and go).  Thread-local destructors of the main thread run before any atexit handler, so
// the HIP runtime is still alive then.
static void release_ctx(ThreadCtx *c)
{
    if (c->device < 0) return;
    if (hipSetDevice(c->device) != hipSuccess) { c->device = -1; return; }
    align_release();
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); c->stream = nullptr; }
    if (c->ws) { hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
    if (c->scratch) { hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    c->device = -1;
}
struct CtxGuard {
    ~CtxGuard() { release_ctx(&g_ctx); }
};
static thread_local CtxGuard g_ctx_guard;

static int bind_device(ThreadCtx *c, int ordinal)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(LARS_ERR_NO_DEVICE, "no HIP device available (%s); liblars_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (ordinal < 0 || ordinal >= count) return fail(LARS_ERR_INVALID, "device ordinal %d out of range [0,%d)", ordinal, count);
    hipDeviceProp_t prop;
    LARS_HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LARS_ERR_NO_DEVICE, "device %d is %s; liblars_hip is built for gfx950 (MI355X) only", ordinal,
                    prop.gcnArchName);
    LARS_HIP_TRY(hipSetDevice(ordinal));
    if (c->stream) { hipStreamDestroy(c->stream); c->stream = nullptr; }
    if (c->ws) { hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
    if (c->scratch) { hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    LARS_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->device = ordinal;
    (void)&g_ctx_guard;                                   // odr-use: constructs this thread's guard
    return LARS_OK;
}

int ensure_ctx(ThreadCtx **out)
{
    ThreadCtx *c = &g_ctx;
    if (c->device < 0) LARS_TRY(bind_device(c, 0));
    else LARS_HIP_TRY(hipSetDevice(c->device));
    *out = c;
    return LARS_OK;
}

static int reserve(void **p, size_t *have, size_t bytes, hipStream_t s)
{
    if (bytes <= *have) return LARS_OK;
    if (*p) {
        LARS_HIP_TRY(hipStreamSynchronize(s));
        LARS_HIP_TRY(hipFree(*p));
        *p = nullptr; *have = 0;
    }
    size_t want = bytes + (bytes >> 3);                  // 12.5 % slack against regrowth
    want = (want + 0xFFFFF) & ~(size_t)0xFFFFF;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(LARS_ERR_OOM, "hipMalloc(%zu) for the library workspace failed: %s", want, hipGetErrorString(e));
    }
    *have = want;
    return LARS_OK;
}
int ws_reserve(ThreadCtx *c, size_t bytes) { return reserve(&c->ws, &c->ws_bytes, bytes, c->stream); }
int scratch_reserve(ThreadCtx *c, size_t bytes) { return reserve(&c->scratch, &c->scratch_bytes, bytes, c->stream); }

static Tuning g_tuning;
Tuning &tuning() { return g_tuning; }

int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LARS_ERR_HIP, "%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return LARS_OK;
}

}  // namespace lars

using namespace lars;

extern "C" {


int lars_device_count(int *count)
{
    if (!count) return fail(LARS_ERR_INVALID, "lars_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(LARS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return LARS_OK;
}

int lars_set_device(int ordinal) { return bind_device(&g_ctx, ordinal); }

int lars_get_device(int *ordinal)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (ordinal) *ordinal = c->device;
    return LARS_OK;
}

int lars_device_name(char *buf, size_t buflen)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    hipDeviceProp_t prop;
    LARS_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    snprintf(buf, buflen, "%s (%s, %d CUs, %.1f GiB)", prop.name, prop.gcnArchName, prop.multiProcessorCount,
             (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return LARS_OK;
}

// LARS_MALLOC_KIND=3 (experiments, tools/allocbench.py): virtual-memory-management allocations -- one address range, physical
// memory created in chunks of LARS_VMM_CHUNK_MB (0 / unset = the whole allocation in ONE handle) and mapped back to back.
namespace {
struct VmmBlock { size_t bytes, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
std::mutex g_vmm_lock;
std::map<void *, VmmBlock> g_vmm;

int vmm_alloc(int device, void **dptr, size_t bytes)
{
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran)
        return fail(LARS_ERR_HIP, "hipMemGetAllocationGranularity failed");
    const char *chunk_env = getenv("LARS_VMM_CHUNK_MB");
    size_t chunk = chunk_env ? (size_t)atoll(chunk_env) << 20 : 0;
    const size_t total = (bytes + gran - 1) / gran * gran;
    if (!chunk || chunk > total) chunk = total;
    chunk = (chunk + gran - 1) / gran * gran;
    void *base = nullptr;
    const char *align_env = getenv("LARS_VMM_ALIGN_MB");          // alignment of the address range (0 / unset = the driver's default)
    const size_t align = align_env ? (size_t)atoll(align_env) << 20 : 0;
    if (hipMemAddressReserve(&base, total, align, nullptr, 0) != hipSuccess) return fail(LARS_ERR_OOM, "hipMemAddressReserve(%zu) failed", total);
    VmmBlock blk;
    blk.bytes = total;
    blk.chunk = chunk;
    // LARS_VMM_SHUFFLE=1: all chunks are created first and then mapped in a pseudo-random order, so that neighbouring
    // addresses are backed by physical memory from unrelated places
    const bool shuffle = getenv("LARS_VMM_SHUFFLE") && getenv("LARS_VMM_SHUFFLE")[0] == '1';
    const size_t nchunks = (total + chunk - 1) / chunk;
    auto undo = [&](size_t mapped) {
        for (size_t i = 0; i < mapped; ++i) hipMemUnmap(static_cast<char *>(base) + i * chunk, (i + 1) * chunk <= total ? chunk : total - i * chunk);
        for (auto &hh : blk.handles) hipMemRelease(hh);
        hipMemAddressFree(base, total);
    };
    for (size_t i = 0; i < nchunks; ++i) {
        const size_t n = (i + 1) * chunk <= total ? chunk : total - i * chunk;
        hipMemGenericAllocationHandle_t h;
        const hipError_t e = hipMemCreate(&h, n, &prop, 0);
        if (e != hipSuccess) { undo(0); return fail(LARS_ERR_OOM, "hipMemCreate(%zu): %s", n, hipGetErrorString(e)); }
        blk.handles.push_back(h);
    }
    if (shuffle && total % chunk == 0) {
        unsigned long long x = 0x9E3779B97F4A7C15ull;
        for (size_t i = nchunks - 1; i > 0; --i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(blk.handles[i], blk.handles[(size_t)(x % (i + 1))]);
        }
    }
    for (size_t i = 0; i < nchunks; ++i) {
        const size_t n = (i + 1) * chunk <= total ? chunk : total - i * chunk;
        const hipError_t e = hipMemMap(static_cast<char *>(base) + i * chunk, n, 0, blk.handles[i], 0);
        if (e != hipSuccess) { undo(i); return fail(LARS_ERR_OOM, "hipMemMap(%zu): %s", n, hipGetErrorString(e)); }
    }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, total, &acc, 1) != hipSuccess) { undo(nchunks); return fail(LARS_ERR_HIP, "hipMemSetAccess failed"); }
    {
        std::lock_guard<std::mutex> g(g_vmm_lock);
        g_vmm[base] = blk;
    }
    *dptr = base;
    return LARS_OK;
}
// true if dptr was a VMM block (and is gone now)
bool vmm_free(void *dptr)
{
    VmmBlock blk;
    {
        std::lock_guard<std::mutex> g(g_vmm_lock);
        auto it = g_vmm.find(dptr);
        if (it == g_vmm.end()) return false;
        blk = it->second;
        g_vmm.erase(it);
    }
    hipMemUnmap(dptr, blk.bytes);
    for (auto &h : blk.handles) hipMemRelease(h);
    hipMemAddressFree(dptr, blk.bytes);
    return true;
}
}  // namespace

int lars_malloc(void **dptr, size_t bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!dptr) return fail(LARS_ERR_INVALID, "lars_malloc: NULL");
    // LARS_MALLOC_KIND (experiments, tools/allocbench.py): 1 uncached, 2 fine-grained device memory, 3 virtual-memory-management
    // blocks, 4 physically contiguous memory (hipDeviceMallocContiguous) instead of plain hipMalloc
    const char *kind = getenv("LARS_MALLOC_KIND");
    hipError_t e;
    if (kind && kind[0] == '3' && bytes >= (64u << 20)) return vmm_alloc(c->device, dptr, bytes);
    if (kind && kind[0] == '1') e = hipExtMallocWithFlags(dptr, bytes ? bytes : 1, hipDeviceMallocUncached);
    else if (kind && kind[0] == '2') e = hipExtMallocWithFlags(dptr, bytes ? bytes : 1, hipDeviceMallocFinegrained);
    else if (kind && kind[0] == '4' && bytes >= (64u << 20)) e = hipExtMallocWithFlags(dptr, bytes, hipDeviceMallocContiguous);
    else e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) { *dptr = nullptr; return fail(LARS_ERR_OOM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return LARS_OK;
}
int lars_free(void *dptr)
{
    if (!dptr) return LARS_OK;
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (vmm_free(dptr)) return LARS_OK;
    LARS_HIP_TRY(hipFree(dptr));
    return LARS_OK;
}
int lars_memset(void *dptr, int value, size_t bytes, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    LARS_HIP_TRY(hipMemsetAsync(dptr, value, bytes, pick_stream(c, stream)));
    return LARS_OK;
}
int lars_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    LARS_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    LARS_HIP_TRY(hipStreamSynchronize(c->stream));
    return LARS_OK;
}
int lars_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    LARS_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    LARS_HIP_TRY(hipStreamSynchronize(c->stream));
    return LARS_OK;
}
int lars_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    LARS_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, pick_stream(c, stream)));
    return LARS_OK;
}
int lars_stream_create(void **stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    hipStream_t s;
    LARS_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return LARS_OK;
}
int lars_stream_destroy(void *stream)
{
    if (stream) LARS_HIP_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return LARS_OK;
}
int lars_synchronize(void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    LARS_HIP_TRY(hipStreamSynchronize(pick_stream(c, stream)));
    return LARS_OK;
}
int lars_shutdown(void)
{
    release_ctx(&g_ctx);
    return LARS_OK;
}

int lars_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    size_t f = 0, t = 0;
    LARS_HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return LARS_OK;
}

int lars_event_create(void **event)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    hipEvent_t e;
    LARS_HIP_TRY(hipEventCreate(&e));
    *event = e;
    return LARS_OK;
}
int lars_event_destroy(void *event)
{
    if (event) LARS_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return LARS_OK;
}
int lars_event_record(void *event, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    LARS_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), pick_stream(c, stream)));
    return LARS_OK;
}
int lars_event_elapsed_ms(void *start, void *stop, float *ms)
{
    LARS_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    LARS_HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return LARS_OK;
}

int lars_set_tuning(const char *key, int value)
{
    if (!key) return fail(LARS_ERR_INVALID, "lars_set_tuning: NULL key");
    Tuning &t = tuning();
    if (!strcmp(key, "fused_impl")) t.fused_impl = value;
    else if (!strcmp(key, "hist_impl")) t.hist_impl = value;
    else if (!strcmp(key, "nt_stores")) t.nt_stores = value;
    else if (!strcmp(key, "blocks_per_tile")) t.blocks_per_tile = value;
    else if (!strcmp(key, "traverse")) t.traverse = value;
    else if (!strcmp(key, "grid_swap")) t.grid_swap = value;
    else if (!strcmp(key, "count_mode")) t.count_mode = value;
    else if (!strcmp(key, "pipe_steps")) t.pipe_steps = value;
    else if (!strcmp(key, "selq_window")) t.selq_window = value;
    else if (!strcmp(key, "selq_list_wgs")) t.selq_list_wgs = value;
    else if (!strcmp(key, "pipe_head")) t.pipe_head = value;
    else if (!strcmp(key, "pipe_trace")) t.pipe_trace = value;
    else if (!strcmp(key, "pipe_cold")) t.pipe_cold = value;
    else if (!strcmp(key, "joint_depth")) t.joint_depth = value;
    else return fail(LARS_ERR_INVALID, "lars_set_tuning: unknown key %s", key);
    return LARS_OK;
}
int lars_get_tuning(const char *key, int *value)
{
    if (!key || !value) return fail(LARS_ERR_INVALID, "lars_get_tuning: NULL");
    const Tuning &t = tuning();
    if (!strcmp(key, "fused_impl")) *value = t.fused_impl;
    else if (!strcmp(key, "hist_impl")) *value = t.hist_impl;
    else if (!strcmp(key, "nt_stores")) *value = t.nt_stores;
    else if (!strcmp(key, "blocks_per_tile")) *value = t.blocks_per_tile;
    else if (!strcmp(key, "traverse")) *value = t.traverse;
    else if (!strcmp(key, "grid_swap")) *value = t.grid_swap;
    else if (!strcmp(key, "count_mode")) *value = t.count_mode;
    else if (!strcmp(key, "pipe_steps")) *value = t.pipe_steps;
    else if (!strcmp(key, "selq_window")) *value = t.selq_window;
    else if (!strcmp(key, "selq_list_wgs")) *value = t.selq_list_wgs;
    else if (!strcmp(key, "pipe_head")) *value = t.pipe_head;
    else if (!strcmp(key, "pipe_trace")) *value = t.pipe_trace;
    else if (!strcmp(key, "pipe_cold")) *value = t.pipe_cold;
    else if (!strcmp(key, "joint_depth")) *value = t.joint_depth;
    else return fail(LARS_ERR_INVALID, "lars_get_tuning: unknown key %s", key);
    return LARS_OK;
}

// Exhaustive device self-check of the rcp+fma quotient against IEEE division.
int lars_d_quot_selfcheck(uint32_t max_den, uint64_t *mismatches, uint32_t first_bad[2])
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!mismatches || !first_bad || max_den < 1) return fail(LARS_ERR_INVALID, "lars_d_quot_selfcheck: bad arguments");
    LARS_TRY(scratch_reserve(c, 64));
    unsigned long long *d = static_cast<unsigned long long *>(c->scratch);
    LARS_HIP_TRY(hipMemsetAsync(d, 0, 64, c->stream));
    LARS_TRY(quot_check_launch(max_den, d, reinterpret_cast<unsigned int *>(d + 1), c->stream));
    unsigned long long h[2];
    LARS_HIP_TRY(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, c->stream));
    LARS_HIP_TRY(hipStreamSynchronize(c->stream));
    *mismatches = h[0];
    first_bad[0] = (uint32_t)(h[1] & 0xFFFFFFFFu);
    first_bad[1] = (uint32_t)(h[1] >> 32);
    return LARS_OK;
}

}  // extern "C"
