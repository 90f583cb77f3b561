// Runtime plumbing of liblars_hip.so: per-thread context, errors, memory,
// streams, events.  (Errors, the ABI version and the host-side fold of statistics records: host_core.cpp.)
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>

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

int lars_malloc(void **dptr, size_t bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!dptr) return fail(LARS_ERR_INVALID, "lars_malloc: NULL");
    const hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) { *dptr = nullptr; return fail(LARS_ERR_OOM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return LARS_OK;
}
int lars_free(void *dptr)
{
    if (!dptr) return LARS_OK;
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
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
int lars_stream_wait_event(void *stream, void *event)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!event) return fail(LARS_ERR_INVALID, "lars_stream_wait_event: NULL event");
    LARS_HIP_TRY(hipStreamWaitEvent(pick_stream(c, stream), reinterpret_cast<hipEvent_t>(event), 0));
    return LARS_OK;
}
int lars_event_elapsed_ms(void *start, void *stop, float *ms)
{
    LARS_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    LARS_HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return LARS_OK;
}

// Which build-time switches of the kernel sources this library was compiled with: 0 for the product.  A laboratory build
// (csrc/Makefile: lablayout, EXTRA=-D...) answers with the bits of what it changed, so that no measurement or test can mistake it.
unsigned int lars_build_flags(void)
{
    unsigned int f = 0;
#ifdef LARS_LAB_LAYOUT
    f |= LARS_BUILD_LAB_LAYOUT;
#endif
#ifdef LARS_IEEE_DIV
    f |= LARS_BUILD_IEEE_DIV;
#endif
#if defined(LARS_COUNT_MODE) && LARS_COUNT_MODE != 0
    f |= LARS_BUILD_COUNT_MODE;
#endif
#if LARS_V2_STATS_THREADS != 1024 || LARS_V2_STATS_WAVES != 8
    f |= LARS_BUILD_STATS_GEOMETRY;
#endif
    return f;
}

int lars_set_tuning(const char *key, int value)
{
    if (!key) return fail(LARS_ERR_INVALID, "lars_set_tuning: NULL key");
    Tuning &t = tuning();
    if (!strcmp(key, "fused_impl")) t.fused_impl = value;
    else if (!strcmp(key, "hist_impl")) t.hist_impl = value;
    else if (!strcmp(key, "nt_stores")) t.nt_stores = value;
    else if (!strcmp(key, "blocks_per_tile")) t.blocks_per_tile = value;
    else if (!strcmp(key, "selq_window")) t.selq_window = value;
    else if (!strcmp(key, "selq_list_wgs")) t.selq_list_wgs = value;
    else if (!strcmp(key, "joint_depth")) {
        if (value != 4 && value != 6 && value != 8 && value != 12) return fail(LARS_ERR_INVALID, "lars_set_tuning: joint_depth is 4, 6, 8 or 12 (got %d)", value);
        t.joint_depth = value;
    }
    else if (!strcmp(key, "joint_window")) {
        if (value < 0 || value > 5) return fail(LARS_ERR_INVALID, "lars_set_tuning: joint_window is 0 .. 5 (got %d)", value);
        t.joint_window = value;
    }
    else if (!strcmp(key, "joint_win_depth")) {
        if (value != 4 && value != 5 && value != 6 && value != 12 && value != 15)
            return fail(LARS_ERR_INVALID, "lars_set_tuning: joint_win_depth is 4, 5, 6, 12 or 15 (got %d)", value);
        t.joint_win_depth = value;
    }
    else if (!strcmp(key, "out_stride_planes")) {
#ifdef LARS_LAB_LAYOUT
        t.out_stride_planes = value;
#else
        return fail(LARS_ERR_INVALID, "lars_set_tuning: out_stride_planes exists in the laboratory build only (make lablayout); this library would ignore it");
#endif
    }
    else if (!strcmp(key, "u16_hist_impl")) {
        if (value != 1 && value != 3 && value != 5) return fail(LARS_ERR_INVALID, "lars_set_tuning: u16_hist_impl is 5, 1 or 3 (got %d)", value);
        t.u16_hist_impl = value;
    }
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
    else if (!strcmp(key, "selq_window")) *value = t.selq_window;
    else if (!strcmp(key, "selq_list_wgs")) *value = t.selq_list_wgs;
    else if (!strcmp(key, "joint_depth")) *value = t.joint_depth;
    else if (!strcmp(key, "joint_window")) *value = t.joint_window;
    else if (!strcmp(key, "joint_win_depth")) *value = t.joint_win_depth;
    else if (!strcmp(key, "out_stride_planes")) *value = t.out_stride_planes;
    else if (!strcmp(key, "u16_hist_impl")) *value = t.u16_hist_impl;
    else if (!strcmp(key, "last_fused_kernel")) *value = t.last_fused_kernel;
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
