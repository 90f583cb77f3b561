// LABORATORY: device allocations of other kinds than plain hipMalloc, for the placement experiments (tools/lab/): uncached,
// fine-grained, physically contiguous, and virtual-memory-management blocks built from chunks.  Round 2 drove these through
// environment variables read inside lars_malloc; the product library no longer knows them.
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "lab.h"

namespace lars {


namespace {
struct VmmBlock { size_t bytes, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
std::mutex g_vmm_lock;
std::map<void *, VmmBlock> g_vmm;

int vmm_alloc(int device, void **dptr, size_t bytes, size_t chunk, size_t align, bool shuffle)
{
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran)
        return fail(LARS_ERR_HIP, "hipMemGetAllocationGranularity failed");
    const size_t total = (bytes + gran - 1) / gran * gran;
    if (!chunk || chunk > total) chunk = total;
    chunk = (chunk + gran - 1) / gran * gran;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, total, align, nullptr, 0) != hipSuccess) return fail(LARS_ERR_OOM, "hipMemAddressReserve(%zu) failed", total);
    VmmBlock blk;
    blk.bytes = total;
    blk.chunk = chunk;
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
        // all chunks are created first and then mapped in a pseudo-random order, so that neighbouring addresses are backed by
        // physical memory from unrelated places
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
}  // namespace

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

}  // namespace lars

using namespace lars;

extern "C" int lars_lab_malloc(void **dptr, size_t bytes, int kind, int chunk_mb, int align_mb, int shuffle)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!dptr) return fail(LARS_ERR_INVALID, "lars_lab_malloc: NULL");
    hipError_t e;
    if (kind == 3) return vmm_alloc(c->device, dptr, bytes, (size_t)(chunk_mb > 0 ? chunk_mb : 0) << 20, (size_t)(align_mb > 0 ? align_mb : 0) << 20, shuffle != 0);
    if (kind == 1) e = hipExtMallocWithFlags(dptr, bytes ? bytes : 1, hipDeviceMallocUncached);
    else if (kind == 2) e = hipExtMallocWithFlags(dptr, bytes ? bytes : 1, hipDeviceMallocFinegrained);
    else if (kind == 4) e = hipExtMallocWithFlags(dptr, bytes ? bytes : 1, hipDeviceMallocContiguous);
    else e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) { *dptr = nullptr; return fail(LARS_ERR_OOM, "lars_lab_malloc(%zu, kind %d): %s", bytes, kind, hipGetErrorString(e)); }
    return LARS_OK;
}

extern "C" int lars_lab_free(void *dptr)
{
    if (!dptr) return LARS_OK;
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (vmm_free(dptr)) return LARS_OK;
    LARS_HIP_TRY(hipFree(dptr));
    return LARS_OK;
}

extern "C" int lars_lab_copy(int mode, int to_device, void *host, void *dev, size_t bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!host || !dev) return fail(LARS_ERR_INVALID, "lars_lab_copy: NULL");
    void *dst = to_device ? dev : host;
    const void *src = to_device ? host : dev;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    if (mode == 0) {
        LARS_HIP_TRY(hipMemcpy(dst, src, bytes, kind));
    } else {
        LARS_HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, c->stream));
        LARS_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return LARS_OK;
}

