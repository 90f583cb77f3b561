// LABORATORY: output arenas assembled from timed groups of physical memory.  Measured in round 3 and found NOT to deliver:
// every candidate group probes alike, and the assembled arena's speed is not predicted by its groups' probe times
// (profiles/r03_arena_assembled.txt).  Kept for tools/lab/arenalab.py; the product chooses among whole plain allocations.
//
// How fast the write-bound fused kernel (three float32 planes per 3 bytes read) runs is a LOCAL, stable property of the
// physical memory its planes live in: 5.2-5.4 TB/s into "slow" memory, 6.1-6.3 into "fast", reproducibly per piece, only
// visible once a launch streams through gigabytes, and only for writes (profiles/r02_placement_*.txt).
// A plain hipMalloc of 12 GiB is almost always of one kind, and only about a third of them come out fast, so an arena
// taken as it comes is a lottery.  Here the arena is BUILT: physical memory is created in groups (the chunks behind
// `group_slots` tile slots of every plane, 3 GiB for three planes of 16 slots of 4096 x 4096 float32), each group is
// mapped on its own and timed with the caller's own fused launch, the fastest groups are kept and mapped back to back
// into one address range with the layout [plane][slot][npix]; the others go back to the driver.
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "lab.h"

namespace lars {
namespace {

struct Piece { size_t offset, bytes; hipMemGenericAllocationHandle_t handle; };
struct Arena { size_t bytes; std::vector<Piece> pieces; };
std::mutex g_lock;
std::map<void *, Arena> g_arenas;

struct Group {
    std::vector<hipMemGenericAllocationHandle_t> handles;      // [plane][chunk]
    float ms = 0.0f;
};

void release_group(Group &g)
{
    for (auto &h : g.handles) hipMemRelease(h);
    g.handles.clear();
}

}  // namespace

// lars_free: true if dptr was an assembled arena (and is gone now)
bool arena_free(void *dptr)
{
    Arena a;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        auto it = g_arenas.find(dptr);
        if (it == g_arenas.end()) return false;
        a = std::move(it->second);
        g_arenas.erase(it);
    }
    hipMemUnmap(dptr, a.bytes);
    for (auto &p : a.pieces) hipMemRelease(p.handle);
    hipMemAddressFree(dptr, a.bytes);
    return true;
}

}  // namespace lars

using namespace lars;

extern "C" int lars_d_output_arena(const lars_fused_args *a, int64_t slots, int64_t group_slots, int max_groups, void **arena,
                                   lars_arena_report *report)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!a || !arena || !a->tiles || a->npix <= 0 || slots <= 0 || group_slots <= 0 || a->ntiles < group_slots)
        return fail(LARS_ERR_INVALID, "lars_d_output_arena: bad arguments (the probe launch needs group_slots tiles)");
    if (slots % group_slots) return fail(LARS_ERR_INVALID, "lars_d_output_arena: slots must be a multiple of group_slots");
    // planes in arena order: the float32 index planes, then the RGBA8 planes (4 bytes per pixel each)
    int plane_kind[6], plane_k[6], nplanes = 0;
    for (int k = 0; k < 3; ++k) if (a->out_index[k]) { plane_kind[nplanes] = 0; plane_k[nplanes++] = k; }
    for (int k = 0; k < 3; ++k) if (a->out_rgba[k]) { plane_kind[nplanes] = 1; plane_k[nplanes++] = k; }
    if (!nplanes) return fail(LARS_ERR_INVALID, "lars_d_output_arena: mark the planes to place with non-NULL out_index / out_rgba");

    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran)
        return fail(LARS_ERR_UNSUPPORTED, "lars_d_output_arena: no virtual memory management on this device");
    const size_t slot_bytes = (size_t)a->npix * 4;
    const size_t group_bytes = slot_bytes * (size_t)group_slots;          // one plane's share of a group
    if (group_bytes % gran)
        return fail(LARS_ERR_UNSUPPORTED, "lars_d_output_arena: %lld slots of %zu bytes are not a multiple of the mapping granularity %zu",
                    (long long)group_slots, slot_bytes, gran);
    size_t chunk = (size_t)(lab_tuning().arena_chunk_mb > 0 ? lab_tuning().arena_chunk_mb : 64) << 20;   // physical pieces (or the whole share if smaller / odd)
    if (chunk % gran || group_bytes % chunk) chunk = group_bytes;
    const size_t va_align = (size_t)(lab_tuning().arena_align_mb > 0 ? lab_tuning().arena_align_mb : 0) << 20;
    const size_t chunks_per_plane = group_bytes / chunk;
    const int need = (int)(slots / group_slots);
    if (max_groups < need) max_groups = need;
    if (max_groups > 32) max_groups = 32;
    hipStream_t s = pick_stream(c, a->stream);
    hipEvent_t ev[2];
    LARS_HIP_TRY(hipEventCreate(&ev[0]));
    LARS_HIP_TRY(hipEventCreate(&ev[1]));

    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    std::vector<Group> groups;
    int status = LARS_OK;
    hipEvent_t t_all[2];
    hipEventCreate(&t_all[0]); hipEventCreate(&t_all[1]);
    hipEventRecord(t_all[0], s);
    const size_t probe_bytes = group_bytes * nplanes;
    void *probe_va = nullptr;
    if (hipMemAddressReserve(&probe_va, probe_bytes, va_align, nullptr, 0) != hipSuccess) {
        status = fail(LARS_ERR_OOM, "lars_d_output_arena: hipMemAddressReserve(%zu) failed", probe_bytes);
        probe_va = nullptr;
    }
    // Candidates are created and timed one by one and ALL kept until the choice is made (released memory would be handed out
    // again).  The search ends once `need` groups lie within 3 % of the fastest seen and at least need + 2 were tried
    // (the classes are ~12 % apart), or at max_groups, or when the device runs short of memory.
    float best = 0.0f;
    while (status == LARS_OK && (int)groups.size() < max_groups) {
        if ((int)groups.size() >= need + 2) {
            int fast = 0;
            for (auto &g : groups) fast += g.ms <= best * 1.03f ? 1 : 0;
            if (fast >= need) break;
        }
        size_t free_b = 0, total_b = 0;
        if ((int)groups.size() >= need && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < probe_bytes + ((size_t)8 << 30))) break;
        Group g;
        bool ok = true;
        for (size_t i = 0; i < (size_t)nplanes * chunks_per_plane && ok; ++i) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) ok = false;
            else g.handles.push_back(h);
        }
        if (ok && lab_tuning().arena_shuffle) {
            static unsigned long long x = 0x9E3779B97F4A7C15ull;
            for (size_t i = g.handles.size() - 1; i > 0; --i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                std::swap(g.handles[i], g.handles[(size_t)(x % (i + 1))]);
            }
        }
        if (!ok) {
            (void)hipGetLastError();
            release_group(g);
            if ((int)groups.size() >= need) break;                        // out of memory: choose among what there is
            status = fail(LARS_ERR_OOM, "lars_d_output_arena: hipMemCreate failed after %zu groups (%d needed)", groups.size(), need);
            break;
        }
        size_t mapped = 0;
        for (; mapped < g.handles.size(); ++mapped)
            if (hipMemMap(static_cast<char *>(probe_va) + mapped * chunk, chunk, 0, g.handles[mapped], 0) != hipSuccess) break;
        if (mapped != g.handles.size() || hipMemSetAccess(probe_va, probe_bytes, &acc, 1) != hipSuccess) {
            if (mapped) hipMemUnmap(probe_va, mapped * chunk);
            release_group(g);
            status = fail(LARS_ERR_HIP, "lars_d_output_arena: mapping a candidate group failed");
            break;
        }
        // the caller's own launch over the first group_slots tiles, planes pointed at the candidate: one warm-up, two timed
        lars_fused_args p = *a;
        p.ntiles = group_slots;
        p.stream = s;
        for (int k = 0; k < 3; ++k) { p.out_index[k] = nullptr; p.out_rgba[k] = nullptr; }
        for (int j = 0; j < nplanes; ++j) {
            void *base = static_cast<char *>(probe_va) + (size_t)j * group_bytes;
            if (plane_kind[j] == 0) p.out_index[plane_k[j]] = static_cast<float *>(base);
            else p.out_rgba[plane_k[j]] = static_cast<uint8_t *>(base);
        }
        int st = lars_d_fused(&p);
        hipEventRecord(ev[0], s);
        if (st == LARS_OK) st = lars_d_fused(&p);
        if (st == LARS_OK) st = lars_d_fused(&p);
        hipEventRecord(ev[1], s);
        hipError_t e = hipEventSynchronize(ev[1]);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev[0], ev[1]);
        hipMemUnmap(probe_va, probe_bytes);
        if (st != LARS_OK || e != hipSuccess) {
            release_group(g);
            status = st != LARS_OK ? st : fail(LARS_ERR_HIP, "lars_d_output_arena: timing a candidate group failed: %s", hipGetErrorString(e));
            break;
        }
        g.ms = ms / 2.0f;
        if (groups.empty() || g.ms < best) best = g.ms;
        groups.push_back(std::move(g));
    }
    if (probe_va) hipMemAddressFree(probe_va, probe_bytes);
    hipEventDestroy(ev[0]); hipEventDestroy(ev[1]);
    if (status == LARS_OK && (int)groups.size() < need)
        status = fail(LARS_ERR_OOM, "lars_d_output_arena: only %zu of %d groups could be created", groups.size(), need);
    if (status != LARS_OK) {
        for (auto &g : groups) release_group(g);
        hipEventDestroy(t_all[0]); hipEventDestroy(t_all[1]);
        return status;
    }

    // keep the `need` fastest (in the order they were created), give the rest back
    std::vector<int> order(groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return groups[x].ms < groups[y].ms; });
    std::vector<int> kept(order.begin(), order.begin() + need);
    std::sort(kept.begin(), kept.end());
    if (report) {
        memset(report, 0, sizeof *report);
        report->kind = 1;
        report->groups_tried = (int)groups.size();
        report->groups_kept = need;
        report->group_bytes = (uint64_t)probe_bytes;
        for (size_t i = 0; i < groups.size() && i < 32; ++i) report->group_ms[i] = groups[i].ms;
        double sum = 0, worst_kept = 0;
        for (int i : kept) { sum += groups[i].ms; worst_kept = std::max(worst_kept, (double)groups[i].ms); }
        report->chosen_ms = (float)(sum / need);
        report->slowest_kept_ms = (float)worst_kept;
        report->rejected = (int)groups.size() - need;
    }
    const size_t plane_bytes = slot_bytes * (size_t)slots, total = plane_bytes * nplanes;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, total, va_align, nullptr, 0) != hipSuccess) {
        for (auto &g : groups) release_group(g);
        hipEventDestroy(t_all[0]); hipEventDestroy(t_all[1]);
        return fail(LARS_ERR_OOM, "lars_d_output_arena: hipMemAddressReserve(%zu) failed", total);
    }
    Arena ar;
    ar.bytes = total;
    bool ok = true;
    for (int gi = 0; gi < need && ok; ++gi) {
        Group &g = groups[kept[gi]];
        for (int j = 0; j < nplanes && ok; ++j)
            for (size_t ch = 0; ch < chunks_per_plane && ok; ++ch) {
                const size_t off = (size_t)j * plane_bytes + (size_t)gi * group_bytes + ch * chunk;
                hipMemGenericAllocationHandle_t h = g.handles[(size_t)j * chunks_per_plane + ch];
                if (hipMemMap(static_cast<char *>(base) + off, chunk, 0, h, 0) != hipSuccess) ok = false;
                else ar.pieces.push_back({off, chunk, h});
            }
    }
    if (ok && hipMemSetAccess(base, total, &acc, 1) != hipSuccess) ok = false;
    std::vector<bool> is_kept(groups.size(), false);
    for (int i : kept) is_kept[i] = true;
    if (!ok) {
        for (auto &p : ar.pieces) hipMemUnmap(static_cast<char *>(base) + p.offset, p.bytes);
        for (auto &g : groups) release_group(g);
        hipMemAddressFree(base, total);
        hipEventDestroy(t_all[0]); hipEventDestroy(t_all[1]);
        return fail(LARS_ERR_HIP, "lars_d_output_arena: mapping the arena failed");
    }
    for (size_t i = 0; i < groups.size(); ++i)
        if (!is_kept[i]) release_group(groups[i]);
    hipEventRecord(t_all[1], s);
    hipEventSynchronize(t_all[1]);
    float search_ms = 0.0f;
    hipEventElapsedTime(&search_ms, t_all[0], t_all[1]);
    hipEventDestroy(t_all[0]); hipEventDestroy(t_all[1]);
    if (report) report->search_ms = search_ms;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        g_arenas[base] = std::move(ar);
    }
    *arena = base;
    return LARS_OK;
}
