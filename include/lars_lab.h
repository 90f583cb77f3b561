/*
 * lars_lab.h -- C ABI of liblars_lab.so, the LABORATORY library next to liblars_hip.so (`make lab`).  Nothing here is part
 * of the product or of the drop-in boundary: these are the experiments behind DESIGN.md / NOTES.md, kept buildable so that
 * their measurements (profiles/) can be repeated.  Loaded only by tools/lab/ and tests/test_gpu_lab.py.
 */
#ifndef LARS_LAB_H
#define LARS_LAB_H

#include "lars_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device memory of a given kind: 0 plain hipMalloc, 1 uncached, 2 fine-grained, 3 virtual-memory-management block built
 * from chunks of chunk_mb MiB (0 = one handle) mapped back to back (shuffle: in a pseudo-random order) into a range
 * aligned to align_mb MiB, 4 physically contiguous.  Free with lars_lab_free (which also frees assembled arenas). */
int lars_lab_malloc(void **dptr, size_t bytes, int kind, int chunk_mb, int align_mb, int shuffle);
int lars_lab_free(void *dptr);
/* "pipe_steps", "pipe_head", "pipe_trace", "pipe_cold" (a timing experiment that reads the wrong tile on purpose),
 * "arena_chunk_mb", "arena_align_mb", "arena_shuffle" */
int lars_lab_set_tuning(const char *key, int value);

/* The whole step in one persistent launch (csrc/lab/pipeline.hip; 0.87x the speed of the separate launches, NOTES.md): channel histograms -> np.percentile(ch, (2, 98)) ->
 * white-balance tables -> the fused pass, ordered tile by tile so that a tile's second read comes out of the 256 MiB
 * Infinity Cache.  Same results as lars_d_channel_hist + lars_d_wb_table + lars_d_fused, bit for bit.  Serves what the
 * headline configuration needs: uint8 tiles with 3 channels, all three float32 planes written, LARS_F_STATS.
 * args->wb_table is the OUTPUT table buffer here ([ntiles][768]); percentiles is [ntiles][3][2]; hist [ntiles][768] or
 * NULL; scratch holds lars_pipeline_scratch_bytes(ntiles, npix) bytes.  If a wait inside the launch times out the
 * statistics records come back poisoned (count 0, NaN sums). */
size_t lars_pipeline_scratch_bytes(int64_t ntiles, int64_t npix);
int lars_d_pipeline(const lars_fused_args *args, double *percentiles, uint32_t *hist, int rgn_variant, void *scratch);

/* An output arena for the planes lars_d_fused writes, ASSEMBLED from physical memory that measured fast.  How fast the
 * write-bound launch runs is a stable, local property of the physical memory behind its planes (two classes ~15 % apart,
 * DESIGN.md section 4), and a plain allocation is of one kind as it comes.  This call creates physical memory in groups --
 * the chunks behind `group_slots` tile slots of every plane -- maps each group on its own, times `a`'s own launch over the
 * first group_slots tiles into it, keeps the slots / group_slots fastest groups and maps them back to back:
 *   *arena = [plane][slots][npix] of 4-byte samples, planes in the order out_index[0..2] then out_rgba[0..2] for every
 *   pointer of `a` that is not NULL (their values only mark which planes exist; the call sets them for its launches).
 * At most max_groups candidates are created (never fewer than needed; the search stops early once enough lie within 3 % of
 * the fastest); the rest is released.  Free with lars_lab_free.  LARS_ERR_UNSUPPORTED when group_slots * npix * 4 is not a
 * multiple of the device's mapping granularity (take lars_malloc then).  a->stats may point at scratch records. */
typedef struct lars_arena_report {
    int32_t  kind;                 /* 1: assembled from timed groups */
    int32_t  groups_tried;
    int32_t  groups_kept;
    int32_t  rejected;
    uint64_t group_bytes;          /* physical memory per group (all planes) */
    float    search_ms;            /* the whole call on the stream's clock */
    float    chosen_ms;            /* mean ms per probe launch of the kept groups */
    float    slowest_kept_ms;
    float    group_ms[32];         /* ms per probe launch of each candidate, in creation order */
} lars_arena_report;
int lars_d_output_arena(const lars_fused_args *a, int64_t slots, int64_t group_slots, int max_groups, void **arena,
                        lars_arena_report *report);

/* Roofline probes (bench.py reports them beside the kernel numbers when the laboratory library is built): kind 0 reads
 * 16 B/lane, 1 reads 12 B/lane (the fused kernel's load shape), 2 copies 16 B/lane
 * (traffic = 2 x bytes), 3 writes 16 B/lane.  src/dst are device buffers of `bytes`. */
int lars_d_probe(int kind, int unroll, int blocks, const void *src, void *dst, int64_t bytes, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* LARS_LAB_H */
