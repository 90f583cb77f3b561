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
 * aligned to align_mb MiB, 4 physically contiguous.  Free with lars_lab_free.  (Round 3 also built output arenas from
 * timed groups of such chunks -- lars_d_output_arena, removed again: NOTES.md, profiles/r03_arena_assembled.txt.) */
int lars_lab_malloc(void **dptr, size_t bytes, int kind, int chunk_mb, int align_mb, int shuffle);
int lars_lab_free(void *dptr);
/* Roofline probes (bench.py reports them beside the kernel numbers when the laboratory library is built): kind 0 reads
 * 16 B/lane, 1 reads 12 B/lane (the fused kernel's load shape), 2 copies 16 B/lane
 * (traffic = 2 x bytes), 3 writes 16 B/lane.  src/dst are device buffers of `bytes`.  Further kinds (csrc/lab/probe.hip):
 * 5-19 shapes of the 12 B read / 48 B written mix, 31-34 / 41-44 shared readers of one chunk, 60-64 the two passes of a
 * tile interleaved tile by tile in one launch (unroll = read-only blocks per tile, blocks = plane-writing blocks per tile,
 * bytes = ntiles x 48 MiB of source, dst = 64 tile slots x 3 planes x 64 MiB; tools/lab/twopass.py). */
int lars_d_probe(int kind, int unroll, int blocks, const void *src, void *dst, int64_t bytes, void *stream);
/* One host <-> device copy two ways (tools/pciebench.py): mode 0 hipMemcpy, 1 hipMemcpyAsync on the calling thread's (non-blocking) library
 * stream + hipStreamSynchronize -- what the host entry points do.  to_device != 0: host -> device. */
int lars_lab_copy(int mode, int to_device, void *host, void *dev, size_t bytes);
/* The fused kernel's traffic mix with the three planes WHERE THE CALLER PUTS THEM (kind 5 packs them behind one another): nquads
 * quads of 12 bytes read from src, one 16-byte vector written to each of d0, d1, d2 per quad.  Device memory comes in two kinds
 * (profiles/r04_arena_two_kinds.txt); this is the probe for a placement that splits the planes between them. */
int lars_d_probe_mix3(const void *src, void *d0, void *d1, void *d2, int64_t nquads, int blocks, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* LARS_LAB_H */
