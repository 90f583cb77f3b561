// The whole step of the plane-writing path in ONE persistent launch: channel histograms -> np.percentile -> white-balance
// tables -> fused de-interleave + white balance + NDVI/GNDVI/NDWI planes + statistics, tile after tile, so that a tile's
// second read (the fused pass) comes out of the 256 MiB Infinity Cache instead of HBM.
//
// Why: the percentile pre-pass has to see a whole tile before the first output pixel can be written, so every input byte
// is read twice (3 + 15 bytes of traffic per pixel for 15 algorithmic ones).  As two launches over a 1024-tile batch the
// second read is 48 GiB later and always misses.  Measured with the bare traffic mix (tools/mallbench.py,
// profiles/r02_mall_probe.txt): re-reading a 48 MiB range while four times as many bytes are written elsewhere moves
// 8.2-8.8 TB/s in total, against 6.4 TB/s when the range is 144 MiB or more -- the cache keeps a tile through the writes
// of its three planes, but not two or three tiles.  The work therefore has to be ordered tile by tile at a grain far
// below what launches can do (a tile's fused pass takes ~30 us).
//
// How: 256 resident workgroups of 1024 threads take work items from one atomic counter.  A tile is cut into N histogram
// items (H) and N fused items (F) of 64 wave-steps (65536 pixels) each; the queue order is
//     H_0 | H_1 F_0 | H_2 F_1 | ...        (inside a segment: `head` H items, one F item, ... then the remaining F items)
// so that tile t + 1 is being counted while tile t is being written and everything in flight belongs to two tiles.
// H item: lane-private LDS histograms (32 copies, no conflicts), folded to one partial histogram per item in global
// memory (plain stores: no atomics on the 768 bins); the workgroup that finishes a tile's last H item folds the partials,
// computes np.percentile(ch, (2, 98)) and the table exactly as k_wb_table does (fused.hip) and publishes ready[tile].
// F item: waits for ready[tile] (it can only wait for items that were handed out before it, so the wait always ends),
// then runs the plane-writing kernel's loop (fused.hip, TRAV 1: a wave owns runs of 1024 pixels, 4 KiB store bursts).
// Results are bit-identical to the two-launch path: same per-pixel arithmetic (fused_device.h), integer / fixed-point
// accumulators (order-independent).
//
// Safety: every wait is bounded (PIPE_SPIN_LIMIT polls of the flag, about two seconds) and raises the launch's abort
// word, which every workgroup checks before taking its next item: the grid always drains.  An aborted launch poisons
// the statistics records (count = 0, sums NaN) so that no caller mistakes it for a result.
#include <string.h>

#include "lab.h"
#include "../device_common.h"
#include "../fused_device.h"

namespace lars {

#ifndef PIPE_THREADS
#define PIPE_THREADS 1024                                 // one workgroup per CU (512 x 2 per CU measured slower: r02_pipeline_ab.txt)
#endif
#define PIPE_WAVES (PIPE_THREADS / 64)
#define PIPE_COPIES (PIPE_THREADS / 32)                   // histogram copies in LDS (copy = lane % PIPE_COPIES): 48 KiB at 512 threads
#define PIPE_CH_PER_PASS (PIPE_THREADS / 256)             // channels the table builder serves at a time
#define PIPE_SPIN_LIMIT 4000000ll

struct PipeParams {
    const uint8_t *tiles;          // [ntiles][npix][3]
    long long npix;
    int ntiles;
    int items;                     // N: items per tile and phase
    long long steps_per_item;      // wave-steps (256 quads = 1024 pixels) per item
    int head;                      // H items handed out before each F item while a segment still has H items
    float *out_index[3];           // [ntiles][npix]
    lars_stats *stats;             // [ntiles][3], initialised by k_stats_init, finalised by k_stats_finalize
    uint8_t *table;                // out [ntiles][768]
    double *pcts;                  // out [ntiles][3][2]
    unsigned int *hist;            // out [ntiles][768] or null
    unsigned int *partial;         // scratch [ntiles][items][768]
    unsigned int *sync;            // [0] next item, [1] abort, [2 + t] H items of tile t done, [2 + ntiles + t] table of tile t ready
    int rgn_variant;
    unsigned int flags;            // bit 29: non-temporal plane stores
    unsigned long long *trace;     // optional [blocks][PIPE_TRACE_ITEMS][6] timestamps (tools/pipebench.py --trace), or null
};
#define PIPE_TRACE_ITEMS 96

// Everything workgroups hand to each other inside the launch (partial histograms, tables, counters, flags) is written and
// read with RELAXED agent-scope atomics: those are performed at the memory side (sc1), past the per-XCD L2s, so they need
// no cache maintenance.  Agent-scope release / acquire would cost a write-back / invalidate of the whole 4 MiB L2 per
// item -- with a few hundred thousand items per launch and the L2s full of dirty plane data that was measured at 25 us
// per item.  Ordering comes from the hardware's own rules instead: a wave's stores have been acknowledged by memory when
// its s_waitcnt vmcnt(0) retires (the workgroup barrier waits for that), and only then does thread 0 bump the counter.
__device__ inline unsigned int ld_coherent(const unsigned int *p)
{
    return __hip_atomic_load(const_cast<unsigned int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_coherent(unsigned int *p, unsigned int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// abort the launch: remember why, and move the work queue past its end so that nobody takes another item
__device__ inline void pipe_abort(unsigned int *sync, unsigned int reason)
{
    st_coherent(&sync[1], reason);
    __hip_atomic_fetch_max(&sync[0], 0xF0000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------
// H item: histogram of `quads [q_lo, q_hi)` of one tile -> partial[768]
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline void pipe_hist_item(const PipeParams &P, unsigned int *s_h, int tid, long long tile, long long chunk)
{
    for (int i = tid; i < 3 * 256 * PIPE_COPIES; i += PIPE_THREADS) s_h[i] = 0;
    __syncthreads();
    const uint8_t *base = P.tiles + tile * P.npix * 3;
    const long long nquads = P.npix >> 2;
    const long long q_lo = chunk * P.steps_per_item * 256;
    long long q_hi = q_lo + P.steps_per_item * 256;
    if (q_hi > nquads) q_hi = nquads;
    char *hb = reinterpret_cast<char *>(s_h);
    const unsigned int lane_off = ((unsigned)tid & (PIPE_COPIES - 1)) << 2;
#define PADD(word, shift, ch)                                                                         \
    atomicAdd(reinterpret_cast<unsigned int *>(hb + (ch) * (256 * PIPE_COPIES * 4) + ((((word) >> (shift)) & 0xFFu) * (PIPE_COPIES * 4)) + lane_off), 1u)
#define PQUAD(a0, a1, a2)                                                                             \
    PADD(a0, 0, 0); PADD(a0, 8, 1); PADD(a0, 16, 2); PADD(a0, 24, 0);                                 \
    PADD(a1, 0, 1); PADD(a1, 8, 2); PADD(a1, 16, 0); PADD(a1, 24, 1);                                 \
    PADD(a2, 0, 2); PADD(a2, 8, 0); PADD(a2, 16, 1); PADD(a2, 24, 2);
    long long q = q_lo + tid;
    // four loads in flight per lane
    for (; q + 3 * PIPE_THREADS < q_hi; q += 4 * PIPE_THREADS) {
        unsigned int w[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int *p = reinterpret_cast<const unsigned int *>(base + (q + (long long)k * PIPE_THREADS) * 12);
            w[k][0] = p[0]; w[k][1] = p[1]; w[k][2] = p[2];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { PQUAD(w[k][0], w[k][1], w[k][2]) }
    }
    for (; q < q_hi; q += PIPE_THREADS) {
        const unsigned int *p = reinterpret_cast<const unsigned int *>(base + q * 12);
        const unsigned int a0 = p[0], a1 = p[1], a2 = p[2];
        PQUAD(a0, a1, a2)
    }
#undef PQUAD
#undef PADD
    __syncthreads();
    for (int bin = tid; bin < 768; bin += PIPE_THREADS) {
        const unsigned int *row = s_h + bin * PIPE_COPIES;
        unsigned int v = 0;
        for (int j = 0; j < PIPE_COPIES; ++j) v += row[(j + bin) & (PIPE_COPIES - 1)];
        st_coherent(&P.partial[(tile * P.items + chunk) * 768 + bin], v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Table of one tile from its N partial histograms (the workgroup that finished the tile's last H item).  The percentile
// arithmetic is k_wb_table<256>'s (fused.hip): numpy's 'linear' method, _lerp in float64, wb_level per sample value.
// Threads [256 c, 256 c + 256) serve channel c; the last 256 threads only keep the barriers company.
// ---------------------------------------------------------------------------------------------------------------------
// Returns the ready word: bit 31 set, the low bits a checksum of the table's 192 words; 0 if the partial histograms never
// added up to the tile's pixel count (the launch is then aborted).  That check is what backs the relaxed-atomics protocol:
// should a partial histogram not be visible yet when its item has been counted as done, the fold is simply repeated.
__device__ inline unsigned int pipe_build_table(const PipeParams &P, unsigned int *s_h, int tid, long long tile)
{
    unsigned int *s_tot = s_h;                                                        // [768]
    unsigned int *s_mass = s_h + 768;                                                 // [3] channel totals (+1 pad)
    unsigned long long *s_scan = reinterpret_cast<unsigned long long *>(s_h + 1024);  // [PIPE_CH_PER_PASS][256]
    double *s_val = reinterpret_cast<double *>(s_h + 1024 + 2 * 1024);               // [4][4]
    double *s_p = s_val + 16;                                                         // [4][2]
    unsigned int *s_part = s_h + 4096;                                                // [PIPE_WAVES][768]
    const long long npix = P.npix;
    for (int attempt = 0;; ++attempt) {
        {
            // fold the N partial histograms: wave w takes the items i = w (mod PIPE_WAVES), lane l the bins l, l + 64, ... (12);
            // 24 independent loads per trip (a one-load-per-trip loop costs a memory round trip per item: 300 us per tile)
            const int wave = tid >> 6, lane = tid & 63;
            unsigned int accb[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) accb[j] = 0;
            const unsigned int *src = P.partial + tile * P.items * 768 + lane;
            int i = wave;
            for (; i + PIPE_WAVES < P.items; i += 2 * PIPE_WAVES) {
                unsigned int a[12], b[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) { a[j] = ld_coherent(&src[(long long)i * 768 + 64 * j]); b[j] = ld_coherent(&src[(long long)(i + PIPE_WAVES) * 768 + 64 * j]); }
#pragma unroll
                for (int j = 0; j < 12; ++j) accb[j] += a[j] + b[j];
            }
            for (; i < P.items; i += PIPE_WAVES) {
#pragma unroll
                for (int j = 0; j < 12; ++j) accb[j] += ld_coherent(&src[(long long)i * 768 + 64 * j]);
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) s_part[wave * 768 + 64 * j + lane] = accb[j];
            if (tid < 4) s_mass[tid] = 0;
        }
        __syncthreads();
        for (int bin = tid; bin < 768; bin += PIPE_THREADS) {
            unsigned int v = 0;
#pragma unroll
            for (int w = 0; w < PIPE_WAVES; ++w) v += s_part[w * 768 + bin];
            s_tot[bin] = v;
            atomicAdd(&s_mass[bin >> 8], v);
        }
        __syncthreads();
        const bool whole = s_mass[0] == (unsigned long long)npix && s_mass[1] == (unsigned long long)npix && s_mass[2] == (unsigned long long)npix;
        if (whole) break;                                       // uniform: everybody reads the same three words
        if (attempt >= 2000) return 0u;
        __builtin_amdgcn_s_sleep(64);
        __syncthreads();
    }
    for (int bin = tid; bin < 768 && P.hist; bin += PIPE_THREADS) P.hist[tile * 768 + bin] = s_tot[bin];
    const double nm1 = (double)(npix - 1);
    double tq[2];
    long long rank[4];
    for (int k = 0; k < 2; ++k) {
        const double qq = (k == 0 ? 2.0 : 98.0) / 100.0;
        const double vi = nm1 * qq;
        const double fl = floor(vi);
        const long long lo = (long long)fl;
        long long hi = lo + 1;
        if (hi > npix - 1) hi = npix - 1;
        rank[2 * k] = lo;
        rank[2 * k + 1] = hi;
        tq[k] = vi - fl;
    }
    uint8_t *s_tab = reinterpret_cast<uint8_t *>(s_part);                // the wave rows are no longer needed
    const int slot = tid >> 8, lt = tid & 255;                          // slot: which of the channels of this pass
    for (int c0 = 0; c0 < 3; c0 += PIPE_CH_PER_PASS) {
        const int ch = c0 + slot;
        const bool live = ch < 3;
        const unsigned long long local = live ? s_tot[ch * 256 + lt] : 0ull;
        __syncthreads();
        s_scan[slot * 256 + lt] = local;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const unsigned long long v = (lt >= off) ? s_scan[slot * 256 + lt - off] : 0ull;
            __syncthreads();
            s_scan[slot * 256 + lt] += v;
            __syncthreads();
        }
        const unsigned long long before = s_scan[slot * 256 + lt] - local;
        if (live && local) {
            for (int r = 0; r < 4; ++r)
                if ((unsigned long long)rank[r] >= before && (unsigned long long)rank[r] < before + local) s_val[ch * 4 + r] = (double)lt;
        }
        __syncthreads();
        if (live && lt < 2) {
            const double a = s_val[ch * 4 + 2 * lt], b = s_val[ch * 4 + 2 * lt + 1], t = tq[lt];
            const double d = b - a;
            double r = a + d * t;
            if (t >= 0.5) r = b - d * (1.0 - t);
            s_p[ch * 2 + lt] = r;
            P.pcts[tile * 6 + ch * 2 + lt] = r;
        }
        __syncthreads();
        if (live) s_tab[ch * 256 + lt] = (uint8_t)wb_level(lt, s_p[ch * 2], s_p[ch * 2 + 1], P.rgn_variant);
    }
    __syncthreads();
    unsigned int word = 0;
    if (tid < 192) {
        word = reinterpret_cast<const unsigned int *>(s_tab)[tid];
        st_coherent(reinterpret_cast<unsigned int *>(P.table + tile * 768) + tid, word);
    }
    // checksum of the 192 words (position-weighted), folded over the first three waves
    unsigned int sum = word * (2u * (unsigned)tid + 1u);
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((tid & 63) == 0 && tid < 192) s_mass[tid >> 6] = sum;
    __syncthreads();
    return 0x80000000u | ((s_mass[0] + s_mass[1] + s_mass[2]) & 0x7FFFFFFFu);
}

// ---------------------------------------------------------------------------------------------------------------------
// F item: wave-steps [chunk * spi, (chunk + 1) * spi) of one tile through the plane-writing loop
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline void pipe_fused_item(const PipeParams &P, const uint8_t *s_lut, double (*s_red)[16], int tid, long long tile,
                                       long long chunk)
{
    const long long npix = P.npix;
    // flags bit 0 (lars_set_tuning("pipe_cold", 1), timing experiments only: wrong results): read the samples of a tile half
    // a launch away, which cannot be in any cache -- what the second read would cost without the Infinity Cache
    const long long src_tile = (P.flags & 1u) ? (tile + P.ntiles / 2) % P.ntiles : tile;
    const uint8_t *base = P.tiles + src_tile * npix * 3;
    float *const oi0 = P.out_index[0] + tile * npix;
    float *const oi1 = P.out_index[1] + tile * npix;
    float *const oi2 = P.out_index[2] + tile * npix;
    const bool nt_st = (P.flags & 0x20000000u) != 0;
    const long long nquads = npix >> 2;
    const long long nsteps = (nquads + 255) >> 8;
    Acc acc[3];
    acc_init(acc[0]); acc_init(acc[1]); acc_init(acc[2]);

    auto do_quad = [&](long long q, unsigned int w0, unsigned int w1, unsigned int w2) {
        unsigned int b[12] = {w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, w0 >> 24,
                              w1 & 0xFF, (w1 >> 8) & 0xFF, (w1 >> 16) & 0xFF, w1 >> 24,
                              w2 & 0xFF, (w2 >> 8) & 0xFF, (w2 >> 16) & 0xFF, w2 >> 24};
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = s_lut[(i % 3) * 256 + b[i]];
        float v0[4], v1[4], v2[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            v0[px] = v1[px] = v2[px] = 0.0f;
            pixel_math<7u, 1, false, true>((float)b[3 * px], (float)b[3 * px + 1], (float)b[3 * px + 2], 0u, v0[px], v1[px], v2[px], acc,
                                           nullptr, nullptr);
        }
        store_plane4(oi0 + q * 4, v0, nt_st);
        store_plane4(oi1 + q * 4, v1, nt_st);
        store_plane4(oi2 + q * 4, v2, nt_st);
    };

    const unsigned int lane = (unsigned)tid & 63u;
    long long st_hi = (chunk + 1) * P.steps_per_item;
    if (st_hi > nsteps) st_hi = nsteps;
    for (long long st = chunk * P.steps_per_item + (tid >> 6); st < st_hi; st += PIPE_WAVES) {
        const long long q0 = st * 256 + lane;
        if (st * 256 + 256 <= nquads) {
            unsigned int w[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int *p = reinterpret_cast<const unsigned int *>(base + (q0 + 64 * j) * 12);
                w[j][0] = p[0]; w[j][1] = p[1]; w[j][2] = p[2];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                do_quad(q0 + 64 * j, w[j][0], w[j][1], w[j][2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int j = 0; j < 4; ++j) {
                const long long q = q0 + 64 * j;
                if (q < nquads) {
                    const unsigned int *p = reinterpret_cast<const unsigned int *>(base + q * 12);
                    do_quad(q, p[0], p[1], p[2]);
                }
            }
        }
    }

    // flush: wave fold -> one row per wave -> thread 0 -> the tile's records (integer atomics: order-independent)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Acc &a = acc[k];
        for (int off = 32; off >= 1; off >>= 1) {
            a.mn = fminf(a.mn, __shfl_xor(a.mn, off));
            a.mx = fmaxf(a.mx, __shfl_xor(a.mx, off));
            a.sum += __shfl_xor(a.sum, off);
            a.above += __shfl_xor(a.above, off);
        }
    }
    const int wave = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s_red[wave][4 * k] = acc[k].sum;
            s_red[wave][4 * k + 1] = (double)acc[k].above;
            s_red[wave][4 * k + 2] = (double)acc[k].mn;
            s_red[wave][4 * k + 3] = (double)acc[k].mx;
        }
    }
    __syncthreads();
    if (tid < 3) {
        double sum = 0, above = 0, mn = __builtin_inf(), mx = -__builtin_inf();
        for (int w = 0; w < PIPE_WAVES; ++w) {
            sum += s_red[w][4 * tid];
            above += s_red[w][4 * tid + 1];
            mn = fmin(mn, s_red[w][4 * tid + 2]);
            mx = fmax(mx, s_red[w][4 * tid + 3]);
        }
        if (above > 0.0 || mn <= mx) {                          // an item without a pixel (cannot happen) adds nothing
            StatsAccView *rec = reinterpret_cast<StatsAccView *>(P.stats + tile * 3) + tid;
            atomicAdd(&rec->sum_fx, (unsigned long long)__double2ll_rn(sum * LARS_FX_SCALE));
            atomicAdd(&rec->above, (unsigned long long)above);
            atomicMin(&rec->min_key, f64_key(mn));
            atomicMax(&rec->max_key, f64_key(mx));
        }
    }
}

__global__ __launch_bounds__(PIPE_THREADS, 4) void k_pipe_u8c3(PipeParams P)
{
    __shared__ __attribute__((aligned(16))) unsigned int s_h[3 * 256 * PIPE_COPIES];         // 48 KiB at 512 threads
    __shared__ __attribute__((aligned(16))) uint8_t s_lut[768];
    __shared__ double s_red[PIPE_WAVES][16];
    __shared__ unsigned int s_ctl[4];

    const int tid = threadIdx.x;
    const unsigned int N = (unsigned)P.items;
    const unsigned int T = (unsigned)P.ntiles;
    const unsigned long long total = (unsigned long long)N + 2ull * N * T;
    const unsigned int h = (unsigned)P.head;
    const unsigned int groups = N / h;                          // groups of (h H items + 1 F item) at the head of a segment
    unsigned int *const done = P.sync + 2;
    unsigned int *const ready = P.sync + 2 + T;

    // the item after the one in hand is always on its way (the queue is a memory-side atomic: a round trip of a microsecond
    // or two that would otherwise sit in front of every item); an abort moves the queue past its end
    unsigned int fetched = 0;
    int ntraced = 0;
    if (tid == 0) fetched = __hip_atomic_fetch_add(&P.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (tid == 0) s_ctl[0] = fetched;
        __syncthreads();
        const unsigned long long item = s_ctl[0];
        if (item >= total) break;                               // uniform over the workgroup
        if (tid == 0) fetched = __hip_atomic_fetch_add(&P.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};
        const bool tracing = P.trace != nullptr && tid == 0 && ntraced < PIPE_TRACE_ITEMS;
        if (tracing) tr[0] = wall_clock64();
        // decode: phase (0 = H, 1 = F), tile, chunk
        unsigned int phase, tile, chunk;
        if (item < N) {
            phase = 0; tile = 0; chunk = (unsigned)item;
        } else {
            const unsigned long long i2 = item - N;
            const unsigned int seg = (unsigned)(i2 / (2ull * N)), j = (unsigned)(i2 % (2ull * N));
            if (j < groups * (h + 1)) {
                const unsigned int g = j / (h + 1), r = j % (h + 1);
                if (r < h) { phase = 0; tile = seg + 1; chunk = g * h + r; }
                else { phase = 1; tile = seg; chunk = g; }
            } else {
                const unsigned int r = j - groups * (h + 1);    // what is left: H items beyond groups * h (none when h divides N), then F
                const unsigned int h_left = N - groups * h;
                if (r < h_left) { phase = 0; tile = seg + 1; chunk = groups * h + r; }
                else { phase = 1; tile = seg; chunk = groups + (r - h_left); }
            }
        }
        if (phase == 0) {
            if (tile < T) {
                pipe_hist_item(P, s_h, tid, tile, chunk);
                if (tracing) tr[1] = tr[2] = wall_clock64();
                // every wave's partial-histogram stores have been acknowledged by memory when it passes the barrier
                // (s_waitcnt vmcnt(0) in front of it); only then does thread 0 count the item as done
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (tid == 0) {
                    const unsigned int prev = __hip_atomic_fetch_add(&done[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_ctl[2] = prev == N - 1u;
                }
                __syncthreads();
                if (tracing) tr[3] = wall_clock64();
                if (s_ctl[2]) {
                    const unsigned int word = pipe_build_table(P, s_h, tid, tile);     // reads the partials past the L2 (ld_coherent)
                    __builtin_amdgcn_s_waitcnt(0);
                    __syncthreads();
                    if (tid == 0) {
                        if (word) st_coherent(&ready[tile], word);
                        else pipe_abort(P.sync, 2u);            // histogram mass never reached npix
                    }
                }
            }
        } else {
            // the table and its ready word in one round trip; repeated (bounded) until the word is there and the table's
            // checksum matches it
            bool good = false;
            for (long long attempt = 0; !good; ++attempt) {
                unsigned int word = 0;
                if (tid < 192) {
                    word = ld_coherent(reinterpret_cast<const unsigned int *>(P.table + (unsigned long long)tile * 768) + tid);
                    reinterpret_cast<unsigned int *>(s_lut)[tid] = word;
                }
                if (tid == 192) {
                    // while the table is not there only this one thread polls, one word per visit (a whole workgroup
                    // re-reading 192 words per visit starves the table builder's own reads)
                    unsigned int rdy, gone = 0u;
                    long long spins = 0;
                    while (!(rdy = ld_coherent(&ready[tile]))) {
                        if ((++spins & 63) == 0 && (gone = ld_coherent(&P.sync[1])) != 0u) break;
                        if (spins > PIPE_SPIN_LIMIT) { pipe_abort(P.sync, 1u); gone = 1u; break; }
                        __builtin_amdgcn_s_sleep(32);
                    }
                    s_ctl[3] = rdy;
                    s_ctl[1] = gone | (attempt > 4000 ? 1u : 0u);
                    if (attempt > 4000) pipe_abort(P.sync, 3u);   // the table never matched its checksum
                }
                unsigned int sum = word * (2u * (unsigned)tid + 1u);
                for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
                if ((tid & 63) == 0 && tid < 192) s_red[0][tid >> 6] = (double)sum;
                __syncthreads();
                const unsigned int total_sum = (unsigned int)s_red[0][0] + (unsigned int)s_red[0][1] + (unsigned int)s_red[0][2];
                good = s_ctl[3] != 0u && (0x80000000u | (total_sum & 0x7FFFFFFFu)) == s_ctl[3];      // uniform
                const bool aborted = s_ctl[1] != 0u;
                __syncthreads();
                if (good) break;
                if (aborted) break;
                __builtin_amdgcn_s_sleep(16);
            }
            if (!good) break;                                   // aborted: uniform exit
            if (tracing) tr[1] = wall_clock64();
            pipe_fused_item(P, s_lut, s_red, tid, tile, chunk);
            if (tracing) tr[2] = tr[3] = wall_clock64();
        }
        __syncthreads();                                        // s_ctl, s_lut, s_red and s_h are free again
        if (tracing) {
            tr[4] = wall_clock64();
            tr[5] = (unsigned long long)phase | ((unsigned long long)tile << 8) | ((unsigned long long)chunk << 32);
            unsigned long long *dst = P.trace + ((unsigned long long)blockIdx.x * PIPE_TRACE_ITEMS + ntraced) * 6;
            for (int j = 0; j < 6; ++j) dst[j] = tr[j];
            ++ntraced;
        }
    }
}

// an aborted launch must not look like a result: zero counts, NaN sums
__global__ void k_pipe_poison(const unsigned int *sync, lars_stats *stats, long long nrec)
{
    if (!sync[1]) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrec) {
        stats[i].count = 0;
        stats[i].sum = __builtin_nan("");
        stats[i].min = __builtin_nan("");
        stats[i].max = __builtin_nan("");
    }
}

}  // namespace lars

using namespace lars;

extern "C" size_t lars_pipeline_scratch_bytes(int64_t ntiles, int64_t npix)
{
    if (ntiles <= 0 || npix <= 0) return 0;
    const long long nsteps = ((npix >> 2) + 255) >> 8;
    long long spi = lab_tuning().pipe_steps > 0 ? lab_tuning().pipe_steps : 64;
    const long long items = (nsteps + spi - 1) / spi;
    return (size_t)ntiles * (size_t)items * 768 * 4 + (size_t)(2 + 2 * ntiles) * 4 + 512 + (size_t)4096 * PIPE_TRACE_ITEMS * 6 * 8;
}

// Histograms, percentile tables and the fused pass of `ntiles` uint8 RGNir tiles in one persistent launch.
// a->wb_table is the OUTPUT table buffer here ([ntiles][768]); percentiles [ntiles][3][2]; hist [ntiles][768] or NULL.
extern "C" int lars_d_pipeline(const lars_fused_args *a, double *percentiles, uint32_t *hist, int rgn_variant, void *scratch)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!a || !a->tiles || !a->wb_table || !a->stats || !percentiles || !scratch || a->ntiles <= 0 || a->npix <= 0)
        return fail(LARS_ERR_INVALID, "lars_d_pipeline: bad arguments");
    if (a->dtype != LARS_U8 || a->channels != 3 || (a->npix & 3) || (reinterpret_cast<uintptr_t>(a->tiles) & 3))
        return fail(LARS_ERR_INVALID, "lars_d_pipeline: uint8 [ntiles][npix][3] tiles, npix a multiple of 4, on 4-byte boundaries");
    if ((a->index_mask & LARS_MASK_ALL) != LARS_MASK_ALL || !a->out_index[0] || !a->out_index[1] || !a->out_index[2] || a->out_wb ||
        a->out_rgba[0] || a->out_rgba[1] || a->out_rgba[2] || (a->flags & (LARS_F_HIST | LARS_F_SUMSQ)))
        return fail(LARS_ERR_INVALID, "lars_d_pipeline: serves all three index planes + basic statistics (use lars_d_fused otherwise)");
    for (int k = 0; k < 3; ++k)
        if (reinterpret_cast<uintptr_t>(a->out_index[k]) & 15) return fail(LARS_ERR_INVALID, "lars_d_pipeline: planes on 16-byte boundaries");
    if (a->ntiles > 65535 || (long long)a->npix * 3 >= (1ll << 31)) return fail(LARS_ERR_INVALID, "lars_d_pipeline: at most 65535 tiles of < 2^31 / 3 pixels");
    hipStream_t s = pick_stream(c, a->stream);

    PipeParams P;
    memset(&P, 0, sizeof P);
    P.tiles = static_cast<const uint8_t *>(a->tiles);
    P.npix = a->npix;
    P.ntiles = (int)a->ntiles;
    const long long nsteps = ((a->npix >> 2) + 255) >> 8;
    P.steps_per_item = lab_tuning().pipe_steps > 0 ? lab_tuning().pipe_steps : 64;
    P.items = (int)((nsteps + P.steps_per_item - 1) / P.steps_per_item);
    P.head = lab_tuning().pipe_head > 0 ? lab_tuning().pipe_head : 2;
    if (P.head > P.items) P.head = P.items;
    for (int k = 0; k < 3; ++k) P.out_index[k] = a->out_index[k];
    P.stats = a->stats;
    P.table = const_cast<uint8_t *>(a->wb_table);
    P.pcts = percentiles;
    P.hist = hist;
    P.partial = static_cast<unsigned int *>(scratch);
    P.sync = P.partial + (size_t)a->ntiles * P.items * 768;
    P.rgn_variant = rgn_variant;
    P.flags = (tuning().nt_stores ? 0x20000000u : 0u) | (lab_tuning().pipe_cold ? 1u : 0u);
    // item timestamps for tools/pipebench.py: behind the sync words, 8-byte aligned, room for 4096 workgroups
    P.trace = lab_tuning().pipe_trace ? reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(P.sync + 2 + 2 * a->ntiles) + 255) & ~(uintptr_t)255) : nullptr;
    if (P.trace) LARS_HIP_TRY(hipMemsetAsync(P.trace, 0, (size_t)4096 * PIPE_TRACE_ITEMS * 6 * 8, s));

    LARS_HIP_TRY(hipMemsetAsync(P.sync, 0, (size_t)(2 + 2 * a->ntiles) * 4, s));
    const long long nrec = a->ntiles * 3;
    stats_init_launch(a->stats, nrec, 7u, s);
    // every workgroup must be resident (the waits assume running producers): one per CU fits by construction
    // (96 KiB of LDS, 1024 threads, <= 128 VGPRs), the occupancy query says how many more
    int per_cu = 0;
    LARS_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pipe_u8c3, PIPE_THREADS, 0));
    if (per_cu < 1) return fail(LARS_ERR_HIP, "lars_d_pipeline: the pipeline kernel does not fit a compute unit");
    hipDeviceProp_t prop;
    LARS_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    long long blocks = (long long)prop.multiProcessorCount * per_cu;
    const long long total = (long long)P.items * (1 + 2ll * a->ntiles);
    if (blocks > total) blocks = total;
    hipLaunchKernelGGL(k_pipe_u8c3, dim3((unsigned)blocks), dim3(PIPE_THREADS), 0, s, P);
    stats_finalize_launch(a->stats, nrec, 7u, (long long)a->npix, s);
    hipLaunchKernelGGL(k_pipe_poison, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, P.sync, a->stats, nrec);
    return launch_check("lars_d_pipeline");
}
