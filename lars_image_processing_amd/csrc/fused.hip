// Hot path kernels: channel histograms -> white-balance tables -> fused
// de-interleave + white balance + NDVI/GNDVI/NDWI + outputs + statistics.
//
// Reference semantics (lars-uav/lars-image-processing):
//   process-images.py:424-447  fix_white_balance   (percentile stretch, uint8 out)
//   process-images.py:449-490  calculate_index     (float32 normalized differences)
//   process-images.py:492-513  analyze_index       (mean/min/max/coverage)
//   process-ndvi.py:97         50-bin histogram on [-1, 1]
//   process-images.py:695      imshow(cmap, vmin=-1, vmax=1) per-pixel colormap
//
// Built with -ffp-contract=off: every float operation below rounds exactly once,
// as the NumPy expressions do.
#include <string.h>

#include "common.h"
#include "device_common.h"
#include "fused_device.h"

namespace lars {

// ===========================================================================
// Per-tile channel histograms (pre-pass of np.percentile, process-images.py:437)
// ===========================================================================
// uint8, 3 interleaved channels, 4-byte aligned tiles: each lane reads 12 bytes
// (4 pixels) per step, a wave reads 768 contiguous bytes.
__global__ __launch_bounds__(256) void k_chan_hist_u8c3(const uint8_t *__restrict__ tiles,
                                                        long long npix, unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_h[4][3 * 256];          // one private copy per wave
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    for (int i = tid; i < 4 * 768; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();

    const long long tile = blockIdx.y;
    const uint8_t *base = tiles + tile * npix * 3;
    const long long nquads = npix >> 2;
    unsigned int *h = s_h[wave];
    for (long long q = (long long)blockIdx.x * 256 + tid; q < nquads; q += (long long)gridDim.x * 256) {
        const unsigned int *p = reinterpret_cast<const unsigned int *>(base + q * 12);
        unsigned int w0 = p[0], w1 = p[1], w2 = p[2];
        // bytes: r0 g0 n0 r1 | g1 n1 r2 g2 | n2 r3 g3 n3
        atomicAdd(&h[0 * 256 + (w0 & 0xFF)], 1u);
        atomicAdd(&h[1 * 256 + ((w0 >> 8) & 0xFF)], 1u);
        atomicAdd(&h[2 * 256 + ((w0 >> 16) & 0xFF)], 1u);
        atomicAdd(&h[0 * 256 + (w0 >> 24)], 1u);
        atomicAdd(&h[1 * 256 + (w1 & 0xFF)], 1u);
        atomicAdd(&h[2 * 256 + ((w1 >> 8) & 0xFF)], 1u);
        atomicAdd(&h[0 * 256 + ((w1 >> 16) & 0xFF)], 1u);
        atomicAdd(&h[1 * 256 + (w1 >> 24)], 1u);
        atomicAdd(&h[2 * 256 + (w2 & 0xFF)], 1u);
        atomicAdd(&h[0 * 256 + ((w2 >> 8) & 0xFF)], 1u);
        atomicAdd(&h[1 * 256 + ((w2 >> 16) & 0xFF)], 1u);
        atomicAdd(&h[2 * 256 + (w2 >> 24)], 1u);
    }
    // tail pixels (npix % 4) by block 0
    if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
        const uint8_t *p = base + (nquads * 4 + tid) * 3;
        atomicAdd(&h[p[0]], 1u);
        atomicAdd(&h[256 + p[1]], 1u);
        atomicAdd(&h[512 + p[2]], 1u);
    }
    __syncthreads();
    unsigned int *gh = hist + tile * 768;
    for (int i = tid; i < 768; i += 256) {
        unsigned int v = s_h[0][i] + s_h[1][i] + s_h[2][i] + s_h[3][i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// Any channel count / sample type / alignment: one pixel per lane per step.
template <typename PIX, int NVAL>
__global__ __launch_bounds__(256) void k_chan_hist_generic(const PIX *__restrict__ tiles, long long npix,
                                                           int channels, unsigned int *__restrict__ hist)
{
    const long long tile = blockIdx.y;
    const PIX *base = tiles + tile * npix * channels;
    unsigned int *gh = hist + tile * 3 * (long long)NVAL;
    if (NVAL == 256) {
        __shared__ unsigned int s_h[3 * 256];
        for (int i = threadIdx.x; i < 768; i += 256) s_h[i] = 0;
        __syncthreads();
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
            const PIX *p = base + i * channels;
            atomicAdd(&s_h[(unsigned)p[0]], 1u);
            atomicAdd(&s_h[256 + (unsigned)p[1]], 1u);
            atomicAdd(&s_h[512 + (unsigned)p[2]], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 768; i += 256)
            if (s_h[i]) atomicAdd(&gh[i], s_h[i]);
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
            const PIX *p = base + i * channels;
            atomicAdd(&gh[(unsigned)p[0]], 1u);
            atomicAdd(&gh[NVAL + (unsigned)p[1]], 1u);
            atomicAdd(&gh[2 * NVAL + (unsigned)p[2]], 1u);
        }
    }
}

// ===========================================================================
// Histogram -> np.percentile(ch, (2, 98)) -> white-balance table
// ===========================================================================
// One block per (channel, tile).  numpy's 'linear' method: virtual index
// (n-1)*q in float64, order statistics floor(vi) and floor(vi)+1 (clamped),
// _lerp: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5.
template <int NVAL>
__global__ __launch_bounds__(256) void k_wb_table(const unsigned int *__restrict__ hist, long long npix,
                                                  uint8_t *__restrict__ table, double *__restrict__ pcts,
                                                  int rgn_variant)
{
    constexpr int PER = NVAL / 256;                   // bins per thread
    __shared__ unsigned long long s_scan[256];
    __shared__ double s_val[4];                       // order statistics: q2.lo q2.hi q98.lo q98.hi
    __shared__ double s_p[2];
    __shared__ unsigned int s_thr[NVAL == 256 ? 4 : 260];

    const int tid = threadIdx.x;
    const long long slot = (long long)blockIdx.y * 3 + blockIdx.x;   // tile*3 + channel
    const unsigned int *h = hist + slot * NVAL;

    unsigned long long local = 0;
    for (int j = 0; j < PER; ++j) local += h[tid * PER + j];
    s_scan[tid] = local;
    __syncthreads();
    // inclusive Hillis-Steele scan over 256 thread sums
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const unsigned long long before = s_scan[tid] - local;   // samples in bins below this thread's

    const double nm1 = (double)(npix - 1);
    double tq[2];
    long long rank[4];
    for (int k = 0; k < 2; ++k) {
        const double q = (k == 0 ? 2.0 : 98.0) / 100.0;
        const double vi = nm1 * q;
        const double fl = floor(vi);
        long long lo = (long long)fl;
        long long hi = lo + 1;
        if (hi > npix - 1) hi = npix - 1;
        rank[2 * k] = lo;
        rank[2 * k + 1] = hi;
        tq[k] = vi - fl;
    }
    // value of order statistic r = the bin whose cumulative count first exceeds r
    unsigned long long cum = before;
    for (int j = 0; j < PER; ++j) {
        const unsigned long long c = h[tid * PER + j];
        if (c) {
            for (int r = 0; r < 4; ++r)
                if ((unsigned long long)rank[r] >= cum && (unsigned long long)rank[r] < cum + c)
                    s_val[r] = (double)(tid * PER + j);
        }
        cum += c;
    }
    __syncthreads();
    if (tid < 2) {
        const double a = s_val[2 * tid], b = s_val[2 * tid + 1], t = tq[tid];
        const double d = b - a;
        double r = a + d * t;
        if (t >= 0.5) r = b - d * (1.0 - t);
        s_p[tid] = r;
        if (pcts) pcts[slot * 2 + tid] = r;
    }
    __syncthreads();
    const double p_lo = s_p[0], p_hi = s_p[1];
    if (NVAL == 256) {
        uint8_t *out = table + slot * 256;
        out[tid] = (uint8_t)wb_level(tid, p_lo, p_hi, rgn_variant);
    } else {
        // uint16: the table lives in the tile's blob together with its threshold form
        u16_fill_blob(table + (long long)blockIdx.y * LARS_U16_BLOB_BYTES, (int)blockIdx.x, p_lo, p_hi, rgn_variant, s_thr, tid);
    }
}

// ===========================================================================
// Statistics records: init / finalize around the fused launch
// ===========================================================================
__global__ void k_stats_init(lars_stats *stats, long long nrec, unsigned int mask)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    const int k = (int)(i % 3);
    if (!((mask >> k) & 1u)) return;
    StatsAccView *a = reinterpret_cast<StatsAccView *>(stats + i);
    a->sum_fx = 0; a->sumsq_fx = 0; a->count = 0; a->above = 0; a->nans = 0;
    a->min_key = ~0ull; a->max_key = 0ull;
    a->threshold = (k == LARS_NDWI) ? 0.0 : (double)0.2f;        // process-images.py:498-502 (float32 compare)
    a->index_id = (unsigned)k; a->reserved = 0;
    for (int b = 0; b < LARS_HIST_BINS; ++b) a->hist[b] = 0;
}

__global__ void k_stats_finalize(lars_stats *stats, long long nrec, unsigned int mask, long long npix)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    const int k = (int)(i % 3);
    if (!((mask >> k) & 1u)) return;
    StatsAccView *a = reinterpret_cast<StatsAccView *>(stats + i);
    const long long s = (long long)a->sum_fx, s2 = (long long)a->sumsq_fx;
    const unsigned long long mnk = a->min_key, mxk = a->max_key;
    lars_stats *o = stats + i;
    o->sum = (double)s * LARS_FX_INV;
    o->sumsq = (double)s2 * LARS_FX_INV;
    o->count = (uint64_t)npix;
    o->min = key_f64(mnk);
    o->max = key_f64(mxk);
}

// Fold of the per-tile records of one index into one record, on the device, with the arithmetic of lars_stats_merge
// (host_core.cpp): the float64 sums are added in tile order by one thread (deterministic), integer fields add, extrema
// fold.  One workgroup per index; records come in final form (after k_stats_finalize / lars_d_stats_joint).
__global__ __launch_bounds__(256) void k_stats_fold(const lars_stats *__restrict__ rec, long long ntiles, unsigned int mask,
                                                    lars_stats *__restrict__ out)
{
    const int k = blockIdx.x;
    if (!((mask >> k) & 1u)) return;
    __shared__ double s_sum[256], s_sq[256];
    __shared__ double s_mn[4], s_mx[4];
    __shared__ unsigned long long s_cnt[4][3];
    const int tid = threadIdx.x;
    double sum = 0.0, sq = 0.0;                                       // thread 0 only
    double mn = __builtin_inf(), mx = -__builtin_inf();
    unsigned long long cnt = 0, above = 0, nans = 0;
    // the 50 bins: every thread walks the records of its own tiles (tid, tid + 256, ...) and adds what is not zero to the block's bins --
    // all loads of a tile independent of each other.  (Until round 5 lane b of each wave walked every fourth tile's bin b, one dependent
    // 8-byte load after the other: 146 us per 1024 tiles, 1.5 % of a statistics-only step, for counts that are zero unless LARS_F_HIST.)
    __shared__ unsigned long long s_hist[LARS_HIST_BINS];
    if (tid < LARS_HIST_BINS) s_hist[tid] = 0ull;
    __syncthreads();
    for (long long t = tid; t < ntiles; t += 256) {
        const unsigned long long *h = reinterpret_cast<const unsigned long long *>(rec[t * 3 + k].hist);
        unsigned long long v[LARS_HIST_BINS];
#pragma unroll
        for (int b = 0; b < LARS_HIST_BINS; ++b) v[b] = h[b];
#pragma unroll
        for (int b = 0; b < LARS_HIST_BINS; ++b)
            if (v[b]) atomicAdd(&s_hist[b], v[b]);
    }
    for (long long base = 0; base < ntiles; base += 256) {
        const long long t = base + tid;
        if (t < ntiles) {
            const lars_stats *r = rec + t * 3 + k;
            s_sum[tid] = r->sum; s_sq[tid] = r->sumsq;
            cnt += r->count; above += r->above; nans += r->nans;
            mn = fmin(mn, r->min); mx = fmax(mx, r->max);
        }
        __syncthreads();
        const int n = (int)(ntiles - base < 256 ? ntiles - base : 256);
        if (tid == 0) {
            // the sums in tile order, one add after the other (lars_stats_merge's order); the operands come out of the LDS eight at a time
            int i = 0;
            if (base == 0) { sum = s_sum[0]; sq = s_sq[0]; i = 1; }
            for (; i < n && (i & 7); ++i) { sum += s_sum[i]; sq += s_sq[i]; }
            for (; i + 8 <= n; i += 8) {
                double a[8], b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { a[j] = s_sum[i + j]; b[j] = s_sq[i + j]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) { sum += a[j]; sq += b[j]; }
            }
            for (; i < n; ++i) { sum += s_sum[i]; sq += s_sq[i]; }
        }
        __syncthreads();
    }
    for (int off = 32; off >= 1; off >>= 1) {
        cnt += __shfl_xor(cnt, off); above += __shfl_xor(above, off); nans += __shfl_xor(nans, off);
        mn = fmin(mn, __shfl_xor(mn, off)); mx = fmax(mx, __shfl_xor(mx, off));
    }
    if ((tid & 63) == 0) {
        const int w = tid >> 6;
        s_cnt[w][0] = cnt; s_cnt[w][1] = above; s_cnt[w][2] = nans; s_mn[w] = mn; s_mx[w] = mx;
    }
    __syncthreads();
    lars_stats *o = out + k;
    if (tid == 0) {
        o->sum = sum; o->sumsq = sq;
        o->count = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
        o->above = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
        o->nans = s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2];
        o->min = fmin(fmin(s_mn[0], s_mn[1]), fmin(s_mn[2], s_mn[3]));
        o->max = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3]));
        o->threshold = rec[k].threshold; o->index_id = rec[k].index_id; o->reserved = rec[k].reserved;
    }
    if (tid < LARS_HIST_BINS) o->hist[tid] = s_hist[tid];
}

// ===========================================================================
// The fused kernel
// ===========================================================================
// Block-wide fold of one accumulator into the tile's record (atomics are
// integer, so the result does not depend on arrival order).
__device__ inline void acc_flush(Acc a, StatsAccView *rec, double (*s_red)[4], int tid)
{
    // wave reduce
    for (int off = 32; off >= 1; off >>= 1) {
        a.mn = fminf(a.mn, __shfl_xor(a.mn, off));
        a.mx = fmaxf(a.mx, __shfl_xor(a.mx, off));
        a.sum += __shfl_xor(a.sum, off);
        a.sumsq += __shfl_xor(a.sumsq, off);
        a.above += __shfl_xor(a.above, off);
    }
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();
    if (lane == 0) {
        s_red[0][wave] = a.sum;
        s_red[1][wave] = a.sumsq;
        s_red[2][wave] = (double)a.above;
        s_red[3][wave] = __builtin_bit_cast(double, ((unsigned long long)f32_key(a.mx) << 32) | f32_key(a.mn));   // packed float keys
    }
    __syncthreads();
    if (tid == 0) {
        double sum = 0, sumsq = 0, above = 0;
        unsigned int mnk = 0xFFFFFFFFu, mxk = 0;
        for (int w = 0; w < 4; ++w) {
            sum += s_red[0][w];
            sumsq += s_red[1][w];
            above += s_red[2][w];
            const unsigned long long kk = __builtin_bit_cast(unsigned long long, s_red[3][w]);
            const unsigned int kmn = (unsigned int)kk, kmx = (unsigned int)(kk >> 32);
            mnk = kmn < mnk ? kmn : mnk;
            mxk = kmx > mxk ? kmx : mxk;
        }
        atomicAdd(&rec->sum_fx, (unsigned long long)__double2ll_rn(sum * LARS_FX_SCALE));
        atomicAdd(&rec->sumsq_fx, (unsigned long long)__double2ll_rn(sumsq * LARS_FX_SCALE));
        atomicAdd(&rec->above, (unsigned long long)above);
        atomicMin(&rec->min_key, f64_key((double)key_f32(mnk)));
        atomicMax(&rec->max_key, f64_key((double)key_f32(mxk)));
    }
}

// ---- fast path: uint8, 3 channels, 4-byte aligned tiles ---------------------
// Lane l of a wave owns pixels 4l..4l+3 of each 256-pixel slab: one 12-byte
// load (a wave reads 768 contiguous bytes), one float4 store per index plane
// (1 KiB contiguous per wave), one 12-byte store of the white-balanced image,
// one 16-byte store per colormapped plane.
//
// The same kernel serves uint16 tiles (PIX = uint16_t): 24 bytes per lane and step, white balance by
// "guess and correct" against the table's threshold form T[k] (smallest v with table[v] >= k, kept in
// LDS): g = trunc((v - p2) * 255/(p98 - p2)) in float64 is within one level of the reference's
// trunc(float32(float64 expression)), and two threshold compares settle it exactly.

// Pixel -> wave mapping: a wave owns runs of 256 * RUN consecutive pixels, the RUN quads of a run unrolled: RUN loads, then per
// plane RUN back-to-back 1 KiB stores.  RUN = 16 (runs of 4096 pixels, 16 KiB bursts per plane) in the headline instantiation (three
// planes + basic statistics), 8 for uint16 NDVI + RGBA, 4 everywhere else -- measured per instantiation in round 3 (the table at
// `constexpr int RUN` below, profiles/r03_waves_ab.txt).  Round 2 had measured runs of 1024 pixels against one quad per trip and
// against 256-pixel slabs a grid stride apart (profiles/r02_traverse_ab.txt): 8 % faster wherever the planes' placement allows
// more than 5.3 TB/s at all, and equal elsewhere; the other mappings are gone from the source.
// CH = 4 (uint8 only): RGBA tiles -- 16 bytes per quad of pixels, one dwordx4 load per lane, repacked in three v_perm_b32 into
// the three dwords the rest of the kernel works on; alpha is ignored and comes back as 0 in the white-balanced image
// (np.zeros_like + range(3), process-images.py:432-435).
// Two-index masks (3, 5, 6) run the MASK = 7 instantiation with the unrequested plane's pointers null and P.mask deciding
// which records are flushed: a launch that is bound by its stores does not notice the spare quotient.
// Resident waves per SIMD: the plane-writing uint8 kernels stream best with THREE (measured in one process against one arena,
// bench.py modes: planes only 0.774 -> 0.808 of 8 TB/s where its 58 registers would allow eight; one plane + statistics 0.682 ->
// 0.689; three planes + statistics, four by its registers, unchanged).  The variants with histograms (124 registers) lose 4 %
// below four and keep what their registers give.
template <typename PIX, unsigned MASK, bool WB, int STATS, int CH = 3>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(1, (sizeof(PIX) == 1 && CH == 3 && MASK != 0u && STATS <= 1) ? 3 : 8))) void k_fused_u8c3(FusedParams P)
{
    constexpr bool U16 = sizeof(PIX) == 2;
    static_assert(CH == 3 || (CH == 4 && !U16), "4-channel fast path: uint8 tiles");
    __shared__ uint8_t s_lut[U16 ? 16 : 3 * 256];
    __shared__ unsigned int s_thr[U16 ? 3 * 260 : 4];
    __shared__ double s_par[U16 ? 6 : 1];
    // sixteen lane-private copies of the three 50-bin histograms, interleaved bin by bin (copy = lane % 16): index values
    // cluster, and with a single copy the LDS atomics of a wave queued on a few words (configs[2]: 0.745 of the roofline
    // against 0.78 without histograms)
    constexpr int HIST_COPIES = 16;
    __shared__ unsigned int s_hist_all[HIST_COPIES * 3 * LARS_HIST_BINS];
    __shared__ HistCell<float> s_edges[LARS_HIST_CELLS];
    __shared__ double s_red[4][4];

    const int tid = threadIdx.x;
    unsigned int *const s_hist = s_hist_all + (tid & (HIST_COPIES - 1));
    const unsigned int bx = blockIdx.x, gx = gridDim.x;
    const long long tile = blockIdx.y;
    const long long npix = P.npix;
    const PIX *base = static_cast<const PIX *>(P.tiles) + tile * npix * CH;

    if (WB) {
        if (U16) {
            const uint8_t *blob = P.wb_table + tile * LARS_U16_BLOB_BYTES;
            const unsigned int *thr = reinterpret_cast<const unsigned int *>(blob + LARS_U16_THR_OFFSET);
            const double *par = reinterpret_cast<const double *>(blob + LARS_U16_PAR_OFFSET);
            for (int i = tid; i < 3 * 260; i += 256) s_thr[i] = thr[i];
            if (tid < 6) s_par[tid] = par[tid];
        } else {
            const uint8_t *t = P.wb_table + tile * 768;
            for (int i = tid; i < 768; i += 256) s_lut[i] = t[i];
        }
    }
    if (STATS >= 2) {
        for (int i = tid; i < HIST_COPIES * 3 * LARS_HIST_BINS; i += 256) s_hist_all[i] = 0;
        hist_cells_init<float>(s_edges, tid);
    }
    if (WB || STATS >= 2) __syncthreads();

    // white balance of one sample of channel c
    auto wb_map = [&](unsigned int v, int c) -> unsigned int {
        if (!U16) return s_lut[c * 256 + v];
        const double y = ((double)v - s_par[2 * c]) * s_par[2 * c + 1];
        int g = (int)y;                                        // saturating; NaN -> 0
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        const unsigned int *t = s_thr + c * 260;
        g += (v >= t[g + 1] ? 1 : 0) - (v < t[g] ? 1 : 0);
        return (unsigned int)g;
    };

    Acc acc[3];
    acc_init(acc[0]); acc_init(acc[1]); acc_init(acc[2]);

#ifdef LARS_LAB_LAYOUT
    const long long ots = P.out_tile_stride;
#else
    const long long ots = npix;
#endif
    float *const oi0 = P.out_index[0] ? P.out_index[0] + tile * ots : nullptr;
    float *const oi1 = P.out_index[1] ? P.out_index[1] + tile * ots : nullptr;
    float *const oi2 = P.out_index[2] ? P.out_index[2] + tile * ots : nullptr;
    uint8_t *const owb = P.out_wb ? P.out_wb + tile * npix * CH : nullptr;
    uint8_t *const oc0 = P.out_rgba[0] ? P.out_rgba[0] + tile * npix * 4 : nullptr;
    uint8_t *const oc1 = P.out_rgba[1] ? P.out_rgba[1] + tile * npix * 4 : nullptr;
    uint8_t *const oc2 = P.out_rgba[2] ? P.out_rgba[2] + tile * npix * 4 : nullptr;
    const unsigned int *lut0 = reinterpret_cast<const unsigned int *>(P.cmap_lut[0]);
    const unsigned int *lut1 = reinterpret_cast<const unsigned int *>(P.cmap_lut[1]);
    const unsigned int *lut2 = reinterpret_cast<const unsigned int *>(P.cmap_lut[2]);

    const long long nquads = npix >> 2;
    const bool nt_st = (P.flags & 0x20000000u) != 0;
    __amdgpu_buffer_rsrc_t rsrc16;
    if (U16) rsrc16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<PIX *>(base), 0, (int)(nquads * 24), 0x00020000);
    constexpr int NW = U16 ? 6 : 3;                          // dwords of one quad of pixels

    auto load_quad = [&](long long q, unsigned int (&w)[NW]) {
        if (U16) {
            const unsigned int off = (unsigned int)q * 24u;
            const fu32x4 a4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc16, off, 0, 0);
            const fu32x2 a2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc16, off + 16u, 0, 0);
            w[0] = a4.x; w[1] = a4.y; w[2] = a4.z; w[NW - 3] = a4.w; w[NW - 2] = a2.x; w[NW - 1] = a2.y;
        } else if (CH == 4) {
            // r0 g0 n0 a0 | r1 g1 n1 a1 | r2 g2 n2 a2 | r3 g3 n3 a3  ->  r0 g0 n0 r1 | g1 n1 r2 g2 | n2 r3 g3 n3
            const uint4 p = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(base) + q * 16);
            w[0] = __builtin_amdgcn_perm(p.y, p.x, 0x04020100u);
            w[1] = __builtin_amdgcn_perm(p.z, p.y, 0x05040201u);
            w[2] = __builtin_amdgcn_perm(p.w, p.z, 0x06050402u);
        } else {
            const unsigned int *p = reinterpret_cast<const unsigned int *>(reinterpret_cast<const uint8_t *>(base) + q * 12);
            w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
        }
    };
    auto do_quad = [&](long long q, const unsigned int (&w)[NW]) {
        unsigned int b[12];
        if (U16) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { b[2 * i] = w[i % NW] & 0xFFFFu; b[2 * i + 1] = w[i % NW] >> 16; }
        } else {
            const unsigned int w0 = w[0], w1 = w[1], w2 = w[2];
            const unsigned int t[12] = {w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, w0 >> 24,
                                        w1 & 0xFF, (w1 >> 8) & 0xFF, (w1 >> 16) & 0xFF, w1 >> 24,
                                        w2 & 0xFF, (w2 >> 8) & 0xFF, (w2 >> 16) & 0xFF, w2 >> 24};
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = t[i];
        }
        if (WB) {
            // channels no requested index reads are only mapped when the white-balanced image is written
            const bool need_r = (MASK & 1u) || owb, need_g = (MASK & 6u) || owb;
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (i % 3 == 2 || (i % 3 == 0 && need_r) || (i % 3 == 1 && need_g)) b[i] = wb_map(b[i], i % 3);
            if (owb && CH == 4) {
                *reinterpret_cast<uint4 *>(owb + q * 16) = make_uint4(b[0] | (b[1] << 8) | (b[2] << 16), b[3] | (b[4] << 8) | (b[5] << 16),
                                                                      b[6] | (b[7] << 8) | (b[8] << 16), b[9] | (b[10] << 8) | (b[11] << 16));
            } else if (owb) {
                unsigned int *o = reinterpret_cast<unsigned int *>(owb + q * 12);
                o[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                o[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                o[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
            }
        }
        float v0[4], v1[4], v2[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            v0[px] = v1[px] = v2[px] = 0.0f;
            pixel_math<MASK, STATS, false, !U16, HIST_COPIES>((float)b[3 * px], (float)b[3 * px + 1], (float)b[3 * px + 2], 0u,
                                                            v0[px], v1[px], v2[px], acc, s_hist, s_edges);
        }
        if ((MASK & 1u) && oi0) store_plane4(oi0 + q * 4, v0, nt_st);
        if ((MASK & 2u) && oi1) store_plane4(oi1 + q * 4, v1, nt_st);
        if ((MASK & 4u) && oi2) store_plane4(oi2 + q * 4, v2, nt_st);
        if ((MASK & 1u) && oc0)
            *reinterpret_cast<uint4 *>(oc0 + q * 16) = make_uint4(lut0[cmap_index(v0[0])], lut0[cmap_index(v0[1])],
                                                                  lut0[cmap_index(v0[2])], lut0[cmap_index(v0[3])]);
        if ((MASK & 2u) && oc1)
            *reinterpret_cast<uint4 *>(oc1 + q * 16) = make_uint4(lut1[cmap_index(v1[0])], lut1[cmap_index(v1[1])],
                                                                  lut1[cmap_index(v1[2])], lut1[cmap_index(v1[3])]);
        if ((MASK & 4u) && oc2)
            *reinterpret_cast<uint4 *>(oc2 + q * 16) = make_uint4(lut2[cmap_index(v2[0])], lut2[cmap_index(v2[1])],
                                                                  lut2[cmap_index(v2[2])], lut2[cmap_index(v2[3])]);
    };

    // A wave owns 256 * RUN consecutive pixels per step: RUN wave-contiguous loads (768 bytes of uint8 samples each), then per
    // plane RUN wave-contiguous 1 KiB stores of consecutive addresses.  (Round 2, RUN = 4 against one 256-pixel slab per wave and
    // step, slabs a grid stride apart, as bare traffic: 6.07-6.38 vs 5.63-5.95 TB/s in three sets of allocations,
    // profiles/r02_stream_probe.txt, kinds 9-13 vs 5.)
    // Quads per lane and step (a wave owns 256 * RUN consecutive pixels: RUN loads of 768 bytes in flight, then per plane RUN
    // back-to-back 1 KiB stores).  Three planes + basic statistics stream better the longer the run, as long as two waves per SIMD
    // stay resident: 4 / 8 / 12 / 16 quads = 0.784 / 0.788 / 0.792 / 0.795 of 8 TB/s into one arena, 20 (one wave left) 0.66
    // (profiles/r03_waves_ab.txt).  The other instantiations gain nothing from more than four or lose (one plane: 0.686 -> 0.672).
    // uint16 tiles, one index (+ its RGBA picture) + basic statistics -- BASELINE configs[4]: 2 / 4 / 8 quads = 0.636 / 0.669 / 0.681;
    // RGBA uint8 tiles: 4 is best (0.787 against 0.770 with 8).
    constexpr int RUN = (!U16 && CH == 3 && MASK == 7u && STATS == 1) ? 16 : (U16 && MASK == 1u && STATS <= 1) ? 8 : 4;
    const long long nsteps = (nquads + 64 * RUN - 1) / (64 * RUN);
    const long long wstride = (long long)gx * 4;
    const unsigned int lane = (unsigned int)tid & 63u;
    for (long long st = (long long)bx * 4 + (tid >> 6); st < nsteps; st += wstride) {
        const long long q0 = st * (64 * RUN) + lane;
        unsigned int w[RUN][NW];
        if (st * (64 * RUN) + 64 * RUN <= nquads) {
#pragma unroll
            for (int j = 0; j < RUN; ++j) load_quad(q0 + 64 * j, w[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RUN; ++j) {
                do_quad(q0 + 64 * j, w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int j = 0; j < RUN; ++j) {
                const long long q = q0 + 64 * j;
                if (q < nquads) { load_quad(q, w[0]); do_quad(q, w[0]); }
            }
        }
    }
    // tail pixels (npix % 4): lanes 0..2 of block 0
    if (bx == 0 && tid < (int)(npix & 3)) {
        const long long i = nquads * 4 + tid;
        unsigned int r = base[i * CH], g = base[i * CH + 1], n = base[i * CH + 2];
        if (WB) { r = wb_map(r, 0); g = wb_map(g, 1); n = wb_map(n, 2); }
        if (WB && owb) {
            owb[i * CH] = (uint8_t)r; owb[i * CH + 1] = (uint8_t)g; owb[i * CH + 2] = (uint8_t)n;
            if (CH == 4) owb[i * CH + 3] = 0;
        }
        float a = 0, bq = 0, c = 0;
        pixel_math<MASK, STATS, false, !U16, HIST_COPIES>((float)r, (float)g, (float)n, 0u, a, bq, c, acc, s_hist, s_edges);
        if ((MASK & 1u) && oi0) oi0[i] = a;
        if ((MASK & 2u) && oi1) oi1[i] = bq;
        if ((MASK & 4u) && oi2) oi2[i] = c;
        if ((MASK & 1u) && oc0) reinterpret_cast<unsigned int *>(oc0)[i] = lut0[cmap_index(a)];
        if ((MASK & 2u) && oc1) reinterpret_cast<unsigned int *>(oc1)[i] = lut1[cmap_index(bq)];
        if ((MASK & 4u) && oc2) reinterpret_cast<unsigned int *>(oc2)[i] = lut2[cmap_index(c)];
    }

    if (STATS >= 1) {
        StatsAccView *rec = reinterpret_cast<StatsAccView *>(P.stats + tile * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if ((MASK & (1u << k)) && (P.mask & (1u << k))) acc_flush(acc[k], rec + k, s_red, tid);
        if (STATS >= 2) {
            __syncthreads();
            for (int i = tid; i < 3 * LARS_HIST_BINS; i += 256) {
                const int k = i / LARS_HIST_BINS;
                unsigned int v = 0;
#pragma unroll
                for (int c = 0; c < HIST_COPIES; ++c) v += s_hist_all[i * HIST_COPIES + ((c + i) & (HIST_COPIES - 1))];
                if ((MASK & (1u << k)) && (P.mask & (1u << k)) && v) atomicAdd(&rec[k].hist[i - k * LARS_HIST_BINS], (unsigned long long)v);
            }
        }
    }
}

// ---- generic path: any channel count >= 3, uint8 / uint16, any alignment ---
template <typename PIX, int NVAL>
__global__ __launch_bounds__(256) void k_fused_generic(FusedParams P)
{
    __shared__ unsigned int s_hist[3 * LARS_HIST_BINS];
    __shared__ HistCell<float> s_edges[LARS_HIST_CELLS];
    __shared__ double s_red[4][4];

    const int tid = threadIdx.x;
    const long long tile = blockIdx.y;
    const long long npix = P.npix;
    const int C = P.channels;
    const unsigned mask = P.mask;
    const bool stats = P.flags & (LARS_F_STATS | LARS_F_HIST | LARS_F_SUMSQ);
    const bool hist = P.flags & (LARS_F_HIST | LARS_F_SUMSQ);
    const bool sumsq = P.flags & LARS_F_SUMSQ;
    const PIX *base = static_cast<const PIX *>(P.tiles) + tile * npix * C;
    const uint8_t *tab = P.wb_table ? P.wb_table + tile * (NVAL == 256 ? 768ll : (long long)LARS_U16_BLOB_BYTES) : nullptr;

    for (int i = tid; i < 3 * LARS_HIST_BINS; i += 256) s_hist[i] = 0;
    hist_cells_init<float>(s_edges, tid);
    __syncthreads();

    Acc acc[3];
    acc_init(acc[0]); acc_init(acc[1]); acc_init(acc[2]);
    float *const oi0 = P.out_index[0] ? P.out_index[0] + tile * npix : nullptr;
    float *const oi1 = P.out_index[1] ? P.out_index[1] + tile * npix : nullptr;
    float *const oi2 = P.out_index[2] ? P.out_index[2] + tile * npix : nullptr;
    uint8_t *const owb = P.out_wb ? P.out_wb + tile * npix * C : nullptr;
    unsigned int *const oc0 = P.out_rgba[0] ? reinterpret_cast<unsigned int *>(P.out_rgba[0]) + tile * npix : nullptr;
    unsigned int *const oc1 = P.out_rgba[1] ? reinterpret_cast<unsigned int *>(P.out_rgba[1]) + tile * npix : nullptr;
    unsigned int *const oc2 = P.out_rgba[2] ? reinterpret_cast<unsigned int *>(P.out_rgba[2]) + tile * npix : nullptr;
    const unsigned int *lut0 = reinterpret_cast<const unsigned int *>(P.cmap_lut[0]);
    const unsigned int *lut1 = reinterpret_cast<const unsigned int *>(P.cmap_lut[1]);
    const unsigned int *lut2 = reinterpret_cast<const unsigned int *>(P.cmap_lut[2]);

    for (long long i = (long long)blockIdx.x * 256 + tid; i < npix; i += (long long)gridDim.x * 256) {
        const PIX *p = base + i * C;
        unsigned int r = p[0], g = p[1], n = p[2];
        if (tab) {
            r = tab[r]; g = tab[NVAL + g]; n = tab[2 * NVAL + n];
            if (owb) {
                uint8_t *o = owb + i * C;
                o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)n;
                for (int c = 3; c < C; ++c) o[c] = 0;            // zeros_like, process-images.py:432
            }
        }
        float a = 0, b = 0, c = 0;
        if (stats) {
            if (sumsq) pixel_math<7u, 3, true>((float)r, (float)g, (float)n, mask, a, b, c, acc, s_hist, s_edges);
            else if (hist) pixel_math<7u, 2, true>((float)r, (float)g, (float)n, mask, a, b, c, acc, s_hist, s_edges);
            else pixel_math<7u, 1, true>((float)r, (float)g, (float)n, mask, a, b, c, acc, s_hist, s_edges);
        } else {
            pixel_math<7u, 0, true>((float)r, (float)g, (float)n, mask, a, b, c, acc, s_hist, s_edges);
        }
        if ((mask & 1u) && oi0) oi0[i] = a;
        if ((mask & 2u) && oi1) oi1[i] = b;
        if ((mask & 4u) && oi2) oi2[i] = c;
        if ((mask & 1u) && oc0) oc0[i] = lut0[cmap_index(a)];
        if ((mask & 2u) && oc1) oc1[i] = lut1[cmap_index(b)];
        if ((mask & 4u) && oc2) oc2[i] = lut2[cmap_index(c)];
    }
    if (stats) {
        StatsAccView *rec = reinterpret_cast<StatsAccView *>(P.stats + tile * 3);
        for (int k = 0; k < 3; ++k)
            if (mask & (1u << k)) acc_flush(acc[k], rec + k, s_red, tid);
        if (hist) {
            __syncthreads();
            for (int i = tid; i < 3 * LARS_HIST_BINS; i += 256) {
                const int k = i / LARS_HIST_BINS;
                if ((mask & (1u << k)) && s_hist[i])
                    atomicAdd(&rec[k].hist[i - k * LARS_HIST_BINS], (unsigned long long)s_hist[i]);
            }
        }
    }
}

// ===========================================================================
// Synthetic tiles
// ===========================================================================
__device__ inline unsigned int mix32(unsigned int x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__device__ inline unsigned int veg_byte(unsigned int v, int ch)
{
    return ch == 0 ? 20u + (v * 3u) / 8u : ch == 1 ? 40u + v / 2u : ch == 2 ? 60u + (v * 3u) / 4u : v;
}

__global__ __launch_bounds__(256) void k_synth_u8(uint8_t *tiles, long long first_tile, long long nbytes,
                                                  int channels, unsigned int seed, int profile)
{
    const long long tile_local = blockIdx.y;
    const unsigned int salt = (unsigned int)(first_tile + tile_local) * 0x9E3779B9u;
    uint8_t *base = tiles + tile_local * nbytes;
    const long long nwords = (nbytes + 3) >> 2;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < nwords; k += (long long)gridDim.x * 256) {
        unsigned int w = mix32(mix32((unsigned int)k + seed) ^ salt);
        if (profile == 1) {
            unsigned int o = 0;
            for (int j = 0; j < 4; ++j) {
                const int ch = (int)((k * 4 + j) % channels);
                o |= (veg_byte((w >> (8 * j)) & 0xFF, ch) & 0xFF) << (8 * j);
            }
            w = o;
        }
        const long long off = k * 4;
        if (off + 4 <= nbytes && ((reinterpret_cast<uintptr_t>(base + off) & 3) == 0)) {
            *reinterpret_cast<unsigned int *>(base + off) = w;
        } else {
            for (int j = 0; j < 4 && off + j < nbytes; ++j) base[off + j] = (uint8_t)(w >> (8 * j));
        }
    }
}

}  // namespace lars

// ===========================================================================
// Launchers (device entry points)
// ===========================================================================
using namespace lars;

namespace lars {
int blocks_per_tile(long long work_items, long long ntiles, int threads, long long target_total)
{
    // target_total workgroups per launch (many more than the 256 CUs hold at once: tools/kbench.py
    // sweeps), but never so many that a block runs fewer than ~16 steps -- every block pays for its
    // LDS tables and its statistics flush
    long long want = (target_total * 256 / threads + ntiles - 1) / ntiles;
    if (tuning().blocks_per_tile > 0) want = tuning().blocks_per_tile;
    long long cap = work_items / ((long long)threads * 16);
    if (cap < 1) cap = 1;
    if (want > cap && tuning().blocks_per_tile <= 0) want = cap;
    const long long hard = (work_items + threads - 1) / threads;
    if (want > hard) want = hard;
    if (want < 1) want = 1;
    if (want > 65535) want = 65535;
    return (int)want;
}
}  // namespace lars

extern "C" int lars_d_channel_hist(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                   uint32_t *hist, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || !hist || ntiles <= 0 || npix <= 0 || channels < 3)
        return fail(LARS_ERR_INVALID, "lars_d_channel_hist: bad arguments");
    if (ntiles > 65535) return fail(LARS_ERR_INVALID, "lars_d_channel_hist: ntiles > 65535 per call");
    hipStream_t s = pick_stream(c, stream);
    const int nval = dtype == LARS_U8 ? 256 : 65536;
    if (dtype != LARS_U8 && dtype != LARS_U16) return fail(LARS_ERR_INVALID, "lars_d_channel_hist: dtype");
    LARS_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)ntiles * 3 * nval * sizeof(uint32_t), s));
    const bool fast = dtype == LARS_U8 && channels == 3 && (ntiles == 1 || (npix & 3) == 0) &&
                      ((reinterpret_cast<uintptr_t>(tiles) & 3) == 0);
    const bool fast4 = dtype == LARS_U8 && channels == 4 && (reinterpret_cast<uintptr_t>(tiles) & 15) == 0 &&
                       (ntiles == 1 || (npix & 3) == 0) && (long long)npix * 4 < (1ll << 30);     // RGBA: 16 bytes per quad of pixels
    if ((fast && tuning().hist_impl != 1 && (long long)npix * 3 < (1ll << 30)) || fast4) {
        // 96 KiB of LDS per block: one 1024-thread block per CU, a few waves of blocks per tile
        long long want = tuning().blocks_per_tile > 0 ? tuning().blocks_per_tile : (1024 + ntiles - 1) / ntiles;
        const long long cap = (npix / 4 + 1023) / 1024;
        if (want > cap) want = cap;
        if (want < 1) want = 1;
        dim3 grid((unsigned)want, (unsigned)ntiles);
        chan_hist_v2_launch(static_cast<const uint8_t *>(tiles), (long long)npix, hist, grid, s, channels);
    } else if (fast) {
        dim3 grid(blocks_per_tile(npix / 4 + 1, ntiles), (unsigned)ntiles);
        hipLaunchKernelGGL(k_chan_hist_u8c3, grid, dim3(256), 0, s, static_cast<const uint8_t *>(tiles),
                           (long long)npix, hist);
    } else if (dtype == LARS_U8) {
        dim3 grid(blocks_per_tile(npix, ntiles), (unsigned)ntiles);
        hipLaunchKernelGGL((k_chan_hist_generic<uint8_t, 256>), grid, dim3(256), 0, s,
                           static_cast<const uint8_t *>(tiles), (long long)npix, channels, hist);
    } else {
        dim3 grid(blocks_per_tile(npix, ntiles), (unsigned)ntiles);
        hipLaunchKernelGGL((k_chan_hist_generic<uint16_t, 65536>), grid, dim3(256), 0, s,
                           static_cast<const uint16_t *>(tiles), (long long)npix, channels, hist);
    }
    return launch_check("lars_d_channel_hist");
}

extern "C" int lars_d_wb_table(const uint32_t *hist, int64_t ntiles, int64_t npix, int dtype, uint8_t *table,
                               double *percentiles, int rgn_variant, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!hist || !table || ntiles <= 0 || npix <= 0) return fail(LARS_ERR_INVALID, "lars_d_wb_table: bad arguments");
    if (ntiles > 65535) return fail(LARS_ERR_INVALID, "lars_d_wb_table: ntiles > 65535 per call");
    hipStream_t s = pick_stream(c, stream);
    dim3 grid(3, (unsigned)ntiles);
    if (dtype == LARS_U8)
        hipLaunchKernelGGL((k_wb_table<256>), grid, dim3(256), 0, s, hist, (long long)npix, table, percentiles, rgn_variant);
    else if (dtype == LARS_U16)
        hipLaunchKernelGGL((k_wb_table<65536>), grid, dim3(256), 0, s, hist, (long long)npix, table, percentiles, rgn_variant);
    else
        return fail(LARS_ERR_INVALID, "lars_d_wb_table: dtype");
    return launch_check("lars_d_wb_table");
}

template <typename PIX, unsigned MASK, bool WB, int CH>
static void launch_fast_stats(int stats_mode, dim3 grid, hipStream_t s, const FusedParams &P)
{
    if (stats_mode == 0) hipLaunchKernelGGL((k_fused_u8c3<PIX, MASK, WB, 0, CH>), grid, dim3(256), 0, s, P);
    else if (stats_mode == 1) hipLaunchKernelGGL((k_fused_u8c3<PIX, MASK, WB, 1, CH>), grid, dim3(256), 0, s, P);
    else if (stats_mode == 3) hipLaunchKernelGGL((k_fused_u8c3<PIX, MASK, WB, 3, CH>), grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((k_fused_u8c3<PIX, MASK, WB, 2, CH>), grid, dim3(256), 0, s, P);
}
template <typename PIX, unsigned MASK, int CH>
static void launch_fast_wb(bool wb, int stats_mode, dim3 grid, hipStream_t s, const FusedParams &P)
{
    if (wb) launch_fast_stats<PIX, MASK, true, CH>(stats_mode, grid, s, P);
    else launch_fast_stats<PIX, MASK, false, CH>(stats_mode, grid, s, P);
}
// mask: the template's mask (two-index masks come in as 7, with P.mask = the requested bits)
template <typename PIX, int CH = 3>
static void launch_fast(unsigned mask, bool wb, int stats_mode, dim3 grid, hipStream_t s, const FusedParams &P)
{
    switch (mask) {
    case 0u: hipLaunchKernelGGL((k_fused_u8c3<PIX, 0u, true, 0, CH>), grid, dim3(256), 0, s, P); break;
    case 1u: launch_fast_wb<PIX, 1u, CH>(wb, stats_mode, grid, s, P); break;
    case 2u: launch_fast_wb<PIX, 2u, CH>(wb, stats_mode, grid, s, P); break;
    case 4u: launch_fast_wb<PIX, 4u, CH>(wb, stats_mode, grid, s, P); break;
    default: launch_fast_wb<PIX, 7u, CH>(wb, stats_mode, grid, s, P); break;
    }
}

extern "C" int lars_d_fused(const lars_fused_args *a)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!a || !a->tiles || a->ntiles <= 0 || a->npix <= 0 || a->channels < 3)
        return fail(LARS_ERR_INVALID, "lars_d_fused: bad arguments");
    if (a->ntiles > 65535) return fail(LARS_ERR_INVALID, "lars_d_fused: ntiles > 65535 per call");
    if (a->dtype != LARS_U8 && a->dtype != LARS_U16) return fail(LARS_ERR_INVALID, "lars_d_fused: dtype");
    const unsigned mask = a->index_mask & LARS_MASK_ALL;
    if (a->index_mask & ~LARS_MASK_ALL) return fail(LARS_ERR_INVALID, "lars_d_fused: unknown index bits in mask");
    // 1 basic, 2 + 50-bin histograms, 3 + histograms and the sum of squares (LARS_F_SUMSQ; implies the histograms)
    const int stats_mode = (a->flags & LARS_F_SUMSQ) ? 3 : (a->flags & LARS_F_HIST) ? 2 : (a->flags & LARS_F_STATS) ? 1 : 0;
    if (stats_mode && !a->stats) return fail(LARS_ERR_INVALID, "lars_d_fused: stats requested but stats == NULL");
    if (a->out_wb && !a->wb_table) return fail(LARS_ERR_INVALID, "lars_d_fused: out_wb needs wb_table");
    for (int k = 0; k < 3; ++k)
        if (a->out_rgba[k] && !a->cmap_lut[k]) return fail(LARS_ERR_INVALID, "lars_d_fused: out_rgba needs cmap_lut");
    if (mask == 0 && !a->out_wb) return fail(LARS_ERR_INVALID, "lars_d_fused: nothing to do");
    hipStream_t s = pick_stream(c, a->stream);

    FusedParams P;
    P.tiles = a->tiles; P.npix = a->npix; P.channels = a->channels; P.wb_table = a->wb_table;
    for (int k = 0; k < 3; ++k) {
        const bool on = (mask >> k) & 1u;
        P.out_index[k] = on ? a->out_index[k] : nullptr;
        P.out_rgba[k] = on ? a->out_rgba[k] : nullptr;
        P.cmap_lut[k] = a->cmap_lut[k];
    }
    P.out_wb = a->out_wb; P.stats = a->stats; P.mask = mask; P.sel_hist = nullptr; P.sel_win = nullptr; P.sel_win_hist = nullptr; P.sel_below = nullptr;
    P.flags = (a->flags & 7u) | (tuning().nt_stores ? 0x20000000u : 0u);
#ifdef LARS_LAB_LAYOUT
    P.out_tile_stride = a->npix * (tuning().out_stride_planes > 1 ? tuning().out_stride_planes : 1);
#endif

    const long long nrec = a->ntiles * 3;
    const bool raw = (a->flags & LARS_F_RAW) != 0;        // the caller brackets the launches with lars_d_stats_begin / _end
    if (stats_mode && !raw)
        hipLaunchKernelGGL(k_stats_init, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, a->stats, nrec, mask);

    const bool aligned = (reinterpret_cast<uintptr_t>(a->tiles) & 3) == 0 &&
                         (!a->out_wb || (reinterpret_cast<uintptr_t>(a->out_wb) & 3) == 0) &&
                         (!P.out_index[0] || (reinterpret_cast<uintptr_t>(P.out_index[0]) & 15) == 0) &&
                         (!P.out_index[1] || (reinterpret_cast<uintptr_t>(P.out_index[1]) & 15) == 0) &&
                         (!P.out_index[2] || (reinterpret_cast<uintptr_t>(P.out_index[2]) & 15) == 0) &&
                         (!P.out_rgba[0] || (reinterpret_cast<uintptr_t>(P.out_rgba[0]) & 15) == 0) &&
                         (!P.out_rgba[1] || (reinterpret_cast<uintptr_t>(P.out_rgba[1]) & 15) == 0) &&
                         (!P.out_rgba[2] || (reinterpret_cast<uintptr_t>(P.out_rgba[2]) & 15) == 0);
    // the specialised kernels exist for "white balance only" (0), one index (1, 2, 4) and all three (7); two-index masks
    // run the three-index kernels with the third plane's pointers null and its record left alone (P.mask)
    const unsigned tmask = (mask == 3u || mask == 5u || mask == 6u) ? 7u : mask;
    const bool ch_ok = a->channels == 3 || (a->channels == 4 && a->dtype == LARS_U8 && (reinterpret_cast<uintptr_t>(a->tiles) & 15) == 0 &&
                                            (!a->out_wb || (reinterpret_cast<uintptr_t>(a->out_wb) & 15) == 0));
    const bool fast = ch_ok && aligned && (a->ntiles == 1 || (a->npix & 3) == 0);
    // fused_impl 0 = automatic: the second-generation kernels win where the launch is read-bound
    // (statistics only); with output planes the launch is write-bound and the lighter first-generation
    // kernel (more resident waves, no LDS table) is as fast or faster (tools/kbench.py)
    const bool any_out = P.out_wb || P.out_index[0] || P.out_index[1] || P.out_index[2] || P.out_rgba[0] ||
                         P.out_rgba[1] || P.out_rgba[2];
    int impl = tuning().fused_impl;
    if (impl == 0) impl = (any_out && mask != 0u) ? 1 : 2;
    // the second-generation kernels address a tile through a raw buffer descriptor with 32-bit offsets
    const bool small_tile = (long long)a->npix * 6 < (1ll << 30);
    int &ran = tuning().last_fused_kernel;                    // which kernel family served the launch (tests, tools)
    if (fast && a->dtype == LARS_U8 && a->channels == 3 && impl >= 2 && small_tile) {
        dim3 grid(blocks_per_tile(a->npix / 4 + 1, a->ntiles, fused_v2_threads(any_out || mask == 0u)), (unsigned)a->ntiles);
        fused_v2_launch(tmask, a->wb_table != nullptr, stats_mode, tuning().nt_stores != 0, grid, s, P);
        ran = 2;
    } else if (fast && a->dtype == LARS_U8 && a->channels == 4) {
        dim3 grid(blocks_per_tile(a->npix / 4 + 1, a->ntiles, 256, 32768), (unsigned)a->ntiles);
        launch_fast<uint8_t, 4>(tmask, a->wb_table != nullptr, stats_mode, grid, s, P);
        ran = 5;
    } else if (fast && a->dtype == LARS_U8) {
        dim3 grid(blocks_per_tile(a->npix / 4 + 1, a->ntiles, 256, 32768), (unsigned)a->ntiles);
        launch_fast<uint8_t>(tmask, a->wb_table != nullptr, stats_mode, grid, s, P);
        ran = 1;
    } else if (fast && a->dtype == LARS_U16 && small_tile) {
        dim3 grid(blocks_per_tile(a->npix / 4 + 1, a->ntiles, 256, 32768), (unsigned)a->ntiles);
        launch_fast<uint16_t>(tmask, a->wb_table != nullptr, stats_mode, grid, s, P);
        ran = 3;
    } else {
        dim3 grid(blocks_per_tile(a->npix, a->ntiles), (unsigned)a->ntiles);
        if (a->dtype == LARS_U8) hipLaunchKernelGGL((k_fused_generic<uint8_t, 256>), grid, dim3(256), 0, s, P);
        else hipLaunchKernelGGL((k_fused_generic<uint16_t, 65536>), grid, dim3(256), 0, s, P);
        ran = 4;
    }
    if (stats_mode && !raw)
        hipLaunchKernelGGL(k_stats_finalize, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, a->stats, nrec, mask,
                           (long long)a->npix);
    return launch_check("lars_d_fused");
}

extern "C" int lars_d_stats_begin(lars_stats *stats, int64_t ntiles, uint32_t index_mask, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!stats || ntiles <= 0 || (index_mask & ~LARS_MASK_ALL)) return fail(LARS_ERR_INVALID, "lars_d_stats_begin: bad arguments");
    stats_init_launch(stats, ntiles * 3, index_mask & LARS_MASK_ALL, pick_stream(c, stream));
    return launch_check("lars_d_stats_begin");
}
extern "C" int lars_d_stats_end(lars_stats *stats, int64_t ntiles, uint32_t index_mask, int64_t npix, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!stats || ntiles <= 0 || npix <= 0 || (index_mask & ~LARS_MASK_ALL)) return fail(LARS_ERR_INVALID, "lars_d_stats_end: bad arguments");
    stats_finalize_launch(stats, ntiles * 3, index_mask & LARS_MASK_ALL, (long long)npix, pick_stream(c, stream));
    return launch_check("lars_d_stats_end");
}

extern "C" int lars_d_stats_fold(const lars_stats *tile_records, int64_t ntiles, uint32_t index_mask, lars_stats *out, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tile_records || !out || ntiles <= 0 || !(index_mask & LARS_MASK_ALL) || (index_mask & ~LARS_MASK_ALL))
        return fail(LARS_ERR_INVALID, "lars_d_stats_fold: bad arguments");
    hipLaunchKernelGGL(k_stats_fold, dim3(3), dim3(256), 0, pick_stream(c, stream), tile_records, (long long)ntiles,
                       index_mask & LARS_MASK_ALL, out);
    return launch_check("lars_d_stats_fold");
}

extern "C" int lars_d_synth_u8(uint8_t *tiles, int64_t ntiles, int64_t first_tile, int64_t npix, int channels,
                               uint32_t seed, int profile, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || ntiles <= 0 || npix <= 0 || channels < 1 || ntiles > 65535)
        return fail(LARS_ERR_INVALID, "lars_d_synth_u8: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    const long long nbytes = (long long)npix * channels;
    dim3 grid(blocks_per_tile((nbytes + 3) / 4, ntiles), (unsigned)ntiles);
    hipLaunchKernelGGL(k_synth_u8, grid, dim3(256), 0, s, tiles, (long long)first_tile, nbytes, channels, seed, profile);
    return launch_check("lars_d_synth_u8");
}

namespace lars {
void stats_init_launch(lars_stats *stats, long long nrec, unsigned int mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_stats_init, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, stats, nrec, mask);
}
void stats_finalize_launch(lars_stats *stats, long long nrec, unsigned int mask, long long npix, hipStream_t s)
{
    hipLaunchKernelGGL(k_stats_finalize, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, stats, nrec, mask, npix);
}
}  // namespace lars
