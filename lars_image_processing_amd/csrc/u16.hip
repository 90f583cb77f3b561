// uint16 white-balance pre-pass: exact np.percentile(ch, (2, 98)) of 16-bit samples
// without a 65536-bin histogram per channel.
//
// Two radix levels (the same idea as the median select in arrays.hip):
//   pass 1  histogram of the HIGH byte of every sample (3 x 256 bins, LDS, conflict-free copies)
//   pick    which high-byte bins hold the four order statistics np.percentile needs
//           (floor/ceil neighbours of the 2nd and 98th percentile) and the ranks inside them
//   pass 2  histogram of the LOW byte of the samples whose high byte is one of those bins
//   table   order statistics -> percentiles (numpy 'linear' + _lerp, float64) -> the 65536-entry
//           uint8 table (process-images.py:438-441 evaluated for every sample value), and its
//           threshold form T[k] = smallest v with table[v] >= k (the table is a monotone staircase)
//           which the fast fused kernel keeps in LDS.
//
// Both data passes stream the tile once at HBM rate; a 3 x 65536-bin histogram would need one
// global atomic per sample instead.
//
// Usually ONE full pass instead of those two (round 5: value windows, see k_u16_count_win below).  Rounds 3-4 predicted candidate high-byte
// BINS from a subsample and counted all high bytes plus the candidates' low bytes in one pass (k_hist_u16_both: 127 vector instructions
// per wave-quad, bound by instruction issue at 0.61 of 8 TB/s; profiles/r04_u16_prepass_counters.txt, r05_u16_prepare_ab.txt) -- removed
// when the value-window pass reached the HBM rate.  The two radix passes stay: as `u16_hist_impl` 1, for tiles the fast path does not
// serve, and as the fall-back of a tile whose window missed.
#include "common.h"
#include "device_common.h"

namespace lars {

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

struct U16Pick {
    unsigned int target[4];          // high byte holding order statistic r (q2.lo, q2.hi, q98.lo, q98.hi)
    unsigned int slot[4];            // pass-2 histogram slot of rank r (ranks sharing a high byte share a slot)
    unsigned long long resid[4];     // rank inside that high-byte bin
    double tq[2];                    // interpolation weights of the two percentiles
};

// ---- pass 1: high-byte histograms --------------------------------------------------------
// lane owns 4 pixels = 24 bytes (6 dwords: r0g0 n0r1 g1n1 r2g2 n2r3 g3n3, two samples per dword)
// every > 1: only every `every`-th grid stride is counted (the subsample that predicts the candidate bins)
// only: null, or flags per tile -- tiles whose flag is 0 are skipped (the classic passes as the fall-back of the one-pass forms)
__global__ __launch_bounds__(1024) void k_hist_u16_hi(const uint16_t *__restrict__ tiles, long long npix,
                                                      unsigned int *__restrict__ hist, int every, const unsigned int *__restrict__ only = nullptr)
{
    __shared__ unsigned int s_h[3 * 256 * 32];             // [channel][bin][copy = lane % 32]
    const int tid = threadIdx.x;
    if (only && !only[blockIdx.y]) return;
    for (int i = tid; i < 3 * 256 * 32; i += 1024) s_h[i] = 0;
    __syncthreads();
    const long long tile = blockIdx.y;
    const uint16_t *base = tiles + tile * npix * 3;
    const long long nquads = npix >> 2;
    const unsigned int lane_off = (tid & 31) << 2;
    char *hb = reinterpret_cast<char *>(s_h);
#define HADD16(word, half, ch)                                                                         \
    atomicAdd(reinterpret_cast<unsigned int *>(hb + (ch) * 32768 + ((((word) >> ((half) * 16 + 8)) & 0xFFu) << 7) + lane_off), 1u)
    const long long step = (long long)gridDim.x * 1024 * every;
    if (nquads > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)(nquads * 24), 0x00020000);
        for (long long q = (long long)blockIdx.x * 1024 + tid; q < nquads; q += step) {
            const unsigned int off = (unsigned int)q * 24u;
            const u32x4v a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            const u32x2v b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0);
            HADD16(a.x, 0, 0); HADD16(a.x, 1, 1); HADD16(a.y, 0, 2); HADD16(a.y, 1, 0);
            HADD16(a.z, 0, 1); HADD16(a.z, 1, 2); HADD16(a.w, 0, 0); HADD16(a.w, 1, 1);
            HADD16(b.x, 0, 2); HADD16(b.x, 1, 0); HADD16(b.y, 0, 1); HADD16(b.y, 1, 2);
        }
    }
    if (every == 1 && blockIdx.x == 0 && tid < (int)(npix & 3)) {
        const uint16_t *p = base + (nquads * 4 + tid) * 3;
        HADD16((unsigned)p[0], 0, 0); HADD16((unsigned)p[1], 0, 1); HADD16((unsigned)p[2], 0, 2);
    }
#undef HADD16
    __syncthreads();
    if (tid < 768) {
        const unsigned int *row = s_h + tid * 32;
        unsigned int v = 0;
        for (int j = 0; j < 32; ++j) v += row[(j + tid) & 31];
        if (v) atomicAdd(&hist[tile * 768 + tid], v);
    }
}

// any channel count / alignment
__global__ __launch_bounds__(256) void k_hist_u16_hi_generic(const uint16_t *__restrict__ tiles, long long npix,
                                                             int channels, unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_h[768];
    for (int i = threadIdx.x; i < 768; i += 256) s_h[i] = 0;
    __syncthreads();
    const long long tile = blockIdx.y;
    const uint16_t *base = tiles + tile * npix * channels;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const uint16_t *p = base + i * channels;
        atomicAdd(&s_h[p[0] >> 8], 1u);
        atomicAdd(&s_h[256 + (p[1] >> 8)], 1u);
        atomicAdd(&s_h[512 + (p[2] >> 8)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 256)
        if (s_h[i]) atomicAdd(&hist[tile * 768 + i], s_h[i]);
}

// ---- pick: which high bytes hold the order statistics -------------------------------------
__global__ __launch_bounds__(256) void k_u16_pick(const unsigned int *__restrict__ hist, long long npix,
                                                  U16Pick *__restrict__ picks, const unsigned int *__restrict__ only = nullptr)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ U16Pick s_pick;
    const int tid = threadIdx.x;
    if (only && !only[blockIdx.y]) return;
    const long long slot = (long long)blockIdx.y * 3 + blockIdx.x;
    const unsigned long long c = hist[slot * 256 + tid];
    s_scan[tid] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const unsigned long long before = s_scan[tid] - c;
    const double nm1 = (double)(npix - 1);
    for (int k = 0; k < 2; ++k) {
        const double q = (k == 0 ? 2.0 : 98.0) / 100.0;
        const double vi = nm1 * q;
        const double fl = floor(vi);
        long long lo = (long long)fl, hi = lo + 1;
        if (hi > npix - 1) hi = npix - 1;
        const long long rank[2] = {lo, hi};
        for (int j = 0; j < 2; ++j) {
            const unsigned long long r = (unsigned long long)rank[j];
            if (c && r >= before && r < before + c) {
                s_pick.target[2 * k + j] = (unsigned)tid;
                s_pick.resid[2 * k + j] = r - before;
            }
        }
        if (tid == 0) s_pick.tq[k] = vi - fl;
    }
    __syncthreads();
    if (tid == 0) {
        for (int r = 0; r < 4; ++r) {
            unsigned int sl = (unsigned)r;
            for (int p = 0; p < r; ++p)
                if (s_pick.target[p] == s_pick.target[r]) { sl = s_pick.slot[p]; break; }
            s_pick.slot[r] = sl;
        }
        picks[slot] = s_pick;
    }
}

// ---- pass 2: low-byte histograms of the samples in the picked high-byte bins ---------------
template <bool FAST>
__global__ __launch_bounds__(1024) void k_hist_u16_lo(const uint16_t *__restrict__ tiles, long long npix, int channels,
                                                      const U16Pick *__restrict__ picks, unsigned int *__restrict__ lohist,
                                                      const unsigned int *__restrict__ only = nullptr)
{
    __shared__ unsigned int s_h[3 * 4 * 256];
    __shared__ unsigned char s_slot[3 * 256];
    const int tid = threadIdx.x;
    const long long tile = blockIdx.y;
    if (only && !only[tile]) return;
    for (int i = tid; i < 3 * 4 * 256; i += 1024) s_h[i] = 0;
    if (tid < 768) s_slot[tid] = 0xFF;
    __syncthreads();
    if (tid < 12) {
        const int c = tid >> 2, r = tid & 3;
        const U16Pick &p = picks[tile * 3 + c];
        if (p.slot[r] == (unsigned)r) s_slot[c * 256 + p.target[r]] = (unsigned char)r;   // first rank of each bin owns the slot
    }
    __syncthreads();
    const uint16_t *base = tiles + tile * npix * channels;
#define LADD(sample, ch)                                                                               \
    {                                                                                                  \
        const unsigned int s_ = (sample);                                                              \
        const unsigned int sl_ = s_slot[(ch) * 256 + (s_ >> 8)];                                       \
        if (sl_ != 0xFFu) atomicAdd(&s_h[((ch) * 4 + sl_) * 256 + (s_ & 0xFFu)], 1u);                  \
    }
    if (FAST) {
        const long long nquads = npix >> 2;
        const long long step = (long long)gridDim.x * 1024;
        if (nquads > 0) {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)(nquads * 24), 0x00020000);
            for (long long q = (long long)blockIdx.x * 1024 + tid; q < nquads; q += step) {
                const unsigned int off = (unsigned int)q * 24u;
                const u32x4v a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                const u32x2v b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0);
                LADD(a.x & 0xFFFFu, 0) LADD(a.x >> 16, 1) LADD(a.y & 0xFFFFu, 2) LADD(a.y >> 16, 0)
                LADD(a.z & 0xFFFFu, 1) LADD(a.z >> 16, 2) LADD(a.w & 0xFFFFu, 0) LADD(a.w >> 16, 1)
                LADD(b.x & 0xFFFFu, 2) LADD(b.x >> 16, 0) LADD(b.y & 0xFFFFu, 1) LADD(b.y >> 16, 2)
            }
        }
        if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
            const uint16_t *p = base + (nquads * 4 + tid) * 3;
            LADD((unsigned)p[0], 0) LADD((unsigned)p[1], 1) LADD((unsigned)p[2], 2)
        }
    } else {
        for (long long i = (long long)blockIdx.x * 1024 + tid; i < npix; i += (long long)gridDim.x * 1024) {
            const uint16_t *p = base + i * channels;
            LADD((unsigned)p[0], 0) LADD((unsigned)p[1], 1) LADD((unsigned)p[2], 2)
        }
    }
#undef LADD
    __syncthreads();
    unsigned int *g = lohist + tile * (3 * 4 * 256);
    for (int i = tid; i < 3 * 4 * 256; i += 1024)
        if (s_h[i]) atomicAdd(&g[i], s_h[i]);
}

#define U16_DEPTH 3                 /* quads per lane in flight in the one-pass count */

// ---- one full pass on VALUE windows (round 5): no histogram per sample at all ---------------------------------------------------
// np.percentile needs, per channel, the order statistics at ranks floor((n - 1) q) and the next one for q = 2 % and 98 %.  A 1/64
// subsample (12-bit histograms: k_u16_sample12) says where in the 16-bit range each mark lies to within a few dozen values
// (k_u16_window: the sample's order statistics six standard deviations of the sampling error either side of the mark, rounded outwards
// to the sample histogram's 16-value bins); the full pass then only has to know, per channel and mark, HOW MANY samples lie below the
// window and the histogram of the samples INSIDE it (k_u16_count_win).  Below: the borrow of one subtraction, added to the lane's
// counter -- v_subrev_co_u32 d, vcc, lo << 16, key; v_addc_co_u32 -- where key is the dword itself for its high sample and the dword
// shifted left by 16 for its low one (no extraction: the other sample sits below the compared bits).  Inside: d < width << 16, true for
// one sample in a thousand, so the block behind it is entered at about one sample position in eight per wave (the candidate BINS of
// round 3-4's pass, 2.3 % of the samples, put some lane of 64 into it at three positions in four: 127 vector instructions per
// wave-quad, 94 of which ran at HBM rate).  A rank that falls outside its window (k_u16_pick_win) flags the tile, and only flagged
// tiles take the two classic radix passes.
#define U16_WIN_MAX 1024            /* values per window: 64 bins of the sample histogram */
#define U16_SAMPLE_BITS 12
#define U16_SAMPLE_BINS 4096        /* v >> 4 */
#define U16_SAMPLE_SPAN 16          /* values per bin of the sample histogram */
struct U16Win { unsigned int lo[2], wd[2]; };     // per (tile, channel): mark m's window = values [lo, lo + wd); wd == 0: none (the tile is flagged)

__global__ __launch_bounds__(1024) void k_u16_sample12(const uint16_t *__restrict__ tiles, long long npix, unsigned int *__restrict__ hist12, int every)
{
    __shared__ unsigned int s_h[3 * U16_SAMPLE_BINS];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * U16_SAMPLE_BINS; i += 1024) s_h[i] = 0;
    __syncthreads();
    const long long tile = blockIdx.y;
    const uint16_t *base = tiles + tile * npix * 3;
    const long long nquads = npix >> 2;
#define SADD(word, half, ch) atomicAdd(&s_h[(ch) * U16_SAMPLE_BINS + (((word) >> ((half) * 16 + 4)) & 0xFFFu)], 1u)
    if (nquads > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)(nquads * 24), 0x00020000);
        // Every `every`-th stretch of 1024 quads of the tile is sampled, whatever the grid: workgroup b takes the sampled stretches b,
        // b + gridDim.x, ... (two in flight per lane; quads past the tile read as zero and are skipped).  A stride that depended on the
        // grid would sample only the head of a tile that has many workgroups to itself.
        const long long nstretch = (nquads + 1023) >> 10;
        const long long nsampled = (nstretch + every - 1) / every;
        for (long long sidx = blockIdx.x; sidx < nsampled; sidx += 2 * (long long)gridDim.x) {
            const long long q = sidx * every * 1024 + tid, q2 = (sidx + gridDim.x) * every * 1024 + tid;
            const unsigned int off = (unsigned int)q * 24u, off2 = (unsigned int)q2 * 24u;
            const bool second = sidx + gridDim.x < nsampled;
            const u32x4v a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            const u32x2v b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0);
            const u32x4v a2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, second ? off2 : off, 0, 0);
            const u32x2v b2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (second ? off2 : off) + 16u, 0, 0);
            if (q < nquads) {
                SADD(a.x, 0, 0); SADD(a.x, 1, 1); SADD(a.y, 0, 2); SADD(a.y, 1, 0); SADD(a.z, 0, 1); SADD(a.z, 1, 2);
                SADD(a.w, 0, 0); SADD(a.w, 1, 1); SADD(b.x, 0, 2); SADD(b.x, 1, 0); SADD(b.y, 0, 1); SADD(b.y, 1, 2);
            }
            if (second && q2 < nquads) {
                SADD(a2.x, 0, 0); SADD(a2.x, 1, 1); SADD(a2.y, 0, 2); SADD(a2.y, 1, 0); SADD(a2.z, 0, 1); SADD(a2.z, 1, 2);
                SADD(a2.w, 0, 0); SADD(a2.w, 1, 1); SADD(b2.x, 0, 2); SADD(b2.x, 1, 0); SADD(b2.y, 0, 1); SADD(b2.y, 1, 2);
            }
        }
    }
#undef SADD
    __syncthreads();
    unsigned int *g = hist12 + tile * (3 * U16_SAMPLE_BINS);
    for (int i = tid; i < 3 * U16_SAMPLE_BINS; i += 1024)
        if (s_h[i]) atomicAdd(&g[i], s_h[i]);
}

// test_wrong (lars_set_tuning("u16_hist_impl", 3)): windows of 64 values at 0, so that (almost) every tile is flagged
__global__ __launch_bounds__(256) void k_u16_window(const unsigned int *__restrict__ hist12, U16Win *__restrict__ win, int test_wrong)
{
    constexpr int PER = U16_SAMPLE_BINS / 256;              // 16 bins per thread
    __shared__ unsigned long long s_scan[256];
    __shared__ unsigned int s_b[2][2];                     // [mark][first, last] bin of the sample histogram
    const int tid = threadIdx.x;
    const long long slot = (long long)blockIdx.y * 3 + blockIdx.x;
    const unsigned int *h = hist12 + slot * U16_SAMPLE_BINS;
    unsigned int mine[PER];
    unsigned long long local = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { mine[j] = h[tid * PER + j]; local += mine[j]; }
    s_scan[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const unsigned long long total = s_scan[255], before = s_scan[tid] - local;
    if (tid < 4) s_b[tid >> 1][tid & 1] = (tid & 1) ? 0u : (unsigned)U16_SAMPLE_BINS;       // first = 4096, last = 0: no window
    __syncthreads();
    if (total > 0) {
        for (int k = 0; k < 2; ++k) {
            // four standard deviations of the sampling error either side of the mark (a miss costs that tile the two classic passes,
            // never a wrong percentile: about one tile in ten thousand)
            const double q = k == 0 ? 0.02 : 0.98;
            const double spread = 4.0 * sqrt((double)total * q * (1.0 - q)) + 2.0;          // ranks of the subsample
            const double centre = (double)(total - 1) * q;
            double rl = floor(centre - spread), rh = ceil(centre + spread + 1.0);          // + 1: the mark's upper neighbour rank
            if (rl < 0.0) rl = 0.0;
            if (rh > (double)(total - 1)) rh = (double)(total - 1);
            const unsigned long long ranks[2] = {(unsigned long long)rl, (unsigned long long)rh};
            for (int e = 0; e < 2; ++e) {
                const unsigned long long r = ranks[e];
                if (local && r >= before && r < before + local) {
                    unsigned long long cum = before;
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        if (r >= cum && r < cum + mine[j]) s_b[k][e] = (unsigned)(tid * PER + j);
                        cum += mine[j];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        U16Win w;
        unsigned int widest = 0;
        for (int k = 0; k < 2; ++k) {
            // one more bin on either side: the subsample's bin edges are the window's edges, and a rank may sit on one
            const int first = (int)s_b[k][0] - 1 < 0 ? 0 : (int)s_b[k][0] - 1;
            const int last = (int)s_b[k][1] + 1 > U16_SAMPLE_BINS - 1 ? U16_SAMPLE_BINS - 1 : (int)s_b[k][1] + 1;
            const int nb = last - first + 1;
            const bool ok = s_b[k][0] < (unsigned)U16_SAMPLE_BINS && nb >= 1 && nb * U16_SAMPLE_SPAN <= U16_WIN_MAX;
            w.lo[k] = ok ? (unsigned)first * U16_SAMPLE_SPAN : 0u;
            w.wd[k] = ok ? (unsigned)nb * U16_SAMPLE_SPAN : 0u;
            if (test_wrong) { w.lo[k] = 0u; w.wd[k] = U16_SAMPLE_SPAN; }
            widest = w.wd[k] > widest ? w.wd[k] : widest;
        }
        // both windows of a channel as wide as the wider one (the counting pass tests "inside either" with ONE compare: min of the two
        // distances against one width); a window without a prediction keeps width 0 and the tile is flagged by the pick
        for (int k = 0; k < 2; ++k)
            if (w.wd[k]) w.wd[k] = widest;
        win[slot] = w;
    }
}

// key: the sample in bits 16..31.  d = key - (lo << 16); the lane's counter takes the borrow (sample < lo): v_subrev_co_u32 + v_addc_co_u32,
// nothing for the scalar unit.  (A scalar count -- s_bcnt1_i32_b64 of the borrow mask + s_add per sample and mark, with a branch per
// sample and mark behind it -- made the pass wait for the CU's one scalar pipe: 150 scalar instructions per wave-quad, 2.46-2.55 ms per 32
// tiles of 8192 x 8192 against 2.66 for round 4's pass.)
__device__ inline unsigned int u16_below(unsigned int key, unsigned int lo16, unsigned int &count)
{
    unsigned int d;
    asm volatile("v_subrev_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "=&v"(d), "+v"(count) : "s"(lo16), "v"(key) : "vcc");
    return d;
}

__global__ __launch_bounds__(1024) void k_u16_count_win(const uint16_t *__restrict__ tiles, long long npix, const U16Win *__restrict__ win,
                                                        unsigned int *__restrict__ below_out /*[ntiles][3][2]*/,
                                                        unsigned int *__restrict__ winhist /*[ntiles][3][2][U16_WIN_MAX]*/)
{
    __shared__ unsigned int s_w[3 * 2 * U16_WIN_MAX];          // 24 KiB
    __shared__ unsigned int s_below[6];
    const int tid = threadIdx.x;
    const long long tile = blockIdx.y;
    for (int i = tid; i < 3 * 2 * U16_WIN_MAX; i += 1024) s_w[i] = 0;
    if (tid < 6) s_below[tid] = 0;
    __syncthreads();
    // the windows, wave-uniform: lo << 16 per channel and mark, one width << 16 per channel (k_u16_window made a channel's two alike)
    unsigned int lo16[3][2], wd16[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const U16Win w = win[tile * 3 + c];
        lo16[c][0] = __builtin_amdgcn_readfirstlane(w.lo[0] << 16);
        lo16[c][1] = __builtin_amdgcn_readfirstlane(w.lo[1] << 16);
        wd16[c] = __builtin_amdgcn_readfirstlane((w.wd[0] > w.wd[1] ? w.wd[0] : w.wd[1]) << 16);          // <= 1024 << 16
    }
    // a mark without a window (width 0) must not catch samples through the other mark's width
    const bool no_win[3][2] = {{win[tile * 3].wd[0] == 0, win[tile * 3].wd[1] == 0}, {win[tile * 3 + 1].wd[0] == 0, win[tile * 3 + 1].wd[1] == 0},
                               {win[tile * 3 + 2].wd[0] == 0, win[tile * 3 + 2].wd[1] == 0}};
    unsigned int below[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};                     // per lane
    const uint16_t *base = tiles + tile * npix * 3;
    const long long nquads = npix >> 2;
    const long long step = (long long)gridDim.x * 1024;
    auto one_sample = [&](unsigned int key, int ch) {
        const unsigned int d0 = u16_below(key, lo16[ch][0], below[ch][0]);
        const unsigned int d1 = u16_below(key, lo16[ch][1], below[ch][1]);
        if (min(d0, d1) < wd16[ch]) {                                              // inside either window: about one sample in a thousand
            if (d0 < wd16[ch] && !no_win[ch][0]) atomicAdd(&s_w[(ch * 2 + 0) * U16_WIN_MAX + (d0 >> 16)], 1u);
            if (d1 < wd16[ch] && !no_win[ch][1]) atomicAdd(&s_w[(ch * 2 + 1) * U16_WIN_MAX + (d1 >> 16)], 1u);
        }
    };
    if (nquads > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)(nquads * 24), 0x00020000);
        u32x4v ra[U16_DEPTH];
        u32x2v rb[U16_DEPTH];
        long long q = (long long)blockIdx.x * 1024 + tid;
#pragma unroll
        for (int d = 0; d < U16_DEPTH; ++d) {
            const unsigned int off = (unsigned int)(q + d * step) * 24u;
            ra[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            rb[d] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0);
        }
        for (; q < nquads; q += U16_DEPTH * step) {
#pragma unroll
            for (int d = 0; d < U16_DEPTH; ++d) {
                const long long qq = q + d * step;
                const u32x4v a = ra[d];
                const u32x2v b = rb[d];
                const unsigned int off = (unsigned int)(qq + U16_DEPTH * step) * 24u;
                ra[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);      // past the tile: zeros, never counted
                rb[d] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0);
                if (qq < nquads) {
                    const unsigned int w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
                    for (int i = 0; i < 12; ++i) one_sample((i & 1) ? w[i >> 1] : (w[i >> 1] << 16), i % 3);
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
        const uint16_t *p = base + (nquads * 4 + tid) * 3;
        for (int ch = 0; ch < 3; ++ch) one_sample((unsigned int)p[ch] << 16, ch);
    }
    // the lanes' six counts -> the wave's (shuffles) -> the block's -> the tile's
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            unsigned int v = below[c][m];
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if ((tid & 63) == 0 && v) atomicAdd(&s_below[c * 2 + m], v);
        }
    __syncthreads();
    if (tid < 6 && s_below[tid]) atomicAdd(&below_out[tile * 6 + tid], s_below[tid]);
    unsigned int *g = winhist + tile * (3 * 2 * U16_WIN_MAX);
    for (int i = tid; i < 3 * 2 * U16_WIN_MAX; i += 1024)
        if (s_w[i]) atomicAdd(&g[i], s_w[i]);
}

// ranks -> values -> percentiles -> table, or the tile's flag.  One block per (channel, tile); a flagged tile keeps whatever the other
// channels wrote: the classic passes overwrite all three.
__global__ __launch_bounds__(256) void k_u16_pick_win(const U16Win *__restrict__ win, const unsigned int *__restrict__ below_in,
                                                      const unsigned int *__restrict__ winhist, long long npix, unsigned int *__restrict__ flags,
                                                      uint8_t *__restrict__ blobs, double *__restrict__ pcts, int rgn_variant)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ double s_val[4];
    __shared__ int s_found[4];
    __shared__ double s_p[2];
    __shared__ unsigned int s_thr[260];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const long long tile = blockIdx.y;
    const long long slot = tile * 3 + c;
    const U16Win w = win[slot];
    if (tid < 4) s_found[tid] = 0;
    __syncthreads();
    const double nm1 = (double)(npix - 1);
    double tq[2];
    for (int k = 0; k < 2; ++k) {
        const double q = (k == 0 ? 2.0 : 98.0) / 100.0;
        const double vi = nm1 * q;
        const double fl = floor(vi);
        tq[k] = vi - fl;
        long long lo = (long long)fl, hi = lo + 1;
        if (hi > npix - 1) hi = npix - 1;
        const unsigned long long below = below_in[tile * 6 + c * 2 + k];
        const unsigned int *h = winhist + ((tile * 3 + c) * 2 + k) * U16_WIN_MAX;
        unsigned int mine[4];
        unsigned long long local = 0;
        for (int j = 0; j < 4; ++j) { mine[j] = h[tid * 4 + j]; local += mine[j]; }
        __syncthreads();
        s_scan[tid] = local;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            unsigned long long v = (tid >= off) ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const unsigned long long before = below + s_scan[tid] - local;
        const long long rank[2] = {lo, hi};
        for (int j2 = 0; j2 < 2; ++j2) {
            const unsigned long long r = (unsigned long long)rank[j2];
            if (local && r >= before && r < before + local) {
                unsigned long long cum = before;
                for (int j = 0; j < 4; ++j) {
                    if (r >= cum && r < cum + mine[j]) { s_val[2 * k + j2] = (double)(w.lo[k] + (unsigned)(tid * 4 + j)); s_found[2 * k + j2] = 1; }
                    cum += mine[j];
                }
            }
        }
    }
    __syncthreads();
    if (!(s_found[0] && s_found[1] && s_found[2] && s_found[3])) {
        if (tid == 0) flags[tile] = 1u;                    // a rank outside its window (or no window): the classic passes take this tile
        return;
    }
    if (tid < 2) {
        const double a = s_val[2 * tid], b = s_val[2 * tid + 1], t = tq[tid];
        const double d = b - a;
        double r = a + d * t;
        if (t >= 0.5) r = b - d * (1.0 - t);
        s_p[tid] = r;
        if (pcts) pcts[slot * 2 + tid] = r;
    }
    __syncthreads();
    u16_fill_blob(blobs + tile * LARS_U16_BLOB_BYTES, c, s_p[0], s_p[1], rgn_variant, s_thr, tid);
}

// ---- table: order statistics -> percentiles -> 65536-entry table + thresholds ----------------
// nslots: low-byte histograms per channel in lohist (4: one per order statistic; ranks that share a high byte share a slot)
__global__ __launch_bounds__(256) void k_wb_table_u16(const unsigned int *__restrict__ lohist, const U16Pick *__restrict__ picks,
                                                      uint8_t *__restrict__ blobs, double *__restrict__ pcts, int rgn_variant, int nslots,
                                                      const unsigned int *__restrict__ only = nullptr)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ double s_val[4];
    __shared__ double s_p[2];
    __shared__ unsigned int s_thr[260];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const long long tile = blockIdx.y;
    if (only && !only[tile]) return;
    const long long slot = tile * 3 + c;
    const U16Pick pk = picks[slot];

    for (int r = 0; r < 4; ++r) {
        // scan the slot of rank r (ranks sharing a high byte rescan the same slot)
        const unsigned int *h = lohist + ((tile * 3 + c) * nslots + pk.slot[r]) * 256;
        const unsigned long long cnt = h[tid];
        __syncthreads();
        s_scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            unsigned long long v = (tid >= off) ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const unsigned long long before = s_scan[tid] - cnt;
        if (cnt && pk.resid[r] >= before && pk.resid[r] < before + cnt) s_val[r] = (double)(pk.target[r] * 256u + (unsigned)tid);
    }
    __syncthreads();
    if (tid < 2) {
        const double a = s_val[2 * tid], b = s_val[2 * tid + 1], t = pk.tq[tid];
        const double d = b - a;
        double r = a + d * t;
        if (t >= 0.5) r = b - d * (1.0 - t);
        s_p[tid] = r;
        if (pcts) pcts[slot * 2 + tid] = r;
    }
    __syncthreads();
    u16_fill_blob(blobs + tile * LARS_U16_BLOB_BYTES, c, s_p[0], s_p[1], rgn_variant, s_thr, tid);
}

}  // namespace lars

using namespace lars;

extern "C" size_t lars_wb_table_bytes(int dtype)
{
    return dtype == LARS_U16 ? (size_t)LARS_U16_BLOB_BYTES : (size_t)768;
}

// One call for the whole white-balance pre-pass of a batch, both sample types.
extern "C" int lars_d_wb_prepare(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype, uint8_t *table,
                                 double *percentiles, int rgn_variant, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || !table || ntiles <= 0 || npix <= 0 || channels < 3 || ntiles > 65535)
        return fail(LARS_ERR_INVALID, "lars_d_wb_prepare: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    if (dtype == LARS_U8) {
        LARS_TRY(scratch_reserve(c, (size_t)ntiles * 768 * sizeof(uint32_t)));
        uint32_t *hist = static_cast<uint32_t *>(c->scratch);
        LARS_TRY(lars_d_channel_hist(tiles, ntiles, npix, channels, dtype, hist, s));
        return lars_d_wb_table(hist, ntiles, npix, dtype, table, percentiles, rgn_variant, s);
    }
    if (dtype != LARS_U16) return fail(LARS_ERR_INVALID, "lars_d_wb_prepare: dtype");
    const size_t hi_bytes = (size_t)ntiles * 768 * 4, pick_bytes = (size_t)ntiles * 3 * sizeof(U16Pick),
                 lo_bytes = (size_t)ntiles * 3 * 4 * 256 * 4,
                 flag_bytes = (size_t)ntiles * 4, s12_bytes = (size_t)ntiles * 3 * U16_SAMPLE_BINS * 4,
                 below_bytes = (size_t)ntiles * 6 * 4, wh_bytes = (size_t)ntiles * 3 * 2 * U16_WIN_MAX * 4, win_bytes = (size_t)ntiles * 3 * sizeof(U16Win);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    LARS_TRY(scratch_reserve(c, up(hi_bytes) + up(pick_bytes) + up(lo_bytes) + up(flag_bytes) + up(s12_bytes) + up(below_bytes) +
                                    up(wh_bytes) + up(win_bytes) + 512));
    char *p = static_cast<char *>(c->scratch);
    unsigned int *hi = reinterpret_cast<unsigned int *>(p);                  p += up(hi_bytes);
    unsigned int *lo = reinterpret_cast<unsigned int *>(p);                  p += up(lo_bytes);
    unsigned int *flags = reinterpret_cast<unsigned int *>(p);               p += up(flag_bytes);
    unsigned int *s12 = reinterpret_cast<unsigned int *>(p);                 p += up(s12_bytes);
    unsigned int *below = reinterpret_cast<unsigned int *>(p);               p += up(below_bytes);
    unsigned int *winhist = reinterpret_cast<unsigned int *>(p);             p += up(wh_bytes);
    U16Pick *picks = reinterpret_cast<U16Pick *>(p);                         p += up(pick_bytes);
    U16Win *win = reinterpret_cast<U16Win *>(p);
    const uint16_t *t16 = static_cast<const uint16_t *>(tiles);
    const bool fast = channels == 3 && (ntiles == 1 || (npix & 3) == 0) && ((reinterpret_cast<uintptr_t>(tiles) & 3) == 0) &&
                      (long long)npix * 6 < (1ll << 30);
    long long want = (1024 + ntiles - 1) / ntiles;
    const long long cap = (npix / 4 + 1023) / 1024;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    dim3 grid((unsigned)want, (unsigned)ntiles);
    const dim3 per_channel(3, (unsigned)ntiles);
    const int impl = tuning().u16_hist_impl;
    if (fast && cap >= 64 && impl != 1) {
        // ONE full pass on value windows (round 5): per channel and mark the count of the samples below a predicted window and the histogram
        // inside it; a rank outside its window flags the tile, and only flagged tiles take the two classic passes (all of them with impl 3)
        LARS_HIP_TRY(hipMemsetAsync(hi, 0, up(hi_bytes) + up(lo_bytes) + up(flag_bytes) + up(s12_bytes) + up(below_bytes) + up(wh_bytes), s));
        hipLaunchKernelGGL(k_u16_sample12, grid, dim3(1024), 0, s, t16, (long long)npix, s12, 64);     // 1 / 64 of the samples: 1 M per 8192 x 8192 tile and channel
        hipLaunchKernelGGL(k_u16_window, per_channel, dim3(256), 0, s, s12, win, impl == 3 ? 1 : 0);
        hipLaunchKernelGGL(k_u16_count_win, grid, dim3(1024), 0, s, t16, (long long)npix, win, below, winhist);
        hipLaunchKernelGGL(k_u16_pick_win, per_channel, dim3(256), 0, s, win, below, winhist, (long long)npix, flags, table, percentiles, rgn_variant);
        hipLaunchKernelGGL(k_hist_u16_hi, grid, dim3(1024), 0, s, t16, (long long)npix, hi, 1, flags);
        hipLaunchKernelGGL(k_u16_pick, per_channel, dim3(256), 0, s, hi, (long long)npix, picks, flags);
        hipLaunchKernelGGL((k_hist_u16_lo<true>), grid, dim3(1024), 0, s, t16, (long long)npix, channels, picks, lo, flags);
        hipLaunchKernelGGL(k_wb_table_u16, per_channel, dim3(256), 0, s, lo, picks, table, percentiles, rgn_variant, 4, flags);
        return launch_check("lars_d_wb_prepare (value windows)");
    }
    LARS_HIP_TRY(hipMemsetAsync(hi, 0, up(hi_bytes) + up(lo_bytes) + up(flag_bytes), s));     // hi, lo, flags are contiguous
    if (fast) hipLaunchKernelGGL(k_hist_u16_hi, grid, dim3(1024), 0, s, t16, (long long)npix, hi, 1, (const unsigned int *)nullptr);
    else hipLaunchKernelGGL(k_hist_u16_hi_generic, dim3((unsigned)want * 4, (unsigned)ntiles), dim3(256), 0, s, t16, (long long)npix, channels, hi);
    hipLaunchKernelGGL(k_u16_pick, per_channel, dim3(256), 0, s, hi, (long long)npix, picks, (const unsigned int *)nullptr);
    if (fast) hipLaunchKernelGGL((k_hist_u16_lo<true>), grid, dim3(1024), 0, s, t16, (long long)npix, channels, picks, lo, (const unsigned int *)nullptr);
    else hipLaunchKernelGGL((k_hist_u16_lo<false>), grid, dim3(1024), 0, s, t16, (long long)npix, channels, picks, lo, (const unsigned int *)nullptr);
    hipLaunchKernelGGL(k_wb_table_u16, per_channel, dim3(256), 0, s, lo, picks, table, percentiles, rgn_variant, 4, (const unsigned int *)nullptr);
    return launch_check("lars_d_wb_prepare");
}
