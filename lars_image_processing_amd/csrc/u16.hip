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
// Round 3: usually ONE full pass.  A 1/16 subsample of the high-byte histograms predicts, per channel, the bins that will
// hold the order statistics (each predicted bin and its two neighbours: up to six candidates); the full pass then counts the
// high bytes of every sample AND the low bytes of the samples in candidate bins (k_hist_u16_both).  The exact high-byte
// histogram settles which bins were needed; a tile whose bins were not all among its candidates (a percentile within a sample
// error of a bin boundary, or a 16-bit image that is not smooth at the 2 % / 98 % marks) is recounted by the classic second pass.
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
__global__ __launch_bounds__(1024) void k_hist_u16_hi(const uint16_t *__restrict__ tiles, long long npix,
                                                      unsigned int *__restrict__ hist, int every)
{
    __shared__ unsigned int s_h[3 * 256 * 32];             // [channel][bin][copy = lane % 32]
    const int tid = threadIdx.x;
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
                                                  U16Pick *__restrict__ picks)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ U16Pick s_pick;
    const int tid = threadIdx.x;
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
                                                      const U16Pick *__restrict__ picks, unsigned int *__restrict__ lohist)
{
    __shared__ unsigned int s_h[3 * 4 * 256];
    __shared__ unsigned char s_slot[3 * 256];
    const int tid = threadIdx.x;
    const long long tile = blockIdx.y;
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

// ---- one full pass: candidate bins from a subsample, high bytes + low bytes of the candidates together --------------------
#define U16_CAND 6                  /* candidate high-byte bins per channel: each predicted bin and its two neighbours */
#define U16_DEPTH 3                 /* quads per lane in flight in the one-pass count */
struct U16Cand { unsigned char bin[U16_CAND]; unsigned char n; unsigned char pad; };

// From the subsample's high-byte histogram: the bin holding the 2 % resp. the 98 % mark of the sample, and a neighbour only where the
// mark sits within six standard deviations of its sampling error from that side of its bin (sigma = sqrt(n q (1 - q)) ranks of the
// subsample, as a fraction of the bin's count).  Round 3 took both neighbours always: six candidate bins per channel = 2.3 % of random
// samples, and the one-pass count -- bound by instruction issue -- pays for every sample position at which ANY lane of a wave holds a
// candidate (profiles/r04_u16_prepass_counters.txt).  A mark that lands outside its candidates costs that tile a recount, never a wrong
// percentile: the exact high-byte histogram of the full pass decides (k_u16_resolve).
__global__ __launch_bounds__(256) void k_u16_predict(const unsigned int *__restrict__ sample_hist, U16Cand *__restrict__ cand)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ int s_t[2];
    __shared__ int s_side[2][2];                           // [mark][0: take the bin below, 1: the bin above]
    const int tid = threadIdx.x;
    const long long slot = (long long)blockIdx.y * 3 + blockIdx.x;
    const unsigned long long c = sample_hist[slot * 256 + tid];
    s_scan[tid] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const unsigned long long total = s_scan[255], before = s_scan[tid] - c;
    if (tid < 2) { s_t[tid] = tid == 0 ? 0 : 255; s_side[tid][0] = 1; s_side[tid][1] = 1; }
    __syncthreads();
    if (total > 0) {
        for (int k = 0; k < 2; ++k) {
            const double q = k == 0 ? 0.02 : 0.98;
            const unsigned long long r = (unsigned long long)((double)(total - 1) * q);
            if (c && r >= before && r < before + c) {
                s_t[k] = tid;
                const double margin = 6.0 * sqrt((double)total * q * (1.0 - q)) + 2.0;        // ranks of the subsample
                s_side[k][0] = (double)(r - before) < margin ? 1 : 0;
                s_side[k][1] = (double)(before + c - 1 - r) < margin ? 1 : 0;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        U16Cand out;
        out.n = 0; out.pad = 0;
        for (int j = 0; j < U16_CAND; ++j) out.bin[j] = 0;
        for (int k = 0; k < 2; ++k)
            for (int d = -1; d <= 1; ++d) {
                if ((d < 0 && !s_side[k][0]) || (d > 0 && !s_side[k][1])) continue;
                const int b = s_t[k] + d;
                if (b < 0 || b > 255) continue;
                bool have = false;
                for (int j = 0; j < out.n; ++j) have |= out.bin[j] == (unsigned char)b;
                if (!have) out.bin[out.n++] = (unsigned char)b;
            }
        // ascending order: the full pass reads the list as two windows of consecutive bins
        for (int i = 1; i < out.n; ++i)
            for (int j = i; j > 0 && out.bin[j] < out.bin[j - 1]; --j) { const unsigned char t = out.bin[j]; out.bin[j] = out.bin[j - 1]; out.bin[j - 1] = t; }
        cand[slot] = out;
    }
}

// test hook (lars_set_tuning("u16_hist_impl", 3)): every channel's candidates = {bin 0}, so that (almost) every tile misses and
// takes the recount
__global__ void k_u16_spoil(U16Cand *cand, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    U16Cand c;
    c.n = 1; c.pad = 0;
    for (int j = 0; j < U16_CAND; ++j) c.bin[j] = 0;
    cand[i] = c;
}

// ONLY_FLAGGED: the recount of the tiles whose candidates missed (low bytes only, candidates = the exact bins by then)
template <bool ONLY_FLAGGED>
__global__ __launch_bounds__(1024) void k_hist_u16_both(const uint16_t *__restrict__ tiles, long long npix, const U16Cand *__restrict__ cand,
                                                        const unsigned int *__restrict__ flags, unsigned int *__restrict__ hist,
                                                        unsigned int *__restrict__ lohist, int allow_windows)
{
    __shared__ unsigned int s_h[ONLY_FLAGGED ? 32 : 3 * 256 * 32];       // high bytes: [channel][bin][copy = lane % 32]
    __shared__ unsigned int s_lo[3 * U16_CAND * 256];
    __shared__ unsigned char s_slot[3 * 256];
    // The candidates of a channel as two windows of consecutive bins [A0, A0 + la) and [B0, B0 + lb), slots 0 .. la - 1 and la .. la + lb - 1:
    // what k_u16_predict produces (each predicted bin and its neighbours).  A sample is then tested with two subtractions and two compares
    // in registers instead of a byte look-up in LDS per sample (twelve random ds_read_u8 per quad were 70 % of this kernel's LDS time).
    // s_win[c] = {A0, la, B0, lb}; s_windows = 0 if some channel's list is not of that form (the recount's exact bins need not be):
    // then the look-up table decides, as before.
    __shared__ unsigned int s_win[3][4];
    __shared__ int s_windows;
    const int tid = threadIdx.x;
    const long long tile = blockIdx.y;
    if (ONLY_FLAGGED && !flags[tile]) return;
    if (!ONLY_FLAGGED)
        for (int i = tid; i < 3 * 256 * 32; i += 1024) s_h[i] = 0;
    for (int i = tid; i < 3 * U16_CAND * 256; i += 1024) s_lo[i] = 0;
    if (tid < 768) s_slot[tid] = 0xFF;
    if (tid == 0) s_windows = (ONLY_FLAGGED || !allow_windows) ? 0 : 1;
    __syncthreads();
    if (tid < 3 * U16_CAND) {
        const int c = tid / U16_CAND, j = tid % U16_CAND;
        const U16Cand cd = cand[tile * 3 + c];
        if (j < cd.n) s_slot[c * 256 + cd.bin[j]] = (unsigned char)j;
        if (!ONLY_FLAGGED && j == 0) {
            int la = 1;
            while (la < cd.n && cd.bin[la] == cd.bin[0] + la) ++la;
            int lb = 0;
            while (la + lb < cd.n && (lb == 0 || cd.bin[la + lb] == cd.bin[la] + lb)) ++lb;
            if (cd.n < 1 || la + lb != cd.n) atomicExch(&s_windows, 0);
            s_win[c][0] = cd.n ? cd.bin[0] : 0u; s_win[c][1] = cd.n ? (unsigned)la : 0u;
            s_win[c][2] = lb ? cd.bin[la] : 0u;  s_win[c][3] = (unsigned)lb;
        }
    }
    __syncthreads();
    const uint16_t *base = tiles + tile * npix * 3;
    const long long nquads = npix >> 2;
    const unsigned int lane_off = (tid & 31) << 2;
    char *hb = reinterpret_cast<char *>(s_h);
#define BADD(sample, ch)                                                                               \
    {                                                                                                  \
        const unsigned int s_ = (sample);                                                              \
        if (!ONLY_FLAGGED)                                                                             \
            atomicAdd(reinterpret_cast<unsigned int *>(hb + (ch) * 32768 + ((s_ >> 8) << 7) + lane_off), 1u); \
        const unsigned int sl_ = s_slot[(ch) * 256 + (s_ >> 8)];                                       \
        if (sl_ != 0xFFu) atomicAdd(&s_lo[((ch) * U16_CAND + sl_) * 256 + (s_ & 0xFFu)], 1u);         \
    }
    const long long step = (long long)gridDim.x * 1024;
    if (nquads > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base), 0, (int)(nquads * 24), 0x00020000);
        // U16_DEPTH quads of every lane in flight: with one, a CU's 16 waves keep 24 KB on the way, which bounds the pass
        // at about half of what HBM delivers.  Offsets past the tile (the ring's last turns) read as zero and are not counted.
        u32x4v ra[U16_DEPTH];
        u32x2v rb[U16_DEPTH];
        // the windows, wave-uniform (scalar registers)
        const bool use_windows = !ONLY_FLAGGED && __builtin_amdgcn_readfirstlane(s_windows) != 0;
        unsigned int wA0[3], wla[3], wB0[3], wlb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            wA0[c] = __builtin_amdgcn_readfirstlane(s_win[c][0]); wla[c] = __builtin_amdgcn_readfirstlane(s_win[c][1]);
            wB0[c] = __builtin_amdgcn_readfirstlane(s_win[c][2]); wlb[c] = __builtin_amdgcn_readfirstlane(s_win[c][3]);
        }
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
                ra[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                rb[d] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0);
                if (qq < nquads && use_windows) {
                    const unsigned int w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int ch = i % 3;
                        const unsigned int hi = (w[i >> 1] >> ((i & 1) * 16 + 8)) & 0xFFu;
                        atomicAdd(reinterpret_cast<unsigned int *>(hb + ch * 32768 + (hi << 7) + lane_off), 1u);
                        const unsigned int ra = hi - wA0[ch], rb = hi - wB0[ch];
                        if ((ra < wla[ch]) | (rb < wlb[ch])) {
                            const unsigned int slot = ra < wla[ch] ? ra : wla[ch] + rb;
                            atomicAdd(&s_lo[(ch * U16_CAND + slot) * 256 + ((w[i >> 1] >> ((i & 1) * 16)) & 0xFFu)], 1u);
                        }
                    }
                } else if (qq < nquads) {
                    // all twelve slot look-ups first, then the twelve counts, then the (rare) candidates: one LDS round trip
                    // per quad instead of one per sample
                    const unsigned int w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
                    unsigned int sl[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        sl[i] = s_slot[(i % 3) * 256 + ((w[i >> 1] >> ((i & 1) * 16 + 8)) & 0xFFu)];
                    if (!ONLY_FLAGGED) {
#pragma unroll
                        for (int i = 0; i < 12; ++i)
                            atomicAdd(reinterpret_cast<unsigned int *>(hb + (i % 3) * 32768 + (((w[i >> 1] >> ((i & 1) * 16 + 8)) & 0xFFu) << 7) + lane_off), 1u);
                    }
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        if (sl[i] != 0xFFu)
                            atomicAdd(&s_lo[((i % 3) * U16_CAND + sl[i]) * 256 + ((w[i >> 1] >> ((i & 1) * 16)) & 0xFFu)], 1u);
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
        const uint16_t *p = base + (nquads * 4 + tid) * 3;
        BADD((unsigned)p[0], 0) BADD((unsigned)p[1], 1) BADD((unsigned)p[2], 2)
    }
#undef BADD
    __syncthreads();
    if (!ONLY_FLAGGED && tid < 768) {
        const unsigned int *row = s_h + tid * 32;
        unsigned int v = 0;
        for (int j = 0; j < 32; ++j) v += row[(j + tid) & 31];
        if (v) atomicAdd(&hist[tile * 768 + tid], v);
    }
    unsigned int *g = lohist + tile * (3 * U16_CAND * 256);
    for (int i = tid; i < 3 * U16_CAND * 256; i += 1024)
        if (s_lo[i]) atomicAdd(&g[i], s_lo[i]);
}

// After the exact pick: point each rank at its candidate's low-byte histogram; a tile with a rank outside its candidates is
// flagged, gets the exact bins as its candidates and its low-byte histograms zeroed (the recount follows).  One block per tile.
__global__ __launch_bounds__(256) void k_u16_resolve(U16Pick *__restrict__ picks, U16Cand *__restrict__ cand, unsigned int *__restrict__ lohist,
                                                     unsigned int *__restrict__ flags)
{
    __shared__ int s_miss;
    const int tid = threadIdx.x;
    const long long tile = blockIdx.x;
    if (tid == 0) s_miss = 0;
    __syncthreads();
    if (tid < 3) {
        U16Pick p = picks[tile * 3 + tid];
        const U16Cand cd = cand[tile * 3 + tid];
        bool miss = false;
        for (int r = 0; r < 4; ++r) {
            int j = 0;
            for (; j < cd.n; ++j)
                if (cd.bin[j] == p.target[r]) break;
            if (j == cd.n) miss = true;
            p.slot[r] = (unsigned)j;
        }
        if (miss) atomicExch(&s_miss, 1);
        else picks[tile * 3 + tid] = p;
    }
    __syncthreads();
    if (!s_miss) {
        if (tid == 0) flags[tile] = 0;
        return;
    }
    if (tid < 3) {
        U16Pick p = picks[tile * 3 + tid];
        U16Cand cd;
        cd.n = 0; cd.pad = 0;
        for (int j = 0; j < U16_CAND; ++j) cd.bin[j] = 0;
        for (int r = 0; r < 4; ++r) {
            int j = 0;
            for (; j < cd.n; ++j)
                if (cd.bin[j] == p.target[r]) break;
            if (j == cd.n) cd.bin[cd.n++] = (unsigned char)p.target[r];
            p.slot[r] = (unsigned)j;
        }
        cand[tile * 3 + tid] = cd;
        picks[tile * 3 + tid] = p;
    }
    unsigned int *g = lohist + tile * (3 * U16_CAND * 256);
    for (int i = tid; i < 3 * U16_CAND * 256; i += 256) g[i] = 0;
    if (tid == 0) flags[tile] = 1;
}

// ---- table: order statistics -> percentiles -> 65536-entry table + thresholds ----------------
// nslots: low-byte histograms per channel in lohist (4: the classic second pass, U16_CAND: the candidate layout)
__global__ __launch_bounds__(256) void k_wb_table_u16(const unsigned int *__restrict__ lohist, const U16Pick *__restrict__ picks,
                                                      uint8_t *__restrict__ blobs, double *__restrict__ pcts, int rgn_variant, int nslots)
{
    __shared__ unsigned long long s_scan[256];
    __shared__ double s_val[4];
    __shared__ double s_p[2];
    __shared__ unsigned int s_thr[260];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const long long tile = blockIdx.y;
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
                 lo_bytes = (size_t)ntiles * 3 * U16_CAND * 256 * 4, cand_bytes = (size_t)ntiles * 3 * sizeof(U16Cand),
                 flag_bytes = (size_t)ntiles * 4;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    LARS_TRY(scratch_reserve(c, 2 * up(hi_bytes) + up(pick_bytes) + up(lo_bytes) + up(cand_bytes) + up(flag_bytes) + 512));
    char *p = static_cast<char *>(c->scratch);
    unsigned int *hi = reinterpret_cast<unsigned int *>(p);                  p += up(hi_bytes);
    unsigned int *hi_sample = reinterpret_cast<unsigned int *>(p);           p += up(hi_bytes);
    unsigned int *lo = reinterpret_cast<unsigned int *>(p);                  p += up(lo_bytes);
    unsigned int *flags = reinterpret_cast<unsigned int *>(p);               p += up(flag_bytes);
    U16Pick *picks = reinterpret_cast<U16Pick *>(p);                         p += up(pick_bytes);
    U16Cand *cand = reinterpret_cast<U16Cand *>(p);
    LARS_HIP_TRY(hipMemsetAsync(hi, 0, 2 * up(hi_bytes) + up(lo_bytes) + up(flag_bytes), s));     // hi, hi_sample, lo, flags are contiguous
    const uint16_t *t16 = static_cast<const uint16_t *>(tiles);
    const bool fast = channels == 3 && (ntiles == 1 || (npix & 3) == 0) && ((reinterpret_cast<uintptr_t>(tiles) & 3) == 0) &&
                      (long long)npix * 6 < (1ll << 30);
    long long want = (1024 + ntiles - 1) / ntiles;
    const long long cap = (npix / 4 + 1023) / 1024;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    dim3 grid((unsigned)want, (unsigned)ntiles);
    const dim3 per_channel(3, (unsigned)ntiles);
    if (fast && tuning().u16_hist_impl != 1 && cap >= 64) {
        // ONE full pass: candidates from a 1/16 subsample, then high bytes + low bytes of the candidate bins together; the exact
        // pick decides which tiles (if any) need the classic recount
        hipLaunchKernelGGL(k_hist_u16_hi, grid, dim3(1024), 0, s, t16, (long long)npix, hi_sample, 16);
        hipLaunchKernelGGL(k_u16_predict, per_channel, dim3(256), 0, s, hi_sample, cand);
        if (tuning().u16_hist_impl == 3)
            hipLaunchKernelGGL(k_u16_spoil, dim3((unsigned)((ntiles * 3 + 255) / 256)), dim3(256), 0, s, cand, (long long)ntiles * 3);
        // u16_hist_impl 4: the same pass with a slot look-up in LDS per sample instead of the window tests in registers (round 3's form)
        hipLaunchKernelGGL((k_hist_u16_both<false>), grid, dim3(1024), 0, s, t16, (long long)npix, cand, flags, hi, lo, tuning().u16_hist_impl == 4 ? 0 : 1);
        hipLaunchKernelGGL(k_u16_pick, per_channel, dim3(256), 0, s, hi, (long long)npix, picks);
        hipLaunchKernelGGL(k_u16_resolve, dim3((unsigned)ntiles), dim3(256), 0, s, picks, cand, lo, flags);
        hipLaunchKernelGGL((k_hist_u16_both<true>), grid, dim3(1024), 0, s, t16, (long long)npix, cand, flags, hi, lo, 0);
        hipLaunchKernelGGL(k_wb_table_u16, per_channel, dim3(256), 0, s, lo, picks, table, percentiles, rgn_variant, U16_CAND);
        return launch_check("lars_d_wb_prepare (one pass)");
    }
    if (fast) hipLaunchKernelGGL(k_hist_u16_hi, grid, dim3(1024), 0, s, t16, (long long)npix, hi, 1);
    else hipLaunchKernelGGL(k_hist_u16_hi_generic, dim3((unsigned)want * 4, (unsigned)ntiles), dim3(256), 0, s, t16, (long long)npix, channels, hi);
    hipLaunchKernelGGL(k_u16_pick, per_channel, dim3(256), 0, s, hi, (long long)npix, picks);
    if (fast) hipLaunchKernelGGL((k_hist_u16_lo<true>), grid, dim3(1024), 0, s, t16, (long long)npix, channels, picks, lo);
    else hipLaunchKernelGGL((k_hist_u16_lo<false>), grid, dim3(1024), 0, s, t16, (long long)npix, channels, picks, lo);
    hipLaunchKernelGGL(k_wb_table_u16, per_channel, dim3(256), 0, s, lo, picks, table, percentiles, rgn_variant, 4);
    return launch_check("lars_d_wb_prepare");
}
