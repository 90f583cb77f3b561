// Exact medians without index planes: radix select on values recomputed from the tiles.
// The first (bucket) pass also exists inside the statistics kernel (fused_v2.hip, SEL).
#include <string.h>

#include <type_traits>

#include "v2_device.h"

namespace lars {

// ---------------------------------------------------------------------------
// Exact medians without materialising the index planes -- of every tile of a batch, or of the whole batch over
// all ranks.  Radix select on values that are RECOMPUTED from the tiles (3 bytes per pixel and
// pass).  Two streams (NDVI, GNDVI; NDWI = -GNDVI shares GNDVI's order statistics) x two tracks (the ranks
// (N-1)/2 and N/2, which may part ways).
//   pass 1   2048 linear buckets of [-1, 1] (selq_bucket: monotone in x, so it is a valid first radix level, and
//            unlike the key's top bits it spreads an index plane over hundreds of LDS words)
//   pick     bucket holding the rank -> its key range [lo, hi) by bisection with the same arithmetic
//   pass 2+  digit d = (key - bias) >> shift of the keys inside the range (anything else lands on a per-lane dummy
//            word: d is compared with the row's usable length by one v_min, no branch); the first digit takes whatever
//            shift makes it fit 1984 bins, then the shift drops by 10 per pass until 0.
// uint8 quotients are 0 or at least 1/510 in magnitude, so a bucket is at most 2^22 keys wide (three passes, four for
// medians below 2^-7 in magnitude) once the bucket around zero is cut down to the single key of +0.0.
// LDS: the 64 KiB white-balance table + one 2048-word row per stream = exactly 80 KiB, two blocks per CU.
// ---------------------------------------------------------------------------
#define SELQ_DIGITS 1984                                  /* digits per row in the later passes; words 1984..2047 are per-lane dummies */
#define SELQ_DIGIT_BITS 10                                /* 2^10 <= SELQ_DIGITS: a picked bin splits into at most that many */
#define SELQ_KEY_MINUS1 0x407FFFFFu                        /* f32_key(-1.0f) */
#define SELQ_KEY_PLUS1 0xBF800000u                         /* f32_key(+1.0f) */
#define SELQ_KEY_ZERO 0x80000000u                          /* f32_key(+0.0f) */

// first-level bucket of x in [-1, 1]: round((x + 1) * 1023.5) in 0..2047, read off the mantissa of t + 2^23
__device__ inline unsigned int selq_bucket(float x)
{
    const float t = __builtin_fmaf(x, 1023.5f, 1023.5f);
    const float u = t + 8388608.0f;
    return __builtin_bit_cast(unsigned int, u) & 0x7FFFFFu;
}
// smallest key in [key(-1), key(+1) + 1] whose bucket is >= b
__device__ inline unsigned int selq_lower_key(unsigned int b)
{
    unsigned int lo = SELQ_KEY_MINUS1, hi = SELQ_KEY_PLUS1 + 1u;
    while (lo < hi) {
        const unsigned int mid = lo + ((hi - lo) >> 1);
        if (selq_bucket(key_f32(mid)) >= b) hi = mid;
        else lo = mid + 1u;
    }
    return lo;
}

struct SelQParams {
    const uint8_t *tiles;
    const uint8_t *wb_table;
    long long npix;
    int first;                            // 1: bucket pass (every value counts, track = lane parity)
    unsigned int bias[4], shift[4];       // [stream * 2 + track], later passes
    unsigned long long *hist;             // [2][2][SELQ_BINS], accumulated with atomics (whole-batch variant)
    // per-tile selection (medians of every tile of a batch, all on the device): state and 32-bit histograms per tile
    struct SelQTile *state;
    unsigned int *hist32;                 // [ntiles][2][2][SELQ_BINS]
};
struct SelQTile {
    unsigned int bias[4];                 // [stream * 2 + track]: key the digits are counted from
    unsigned int shift[4];
    unsigned int rank[4];                 // rank still to find at or above bias
    unsigned int done;                    // bit c: combo c has had its shift-0 pass (bias is the key of the order statistic)
    unsigned int pad[3];
};

template <bool WB, bool PER_TILE>
__global__ __launch_bounds__(1024, 8) void k_selq_pass(SelQParams P)
{
    // 64 KiB table + one 2048-word row per stream = exactly 80 KiB: two blocks per CU, 8 waves per SIMD
    __shared__ __attribute__((aligned(16))) char s_tab[WB ? V2_TABLE_BYTES : 16];
    __shared__ unsigned int s_h[2 * SELQ_BINS];
    const int tid = threadIdx.x;
    const unsigned int lane_off4 = (tid & 63u) << 2;
    const long long tile = blockIdx.y;
    const long long npix = P.npix;
    const uint8_t *base = P.tiles + tile * npix * 3;
    // a tile whose four order statistics are settled skips the spare passes (uniform per block)
    if (PER_TILE && !P.first && P.state[tile].done == 0xFu) return;
    if (WB) {
        const uint8_t *t = P.wb_table + tile * 768;
        unsigned int *tab = reinterpret_cast<unsigned int *>(s_tab);
        for (int i = tid; i < 256 * 64; i += 1024) {
            const int v = i >> 6;
            tab[i] = (unsigned)t[v] | ((unsigned)t[256 + v] << 8) | ((unsigned)t[512 + v] << 16);
        }
    }
    for (int i = tid; i < 2 * SELQ_BINS; i += 1024) s_h[i] = 0;
    __syncthreads();

    const unsigned int *bias = PER_TILE ? P.state[tile].bias : P.bias;
    const unsigned int *shft = PER_TILE ? P.state[tile].shift : P.shift;
    const unsigned int ba[2] = {bias[0], bias[1]}, bb[2] = {bias[2], bias[3]};
    const unsigned int sa[2] = {shft[0], shft[1]}, sb[2] = {shft[2], shft[3]};
    const long long nquads = npix >> 2;
    const unsigned int sign_bit = 0x80000000u;
    const unsigned int dummy_idx = SELQ_DIGITS + (tid & 63u);     // a value outside the range adds to its lane's dummy word

    // one sweep over the tile: MODE 0 counts buckets, MODE 1 digits (key - bias) >> shift of both streams; no divergence
    // (a digit is compared with the row's usable length by one v_min)
    auto sweep = [&](auto mode_tag, unsigned int b0, unsigned int s0, unsigned int b1, unsigned int s1) {
        constexpr int MODE = decltype(mode_tag)::value;
        auto push_n = [&](int stream, const float *x, int nval, unsigned int bs, unsigned int sh) {
            for (int j = 0; j < nval; ++j) {
                if (MODE == 0) {
                    atomicAdd(&s_h[stream * SELQ_BINS + selq_bucket(x[j])], 1u);
                    continue;
                }
                const unsigned int bits_j = __builtin_bit_cast(unsigned int, x[j]);
                unsigned int key;
                // order-preserving key: x >= 0 -> bits | 2^31, x < 0 -> ~bits
                asm("v_ashrrev_i32 %0, 31, %1\n\tv_or_b32 %0, %2, %0\n\tv_xor_b32 %0, %0, %1" : "=&v"(key) : "v"(bits_j), "v"(sign_bit));
                const unsigned int d = (key - bs) >> sh;
                atomicAdd(&s_h[stream * SELQ_BINS + (d < dummy_idx ? d : dummy_idx)], 1u);
            }
        };
        for_each_quad_ring<1024>(base, nquads, [&](long long, unsigned int w0, unsigned int w1, unsigned int w2) {
            const unsigned int wr[4] = {w0, w0, w1, w2}, wg[4] = {w0, w1, w1, w2}, wn[4] = {w0, w1, w2, w2};
            constexpr int br[4] = {0, 3, 2, 1}, bg[4] = {1, 0, 3, 2}, bn[4] = {2, 1, 0, 3};
            float fn[4], fr[4], fg[4], qv[4], qg[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                fn[px] = sample<WB>(wn[px], bn[px], 2, lane_off4, s_tab);
                fr[px] = sample<WB>(wr[px], br[px], 0, lane_off4, s_tab);
                fg[px] = sample<WB>(wg[px], bg[px], 1, lane_off4, s_tab);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 N = {fn[2 * h], fn[2 * h + 1]}, R = {fr[2 * h], fr[2 * h + 1]}, G = {fg[2 * h], fg[2 * h + 1]};
                const f32x2 Ne = N + (f32x2){LARS_DEN_EPS, LARS_DEN_EPS};
                const f32x2 v = exact_quot2(N - R, Ne + R), g = exact_quot2(N - G, Ne + G);
                qv[2 * h] = v.x; qv[2 * h + 1] = v.y; qg[2 * h] = g.x; qg[2 * h + 1] = g.y;
            }
            push_n(0, qv, 4, b0, s0);
            push_n(1, qg, 4, b1, s1);
        });
        if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
            const long long i = nquads * 4 + tid;
            unsigned int r = base[i * 3], g = base[i * 3 + 1], n = base[i * 3 + 2];
            if (WB) {
                const unsigned int *tab = reinterpret_cast<const unsigned int *>(s_tab);
                r = tab[r * 64] & 0xFFu; g = (tab[g * 64] >> 8) & 0xFFu; n = (tab[n * 64] >> 16) & 0xFFu;
            }
            const float tv = norm_diff_fast((float)n, (float)r), tg = norm_diff_fast((float)n, (float)g);
            push_n(0, &tv, 1, b0, s0);
            push_n(1, &tg, 1, b1, s1);
        }
    };
    // rows -> the histogram of `track` (the dummy words are not part of it)
    auto flush = [&](int track, int nbins) {
        __syncthreads();
        for (int i = tid; i < 2 * SELQ_BINS; i += 1024) {
            const int stream = i >> 11, bin = i & (SELQ_BINS - 1);
            const unsigned int v = bin < nbins ? s_h[i] : 0u;
            if (v) {
                const long long at = (stream * 2 + track) * SELQ_BINS + bin;
                if (PER_TILE) atomicAdd(&P.hist32[tile * (4 * SELQ_BINS) + at], v);
                else atomicAdd(&P.hist[at], (unsigned long long)v);
            }
        }
    };
    if (P.first) {
        sweep(std::integral_constant<int, 0>{}, 0u, 0u, 0u, 0u);
        flush(0, SELQ_BINS);
    } else {
        // both ranks of a stream usually share (bias, shift): one sweep, counted under track 0.  Otherwise a second sweep
        // recounts for track 1 (rare: the two middle ranks straddle a bin boundary).
        const bool split = ba[0] != ba[1] || sa[0] != sa[1] || bb[0] != bb[1] || sb[0] != sb[1];
        sweep(std::integral_constant<int, 1>{}, ba[0], sa[0], bb[0], sb[0]);
        flush(0, SELQ_DIGITS);
        if (split) {
            __syncthreads();
            for (int i = tid; i < 2 * SELQ_BINS; i += 1024) s_h[i] = 0;
            __syncthreads();
            sweep(std::integral_constant<int, 1>{}, ba[1], sa[1], bb[1], sb[1]);
            flush(1, SELQ_DIGITS);
        }
    }
}

__global__ void k_selq_init(SelQTile *state, long long ntiles, long long npix, unsigned int streams)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntiles) {
        SelQTile t;
        for (int c = 0; c < 4; ++c) { t.bias[c] = 0u; t.shift[c] = 0u; t.rank[c] = (unsigned int)((c & 1) ? npix / 2 : (npix - 1) / 2); }
        t.done = ((streams & 1u) ? 0u : 0x3u) | ((streams & 2u) ? 0u : 0xCu);      // a stream nobody asked for is settled
        t.pad[0] = t.pad[1] = t.pad[2] = 0u;
        state[i] = t;
    }
}

// one block per tile, one wave per (stream, track): find the bin whose cumulative count covers the rank and
// narrow the key range
__global__ __launch_bounds__(256) void k_selq_pick(SelQTile *state, unsigned int *hist32, int first)
{
    const long long tile = blockIdx.x;
    unsigned int *h = hist32 + tile * (4 * SELQ_BINS);
    const int combo = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned int done = state[tile].done;
    if (!first && done == 0xFu) return;                     // the pass did not run for this tile (uniform per block)
    const bool active = !((done >> combo) & 1u);            // settled combos (or streams nobody asked for) only help zeroing
    // a later pass counted a (bias, shift) shared by both tracks once, under track 0 (decided before anything changes)
    const bool shared = !first && state[tile].bias[combo & 2] == state[tile].bias[combo | 1] &&
                        state[tile].shift[combo & 2] == state[tile].shift[combo | 1];
    const unsigned int rank = state[tile].rank[combo], bias = state[tile].bias[combo], shift = state[tile].shift[combo];
    __syncthreads();
    if (active) {
        const unsigned int *mine = h + (shared ? (combo & 2) : combo) * SELQ_BINS;
        const unsigned int *twin = h + (combo ^ 1) * SELQ_BINS;    // bucket pass: the two tracks are two copies
        unsigned int c[32], local = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int bin_j = lane * 32 + j;
            c[j] = first ? mine[bin_j] + twin[bin_j] : (bin_j < SELQ_DIGITS ? mine[bin_j] : 0u);
            local += c[j];
        }
        unsigned int incl = local;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        unsigned int cum = incl - local;
        if (rank >= cum && rank < incl) {                   // exactly one lane (the bins up to the range's end hold >= rank + 1 values)
            int d = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (rank >= cum + c[j]) { cum += c[j]; d = j + 1; }
                else break;
            }
            const unsigned int bin = (unsigned int)(lane * 32 + d);
            unsigned int nbias, nshift;
            if (first) {
                unsigned int lo = selq_lower_key(bin);
                unsigned int hi = bin >= SELQ_BINS - 1 ? SELQ_KEY_PLUS1 + 1u : selq_lower_key(bin + 1u);
                if (lo <= SELQ_KEY_ZERO && SELQ_KEY_ZERO < hi) { lo = SELQ_KEY_ZERO; hi = SELQ_KEY_ZERO + 1u; }   // only +0.0 lives there
                const unsigned int span = hi - lo - 1u;    // largest offset inside the range
                nshift = 0u;
                while ((span >> nshift) >= SELQ_DIGITS) ++nshift;      // the first digit must fit the row
                nbias = lo;
            } else {
                nbias = bias + (bin << shift);
                nshift = shift > SELQ_DIGIT_BITS ? shift - SELQ_DIGIT_BITS : 0u;
            }
            state[tile].bias[combo] = nbias;
            state[tile].shift[combo] = nshift;
            state[tile].rank[combo] = rank - cum;
            if (!first && shift == 0u) atomicOr(&state[tile].done, 1u << combo);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * SELQ_BINS; i += 256) h[i] = 0u;
}

// a selection that is not settled after the last pass comes back as NaN instead of a wrong value
__global__ void k_selq_finish_checked(const SelQTile *state, long long ntiles, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntiles * 4) out[i] = ((state[i >> 2].done >> (i & 3)) & 1u) ? key_f32(state[i >> 2].bias[i & 3]) : __builtin_nanf("");
}

}  // namespace lars

using namespace lars;

namespace lars {

int selq_pass_launch(const uint8_t *tiles, const uint8_t *wb_table, long long ntiles, long long npix, int first,
                     const unsigned int bias[4], const unsigned int shift[4], unsigned long long *hist, hipStream_t s)
{
    SelQParams P;
    memset(&P, 0, sizeof P);
    P.tiles = tiles; P.wb_table = wb_table; P.npix = npix; P.first = first;
    for (int c = 0; c < 4; ++c) { P.bias[c] = bias[c]; P.shift[c] = shift[c]; }
    P.hist = hist;
    long long bpt = (2048 + ntiles - 1) / ntiles;                  // ~2048 workgroups per launch
    const long long cap = (npix / 4 + 1024 * 8 - 1) / (1024 * 8);  // at least ~8 steps per block (64 KiB table each)
    if (bpt > cap) bpt = cap;
    if (bpt < 1) bpt = 1;
    dim3 grid((unsigned)bpt, (unsigned)ntiles);
    if (wb_table) hipLaunchKernelGGL((k_selq_pass<true, false>), grid, dim3(1024), 0, s, P);
    else hipLaunchKernelGGL((k_selq_pass<false, false>), grid, dim3(1024), 0, s, P);
    return launch_check("k_selq_pass");
}

size_t selq_tile_scratch_bytes(long long ntiles)
{
    return (size_t)ntiles * (sizeof(SelQTile) + 4 * SELQ_BINS * sizeof(unsigned int)) + 512;
}

// medians of every tile: bucket pass + two digit passes, picks on the device, no host round trip
// selq_tile_prepare: state + zeroed histograms (before a fused statistics + bucket pass); selq_tile_hist32: where that
// pass adds its counts; selq_tile_medians_launch(..., first_pass_done): the remaining passes.
static void selq_scratch_layout(void *scratch, long long ntiles, SelQTile **state, unsigned int **hist32)
{
    *state = static_cast<SelQTile *>(scratch);
    *hist32 = reinterpret_cast<unsigned int *>(static_cast<char *>(scratch) + (((size_t)ntiles * sizeof(SelQTile) + 255) & ~(size_t)255));
}
unsigned int *selq_tile_hist32(void *scratch, long long ntiles)
{
    SelQTile *state; unsigned int *hist32;
    selq_scratch_layout(scratch, ntiles, &state, &hist32);
    return hist32;
}
int selq_tile_prepare(void *scratch, long long ntiles, long long npix, hipStream_t s, unsigned int streams)
{
    SelQTile *state; unsigned int *hist32;
    selq_scratch_layout(scratch, ntiles, &state, &hist32);
    hipLaunchKernelGGL(k_selq_init, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, state, ntiles, npix, streams);
    if (hipMemsetAsync(hist32, 0, (size_t)ntiles * 4 * SELQ_BINS * sizeof(unsigned int), s) != hipSuccess)
        return fail(LARS_ERR_HIP, "hipMemsetAsync failed (tile median scratch)");
    return launch_check("selq_tile_prepare");
}

int selq_tile_medians_launch(const uint8_t *tiles, const uint8_t *wb_table, long long ntiles, long long npix, float *out_pairs,
                             void *scratch, hipStream_t s, bool first_pass_done)
{
    SelQTile *state; unsigned int *hist32;
    selq_scratch_layout(scratch, ntiles, &state, &hist32);
    if (!first_pass_done) LARS_TRY(selq_tile_prepare(scratch, ntiles, npix, s, 3u));
    long long bpt = (2048 + ntiles - 1) / ntiles;
    const long long cap = (npix / 4 + 1024 * 8 - 1) / (1024 * 8);
    if (bpt > cap) bpt = cap;
    if (bpt < 1) bpt = 1;
    dim3 grid((unsigned)bpt, (unsigned)ntiles);
    // bucket pass + up to five digit passes (a first digit of < 1984 values, then 10 bits per pass: a 2^32-wide range
    // needs shifts 22, 12, 2, 0); uint8 tiles need two, exceptionally three -- a settled tile's blocks return at once
    for (int p = 0; p < 6; ++p) {
        SelQParams P;
        memset(&P, 0, sizeof P);
        P.tiles = tiles; P.wb_table = wb_table; P.npix = npix; P.first = p == 0;
        P.state = state; P.hist32 = hist32;
        if (p == 0 && first_pass_done) { /* counted by the statistics kernel */ }
        else if (wb_table) hipLaunchKernelGGL((k_selq_pass<true, true>), grid, dim3(1024), 0, s, P);
        else hipLaunchKernelGGL((k_selq_pass<false, true>), grid, dim3(1024), 0, s, P);
        hipLaunchKernelGGL(k_selq_pick, dim3((unsigned)ntiles), dim3(256), 0, s, state, hist32, p == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL(k_selq_finish_checked, dim3((unsigned)((ntiles * 4 + 255) / 256)), dim3(256), 0, s, state, ntiles, out_pairs);
    return launch_check("selq_tile_medians");
}

}  // namespace lars

// ===========================================================================
// entry points
// ===========================================================================
extern "C" int lars_d_quotient_digit_hist(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                          const uint8_t *wb_table, int first, const uint32_t bias[4], const uint32_t shift[4],
                                          uint64_t *hist, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || !hist || !bias || !shift || ntiles <= 0 || npix <= 0)
        return fail(LARS_ERR_INVALID, "lars_d_quotient_digit_hist: bad arguments");
    if (dtype != LARS_U8 || channels != 3 || (reinterpret_cast<uintptr_t>(tiles) & 3) || (ntiles > 1 && (npix & 3)))
        return fail(LARS_ERR_INVALID, "lars_d_quotient_digit_hist: uint8 [ntiles][npix][3] tiles on 4-byte boundaries are required");
    for (int k = 0; k < 4; ++k)
        if (shift[k] > 31u) return fail(LARS_ERR_INVALID, "lars_d_quotient_digit_hist: shift must be below 32");
    if (ntiles > 65535) return fail(LARS_ERR_INVALID, "lars_d_quotient_digit_hist: at most 65535 tiles per launch");
    return selq_pass_launch(static_cast<const uint8_t *>(tiles), wb_table, ntiles, npix, first ? 1 : 0, bias, shift,
                            reinterpret_cast<unsigned long long *>(hist), pick_stream(c, stream));
}

extern "C" size_t lars_quotient_median_scratch_bytes(int64_t ntiles) { return selq_tile_scratch_bytes(ntiles > 0 ? ntiles : 1); }

extern "C" int lars_d_quotient_median_pairs(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                            const uint8_t *wb_table, float *out_pairs, void *scratch, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || !out_pairs || !scratch || ntiles <= 0 || npix <= 0)
        return fail(LARS_ERR_INVALID, "lars_d_quotient_median_pairs: bad arguments");
    if (dtype != LARS_U8 || channels != 3 || (reinterpret_cast<uintptr_t>(tiles) & 3) || (ntiles > 1 && (npix & 3)))
        return fail(LARS_ERR_INVALID, "lars_d_quotient_median_pairs: uint8 [ntiles][npix][3] tiles on 4-byte boundaries are required");
    if (ntiles > 65535 || npix >= (1ll << 32)) return fail(LARS_ERR_INVALID, "lars_d_quotient_median_pairs: at most 65535 tiles of < 2^32 pixels");
    return selq_tile_medians_launch(static_cast<const uint8_t *>(tiles), wb_table, ntiles, npix, out_pairs, scratch,
                                    pick_stream(c, stream), false);
}

// Statistics AND the exact median of every tile in four passes over the tiles instead of five: the statistics kernel
// also counts the select's bucket pass.  Same constraints as lars_d_quotient_median_pairs; no output planes.
extern "C" int lars_d_stats_medians(const lars_fused_args *a, float *out_pairs, void *scratch)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!a || !a->tiles || !a->stats || !out_pairs || !scratch || a->ntiles <= 0 || a->npix <= 0)
        return fail(LARS_ERR_INVALID, "lars_d_stats_medians: bad arguments");
    if (a->dtype != LARS_U8 || a->channels != 3 || (reinterpret_cast<uintptr_t>(a->tiles) & 3) || (a->ntiles > 1 && (a->npix & 3)))
        return fail(LARS_ERR_INVALID, "lars_d_stats_medians: uint8 [ntiles][npix][3] tiles on 4-byte boundaries are required");
    const unsigned mask = a->index_mask & LARS_MASK_ALL;
    if (mask != 1u && mask != 2u && mask != 4u && mask != 7u)
        return fail(LARS_ERR_INVALID, "lars_d_stats_medians: index_mask must be one index or all three");
    if (a->out_wb || a->out_index[0] || a->out_index[1] || a->out_index[2] || a->out_rgba[0] || a->out_rgba[1] || a->out_rgba[2])
        return fail(LARS_ERR_INVALID, "lars_d_stats_medians: no output planes (use lars_d_fused + lars_d_median_pair_batch_f32)");
    if (a->ntiles > 65535 || a->npix >= (1ll << 32) || (long long)a->npix * 6 >= (1ll << 30))
        return fail(LARS_ERR_INVALID, "lars_d_stats_medians: at most 65535 tiles of < 2^30 / 6 pixels");
    hipStream_t s = pick_stream(c, a->stream);
    const int stats_mode = (a->flags & LARS_F_SUMSQ) ? 3 : (a->flags & LARS_F_HIST) ? 2 : 1;
    const uint8_t *tiles = static_cast<const uint8_t *>(a->tiles);

    FusedParams P;
    memset(&P, 0, sizeof P);
    P.tiles = a->tiles; P.npix = a->npix; P.channels = 3; P.wb_table = a->wb_table; P.stats = a->stats; P.mask = mask;
    P.flags = a->flags & 7u;
    LARS_TRY(selq_tile_prepare(scratch, a->ntiles, a->npix, s, ((mask & 1u) ? 1u : 0u) | ((mask & 6u) ? 2u : 0u)));
    P.sel_hist = selq_tile_hist32(scratch, a->ntiles);
    const long long nrec = a->ntiles * 3;
    stats_init_launch(a->stats, nrec, mask, s);
    dim3 grid(blocks_per_tile(a->npix / 4 + 1, a->ntiles, fused_v2_threads(false), 8192), (unsigned)a->ntiles);
    fused_v2_sel_launch(mask, a->wb_table != nullptr, stats_mode, grid, s, P);
    stats_finalize_launch(a->stats, nrec, mask, (long long)a->npix, s);
    LARS_TRY(launch_check("lars_d_stats_medians"));
    return selq_tile_medians_launch(tiles, a->wb_table, a->ntiles, a->npix, out_pairs, scratch, s, true);
}
