// Exact medians without index planes: radix select on values recomputed from the tiles.
// The first (bucket) pass also exists inside the statistics kernel (fused_v2.hip, SEL).
//
// WHAT STILL DEPENDS ON THIS FILE (round 4).  Per-tile medians of uint8 RGNir / RGBA batches come out of the one-read route's
// finish kernel (joint.hip) since round 3.  These passes remain for: TileBatch.global_medians (ONE median over all tiles of all
// ranks: histograms are summed over the communicator between the passes, which a per-tile finish cannot do), tile_medians /
// lars_d_quotient_median_pairs on tables the caller supplied or with planes that carry a white-balanced image, lars_d_stats_medians
// on the per-pixel route, and the other side of the median checks in tests/test_gpu_joint.py and bench.py's self-check.
#include <string.h>

#include <type_traits>

#include "v2_device.h"

namespace lars {

// ---------------------------------------------------------------------------
// Exact medians without materialising the index planes -- of every tile of a batch, or of the whole batch over
// all ranks.  A two-level select on values that are RECOMPUTED from the tiles (3 bytes per pixel and
// pass).  Two streams (NDVI, GNDVI; NDWI = -GNDVI shares GNDVI's order statistics) x two tracks (the ranks
// (N-1)/2 and N/2, which may part ways).
//   pass 1   2048 linear buckets of [-1, 1]: bucket = floor(t) - 2048, t = fma(x, 1023.5, 3071.5) (v2_device.h).  Monotone in
//            x, so a valid first radix level, and unlike a float key's top bits it spreads an index plane over hundreds
//            of LDS words.
//   pick     bucket holding the rank, rank inside it
//   pass 2   slot = (fraction bits of t) >> 2 of the values whose t falls into the picked bucket -- one integer
//            subtract tells both ("t bits - bits of the bucket's first t" is below 4096 exactly for the bucket's members);
//            anything else lands on a per-lane dummy word (one v_min, no branch).
//   pick     slot holding the rank.  A slot holds ONE distinct value: quotients of bytes are fractions n/d with
//            d <= 510, any two of which differ by >= 1/(510 * 509) = 16.1 units of the fraction, 15 after both
//            roundings, and a slot is 4 units wide.  The value itself is found by trying every denominator: n =
//            rint(centre * d), q = n / d with the kernels' own quotient, accepted when its (bucket, slot) is the
//            picked one.
// Always two passes; both see the same bits of t, so they cannot disagree about membership.
// LDS: the 64 KiB white-balance table + one 2048-word row per stream = exactly 80 KiB, two blocks per CU.
// ---------------------------------------------------------------------------

struct SelQParams {
    const uint8_t *tiles;
    const uint8_t *wb_table;
    long long npix;
    int first;                            // 1: bucket pass (every value counts, under track 0); 0: slot pass; 2: window pass;
                                          // 3 (whole batch only): bucket pass over every 16th grid stride (the prediction's sample)
    unsigned int bucket[4];               // [stream * 2 + track], second pass
    unsigned long long *hist;             // [2][2][SELQ_BINS], accumulated with atomics (whole-batch variant)
    // per-tile selection (medians of every tile of a batch, all on the device): state and 32-bit histograms per tile
    struct SelQTile *state;
    unsigned int *hist32;                 // [ntiles][2][2][SELQ_BINS]
    const unsigned int *tile_list;        // per-tile mode: blockIdx.y -> tile (the tiles the one-pass route did not serve), or null
    // window pass (first == 2): what the statistics kernel counts with SEL == 2 (fused_v2.hip), without the statistics
    const unsigned int *win;              // [ntiles][2]: first slot of each stream's predicted window
    unsigned int *win_hist;               // [ntiles][2][SELQ_WIN_SLOTS]
    unsigned int *below;                  // [ntiles][2]
};
struct SelQTile {
    unsigned int bucket[4];               // [stream * 2 + track]: bucket picked after pass 1
    unsigned int rank[4];                 // rank still to find inside the bucket
    unsigned int streams;                 // bit s: stream s was asked for
    unsigned int pad[3];
};

// STREAMS: bit 0 NDVI, bit 1 GNDVI -- a stream nobody asked for is neither sampled nor divided nor counted
template <bool WB, bool PER_TILE, unsigned STREAMS>
__global__ __launch_bounds__(1024, 8) void k_selq_pass(SelQParams P)
{
    // 64 KiB table + one 2048-word row per stream = exactly 80 KiB: two blocks per CU, 8 waves per SIMD
    __shared__ __attribute__((aligned(16))) char s_tab[WB ? V2_TABLE_BYTES : 16];
    __shared__ unsigned int s_h[2 * SELQ_BINS];
    const int tid = threadIdx.x;
    const unsigned int lane_off4 = (tid & 63u) << 2;
    const long long tile = (PER_TILE && P.tile_list) ? (long long)P.tile_list[blockIdx.y] : (long long)blockIdx.y;
    const long long npix = P.npix;
    const uint8_t *base = P.tiles + tile * npix * 3;
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

    const unsigned int *bk = PER_TILE ? P.state[tile].bucket : P.bucket;
    const unsigned int ba[2] = {bk[0], bk[1]}, bb[2] = {bk[2], bk[3]};
    const long long nquads = npix >> 2;
    const unsigned int dummy_idx = SELQ_SLOTS + (tid & 63u);      // a value outside the bucket adds to its lane's dummy word

    // one sweep over the tile: MODE 0 counts buckets, MODE 1 the slots inside bucket b0 (NDVI) / b1 (GNDVI); no divergence
    auto sweep = [&](auto mode_tag, unsigned int b0, unsigned int b1, int every = 1) {
        constexpr int MODE = decltype(mode_tag)::value;
        const unsigned int t0[2] = {SELQ_T_BITS | (b0 << 12), SELQ_T_BITS | (b1 << 12)};     // bits of the bucket's first t
        // MODE 2: b0 / b1 carry the windows' first slots; the row of a stream is 64 "below" words | the slots | 64 "above" words
        typedef __attribute__((address_space(3))) unsigned int lds_u32;
        const unsigned int row_rel = (unsigned int)(unsigned long long)(lds_u32 *)s_h - (SELQ_WIN_MAGIC_BITS << 2);
        const float bias[2] = {selq_window_bias((int)b0), selq_window_bias((int)b1)};
        const int win_lo = (int)(SELQ_WIN_MAGIC_BITS + (tid & 63u)), win_hi = (int)(SELQ_WIN_MAGIC_BITS + 64 + SELQ_WIN_SLOTS + (tid & 63u));
        auto push_n = [&](int stream, const float *x, int nval) {
            for (int j = 0; j < nval; ++j) {
                if (MODE == 2) {
                    selq_window_add(__builtin_fmaf(x[j], SELQ_WIN_SCALE, bias[stream]), row_rel + stream * SELQ_BINS * 4, win_lo, win_hi);
                    continue;
                }
                const float t = selq_t(x[j]);
                if (MODE == 0) {
                    atomicAdd(&s_h[stream * SELQ_BINS + selq_bucket_of(t)], 1u);
                    continue;
                }
                const unsigned int d = (__builtin_bit_cast(unsigned int, t) - t0[stream]) >> 2;    // < 1024 inside the bucket
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
                if (STREAMS & 1u) fr[px] = sample<WB>(wr[px], br[px], 0, lane_off4, s_tab);
                if (STREAMS & 2u) fg[px] = sample<WB>(wg[px], bg[px], 1, lane_off4, s_tab);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 N = {fn[2 * h], fn[2 * h + 1]};
                const f32x2 Ne = N + (f32x2){LARS_DEN_EPS, LARS_DEN_EPS};
                if (STREAMS & 1u) {
                    const f32x2 R = {fr[2 * h], fr[2 * h + 1]};
                    const f32x2 v = exact_quot2(N - R, Ne + R);
                    qv[2 * h] = v.x; qv[2 * h + 1] = v.y;
                }
                if (STREAMS & 2u) {
                    const f32x2 G = {fg[2 * h], fg[2 * h + 1]};
                    const f32x2 g = exact_quot2(N - G, Ne + G);
                    qg[2 * h] = g.x; qg[2 * h + 1] = g.y;
                }
            }
            if (STREAMS & 1u) push_n(0, qv, 4);
            if (STREAMS & 2u) push_n(1, qg, 4);
        }, every);
        if (every == 1 && blockIdx.x == 0 && tid < (int)(npix & 3)) {
            const long long i = nquads * 4 + tid;
            unsigned int r = base[i * 3], g = base[i * 3 + 1], n = base[i * 3 + 2];
            if (WB) {
                const unsigned int *tab = reinterpret_cast<const unsigned int *>(s_tab);
                r = tab[r * 64] & 0xFFu; g = (tab[g * 64] >> 8) & 0xFFu; n = (tab[n * 64] >> 16) & 0xFFu;
            }
            const float tv = norm_diff_fast((float)n, (float)r), tg = norm_diff_fast((float)n, (float)g);
            if (STREAMS & 1u) push_n(0, &tv, 1);
            if (STREAMS & 2u) push_n(1, &tg, 1);
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
    if (P.first == 2) {
        static_assert(64 + SELQ_WIN_SLOTS + 64 == SELQ_BINS, "a window row is a bucket row");
        // per tile: the predicted windows of this tile; whole batch: one window per stream in bucket[0] / bucket[2]
        sweep(std::integral_constant<int, 2>{}, PER_TILE ? P.win[tile * 2] : P.bucket[0], PER_TILE ? P.win[tile * 2 + 1] : P.bucket[2]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the window atomics are inline asm
        __syncthreads();
        for (int i = tid; i < 2 * SELQ_BINS; i += 1024) {
            const unsigned int v = s_h[i];
            if (!v) continue;
            const int stream = i >> 11, w = i & (SELQ_BINS - 1);
            if (!PER_TILE) atomicAdd(&P.hist[(stream * 2) * SELQ_BINS + w], (unsigned long long)v);     // the whole row, under track 0
            else if (w < 64) atomicAdd(&P.below[tile * 2 + stream], v);
            else if (w < 64 + SELQ_WIN_SLOTS) atomicAdd(&P.win_hist[(tile * 2 + stream) * SELQ_WIN_SLOTS + (w - 64)], v);
        }
    } else if (!PER_TILE && P.first == 3) {
        sweep(std::integral_constant<int, 0>{}, 0u, 0u, 16);
        flush(0, SELQ_BINS);
    } else if (P.first) {
        sweep(std::integral_constant<int, 0>{}, 0u, 0u);
        flush(0, SELQ_BINS);
    } else {
        // both ranks of a stream usually share the bucket: one sweep, counted under track 0.  Otherwise a second sweep
        // recounts for track 1 (rare: the two middle ranks straddle a bucket boundary).
        const bool split = ba[0] != ba[1] || bb[0] != bb[1];
        sweep(std::integral_constant<int, 1>{}, ba[0], bb[0]);
        flush(0, SELQ_SLOTS);
        if (split) {
            __syncthreads();
            for (int i = tid; i < 2 * SELQ_BINS; i += 1024) s_h[i] = 0;
            __syncthreads();
            sweep(std::integral_constant<int, 1>{}, ba[1], bb[1]);
            flush(1, SELQ_SLOTS);
        }
    }
}

__global__ void k_selq_init(SelQTile *state, long long ntiles, long long npix, unsigned int streams)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntiles) {
        SelQTile t;
        for (int c = 0; c < 4; ++c) { t.bucket[c] = 0u; t.rank[c] = (unsigned int)((c & 1) ? npix / 2 : (npix - 1) / 2); }
        t.streams = streams;
        t.pad[0] = t.pad[1] = t.pad[2] = 0u;
        state[i] = t;
    }
}

// the quotient of bytes whose position is (bucket, slot): try every denominator (see the header of this file).
// One wave; every lane returns the value (NaN if there is none, which a consistent pair of passes cannot produce).
__device__ inline float selq_value_of(unsigned int bucket, unsigned int slot, int lane)
{
    const double centre_t = 2048.0 + (double)bucket + ((double)slot * 4.0 + 2.0) / 4096.0;
    const double centre = (centre_t - 3071.5) / 1023.5;
    const unsigned int want = SELQ_T_BITS | (bucket << 12);
    float found = __builtin_nanf("");
    for (int den = 1 + lane; den <= 510; den += 64) {
        const float n = (float)__builtin_rint(centre * (double)den);
        if (__builtin_fabsf(n) > (float)den) continue;
        const float q = exact_quot(n, (float)den);
        const unsigned int d = __builtin_bit_cast(unsigned int, selq_t(q)) - want;
        if (d < 4096u && (d >> 2) == slot) found = q + 0.0f;           // -0/den -> +0.0, like the kernels' (a - b) / (a + b)
    }
    // any lane that found one found the same value
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(found, off);
        if (found != found) found = o;
    }
    return found;
}

// one block per tile, one wave per (stream, track): find the bin whose cumulative count covers the rank.
// first: bucket -> state; second: slot -> the value, written to out[tile][stream][track]
__global__ __launch_bounds__(256) void k_selq_pick(SelQTile *state, unsigned int *hist32, int first, float *out, const unsigned int *tile_list)
{
    const long long tile = tile_list ? (long long)tile_list[blockIdx.x] : (long long)blockIdx.x;
    unsigned int *h = hist32 + tile * (4 * SELQ_BINS);
    const int combo = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool active = (state[tile].streams >> (combo >> 1)) & 1u;
    // the second pass counted a bucket shared by both tracks once, under track 0
    const bool shared = !first && state[tile].bucket[combo & 2] == state[tile].bucket[combo | 1];
    const unsigned int rank = state[tile].rank[combo], bucket = state[tile].bucket[combo];
    __syncthreads();
    if (!active && !first && lane == 0) out[tile * 4 + combo] = __builtin_nanf("");
    if (active) {
        // bucket pass: everything is counted under track 0
        const unsigned int *mine = h + ((first || shared) ? (combo & 2) : combo) * SELQ_BINS;
        const int nbins = first ? SELQ_BINS : SELQ_SLOTS;
        unsigned int c[32], local = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int bin_j = lane * 32 + j;
            c[j] = bin_j < nbins ? mine[bin_j] : 0u;
            local += c[j];
        }
        unsigned int incl = local;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        unsigned int cum = incl - local;
        const bool holder = rank >= cum && rank < incl;     // exactly one lane (the bins hold >= rank + 1 values)
        int d = 0;
        if (holder) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (rank >= cum + c[j]) { cum += c[j]; d = j + 1; }
                else break;
            }
        }
        const unsigned long long who = __ballot(holder);
        if (first) {
            if (holder) {
                state[tile].bucket[combo] = (unsigned int)(lane * 32 + d);
                state[tile].rank[combo] = rank - cum;
            }
        } else {
            float v = __builtin_nanf("");
            if (who) {
                const int src = __ffsll((long long)who) - 1;
                const unsigned int slot = (unsigned int)__shfl(lane * 32 + d, src);
                v = selq_value_of(bucket, slot, lane);
            }
            if (lane == 0) out[tile * 4 + combo] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * SELQ_BINS; i += 256) h[i] = 0u;
}

// ---------------------------------------------------------------------------------------------------------------------
// One-pass medians: predict where each tile's median will fall, and let the statistics kernel count -- instead of the 2048
// buckets -- the values below a window of 3.75 buckets around the prediction and the slots inside it (fused_v2.hip,
// SEL == 2).  If a rank falls inside the window the order statistic is exact from those counts; tiles where it does not
// (known exactly: below <= rank < below + window mass) take the two classic passes.
// ---------------------------------------------------------------------------------------------------------------------
// The prediction: where the distribution of a subsample (every SUB-th 1024-pixel step, at most 1024 steps: a sixteenth of a
// 4096 x 4096 tile), counted in quarter buckets, passes one half.  In ranks the standard error of a sample median is sqrt(n) / 2 (512 of n = 2^20); how
// many buckets that is depends on the tile (half a bucket for a density of 1 per unit of the index, two buckets where the
// median sits among the sparse quotients around 0).  One block per tile; win[tile][stream] = the window's first slot (an int,
// v2_device.h).
template <bool WB>
__global__ __launch_bounds__(1024) void k_selq_predict(const uint8_t *__restrict__ tiles, const uint8_t *__restrict__ wb_table, long long npix,
                                                       unsigned int streams, unsigned int *__restrict__ win)
{
    // the sample is counted in quarter buckets (PRED_BINS per stream): the window is 15 of them, and where it can start
    // decides how many ranks of the sample lie between its ends and the sample's middle
    constexpr int PRED_SUB = 4, PRED_BINS = SELQ_BINS * PRED_SUB, PRED_WIN = SELQ_WIN_SLOTS * PRED_SUB / SELQ_WIN_PER_BUCKET;
    static_assert(PRED_WIN * SELQ_WIN_PER_BUCKET == SELQ_WIN_SLOTS * PRED_SUB, "the window is a whole number of sample bins");
    __shared__ __attribute__((aligned(16))) char s_tab[WB ? V2_TABLE_BYTES : 16];
    __shared__ unsigned int s_h[2 * PRED_BINS];                    // 64 KiB: one block per CU, as the grid has it anyway
    const int tid = threadIdx.x;
    const unsigned int lane = tid & 63u, lane_off4 = lane << 2;
    const long long tile = blockIdx.x;
    const uint8_t *base = tiles + tile * npix * 3;
    if (WB) {
        const uint8_t *t = wb_table + tile * 768;
        unsigned int *tab = reinterpret_cast<unsigned int *>(s_tab);
        for (int i = tid; i < 256 * 64; i += 1024) {
            const int v = i >> 6;
            tab[i] = (unsigned)t[v] | ((unsigned)t[256 + v] << 8) | ((unsigned)t[512 + v] << 16);
        }
    }
    for (int i = tid; i < 2 * PRED_BINS; i += 1024) s_h[i] = 0;
    __syncthreads();
    auto pred_bin = [](float t) -> unsigned int { return (__builtin_bit_cast(unsigned int, t) >> 10) & (unsigned)(PRED_BINS - 1); };
    const long long nquads = npix >> 2;
    const long long nsteps = nquads >> 8;                          // complete steps only
    const long long sub = nsteps > 1024 ? nsteps / 1024 : 1;
    // the next step's twelve loads are issued before the current step is counted: the kernel is one block of 16 waves per
    // tile and CU, so nothing else hides the latency of a wave's loads
    auto load_step = [&](long long k, unsigned int (&w)[4][3]) {
        const long long q0 = k * sub * 256 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int *p = reinterpret_cast<const unsigned int *>(base + (q0 + 64 * j) * 12);
            w[j][0] = p[0]; w[j][1] = p[1]; w[j][2] = p[2];
        }
    };
    auto count_step = [&](const unsigned int (&w)[4][3]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int w0 = w[j][0], w1 = w[j][1], w2 = w[j][2];
            const unsigned int wr[4] = {w0, w0, w1, w2}, wg[4] = {w0, w1, w1, w2}, wn[4] = {w0, w1, w2, w2};
            constexpr int br[4] = {0, 3, 2, 1}, bg[4] = {1, 0, 3, 2}, bn[4] = {2, 1, 0, 3};
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float fn = sample<WB>(wn[px], bn[px], 2, lane_off4, s_tab);
                if (streams & 1u) {
                    const float fr = sample<WB>(wr[px], br[px], 0, lane_off4, s_tab);
                    atomicAdd(&s_h[pred_bin(selq_t(norm_diff_fast(fn, fr)))], 1u);
                }
                if (streams & 2u) {
                    const float fg = sample<WB>(wg[px], bg[px], 1, lane_off4, s_tab);
                    atomicAdd(&s_h[PRED_BINS + pred_bin(selq_t(norm_diff_fast(fn, fg)))], 1u);
                }
            }
        }
    };
    const long long k_end = nsteps / sub < 1024 ? (nsteps + sub - 1) / sub : 1024;      // sampled steps: k * sub < nsteps, k < 1024
    long long k = tid >> 6;
    if (k < k_end) {
        unsigned int cur[4][3], nxt[4][3];
        load_step(k, cur);
        for (; k + 16 < k_end; k += 16) {
            load_step(k + 16, nxt);
            count_step(cur);
#pragma unroll
            for (int j = 0; j < 4; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; cur[j][2] = nxt[j][2]; }
        }
        count_step(cur);
    }
    __syncthreads();
    // waves 0 and 1: the sample's cumulative counts of stream 0 / 1, written over the counts (lane l owns PRED_BINS / 64
    // consecutive bins), and the bin m that holds the sample's middle rank
    __shared__ unsigned int s_m[2][2];
    const int stream = tid >> 6;
    if (stream < 2) {
        constexpr int PER = PRED_BINS / 64;
        unsigned int *mine = s_h + stream * PRED_BINS + lane * PER;
        unsigned int local = 0;
        for (int j = 0; j < PER; ++j) local += mine[j];
        unsigned int incl = local;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off);
            if ((int)lane >= off) incl += o;
        }
        const unsigned int total = __shfl(incl, 63);
        const unsigned int mid = total / 2;
        unsigned int cum = incl - local;
        for (int j = 0; j < PER; ++j) {
            const unsigned int c = mine[j];
            if (mid >= cum && mid < cum + c) s_m[stream][0] = (unsigned int)((int)lane * PER + j);
            cum += c;
            mine[j] = cum;
        }
        if (lane == 0) s_m[stream][1] = total;
    }
    __syncthreads();
    // Index values of bytes are atoms, not a density (around 0 the distinct quotients lie two buckets apart and one of them
    // can hold 0.6 % of a tile: 13 standard errors of the sample's median, which is sqrt(n) / 2 ranks), so the bucket of the
    // sample's median may be a neighbour of the tile's.  Of the PRED_WIN windows (3.75 buckets, starting on any quarter
    // bucket) that contain m, take the one that keeps the sample's middle rank farthest from both of its ends, in ranks.
    if (tid < 2) {
        const unsigned int *C = s_h + tid * PRED_BINS;                 // inclusive cumulative counts
        const unsigned int total = s_m[tid][1];
        if (!total) win[tile * 2 + tid] = (unsigned int)SELQ_WIN_BOTTOM;
        else {
            const int m = (int)s_m[tid][0];
            const long long mid = total / 2;
            int best = m;
            long long best_margin = -1;
            for (int b = m - PRED_WIN + 1; b <= m; ++b) {
                const int lo = b < 0 ? 0 : (b > PRED_BINS - PRED_WIN ? PRED_BINS - PRED_WIN : b);
                const long long below = lo > 0 ? (long long)C[lo - 1] : 0ll;
                const long long margin_lo = mid - below, margin_hi = (long long)C[lo + PRED_WIN - 1] - 1 - mid;
                const long long margin = margin_lo < margin_hi ? margin_lo : margin_hi;
                if (margin > best_margin) { best_margin = margin; best = lo; }
            }
            // sample bin b starts at t = 2048 + b / PRED_SUB, i.e. at slot sigma = (t - 3071.5) * 512
            int ws = best * (SELQ_WIN_PER_BUCKET / PRED_SUB) + SELQ_WIN_BOTTOM;
            if (ws < SELQ_WIN_BOTTOM) ws = SELQ_WIN_BOTTOM;
            if (ws > -SELQ_WIN_BOTTOM + 1 - SELQ_WIN_SLOTS) ws = -SELQ_WIN_BOTTOM + 1 - SELQ_WIN_SLOTS;
            win[tile * 2 + tid] = (unsigned int)ws;
        }
    }
}

__global__ void k_selq_fill(unsigned int *p, long long n, unsigned int v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// the quotient of bytes in slot `slot` of the window that starts at slot ws (see selq_value_of): every denominator is tried,
// with the kernel's own fma (same bias, so the same rounding) deciding which slot a candidate falls into
__device__ inline float selq_window_value(int ws, unsigned int slot, int lane)
{
    const double centre = (double)(ws + (int)slot) / (double)SELQ_WIN_SCALE;
    const float bias = selq_window_bias(ws);
    float found = __builtin_nanf("");
    for (int den = 1 + lane; den <= 510; den += 64) {
        const float n = (float)__builtin_rint(centre * (double)den);
        if (__builtin_fabsf(n) > (float)den) continue;
        const float q = exact_quot(n, (float)den);
        if (selq_window_word(q, bias) == 64 + (int)slot) found = q + 0.0f;      // -0/den -> +0.0
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(found, off);
        if (found != found) found = o;
    }
    return found;
}

// After the statistics kernel: for every requested (stream, track) the rank is looked up in the window's counts
// (below <= rank < below + mass of the window); a tile whose every order statistic was found is marked done (pad[2]) and
// never sees the classic passes.  One block per tile, one wave per (stream, track); lane l owns 30 of the 1920 slots.
__global__ __launch_bounds__(256) void k_selq_pick_window(SelQTile *state, const unsigned int *win, const unsigned int *win_hist,
                                                          const unsigned int *below, float *out, unsigned int *pending, unsigned int *tile_list)
{
    __shared__ unsigned int s_ok[4];
    const long long tile = blockIdx.x;
    const int combo = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int stream = combo >> 1;
    const bool active = (state[tile].streams >> stream) & 1u;
    const unsigned int rank = state[tile].rank[combo];              // absolute: (N - 1) / 2 or N / 2 (k_selq_init)
    const int ws = (int)win[tile * 2 + stream];
    const unsigned int lo = below[tile * 2 + stream];
    bool ok = !active;
    float v = __builtin_nanf("");
    if (active && rank >= lo) {
        constexpr int PER = SELQ_WIN_SLOTS / 64;
        static_assert(SELQ_WIN_SLOTS == 64 * PER, "whole slots per lane");
        const unsigned int *mine = win_hist + (tile * 2 + stream) * SELQ_WIN_SLOTS + lane * PER;
        const unsigned int r = rank - lo;
        unsigned int c[PER], local = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { c[j] = mine[j]; local += c[j]; }
        unsigned int incl = local;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        unsigned int cum = incl - local;
        const bool holder = r >= cum && r < incl;
        int d = 0;
        if (holder) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (r >= cum + c[j]) { cum += c[j]; d = j + 1; }
                else break;
            }
        }
        const unsigned long long who = __ballot(holder);
        if (who) {
            const int src = __ffsll((long long)who) - 1;
            const unsigned int slot = (unsigned int)__shfl(lane * PER + d, src);
            v = selq_window_value(ws, slot, lane);
            ok = v == v;
        }
    }
    if (lane == 0) {
        s_ok[combo] = ok ? 1u : 0u;
        if (active && ok) out[tile * 4 + combo] = v;
        if (!active) out[tile * 4 + combo] = __builtin_nanf("");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int done = s_ok[0] & s_ok[1] & s_ok[2] & s_ok[3];
        state[tile].pad[2] = done;
        if (!done) tile_list[atomicAdd(pending, 1u)] = (unsigned int)tile;      // tiles that still need the classic passes
    }
}

}  // namespace lars

using namespace lars;

namespace lars {

static dim3 selq_grid(long long ntiles, long long npix, long long total = 2048)
{
    long long bpt = (total + ntiles - 1) / ntiles;                 // ~2048 workgroups per launch
    const long long cap = (npix / 4 + 1024 * 8 - 1) / (1024 * 8);  // at least ~8 steps per block (64 KiB table each)
    if (bpt > cap) bpt = cap;
    if (bpt < 1) bpt = 1;
    return dim3((unsigned)bpt, (unsigned)ntiles);
}

template <bool PER_TILE>
static void selq_launch(bool wb, unsigned streams, dim3 grid, hipStream_t s, const SelQParams &P)
{
#define SELQ_GO(W, S) hipLaunchKernelGGL((k_selq_pass<W, PER_TILE, S>), grid, dim3(1024), 0, s, P)
    if (wb) { if (streams == 1u) SELQ_GO(true, 1u); else if (streams == 2u) SELQ_GO(true, 2u); else SELQ_GO(true, 3u); }
    else { if (streams == 1u) SELQ_GO(false, 1u); else if (streams == 2u) SELQ_GO(false, 2u); else SELQ_GO(false, 3u); }
#undef SELQ_GO
}

int selq_pass_launch(const uint8_t *tiles, const uint8_t *wb_table, long long ntiles, long long npix, int first,
                     const unsigned int bucket[4], unsigned long long *hist, hipStream_t s, unsigned streams)
{
    SelQParams P;
    memset(&P, 0, sizeof P);
    P.tiles = tiles; P.wb_table = wb_table; P.npix = npix; P.first = first;
    for (int c = 0; c < 4; ++c) P.bucket[c] = bucket[c];
    P.hist = hist;
    selq_launch<false>(wb_table != nullptr, streams, selq_grid(ntiles, npix), s, P);
    return launch_check("k_selq_pass");
}

size_t selq_tile_scratch_bytes(long long ntiles)
{
    return (size_t)ntiles * (sizeof(SelQTile) + 4 * SELQ_BINS * sizeof(unsigned int) + 5 * sizeof(unsigned int) +
                             2 * SELQ_WIN_SLOTS * sizeof(unsigned int)) + 2048 + 512;
}
// behind the per-tile histograms: the predicted windows [ntiles][2], the counts below them [ntiles][2], the slot counts
// [ntiles][2][SELQ_WIN_SLOTS] and one word for the number of tiles the window did not serve
struct SelQWindow {
    unsigned int *win, *below, *win_hist, *pending, *tile_list;
    size_t zero_bytes;                     // below .. pending are contiguous: one memset
};
static SelQWindow selq_window_layout(void *scratch, long long ntiles)
{
    SelQWindow w;
    char *p = reinterpret_cast<char *>(selq_tile_hist32(scratch, ntiles) + (size_t)ntiles * 4 * SELQ_BINS);
    p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255);
    w.win = reinterpret_cast<unsigned int *>(p);
    p += ((size_t)ntiles * 2 * sizeof(unsigned int) + 255) & ~(size_t)255;
    w.below = reinterpret_cast<unsigned int *>(p);
    w.win_hist = w.below + (((size_t)ntiles * 2 + 63) & ~(size_t)63);
    w.pending = w.win_hist + (size_t)ntiles * 2 * SELQ_WIN_SLOTS;
    w.zero_bytes = (size_t)(reinterpret_cast<char *>(w.pending + 1) - reinterpret_cast<char *>(w.below));
    w.tile_list = w.pending + 64;
    return w;
}

// medians of every tile: bucket pass + slot pass, picks on the device, no host round trip
// selq_tile_prepare: state + zeroed histograms (before a fused statistics + bucket pass); selq_tile_hist32: where that
// pass adds its counts; selq_tile_medians_launch(..., first_pass_done): the remaining pass.
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
                             void *scratch, hipStream_t s, bool first_pass_done, unsigned streams, bool windowed)
{
    SelQTile *state; unsigned int *hist32;
    selq_scratch_layout(scratch, ntiles, &state, &hist32);
    const unsigned int *tile_list = nullptr;
    long long nrun = ntiles;
    if (windowed) {
        // the statistics kernel counted the windows (not the buckets): finish every tile whose ranks fall inside
        const SelQWindow w = selq_window_layout(scratch, ntiles);
        hipLaunchKernelGGL(k_selq_pick_window, dim3((unsigned)ntiles), dim3(256), 0, s, state, w.win, w.win_hist, w.below, out_pairs, w.pending,
                           w.tile_list);
        // How many tiles are left decides whether (and over how many tiles) the classic passes are launched at all, so the
        // host looks at the count -- the one place where a device entry point waits for its stream (the callers read the
        // medians back right after it anyway).
        unsigned int left = 1;
        LARS_HIP_TRY(hipMemcpyAsync(&left, w.pending, sizeof left, hipMemcpyDeviceToHost, s));
        LARS_HIP_TRY(hipStreamSynchronize(s));
        if (left == 0) return launch_check("selq_tile_medians (window)");
        first_pass_done = false;                              // the tiles on the list take both classic passes
        tile_list = w.tile_list;
        nrun = left;
    }
    if (!first_pass_done && !windowed) LARS_TRY(selq_tile_prepare(scratch, ntiles, npix, s, streams));
    // listed tiles: as many workgroups as a whole batch gets (the passes are bound by the steps a wave runs through, not by
    // what a workgroup pays for its table and counters: tools/selqbench.py, 64 .. 2048 workgroups)
    const dim3 grid = selq_grid(nrun, npix, tile_list && tuning().selq_list_wgs > 0 ? tuning().selq_list_wgs : 2048);
    for (int p = 0; p < 2; ++p) {
        SelQParams P;
        memset(&P, 0, sizeof P);
        P.tiles = tiles; P.wb_table = wb_table; P.npix = npix; P.first = p == 0;
        P.state = state; P.hist32 = hist32; P.tile_list = tile_list;
        if (p == 0 && first_pass_done) { /* counted by the statistics kernel */ }
        else selq_launch<true>(wb_table != nullptr, streams, grid, s, P);
        hipLaunchKernelGGL(k_selq_pick, dim3((unsigned)nrun), dim3(256), 0, s, state, hist32, p == 0 ? 1 : 0, out_pairs, tile_list);
    }
    return launch_check("selq_tile_medians");
}

}  // namespace lars

// ===========================================================================
// entry points
// ===========================================================================
extern "C" int lars_d_quotient_select_hist(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                           const uint8_t *wb_table, uint32_t streams, int first, const uint32_t bucket[4],
                                           uint64_t *hist, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || !hist || !bucket || ntiles <= 0 || npix <= 0 || streams < 1u || streams > 3u)
        return fail(LARS_ERR_INVALID, "lars_d_quotient_select_hist: bad arguments");
    if (dtype != LARS_U8 || channels != 3 || (reinterpret_cast<uintptr_t>(tiles) & 3) || (ntiles > 1 && (npix & 3)))
        return fail(LARS_ERR_INVALID, "lars_d_quotient_select_hist: uint8 [ntiles][npix][3] tiles on 4-byte boundaries are required");
    if (first < 0 || first > 3) return fail(LARS_ERR_INVALID, "lars_d_quotient_select_hist: first must be 0 .. 3");
    for (int k = 0; k < 4 && first != 2; ++k)
        if (bucket[k] >= SELQ_BINS) return fail(LARS_ERR_INVALID, "lars_d_quotient_select_hist: bucket must be below 2048");
    if (first == 2)
        for (int k = 0; k < 4; k += 2)
            if ((int)bucket[k] < SELQ_WIN_BOTTOM || (int)bucket[k] > -SELQ_WIN_BOTTOM + 1 - SELQ_WIN_SLOTS)
                return fail(LARS_ERR_INVALID, "lars_d_quotient_select_hist: window start outside [-524032, 522113]");
    if (ntiles > 65535 || (long long)npix * 6 >= (1ll << 30))
        return fail(LARS_ERR_INVALID, "lars_d_quotient_select_hist: at most 65535 tiles of < 2^30 / 6 pixels per launch");
    return selq_pass_launch(static_cast<const uint8_t *>(tiles), wb_table, ntiles, npix, first, bucket,
                            reinterpret_cast<unsigned long long *>(hist), pick_stream(c, stream), streams);
}

extern "C" size_t lars_quotient_median_scratch_bytes(int64_t ntiles) { return selq_tile_scratch_bytes(ntiles > 0 ? ntiles : 1); }

extern "C" int lars_d_quotient_median_pairs(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                            const uint8_t *wb_table, uint32_t streams, float *out_pairs, void *scratch, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!tiles || !out_pairs || !scratch || ntiles <= 0 || npix <= 0 || streams < 1u || streams > 3u)
        return fail(LARS_ERR_INVALID, "lars_d_quotient_median_pairs: bad arguments");
    if (dtype != LARS_U8 || channels != 3 || (reinterpret_cast<uintptr_t>(tiles) & 3) || (ntiles > 1 && (npix & 3)))
        return fail(LARS_ERR_INVALID, "lars_d_quotient_median_pairs: uint8 [ntiles][npix][3] tiles on 4-byte boundaries are required");
    if (ntiles > 65535 || (long long)npix * 6 >= (1ll << 30))
        return fail(LARS_ERR_INVALID, "lars_d_quotient_median_pairs: at most 65535 tiles of < 2^30 / 6 pixels");
    hipStream_t s = pick_stream(c, stream);
    const uint8_t *t8 = static_cast<const uint8_t *>(tiles);
    if (tuning().selq_window == 0)
        return selq_tile_medians_launch(t8, wb_table, ntiles, npix, out_pairs, scratch, s, false, streams, false);
    // one pass where the predicted window holds the ranks (see lars_d_stats_medians): prediction from a subsample, ONE
    // sweep that counts the windows (k_selq_pass, first == 2), and the two classic passes only over the tiles that missed
    LARS_TRY(selq_tile_prepare(scratch, ntiles, npix, s, streams));
    const SelQWindow w = selq_window_layout(scratch, ntiles);
    LARS_HIP_TRY(hipMemsetAsync(w.below, 0, w.zero_bytes, s));
    if (wb_table) hipLaunchKernelGGL((k_selq_predict<true>), dim3((unsigned)ntiles), dim3(1024), 0, s, t8, wb_table, (long long)npix, streams, w.win);
    else hipLaunchKernelGGL((k_selq_predict<false>), dim3((unsigned)ntiles), dim3(1024), 0, s, t8, wb_table, (long long)npix, streams, w.win);
    if (tuning().selq_window == 2)
        hipLaunchKernelGGL(k_selq_fill, dim3((unsigned)((ntiles * 2 + 255) / 256)), dim3(256), 0, s, w.win, ntiles * 2, (unsigned)SELQ_WIN_BOTTOM);
    SelQParams P;
    memset(&P, 0, sizeof P);
    P.tiles = t8; P.wb_table = wb_table; P.npix = npix; P.first = 2;
    selq_scratch_layout(scratch, ntiles, &P.state, &P.hist32);
    P.win = w.win; P.win_hist = w.win_hist; P.below = w.below;
    selq_launch<true>(wb_table != nullptr, streams, selq_grid(ntiles, npix), s, P);
    LARS_TRY(launch_check("lars_d_quotient_median_pairs (window pass)"));
    return selq_tile_medians_launch(t8, wb_table, ntiles, npix, out_pairs, scratch, s, true, streams, true);
}

// Statistics AND the exact median of every tile in two passes over the tiles (three with the white-balance histogram
// pass): the statistics kernel also counts the select's bucket pass, one slot pass follows.  Same constraints as lars_d_quotient_median_pairs; no output planes.
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
    if (a->ntiles > 65535 || (long long)a->npix * 6 >= (1ll << 30))
        return fail(LARS_ERR_INVALID, "lars_d_stats_medians: at most 65535 tiles of < 2^30 / 6 pixels");
    hipStream_t s = pick_stream(c, a->stream);
    const int stats_mode = (a->flags & LARS_F_SUMSQ) ? 3 : (a->flags & LARS_F_HIST) ? 2 : 1;
    const uint8_t *tiles = static_cast<const uint8_t *>(a->tiles);

    FusedParams P;
    memset(&P, 0, sizeof P);
    P.tiles = a->tiles; P.npix = a->npix; P.channels = 3; P.wb_table = a->wb_table; P.stats = a->stats; P.mask = mask;
    P.flags = a->flags & 7u;
    const unsigned streams = ((mask & 1u) ? 1u : 0u) | ((mask & 6u) ? 2u : 0u);
    LARS_TRY(selq_tile_prepare(scratch, a->ntiles, a->npix, s, streams));
    P.sel_hist = selq_tile_hist32(scratch, a->ntiles);
    // one-pass medians: predict a window of SELQ_WIN buckets per tile and stream from a subsample; the statistics kernel
    // then counts the window's slots as well, and only tiles whose median bucket falls outside take the slot pass.
    // lars_set_tuning("selq_window", 0) = always two passes; 2 = predict, then point every window at bucket 0 (tests the fallback)
    const bool windowed = stats_mode == 1 && tuning().selq_window != 0;
    if (windowed) {
        const SelQWindow w = selq_window_layout(scratch, a->ntiles);
        LARS_HIP_TRY(hipMemsetAsync(w.below, 0, w.zero_bytes, s));
        if (a->wb_table) hipLaunchKernelGGL((k_selq_predict<true>), dim3((unsigned)a->ntiles), dim3(1024), 0, s, tiles, a->wb_table, (long long)a->npix, streams, w.win);
        else hipLaunchKernelGGL((k_selq_predict<false>), dim3((unsigned)a->ntiles), dim3(1024), 0, s, tiles, a->wb_table, (long long)a->npix, streams, w.win);
        if (tuning().selq_window == 2) {
            // test hook: every window at the bottom of the range, so that every tile misses and takes the classic passes
            hipLaunchKernelGGL(k_selq_fill, dim3((unsigned)((a->ntiles * 2 + 255) / 256)), dim3(256), 0, s, w.win, a->ntiles * 2, (unsigned)SELQ_WIN_BOTTOM);
        }
        P.sel_win = w.win;
        P.sel_win_hist = w.win_hist;
        P.sel_below = w.below;
    }
    const long long nrec = a->ntiles * 3;
    stats_init_launch(a->stats, nrec, mask, s);
    dim3 grid(blocks_per_tile(a->npix / 4 + 1, a->ntiles, fused_v2_threads(false), 8192), (unsigned)a->ntiles);
    fused_v2_sel_launch(mask, a->wb_table != nullptr, stats_mode, grid, s, P);
    stats_finalize_launch(a->stats, nrec, mask, (long long)a->npix, s);
    LARS_TRY(launch_check("lars_d_stats_medians"));
    return selq_tile_medians_launch(tiles, a->wb_table, a->ntiles, a->npix, out_pairs, scratch, s, true, streams, windowed);
}
