// One-read statistics of uint8 RGNir tiles through joint byte-pair histograms.
//
// Every statistic analyze_index reports (process-images.py:506-512: mean, MEDIAN, min, max, coverage), the 50-bin
// histogram (process-ndvi.py:97) and the percentiles of the white balance itself (process-images.py:437) are exact
// functions of how often each pair of raw bytes occurs: NDVI = (n' - r') / (n' + r') with n' = T_n[n], r' = T_r[r], and
// the tables T follow from the channel histograms, which are the marginals of the pair counts.  So the tile is read
// ONCE, by a kernel that does no table look-up, no quotient and no float64 sum per pixel -- it only counts the pair
// (n, r) for the NDVI stream and (n, g) for the GNDVI / NDWI stream -- and a small second kernel per tile and stream
// derives everything from the 65536 counts: marginals -> np.percentile -> white-balance tables -> the value of every
// cell (the same correctly rounded float32 quotient the fused kernels produce) -> sums in 2^-32 fixed point, extrema,
// coverage, bins, and the exact median by a weighted two-level select.  The records are the same bits the per-pixel
// kernels produce, because every ingredient is a function of the counts.
//
// k_joint_count: one 1024-thread workgroup per (tile chunk, stream) with the stream's 65536 counters in LDS as 16-bit
// pairs: dword D holds in its low half the pixels of BOTH its cells (D, h = 0) and (D, 1) and in its high half those of (D, 1),
// so one ds_add of 1 | (h << 16) per pixel counts it and the high half can never pass the low one.  128 KiB of the
// 160 KiB: one workgroup per CU.  With two streams the two workgroups of a tile chunk sit 8 apart in dispatch order (the same
// XCD, at the same time): the second reader of a line finds it in that XCD's L2 or in the Infinity Cache.
//
// 16-bit sums cannot wrap between two sweeps: a workgroup adds 12 x 4096 = 49152 pixels per period, and at every period's
// end (two barriers) it moves each dword whose sum has reached 16384 onto a list in LDS (16383 + 49152 = 65535); a workgroup
// counts at most 2^24 pixels, so at most 1024 moves.  Round 4 built the barrier-free alternative -- every add returns the dword's
// previous value, the lane whose add passes the mark moves the dword (ds_sub of exactly what it held) -- and measured it: in
// isolation 7.8 against 10.5 cycles per wave-add (tools/lab/ldsatomics.py), in this kernel nothing (2.92 against 2.96 ms per
// 256 tiles, two streams; one stream 2.30 against 2.24: the sweeps hide under the memory waits), and where one colour takes
// most of the pixels the mover falls a full mark behind its add and the pair runs away.  Removed again (commit 1c850f5).
// Cell of a pixel (x = r or g, n = NIR): m = (n + 5 x) & 255, dword D = x << 7 | (m & 127), half h = m >> 7.  The LDS bank of a
// dword is D & 31 = (n + 5 x) & 31: the 5 x 5 neighbourhood of cells that a wave's 64 pixels of a smooth image fall into lands in
// 25 DIFFERENT banks (and dwords), as do common shifts of both samples (shading: +6 per level) and shifts of one sample alone.
// Round 3's n ^ 2x put such a neighbourhood into about six banks (tools/lab/banksim.py: 10-12 lanes on the fullest bank against
// 5.4 for independent samples; the linear form: 5.5-5.7).
//
// Since round 5 this file is one of two counting kernels.  With both value streams, white balance and no channel histograms wanted,
// lars_d_stats_joint first asks k_joint_predict (joint_win.hip) for windows on red and green; tiles whose windows fit are counted by
// k_joint_count_win -- both tables in ONE workgroup, one reader per byte -- and published in this file's layout; everything else
// (one value stream, windows that do not fit, channel histograms, and the recount of a tile whose window missed) is counted here.
// k_joint_finish serves both: it walks only the blocks of 8 table rows the counting kernels report as occupied, and for windowed
// counts it checks that np.percentile's order statistics fell strictly inside the window before it writes a record.
#include <vector>

#include "joint_device.h"

namespace lars {

// CH = 4: RGBA tiles, one 16-byte load per lane and step, repacked into the three dwords of an RGB quad.
template <int DEPTH, int CH = 3>
__global__ __launch_bounds__(JH_THREADS, 4) void k_joint_count(JointCountParams P)
{
    // a period is a whole number of ring turns and adds at most 65535 - 16383 = 49152 pixels (12 steps) to any one dword
    constexpr int PERIOD = (JH_PERIOD_STEPS % DEPTH == 0) ? JH_PERIOD_STEPS : DEPTH;
    static_assert(PERIOD % DEPTH == 0 && PERIOD <= JH_PERIOD_STEPS, "a period is a whole number of ring turns, at most 12 steps");
    __shared__ __attribute__((aligned(16))) unsigned int s_tab[JH_DWORDS];          // 128 KiB
    __shared__ uint2 s_list[JH_LIST_CAP];                                          // moved dwords: (D, value)
    __shared__ unsigned int s_nlist;
    __shared__ unsigned int s_jlo, s_jhi;                                          // blocks of 8 rows (1024 dwords) that hold a count

    const int tid = threadIdx.x;
    // unit = (tile, chunk); with two streams the two workgroups of a unit are 8 apart in dispatch order
    long long unit;
    int role;
    if (P.S == 2) {
        const unsigned int b = blockIdx.x;
        unit = (long long)(b >> 4) * 8 + (b & 7u);
        role = (int)((b >> 3) & 1u);
    } else {
        unit = blockIdx.x;
        role = 0;
    }
    const long long tile = unit / P.K;
    const int chunk = (int)(unit - tile * P.K);
    if (tile >= P.ntiles) return;
    if (P.win) {
        // windowed tiles belong to k_joint_count_win; the second pass counts only the tiles whose window missed (k_joint_finish's flag)
        const JointWin wn = P.win[tile];
        if (P.pass == 0 ? wn.mode != 0u : wn.flag == 0u) return;
    }
    const bool green = P.S == 2 ? role == 1 : (P.streams == 2u);                   // which sample pairs with NIR

    uint4 *tab4 = reinterpret_cast<uint4 *>(s_tab);
    for (int i = tid; i < JH_DWORDS / 4; i += JH_THREADS) tab4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) { s_nlist = 0; s_jlo = JH_DWORDS / JH_THREADS; s_jhi = 0u; }
    __syncthreads();
    char *tab = reinterpret_cast<char *>(s_tab);

    const long long nquads_tile = P.npix >> 2;
    const long long q_begin = (long long)chunk * P.chunk_quads;
    long long q_end = chunk == P.K - 1 ? nquads_tile : q_begin + P.chunk_quads;
    if (q_end > nquads_tile) q_end = nquads_tile;
    const long long nq = q_end > q_begin ? q_end - q_begin : 0;
    const uint8_t *tile_base = P.tiles + tile * P.npix * CH;

    // byte selectors (wave-uniform): bytes of a quad r0 g0 n0 r1 | g1 n1 r2 g2 | n2 r3 g3 n3
    // pixels 0, 1 from (w0, w1): perm(src0 = w1, src1 = w0) -> selector = byte offset 0..7
    // pixels 2, 3 from (w1, w2): perm(src0 = w2, src1 = w1) -> selector = byte offset - 4
    const unsigned int xo = green ? 1u : 0u;
    const unsigned int seln01 = 0x02u | (0x0cu << 8) | (0x05u << 16) | (0x0cu << 24);           // n0 0 n1 0
    const unsigned int selx01 = (0u + xo) | (0x0cu << 8) | ((3u + xo) << 16) | (0x0cu << 24);   // x0 0 x1 0
    const unsigned int seln23 = 0x04u | (0x0cu << 8) | (0x07u << 16) | (0x0cu << 24);           // n2 0 n3 0
    const unsigned int selx23 = (2u + xo) | (0x0cu << 8) | ((5u + xo) << 16) | (0x0cu << 24);

    // Two pixels at a time, one in each 16-bit half: nn = n | n' << 16, px = x | x' << 16.  m2 = nn + 5 px holds n + 5 x (< 2048) in
    // each half, W = px << 7 | (m2 & 0x7F007F) the two dword indices D; the half bit h is bit 7 of each m.
    auto count_pair = [&](unsigned int nn, unsigned int px) {
        const unsigned int m2 = jh_mad5(px, nn);                                // v_mad_u32_u24: px < 2^24
        const unsigned int W = (px << 7) | (m2 & 0x007F007Fu);
        const unsigned int a0 = (W << 2) & 0x1FFFCu, a1 = (W >> 14) & 0x1FFFCu;
        const unsigned int v0 = ((m2 << 9) & 0x10000u) | 1u, v1 = ((m2 >> 7) & 0x10000u) | 1u;
        jh_add(a0, v0, tab);
        jh_add(a1, v1, tab);
    };
    // Runs of equal pixels would queue the lanes of a wave on a few LDS words per atomic: the lanes that start a run add its whole count
    // (joint_device.h).  For textured content the test is one v_mov_b32_dpp + one compare per quad.
    auto count_pair_n = [&](unsigned int nn, unsigned int px, unsigned int n) {
        const unsigned int m2 = jh_mad5(px, nn);
        const unsigned int W = (px << 7) | (m2 & 0x007F007Fu);
        const unsigned int a0 = (W << 2) & 0x1FFFCu, v0 = (((m2 << 9) & 0x10000u) | 1u) * n;
        const unsigned int a1 = (W >> 14) & 0x1FFFCu, v1 = (((m2 >> 7) & 0x10000u) | 1u) * n;
        jh_add(a0, v0, tab);
        jh_add(a1, v1, tab);
    };
    auto do_quad = [&](unsigned int w0, unsigned int w1, unsigned int w2) {
        if (jh_mostly_runs(w0)) {
            const bool head = ((jh_prev_lane(w0) ^ w0) | (jh_prev_lane(w1) ^ w1) | (jh_prev_lane(w2) ^ w2)) != 0u;   // every lane takes part in the three DPP moves
            const unsigned int n = jh_run_length(head, tid & 63);
            if (head) {
                count_pair_n(__builtin_amdgcn_perm(w1, w0, seln01), __builtin_amdgcn_perm(w1, w0, selx01), n);
                count_pair_n(__builtin_amdgcn_perm(w2, w1, seln23), __builtin_amdgcn_perm(w2, w1, selx23), n);
            }
            return;
        }
        count_pair(__builtin_amdgcn_perm(w1, w0, seln01), __builtin_amdgcn_perm(w1, w0, selx01));
        count_pair(__builtin_amdgcn_perm(w2, w1, seln23), __builtin_amdgcn_perm(w2, w1, selx23));
    };

    // tail pixels of the tile (npix % 4): its last chunk, before the first period (which starts from zero counts)
    if (chunk == P.K - 1 && tid < (int)(P.npix & 3)) {
        const uint8_t *p = tile_base + (nquads_tile * 4 + tid) * CH;
        const unsigned int n = p[2], x = green ? p[1] : p[0];
        const unsigned int m = jh_m(n, x);
        jh_add(((x << 7) | (m & 127u)) << 2, ((m >> 7) << 16) | 1u, tab);
    }

    // A scan: every dword whose sum (low half) has reached 16384 moves onto the list.  Between the two barriers nobody adds.
    auto scan = [&]() {
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < JH_DWORDS / 4 / JH_THREADS; ++i) {
            const int idx = tid + i * JH_THREADS;
            const uint4 v = tab4[idx];
            if ((v.x | v.y | v.z | v.w) & JH_PROMOTE_MASK) {
                const unsigned int c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (c[k] & JH_PROMOTE_MASK) {
                        const unsigned int slot = atomicAdd(&s_nlist, 1u);
                        if (slot < JH_LIST_CAP) s_list[slot] = make_uint2((unsigned)(idx * 4 + k), c[k]);
                        else atomicExch(P.error, 1u);
                        s_tab[idx * 4 + k] = 0u;
                    }
                }
            }
        }
        __syncthreads();
    };

    // Software pipeline: DEPTH 12-byte buffer loads in flight per lane; a step = 1024 quads = 12 KiB contiguous.
    const long long nfull = nq >> 10;                       // steps in which every lane has a quad
    const int rem = (int)(nq & 1023);
    if (nq > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(tile_base + q_begin * (CH * 4)), 0, (int)(nq * (CH * 4)), 0x00020000);
        const unsigned int voff = (unsigned int)tid * (CH * 4u);
        constexpr unsigned int STEP_B = JH_THREADS * CH * 4u;
        typedef unsigned int u32x4j __attribute__((ext_vector_type(4)));
        typename std::conditional<CH == 4, u32x4j, u32x3>::type w[DEPTH];
        auto load = [&](unsigned int soff_b) {
            if constexpr (CH == 4) return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff_b, 0);
            else return __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff_b, 0);
        };
        auto count = [&](const auto &v) {
            if constexpr (CH == 4)
                do_quad(__builtin_amdgcn_perm(v.y, v.x, 0x04020100u), __builtin_amdgcn_perm(v.z, v.y, 0x05040201u),
                        __builtin_amdgcn_perm(v.w, v.z, 0x06050402u));
            else do_quad(v.x, v.y, v.z);
        };
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) w[k] = load((unsigned)k * STEP_B);
        long long it = 0;
        unsigned int soff = DEPTH * STEP_B;
        int since = 0;
        for (; it + DEPTH <= nfull; it += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                count(w[k]);
                __builtin_amdgcn_sched_barrier(0);
                w[k] = load(soff + (unsigned)k * STEP_B);                 // past the end: zeros, never counted
                __builtin_amdgcn_sched_barrier(0);
            }
            soff += DEPTH * STEP_B;
            since += DEPTH;
            if (since == PERIOD) {
                scan();
                since = 0;
            }
        }
        // the last (fewer than DEPTH) full steps and the ragged one: since + DEPTH <= PERIOD, no scan needed
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const long long step = it + k;
            if (step < nfull || (step == nfull && tid < rem)) count(w[k]);
        }
    }
    __syncthreads();

    // publish: (cell (D, 0), cell (D, 1)) = (low - high, high), 32 bytes per lane and trip -- and which blocks of 8 rows (1024 dwords: the unit
    // k_joint_finish walks) hold a count at all: a tile's samples seldom span all 256 values, and the finish kernel skips the rest
    const long long slot = (tile * P.S + role) * P.K + chunk;
    unsigned int *out = P.part + slot * (long long)(2 * JH_DWORDS);
    for (int i = tid; i < JH_DWORDS / 4; i += JH_THREADS) {
        const uint4 v = tab4[i];
        uint4 *o = reinterpret_cast<uint4 *>(out + (long long)i * 8);
        o[0] = make_uint4((v.x & 0xFFFFu) - (v.x >> 16), v.x >> 16, (v.y & 0xFFFFu) - (v.y >> 16), v.y >> 16);
        o[1] = make_uint4((v.z & 0xFFFFu) - (v.z >> 16), v.z >> 16, (v.w & 0xFFFFu) - (v.w >> 16), v.w >> 16);
        const unsigned int jb = (unsigned)i >> 8;                                  // the same for the 64 lanes of a wave
        if (__builtin_amdgcn_ballot_w64((v.x | v.y | v.z | v.w) != 0u) != 0ull && (tid & 63) == 0) {
            atomicMin(&s_jlo, jb);
            atomicMax(&s_jhi, jb + 1u);
        }
    }
    const unsigned int nlist = s_nlist < JH_LIST_CAP ? s_nlist : JH_LIST_CAP;
    if (nlist) {
        // the stores above have to be in the L2 before the adds to the same words: the barrier's own workgroup-scope release (stores
        // acknowledged) orders them, and the adds are workgroup-scope too -- every access stays in this XCD's L2.  (A device-scope
        // __threadfence here writes the XCD's L2 back for every workgroup that has a list: + 40 % on the launch of any batch whose
        // chunks hold a cell of 16384 pixels: profiles/r05_joint_publish_fence.txt.)
        __syncthreads();
        for (unsigned int e = tid; e < nlist; e += JH_THREADS) {
            const uint2 m = s_list[e];
            const unsigned int hi = m.y >> 16, lo = (m.y & 0xFFFFu) - hi;
            if (lo) jh_publish_add(&out[2 * (long long)m.x], lo);
            if (hi) jh_publish_add(&out[2 * (long long)m.x + 1], hi);
            atomicMin(&s_jlo, m.x >> 10);
            atomicMax(&s_jhi, (m.x >> 10) + 1u);
        }
    }
    __syncthreads();
    if (tid == 0) P.rows[slot] = make_uint2(s_jlo, s_jhi);
}

// ---------------------------------------------------------------------------
// counts -> marginals -> percentiles -> tables -> statistics + exact medians
// ---------------------------------------------------------------------------
struct JointFinishParams {
    const unsigned int *part;
    const uint2 *rows;                        // [ntiles][S][K]: blocks of 8 rows that hold counts (see JointCountParams)
    long long npix;
    int K, S;
    unsigned int streams;                     // as JointCountParams
    unsigned int mask;                        // LARS_MASK_*: which records to write
    unsigned int flags;                       // LARS_F_HIST / LARS_F_SUMSQ
    int wb;                                   // percentile white balance (process-images.py:437-441) or raw samples
    int rgn_variant;
    lars_stats *stats;                        // [ntiles][3]
    uint8_t *table_out;                       // [ntiles][3][256] or null
    double *pcts_out;                         // [ntiles][3][2] or null
    unsigned int *hist_out;                   // [ntiles][3][256] or null
    float *out_pairs;                         // [ntiles][2 streams][2] or null: the two middle order statistics
    JointWin *win;                            // [ntiles] or null: the windows the tiles were counted with (joint_win.hip)
    int pass;                                 // 0: every tile, windows checked; 1: only the tiles flagged by pass 0 (now counted on full tables)
};

// Exclusive prefix of one value per thread over the first 256 threads of the block (4 waves): wave scans + one barrier.
// Every thread of the block must call it (threads >= 256 pass 0 and get nothing useful back).  s_w: 4 words of LDS.
__device__ inline unsigned long long jf_prefix256(unsigned long long v, int tid, unsigned long long *s_w)
{
    unsigned long long inc = v;
    const int lane = tid & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();                                       // s_w may still be read by the previous call
    if (tid < 256 && lane == 63) s_w[tid >> 6] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (int w = 0; w < (tid >> 6) && w < 4; ++w) base += s_w[w];
    return base + inc - v;
}

// Position of two ranks in a histogram of NB bins (NB = 2048 or 1024): s_bin[k] = first bin whose cumulative count
// exceeds rank[k], s_before[k] = the count below that bin.  The first 256 threads own NB / 256 consecutive bins each.
template <int NB>
__device__ inline void jf_pick(const unsigned int *h, unsigned long long rank0, unsigned long long rank1,
                               unsigned long long *s_w, unsigned int *s_bin /*[2]*/, unsigned long long *s_before /*[2]*/, int tid)
{
    constexpr int PER = NB / 256;
    unsigned int mine[PER];
    unsigned long long local = 0;
    if (tid < 256) {
#pragma unroll
        for (int j = 0; j < PER; ++j) { mine[j] = h[tid * PER + j]; local += mine[j]; }
    }
    const unsigned long long before = jf_prefix256(local, tid, s_w);
    if (tid < 256) {
        const unsigned long long rk[2] = {rank0, rank1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (rk[k] >= before && rk[k] < before + local) {
                unsigned long long cum = before;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    if (rk[k] >= cum && rk[k] < cum + mine[j]) { s_bin[k] = (unsigned)(tid * PER + j); s_before[k] = cum; }
                    cum += mine[j];
                }
            }
        }
    }
    __syncthreads();
}

// 8 waves per SIMD = 64 registers: TWO of these 16-wave blocks share a CU.  Left to itself the compiler took 68 and the second half of the
// blocks waited for the first (profiles/r05_finish_block_timeline.txt: 97 -> 74 us per 256 tiles, with medians 177 -> 129).
__global__ __launch_bounds__(JH_THREADS, 8) void k_joint_finish(JointFinishParams P)
{
    __shared__ unsigned int s_hn[256], s_hx[256];
    __shared__ float s_fn[256], s_fx[256];
    __shared__ double s_ord[2][4], s_p[2][2];
    __shared__ unsigned int s_bucket[SELQ_BINS];
    __shared__ unsigned int s_slot[2][SELQ_SLOTS];
    __shared__ unsigned long long s_w[2][4];
    __shared__ unsigned int s_bin[2];
    __shared__ unsigned long long s_before[2];
    __shared__ unsigned int s_h50[2][LARS_HIST_BINS + 2];          // [plain | negated][bin + 1]
    __shared__ unsigned long long s_acc[4];                          // sum_fx, above (> 0.2f), below zero (< 0), unused
    __shared__ double s_wsq[JH_THREADS / 64];                        // per-wave sums of squares, folded in wave order
    __shared__ unsigned int s_mnk, s_mxk;
    __shared__ float s_med[2];

    const int tid = threadIdx.x;
    const int role = blockIdx.x;
    const long long tile = blockIdx.y;
    const bool green = P.S == 2 ? role == 1 : (P.streams == 2u);
    const int xch = green ? 1 : 0;                                   // channel of the sample paired with NIR
    const unsigned int *part = P.part + ((tile * P.S + role) * P.K) * (long long)(2 * JH_DWORDS);
    const bool want_hist = (P.flags & (LARS_F_HIST | LARS_F_SUMSQ)) != 0;
    const bool want_sq = (P.flags & LARS_F_SUMSQ) != 0;
    const bool want_v = !green && (P.mask & LARS_MASK_NDVI);
    const bool want_g = green && (P.mask & LARS_MASK_GNDVI);
    const bool want_w = green && (P.mask & LARS_MASK_NDWI);
    const bool medians = P.out_pairs != nullptr;
    // Windowed counts (joint_win.hip): samples of this stream's channel below win_lo were counted as win_lo, above win_hi as win_hi.
    bool windowed = false;
    unsigned int win_lo = 0u, win_hi = 255u, win_nlo = 0u, win_nhi = 255u;
    if (P.win) {
        const JointWin wn = P.win[tile];                                 // pass 0: the other stream's block may set .flag meanwhile; nothing else changes
        if (P.pass == 1 && wn.flag == 0u) return;
        if (P.pass == 0 && wn.mode != 0u && P.wb) {
            windowed = true;
            win_lo = green ? wn.lo_g : wn.lo_r;
            win_hi = win_lo + (green ? wn.ng : wn.nr) - 1u;
            win_nlo = wn.lo_n;                                            // NIR whole (mode 1): 0 .. 255, nothing to check
            win_nhi = wn.lo_n + wn.nn - 1u;
        }
    }

    if (tid < 256) { s_hn[tid] = 0; s_hx[tid] = 0; }
    for (int i = tid; i < SELQ_BINS; i += JH_THREADS) s_bucket[i] = 0;
    for (int i = tid; i < 2 * SELQ_SLOTS; i += JH_THREADS) (&s_slot[0][0])[i] = 0;
    if (tid < 2 * (LARS_HIST_BINS + 2)) (&s_h50[0][0])[tid] = 0;
    if (tid < 4) s_acc[tid] = 0;
    if (tid == 0) { s_mnk = 0xFFFFFFFFu; s_mxk = 0u; }
    if (tid < 2) s_med[tid] = __builtin_nanf("");

    // This thread's 64 cells: dwords D = j * 1024 + tid (j < 32), halves h = 0, 1; x = D >> 7 (the same for the 64 lanes of a
    // wave), n = ((D & 127 | h << 7) - 5 x) & 255 (jh_n_of).  Every pass below walks them in four groups of eight dwords, the next
    // group's loads (L2 hits: the counting kernel has just written them) in flight while the current one is worked on.
    constexpr int NJ = JH_DWORDS / JH_THREADS;                       // 32 blocks of 1024 dwords = 8 rows each
    constexpr int GRP = 8;
    // only the blocks of 8 rows in which some chunk of the tile counted something (windowed counts: the blocks of the window, the only ones
    // k_joint_count_win published): a tile whose samples span 96 of the 256 values is walked in 13 blocks instead of 32
    int j_lo = NJ, j_hi = 0;
    for (int k = 0; k < P.K; ++k) {
        const uint2 rr = P.rows[(tile * P.S + role) * P.K + k];
        j_lo = min(j_lo, (int)rr.x);
        j_hi = max(j_hi, (int)rr.y);
    }
    if (j_hi > NJ) j_hi = NJ;
    auto for_cells = [&](auto &&f) {
        uint2 buf[2][GRP];
        auto fetch = [&](int g, uint2 (&dst)[GRP]) {
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int j = j_lo + g * GRP + i;
                dst[i] = j < j_hi ? *reinterpret_cast<const uint2 *>(part + 2 * ((long long)j * JH_THREADS + tid)) : make_uint2(0u, 0u);
            }
            for (int k = 1; k < P.K; ++k) {
                const unsigned int *pk = part + (long long)k * (2 * JH_DWORDS);
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int j = j_lo + g * GRP + i;
                    if (j < j_hi) {
                        const uint2 v = *reinterpret_cast<const uint2 *>(pk + 2 * ((long long)j * JH_THREADS + tid));
                        dst[i].x += v.x; dst[i].y += v.y;
                    }
                }
            }
        };
        auto work = [&](int g, const uint2 (&src)[GRP]) {
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int j = j_lo + g * GRP + i;
                if (j < j_hi) f((unsigned)j * JH_THREADS + (unsigned)tid, src[i].x, src[i].y);       // uniform over the block
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const int ng = (j_hi - j_lo + GRP - 1) / GRP;
        fetch(0, buf[0]);
#pragma unroll 1
        for (int g = 0; g < ng; g += 2) {
            if (g + 1 < ng) fetch(g + 1, buf[1]);
            work(g, buf[0]);
            if (g + 2 < ng) fetch(g + 2, buf[0]);
            if (g + 1 < ng) work(g + 1, buf[1]);
        }
    };
    __syncthreads();

    // ---- marginals: the channel histograms np.percentile needs
    for_cells([&](unsigned int D, unsigned int c0, unsigned int c1) {
        const unsigned int x = D >> 7;
        const unsigned int n0 = jh_n_of(D, 0u), n1 = jh_n_of(D, 1u);
        if (c0) atomicAdd(&s_hn[n0], c0);
        if (c1) atomicAdd(&s_hn[n1], c1);
        unsigned int t = c0 + c1;
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
        if ((tid & 63) == 0 && t) atomicAdd(&s_hx[x], t);            // D >> 7 is the same for the 64 lanes of a wave
    });
    __syncthreads();
    if (P.hist_out) {
        unsigned int *ho = P.hist_out + tile * 768;
        if (tid < 256) {
            ho[xch * 256 + tid] = s_hx[tid];
            if (!green || P.S == 1) ho[512 + tid] = s_hn[tid];
        }
    }

    // ---- np.percentile(ch, (2, 98)), 'linear' + numpy's _lerp (the arithmetic of k_wb_table, fused.hip).
    // Threads 0..255 hold the NIR histogram, 256..511 the paired channel's: one bin each, prefix by wave scans.
    const long long npix = P.npix;
    if (P.wb) {
        const int ch = (tid >> 8) & 1, bin = tid & 255;
        const unsigned long long cnt = tid < 512 ? (unsigned long long)(ch ? s_hx : s_hn)[bin] : 0ull;
        // two independent 256-thread prefixes: threads 256..511 are shifted down for the helper
        unsigned long long before;
        {
            unsigned long long inc = cnt;
            const int lane = tid & 63;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            if (tid < 512 && lane == 63) s_w[ch][(tid >> 6) & 3] = inc;
            __syncthreads();
            unsigned long long base = 0;
            for (int w = 0; w < ((tid >> 6) & 3); ++w) base += s_w[ch][w];
            before = base + inc - cnt;
        }
        if (tid < 512 && cnt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double q = ((r >> 1) == 0 ? 2.0 : 98.0) / 100.0;
                const double vi = (double)(npix - 1) * q;
                long long rank = (long long)floor(vi);
                if (r & 1) { rank += 1; if (rank > npix - 1) rank = npix - 1; }
                if ((unsigned long long)rank >= before && (unsigned long long)rank < before + cnt) s_ord[ch][r] = (double)bin;
            }
        }
        __syncthreads();
        if (windowed) {
            // The clamped channel's cumulative counts are exact from win_lo up to win_hi - 1, so an order statistic found strictly
            // inside the window IS the tile's; one found on an edge (that is not the range's edge) may really lie beyond it.  All four
            // inside: win_lo < p2 and p98 < win_hi, every clamped sample has the level of its edge (0 resp. 255, process-images.py:438)
            // and the counts are the tile's statistics.  Otherwise: flag the tile, it is counted again on full tables.
            bool missed = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned int b = (unsigned int)s_ord[1][r], bn = (unsigned int)s_ord[0][r];
                if ((win_lo > 0u && b <= win_lo) || (win_hi < 255u && b >= win_hi)) missed = true;
                if ((win_nlo > 0u && bn <= win_nlo) || (win_nhi < 255u && bn >= win_nhi)) missed = true;
            }
            if (missed) {
                if (tid == 0) P.win[tile].flag = 1u;
                return;
            }
        }
        if (tid < 4) {
            const int pc = tid >> 1, k = tid & 1;
            const double q = (k == 0 ? 2.0 : 98.0) / 100.0;
            const double vi = (double)(npix - 1) * q;
            const double t = vi - floor(vi);
            const double a = s_ord[pc][2 * k], b = s_ord[pc][2 * k + 1];
            const double d = b - a;
            double r = a + d * t;
            if (t >= 0.5) r = b - d * (1.0 - t);
            s_p[pc][k] = r;
            if (P.pcts_out) {
                const int cc = pc ? xch : 2;
                if (pc || !green || P.S == 1) P.pcts_out[(tile * 3 + cc) * 2 + k] = r;
            }
        }
        __syncthreads();
    }
    if (tid < 512) {
        const int ch = tid >> 8, v = tid & 255;
        const unsigned int level = P.wb ? wb_level(v, s_p[ch][0], s_p[ch][1], P.rgn_variant) : (unsigned)v;
        (ch ? s_fx : s_fn)[v] = (float)level;
        if (P.wb && P.table_out) {
            const int cc = ch ? xch : 2;
            if (ch || !green || P.S == 1) P.table_out[(tile * 3 + cc) * 256 + v] = (uint8_t)level;
        }
    }
    __syncthreads();

    // ---- statistics of the stream's quotient (n' - x') / (n' + x') over the cells
    {
        long long sum_fx = 0;
        unsigned long long above = 0, below0 = 0;
        double sumsq = 0.0;
        float mn = __builtin_inff(), mx = -__builtin_inff();
        for_cells([&](unsigned int D, unsigned int c0, unsigned int c1) {
            const unsigned int cc[2] = {c0, c1};
            const unsigned int x = D >> 7;
            const float fx = s_fx[x];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned int cv = cc[h];
                if (!cv) continue;
                const unsigned int n = jh_n_of(D, (unsigned)h);
                const float q = norm_diff(s_fn[n], fx);
                const long long cnt = (long long)cv;
                sum_fx += cnt * (long long)((double)q * LARS_FX_SCALE);          // q is a multiple of 2^-32: exact
                mn = fminf(mn, q);
                mx = fmaxf(mx, q);
                if (q > 0.2f) above += (unsigned long long)cnt;
                if (q < 0.0f) below0 += (unsigned long long)cnt;
                if (want_sq) sumsq += (double)cnt * ((double)q * (double)q);
                if (want_hist) {
                    // bin + 1 in the low mantissa bits (hist_pos2, fused_v2.hip); 51 = the closed right edge
                    const float u = __builtin_fmaf(q, 25.0f, V2_HIST_MAGIC_C) + 8388608.0f;
                    atomicAdd(&s_h50[0][__builtin_bit_cast(unsigned int, u) & 0x7FFFFFu], cv);
                    if (want_w) {
                        const float uw = __builtin_fmaf(q, -25.0f, V2_HIST_MAGIC_C) + 8388608.0f;
                        atomicAdd(&s_h50[1][__builtin_bit_cast(unsigned int, uw) & 0x7FFFFFu], cv);
                    }
                }
                if (medians) atomicAdd(&s_bucket[selq_bucket_of(selq_t(q))], cv);
            }
        });
        for (int off = 32; off >= 1; off >>= 1) {
            sum_fx += __shfl_xor(sum_fx, off);
            above += __shfl_xor(above, off);
            below0 += __shfl_xor(below0, off);
            sumsq += __shfl_xor(sumsq, off);
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if ((tid & 63) == 0) {
            atomicAdd(&s_acc[0], (unsigned long long)sum_fx);
            atomicAdd(&s_acc[1], above);
            atomicAdd(&s_acc[2], below0);
            s_wsq[tid >> 6] = sumsq;
            atomicMin(&s_mnk, f32_key(mn));
            atomicMax(&s_mxk, f32_key(mx));
        }
    }
    __syncthreads();

    // ---- exact median: weighted two-level select (bucket, then slot: a slot holds one distinct quotient of bytes)
    if (medians) {
        const unsigned long long rank0 = (unsigned long long)((npix - 1) / 2), rank1 = (unsigned long long)(npix / 2);
        jf_pick<SELQ_BINS>(s_bucket, rank0, rank1, s_w[0], s_bin, s_before, tid);
        const unsigned int bk0 = s_bin[0], bk1 = s_bin[1];
        const unsigned long long in0 = rank0 - s_before[0], in1 = rank1 - s_before[1];
        __syncthreads();
        for_cells([&](unsigned int D, unsigned int c0, unsigned int c1) {
            const unsigned int cc[2] = {c0, c1};
            const unsigned int x = D >> 7;
            const float fx = s_fx[x];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned int cv = cc[h];
                if (!cv) continue;
                const unsigned int n = jh_n_of(D, (unsigned)h);
                const float t = selq_t(norm_diff(s_fn[n], fx));
                const unsigned int b = selq_bucket_of(t), sl = (__builtin_bit_cast(unsigned int, t) & 0xFFFu) >> 2;
                if (b == bk0) atomicAdd(&s_slot[0][sl], cv);
                if (b == bk1) atomicAdd(&s_slot[1][sl], cv);
            }
        });
        __syncthreads();
        // slot of each track inside its bucket (one histogram per call)
        jf_pick<SELQ_SLOTS>(s_slot[0], in0, in0, s_w[0], s_bin, s_before, tid);
        const unsigned int slot0 = s_bin[0];
        __syncthreads();
        jf_pick<SELQ_SLOTS>(s_slot[1], in1, in1, s_w[0], s_bin, s_before, tid);
        const unsigned int slot1 = s_bin[0];
        __syncthreads();
        for_cells([&](unsigned int D, unsigned int c0, unsigned int c1) {
            const unsigned int cc[2] = {c0, c1};
            const unsigned int x = D >> 7;
            const float fx = s_fx[x];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!cc[h]) continue;
                const unsigned int n = jh_n_of(D, (unsigned)h);
                const float q = norm_diff(s_fn[n], fx);
                const float t = selq_t(q);
                const unsigned int b = selq_bucket_of(t), sl = (__builtin_bit_cast(unsigned int, t) & 0xFFFu) >> 2;
                if (b == bk0 && sl == slot0) s_med[0] = q;            // every writer holds the same value
                if (b == bk1 && sl == slot1) s_med[1] = q;
            }
        });
        __syncthreads();
        if (tid < 2) P.out_pairs[(tile * 2 + (green ? 1 : 0)) * 2 + tid] = s_med[tid];
        if (P.S == 1 && tid >= 2 && tid < 4) P.out_pairs[(tile * 2 + (green ? 0 : 1)) * 2 + (tid - 2)] = __builtin_nanf("");   // the stream nobody asked for
    }

    // ---- the records, in their final form (what k_stats_init + the fused kernel + k_stats_finalize leave behind)
    if (tid < 2) {
        // tid 0: NDVI resp. GNDVI; tid 1: NDWI = -GNDVI (sum negates, extrema swap, coverage counts q < 0)
        const bool neg = tid == 1;
        const bool on = neg ? want_w : (want_v || want_g);
        if (on) {
            const int k = neg ? LARS_NDWI : (green ? LARS_GNDVI : LARS_NDVI);
            lars_stats *o = P.stats + tile * 3 + k;
            const long long s = (long long)s_acc[0];
            const double mnd = (double)key_f32(s_mnk), mxd = (double)key_f32(s_mxk);
            double sq = 0.0;
            for (int w = 0; w < JH_THREADS / 64; ++w) sq += s_wsq[w];
            o->sum = (double)(neg ? -s : s) * LARS_FX_INV;
            o->sumsq = want_sq ? (double)__double2ll_rn(sq * LARS_FX_SCALE) * LARS_FX_INV : 0.0;
            o->count = (uint64_t)npix;
            o->above = neg ? s_acc[2] : s_acc[1];
            o->nans = 0;
            o->min = neg ? 0.0 - mxd : mnd;
            o->max = neg ? 0.0 - mnd : mxd;
            o->threshold = neg ? 0.0 : (double)0.2f;
            o->index_id = (unsigned)k;
            o->reserved = 0;
        }
    }
    if (tid < 2 * LARS_HIST_BINS) {
        const bool neg = tid >= LARS_HIST_BINS;
        const int b = neg ? tid - LARS_HIST_BINS : tid;
        const bool on = neg ? want_w : (want_v || want_g);
        if (on) {
            const int k = neg ? LARS_NDWI : (green ? LARS_GNDVI : LARS_NDVI);
            unsigned long long v = 0;
            if (want_hist) {
                v = s_h50[neg ? 1 : 0][b + 1];
                if (b == LARS_HIST_BINS - 1) v += s_h50[neg ? 1 : 0][LARS_HIST_BINS + 1];      // x == 1.0: the closed last bin
            }
            P.stats[tile * 3 + k].hist[b] = v;
        }
    }
}
}  // namespace lars

using namespace lars;

// K: chunks per tile.  Enough workgroups to fill the chip (one per CU, a few rounds), at least npix / 2^24 (the 16-bit
// counters' list bound), chunks of whole steps.
static int joint_chunks(long long ntiles, long long npix, int S)
{
    const long long nquads = npix >> 2;
    long long k_min = (npix + JH_MAX_WG_PIXELS - 1) / JH_MAX_WG_PIXELS;
    if (k_min < 1) k_min = 1;
    long long k = tuning().blocks_per_tile > 0 ? tuning().blocks_per_tile : (1024 + ntiles * S - 1) / (ntiles * S);
    const long long k_max = nquads / (64 * 1024) > 1 ? nquads / (64 * 1024) : 1;          // at least 64 steps per workgroup
    if (k > k_max) k = k_max;
    if (k < k_min) k = k_min;
    if (k > 4096) k = 4096;
    return (int)k;
}
static long long joint_chunk_quads(long long npix, int K)
{
    const long long nquads = npix >> 2;
    long long cq = (nquads + K - 1) / K;
    cq = (cq + 1023) & ~1023ll;
    return cq > 0 ? cq : 1024;
}

// Scratch: [256 B: error flag][JointWin x ntiles, padded to 256 B][occupied row blocks per (tile, stream, chunk)][moved-dword lists of the
// windowed workgroups][pair counts]
struct JointScratch {
    size_t win_off, rows_off, list_off, part_off, total;
};
static JointScratch joint_scratch_layout(long long ntiles, int S, int K)
{
    JointScratch L;
    L.win_off = 256;
    L.rows_off = L.win_off + (((size_t)ntiles * sizeof(JointWin) + 255) & ~(size_t)255);
    L.list_off = L.rows_off + (((size_t)ntiles * S * K * sizeof(uint2) + 255) & ~(size_t)255);
    const size_t list_bytes = S == 2 ? (size_t)ntiles * K * JW_LIST_CAP * sizeof(uint2) : 0;
    L.part_off = L.list_off + list_bytes;
    L.total = L.part_off + (size_t)ntiles * S * K * (2 * JH_DWORDS) * sizeof(unsigned int);
    return L;
}

extern "C" size_t lars_joint_scratch_bytes(int64_t ntiles, int64_t npix, uint32_t index_mask)
{
    if (ntiles <= 0 || npix <= 0) return 0;
    const int S = ((index_mask & 1u) ? 1 : 0) + ((index_mask & 6u) ? 1 : 0);
    if (S == 0) return 0;
    const int K = joint_chunks(ntiles, npix, S);
    return joint_scratch_layout(ntiles, S, K).total;
}

extern "C" int lars_d_stats_joint(const lars_fused_args *a, int white_balance, int rgn_variant, double *percentiles,
                                  uint32_t *hist, float *out_pairs, void *scratch, size_t scratch_bytes)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!a || !a->tiles || !a->stats || !scratch || a->ntiles <= 0 || a->npix <= 0)
        return fail(LARS_ERR_INVALID, "lars_d_stats_joint: bad arguments");
    const bool c3 = a->channels == 3 && !(reinterpret_cast<uintptr_t>(a->tiles) & 3);
    const bool c4 = a->channels == 4 && !(reinterpret_cast<uintptr_t>(a->tiles) & 15);
    if (a->dtype != LARS_U8 || !(c3 || c4) || (a->ntiles > 1 && (a->npix & 3)))
        return fail(LARS_ERR_INVALID, "lars_d_stats_joint: uint8 tiles [ntiles][npix][3] on 4-byte or [ntiles][npix][4] on 16-byte boundaries are required");
    const unsigned mask = a->index_mask & LARS_MASK_ALL;
    if (!mask || (a->index_mask & ~LARS_MASK_ALL)) return fail(LARS_ERR_INVALID, "lars_d_stats_joint: index_mask");
    if (a->out_wb || a->out_index[0] || a->out_index[1] || a->out_index[2] || a->out_rgba[0] || a->out_rgba[1] || a->out_rgba[2])
        return fail(LARS_ERR_INVALID, "lars_d_stats_joint: no output planes (lars_d_fused writes those)");
    if (a->ntiles > 65535 || (long long)a->npix >= (1ll << 32))
        return fail(LARS_ERR_INVALID, "lars_d_stats_joint: at most 65535 tiles per call, fewer than 2^32 pixels per tile (the per-cell counts are uint32)");
    hipStream_t s = pick_stream(c, a->stream);
    const unsigned streams = ((mask & 1u) ? 1u : 0u) | ((mask & 6u) ? 2u : 0u);
    const int S = streams == 3u ? 2 : 1;
    const int K = joint_chunks(a->ntiles, a->npix, S);
    const JointScratch L = joint_scratch_layout(a->ntiles, S, K);
    if (scratch_bytes < L.total)
        return fail(LARS_ERR_INVALID, "lars_d_stats_joint: scratch holds %zu bytes, %zu are needed (lars_joint_scratch_bytes, with the "
                                      "tuning in force at the launch)", scratch_bytes, L.total);
    // Windowed tables, one reader per tile chunk (joint_win.hip): two streams, percentile white balance (the windows ARE its clipping),
    // tiles large enough to pay for a window, and no channel histograms wanted -- those would come out clamped to the windows.
    // lars_set_tuning("joint_window", 0) = never, 2 = windows that miss on purpose (exercises the recount), 3 = tiles of any size (tests).
    // 4 = three windows (NIR as well) wherever they fit, before two are tried, tiles of any size; 5 = as 4 with NIR windows that miss (tests).
    const int window = tuning().joint_window;
    const bool windowed = S == 2 && white_balance && !hist && window != 0 && (a->npix >= JW_MIN_PIXELS || window >= 2);

    char *base = static_cast<char *>(scratch);
    unsigned int *error = reinterpret_cast<unsigned int *>(base);
    JointWin *win = reinterpret_cast<JointWin *>(base + L.win_off);
    LARS_HIP_TRY(hipMemsetAsync(error, 0, 256, s));
    const uint8_t *tiles = static_cast<const uint8_t *>(a->tiles);
    if (windowed) {
        joint_predict_launch(tiles, a->ntiles, a->npix, a->channels, win, window == 2 ? 1 : window == 4 ? 2 : window == 5 ? 3 : 0, s);
        LARS_TRY(launch_check("lars_d_stats_joint (predict)"));
    } else {
        LARS_HIP_TRY(hipMemsetAsync(win, 0, (size_t)a->ntiles * sizeof(JointWin), s));        // lars_joint_window_report: nothing windowed
    }
    JointCountParams C;
    memset(&C, 0, sizeof C);
    C.tiles = tiles; C.npix = a->npix; C.ntiles = a->ntiles;
    C.chunk_quads = joint_chunk_quads(a->npix, K);
    C.part = reinterpret_cast<unsigned int *>(base + L.part_off); C.error = error; C.K = K; C.S = S; C.streams = streams;
    C.win = windowed ? win : nullptr; C.pass = 0; C.list = reinterpret_cast<uint2 *>(base + L.list_off);
    C.rows = reinterpret_cast<uint2 *>(base + L.rows_off);
    const long long units = (long long)a->ntiles * K;
    const long long nwg = S == 2 ? ((units + 7) / 8) * 16 : units;
    if (nwg > 0x7FFFFFFFll) return fail(LARS_ERR_INVALID, "lars_d_stats_joint: too many workgroups");
    // joint_depth: 12-byte (RGBA: 16-byte) loads in flight per lane; 6 by default -- 4, 8 and 12 measure the same within noise
    // (profiles/r04_joint_depths.txt), although the bare two-reader pattern gains 5 % from 6 to 12 (profiles/r04_shared_readers_depths.txt)
    const int depth = tuning().joint_depth;
    auto count_full = [&]() {
#define LARS_JOINT(DD, CC) hipLaunchKernelGGL((k_joint_count<DD, CC>), dim3((unsigned)nwg), dim3(JH_THREADS), 0, s, C)
        if (c4) LARS_JOINT(6, 4);
        else if (depth == 4) LARS_JOINT(4, 3);
        else if (depth == 8) LARS_JOINT(8, 3);
        else if (depth == 12) LARS_JOINT(12, 3);
        else LARS_JOINT(6, 3);
#undef LARS_JOINT
    };
    JointFinishParams F;
    memset(&F, 0, sizeof F);
    F.part = C.part; F.rows = C.rows; F.npix = a->npix; F.K = K; F.S = S; F.streams = streams; F.mask = mask;
    F.flags = a->flags & (LARS_F_HIST | LARS_F_SUMSQ); F.wb = white_balance ? 1 : 0; F.rgn_variant = rgn_variant;
    F.stats = a->stats; F.table_out = white_balance ? const_cast<uint8_t *>(a->wb_table) : nullptr;
    F.pcts_out = white_balance ? percentiles : nullptr; F.hist_out = hist; F.out_pairs = out_pairs;
    F.win = C.win; F.pass = 0;

    if (windowed) {
        joint_count_win_launch(C, a->channels, tuning().joint_win_depth, s);
        LARS_TRY(launch_check("lars_d_stats_joint (count, windowed)"));
    }
    count_full();                                          // the tiles without a window (every tile when nothing is windowed)
    LARS_TRY(launch_check("lars_d_stats_joint (count)"));
    hipLaunchKernelGGL(k_joint_finish, dim3((unsigned)S, (unsigned)a->ntiles), dim3(JH_THREADS), 0, s, F);
    LARS_TRY(launch_check("lars_d_stats_joint (finish)"));
    if (windowed) {
        // the tiles whose window missed a percentile: counted again on full tables -- workgroups of all other tiles leave at once
        C.pass = 1; F.pass = 1;
        count_full();
        LARS_TRY(launch_check("lars_d_stats_joint (recount)"));
        hipLaunchKernelGGL(k_joint_finish, dim3((unsigned)S, (unsigned)a->ntiles), dim3(JH_THREADS), 0, s, F);
        LARS_TRY(launch_check("lars_d_stats_joint (finish of the recount)"));
    }
    return LARS_OK;
}

// How the last windowed lars_d_stats_joint on this scratch went (after its stream has finished): tiles counted on windowed
// tables, and tiles among them whose window missed and which were counted again.  Laboratory / test helper.
extern "C" int lars_joint_window_report(const void *scratch, int64_t ntiles, int64_t *windowed, int64_t *recounted)
{
    if (!scratch || ntiles <= 0 || !windowed || !recounted) return fail(LARS_ERR_INVALID, "lars_joint_window_report: bad arguments");
    std::vector<JointWin> w((size_t)ntiles);
    LARS_HIP_TRY(hipMemcpy(w.data(), static_cast<const char *>(scratch) + 256, (size_t)ntiles * sizeof(JointWin), hipMemcpyDeviceToHost));
    int64_t nw = 0, nr = 0;
    for (const JointWin &x : w) { nw += x.mode != 0u; nr += x.flag != 0u; }
    *windowed = nw; *recounted = nr;
    return LARS_OK;
}

// ... and by table form: counts[0] tiles on full tables (two readers), [1] on windowed red and green rows with NIR whole, [2] on three windows.
extern "C" int lars_joint_window_modes(const void *scratch, int64_t ntiles, int64_t counts[3])
{
    if (!scratch || ntiles <= 0 || !counts) return fail(LARS_ERR_INVALID, "lars_joint_window_modes: bad arguments");
    std::vector<JointWin> w((size_t)ntiles);
    LARS_HIP_TRY(hipMemcpy(w.data(), static_cast<const char *>(scratch) + 256, (size_t)ntiles * sizeof(JointWin), hipMemcpyDeviceToHost));
    counts[0] = counts[1] = counts[2] = 0;
    for (const JointWin &x : w) counts[x.mode < 3u ? x.mode : 0u] += 1;
    return LARS_OK;
}
