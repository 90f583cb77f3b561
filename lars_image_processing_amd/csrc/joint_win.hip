// One-read statistics, ONE reader per tile chunk: both byte-pair tables of a chunk in one workgroup's LDS (joint.hip has the
// method; this file the windowed tables and their prediction).
//
// The two-stream launch of k_joint_count puts the (NIR, red) and the (NIR, green) table of a tile chunk into two workgroups on two
// CUs, because 2 x 65536 counters of 16 bits are 256 KiB and a CU has 160: every byte of the tile then travels from the L2 into
// two CUs, and that intake -- not HBM, not the LDS atomics -- set the launch's pace (profiles/r04_joint_hist_counters.txt).
// But the white balance itself says which cells matter: process-images.py:438 maps every sample <= p2 to 0 and every sample >= p98
// to 255, so below a window start lo < p2 and above a window end hi > p98 the samples need not be told apart -- they count as lo
// resp. hi, and every statistic of the white-balanced quotients comes out the same.  With windows on red and green (NIR keeps its
// 256 values; no clamp for it) the tables take (nr + ng) rows of 256 cells: both fit one CU when nr + ng <= 306.  Where they do not,
// NIR -- white-balanced through its own percentiles like the other two -- gets a window as well and a row shrinks to nn cells: three
// windows of up to about 196 values each (77 % of the 8-bit range) still fit (k_joint_count_win<.., NWIN = true>, JointWin mode 2).
//
//   k_joint_predict     per tile: channel histograms of a subsample (1024 segments of 64 pixels spread over the tile) ->
//                       [lo, hi] per channel with a safe and a tight margin for the sampling error -> mode 1 (red and green windows, NIR
//                       whole) if the rows fit, else mode 2 (three windows), else 0 (full tables, two readers)
//   k_joint_count_win   one workgroup per (tile, chunk) of the mode-1 tiles: per pair of pixels 3 v_perm_b32 (n | n' << 16, r | r' << 16,
//                       g | g' << 16), per window 2 packed clamps (v_pk_sub_u16 clamp, v_pk_min_u16) and 1 v_pk_mad_u16 for both dword
//                       indices, 4 LDS atomics.  Publishes the counts in the FULL tables' layout (zeros outside the windows), so that
//                       k_joint_finish reads one format
//   k_joint_finish      derives the exact percentiles from the marginals as ever and checks that their order statistics lie strictly
//                       inside the window (or the window's edge is the range's edge): then lo < p2 and p98 < hi hold and the clamped
//                       counts ARE the tile's statistics.  Otherwise it flags the tile and k_joint_count counts it again on full tables.
//
// Layout in LDS: row x' (red rows 0 .. nr - 1, green rows nr .. nr + ng - 1) takes JW_PITCH = 133 dwords, dword x' * 133 + (n & 127)
// holds the cells n & 127 (low half: both cells, as in joint.hip) and n | 128 (high half).  133 = 128 + 5: the bank of a dword is
// (n + 5 x') mod 32, the full tables' bank pattern (smooth imagery: a 5 x 5 neighbourhood of pairs lands in 25 banks).
#include "joint_device.h"

namespace lars {

struct JointPredictParams {
    const uint8_t *tiles;
    long long npix;
    long long ntiles;
    JointWin *win;
    int test_mode;                            // lars_set_tuning("joint_window", ..): 0 as measured; 1 (setting 2): one-row red and green windows at the
                                              // median, so that every tile misses; 2 (setting 4): three windows wherever they fit, before two are tried;
                                              // 3 (setting 5): as 2 with a one-row NIR window at the median (its check in k_joint_finish misses)
};

#define JP_SEGMENTS 1024                      /* of 16 quads = 64 pixels each: 1 / 256 of a 4096 x 4096 tile */

__device__ inline unsigned int jp_mix(unsigned int x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

template <int CH>
__global__ __launch_bounds__(JH_THREADS) void k_joint_predict(JointPredictParams P)
{
    // red, green and NIR histograms of the sample in 16 copies: lane l adds to copy l & 15 -- at most two lanes of a half-wave on one word,
    // whatever the image (a smooth one puts all 64 lanes into two or three bins); 48 KiB per workgroup, three workgroups per CU (two and
    // four measured the same: the pass is bound by its scattered reads)
    __shared__ unsigned int s_h[3 * 256 * 16];                 // 48 KiB: [channel][bin][copy]
    const int tid = threadIdx.x;
    const long long tile = blockIdx.x;
    for (int i = tid; i < 3 * 256 * 16 / 4; i += JH_THREADS) reinterpret_cast<uint4 *>(s_h)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const long long nquads = P.npix >> 2;
    const uint8_t *base = P.tiles + tile * P.npix * CH;
    const unsigned int copy = (unsigned)(tid & 15);
    auto add = [&](unsigned int ch, unsigned int v) {
        __hip_atomic_fetch_add(&s_h[(((ch << 8) | v) << 4) | copy], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto count_quad = [&](unsigned int w0, unsigned int w1, unsigned int w2) {
        // r0 g0 n0 r1 | g1 n1 r2 g2 | n2 r3 g3 n3
        add(0u, w0 & 255u); add(1u, (w0 >> 8) & 255u); add(0u, w0 >> 24); add(1u, w1 & 255u);
        add(0u, (w1 >> 16) & 255u); add(1u, w1 >> 24); add(0u, (w2 >> 8) & 255u); add(1u, (w2 >> 16) & 255u);
        add(2u, (w0 >> 16) & 255u); add(2u, (w1 >> 8) & 255u); add(2u, w2 & 255u); add(2u, w2 >> 24);
    };
    auto load_quad = [&](long long q, unsigned int &w0, unsigned int &w1, unsigned int &w2) {
        if constexpr (CH == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(base + q * 16);
            w0 = __builtin_amdgcn_perm(v.y, v.x, 0x04020100u); w1 = __builtin_amdgcn_perm(v.z, v.y, 0x05040201u);
            w2 = __builtin_amdgcn_perm(v.w, v.z, 0x06050402u);
        } else {
            const unsigned int *p = reinterpret_cast<const unsigned int *>(base + q * 12);
            w0 = p[0]; w1 = p[1]; w2 = p[2];
        }
    };
    long long sampled;                                       // pixels in the histograms
    if (nquads >= (long long)JP_SEGMENTS * 128) {
        // segment i (16 quads = 64 pixels = 192 contiguous bytes; a wave reads four per load) starts somewhere inside its own stretch of
        // nquads / 1024 quads; a lane's loads all go out before the first is counted (latency once, not 16 times).  The pass is bound by
        // its scattered reads (0.4 % of the batch, but in pieces of two or three 128-byte lines: 95 us per 1024 tiles with segments of
        // 128 pixels, what 550 MB cost at HBM rate); what a smooth image needs is many segments, not long ones.
        const long long stretch = nquads / JP_SEGMENTS;
        const unsigned int span = (unsigned int)(stretch - 15);            // npix < 2^32: a segment starts at stretch * i + [0, span)
        const int wave = tid >> 6, lane = tid & 63;
        constexpr int NJ = JP_SEGMENTS / 64;                               // 16 loads per lane
        unsigned int w[NJ][3];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned int i = (unsigned)((j * 16 + wave) * 4 + (lane >> 4));
            const long long q = (long long)i * stretch + (long long)__umulhi(jp_mix(i + (unsigned)tile * 0x9E3779B9u), span) + (lane & 15);
            load_quad(q, w[j][0], w[j][1], w[j][2]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) count_quad(w[j][0], w[j][1], w[j][2]);
        sampled = (long long)JP_SEGMENTS * 64;
    } else {
        for (long long q = tid; q < nquads; q += JH_THREADS) {
            unsigned int w0, w1, w2;
            load_quad(q, w0, w1, w2);
            count_quad(w0, w1, w2);
        }
        sampled = nquads * 4;
    }
    __syncthreads();
    // totals of the 16 copies, their running sums (wave scans + one barrier), and the order statistics that bound the windows
    __shared__ unsigned int s_wsum[3][4];
    __shared__ unsigned int s_lo[2][3], s_hi[2][3];          // [margin][channel]
    unsigned int tot = 0, inc = 0;
    if (tid < 768) {
#pragma unroll 8
        for (int k = 0; k < 16; ++k) tot += s_h[tid * 16 + ((k + tid) & 15)];    // rotated: fewer lanes per bank
        inc = tot;
        const int lane = tid & 63;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wsum[tid >> 8][(tid >> 6) & 3] = inc;
    }
    __syncthreads();
    if (tid < 768 && sampled > 0) {
        // The window of a channel: from the sample's order statistics at 0.5 % and 99.5 %, one more value on either side.  The margin is
        // for imagery, not for independent samples: the 64 pixels of a segment of a smooth image are nearly one observation, so the
        // sample is worth its 1024 segments -- a value that truly holds 2 % of the tile below it shows fewer than 0.5 % of the sample
        // there about once in 10^5 tiles even then.  (0.4 % of margin, enough for independent pixels, missed on 19 % of the tiles of
        // tools/jointbench.py's smooth content: profiles/r05_joint_window_first.txt.)
        // A second, tight set of windows (1.5 % and 98.5 %) is for the tiles whose safe windows do not fit: a photograph with 1 % of
        // overexposed pixels has its 99.5 % order statistic at 255 and its p98 a hundred values below.  A tight window misses more
        // often, and a miss costs a second count -- but not fitting costs the second READER for certain.
        const int ch = tid >> 8, bin = tid & 255;
        unsigned int before = inc - tot;
        for (int w = 0; w < ((tid >> 6) & 3); ++w) before += s_wsum[ch][w];
        const bool wrong = ch < 2 ? P.test_mode == 1 : P.test_mode == 3;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            double dq = m ? 0.005 : 0.015;
            if (sampled == nquads * 4) dq = 0.0;              // everything was counted (but the tile's last npix % 4 pixels)
            double ql = 0.02 - dq, qh = 0.98 + dq;
            if (wrong) ql = qh = 0.5;
            const long long rl = (long long)floor((double)(sampled - 1) * ql), rh = (long long)ceil((double)(sampled - 1) * qh);
            const int margin = wrong ? 0 : 1;
            if (tot && rl >= (long long)before && rl < (long long)before + tot) s_lo[m][ch] = (unsigned)(bin - margin < 0 ? 0 : bin - margin);
            if (tot && rh >= (long long)before && rh < (long long)before + tot) s_hi[m][ch] = (unsigned)(bin + margin > 255 ? 255 : bin + margin);
        }
    }
    __syncthreads();
    if (tid == 0) {
        // Which tables the tile is counted on: safe margins before tight ones (a miss costs a second count: 2.2 x), and at either margin
        // windows on red and green with NIR whole before windows on all three channels (that kernel is 1-4 % slower); else full tables
        // and two readers.
        JointWin w;
        w.flag = 0u;
        w.lo_r = w.lo_g = w.lo_n = 0; w.nr = w.ng = w.nn = 256; w.pitch = (unsigned short)JW_PITCH; w.half = 128; w.mode = 0u;
        auto two = [&](int m) {
            const unsigned int nr = s_hi[m][0] - s_lo[m][0] + 1u, ng = s_hi[m][1] - s_lo[m][1] + 1u;
            if (nr + ng > (unsigned)JW_MAX_ROWS) return false;
            w.lo_r = (unsigned short)s_lo[m][0]; w.nr = (unsigned short)nr; w.lo_g = (unsigned short)s_lo[m][1]; w.ng = (unsigned short)ng;
            w.lo_n = 0; w.nn = 256; w.pitch = (unsigned short)JW_PITCH; w.half = 128; w.mode = 1u;
            return true;
        };
        auto three = [&](int m) {
            const unsigned int nr = s_hi[m][0] - s_lo[m][0] + 1u, ng = s_hi[m][1] - s_lo[m][1] + 1u, nn = s_hi[m][2] - s_lo[m][2] + 1u;
            const unsigned int half = (nn + 1u) >> 1, pitch = jw_pitch_for(half);
            if ((nr + ng) * pitch + 3u > (unsigned)JW_TAB_DWORDS) return false;
            w.lo_r = (unsigned short)s_lo[m][0]; w.nr = (unsigned short)nr; w.lo_g = (unsigned short)s_lo[m][1]; w.ng = (unsigned short)ng;
            w.lo_n = (unsigned short)s_lo[m][2]; w.nn = (unsigned short)nn; w.pitch = (unsigned short)pitch; w.half = (unsigned short)half; w.mode = 2u;
            return true;
        };
        if (sampled > 0) {
            if (P.test_mode >= 2) { three(0) || three(1) || two(0) || two(1); }
            else { two(0) || three(0) || two(1) || three(1); }
        }
        P.win[tile] = w;
    }
}

// ---- packed 16-bit arithmetic the compiler is not left to choose ------------------------------------------------------------
__device__ inline unsigned int jw_sub_sat(unsigned int a, unsigned int b)        // per half: max(a - b, 0)
{
    unsigned int r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline unsigned int jw_min(unsigned int a, unsigned int b)
{
    unsigned int r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline unsigned int jw_mad(unsigned int a, unsigned int b, unsigned int c)   // per half: a * b + c
{
    unsigned int r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// 1 | h0 << 16 for u = h0 << 7 | h1 << 23: the product's bit 32 falls off
__device__ inline unsigned int jw_val0(unsigned int u)
{
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(r) : "v"(u), "s"(512u));
    return r;
}

// byte address of the dword whose index is the low (HALF = 0) or high (HALF = 1) 16 bits of d: one SDWA shift, the half selected by the operand
template <int HALF>
__device__ inline unsigned int jw_addr(unsigned int d, unsigned int two)
{
    unsigned int r;
    if (HALF) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(two), "v"(d));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(two), "v"(d));
    return r;
}

// NWIN: NIR has a window as well (JointWin mode 2): rows of win.pitch dwords, cell n' = clamp(n - lo_n) in dword n' mod half, half n' >= half.
// Three more packed instructions per pixel pair than the NIR-whole form, whose rows are 128 + 5 dwords and whose half is bit 7 of n.
template <int DEPTH, int CH = 3, bool NWIN = false>
__global__ __launch_bounds__(JH_THREADS, 4) void k_joint_count_win(JointCountParams P)
{
    // a period is a whole number of ring turns and adds at most 65535 - 4095 = 61440 pixels (15 steps) to any one dword
    constexpr int PERIOD = (JW_PERIOD_STEPS % DEPTH == 0) ? JW_PERIOD_STEPS : ((12 % DEPTH == 0) ? 12 : DEPTH);
    static_assert(PERIOD % DEPTH == 0 && PERIOD <= JW_PERIOD_STEPS, "a period is a whole number of ring turns, at most 15 steps");
    __shared__ __attribute__((aligned(16))) unsigned int s_tab[JW_TAB_DWORDS];      // 159 KiB
    __shared__ unsigned int s_nlist;

    const int tid = threadIdx.x;
    const long long unit = blockIdx.x;
    const long long tile = unit / P.K;
    const int chunk = (int)(unit - tile * P.K);
    if (tile >= P.ntiles) return;
    const JointWin win = P.win[tile];
    if (win.mode != (NWIN ? 2u : 1u)) return;                                      // another kernel's tile
    const unsigned int nr = win.nr, ng = win.ng, lo_r = win.lo_r, lo_g = win.lo_g;
    const unsigned int lo_n = NWIN ? win.lo_n : 0u, nn = NWIN ? win.nn : 256u;
    const unsigned int pitch = NWIN ? win.pitch : JW_PITCH, half = NWIN ? win.half : 128u;
    const unsigned int rows = nr + ng;
    const int ntab4 = (int)((rows * pitch + 3u) >> 2);                             // uint4s in use

    uint4 *tab4 = reinterpret_cast<uint4 *>(s_tab);
    for (int i = tid; i < ntab4; i += JH_THREADS) tab4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    char *tab = reinterpret_cast<char *>(s_tab);

    const long long nquads_tile = P.npix >> 2;
    const long long q_begin = (long long)chunk * P.chunk_quads;
    long long q_end = chunk == P.K - 1 ? nquads_tile : q_begin + P.chunk_quads;
    if (q_end > nquads_tile) q_end = nquads_tile;
    const long long nq = q_end > q_begin ? q_end - q_begin : 0;
    const uint8_t *tile_base = P.tiles + tile * P.npix * CH;

    // Which table a lane's FIRST and which its SECOND add of a pixel goes to depends on the lane's parity: even lanes red then green, odd
    // lanes green then red.  One LDS instruction then spreads its 64 lanes over both tables -- on smooth imagery, where the pixels of a
    // wave fall into a 5 x 5 neighbourhood of cells per table, that halves the lanes per cell and bank.  It costs nothing but registers:
    // the byte selectors of v_perm_b32 and the windows' constants are per-lane values instead of scalars.  Both halves of a dword alike.
    const bool odd = (tid & 1) != 0;
    const unsigned int lo_r2 = lo_r * 0x10001u, lo_g2 = lo_g * 0x10001u;
    const unsigned int nr1_2 = (nr - 1u) * 0x10001u, ng1_2 = (ng - 1u) * 0x10001u;
    const unsigned int base_g2 = (nr * pitch) * 0x10001u;                              // first dword of the green rows: < 2^16
    const unsigned int lo_f = odd ? lo_g2 : lo_r2, lo_s = odd ? lo_r2 : lo_g2;
    const unsigned int n1_f = odd ? ng1_2 : nr1_2, n1_s = odd ? nr1_2 : ng1_2;
    const unsigned int base_f = odd ? base_g2 : 0u, base_s = odd ? 0u : base_g2;
    const unsigned int pitch2 = pitch * 0x10001u;
    unsigned int lo_n2 = lo_n * 0x10001u, nn1_2 = (nn - 1u) * 0x10001u;                // NWIN: the NIR clamp,
    unsigned int halfm1_2 = (half - 1u) * 0x10001u;                                    // ... and n' -> (dword, half)
    const unsigned int neghalf_2 = ((0x10000u - half) & 0xFFFFu) * 0x10001u;
    unsigned int one2 = 0x00010001u;
    // operands of the packed instructions below that take no scalar register: keep them in vector registers, not copied in per use
    asm volatile("" : "+v"(lo_n2), "+v"(nn1_2), "+v"(halfm1_2), "+v"(one2));

    // bytes of a quad r0 g0 n0 r1 | g1 n1 r2 g2 | n2 r3 g3 n3; pixels 0, 1 from perm(w1, w0), pixels 2, 3 from perm(w2, w1)
    const unsigned int seln01 = 0x0c050c02u, selr01 = 0x0c030c00u, selg01 = 0x0c040c01u;
    const unsigned int seln23 = 0x0c070c04u, selr23 = 0x0c050c02u, selg23 = 0x0c060c03u;
    const unsigned int self01 = odd ? selg01 : selr01, sels01 = odd ? selr01 : selg01;
    const unsigned int self23 = odd ? selg23 : selr23, sels23 = odd ? selr23 : selg23;

    unsigned int two = 2u;
    asm volatile("" : "+v"(two));                                                  // a VGPR that holds 2 (SDWA takes no constants)
    // Two pixels at a time, one in each 16-bit half; xf / xs: the samples paired with NIR in the lane's first and second table.
    // mul = 1 but in runs of equal quads, where the lane that starts a run adds the run's length.
    auto count_pair = [&](unsigned int nn_, unsigned int xf, unsigned int xs, unsigned int mul) {
        unsigned int t, v0, v1;
        if constexpr (NWIN) {
            const unsigned int nc = jw_min(jw_sub_sat(nn_, lo_n2), nn1_2);           // n' = clamp(n - lo_n, 0, nn - 1)
            const unsigned int h = jw_min(jw_sub_sat(nc, halfm1_2), one2);          // n' >= half, per pixel
            t = jw_mad(h, neghalf_2, nc);                                           // n' - h * half (mod 2^16)
            v0 = (h << 16) | 1u;                                                    // 1 | h << 16 of the low pixel
            v1 = (h & 0x10000u) | 1u;                                               // of the high pixel
        } else {
            t = nn_ & 0x007F007Fu;
            const unsigned int u = nn_ & 0x00800080u;
            v0 = jw_val0(u);                                                        // 1 | h << 16 of the low pixel
            v1 = (u >> 7) | 1u;                                                     // of the high pixel (bit 0 is set either way)
        }
        if (mul != 1u) { v0 *= mul; v1 = (v1 & 0x10001u) * mul; }
        const unsigned int df = jw_mad(jw_min(jw_sub_sat(xf, lo_f), n1_f), pitch2, t + base_f);
        const unsigned int ds = jw_mad(jw_min(jw_sub_sat(xs, lo_s), n1_s), pitch2, t + base_s);
        jh_add(jw_addr<0>(df, two), v0, tab);
        jh_add(jw_addr<1>(df, two), v1, tab);
        jh_add(jw_addr<0>(ds, two), v0, tab);
        jh_add(jw_addr<1>(ds, two), v1, tab);
    };
    auto do_quad = [&](unsigned int w0, unsigned int w1, unsigned int w2) {
        // runs of equal pixels: the lanes that start a run add its whole count, the others nothing (joint_device.h)
        if (jh_mostly_runs(w0)) {
            const bool head = ((jh_prev_lane(w0) ^ w0) | (jh_prev_lane(w1) ^ w1) | (jh_prev_lane(w2) ^ w2)) != 0u;   // every lane takes part in the three DPP moves
            const unsigned int n = jh_run_length(head, tid & 63);
            if (head) {
                count_pair(__builtin_amdgcn_perm(w1, w0, seln01), __builtin_amdgcn_perm(w1, w0, self01), __builtin_amdgcn_perm(w1, w0, sels01), n);
                count_pair(__builtin_amdgcn_perm(w2, w1, seln23), __builtin_amdgcn_perm(w2, w1, self23), __builtin_amdgcn_perm(w2, w1, sels23), n);
            }
            return;
        }
        count_pair(__builtin_amdgcn_perm(w1, w0, seln01), __builtin_amdgcn_perm(w1, w0, self01), __builtin_amdgcn_perm(w1, w0, sels01), 1u);
        count_pair(__builtin_amdgcn_perm(w2, w1, seln23), __builtin_amdgcn_perm(w2, w1, self23), __builtin_amdgcn_perm(w2, w1, sels23), 1u);
    };

    // tail pixels of the tile (npix % 4): its last chunk, before the first period
    if (chunk == P.K - 1 && tid < (int)(P.npix & 3)) {
        const uint8_t *p = tile_base + (nquads_tile * 4 + tid) * CH;
        unsigned int n = p[2] > lo_n ? p[2] - lo_n : 0u;
        n = n < nn - 1u ? n : nn - 1u;
        unsigned int r = p[0] > lo_r ? p[0] - lo_r : 0u, g = p[1] > lo_g ? p[1] - lo_g : 0u;
        r = r < nr - 1u ? r : nr - 1u;
        g = g < ng - 1u ? g : ng - 1u;
        const unsigned int h = n >= half ? 1u : 0u, t = n - h * half;
        const unsigned int v = (h << 16) | 1u;
        jh_add((r * pitch + t) << 2, v, tab);
        jh_add(((nr + g) * pitch + t) << 2, v, tab);
    }

    // A scan: every dword whose sum (low half) has reached 4096 moves onto the workgroup's list (in global memory: the LDS is the
    // tables').  Between the two barriers nobody adds.
    uint2 *list = P.list + unit * JW_LIST_CAP;
    auto scan = [&]() {
        __syncthreads();
#pragma unroll 1
        for (int idx = tid; idx < ntab4; idx += JH_THREADS) {
            const uint4 v = tab4[idx];
            if ((v.x | v.y | v.z | v.w) & JW_PROMOTE_MASK) {
                const unsigned int c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (c[k] & JW_PROMOTE_MASK) {
                        const unsigned int slot = atomicAdd(&s_nlist, 1u);
                        if (slot < JW_LIST_CAP) list[slot] = make_uint2((unsigned)(idx * 4 + k), c[k]);
                        else atomicExch(P.error, 1u);
                        s_tab[idx * 4 + k] = 0u;
                    }
                }
            }
        }
        __syncthreads();
    };

    const long long nfull = nq >> 10;
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
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const long long step = it + k;
            if (step < nfull || (step == nfull && tid < rem)) count(w[k]);
        }
    }
    __syncthreads();

    // Publish in the full tables' layout (joint.hip): stream s, dword D = x << 7 | (m & 127) with m = (n + 5 x) & 255 holds the cells
    // (D, 0), (D, 1) = the counts of n0 = (D & 127) - 5 x and of n0 ^ 128 -- with NIR whole the two halves of ONE windowed dword, with a
    // NIR window two cells looked up one by one (zero outside the window).  Rows outside the red / green window are zero and, but for those
    // that share a block of 8 rows with the window, not even written: k_joint_finish skips them.
    auto full_cells = [&](unsigned int row, unsigned int n0) {      // (table row, n of the dword's first cell) -> (count of n0, count of n0 ^ 128)
        if constexpr (NWIN) {
            auto cell = [&](unsigned int n) {
                const unsigned int np = n - lo_n;                   // wraps below the window
                if (np >= nn) return 0u;
                const unsigned int hh = np >= half ? 1u : 0u;
                const unsigned int v = s_tab[row * pitch + np - hh * half];
                return hh ? v >> 16 : (v & 0xFFFFu) - (v >> 16);
            };
            return make_uint2(cell(n0), cell(n0 ^ 128u));
        } else {
            const unsigned int v = s_tab[row * JW_PITCH + (n0 & 127u)];
            const unsigned int hi = v >> 16, lo = (v & 0xFFFFu) - hi;  // cells n & 127 (bit 7 clear) and n | 128
            return (n0 & 128u) ? make_uint2(hi, lo) : make_uint2(lo, hi);
        }
    };
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        unsigned int *out = P.part + ((tile * 2 + s) * P.K + chunk) * (long long)(2 * JH_DWORDS);
        const unsigned int lo = s ? lo_g : lo_r, nx = s ? ng : nr, row0 = s ? nr : 0u;
        // only the blocks of 8 rows that touch the window: k_joint_finish walks exactly these
        const int d_begin = (int)(lo >> 3) << 10, d_end = (int)(((lo + nx - 1u) >> 3) + 1u) << 10;
        if (tid == 0) P.rows[(tile * 2 + s) * P.K + chunk] = make_uint2(lo >> 3, ((lo + nx - 1u) >> 3) + 1u);
        for (int D = d_begin + tid; D < d_end; D += JH_THREADS) {
            const unsigned int x = (unsigned)D >> 7;
            uint2 c = make_uint2(0u, 0u);
            if (x >= lo && x < lo + nx) c = full_cells(row0 + x - lo, (((unsigned)D & 127u) - JH_K * x) & 255u);
            *reinterpret_cast<uint2 *>(out + 2 * (long long)D) = c;
        }
    }
    const unsigned int nlist = s_nlist < JW_LIST_CAP ? s_nlist : JW_LIST_CAP;
    if (nlist) {
        // the stores above have to be in the L2 before the adds to the same words: the barrier's own workgroup-scope release (stores
        // acknowledged) orders them, and the adds are workgroup-scope too -- every access stays in this XCD's L2.  (A device-scope
        // __threadfence here writes the XCD's L2 back for every workgroup that has a list: + 40 % on the launch of any batch whose
        // chunks hold a cell of 16384 pixels: profiles/r05_joint_publish_fence.txt.)
        __syncthreads();
        for (unsigned int e = tid; e < nlist; e += JH_THREADS) {
            const unsigned long long mv = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(&list[e]));   // past this CU's L1
            const uint2 m = make_uint2((unsigned int)mv, (unsigned int)(mv >> 32));
            const unsigned int row = m.x / pitch, nl = m.x - row * pitch;           // the dword's first cell: n' = nl (NIR whole: n & 127)
            const int s = row >= nr ? 1 : 0;
            const unsigned int x = s ? lo_g + (row - nr) : lo_r + row;
            unsigned int *out = P.part + ((tile * 2 + s) * P.K + chunk) * (long long)(2 * JH_DWORDS);
            const unsigned int hi = m.y >> 16, lo = (m.y & 0xFFFFu) - hi;
            // cells n = lo_n + nl (count lo) and n = lo_n + nl + half (count hi), each at (D, h) of the full layout
            const unsigned int m0 = jh_m(lo_n + nl, x), m1 = jh_m(lo_n + nl + half, x);
            if (lo) jh_publish_add(&out[2 * (long long)((x << 7) | (m0 & 127u)) + (m0 >> 7)], lo);
            if (hi) jh_publish_add(&out[2 * (long long)((x << 7) | (m1 & 127u)) + (m1 >> 7)], hi);
        }
    }
}

// ---- launchers (called by lars_d_stats_joint, joint.hip) --------------------------------------------------------------------
void joint_predict_launch(const uint8_t *tiles, long long ntiles, long long npix, int channels, JointWin *win, int test_mode, hipStream_t s)
{
    JointPredictParams P;
    P.tiles = tiles; P.npix = npix; P.ntiles = ntiles; P.win = win; P.test_mode = test_mode;
    if (channels == 4) hipLaunchKernelGGL((k_joint_predict<4>), dim3((unsigned)ntiles), dim3(JH_THREADS), 0, s, P);
    else hipLaunchKernelGGL((k_joint_predict<3>), dim3((unsigned)ntiles), dim3(JH_THREADS), 0, s, P);
}
void joint_count_win_launch(const JointCountParams &C, int channels, int depth, hipStream_t s)
{
    const long long units = C.ntiles * C.K;
#define LARS_JOINT_WIN(DD, CC) hipLaunchKernelGGL((k_joint_count_win<DD, CC, false>), dim3((unsigned)units), dim3(JH_THREADS), 0, s, C)
    if (channels == 4) LARS_JOINT_WIN(5, 4);
    else if (depth == 4) LARS_JOINT_WIN(4, 3);
    else if (depth == 5) LARS_JOINT_WIN(5, 3);
    else if (depth == 6) LARS_JOINT_WIN(6, 3);
    else if (depth == 12) LARS_JOINT_WIN(12, 3);
    else LARS_JOINT_WIN(15, 3);
#undef LARS_JOINT_WIN
    // the tiles with a NIR window as well (the depth of the load ring is noise: one build per pixel format)
    if (channels == 4) hipLaunchKernelGGL((k_joint_count_win<5, 4, true>), dim3((unsigned)units), dim3(JH_THREADS), 0, s, C);
    else hipLaunchKernelGGL((k_joint_count_win<15, 3, true>), dim3((unsigned)units), dim3(JH_THREADS), 0, s, C);
}

}  // namespace lars
