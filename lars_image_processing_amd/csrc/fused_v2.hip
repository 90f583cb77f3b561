// The per-pixel statistics kernels ("classic" route), written for the CDNA4 issue budget.
//
// WHAT STILL DEPENDS ON THIS FILE (round 4).  Since round 3 the statistics of uint8 RGNir / RGBA batches come from the one-read
// route (joint.hip) by default; these kernels serve
//   * statistics-only passes the one-read route does not take: uint16 tiles, channel counts other than 3 / 4, unaligned tiles,
//     passes over tables the caller supplied (process(recompute_tables=False)), and route="auto" on content where it measures faster;
//   * lars_d_stats_medians / lars_d_quotient_select_hist (the first select pass lives inside k_fused_v2<..., SEL>) and with them
//     TileBatch.global_medians, the exact median over all tiles of all ranks;
//   * k_chan_hist_u8c3_v2: the channel-histogram pre-pass of every launch that writes planes from tables (the bench headline);
//   * the other side of every route-equality test (tests/test_gpu_joint.py, tests/fuzz_routes.py) and of bench.py's self-check.
//
// The stats-only configurations move 3 bytes per pixel, so at HBM rate a CU has
// only ~35 vector-instruction slots per pixel.  What this file does about it:
//
//   * white-balance table as ONE packed dword per sample value
//     {R'[v], G'[v], N'[v], 0}, replicated 64x in LDS so that lane l only ever
//     touches bank l%32 (conflict-free ds_read_b32); the byte address
//     (v << 8 | lane*4) is built by a single v_perm_b32 and the wanted channel
//     is converted with v_cvt_f32_ubyte{0,1,2}: 2 VALU + 1 LDS per sample.
//   * IEEE-exact float32 quotient as rcp + mul + 2 fma (no div_scale / div_fmas
//     / div_fixup); tests/test_gpu_parity.py proves it bit-identical to the
//     correctly rounded quotient for every operand pair the domain allows.
//   * coverage counters live in scalar registers: v_cmp + s_bcnt1 + s_add.
//   * NDWI statistics are derived from GNDVI's at flush time (NDWI == -GNDVI
//     exactly): sum negates, extrema swap, only the coverage count (g < 0)
//     is kept per pixel.
//   * sums stay exact: float32 index values of uint8 tiles are multiples of
//     2^-32, double adds of them are exact, blocks publish int64 fixed point.
//   * the channel histogram pre-pass uses 32 lane-private copies of each
//     256-bin histogram (96 KiB of LDS) so ds_add_u32 never conflicts.
//
// Reference semantics: see fused.hip.  Built with -ffp-contract=off.
#include <string.h>

#include <type_traits>

#include "v2_device.h"

namespace lars {

// Exhaustive self-check kernel: counts operand pairs where exact_quot differs
// from the compiler's correctly rounded division.  num in [-den, den].
__global__ __launch_bounds__(256) void k_quot_check(unsigned int max_den, unsigned long long *mismatches,
                                                    unsigned int *first_bad)
{
    unsigned long long bad = 0;
    for (unsigned int den = blockIdx.x + 1; den <= max_den; den += gridDim.x) {
        const float fd = (float)den;
        for (long long num = -(long long)den + threadIdx.x; num <= (long long)den; num += 256) {
            const float fn = (float)num;
            const float want = fn / fd;
            const float got = exact_quot(fn, fd);
            if (__builtin_bit_cast(unsigned int, want) != __builtin_bit_cast(unsigned int, got)) {
                if (!bad) { first_bad[0] = den; first_bad[1] = (unsigned int)(int)num; }
                ++bad;
            }
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------------------
// channel histograms, conflict-free
// ---------------------------------------------------------------------------
// LDS: [3 channels][256 bins][32 copies] u32 = 96 KiB; copy = lane % 32.
// CH = 4: RGBA tiles, one 16-byte load per lane repacked into the three dwords of an RGB quad (alpha is not counted).
template <int CH>
__global__ __launch_bounds__(1024) void k_chan_hist_u8c3_v2(const uint8_t *__restrict__ tiles, long long npix,
                                                            unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_h[3 * 256 * 32];             // 96 KiB
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * 256 * 32; i += 1024) s_h[i] = 0;
    __syncthreads();

    const long long tile = blockIdx.y;
    const uint8_t *base = tiles + tile * npix * CH;
    const long long nquads = npix >> 2;
    const unsigned int lane_off = (tid & 31) << 2;         // byte offset of this lane's copy
    char *hb = reinterpret_cast<char *>(s_h);
#define HADD(word, shift, ch)                                                                          \
    atomicAdd(reinterpret_cast<unsigned int *>(hb + (ch) * 32768 + ((((word) >> (shift)) & 0xFFu) << 7) + lane_off), 1u)
    // software pipeline: four 12-byte buffer loads in flight per lane (48 KiB per CU at 16 waves)
    const long long step = (long long)gridDim.x * 1024;
    const long long q0 = (long long)blockIdx.x * 1024 + tid;
    const long long niter = (nquads + step - 1) / step;            // same for every lane of the grid
#define HQUAD(a0, a1, a2)                                                                              \
    HADD(a0, 0, 0); HADD(a0, 8, 1); HADD(a0, 16, 2); HADD(a0, 24, 0);                                 \
    HADD(a1, 0, 1); HADD(a1, 8, 2); HADD(a1, 16, 0); HADD(a1, 24, 1);                                 \
    HADD(a2, 0, 2); HADD(a2, 8, 0); HADD(a2, 16, 1); HADD(a2, 24, 2);
    if (niter > 0 && CH == 4) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(nquads * 16), 0x00020000);
        const unsigned int voff = (unsigned int)q0 * 16u;
        const unsigned int step_b = (unsigned int)step * 16u;
        typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
        u32x4h w[4];
#define HQUAD4(p)                                                                                       \
        {                                                                                               \
            const unsigned int a0 = __builtin_amdgcn_perm((p).y, (p).x, 0x04020100u);                  \
            const unsigned int a1 = __builtin_amdgcn_perm((p).z, (p).y, 0x05040201u);                  \
            const unsigned int a2 = __builtin_amdgcn_perm((p).w, (p).z, 0x06050402u);                  \
            HQUAD(a0, a1, a2)                                                                           \
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (unsigned)k * step_b, 0);
        long long it = 0;
        unsigned int soff = 4u * step_b;
        for (; it + 4 <= niter - 1; it += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                HQUAD4(w[k])
                __builtin_amdgcn_sched_barrier(0);
                w[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)k * step_b, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            soff += 4u * step_b;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (it + k < niter && q0 + (it + k) * step < nquads) { HQUAD4(w[k]) }
        }
#undef HQUAD4
    } else if (niter > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(nquads * 12), 0x00020000);
        const unsigned int voff = (unsigned int)q0 * 12u;
        const unsigned int step_b = (unsigned int)step * 12u;
        u32x3 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, (unsigned)k * step_b, 0);
        long long it = 0;
        unsigned int soff = 4u * step_b;
        for (; it + 4 <= niter - 1; it += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                HQUAD(w[k].x, w[k].y, w[k].z)
                __builtin_amdgcn_sched_barrier(0);
                w[k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff + (unsigned)k * step_b, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            soff += 4u * step_b;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (it + k < niter && q0 + (it + k) * step < nquads) { HQUAD(w[k].x, w[k].y, w[k].z) }
        }
    }
#undef HQUAD
    if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
        const uint8_t *p = base + (nquads * 4 + tid) * CH;
        HADD((unsigned)p[0], 0, 0); HADD((unsigned)p[1], 0, 1); HADD((unsigned)p[2], 0, 2);
    }
#undef HADD
    __syncthreads();
    // fold the 32 copies: 768 bins, one per thread (rotated start keeps banks apart)
    if (tid < 768) {
        const unsigned int *row = s_h + tid * 32;
        unsigned int v = 0;
        for (int j = 0; j < 32; ++j) v += row[(j + tid) & 31];
        if (v) atomicAdd(&hist[tile * 768 + tid], v);
    }
}

// ---------------------------------------------------------------------------
// fused kernel, generation 2 (uint8, 3 channels, 4-byte aligned tiles)
// ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ inline void store4(float *dst, float a, float b, float c, float d)
{
    f32x4 v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst));
    else *reinterpret_cast<f32x4 *>(dst) = v;
}

struct WaveAcc {
    float mn, mx;
    double sum, sumsq;
};

// LDS layout (dynamic): [WB table 64 KiB][hist 3*50*32 u32][edges 51 f32][reduce scratch]
#define V2_HIST_COPIES 16    // copies of each index histogram, copy = lane % 16 (two blocks per CU still fit)
#define V2_HIST_SHIFT 6      // log2(V2_HIST_COPIES * 4): bytes between consecutive bins
#define V2_HIST_ROWS 51      // bins 0..50; 50 (x == 1.0, the closed right edge) folds into 49 at the flush
#define V2_HIST_WORDS (3 * V2_HIST_ROWS * V2_HIST_COPIES)
#define V2_WIN_ROW (64 + SELQ_WIN_SLOTS + 64)   // one stream's row: below words, slots, above words = 2048

// Coverage counters live in scalar registers.  The compare mask goes through VCC inside ONE asm
// statement (v_cmp -> s_bcnt1), so it never occupies an allocatable SGPR pair: with the
// ballot/popcount form hipcc kept a dozen 64-bit masks alive per quad, ran out of SGPRs and spilled
// them with v_writelane_b32 (4.5 extra VALU instructions per pixel in the three-index kernel).
// Inactive lanes contribute zero bits; scalar instructions ignore EXEC.
// LARS_COUNT_MODE (build-time, tools/kbench A/B): 0 scalar counters everywhere, 1 per-lane vector
// counters everywhere (v_cmp + v_addc, no SALU), 2 NDVI on the vector pipe and GNDVI/NDWI on the scalar pipe.
// (Also tried and dropped: per-lane float counters fed by a saturating packed fma, sat((x - thr) * 2^40) --
// exact, no compare / SALU / VCC, but the same VALU time and 2 % slower at 8 waves per SIMD.)
#ifndef LARS_COUNT_MODE
#define LARS_COUNT_MODE 0
#endif
__device__ inline void vcount_gt(unsigned int &lane_counter, float x, float thr) { lane_counter += (x > thr) ? 1u : 0u; }
__device__ inline void vcount_lt(unsigned int &lane_counter, float x, float thr) { lane_counter += (x < thr) ? 1u : 0u; }
__device__ inline unsigned int wave_sum_u32(unsigned int v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ inline void count_gt(unsigned int &counter, float x, float thr)      // counter += #lanes(x > thr)
{
    unsigned int c;
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_bcnt1_i32_b64 %0, vcc" : "=s"(c) : "s"(thr), "v"(x) : "vcc", "scc");
    counter += c;
}
__device__ inline void count_lt(unsigned int &counter, float x, float thr)      // counter += #lanes(x < thr)
{
    unsigned int c;
    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_bcnt1_i32_b64 %0, vcc" : "=s"(c) : "s"(thr), "v"(x) : "vcc", "scc");
    counter += c;
}

// four samples at once: one asm statement per counter and quad (hipcc pads every asm statement that
// touches VCC with an s_nop, and every scalar instruction costs a slot of the CU's single scalar pipe)
__device__ inline void count4_gt(unsigned int &counter, const float (&x)[4], float thr)
{
    unsigned int c;
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_bcnt1_i32_b64 %0, vcc\n\t"
                 "v_cmp_lt_f32 vcc, %1, %3\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_u32 %0, %0, vcc_lo\n\t"
                 "v_cmp_lt_f32 vcc, %1, %4\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_u32 %0, %0, vcc_lo\n\t"
                 "v_cmp_lt_f32 vcc, %1, %5\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_u32 %0, %0, vcc_lo"
                 : "=&s"(c) : "s"(thr), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "vcc", "scc");
    counter += c;
}
__device__ inline void count4_lt(unsigned int &counter, const float (&x)[4], float thr)
{
    unsigned int c;
    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_bcnt1_i32_b64 %0, vcc\n\t"
                 "v_cmp_gt_f32 vcc, %1, %3\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_u32 %0, %0, vcc_lo\n\t"
                 "v_cmp_gt_f32 vcc, %1, %4\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_u32 %0, %0, vcc_lo\n\t"
                 "v_cmp_gt_f32 vcc, %1, %5\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_u32 %0, %0, vcc_lo"
                 : "=&s"(c) : "s"(thr), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "vcc", "scc");
    counter += c;
}

// LARS_COUNT_MODE 3 / template CM == 3: per-lane float counters, two values per instruction:
//   c = sat(fma(x, +-2^40, -+thr * 2^40))   is exactly 1.0 where x > thr (resp. x < thr) and 0.0 elsewhere -- a value
// above the threshold is above it by at least one unit in the last place (>= 2^-32 here), which the factor turns into
// >= 256 -- and   counter += c.   No compare, no VCC, no scalar instruction.  Per-lane sums stay below 2^24.
__device__ inline void fcount2(f32x2 &counter, f32x2 x, f32x2 k, f32x2 b)
{
    f32x2 c;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(c) : "v"(x), "v"(k), "v"(b));
    counter += c;
}

// (Round 2 timed this kernel with single ingredients compiled out -- minimum / maximum, the float64 sums, the coverage counters, the
// quotient's correction step: profiles/r02_ablation_stats_kernel.txt.  Those builds gave wrong results by construction; the switch
// left the sources in round 5.)
template <int STATS, bool COUNT = true>
__device__ inline void push(WaveAcc &a, unsigned int &above, float x, float thr)
{
    a.mn = fminf(a.mn, x);
    a.mx = fmaxf(a.mx, x);
    const double xd = (double)x;
    a.sum += xd;
    if (STATS >= 3) a.sumsq += xd * xd;
    if (COUNT) {
        if (LARS_COUNT_MODE == 0) count_gt(above, x, thr);           // wave-uniform scalar counter
        else vcount_gt(above, x, thr);                               // per-lane counter, folded at flush
    }
}

// Histogram bin of an index value of a uint8 tile without looking at the edges.  x = (a-b)/(a+b)
// with bytes a, b, so the exact position T = 25 x + 25 = 50 a / (a + b) is either an integer or at
// least 1/510 away from one, while t = fma(x, 25, 25.5001) carries an error below 4e-6: adding 2^23
// rounds t to nearest and leaves floor(T) + 1 in the low mantissa bits.  On an exact hit the float32
// quotient is >= the float32 edge for every edge of numpy's linspace (checked for all 51 edges and,
// exhaustively over the 65536 byte pairs, by tests/test_properties_cpu.py), so floor(T) is numpy's bin.
// One packed fma + one packed add per pixel pair, one v_lshl_add_u32 per pixel, no LDS lookup.
// sign = -1 bins -x (NDWI from the GNDVI quotient).  `base` is the lane's byte address of copy
// lane % 16 of the index's bin 0, minus the constant part of the shifted float bits.
#define V2_HIST_MAGIC_BITS 0x4B000001u                   /* float bits of 2^23 + 1: "bin 0" */
__device__ inline f32x2 hist_pos2(f32x2 x, float sign)
{
    const f32x2 k = {25.0f * sign, 25.0f * sign}, c = {V2_HIST_MAGIC_C, V2_HIST_MAGIC_C}, big = {8388608.0f, 8388608.0f};
    return __builtin_elementwise_fma(x, k, c) + big;
}
__device__ inline void hist_add_pos(float pos, unsigned int base)
{
    const unsigned int addr = (__builtin_bit_cast(unsigned int, pos) << V2_HIST_SHIFT) + base;
    asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
}
__device__ inline void hist_add(float x, float sign, unsigned int base)
{
    hist_add_pos(__builtin_fmaf(x, 25.0f * sign, V2_HIST_MAGIC_C) + 8388608.0f, base);
}

// Block size: the 64 KiB table allows two blocks per CU.  A wave issues at most one vector instruction
// per ~12 cycles (tools/issuebench.py), so the VALU-bound statistics-only variants want all 8 waves per
// SIMD: 1024 threads per block.  With output planes the kernel needs more registers: 512.
template <bool OUT> struct V2Block { static constexpr int threads = OUT ? 512 : LARS_V2_STATS_THREADS; };

// SEL >= 1: also count every NDVI / GNDVI value in the 2048 linear buckets of the exact-median select (its first pass,
// fused: P.sel_hist[tile][stream][track 0][bucket]).  With the 64 KiB table that is exactly 80 KiB of LDS (two blocks
// per CU) because the reduction scratch then reuses the table's space once the loop is over.
template <unsigned MASK, bool WB, int STATS, bool OUT, bool NT, int SEL = 0, int CM = LARS_COUNT_MODE>
__global__ __launch_bounds__(V2Block<OUT>::threads, OUT ? 4 : LARS_V2_STATS_WAVES) void k_fused_v2(FusedParams P)
{
    constexpr int NTHR = V2Block<OUT>::threads;
    constexpr int NWAVES = NTHR / 64;
    constexpr bool RED_ALIASES_TABLE = WB && SEL;
    __shared__ __attribute__((aligned(16))) char s_mem[(WB ? V2_TABLE_BYTES : 0) + (STATS >= 2 ? V2_HIST_WORDS * 4 : 0) +
                                                       (SEL == 1 ? 2 * SELQ_BINS * 4 : 0) + (SEL == 2 ? 2 * V2_WIN_ROW * 4 : 0) +
                                                       (RED_ALIASES_TABLE ? 0 : NWAVES * 16 * sizeof(double))];
    char *s_tab = s_mem;                                                             // 64 KiB when WB
    unsigned int *s_hist = reinterpret_cast<unsigned int *>(s_mem + (WB ? V2_TABLE_BYTES : 0));
    unsigned int *s_sel = s_hist + (STATS >= 2 ? V2_HIST_WORDS : 0);                 // [2 streams][SELQ_BINS] when SEL
    unsigned int *s_win = s_sel + (SEL == 1 ? 2 * SELQ_BINS : 0);                    // [2 streams][V2_WIN_ROW] when SEL == 2 (instead of the buckets)
    double *s_red = RED_ALIASES_TABLE ? reinterpret_cast<double *>(s_mem)
                                      : reinterpret_cast<double *>(s_win + (SEL == 2 ? 2 * V2_WIN_ROW : 0));     // [NWAVES][16]

    constexpr bool NEED_R = (MASK & 1u) != 0;
    constexpr bool NEED_G = (MASK & 6u) != 0;
    constexpr bool WANT_NDVI = (MASK & 1u) != 0, WANT_GNDVI = (MASK & 2u) != 0, WANT_NDWI = (MASK & 4u) != 0;

    const int tid = threadIdx.x;
    const unsigned int lane = tid & 63;
    // byte address (LDS offset) of this lane's copy of bin 0 of index k, less the shifted bits of "bin 0"
    typedef __attribute__((address_space(3))) unsigned int lds_u32;
    const unsigned int hist_lds = STATS >= 2 ? (unsigned int)(unsigned long long)(lds_u32 *)s_hist : 0u;
    const unsigned int hb0 = hist_lds + (((unsigned)tid & (V2_HIST_COPIES - 1)) << 2) - (V2_HIST_MAGIC_BITS << V2_HIST_SHIFT);
    const unsigned int hb1 = hb0 + V2_HIST_ROWS * V2_HIST_COPIES * 4, hb2 = hb1 + V2_HIST_ROWS * V2_HIST_COPIES * 4;
    const unsigned int sel_lds = SEL == 1 ? (unsigned int)(unsigned long long)(lds_u32 *)s_sel : 0u;
    const unsigned int sb0 = sel_lds, sb1 = sb0 + SELQ_BINS * 4;     // NDVI row, GNDVI row
    // SEL == 2: per stream the count of values below its predicted window and the slot counts inside it (the one-pass
    // median, select_q.hip) -- instead of the 2048 buckets, same 16 KiB
    const unsigned int win_lds = SEL == 2 ? (unsigned int)(unsigned long long)(lds_u32 *)s_win : 0u;
    const unsigned int wr0 = win_lds - (SELQ_WIN_MAGIC_BITS << 2), wr1 = wr0 + V2_WIN_ROW * 4;     // see selq_window_add
    float wt0 = 0, wt1 = 0;                                          // fma bias of each stream's window (selq_window_bias)
    const int win_lo = (int)(SELQ_WIN_MAGIC_BITS + lane), win_hi = (int)(SELQ_WIN_MAGIC_BITS + 64 + SELQ_WIN_SLOTS + lane);
    const unsigned int lane_off4 = lane << 2;
    const long long tile = blockIdx.y;
    const long long npix = P.npix;
    const uint8_t *base = static_cast<const uint8_t *>(P.tiles) + tile * npix * 3;

    if (WB) {
        const uint8_t *t = P.wb_table + tile * 768;
        unsigned int *tab = reinterpret_cast<unsigned int *>(s_tab);
        // entry (v, copy) at dword v*64 + copy
        for (int i = tid; i < 256 * 64; i += NTHR) {
            const int v = i >> 6;
            tab[i] = (unsigned)t[v] | ((unsigned)t[256 + v] << 8) | ((unsigned)t[512 + v] << 16);
        }
    }
    if (STATS >= 2) {
        for (int i = tid; i < V2_HIST_WORDS; i += NTHR) s_hist[i] = 0;
    }
    if (SEL == 1) {
        for (int i = tid; i < 2 * SELQ_BINS; i += NTHR) s_sel[i] = 0;
    }
    if (SEL == 2) {
        for (int i = tid; i < 2 * V2_WIN_ROW; i += NTHR) s_win[i] = 0;
        wt0 = selq_window_bias((int)__builtin_amdgcn_readfirstlane(P.sel_win[blockIdx.y * 2]));         // wave-uniform
        wt1 = selq_window_bias((int)__builtin_amdgcn_readfirstlane(P.sel_win[blockIdx.y * 2 + 1]));
    }
    if (WB || STATS >= 2 || SEL) __syncthreads();

    WaveAcc acc_v, acc_g;                                  // NDVI, GNDVI-quotient (NDWI derives from it)
    acc_v.mn = acc_g.mn = __builtin_inff(); acc_v.mx = acc_g.mx = -__builtin_inff();
    acc_v.sum = acc_g.sum = 0; acc_v.sumsq = acc_g.sumsq = 0;
    unsigned int above_v = 0, above_g = 0, above_w = 0;
    f32x2 fc_v = {0.0f, 0.0f}, fc_g = {0.0f, 0.0f}, fc_w = {0.0f, 0.0f};          // CM == 3: float counters
    const f32x2 fc_kp = {1099511627776.0f, 1099511627776.0f}, fc_kn = {-1099511627776.0f, -1099511627776.0f};
    const f32x2 fc_bt = {-0.2f * 1099511627776.0f, -0.2f * 1099511627776.0f}, fc_b0 = {0.0f, 0.0f};

    // OUT == false: statistics only, every output pointer is known to be null at compile time
    float *const oi0 = (OUT && P.out_index[0]) ? P.out_index[0] + tile * npix : nullptr;
    float *const oi1 = (OUT && P.out_index[1]) ? P.out_index[1] + tile * npix : nullptr;
    float *const oi2 = (OUT && P.out_index[2]) ? P.out_index[2] + tile * npix : nullptr;
    uint8_t *const owb = (OUT && P.out_wb) ? P.out_wb + tile * npix * 3 : nullptr;
    uint8_t *const oc0 = (OUT && P.out_rgba[0]) ? P.out_rgba[0] + tile * npix * 4 : nullptr;
    uint8_t *const oc1 = (OUT && P.out_rgba[1]) ? P.out_rgba[1] + tile * npix * 4 : nullptr;
    uint8_t *const oc2 = (OUT && P.out_rgba[2]) ? P.out_rgba[2] + tile * npix * 4 : nullptr;
    const unsigned int *lut0 = reinterpret_cast<const unsigned int *>(P.cmap_lut[0]);
    const unsigned int *lut1 = reinterpret_cast<const unsigned int *>(P.cmap_lut[1]);
    const unsigned int *lut2 = reinterpret_cast<const unsigned int *>(P.cmap_lut[2]);

    const long long nquads = npix >> 2;
    const long long stride = (long long)gridDim.x * NTHR;

    auto do_quad = [&](long long q, unsigned int w0, unsigned int w1, unsigned int w2) {
        // bytes: r0 g0 n0 r1 | g1 n1 r2 g2 | n2 r3 g3 n3
        const unsigned int wr[4] = {w0, w0, w1, w2}, wg[4] = {w0, w1, w1, w2}, wn[4] = {w0, w1, w2, w2};
        constexpr int br[4] = {0, 3, 2, 1}, bg[4] = {1, 0, 3, 2}, bn[4] = {2, 1, 0, 3};
        if (WB && owb) {
            unsigned int e[12];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                e[3 * px] = sample_entry<WB>(wr[px], br[px], lane_off4, s_tab) & 0xFFu;
                e[3 * px + 1] = (sample_entry<WB>(wg[px], bg[px], lane_off4, s_tab) >> 8) & 0xFFu;
                e[3 * px + 2] = (sample_entry<WB>(wn[px], bn[px], lane_off4, s_tab) >> 16) & 0xFFu;
            }
            unsigned int *o = reinterpret_cast<unsigned int *>(owb + q * 12);
            o[0] = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24);
            o[1] = e[4] | (e[5] << 8) | (e[6] << 16) | (e[7] << 24);
            o[2] = e[8] | (e[9] << 8) | (e[10] << 16) | (e[11] << 24);
        }
        if (MASK == 0u) return;
        // samples -> float (white balanced through the LDS table when WB)
        float fn[4], fr[4], fg[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            fn[px] = sample<WB>(wn[px], bn[px], 2, lane_off4, s_tab);
            if (NEED_R) fr[px] = sample<WB>(wr[px], br[px], 0, lane_off4, s_tab);
            if (NEED_G) fg[px] = sample<WB>(wg[px], bg[px], 1, lane_off4, s_tab);
        }
        // quotients, two pixels per packed instruction (v_pk_add/mul/fma_f32)
        float v0[4], v1[4], v2[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 N = {fn[2 * h], fn[2 * h + 1]};
            const f32x2 Ne = N + (f32x2){LARS_DEN_EPS, LARS_DEN_EPS};
            if (WANT_NDVI) {
                const f32x2 R = {fr[2 * h], fr[2 * h + 1]};
                const f32x2 x = exact_quot2(N - R, Ne + R);
                v0[2 * h] = x.x; v0[2 * h + 1] = x.y;
                if (STATS >= 1 && CM == 3) fcount2(fc_v, x, fc_kp, fc_bt);
                if (STATS >= 2) {
                    const f32x2 p = hist_pos2(x, 1.0f);
                    hist_add_pos(p.x, hb0); hist_add_pos(p.y, hb0);
                }
                if (SEL == 1) {
                    const f32x2 p = selq_t2(x);
                    selq_add_bucket(p.x, sb0); selq_add_bucket(p.y, sb0);
                }
                if (SEL == 2) {
                    const f32x2 p = __builtin_elementwise_fma(x, (f32x2){SELQ_WIN_SCALE, SELQ_WIN_SCALE}, (f32x2){wt0, wt0});
                    selq_window_add(p.x, wr0, win_lo, win_hi); selq_window_add(p.y, wr0, win_lo, win_hi);
                }
            }
            if (NEED_G) {
                const f32x2 G = {fg[2 * h], fg[2 * h + 1]};
                const f32x2 x = exact_quot2(N - G, Ne + G);
                v1[2 * h] = x.x; v1[2 * h + 1] = x.y;
                if (STATS >= 1 && CM == 3 && WANT_GNDVI) fcount2(fc_g, x, fc_kp, fc_bt);
                if (STATS >= 1 && CM == 3 && WANT_NDWI) fcount2(fc_w, x, fc_kn, fc_b0);          // -x > 0
                if (STATS >= 2 && WANT_GNDVI) {
                    const f32x2 p = hist_pos2(x, 1.0f);
                    hist_add_pos(p.x, hb1); hist_add_pos(p.y, hb1);
                }
                if (SEL == 1) {
                    const f32x2 p = selq_t2(x);
                    selq_add_bucket(p.x, sb1); selq_add_bucket(p.y, sb1);
                }
                if (SEL == 2) {
                    const f32x2 p = __builtin_elementwise_fma(x, (f32x2){SELQ_WIN_SCALE, SELQ_WIN_SCALE}, (f32x2){wt1, wt1});
                    selq_window_add(p.x, wr1, win_lo, win_hi); selq_window_add(p.y, wr1, win_lo, win_hi);
                }
                if (STATS >= 2 && WANT_NDWI) {
                    const f32x2 p = hist_pos2(x, -1.0f);
                    hist_add_pos(p.x, hb2); hist_add_pos(p.y, hb2);
                }
            }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            if (WANT_NDVI) {
                const float x = v0[px];
                if (STATS >= 1) push<STATS, (CM == 1 || CM == 2)>(acc_v, above_v, x, 0.2f);
            }
            if (NEED_G) {
                const float x = v1[px];
                if (STATS >= 1) {
                    acc_g.mn = fminf(acc_g.mn, x);
                    acc_g.mx = fmaxf(acc_g.mx, x);
                    const double xd = (double)x;
                    acc_g.sum += xd;
                    if (STATS >= 3) acc_g.sumsq += xd * xd;
                    if (CM == 1) {
                        if (WANT_GNDVI) vcount_gt(above_g, x, 0.2f);
                        if (WANT_NDWI) vcount_lt(above_w, x, 0.0f);           // -x > 0
                    }
                }
                if (WANT_NDWI) {
                    v2[px] = 0.0f - x;                     // +0.0 where the quotient is zero
                }
            }
        }
        if (STATS >= 1 && CM == 0) {
            if (WANT_NDVI) count4_gt(above_v, v0, 0.2f);
            if (WANT_GNDVI) count4_gt(above_g, v1, 0.2f);
            if (WANT_NDWI) count4_lt(above_w, v1, 0.0f);                   // -x > 0
        } else if (STATS >= 1 && CM == 2) {
            if (WANT_GNDVI) count4_gt(above_g, v1, 0.2f);
            if (WANT_NDWI) count4_lt(above_w, v1, 0.0f);
        }
        if (WANT_NDVI && oi0) {
            store4<NT>(oi0 + q * 4, v0[0], v0[1], v0[2], v0[3]);
        }
        if (WANT_GNDVI && oi1) {
            store4<NT>(oi1 + q * 4, v1[0], v1[1], v1[2], v1[3]);
        }
        if (WANT_NDWI && oi2) {
            store4<NT>(oi2 + q * 4, v2[0], v2[1], v2[2], v2[3]);
        }
        if (WANT_NDVI && oc0)
            *reinterpret_cast<uint4 *>(oc0 + q * 16) = make_uint4(lut0[cmap_index(v0[0])], lut0[cmap_index(v0[1])],
                                                                  lut0[cmap_index(v0[2])], lut0[cmap_index(v0[3])]);
        if (WANT_GNDVI && oc1)
            *reinterpret_cast<uint4 *>(oc1 + q * 16) = make_uint4(lut1[cmap_index(v1[0])], lut1[cmap_index(v1[1])],
                                                                  lut1[cmap_index(v1[2])], lut1[cmap_index(v1[3])]);
        if (WANT_NDWI && oc2)
            *reinterpret_cast<uint4 *>(oc2 + q * 16) = make_uint4(lut2[cmap_index(v2[0])], lut2[cmap_index(v2[1])],
                                                                  lut2[cmap_index(v2[2])], lut2[cmap_index(v2[3])]);
    };

    // Software pipeline: four 12-byte loads in flight per lane while the oldest quad is processed.
    // buffer_load with the tile as a raw buffer: the lane offset is fixed, the per-iteration offset
    // is a scalar (no vector address arithmetic), and loads past the end return zeros (no clamps).
    const long long q0 = (long long)blockIdx.x * NTHR + tid;
    const long long niter = (nquads + stride - 1) / stride;          // same for every lane of the grid
    if (niter > 0) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(nquads * 12), 0x00020000);
        const unsigned int voff = (unsigned int)q0 * 12u;
        const unsigned int step_b = (unsigned int)stride * 12u;
        u32x3 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, (unsigned)k * step_b, 0);
        long long it = 0;
        unsigned int soff = 4u * step_b;
        // every lane's quad is in range while it < niter - 1
        for (; it + 4 <= niter - 1; it += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // consume slot k, then refill it: the refill lands in the registers just freed
                // (no copies, no vmcnt(0) at the loop head) and has three quads of work to hide behind
                do_quad(q0 + (it + k) * stride, w[k].x, w[k].y, w[k].z);
                __builtin_amdgcn_sched_barrier(0);
                w[k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff + (unsigned)k * step_b, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            soff += 4u * step_b;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long qq = q0 + (it + k) * stride;
            if (it + k < niter && qq < nquads) do_quad(qq, w[k].x, w[k].y, w[k].z);
        }
    }

    // tail pixels (npix % 4): lanes 0..2 of block 0, scalar path
    if (blockIdx.x == 0 && tid < (int)(npix & 3)) {
        const long long i = nquads * 4 + tid;
        unsigned int r = base[i * 3], g = base[i * 3 + 1], n = base[i * 3 + 2];
        if (WB) {
            const unsigned int *tab = reinterpret_cast<const unsigned int *>(s_tab);
            r = tab[r * 64] & 0xFFu; g = (tab[g * 64] >> 8) & 0xFFu; n = (tab[n * 64] >> 16) & 0xFFu;
            if (owb) { owb[i * 3] = (uint8_t)r; owb[i * 3 + 1] = (uint8_t)g; owb[i * 3 + 2] = (uint8_t)n; }
        }
        const float fr = (float)r, fg = (float)g, fn = (float)n;
        if (WANT_NDVI) {
            const float x = norm_diff_fast(fn, fr);
            if (STATS >= 1) push<STATS>(acc_v, above_v, x, 0.2f);
            if (STATS >= 2) hist_add(x, 1.0f, hb0);
            if (SEL == 1) selq_add_bucket(selq_t(x), sb0);
            if (SEL == 2) selq_window_add(__builtin_fmaf(x, SELQ_WIN_SCALE, wt0), wr0, win_lo, win_hi);
            if (oi0) oi0[i] = x;
            if (oc0) reinterpret_cast<unsigned int *>(oc0)[i] = lut0[cmap_index(x)];
        }
        if (NEED_G) {
            const float x = norm_diff_fast(fn, fg);
            if (SEL == 1) selq_add_bucket(selq_t(x), sb1);
            if (SEL == 2) selq_window_add(__builtin_fmaf(x, SELQ_WIN_SCALE, wt1), wr1, win_lo, win_hi);
            if (STATS >= 1) {
                acc_g.mn = fminf(acc_g.mn, x); acc_g.mx = fmaxf(acc_g.mx, x);
                const double xd = (double)x;
                acc_g.sum += xd;
                if (STATS >= 3) acc_g.sumsq += xd * xd;
                if (CM == 1) {
                    if (WANT_GNDVI) vcount_gt(above_g, x, 0.2f);
                    if (WANT_NDWI) vcount_lt(above_w, x, 0.0f);
                } else {
                    if (WANT_GNDVI) count_gt(above_g, x, 0.2f);
                    if (WANT_NDWI) count_lt(above_w, x, 0.0f);
                }
            }
            if (WANT_GNDVI) {
                if (STATS >= 2) hist_add(x, 1.0f, hb1);
                if (oi1) oi1[i] = x;
                if (oc1) reinterpret_cast<unsigned int *>(oc1)[i] = lut1[cmap_index(x)];
            }
            if (WANT_NDWI) {
                const float w = 0.0f - x;
                if (STATS >= 2) hist_add(x, -1.0f, hb2);
                if (oi2) oi2[i] = w;
                if (oc2) reinterpret_cast<unsigned int *>(oc2)[i] = lut2[cmap_index(w)];
            }
        }
    }

    // the histogram atomics are inline asm: the compiler does not count them, wait for them by hand
    if (STATS >= 2 || SEL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (RED_ALIASES_TABLE) __syncthreads();               // every wave is done with the table before its space is reused
    if (STATS >= 1) {
        // wave fold (scalar counters are already wave totals; per-lane ones are summed here)
        if (CM == 3) {
            above_v = wave_sum_u32((unsigned int)(fc_v.x + fc_v.y));
            above_g = wave_sum_u32((unsigned int)(fc_g.x + fc_g.y));
            above_w = wave_sum_u32((unsigned int)(fc_w.x + fc_w.y));
        }
        if ((CM == 1 || CM == 2)) above_v = wave_sum_u32(above_v);
        if (CM == 1) { above_g = wave_sum_u32(above_g); above_w = wave_sum_u32(above_w); }
        for (int off = 32; off >= 1; off >>= 1) {
            if (WANT_NDVI) {
                acc_v.mn = fminf(acc_v.mn, __shfl_xor(acc_v.mn, off)); acc_v.mx = fmaxf(acc_v.mx, __shfl_xor(acc_v.mx, off));
                acc_v.sum += __shfl_xor(acc_v.sum, off);
                if (STATS >= 3) acc_v.sumsq += __shfl_xor(acc_v.sumsq, off);
            }
            if (NEED_G) {
                acc_g.mn = fminf(acc_g.mn, __shfl_xor(acc_g.mn, off)); acc_g.mx = fmaxf(acc_g.mx, __shfl_xor(acc_g.mx, off));
                acc_g.sum += __shfl_xor(acc_g.sum, off);
                if (STATS >= 3) acc_g.sumsq += __shfl_xor(acc_g.sumsq, off);
            }
        }
        const int wave = tid >> 6;
        double *row = s_red + wave * 16;
        if (lane == 0) {
            row[0] = acc_v.sum; row[1] = acc_v.sumsq; row[2] = (double)acc_v.mn; row[3] = (double)acc_v.mx; row[4] = (double)above_v;
            row[5] = acc_g.sum; row[6] = acc_g.sumsq; row[7] = (double)acc_g.mn; row[8] = (double)acc_g.mx;
            row[9] = (double)above_g; row[10] = (double)above_w;
        }
        __syncthreads();
        if (tid < 11) {
            // column tid of the wave rows: sums (exact), minima (columns 2, 7), maxima (3, 8)
            const bool is_min = tid == 2 || tid == 7, is_max = tid == 3 || tid == 8;
            double t = s_red[tid];
#pragma unroll 1
            for (int w = 1; w < NWAVES; ++w) {
                const double o = s_red[w * 16 + tid];
                t = is_min ? fmin(t, o) : is_max ? fmax(t, o) : t + o;
            }
            s_red[tid] = t;
        }
        __syncthreads();
        if (tid == 0) {
            double t[11];
            for (int j = 0; j < 11; ++j) t[j] = s_red[j];
            StatsAccView *rec = reinterpret_cast<StatsAccView *>(P.stats + tile * 3);
            if (WANT_NDVI && (P.mask & 1u)) {
                atomicAdd(&rec[0].sum_fx, (unsigned long long)__double2ll_rn(t[0] * LARS_FX_SCALE));
                if (STATS >= 3) atomicAdd(&rec[0].sumsq_fx, (unsigned long long)__double2ll_rn(t[1] * LARS_FX_SCALE));
                atomicAdd(&rec[0].above, (unsigned long long)t[4]);
                atomicMin(&rec[0].min_key, f64_key(t[2]));
                atomicMax(&rec[0].max_key, f64_key(t[3]));
            }
            if (WANT_GNDVI && (P.mask & 2u)) {
                atomicAdd(&rec[1].sum_fx, (unsigned long long)__double2ll_rn(t[5] * LARS_FX_SCALE));
                if (STATS >= 3) atomicAdd(&rec[1].sumsq_fx, (unsigned long long)__double2ll_rn(t[6] * LARS_FX_SCALE));
                atomicAdd(&rec[1].above, (unsigned long long)t[9]);
                atomicMin(&rec[1].min_key, f64_key(t[7]));
                atomicMax(&rec[1].max_key, f64_key(t[8]));
            }
            if (WANT_NDWI && (P.mask & 4u)) {
                // NDWI = -GNDVI: sum negates, squares equal, extrema swap (0.0 - x keeps zeros positive,
                // which the double keys below reproduce: -(+0.0) never occurs because min/max of the
                // quotient only reach 0 as +0.0 and 0.0 - 0.0 = +0.0)
                atomicAdd(&rec[2].sum_fx, (unsigned long long)__double2ll_rn((0.0 - t[5]) * LARS_FX_SCALE));
                if (STATS >= 3) atomicAdd(&rec[2].sumsq_fx, (unsigned long long)__double2ll_rn(t[6] * LARS_FX_SCALE));
                atomicAdd(&rec[2].above, (unsigned long long)t[10]);
                atomicMin(&rec[2].min_key, f64_key(0.0 - t[8]));
                atomicMax(&rec[2].max_key, f64_key(0.0 - t[7]));
            }
        }
        if (STATS >= 2) {
            __syncthreads();
            StatsAccView *rec = reinterpret_cast<StatsAccView *>(P.stats + tile * 3);
            if (tid < 3 * LARS_HIST_BINS) {
                const int k = tid / LARS_HIST_BINS;
                if ((MASK & (1u << k)) && (P.mask & (1u << k))) {
                    const int b = tid - k * LARS_HIST_BINS;
                    const unsigned int *rowh = s_hist + (k * V2_HIST_ROWS + b) * V2_HIST_COPIES;
                    unsigned int v = 0;
                    // row 50 holds x == 1.0: numpy's last bin is closed on the right
                    const int nwords = (b == LARS_HIST_BINS - 1) ? 2 * V2_HIST_COPIES : V2_HIST_COPIES;
                    for (int j = 0; j < nwords; ++j) v += rowh[j];
                    if (v) atomicAdd(&rec[k].hist[b], (unsigned long long)v);
                }
            }
        }
        if (SEL == 1) {
            // the barrier of the statistics flush above already ordered every wave's bucket atomics
            unsigned int *h = P.sel_hist + tile * (4 * SELQ_BINS);
            for (int i = tid; i < 2 * SELQ_BINS; i += NTHR) {
                const unsigned int v = s_sel[i];
                if (v) atomicAdd(&h[(i >> 11) * (2 * SELQ_BINS) + (i & (SELQ_BINS - 1))], v);      // stream row, track 0
            }
        }
        if (SEL == 2) {
            // words 0..63 of a row: values below the window; 64 .. 64 + SLOTS - 1: the slots; the rest: above (not needed)
            for (int i = tid; i < 2 * V2_WIN_ROW; i += NTHR) {
                const unsigned int v = s_win[i];
                if (!v) continue;
                const int stream = i / V2_WIN_ROW, w = i % V2_WIN_ROW;
                if (w < 64) atomicAdd(&P.sel_below[tile * 2 + stream], v);
                else if (w < 64 + SELQ_WIN_SLOTS) atomicAdd(&P.sel_win_hist[(tile * 2 + stream) * SELQ_WIN_SLOTS + (w - 64)], v);
            }
        }
    }
}


}  // namespace lars

// ===========================================================================
// launch glue used by fused.hip's entry points
// ===========================================================================
using namespace lars;

namespace lars {


template <unsigned MASK, bool WB, int STATS>
static void v2_launch_out(bool out, bool nt, dim3 grid, hipStream_t s, const FusedParams &P)
{
    if (!out) hipLaunchKernelGGL((k_fused_v2<MASK, WB, STATS, false, false>), grid, dim3(V2Block<false>::threads), 0, s, P);
    else if (nt) hipLaunchKernelGGL((k_fused_v2<MASK, WB, STATS, true, true>), grid, dim3(512), 0, s, P);
    else hipLaunchKernelGGL((k_fused_v2<MASK, WB, STATS, true, false>), grid, dim3(512), 0, s, P);
}
template <unsigned MASK, bool WB>
static void v2_launch_stats(int stats, bool out, bool nt, dim3 grid, hipStream_t s, const FusedParams &P)
{
    if (stats == 0) v2_launch_out<MASK, WB, 0>(out, nt, grid, s, P);
    else if (stats == 1) v2_launch_out<MASK, WB, 1>(out, nt, grid, s, P);
    else if (stats == 2) v2_launch_out<MASK, WB, 2>(out, nt, grid, s, P);
    else v2_launch_out<MASK, WB, 3>(out, nt, grid, s, P);
}
template <unsigned MASK>
static void v2_launch_wb(bool wb, int stats, bool out, bool nt, dim3 grid, hipStream_t s, const FusedParams &P)
{
    if (wb) v2_launch_stats<MASK, true>(stats, out, nt, grid, s, P);
    else v2_launch_stats<MASK, false>(stats, out, nt, grid, s, P);
}

template <unsigned MASK>
static void v2_launch_sel(bool wb, int stats, dim3 grid, hipStream_t s, const FusedParams &P)
{
    const dim3 block(V2Block<false>::threads);
    // with a predicted window (P.sel_win) the slots inside it are counted as well: basic statistics only
    if (P.sel_win && stats == 1) {
        if (wb) hipLaunchKernelGGL((k_fused_v2<MASK, true, 1, false, false, 2>), grid, block, 0, s, P);
        else hipLaunchKernelGGL((k_fused_v2<MASK, false, 1, false, false, 2>), grid, block, 0, s, P);
        return;
    }
    if (wb && stats >= 3) hipLaunchKernelGGL((k_fused_v2<MASK, true, 3, false, false, 1>), grid, block, 0, s, P);
    else if (wb && stats == 2) hipLaunchKernelGGL((k_fused_v2<MASK, true, 2, false, false, 1>), grid, block, 0, s, P);
    else if (wb) hipLaunchKernelGGL((k_fused_v2<MASK, true, 1, false, false, 1>), grid, block, 0, s, P);
    else if (stats >= 3) hipLaunchKernelGGL((k_fused_v2<MASK, false, 3, false, false, 1>), grid, block, 0, s, P);
    else if (stats == 2) hipLaunchKernelGGL((k_fused_v2<MASK, false, 2, false, false, 1>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_fused_v2<MASK, false, 1, false, false, 1>), grid, block, 0, s, P);
}
// statistics + the select's bucket pass in one kernel (mask 1, 2, 4 or 7; no output planes)
void fused_v2_sel_launch(unsigned mask, bool wb, int stats, dim3 grid, hipStream_t s, const FusedParams &P)
{
    switch (mask) {
    case 1u: v2_launch_sel<1u>(wb, stats, grid, s, P); break;
    case 2u: v2_launch_sel<2u>(wb, stats, grid, s, P); break;
    case 4u: v2_launch_sel<4u>(wb, stats, grid, s, P); break;
    default: v2_launch_sel<7u>(wb, stats, grid, s, P); break;
    }
}

int fused_v2_threads(bool any_out) { return any_out ? V2Block<true>::threads : V2Block<false>::threads; }

void fused_v2_launch(unsigned mask, bool wb, int stats, bool nt, dim3 grid, hipStream_t s, const FusedParams &P)
{
    const bool out = P.out_wb || P.out_index[0] || P.out_index[1] || P.out_index[2] || P.out_rgba[0] || P.out_rgba[1] ||
                     P.out_rgba[2];
    switch (mask) {
    case 0u: hipLaunchKernelGGL((k_fused_v2<0u, true, 0, true, false>), grid, dim3(512), 0, s, P); break;
    case 1u: v2_launch_wb<1u>(wb, stats, out, nt, grid, s, P); break;
    case 2u: v2_launch_wb<2u>(wb, stats, out, nt, grid, s, P); break;
    case 4u: v2_launch_wb<4u>(wb, stats, out, nt, grid, s, P); break;
    default: v2_launch_wb<7u>(wb, stats, out, nt, grid, s, P); break;
    }
}

void chan_hist_v2_launch(const uint8_t *tiles, long long npix, unsigned int *hist, dim3 grid, hipStream_t s, int channels)
{
    if (channels == 4) hipLaunchKernelGGL(k_chan_hist_u8c3_v2<4>, grid, dim3(1024), 0, s, tiles, npix, hist);
    else hipLaunchKernelGGL(k_chan_hist_u8c3_v2<3>, grid, dim3(1024), 0, s, tiles, npix, hist);
}

int quot_check_launch(unsigned int max_den, unsigned long long *mismatches_dev, unsigned int *first_bad_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_quot_check, dim3(2048), dim3(256), 0, s, max_den, mismatches_dev, first_bad_dev);
    return launch_check("k_quot_check");
}

}  // namespace lars
