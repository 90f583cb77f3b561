// Device-side pieces shared by the fused kernels (fused.hip, fused_v2.hip).
#pragma once
#include "common.h"

namespace lars {

struct FusedParams {
    const void *tiles;
    long long npix;
    int channels;
    const uint8_t *wb_table;       // [ntiles][3][NVAL] or null
    float *out_index[3];
    uint8_t *out_wb;
    uint8_t *out_rgba[3];
    const uint8_t *cmap_lut[3];
    lars_stats *stats;
    unsigned int mask;             // runtime copy (generic kernel)
    unsigned int flags;
    unsigned int *sel_hist;        // [ntiles][2 streams][2 tracks][2048]: bucket pass of the median select, or null
    const unsigned int *sel_win;   // [ntiles][2]: first slot (int) of each stream's predicted window (select_q.hip, v2_device.h), or null
    unsigned int *sel_win_hist;    // [ntiles][2][SELQ_WIN_SLOTS]: slot counts inside the window
    unsigned int *sel_below;       // [ntiles][2]: values below the window
#ifdef LARS_LAB_LAYOUT
    // laboratory build only (make lablayout; tools/lab/interleave.py): pixels between consecutive tiles of an index plane.  Kept out of
    // the product build on purpose: three more scalars in the headline instantiation took it from 253 to 264 registers = from two
    // resident waves per SIMD to one, and every arena then ran at the slow class's level (NOTES.md, round 4)
    long long out_tile_stride;
#endif
};

struct Acc {
    float mn, mx;
    double sum, sumsq;
    unsigned int above;
};
__device__ inline void acc_init(Acc &a) { a.mn = __builtin_inff(); a.mx = -__builtin_inff(); a.sum = 0; a.sumsq = 0; a.above = 0; }

// IEEE float32 (a-b)/(a+b); +0.0 where a+b == 0 (the reference's epsilon only
// matters there: process-images.py:464-482).
// The quotient is rcp + mul + 2 fma instead of the compiler's 12-instruction IEEE division:
// bit-identical for every operand pair of the uint8 / uint16 domains (lars_d_quot_selfcheck
// proves it exhaustively on the device).  LARS_IEEE_DIV keeps the plain division for A/B runs.
__device__ inline float norm_diff(float a, float b)
{
    const float s = a + b;
    const float d = a - b;
#ifdef LARS_IEEE_DIV
    return d / (s == 0.0f ? 1.0f : s);
#else
    const float den = fmaxf(s, 1.0f);
    const float r = __builtin_amdgcn_rcpf(den);
    const float q0 = d * r;
    const float e = __builtin_fmaf(-q0, den, d);
    return __builtin_fmaf(e, r, q0);
#endif
}

// Bin of numpy.histogram(bins=50, range=(-1,1)) for x in [-1, 1] without searching the edges.
// [-1, 1] is cut into 64 cells of width 1/32 (< the bin width 0.04, so a cell holds at most one
// edge).  cell = trunc((x+1)*32); entry = {the edge inside the cell or +inf, number of edges below
// the cell - 1}; bin = base + (x >= edge).  Rounding of x+1 can only move x onto a cell boundary
// from below, and edges that coincide with a boundary (-1, 0) are stored as that cell's edge, so
// the compare still classifies such an x correctly.  The last edge (1.0) is not counted: the last
// bin is closed.
#define LARS_HIST_CELLS 65
template <typename T>
struct HistCell {
    T edge;
    long long base;      // same size as a double edge; int would do
};
template <>
struct HistCell<float> {
    float edge;
    int base;
};
template <typename T>
__device__ inline void hist_cells_init(HistCell<T> *cells, int tid)
{
    if (tid < LARS_HIST_CELLS) {
        const T lo = (T)tid / (T)32 - (T)1, hi = (T)(tid + 1) / (T)32 - (T)1;
        int below = 0;
        T inside = (T)__builtin_inf();
        for (int i = 0; i < LARS_HIST_BINS; ++i) {            // edges 0..49; edge 50 closes the last bin
            const T e = (T)hist_edge_f64(i);
            if (e < lo) ++below;
            else if (e < hi) inside = e;
        }
        cells[tid].edge = inside;
        cells[tid].base = below - 1;
    }
}
template <typename T>
__device__ inline int hist_bin_cell(T x, const HistCell<T> *cells)
{
    const int c = (int)__builtin_fma(x, (T)32, (T)32);
    const HistCell<T> e = cells[c];
    return (int)e.base + (x >= e.edge ? 1 : 0);
}
__device__ inline int hist_bin_f32(float x, const HistCell<float> *cells)
{
    const int c = (int)__builtin_fmaf(x, 32.0f, 32.0f);
    const HistCell<float> e = cells[c];
    return e.base + (x >= e.edge ? 1 : 0);
}

__device__ inline unsigned int cmap_index(float x)
{
    const float s = (x + 1.0f) * 128.0f;
    int i = (int)s;
    i = i < 0 ? 0 : (i > 255 ? 255 : i);
    return (unsigned)i;
}


// White-balance level of sample value v: process-images.py:438/:441 (float64 arithmetic, clip,
// float32 store, truncating uint8 cast) or process-rgn.py:29-33/:44 (inner clip, direct cast).
__device__ inline unsigned int wb_level(int v, double p_lo, double p_hi, int rgn_variant)
{
    double x = (double)v;
    if (rgn_variant) x = fmin(fmax(x, p_lo), p_hi);
    double y = (x - p_lo) / (p_hi - p_lo) * 255.0;
    if (y != y) return 0u;                                  // NaN -> uint8 cast gives 0
    y = y < 0.0 ? 0.0 : (y > 255.0 ? 255.0 : y);
    return rgn_variant ? (unsigned int)(int)y : (unsigned int)(int)(float)y;
}

// Fill one channel of a uint16 tile's table blob: the 65536-entry table, its threshold form
// T[k] = smallest v with table[v] >= k (T[0] = 0, T[256..] = 65536), and (p2, 255/(p98-p2)).
// 256 threads; s_thr is 260 words of LDS.
__device__ inline void u16_fill_blob(uint8_t *blob, int c, double p_lo, double p_hi, int rgn_variant,
                                     unsigned int *s_thr, int tid)
{
    for (int k = tid; k < 260; k += 256) s_thr[k] = 65536u;
    __syncthreads();
    uint8_t *out = blob + (long long)c * 65536;
    unsigned int prev = tid == 0 ? 0u : wb_level(tid * 256 - 1, p_lo, p_hi, rgn_variant);
    for (int j = 0; j < 256; ++j) {
        const int v = tid * 256 + j;
        const unsigned int o = wb_level(v, p_lo, p_hi, rgn_variant);
        out[v] = (uint8_t)o;
        for (unsigned int k = prev + 1; k <= o; ++k) s_thr[k] = (unsigned)v;     // monotone staircase
        prev = o > prev ? o : prev;
    }
    __syncthreads();
    unsigned int *thr = reinterpret_cast<unsigned int *>(blob + LARS_U16_THR_OFFSET) + c * 260;
    for (int k = tid; k < 260; k += 256) thr[k] = k == 0 ? 0u : (k >= 256 ? 65536u : s_thr[k]);
    if (tid == 0) {
        double *par = reinterpret_cast<double *>(blob + LARS_U16_PAR_OFFSET) + c * 2;
        const double span = p_hi - p_lo;
        par[0] = p_lo;
        par[1] = span > 0.0 ? 255.0 / span : 1e300;        // degenerate channel: <= p -> 0, above -> 255
    }
}

}  // namespace lars
