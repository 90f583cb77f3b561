// Per-pixel arithmetic of the plane-writing kernels (fused.hip).  Reference semantics: see fused.hip.
#pragma once
#include "common.h"
#include "device_common.h"

namespace lars {

// U8DOM: x is a quotient of uint8 samples, whose histogram bin can be read off the mantissa of
// fma(x, 25, 25.5001) + 2^23 (see hist_pos2 in fused_v2.hip; exhaustively equal to numpy's bin) instead of the cell table
// HSTRIDE: words between consecutive bins of the LDS histogram (lane-private copies interleaved bin by bin)
template <int STATS, bool U8DOM = false, int HSTRIDE = 1>
__device__ inline void acc_push(Acc &a, float x, float thr, unsigned int *s_hist, const HistCell<float> *s_edges)
{
    if (STATS >= 1) {
        a.mn = fminf(a.mn, x);
        a.mx = fmaxf(a.mx, x);
        const double xd = (double)x;
        a.sum += xd;
        if (STATS >= 3) a.sumsq += xd * xd;
        a.above += (x > thr) ? 1u : 0u;
    }
    if (STATS >= 2) {
        if (U8DOM) {
            const float u = __builtin_fmaf(x, 25.0f, 25.5001f) + 8388608.0f;
            const unsigned int b1 = __builtin_bit_cast(unsigned int, u) & 0x7FFFFFu;      // bin + 1, 51 for x == 1.0
            atomicAdd(&s_hist[(b1 > 50u ? 49u : b1 - 1u) * HSTRIDE], 1u);
        } else {
            atomicAdd(&s_hist[hist_bin_f32(x, s_edges) * HSTRIDE], 1u);
        }
    }
}

// One pixel: white-balanced (or raw) band values in, everything out.
template <unsigned MASK, int STATS, bool RT_MASK, bool U8DOM = false, int HSTRIDE = 1>
__device__ inline void pixel_math(float r, float g, float n, unsigned rt_mask,
                                  float &o_ndvi, float &o_gndvi, float &o_ndwi,
                                  Acc *acc, unsigned int *s_hist, const HistCell<float> *s_edges)
{
    const bool want_ndvi = RT_MASK ? (rt_mask & 1u) : (MASK & 1u);
    const bool want_gndvi = RT_MASK ? (rt_mask & 2u) : (MASK & 2u);
    const bool want_ndwi = RT_MASK ? (rt_mask & 4u) : (MASK & 4u);
    if (want_ndvi) {
        o_ndvi = norm_diff(n, r);
        acc_push<STATS, U8DOM, HSTRIDE>(acc[0], o_ndvi, 0.2f, s_hist + 0 * LARS_HIST_BINS * HSTRIDE, s_edges);
    }
    float gq = 0.0f;
    if (want_gndvi || want_ndwi) gq = norm_diff(n, g);
    if (want_gndvi) {
        o_gndvi = gq;
        acc_push<STATS, U8DOM, HSTRIDE>(acc[1], o_gndvi, 0.2f, s_hist + 1 * LARS_HIST_BINS * HSTRIDE, s_edges);
    }
    if (want_ndwi) {
        // (g-n)/(g+n) == -(n-g)/(n+g) bit for bit, and +0.0 where the quotient is zero
        o_ndwi = 0.0f - gq;
        acc_push<STATS, U8DOM, HSTRIDE>(acc[2], o_ndwi, 0.0f, s_hist + 2 * LARS_HIST_BINS * HSTRIDE, s_edges);
    }
}

typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
typedef float ff32x4 __attribute__((ext_vector_type(4)));
__device__ inline void store_plane4(float *dst, const float (&v)[4], bool nt)
{
    const ff32x4 x = {v[0], v[1], v[2], v[3]};
    if (nt) __builtin_nontemporal_store(x, reinterpret_cast<ff32x4 *>(dst));
    else *reinterpret_cast<ff32x4 *>(dst) = x;
}
typedef unsigned int fu32x2 __attribute__((ext_vector_type(2)));

}  // namespace lars
