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
};

struct Acc {
    float mn, mx;
    double sum, sumsq;
    unsigned int above;
};
__device__ inline void acc_init(Acc &a) { a.mn = __builtin_inff(); a.mx = -__builtin_inff(); a.sum = 0; a.sumsq = 0; a.above = 0; }

// IEEE float32 (a-b)/(a+b); +0.0 where a+b == 0 (the reference's epsilon only
// matters there: process-images.py:464-482).
__device__ inline float norm_diff(float a, float b)
{
    const float s = a + b;
    const float d = a - b;
    return d / (s == 0.0f ? 1.0f : s);
}

// bin of numpy.histogram(bins=50, range=(-1,1)) for float32 x in [-1, 1]
__device__ inline int hist_bin_f32(float x, const float *edges)
{
    int b = (int)((x + 1.0f) * 25.0f);
    b = b < 0 ? 0 : (b > LARS_HIST_BINS - 1 ? LARS_HIST_BINS - 1 : b);
    if (x < edges[b]) --b;
    else if (b != LARS_HIST_BINS - 1 && x >= edges[b + 1]) ++b;
    return b;
}

__device__ inline unsigned int cmap_index(float x)
{
    const float s = (x + 1.0f) * 128.0f;
    int i = (int)s;
    i = i < 0 ? 0 : (i > 255 ? 255 : i);
    return (unsigned)i;
}


}  // namespace lars
