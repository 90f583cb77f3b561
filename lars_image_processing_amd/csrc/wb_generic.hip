// White balance of images whose samples are neither uint8 nor uint16 (process-images.py:424-447 accepts whatever
// `astype(np.float32)` accepts, :431).  The host side has already applied that cast (it is the reference's own first
// line); this file does the rest on float32 samples:
//   np.percentile(channel, (2, 98))   exact order statistics by radix select (arrays.hip) + numpy's `_lerp`
//   clip((channel - p2) / (p98 - p2) * 255, 0, 255)   float64 arithmetic (float32 array (-) float64 scalar, NumPy 2)
//   stored into a float32 array, then .astype(uint8)  round to float32, truncate
// NaN samples: np.percentile returns NaN and the reference's output is whatever the platform's float -> uint8 cast makes
// of NaN; not reproduced here (documented "parity unpinned" in tests/test_gpu_parity.py).
#include "common.h"

namespace lars {

int rank_pair_f32(const float *x, int64_t n, int64_t items, int64_t stride, int64_t k0, int64_t k1, float *out_dev, void *scratch,
                  void *stream);

// interleaved [npix][C] -> three planes [3][npix]
__global__ __launch_bounds__(256) void k_wbg_split(const float *__restrict__ img, long long npix, int channels, float *__restrict__ planes)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const float *p = img + i * channels;
        planes[i] = p[0];
        planes[npix + i] = p[1];
        planes[2 * npix + i] = p[2];
    }
}

// numpy's 'linear' percentile from the two neighbouring order statistics.  For float32 data `_lerp` takes the difference
// b - a in float32 (both operands are float32 arrays) and everything after it in float64 (gamma is float64):
//   r = a + (b - a) * t ;  where t >= 0.5:  r = b - (b - a) * (1 - t)
// pairs[q][channel][2] (q = 0: 2nd percentile, 1: 98th) -> pcts[channel][2]
__global__ void k_wbg_percentiles(const float *__restrict__ pairs, long long n, double *__restrict__ pcts)
{
    const int t = threadIdx.x;
    if (t >= 6) return;
    const int q = t / 3, ch = t % 3;
    const double vi = (double)(n - 1) * ((q == 0 ? 2.0 : 98.0) / 100.0);
    const double g = vi - floor(vi);
    const float a = pairs[(q * 3 + ch) * 2], b = pairs[(q * 3 + ch) * 2 + 1];
    const float d32 = b - a;
    const double d = (double)d32;
    double r = (double)a + d * g;
    if (g >= 0.5) r = (double)b - d * (1.0 - g);
    pcts[ch * 2 + q] = r;
}

__global__ __launch_bounds__(256) void k_wbg_map(const float *__restrict__ img, long long npix, int channels,
                                                 const double *__restrict__ pcts, uint8_t *__restrict__ out)
{
    const double lo0 = pcts[0], sp0 = pcts[1] - pcts[0], lo1 = pcts[2], sp1 = pcts[3] - pcts[2], lo2 = pcts[4], sp2 = pcts[5] - pcts[4];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const float *p = img + i * channels;
        uint8_t *o = out + i * channels;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double lo = c == 0 ? lo0 : c == 1 ? lo1 : lo2, sp = c == 0 ? sp0 : c == 1 ? sp1 : sp2;
            double v = ((double)p[c] - lo) / sp * 255.0;            // IEEE: x/0 -> +-inf, 0/0 -> NaN, as NumPy
            unsigned int u = 0;                                      // NaN survives np.clip and casts to 0
            if (v == v) {
                v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
                u = (unsigned int)(float)v;                          // float32 store (:438), truncating cast (:441)
            }
            o[c] = (uint8_t)u;
        }
        for (int c = 3; c < channels; ++c) o[c] = 0;                 // zeros_like (:432)
    }
}

}  // namespace lars

using namespace lars;

extern "C" size_t lars_select_scratch_bytes(void);

// Host entry point: float32 [h][w][channels] in, uint8 [h][w][channels] out, optional percentiles[3][2].
extern "C" int lars_h_fix_white_balance_f32(const float *img, int64_t h, int64_t w, int channels, uint8_t *out, double *percentiles)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!img || !out || h <= 0 || w <= 0 || channels < 3) return fail(LARS_ERR_INVALID, "lars_h_fix_white_balance_f32: bad arguments");
    const long long npix = (long long)h * w;
    const size_t sel_bytes = (lars_select_scratch_bytes() + 255) & ~(size_t)255;
    const size_t img_bytes = ((size_t)npix * channels * 4 + 255) & ~(size_t)255;
    const size_t planes_bytes = ((size_t)npix * 3 * 4 + 255) & ~(size_t)255;
    const size_t out_bytes = ((size_t)npix * channels + 255) & ~(size_t)255;
    LARS_TRY(ws_reserve(c, img_bytes + planes_bytes + out_bytes + 3 * sel_bytes + 1024));
    char *base = static_cast<char *>(c->ws);
    float *d_img = reinterpret_cast<float *>(base);
    float *d_planes = reinterpret_cast<float *>(base + img_bytes);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(base + img_bytes + planes_bytes);
    char *d_sel = base + img_bytes + planes_bytes + out_bytes;
    float *d_pairs = reinterpret_cast<float *>(d_sel + 3 * sel_bytes);           // [2][3][2]
    double *d_pcts = reinterpret_cast<double *>(d_sel + 3 * sel_bytes + 256);     // [3][2]
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(d_img, img, (size_t)npix * channels * 4, hipMemcpyHostToDevice, s));
    long long nb = (npix + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_wbg_split, dim3((unsigned)nb), dim3(256), 0, s, d_img, npix, channels, d_planes);
    for (int q = 0; q < 2; ++q) {
        const double vi = (double)(npix - 1) * ((q == 0 ? 2.0 : 98.0) / 100.0);
        const long long k0 = (long long)floor(vi);
        const long long k1 = k0 + 1 > npix - 1 ? npix - 1 : k0 + 1;
        LARS_TRY(rank_pair_f32(d_planes, npix, 3, npix, k0, k1, d_pairs + q * 6, d_sel, s));
    }
    hipLaunchKernelGGL(k_wbg_percentiles, dim3(1), dim3(64), 0, s, d_pairs, npix, d_pcts);
    hipLaunchKernelGGL(k_wbg_map, dim3((unsigned)nb), dim3(256), 0, s, d_img, npix, channels, d_pcts, d_out);
    LARS_TRY(launch_check("lars_h_fix_white_balance_f32"));
    LARS_HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)npix * channels, hipMemcpyDeviceToHost, s));
    if (percentiles) LARS_HIP_TRY(hipMemcpyAsync(percentiles, d_pcts, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}
