// LANCZOS down-scale in front of the hot path.
//
// Reference: preprocess_large_image(img_array, max_dimension=1024), process-images.py:398-422:
// PIL.Image.fromarray(img).resize((new_w, new_h), Image.Resampling.LANCZOS).  Pillow's algorithm
// (src/libImaging/Resample.c, 8 bits per channel): float64 weights of the truncated sinc per output
// sample, normalised, rounded to 22-bit fixed point; horizontal integer pass into a uint8
// intermediate, then the vertical pass; accumulators start at 2^21, result clip8(acc >> 22).
// Four-channel images are RGBA to Pillow and are premultiplied by alpha before and divided after
// (Image.resize: RGBA -> RGBa -> resize -> RGBA).
//
// The weights are computed on the host with the same double arithmetic and libm sin() as Pillow
// (-ffp-contract=off); the two passes and the alpha handling run on the GPU in integers, so the
// result is bit-identical to Pillow's (tests/golden/resize_outputs.npz: outputs of the reference).
#include <cmath>
#include <vector>

#include "common.h"

namespace lars {

#define RS_PRECISION_BITS (32 - 8 - 2)

__device__ inline uint8_t rs_clip8(int acc)
{
    const int v = acc >> RS_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one pass along x: in [rows][w_in][C] -> out [rows][w_out][C]
template <int C>
__global__ __launch_bounds__(256) void k_resample_h(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int rows, int w_in,
                                                    int w_out, int ksize, const int *__restrict__ bounds,
                                                    const int *__restrict__ kk)
{
    const long long n = (long long)rows * w_out;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w_out), xx = (int)(i - (long long)y * w_out);
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int *k = kk + (long long)xx * ksize;
        const uint8_t *p = in + ((long long)y * w_in + xmin) * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) {
            const int kv = k[x];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)p[x * C + c] * kv;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[i * C + c] = rs_clip8(acc[c]);
    }
}

// one pass along y: in [h_in][w][C] -> out [h_out][w][C]
template <int C>
__global__ __launch_bounds__(256) void k_resample_v(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int h_in, int h_out,
                                                    int w, int ksize, const int *__restrict__ bounds, const int *__restrict__ kk)
{
    const long long n = (long long)h_out * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int yy = (int)(i / w), x = (int)(i - (long long)yy * w);
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int *k = kk + (long long)yy * ksize;
        const uint8_t *p = in + ((long long)ymin * w + x) * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) {
            const int kv = k[y];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)p[(long long)y * w * C + c] * kv;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[i * C + c] = rs_clip8(acc[c]);
    }
}

// Pillow Convert.c rgbA2rgba / rgba2rgbA
__global__ __launch_bounds__(256) void k_premultiply_rgba(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long npix)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const unsigned int px = reinterpret_cast<const unsigned int *>(in)[i];
        const unsigned int a = px >> 24;
        unsigned int o = px & 0xFF000000u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned int t = ((px >> (8 * c)) & 0xFFu) * a + 128u;
            o |= ((((t >> 8) + t) >> 8) & 0xFFu) << (8 * c);
        }
        reinterpret_cast<unsigned int *>(out)[i] = o;
    }
}
__global__ __launch_bounds__(256) void k_unpremultiply_rgba(uint8_t *__restrict__ img, long long npix)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const unsigned int px = reinterpret_cast<unsigned int *>(img)[i];
        const unsigned int a = px >> 24;
        if (a == 0u || a == 255u) continue;
        unsigned int o = px & 0xFF000000u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            unsigned int v = (255u * ((px >> (8 * c)) & 0xFFu)) / a;
            v = v > 255u ? 255u : v;
            o |= v << (8 * c);
        }
        reinterpret_cast<unsigned int *>(img)[i] = o;
    }
}

// Resample.c: lanczos_filter / precompute_coeffs / normalize_coeffs_8bpc for the box (0, in_size)
static double rs_sinc(double x)
{
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return std::sin(x) / x;
}
static double rs_lanczos(double x)
{
    if (-3.0 <= x && x < 3.0) return rs_sinc(x) * rs_sinc(x / 3);
    return 0.0;
}
static int rs_coeffs(int in_size, int out_size, std::vector<int> &bounds, std::vector<int> &kk)
{
    const float in0 = 0.0f, in1 = (float)in_size;
    const double scale = (double)(in1 - in0) / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k((size_t)ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = rs_lanczos((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (1 << RS_PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

template <int C>
static void launch_passes(hipStream_t s, const uint8_t *src, uint8_t *tmp, uint8_t *dst, int h, int w, int nh, int nw,
                          const int *bh, const int *kh, int ksh, const int *bv, const int *kv, int ksv)
{
    const uint8_t *cur = src;
    if (nw != w) {
        uint8_t *o = nh != h ? tmp : dst;
        const long long n = (long long)h * nw;
        hipLaunchKernelGGL((k_resample_h<C>), dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, s, cur, o, h,
                           w, nw, ksh, bh, kh);
        cur = o;
    }
    if (nh != h) {
        const long long n = (long long)nh * nw;
        hipLaunchKernelGGL((k_resample_v<C>), dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, s, cur, dst,
                           h, nh, nw, ksv, bv, kv);
    }
}

}  // namespace lars

using namespace lars;

// PIL.Image.resize((new_w, new_h), LANCZOS) of a host uint8 image [h][w][channels], channels 1, 3 or 4
// (4 = RGBA: premultiplied-alpha path) -- the arithmetic of preprocess_large_image, process-images.py:419-420.
extern "C" int lars_h_resize_lanczos_u8(const uint8_t *img, int64_t h, int64_t w, int channels, int64_t new_h, int64_t new_w,
                                        uint8_t *out)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!img || !out || h <= 0 || w <= 0 || new_h <= 0 || new_w <= 0 || h > (1 << 24) || w > (1 << 24))
        return fail(LARS_ERR_INVALID, "lars_h_resize_lanczos_u8: bad arguments");
    if (channels != 1 && channels != 3 && channels != 4)
        return fail(LARS_ERR_UNSUPPORTED, "lars_h_resize_lanczos_u8: 1, 3 or 4 channels (got %d)", channels);
    std::vector<int> bh, kh, bv, kv;
    const int ksh = rs_coeffs((int)w, (int)new_w, bh, kh);
    const int ksv = rs_coeffs((int)h, (int)new_h, bv, kv);
    const size_t in_bytes = (size_t)h * w * channels, tmp_bytes = (size_t)h * new_w * channels,
                 out_bytes = (size_t)new_h * new_w * channels;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t need = al(in_bytes) * 2 + al(tmp_bytes) + al(out_bytes) + al(bh.size() * 4) + al(kh.size() * 4) + al(bv.size() * 4) +
                        al(kv.size() * 4) + 1024;
    LARS_TRY(ws_reserve(c, need));
    char *p = static_cast<char *>(c->ws);
    uint8_t *d_in = reinterpret_cast<uint8_t *>(p); p += al(in_bytes);
    uint8_t *d_pre = reinterpret_cast<uint8_t *>(p); p += al(in_bytes);
    uint8_t *d_tmp = reinterpret_cast<uint8_t *>(p); p += al(tmp_bytes);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(p); p += al(out_bytes);
    int *d_bh = reinterpret_cast<int *>(p); p += al(bh.size() * 4);
    int *d_kh = reinterpret_cast<int *>(p); p += al(kh.size() * 4);
    int *d_bv = reinterpret_cast<int *>(p); p += al(bv.size() * 4);
    int *d_kv = reinterpret_cast<int *>(p);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(d_in, img, in_bytes, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(d_bh, bh.data(), bh.size() * 4, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(d_kh, kh.data(), kh.size() * 4, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(d_bv, bv.data(), bv.size() * 4, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(d_kv, kv.data(), kv.size() * 4, hipMemcpyHostToDevice, s));
    const uint8_t *src = d_in;
    if (channels == 4) {
        const long long npix = (long long)h * w;
        hipLaunchKernelGGL(k_premultiply_rgba, dim3((unsigned)((npix + 255) / 256 > 8192 ? 8192 : (npix + 255) / 256)), dim3(256), 0, s, d_in,
                           d_pre, npix);
        src = d_pre;
    }
    if (new_h == h && new_w == w) {
        LARS_HIP_TRY(hipMemcpyAsync(d_out, src, in_bytes, hipMemcpyDeviceToDevice, s));
    } else if (channels == 1) {
        launch_passes<1>(s, src, d_tmp, d_out, (int)h, (int)w, (int)new_h, (int)new_w, d_bh, d_kh, ksh, d_bv, d_kv, ksv);
    } else if (channels == 3) {
        launch_passes<3>(s, src, d_tmp, d_out, (int)h, (int)w, (int)new_h, (int)new_w, d_bh, d_kh, ksh, d_bv, d_kv, ksv);
    } else {
        launch_passes<4>(s, src, d_tmp, d_out, (int)h, (int)w, (int)new_h, (int)new_w, d_bh, d_kh, ksh, d_bv, d_kv, ksv);
    }
    if (channels == 4) {
        const long long npix = (long long)new_h * new_w;
        hipLaunchKernelGGL(k_unpremultiply_rgba, dim3((unsigned)((npix + 255) / 256 > 8192 ? 8192 : (npix + 255) / 256)), dim3(256), 0, s,
                           d_out, npix);
    }
    LARS_TRY(launch_check("lars_h_resize_lanczos_u8"));
    LARS_HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}
