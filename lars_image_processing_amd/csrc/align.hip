// Registration and change detection around the hot path.
//
// Reference:
//   process-images.py:515-565  align_images: rgb2gray -> phase_cross_correlation -> ndimage.shift
//   process-images.py:885-989  create_change_detection_visualization: index(early), index(aligned late),
//                              diff = late - early, imshow(diff, cmap='bwr', vmin=-0.5, vmax=0.5)
//
// rgb2gray and phase_cross_correlation are scikit-image functions (not installed where this was
// written: their parity is anchored on known displacements, see oracle/align_oracle.py); the shift
// itself (scipy.ndimage.shift, order=1, mode='reflect', integer shift) and the colormap are bit-exact.
//
// The two 2-D FFTs and the inverse run in rocFFT through hipFFT, loaded on first use with dlopen
// (a plain library FFT: 1024 x 1024 complex128 at most, the images were down-scaled before).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace lars {

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// skimage.color.rgb2gray on uint8: x * (1/255) in float64, then the ITU-R 709 weights.
__global__ __launch_bounds__(256) void k_gray_c128(const uint8_t *__restrict__ img, long long npix, int channels,
                                                   double2 *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        double g;
        if (channels == 1) {
            g = (double)img[i];
        } else {
            const uint8_t *p = img + i * channels;
            const double k = 1.0 / 255.0;
            const double r = (double)p[0] * k, gg = (double)p[1] * k, b = (double)p[2] * k;
            g = r * 0.2125 + gg * 0.7154 + b * 0.0721;
        }
        out[i] = make_double2(g, 0.0);
    }
}

// cross-power spectrum, phase normalised: F * conj(M) / max(|F * conj(M)|, 100 eps); result in F
__global__ __launch_bounds__(256) void k_cross_power(double2 *__restrict__ f, const double2 *__restrict__ m, long long n)
{
    const double floor_mag = 100.0 * 2.220446049250313e-16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double2 a = f[i], b = m[i];
        const double re = a.x * b.x + a.y * b.y;
        const double im = a.y * b.x - a.x * b.y;
        const double mag = fmax(hypot(re, im), floor_mag);
        f[i] = make_double2(re / mag, im / mag);
    }
}

struct Peak {
    double value;
    long long index;
};
__device__ inline Peak better(Peak a, Peak b)          // np.argmax: the first of equal maxima
{
    if (b.value > a.value || (b.value == a.value && b.index < a.index)) return b;
    return a;
}
__device__ inline Peak block_peak(Peak p, Peak *s_p)
{
    for (int off = 32; off >= 1; off >>= 1) {
        Peak o;
        o.value = __shfl_xor(p.value, off);
        o.index = __shfl_xor(p.index, off);
        p = better(p, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_p[wave] = p;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) p = better(p, s_p[w]);
    return p;                                            // valid in thread 0
}
__global__ __launch_bounds__(256) void k_peak_stage1(const double2 *__restrict__ cc, long long n, Peak *__restrict__ partial)
{
    __shared__ Peak s_p[4];
    Peak p = {-1.0, 0x7fffffffffffffffll};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double2 v = cc[i];
        Peak q = {hypot(v.x, v.y), i};
        if (q.value != q.value) q.value = -1.0;          // NaN never wins (np.argmax would return it; the inputs are finite)
        p = better(p, q);
    }
    p = block_peak(p, s_p);
    if (threadIdx.x == 0) partial[blockIdx.x] = p;
}
// final fold + the wrap of skimage: shift > fix(size / 2) -> shift - size; an axis of length 1 -> 0
__global__ __launch_bounds__(256) void k_peak_stage2(const Peak *__restrict__ partial, int nparts, long long h, long long w,
                                                     long long *__restrict__ shift)
{
    __shared__ Peak s_p[4];
    Peak p = {-1.0, 0x7fffffffffffffffll};
    for (int i = threadIdx.x; i < nparts; i += 256) p = better(p, partial[i]);
    p = block_peak(p, s_p);
    if (threadIdx.x == 0) {
        long long py = p.index / w, px = p.index % w;
        if (py > h / 2) py -= h;
        if (px > w / 2) px -= w;
        if (h == 1) py = 0;
        if (w == 1) px = 0;
        shift[0] = py;
        shift[1] = px;
    }
}

// ndimage.shift(order=1, mode='reflect') with an integer shift: out[y][x] = in[R(y - dy)][R(x - dx)],
// R = half-sample symmetric reflection (d c b a | a b c d | d c b a)
__device__ inline long long reflect_index(long long i, long long n)
{
    const long long period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i >= n ? period - 1 - i : i;
}
__global__ __launch_bounds__(256) void k_shift_reflect_u8(const uint8_t *__restrict__ img, long long h, long long w, int channels,
                                                          const long long *__restrict__ shift, uint8_t *__restrict__ out)
{
    const long long dy = shift[0], dx = shift[1];
    const long long npix = h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long y = i / w, x = i - y * w;
        const long long sy = reflect_index(y - dy, h), sx = reflect_index(x - dx, w);
        const uint8_t *p = img + (sy * w + sx) * channels;
        uint8_t *o = out + i * channels;
        for (int c = 0; c < channels; ++c) o[c] = p[c];
    }
}

__global__ __launch_bounds__(256) void k_diff_f32(const float *__restrict__ early, const float *__restrict__ late, long long n,
                                                  float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = late[i] - early[i];                     // process-images.py:923
}

// cmap(Normalize(vmin, vmax)(x), bytes=True), matplotlib 3.10 colors.py: norm = (x - vmin) / (vmax - vmin) in
// float32; xa = norm * 256; xa == 256 -> 255; xa < 0 -> under (first colour); xa >= 256 -> over (last colour);
// NaN -> bad (transparent black)
__global__ __launch_bounds__(256) void k_colormap_norm(const float *__restrict__ x, long long n, float vmin, float span,
                                                       const unsigned int *__restrict__ lut, unsigned int *__restrict__ out)
{
    __shared__ unsigned int s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        unsigned int o = 0u;
        if (v == v) {
            const float xa = ((v - vmin) / span) * 256.0f;
            int idx;
            if (xa < 0.0f) idx = 0;
            else if (xa >= 256.0f) idx = 255;            // 256 itself folds to 255, above it the over colour = last colour
            else idx = (int)xa;
            o = s_lut[idx];
        }
        out[i] = o;
    }
}

// ---------------------------------------------------------------------------
// hipFFT, loaded lazily
// ---------------------------------------------------------------------------
namespace {

typedef void *fft_handle;      // hipfftHandle is an opaque pointer on ROCm
enum { FFT_Z2Z = 0x69, FFT_FORWARD = -1, FFT_BACKWARD = 1 };

struct HipFft {
    void *lib = nullptr;
    int (*Plan2d)(fft_handle *, int, int, int) = nullptr;
    int (*SetStream)(fft_handle, hipStream_t) = nullptr;
    int (*ExecZ2Z)(fft_handle, void *, void *, int) = nullptr;
    int (*Destroy)(fft_handle) = nullptr;
};
HipFft g_fft;

int load_hipfft()
{
    if (g_fft.lib) return LARS_OK;
    const char *env = getenv("LARS_HIPFFT_LIB");
    const char *names[] = {env, "libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"};
    void *lib = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) return fail(LARS_ERR_HIP, "cannot load libhipfft.so (set LARS_HIPFFT_LIB): %s", dlerror());
#define SYM(field, name)                                                                   \
    *(void **)(&g_fft.field) = dlsym(lib, name);                                           \
    if (!g_fft.field) { dlclose(lib); return fail(LARS_ERR_HIP, "libhipfft.so lacks %s", name); }
    SYM(Plan2d, "hipfftPlan2d")
    SYM(SetStream, "hipfftSetStream")
    SYM(ExecZ2Z, "hipfftExecZ2Z")
    SYM(Destroy, "hipfftDestroy")
#undef SYM
    g_fft.lib = lib;
    return LARS_OK;
}

// one cached plan per thread (the Streamlit sessions align pairs of equally sized images)
struct PlanCache {
    fft_handle plan = nullptr;
    long long h = 0, w = 0;
    bool valid = false;
};
thread_local PlanCache t_plan;

int get_plan(long long h, long long w, fft_handle *out)
{
    LARS_TRY(load_hipfft());
    if (t_plan.valid && t_plan.h == h && t_plan.w == w) { *out = t_plan.plan; return LARS_OK; }
    if (t_plan.valid) { g_fft.Destroy(t_plan.plan); t_plan.valid = false; }
    fft_handle p = nullptr;
    const int rc = g_fft.Plan2d(&p, (int)h, (int)w, FFT_Z2Z);
    if (rc != 0) return fail(LARS_ERR_HIP, "hipfftPlan2d(%lld, %lld) failed: %d", h, w, rc);
    t_plan.plan = p; t_plan.h = h; t_plan.w = w; t_plan.valid = true;
    *out = p;
    return LARS_OK;
}

int grid_for(long long n)
{
    long long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

#define PEAK_PARTS 1024

}  // namespace

void align_release()
{
    if (t_plan.valid && g_fft.Destroy) { g_fft.Destroy(t_plan.plan); t_plan.valid = false; }
}

}  // namespace lars

using namespace lars;

extern "C" {

int lars_d_gray_c128(const uint8_t *img, int64_t npix, int channels, double *out_c128, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!img || !out_c128 || npix <= 0 || (channels != 1 && channels != 3))
        return fail(LARS_ERR_INVALID, "lars_d_gray_c128: a [npix][3] (or [npix]) uint8 image is required");
    hipLaunchKernelGGL(k_gray_c128, dim3(grid_for(npix)), dim3(256), 0, pick_stream(c, stream), img, (long long)npix, channels,
                       reinterpret_cast<double2 *>(out_c128));
    return launch_check("lars_d_gray_c128");
}

size_t lars_phase_scratch_bytes(void) { return PEAK_PARTS * sizeof(Peak); }

int lars_d_phase_correlation(double *fixed_c128, double *moving_c128, int64_t h, int64_t w, int64_t *shift_dev, void *scratch,
                             void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!fixed_c128 || !moving_c128 || !shift_dev || !scratch || h <= 0 || w <= 0 || h > 32768 || w > 32768)
        return fail(LARS_ERR_INVALID, "lars_d_phase_correlation: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    fft_handle plan;
    LARS_TRY(get_plan(h, w, &plan));
    int rc = g_fft.SetStream(plan, s);
    if (rc == 0) rc = g_fft.ExecZ2Z(plan, fixed_c128, fixed_c128, FFT_FORWARD);
    if (rc == 0) rc = g_fft.ExecZ2Z(plan, moving_c128, moving_c128, FFT_FORWARD);
    if (rc != 0) return fail(LARS_ERR_HIP, "hipfftExecZ2Z (forward) failed: %d", rc);
    const long long n = (long long)h * w;
    hipLaunchKernelGGL(k_cross_power, dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<double2 *>(fixed_c128),
                       reinterpret_cast<const double2 *>(moving_c128), n);
    rc = g_fft.ExecZ2Z(plan, fixed_c128, fixed_c128, FFT_BACKWARD);
    if (rc != 0) return fail(LARS_ERR_HIP, "hipfftExecZ2Z (inverse) failed: %d", rc);
    Peak *parts = static_cast<Peak *>(scratch);
    int nparts = grid_for(n);
    if (nparts > PEAK_PARTS) nparts = PEAK_PARTS;
    hipLaunchKernelGGL(k_peak_stage1, dim3(nparts), dim3(256), 0, s, reinterpret_cast<const double2 *>(fixed_c128), n, parts);
    hipLaunchKernelGGL(k_peak_stage2, dim3(1), dim3(256), 0, s, parts, nparts, (long long)h, (long long)w,
                       reinterpret_cast<long long *>(shift_dev));
    return launch_check("lars_d_phase_correlation");
}

int lars_d_shift_reflect_u8(const uint8_t *img, int64_t h, int64_t w, int channels, const int64_t *shift_dev, uint8_t *out,
                            void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!img || !out || !shift_dev || h <= 0 || w <= 0 || channels < 1 || img == out)
        return fail(LARS_ERR_INVALID, "lars_d_shift_reflect_u8: bad arguments");
    hipLaunchKernelGGL(k_shift_reflect_u8, dim3(grid_for((long long)h * w)), dim3(256), 0, pick_stream(c, stream), img,
                       (long long)h, (long long)w, channels, reinterpret_cast<const long long *>(shift_dev), out);
    return launch_check("lars_d_shift_reflect_u8");
}

int lars_d_diff_f32(const float *early, const float *late, int64_t n, float *out_diff, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!early || !late || !out_diff || n <= 0) return fail(LARS_ERR_INVALID, "lars_d_diff_f32: bad arguments");
    hipLaunchKernelGGL(k_diff_f32, dim3(grid_for(n)), dim3(256), 0, pick_stream(c, stream), early, late, (long long)n, out_diff);
    return launch_check("lars_d_diff_f32");
}

int lars_d_colormap_norm_f32(const float *x, int64_t n, float vmin, float vmax, const uint8_t *lut_rgba, uint8_t *out_rgba,
                             void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !lut_rgba || !out_rgba || n <= 0 || !(vmax > vmin))
        return fail(LARS_ERR_INVALID, "lars_d_colormap_norm_f32: bad arguments (vmax must exceed vmin)");
    hipLaunchKernelGGL(k_colormap_norm, dim3(grid_for(n)), dim3(256), 0, pick_stream(c, stream), x, (long long)n, vmin,
                       vmax - vmin, reinterpret_cast<const unsigned int *>(lut_rgba), reinterpret_cast<unsigned int *>(out_rgba));
    return launch_check("lars_d_colormap_norm_f32");
}

}  // extern "C"
