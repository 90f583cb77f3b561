// Array-level kernels around the hot path: statistics and medians of index
// arrays, the 4-argument band-plane index, the float64 NDVI flavour, colormaps.
//
// Reference semantics (lars-uav/lars-image-processing):
//   process-images.py:506-512   analyze_index          (mean / median / min / max / coverage)
//   process-ndvi.py:60-71       analyze_ndvi_statistics (+ population std)
//   backend-process.py:28-38    calculate_index(red, green, nir, index_type)
//   process-ndvi.py:18-31       float64 NDVI
//   process-images.py:695       imshow(cmap, vmin=-1, vmax=1)
#include "common.h"
#include "device_common.h"

namespace lars {

// ===========================================================================
// Statistics of a float32 / float64 array
// ===========================================================================
// Deterministic: every block writes one partial record, a single block folds
// them in block order.
struct ArrPartial {
    double sum, sumsq, mn, mx;
    unsigned long long above, nans, count;
    unsigned long long pad;
};

template <typename T>
__global__ __launch_bounds__(256) void k_array_stats(const T *__restrict__ x, long long n, T thr, int want_hist,
                                                     ArrPartial *__restrict__ partials,
                                                     unsigned long long *__restrict__ ghist)
{
    __shared__ unsigned int s_hist[LARS_HIST_BINS];
    __shared__ HistCell<T> s_edges[LARS_HIST_CELLS];
    __shared__ ArrPartial s_part[4];
    const int tid = threadIdx.x;
    if (want_hist) {
        if (tid < LARS_HIST_BINS) s_hist[tid] = 0;
        hist_cells_init<T>(s_edges, tid);
        __syncthreads();
    }
    double sum = 0, sumsq = 0;
    double mn = __builtin_inf(), mx = -__builtin_inf();
    unsigned long long above = 0, nans = 0, count = 0;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n; i += (long long)gridDim.x * 256) {
        const T v = x[i];
        if (v != v) { ++nans; continue; }
        const double d = (double)v;
        sum += d; sumsq += d * d;
        mn = fmin(mn, d); mx = fmax(mx, d);
        above += (v > thr) ? 1 : 0;
        ++count;
        if (want_hist && v >= (T)-1 && v <= (T)1) atomicAdd(&s_hist[hist_bin_cell<T>(v, s_edges)], 1u);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        sum += __shfl_xor(sum, off); sumsq += __shfl_xor(sumsq, off);
        mn = fmin(mn, __shfl_xor(mn, off)); mx = fmax(mx, __shfl_xor(mx, off));
        above += __shfl_xor(above, off); nans += __shfl_xor(nans, off); count += __shfl_xor(count, off);
    }
    if ((tid & 63) == 0) {
        ArrPartial p; p.sum = sum; p.sumsq = sumsq; p.mn = mn; p.mx = mx; p.above = above; p.nans = nans; p.count = count; p.pad = 0;
        s_part[tid >> 6] = p;
    }
    __syncthreads();
    if (tid == 0) {
        ArrPartial p = s_part[0];
        for (int w = 1; w < 4; ++w) {
            p.sum += s_part[w].sum; p.sumsq += s_part[w].sumsq;
            p.mn = fmin(p.mn, s_part[w].mn); p.mx = fmax(p.mx, s_part[w].mx);
            p.above += s_part[w].above; p.nans += s_part[w].nans; p.count += s_part[w].count;
        }
        partials[blockIdx.x] = p;
    }
    if (want_hist && tid < LARS_HIST_BINS && s_hist[tid]) atomicAdd(&ghist[tid], (unsigned long long)s_hist[tid]);
}

// one block of 256 threads folds the per-block partials (a fixed tree: the double sums do not depend on timing)
__global__ __launch_bounds__(256) void k_array_stats_fold(const ArrPartial *__restrict__ partials, int nblocks, double thr,
                                                          const unsigned long long *__restrict__ ghist, int want_hist, lars_stats *out)
{
    __shared__ ArrPartial s_p[4];
    const int tid = threadIdx.x;
    ArrPartial p;
    p.sum = 0; p.sumsq = 0; p.mn = __builtin_inf(); p.mx = -__builtin_inf(); p.above = 0; p.nans = 0; p.count = 0; p.pad = 0;
    for (int b = tid; b < nblocks; b += 256) {
        const ArrPartial q = partials[b];
        p.sum += q.sum; p.sumsq += q.sumsq;
        p.mn = fmin(p.mn, q.mn); p.mx = fmax(p.mx, q.mx);
        p.above += q.above; p.nans += q.nans; p.count += q.count;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        p.sum += __shfl_xor(p.sum, off); p.sumsq += __shfl_xor(p.sumsq, off);
        p.mn = fmin(p.mn, __shfl_xor(p.mn, off)); p.mx = fmax(p.mx, __shfl_xor(p.mx, off));
        p.above += __shfl_xor(p.above, off); p.nans += __shfl_xor(p.nans, off); p.count += __shfl_xor(p.count, off);
    }
    if ((tid & 63) == 0) s_p[tid >> 6] = p;
    __syncthreads();
    if (tid == 0) {
        p = s_p[0];
        for (int w = 1; w < 4; ++w) {
            p.sum += s_p[w].sum; p.sumsq += s_p[w].sumsq;
            p.mn = fmin(p.mn, s_p[w].mn); p.mx = fmax(p.mx, s_p[w].mx);
            p.above += s_p[w].above; p.nans += s_p[w].nans; p.count += s_p[w].count;
        }
        out->sum = p.sum; out->sumsq = p.sumsq; out->count = p.count + p.nans; out->above = p.above; out->nans = p.nans;
        out->min = p.mn; out->max = p.mx; out->threshold = thr; out->index_id = 0xFFFFFFFFu; out->reserved = 0;
    }
    if (tid < LARS_HIST_BINS) out->hist[tid] = want_hist ? ghist[tid] : 0ull;
}

// sum of squared deviations from the mean (np.std's second pass, process-ndvi.py:65)
template <typename T>
__global__ __launch_bounds__(256) void k_sumsqdev(const T *__restrict__ x, long long n, const lars_stats *st,
                                                  double *__restrict__ partials)
{
    __shared__ double s_w[4];
    const double mean = st->sum / (double)st->count;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double d = (double)x[i] - mean;
        acc += d * d;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void k_fold_f64(const double *partials, int nblocks, double *out)
{
    __shared__ double s_w[4];
    double acc = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partials[b];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ===========================================================================
// np.median: radix select of the two middle order statistics
// ===========================================================================
#define SEL_BITS 11
#define SEL_BINS (1 << SEL_BITS)
struct SelectState {
    unsigned long long prefix;      // key bits decided so far
    unsigned long long k;           // rank still to find inside the current prefix
    unsigned long long k0;          // the requested rank ((n-1)/2 for a median)
    unsigned long long k1;          // the second requested rank: k0 or k0 + 1 (n/2 for a median)
    unsigned long long n;
    unsigned long long c_le;        // # keys <= selected key
    unsigned long long next_key;    // smallest key > selected key
    unsigned int hist[SEL_BINS];
};

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    typedef unsigned int type;
    static constexpr int BITS = 32;
    __device__ static inline unsigned long long key(float x) { return f32_key(x); }
    __device__ static inline float val(unsigned long long k) { return key_f32((unsigned int)k); }
};
template <> struct KeyOf<double> {
    typedef unsigned long long type;
    static constexpr int BITS = 64;
    __device__ static inline unsigned long long key(double x) { return f64_key(x); }
    __device__ static inline double val(unsigned long long k) { return key_f64(k); }
};

// All select kernels are batched: blockIdx.y = item (one array of n values, e.g. one tile's plane),
// st[item] its state, x + item * stride its data.
__global__ void k_sel_init(SelectState *st, long long n, long long k0, long long k1)
{
    SelectState *s = st + blockIdx.y;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid == 0) {
        s->prefix = 0; s->k0 = (unsigned long long)k0; s->k1 = (unsigned long long)k1; s->k = s->k0; s->n = (unsigned long long)n;
        s->c_le = 0; s->next_key = ~0ull;
    }
    if (tid < SEL_BINS) s->hist[tid] = 0;
}

// histogram of digit [shift, shift+bits) over keys that match the prefix above it
template <typename T>
__global__ __launch_bounds__(256) void k_sel_hist(const T *__restrict__ x, long long n, long long stride, SelectState *st,
                                                  int shift, int bits)
{
    __shared__ unsigned int s_h[SEL_BINS];
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) s_h[i] = 0;
    __syncthreads();
    SelectState *s = st + blockIdx.y;
    x += (long long)blockIdx.y * stride;
    const unsigned long long prefix = s->prefix;
    const int hi = shift + bits;                                   // bits above the digit
    const bool top = hi >= KeyOf<T>::BITS;
    const unsigned long long dmask = (1ull << bits) - 1ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned long long key = KeyOf<T>::key(x[i]);
        if (top || (key >> hi) == (prefix >> hi)) atomicAdd(&s_h[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += 256)
        if (s_h[i]) atomicAdd(&s->hist[i], s_h[i]);
}

__global__ __launch_bounds__(256) void k_sel_pick(SelectState *st, int shift, int bits)
{
    // one block per item: find the digit whose cumulative count covers rank k
    __shared__ unsigned long long s_cum[256];
    SelectState *s = st + blockIdx.y;
    const int tid = threadIdx.x;
    const int nb = 1 << bits;
    const int per = (nb + 255) / 256;
    unsigned long long local = 0;
    for (int j = 0; j < per; ++j) { const int b = tid * per + j; if (b < nb) local += s->hist[b]; }
    s_cum[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = tid >= off ? s_cum[tid - off] : 0;
        __syncthreads();
        s_cum[tid] += v;
        __syncthreads();
    }
    const unsigned long long k = s->k;
    unsigned long long cum = s_cum[tid] - local;
    __syncthreads();
    for (int j = 0; j < per; ++j) {
        const int b = tid * per + j;
        if (b < nb) {
            const unsigned long long c = s->hist[b];
            if (c && k >= cum && k < cum + c) {
                s->prefix |= ((unsigned long long)b) << shift;
                s->k = k - cum;
            }
            cum += c;
        }
    }
    __syncthreads();
    for (int i = tid; i < SEL_BINS; i += 256) s->hist[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_sel_next(const T *__restrict__ x, long long n, long long stride, SelectState *st)
{
    __shared__ unsigned long long s_c[4], s_m[4];
    SelectState *s = st + blockIdx.y;
    x += (long long)blockIdx.y * stride;
    const unsigned long long sel = s->prefix;
    unsigned long long c_le = 0, nxt = ~0ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned long long key = KeyOf<T>::key(x[i]);
        if (key <= sel) ++c_le;
        else nxt = key < nxt ? key : nxt;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        c_le += __shfl_xor(c_le, off);
        const unsigned long long o = __shfl_xor(nxt, off);
        nxt = o < nxt ? o : nxt;
    }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = c_le; s_m[threadIdx.x >> 6] = nxt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        unsigned long long m = s_m[0];
        for (int w = 1; w < 4; ++w) m = s_m[w] < m ? s_m[w] : m;
        atomicAdd(&s->c_le, c);
        atomicMin(&s->next_key, m);
    }
}

template <typename T>
__global__ void k_sel_finish(const SelectState *st, T *out, long long items)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < items) {
        const SelectState *s = st + i;
        const T v1 = KeyOf<T>::val(s->prefix);
        T v2 = v1;
        const unsigned long long k2 = s->k1;                      // k0 or k0 + 1
        if (k2 != s->k0 && k2 >= s->c_le) v2 = KeyOf<T>::val(s->next_key);
        out[2 * i] = v1;
        out[2 * i + 1] = v2;
    }
}

// ===========================================================================
// Elementwise kernels
// ===========================================================================
// backend-process.py:28-38: float32 planes, epsilon added in float32, np.clip.
__global__ __launch_bounds__(256) void k_index_planes(const float *__restrict__ red, const float *__restrict__ green,
                                                      const float *__restrict__ nir, long long n, int index_id,
                                                      float *__restrict__ out)
{
    const float *pa = index_id == LARS_NDWI ? green : nir;
    const float *pb = index_id == LARS_NDVI ? red : (index_id == LARS_GNDVI ? green : nir);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = pa[i], b = pb[i];
        float s = a + b;
        s = s + 1e-10f;
        float q = (a - b) / s;
        if (q == q) q = q < -1.0f ? -1.0f : (q > 1.0f ? 1.0f : q);    // np.clip keeps NaN
        out[i] = q;
    }
}

// process-ndvi.py:18-31: float64 NDVI of an interleaved image.
template <typename PIX>
__global__ __launch_bounds__(256) void k_ndvi_f64(const PIX *__restrict__ img, long long npix, int channels,
                                                  double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const PIX *p = img + i * channels;
        const double red = (double)p[0], nir = (double)p[2];
        double s = nir + red;
        s = s + 1e-10;
        double q = (nir - red) / s;
        q = q < -1.0 ? -1.0 : (q > 1.0 ? 1.0 : q);
        out[i] = q;
    }
}

// imshow(cmap, vmin=-1, vmax=1): Normalize in float32, index int(norm*256) with 256 -> 255.
__global__ __launch_bounds__(256) void k_colormap(const float *__restrict__ x, long long n,
                                                  const unsigned int *__restrict__ lut, unsigned int *__restrict__ out)
{
    __shared__ unsigned int s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        unsigned int o = 0u;                                           // NaN -> masked -> transparent black
        if (v == v) {
            const float s = (v + 1.0f) * 128.0f;
            int idx = (int)s;
            idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
            o = s_lut[idx];
        }
        out[i] = o;
    }
}

// The colormap ENTRY of every sample -- min(int((x + 1f) * 128f), 255), what Normalize(-1, 1) + Colormap.__call__ look up
// (process-images.py:690-695, backend-process.py:40-47) -- one byte per pixel: a palette image's pixels.
// Four samples per lane: one 16-byte load, one 4-byte store.  x must be 16-byte aligned, out 4-byte aligned.
__global__ __launch_bounds__(256) void k_colormap_entry(const float *__restrict__ x, long long n, uint8_t *__restrict__ out)
{
    const long long nvec = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        reinterpret_cast<unsigned int *>(out)[i] = cmap_index(v.x) | (cmap_index(v.y) << 8) | (cmap_index(v.z) << 16) | (cmap_index(v.w) << 24);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[nvec * 4 + threadIdx.x] = (uint8_t)cmap_index(x[nvec * 4 + threadIdx.x]);
}

}  // namespace lars

using namespace lars;

static int grid_for(long long n)
{
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}
// reductions that end in atomics or a per-block partial: fewer blocks with at least 16 elements per thread
static int grid_for_reduce(long long n)
{
    long long b = (n + 4095) / 4096;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
static int array_stats_impl(const T *x, int64_t n, T thr, int want_hist, lars_stats *out_dev, double *sumsqdev_dev,
                            void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || n <= 0 || !out_dev) return fail(LARS_ERR_INVALID, "lars_d_array_stats: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    const int nb = grid_for_reduce(n);
    const size_t need = (size_t)nb * sizeof(ArrPartial) + LARS_HIST_BINS * sizeof(unsigned long long) + (size_t)nb * sizeof(double);
    LARS_TRY(scratch_reserve(c, need));
    ArrPartial *parts = static_cast<ArrPartial *>(c->scratch);
    unsigned long long *ghist = reinterpret_cast<unsigned long long *>(parts + nb);
    double *dparts = reinterpret_cast<double *>(ghist + LARS_HIST_BINS);
    LARS_HIP_TRY(hipMemsetAsync(ghist, 0, LARS_HIST_BINS * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((k_array_stats<T>), dim3(nb), dim3(256), 0, s, x, (long long)n, thr, want_hist, parts, ghist);
    hipLaunchKernelGGL(k_array_stats_fold, dim3(1), dim3(256), 0, s, parts, nb, (double)thr, ghist, want_hist, out_dev);
    if (sumsqdev_dev) {
        hipLaunchKernelGGL((k_sumsqdev<T>), dim3(nb), dim3(256), 0, s, x, (long long)n, out_dev, dparts);
        hipLaunchKernelGGL(k_fold_f64, dim3(1), dim3(256), 0, s, dparts, nb, sumsqdev_dev);
    }
    return launch_check("lars_d_array_stats");
}

extern "C" int lars_d_array_stats_f32(const float *x, int64_t n, float threshold, int want_hist, lars_stats *out_dev,
                                      void *stream)
{
    return array_stats_impl<float>(x, n, threshold, want_hist, out_dev, nullptr, stream);
}
extern "C" int lars_d_array_stats_f64(const double *x, int64_t n, double threshold, int want_hist, lars_stats *out_dev,
                                      double *out_sumsqdev_dev, void *stream)
{
    return array_stats_impl<double>(x, n, threshold, want_hist, out_dev, out_sumsqdev_dev, stream);
}

extern "C" size_t lars_select_scratch_bytes(void) { return sizeof(SelectState); }

// the order statistics of ranks k0 and k1 (k1 = k0 or k0 + 1) of `items` arrays: out_dev[items][2]
template <typename T>
static int rank_pair_impl(const T *x, int64_t n, int64_t items, int64_t stride, int64_t k0, int64_t k1, T *out_dev, void *scratch,
                          void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || n <= 0 || items <= 0 || items > 65535 || !out_dev || !scratch || k0 < 0 || k1 < k0 || k1 > k0 + 1 || k1 >= n)
        return fail(LARS_ERR_INVALID, "rank select: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    SelectState *st = static_cast<SelectState *>(scratch);
    int nb = grid_for_reduce(n);
    if (items > 1) {                                      // enough blocks in total, not per item
        long long want = (4096 + items - 1) / items;
        if (nb > want) nb = (int)(want < 1 ? 1 : want);
    }
    const unsigned it = (unsigned)items;
    hipLaunchKernelGGL(k_sel_init, dim3(SEL_BINS / 256, it), dim3(256), 0, s, st, (long long)n, (long long)k0, (long long)k1);
    int hi = KeyOf<T>::BITS;
    while (hi > 0) {
        const int bits = hi >= SEL_BITS ? SEL_BITS : hi;
        const int shift = hi - bits;
        hipLaunchKernelGGL((k_sel_hist<T>), dim3(nb, it), dim3(256), 0, s, x, (long long)n, (long long)stride, st, shift, bits);
        hipLaunchKernelGGL(k_sel_pick, dim3(1, it), dim3(256), 0, s, st, shift, bits);
        hi = shift;
    }
    hipLaunchKernelGGL((k_sel_next<T>), dim3(nb, it), dim3(256), 0, s, x, (long long)n, (long long)stride, st);
    hipLaunchKernelGGL((k_sel_finish<T>), dim3((it + 63) / 64), dim3(64), 0, s, st, out_dev, (long long)items);
    return launch_check("rank select");
}
template <typename T>
static int median_pair_impl(const T *x, int64_t n, int64_t items, int64_t stride, T *out_dev, void *scratch, void *stream)
{
    if (n <= 0) return fail(LARS_ERR_INVALID, "lars_d_median_pair: bad arguments");
    return rank_pair_impl<T>(x, n, items, stride, (n - 1) / 2, n / 2, out_dev, scratch, stream);
}
namespace lars {
// np.percentile's two neighbours (ranks k0, min(k0 + 1, n - 1)) of `items` float32 arrays, for wb_generic.hip
int rank_pair_f32(const float *x, int64_t n, int64_t items, int64_t stride, int64_t k0, int64_t k1, float *out_dev, void *scratch,
                  void *stream)
{
    return rank_pair_impl<float>(x, n, items, stride, k0, k1, out_dev, scratch, stream);
}
}  // namespace lars
extern "C" int lars_d_median_pair_f32(const float *x, int64_t n, float *out_dev, void *scratch, void *stream)
{
    return median_pair_impl<float>(x, n, 1, n, out_dev, scratch, stream);
}
extern "C" int lars_d_median_pair_f64(const double *x, int64_t n, double *out_dev, void *scratch, void *stream)
{
    return median_pair_impl<double>(x, n, 1, n, out_dev, scratch, stream);
}
// items arrays of n float32 values each, `stride` values apart (planes of a batch): out_dev[items][2]
extern "C" int lars_d_median_pair_batch_f32(const float *x, int64_t n, int64_t items, int64_t stride, float *out_dev,
                                            void *scratch, void *stream)
{
    return median_pair_impl<float>(x, n, items, stride, out_dev, scratch, stream);
}

extern "C" int lars_d_index_planes_f32(const float *red, const float *green, const float *nir, int64_t n, int index_id,
                                       float *out, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!red || !green || !nir || !out || n <= 0 || index_id < 0 || index_id > 2)
        return fail(LARS_ERR_INVALID, "lars_d_index_planes_f32: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    hipLaunchKernelGGL(k_index_planes, dim3(grid_for(n)), dim3(256), 0, s, red, green, nir, (long long)n, index_id, out);
    return launch_check("lars_d_index_planes_f32");
}

extern "C" int lars_d_ndvi_f64(const void *img, int64_t npix, int channels, int dtype, double *out, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!img || !out || npix <= 0 || channels < 3) return fail(LARS_ERR_INVALID, "lars_d_ndvi_f64: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    if (dtype == LARS_U8)
        hipLaunchKernelGGL((k_ndvi_f64<uint8_t>), dim3(grid_for(npix)), dim3(256), 0, s, static_cast<const uint8_t *>(img),
                           (long long)npix, channels, out);
    else if (dtype == LARS_U16)
        hipLaunchKernelGGL((k_ndvi_f64<uint16_t>), dim3(grid_for(npix)), dim3(256), 0, s,
                           static_cast<const uint16_t *>(img), (long long)npix, channels, out);
    else
        return fail(LARS_ERR_INVALID, "lars_d_ndvi_f64: dtype");
    return launch_check("lars_d_ndvi_f64");
}

// classification mask: index > threshold in float32 (process-images.py:511, :657 -- the array np.mean averages)
__global__ __launch_bounds__(256) void k_threshold_mask(const float *__restrict__ x, long long n, float thr, uint8_t *__restrict__ out)
{
    const long long nvec = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        reinterpret_cast<unsigned int *>(out)[i] = (v.x > thr ? 1u : 0u) | (v.y > thr ? 0x100u : 0u) | (v.z > thr ? 0x10000u : 0u) |
                                                   (v.w > thr ? 0x1000000u : 0u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = nvec * 4 + threadIdx.x;
        out[i] = x[i] > thr ? 1 : 0;
    }
}

extern "C" int lars_d_threshold_mask_f32(const float *x, int64_t n, float threshold, uint8_t *out_mask, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !out_mask || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out_mask) & 3))
        return fail(LARS_ERR_INVALID, "lars_d_threshold_mask_f32: bad arguments (x 16-byte, out_mask 4-byte aligned)");
    hipLaunchKernelGGL(k_threshold_mask, dim3(grid_for((n + 3) / 4)), dim3(256), 0, pick_stream(c, stream), x, (long long)n, threshold,
                       out_mask);
    return launch_check("lars_d_threshold_mask_f32");
}

extern "C" int lars_d_colormap_f32(const float *x, int64_t n, const uint8_t *lut_rgba, uint8_t *out_rgba, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !lut_rgba || !out_rgba || n <= 0) return fail(LARS_ERR_INVALID, "lars_d_colormap_f32: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    hipLaunchKernelGGL(k_colormap, dim3(grid_for(n)), dim3(256), 0, s, x, (long long)n,
                       reinterpret_cast<const unsigned int *>(lut_rgba), reinterpret_cast<unsigned int *>(out_rgba));
    return launch_check("lars_d_colormap_f32");
}

extern "C" int lars_d_colormap_entry_f32(const float *x, int64_t n, uint8_t *out_entry, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !out_entry || n <= 0) return fail(LARS_ERR_INVALID, "lars_d_colormap_entry_f32: bad arguments");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out_entry) & 3))
        return fail(LARS_ERR_INVALID, "lars_d_colormap_entry_f32: x on a 16-byte, out_entry on a 4-byte boundary");
    hipStream_t s = pick_stream(c, stream);
    hipLaunchKernelGGL(k_colormap_entry, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, x, (long long)n, out_entry);
    return launch_check("lars_d_colormap_entry_f32");
}
