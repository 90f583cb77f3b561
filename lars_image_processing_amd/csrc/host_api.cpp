// Host entry points (lars_h_*): host pointers in, host pointers out.  These are
// what the Python mirror of the reference's functions binds (one call per
// reference function).  Each call stages through the calling
// thread's grow-only device workspace and stream, so concurrent Streamlit
// sessions (threads) do not share mutable state.
#include <string.h>

#include "common.h"

using namespace lars;

namespace {

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base(static_cast<char *>(b)) {}
    template <typename T>
    T *take(size_t count)
    {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct ImageJob {
    const void *img;
    int64_t h, w;
    int channels, dtype;
    int apply_wb, wb_variant;
    uint32_t mask;
    int want_stats, want_hist, want_median;
    uint8_t *out_wb;
    float *out_index[3];
    lars_stats *stats;
    float *medians;               // [3][2]
    uint8_t *out_rgba[3];
    const uint8_t *cmap[3];
    double *pcts;                 // [3][2]
};

struct Layout {
    uint8_t *img; uint32_t *hist; uint8_t *table; double *pcts; uint8_t *wb;
    float *idx[3]; lars_stats *stats; float *med; char *sel; uint8_t *rgba[3]; uint8_t *cmap[3];
    uint8_t *entry[3];             // colormap entry planes (out_rgba[k] without a cmap_lut[k]): one byte per pixel
    float *pairs; char *selq;      // statistics + medians without planes (lars_d_stats_medians)
    size_t total;
};

// Medians of uint8 RGNir images come from the two-level select on recomputed values (select_q.hip): two passes over the
// 3-byte pixels instead of four or more over each 4-byte plane, whether or not the planes are written as well.
bool select_route(const ImageJob &j)
{
    if (!j.want_median || !j.medians || !j.mask || j.dtype != LARS_U8 || j.channels != 3) return false;
    return (long long)j.h * j.w * 6 < (1ll << 30);
}
// ... and when no plane is wanted (one index or all three), the statistics kernel counts the select's first pass itself:
// the planes never exist on the device either
bool recompute_route(const ImageJob &j)
{
    if (!select_route(j) || !j.want_stats) return false;
    if (j.mask != 1u && j.mask != 2u && j.mask != 4u && j.mask != 7u) return false;
    for (int k = 0; k < 3; ++k)
        if (j.out_index[k] || j.out_rgba[k]) return false;
    return true;
}

Layout plan(const ImageJob &j, void *base)
{
    Carver c(base);
    Layout L;
    const size_t npix = (size_t)j.h * j.w;
    const size_t esz = j.dtype == LARS_U8 ? 1 : 2;
    const size_t nval = j.dtype == LARS_U8 ? 256 : 65536;
    L.img = c.take<uint8_t>(npix * j.channels * esz);
    L.hist = nullptr;
    (void)nval;
    L.table = j.apply_wb ? c.take<uint8_t>(lars_wb_table_bytes(j.dtype)) : nullptr;
    L.pcts = j.apply_wb ? c.take<double>(6) : nullptr;
    L.wb = (j.apply_wb && j.out_wb) ? c.take<uint8_t>(npix * j.channels) : nullptr;
    const bool select = select_route(j);
    for (int k = 0; k < 3; ++k) {
        const bool on = (j.mask >> k) & 1u;
        const bool entries = on && j.out_rgba[k] && !j.cmap[k];             // the entry plane is derived from the float32 plane on the device
        L.idx[k] = (on && (j.out_index[k] || entries || (j.want_median && !select))) ? c.take<float>(npix) : nullptr;
        L.rgba[k] = (on && j.out_rgba[k] && j.cmap[k]) ? c.take<uint8_t>(npix * 4) : nullptr;
        L.cmap[k] = (on && j.out_rgba[k] && j.cmap[k]) ? c.take<uint8_t>(1024) : nullptr;
        L.entry[k] = entries ? c.take<uint8_t>(npix + 4) : nullptr;
    }
    L.stats = c.take<lars_stats>(3);
    L.med = c.take<float>(6);
    L.sel = c.take<char>(3 * ((lars_select_scratch_bytes() + 255) & ~(size_t)255));
    L.pairs = select ? c.take<float>(4) : nullptr;
    L.selq = select ? c.take<char>(lars_quotient_median_scratch_bytes(1)) : nullptr;
    L.total = c.off + 256;
    return L;
}

int run_image(const ImageJob &j)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!j.img || j.h <= 0 || j.w <= 0 || j.channels < 3)
        return fail(LARS_ERR_INVALID, "image must be a non-empty [h][w][>=3] array");
    if (j.dtype != LARS_U8 && j.dtype != LARS_U16) return fail(LARS_ERR_INVALID, "dtype must be LARS_U8 or LARS_U16");
    if (j.mask & ~LARS_MASK_ALL) return fail(LARS_ERR_INVALID, "unknown index bits in mask");
    const size_t npix = (size_t)j.h * j.w;
    const size_t esz = j.dtype == LARS_U8 ? 1 : 2;
    Layout L = plan(j, nullptr);
    LARS_TRY(ws_reserve(c, L.total));
    L = plan(j, c->ws);
    hipStream_t s = c->stream;

    LARS_HIP_TRY(hipMemcpyAsync(L.img, j.img, npix * j.channels * esz, hipMemcpyHostToDevice, s));
    if (j.apply_wb) {
        LARS_TRY(lars_d_wb_prepare(L.img, 1, (int64_t)npix, j.channels, j.dtype, L.table, L.pcts, j.wb_variant, s));
    }
    const bool stats = j.want_stats && j.mask;
    if (j.mask || L.wb) {
        lars_fused_args a;
        memset(&a, 0, sizeof a);
        a.tiles = L.img; a.ntiles = 1; a.npix = (int64_t)npix; a.channels = j.channels; a.dtype = j.dtype;
        a.wb_table = j.apply_wb ? L.table : nullptr;
        a.index_mask = j.mask;
        a.flags = stats ? (LARS_F_STATS | (j.want_hist ? LARS_F_HIST : 0u)) : 0u;
        for (int k = 0; k < 3; ++k) {
            a.out_index[k] = L.idx[k];
            a.out_rgba[k] = L.rgba[k];
            a.cmap_lut[k] = L.cmap[k];
            if (L.cmap[k]) LARS_HIP_TRY(hipMemcpyAsync(L.cmap[k], j.cmap[k], 1024, hipMemcpyHostToDevice, s));
        }
        a.out_wb = L.wb;
        a.stats = stats ? L.stats : nullptr;
        a.stream = s;
        if (recompute_route(j)) {
            if (L.wb) {                                     // the white-balanced image on its own, then statistics + medians
                lars_fused_args w = a;
                w.index_mask = 0; w.flags = 0; w.stats = nullptr;
                LARS_TRY(lars_d_fused(&w));
                a.out_wb = nullptr;
            }
            LARS_TRY(lars_d_stats_medians(&a, L.pairs, L.selq));
        } else {
            LARS_TRY(lars_d_fused(&a));
            if (L.pairs)
                LARS_TRY(lars_d_quotient_median_pairs(L.img, 1, (int64_t)npix, 3, LARS_U8, a.wb_table,
                                                      ((j.mask & 1u) ? 1u : 0u) | ((j.mask & 6u) ? 2u : 0u), L.pairs, L.selq, s));
        }
    }
    const size_t selsz = (lars_select_scratch_bytes() + 255) & ~(size_t)255;
    for (int k = 0; k < 3; ++k) {
        if (!((j.mask >> k) & 1u) || L.pairs) continue;
        if (j.want_median && j.medians)
            LARS_TRY(lars_d_median_pair_f32(L.idx[k], (int64_t)npix, L.med + 2 * k, L.sel + k * selsz, s));
    }
    // results back
    if (L.wb) LARS_HIP_TRY(hipMemcpyAsync(j.out_wb, L.wb, npix * j.channels, hipMemcpyDeviceToHost, s));
    for (int k = 0; k < 3; ++k) {
        if (!((j.mask >> k) & 1u)) continue;
        if (j.out_index[k]) LARS_HIP_TRY(hipMemcpyAsync(j.out_index[k], L.idx[k], npix * 4, hipMemcpyDeviceToHost, s));
        if (L.rgba[k]) LARS_HIP_TRY(hipMemcpyAsync(j.out_rgba[k], L.rgba[k], npix * 4, hipMemcpyDeviceToHost, s));
        if (L.entry[k]) {                                   // one byte per pixel crosses PCIe instead of four (or a float32 plane)
            LARS_TRY(lars_d_colormap_entry_f32(L.idx[k], (int64_t)npix, L.entry[k], s));
            LARS_HIP_TRY(hipMemcpyAsync(j.out_rgba[k], L.entry[k], npix, hipMemcpyDeviceToHost, s));
        }
    }
    lars_stats hstats[3];
    float hmed[6];
    double hp[6];
    if (stats) LARS_HIP_TRY(hipMemcpyAsync(hstats, L.stats, sizeof hstats, hipMemcpyDeviceToHost, s));
    float hpairs[4] = {0, 0, 0, 0};
    if (L.pairs) LARS_HIP_TRY(hipMemcpyAsync(hpairs, L.pairs, sizeof hpairs, hipMemcpyDeviceToHost, s));
    else if (j.want_median && j.medians && j.mask) LARS_HIP_TRY(hipMemcpyAsync(hmed, L.med, sizeof hmed, hipMemcpyDeviceToHost, s));
    if (j.apply_wb && j.pcts) LARS_HIP_TRY(hipMemcpyAsync(hp, L.pcts, sizeof hp, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    if (L.pairs) {
        // {NDVI middles, GNDVI middles}; NDWI = -GNDVI: its two middles are the negated GNDVI middles in reverse order
        hmed[0] = hpairs[0]; hmed[1] = hpairs[1]; hmed[2] = hpairs[2]; hmed[3] = hpairs[3];
        hmed[4] = 0.0f - hpairs[3]; hmed[5] = 0.0f - hpairs[2];
        for (int k = 0; k < 3; ++k)
            if (((j.mask >> k) & 1u) && (hmed[2 * k] != hmed[2 * k] || hmed[2 * k + 1] != hmed[2 * k + 1]))
                return fail(LARS_ERR_HIP, "exact median select did not settle");
    }
    for (int k = 0; k < 3; ++k) {
        if (!((j.mask >> k) & 1u)) continue;
        if (stats && j.stats) j.stats[k] = hstats[k];
        if (j.want_median && j.medians) { j.medians[2 * k] = hmed[2 * k]; j.medians[2 * k + 1] = hmed[2 * k + 1]; }
    }
    if (j.apply_wb && j.pcts) memcpy(j.pcts, hp, sizeof hp);
    return LARS_OK;
}

}  // namespace

template <typename T>
static int analyze_impl(const T *x, int64_t n, T thr, int want_hist, lars_stats *out, T *median_pair, double *sumsqdev)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || n <= 0 || !out) return fail(LARS_ERR_INVALID, "lars_h_analyze: bad arguments");
    Carver cv(nullptr);
    cv.take<T>(n); cv.take<lars_stats>(1); cv.take<T>(2); cv.take<double>(1); cv.take<char>(lars_select_scratch_bytes());
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    T *dx = d.take<T>(n);
    lars_stats *dst = d.take<lars_stats>(1);
    T *dmed = d.take<T>(2);
    double *dss = d.take<double>(1);
    char *dsel = d.take<char>(lars_select_scratch_bytes());
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * sizeof(T), hipMemcpyHostToDevice, s));
    if (sizeof(T) == 4) {
        LARS_TRY(lars_d_array_stats_f32(reinterpret_cast<const float *>(dx), n, (float)thr, want_hist, dst, s));
        if (median_pair)
            LARS_TRY(lars_d_median_pair_f32(reinterpret_cast<const float *>(dx), n, reinterpret_cast<float *>(dmed), dsel, s));
    } else {
        LARS_TRY(lars_d_array_stats_f64(reinterpret_cast<const double *>(dx), n, (double)thr, want_hist, dst,
                                        sumsqdev ? dss : nullptr, s));
        if (median_pair)
            LARS_TRY(lars_d_median_pair_f64(reinterpret_cast<const double *>(dx), n, reinterpret_cast<double *>(dmed), dsel, s));
    }
    LARS_HIP_TRY(hipMemcpyAsync(out, dst, sizeof(lars_stats), hipMemcpyDeviceToHost, s));
    if (median_pair) LARS_HIP_TRY(hipMemcpyAsync(median_pair, dmed, 2 * sizeof(T), hipMemcpyDeviceToHost, s));
    if (sumsqdev) LARS_HIP_TRY(hipMemcpyAsync(sumsqdev, dss, sizeof(double), hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

extern "C" {

int lars_h_fix_white_balance(const void *img, int64_t h, int64_t w, int channels, int dtype, int variant, uint8_t *out,
                             double *percentiles)
{
    if (!out) return fail(LARS_ERR_INVALID, "lars_h_fix_white_balance: out == NULL");
    ImageJob j;
    memset(&j, 0, sizeof j);
    j.img = img; j.h = h; j.w = w; j.channels = channels; j.dtype = dtype;
    j.apply_wb = 1; j.wb_variant = variant; j.mask = 0; j.out_wb = out; j.pcts = percentiles;
    return run_image(j);
}

int lars_h_calculate_index(const void *img, int64_t h, int64_t w, int channels, int dtype, uint32_t index_mask,
                           float *const out[3], lars_stats *stats, int want_hist)
{
    if (!index_mask) return fail(LARS_ERR_INVALID, "lars_h_calculate_index: empty index mask");
    ImageJob j;
    memset(&j, 0, sizeof j);
    j.img = img; j.h = h; j.w = w; j.channels = channels; j.dtype = dtype;
    j.mask = index_mask;
    for (int k = 0; k < 3; ++k) j.out_index[k] = out ? out[k] : nullptr;
    j.stats = stats; j.want_stats = stats != nullptr; j.want_hist = want_hist;
    return run_image(j);
}

int lars_h_process_image(const void *img, int64_t h, int64_t w, int channels, int dtype, int apply_wb,
                         uint32_t index_mask, int want_hist, uint8_t *out_wb, float *const out_index[3],
                         lars_stats *stats, float *medians, uint8_t *const out_rgba[3], const uint8_t *const cmap_lut[3])
{
    ImageJob j;
    memset(&j, 0, sizeof j);
    j.img = img; j.h = h; j.w = w; j.channels = channels; j.dtype = dtype;
    j.apply_wb = apply_wb; j.mask = index_mask; j.out_wb = out_wb;
    for (int k = 0; k < 3; ++k) {
        j.out_index[k] = out_index ? out_index[k] : nullptr;
        j.out_rgba[k] = out_rgba ? out_rgba[k] : nullptr;
        j.cmap[k] = cmap_lut ? cmap_lut[k] : nullptr;
    }
    j.stats = stats; j.want_stats = stats != nullptr; j.want_hist = want_hist;
    j.medians = medians; j.want_median = medians != nullptr;
    if (!index_mask && !(apply_wb && out_wb)) return fail(LARS_ERR_INVALID, "lars_h_process_image: nothing requested");
    return run_image(j);
}

int lars_h_calculate_index_planes(const float *red, const float *green, const float *nir, int64_t n, int index_id,
                                  float *out)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!red || !green || !nir || !out || n <= 0) return fail(LARS_ERR_INVALID, "lars_h_calculate_index_planes: bad arguments");
    Carver cv(nullptr);
    cv.take<float>(n); cv.take<float>(n); cv.take<float>(n); cv.take<float>(n);
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    float *dr = d.take<float>(n), *dg = d.take<float>(n), *dn = d.take<float>(n), *dout = d.take<float>(n);
    hipStream_t s = c->stream;
    // only the two bands the index reads cross PCIe
    const float *ha = index_id == LARS_NDWI ? green : nir;
    const float *hb = index_id == LARS_NDVI ? red : (index_id == LARS_GNDVI ? green : nir);
    float *da = index_id == LARS_NDWI ? dg : dn;
    float *db = index_id == LARS_NDVI ? dr : (index_id == LARS_GNDVI ? dg : dn);
    LARS_HIP_TRY(hipMemcpyAsync(da, ha, (size_t)n * 4, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(db, hb, (size_t)n * 4, hipMemcpyHostToDevice, s));
    LARS_TRY(lars_d_index_planes_f32(dr, dg, dn, n, index_id, dout, s));
    LARS_HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

int lars_h_ndvi_f64(const void *img, int64_t h, int64_t w, int channels, int dtype, double *out)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!img || !out || h <= 0 || w <= 0 || channels < 3) return fail(LARS_ERR_INVALID, "lars_h_ndvi_f64: bad arguments");
    if (dtype != LARS_U8 && dtype != LARS_U16) return fail(LARS_ERR_INVALID, "lars_h_ndvi_f64: dtype");
    const size_t npix = (size_t)h * w, esz = dtype == LARS_U8 ? 1 : 2;
    Carver cv(nullptr);
    cv.take<uint8_t>(npix * channels * esz); cv.take<double>(npix);
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    uint8_t *dimg = d.take<uint8_t>(npix * channels * esz);
    double *dout = d.take<double>(npix);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(dimg, img, npix * channels * esz, hipMemcpyHostToDevice, s));
    LARS_TRY(lars_d_ndvi_f64(dimg, (int64_t)npix, channels, dtype, dout, s));
    LARS_HIP_TRY(hipMemcpyAsync(out, dout, npix * 8, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

int lars_h_analyze_f32(const float *x, int64_t n, float threshold, int want_hist, lars_stats *out, float median_pair[2])
{
    return analyze_impl<float>(x, n, threshold, want_hist, out, median_pair, nullptr);
}
int lars_h_analyze_f64(const double *x, int64_t n, double threshold, int want_hist, lars_stats *out, double median_pair[2],
                       double *sumsqdev)
{
    return analyze_impl<double>(x, n, threshold, want_hist, out, median_pair, sumsqdev);
}

int lars_h_threshold_mask_f32(const float *x, int64_t n, float threshold, uint8_t *out_mask)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !out_mask || n <= 0) return fail(LARS_ERR_INVALID, "lars_h_threshold_mask_f32: bad arguments");
    Carver cv(nullptr);
    cv.take<float>(n); cv.take<uint8_t>((size_t)n + 4);
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    float *dx = d.take<float>(n);
    uint8_t *dm = d.take<uint8_t>((size_t)n + 4);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    LARS_TRY(lars_d_threshold_mask_f32(dx, n, threshold, dm, s));
    LARS_HIP_TRY(hipMemcpyAsync(out_mask, dm, (size_t)n, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

int lars_h_colormap_f32(const float *x, int64_t n, const uint8_t *lut_rgba, uint8_t *out_rgba)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !lut_rgba || !out_rgba || n <= 0) return fail(LARS_ERR_INVALID, "lars_h_colormap_f32: bad arguments");
    Carver cv(nullptr);
    cv.take<float>(n); cv.take<uint8_t>(1024); cv.take<uint8_t>((size_t)n * 4);
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    float *dx = d.take<float>(n);
    uint8_t *dl = d.take<uint8_t>(1024);
    uint8_t *dout = d.take<uint8_t>((size_t)n * 4);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(dl, lut_rgba, 1024, hipMemcpyHostToDevice, s));
    LARS_TRY(lars_d_colormap_f32(dx, n, dl, dout, s));
    LARS_HIP_TRY(hipMemcpyAsync(out_rgba, dout, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

int lars_h_colormap_norm_f32(const float *x, int64_t n, float vmin, float vmax, const uint8_t *lut_rgba, uint8_t *out_rgba)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!x || !lut_rgba || !out_rgba || n <= 0) return fail(LARS_ERR_INVALID, "lars_h_colormap_norm_f32: bad arguments");
    Carver cv(nullptr);
    cv.take<float>(n); cv.take<uint8_t>(1024); cv.take<uint8_t>((size_t)n * 4);
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    float *dx = d.take<float>(n);
    uint8_t *dl = d.take<uint8_t>(1024);
    uint8_t *dout = d.take<uint8_t>((size_t)n * 4);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(dl, lut_rgba, 1024, hipMemcpyHostToDevice, s));
    LARS_TRY(lars_d_colormap_norm_f32(dx, n, vmin, vmax, dl, dout, s));
    LARS_HIP_TRY(hipMemcpyAsync(out_rgba, dout, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    return LARS_OK;
}

// device-side registration of `moving` (device, uint8 [h][w][channels]) to `fixed`: estimate + apply
static int align_on_device(ThreadCtx *c, const uint8_t *d_fixed, const uint8_t *d_moving, int64_t h, int64_t w, int channels,
                           double *d_fa, double *d_fb, void *d_scratch, int64_t *d_shift, uint8_t *d_aligned, hipStream_t s)
{
    const int64_t npix = h * w;
    LARS_TRY(lars_d_gray_c128(d_fixed, npix, channels, d_fa, s));
    LARS_TRY(lars_d_gray_c128(d_moving, npix, channels, d_fb, s));
    LARS_TRY(lars_d_phase_correlation(d_fa, d_fb, h, w, d_shift, d_scratch, s));
    LARS_TRY(lars_d_shift_reflect_u8(d_moving, h, w, channels, d_shift, d_aligned, s));
    return LARS_OK;
}

int lars_h_align_images(const uint8_t *fixed, const uint8_t *moving, int64_t h, int64_t w, int channels, uint8_t *out_aligned,
                        double shift[2])
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!fixed || !moving || !out_aligned || h <= 0 || w <= 0 || (channels != 1 && channels != 3))
        return fail(LARS_ERR_INVALID, "lars_h_align_images: two uint8 [h][w][3] (or [h][w]) images of the same shape are required");
    const size_t npix = (size_t)h * w, nbytes = npix * channels;
    Carver cv(nullptr);
    cv.take<uint8_t>(nbytes); cv.take<uint8_t>(nbytes); cv.take<uint8_t>(nbytes);
    cv.take<double>(npix * 2); cv.take<double>(npix * 2); cv.take<char>(lars_phase_scratch_bytes()); cv.take<int64_t>(2);
    LARS_TRY(ws_reserve(c, cv.off + 256));
    Carver d(c->ws);
    uint8_t *df = d.take<uint8_t>(nbytes), *dm = d.take<uint8_t>(nbytes), *da = d.take<uint8_t>(nbytes);
    double *fa = d.take<double>(npix * 2), *fb = d.take<double>(npix * 2);
    char *sc = d.take<char>(lars_phase_scratch_bytes());
    int64_t *dshift = d.take<int64_t>(2);
    hipStream_t s = c->stream;
    LARS_HIP_TRY(hipMemcpyAsync(df, fixed, nbytes, hipMemcpyHostToDevice, s));
    LARS_HIP_TRY(hipMemcpyAsync(dm, moving, nbytes, hipMemcpyHostToDevice, s));
    LARS_TRY(align_on_device(c, df, dm, h, w, channels, fa, fb, sc, dshift, da, s));
    int64_t hs[2] = {0, 0};
    LARS_HIP_TRY(hipMemcpyAsync(out_aligned, da, nbytes, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipMemcpyAsync(hs, dshift, sizeof hs, hipMemcpyDeviceToHost, s));
    LARS_HIP_TRY(hipStreamSynchronize(s));
    if (shift) { shift[0] = (double)hs[0]; shift[1] = (double)hs[1]; }
    return LARS_OK;
}

int lars_h_change_detection(const uint8_t *early, const uint8_t *late, int64_t h, int64_t w, int channels, int wb_early,
                            int wb_late, int align, int index_id, float *out_early, float *out_late, float *out_diff,
                            uint8_t *out_rgba_diff, const uint8_t *lut_rgba, float vmin, float vmax,
                            uint8_t *out_aligned_late, double shift[2])
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!early || !late || h <= 0 || w <= 0 || channels < 3)
        return fail(LARS_ERR_INVALID, "lars_h_change_detection: two uint8 [h][w][>=3] images of the same shape are required");
    if (index_id < 0 || index_id > 2) return fail(LARS_ERR_INVALID, "lars_h_change_detection: index_id");
    if (align && channels != 3) return fail(LARS_ERR_INVALID, "lars_h_change_detection: registration needs 3-channel images (rgb2gray)");
    if (out_rgba_diff && !lut_rgba) return fail(LARS_ERR_INVALID, "lars_h_change_detection: out_rgba_diff needs lut_rgba");
    const size_t npix = (size_t)h * w, nbytes = npix * channels;
    const size_t tbytes = lars_wb_table_bytes(LARS_U8);
    Carver cv(nullptr);
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0 sizes the workspace, pass 1 carves it
        if (pass == 1) { LARS_TRY(ws_reserve(c, cv.off + 256)); cv = Carver(c->ws); }
        Carver &d = cv;
        uint8_t *d_e = d.take<uint8_t>(nbytes), *d_l = d.take<uint8_t>(nbytes);
        uint8_t *d_ewb = d.take<uint8_t>(nbytes), *d_lwb = d.take<uint8_t>(nbytes), *d_al = d.take<uint8_t>(nbytes);
        uint8_t *d_tab = d.take<uint8_t>(tbytes);
        double *d_pct = d.take<double>(6);
        double *fa = d.take<double>(align ? npix * 2 : 1), *fb = d.take<double>(align ? npix * 2 : 1);
        char *sc = d.take<char>(lars_phase_scratch_bytes());
        int64_t *d_shift = d.take<int64_t>(2);
        float *d_ie = d.take<float>(npix), *d_il = d.take<float>(npix), *d_df = d.take<float>(npix);
        uint8_t *d_rgba = d.take<uint8_t>(out_rgba_diff ? npix * 4 : 1), *d_lut = d.take<uint8_t>(1024);
        if (pass == 0) continue;

        hipStream_t s = c->stream;
        LARS_HIP_TRY(hipMemcpyAsync(d_e, early, nbytes, hipMemcpyHostToDevice, s));
        LARS_HIP_TRY(hipMemcpyAsync(d_l, late, nbytes, hipMemcpyHostToDevice, s));
        LARS_HIP_TRY(hipMemsetAsync(d_shift, 0, 2 * sizeof(int64_t), s));
        // white balance where the caller holds no cached corrected array (process-images.py:894-902)
        const uint8_t *src[2] = {d_e, d_l};
        uint8_t *wb_out[2] = {d_ewb, d_lwb};
        const int want_wb[2] = {wb_early, wb_late};
        for (int k = 0; k < 2; ++k) {
            if (!want_wb[k]) continue;
            LARS_TRY(lars_d_wb_prepare(src[k], 1, (int64_t)npix, channels, LARS_U8, d_tab, d_pct, 0, s));
            lars_fused_args a;
            memset(&a, 0, sizeof a);
            a.tiles = src[k]; a.ntiles = 1; a.npix = (int64_t)npix; a.channels = channels; a.dtype = LARS_U8;
            a.wb_table = d_tab; a.out_wb = wb_out[k]; a.stream = s;
            LARS_TRY(lars_d_fused(&a));
            src[k] = wb_out[k];
        }
        // registration of the late image onto the early one (:905-908)
        const uint8_t *late_final = src[1];
        if (align) {
            LARS_TRY(align_on_device(c, src[0], src[1], h, w, channels, fa, fb, sc, d_shift, d_al, s));
            late_final = d_al;
        }
        // indices (:911-919), difference (:923) and its colour map (:956)
        const uint8_t *imgs[2] = {src[0], late_final};
        float *idx[2] = {d_ie, d_il};
        for (int k = 0; k < 2; ++k) {
            lars_fused_args a;
            memset(&a, 0, sizeof a);
            a.tiles = imgs[k]; a.ntiles = 1; a.npix = (int64_t)npix; a.channels = channels; a.dtype = LARS_U8;
            a.index_mask = 1u << index_id; a.out_index[index_id] = idx[k]; a.stream = s;
            LARS_TRY(lars_d_fused(&a));
        }
        LARS_TRY(lars_d_diff_f32(d_ie, d_il, (int64_t)npix, d_df, s));
        if (out_rgba_diff) {
            LARS_HIP_TRY(hipMemcpyAsync(d_lut, lut_rgba, 1024, hipMemcpyHostToDevice, s));
            LARS_TRY(lars_d_colormap_norm_f32(d_df, (int64_t)npix, vmin, vmax, d_lut, d_rgba, s));
            LARS_HIP_TRY(hipMemcpyAsync(out_rgba_diff, d_rgba, npix * 4, hipMemcpyDeviceToHost, s));
        }
        int64_t hs[2] = {0, 0};
        if (out_early) LARS_HIP_TRY(hipMemcpyAsync(out_early, d_ie, npix * 4, hipMemcpyDeviceToHost, s));
        if (out_late) LARS_HIP_TRY(hipMemcpyAsync(out_late, d_il, npix * 4, hipMemcpyDeviceToHost, s));
        if (out_diff) LARS_HIP_TRY(hipMemcpyAsync(out_diff, d_df, npix * 4, hipMemcpyDeviceToHost, s));
        if (out_aligned_late) LARS_HIP_TRY(hipMemcpyAsync(out_aligned_late, late_final, nbytes, hipMemcpyDeviceToHost, s));
        LARS_HIP_TRY(hipMemcpyAsync(hs, d_shift, sizeof hs, hipMemcpyDeviceToHost, s));
        LARS_HIP_TRY(hipStreamSynchronize(s));
        if (shift) { shift[0] = (double)hs[0]; shift[1] = (double)hs[1]; }
    }
    return LARS_OK;
}

}  // extern "C"
