/*
 * lars_hip.h -- C ABI of liblars_hip.so: the MI355X (gfx950) implementation of
 * the per-pixel multispectral index path of lars-uav/lars-image-processing.
 *
 * The reference has no FFI; its boundary for this path is three Python call
 * signatures (file:line in the upstream repository):
 *
 *   fix_white_balance(img_array)              process-images.py:424-447
 *                                             backend-process.py:17-26, process-rgn.py:4-49
 *   calculate_index(img_array, index_type)    process-images.py:449-490
 *   calculate_index(red, green, nir, type)    backend-process.py:28-38
 *   calculate_ndvi(image_path, ...)           process-ndvi.py:5-48   (float64 flavour)
 *   analyze_index(index_array, index_type)    process-images.py:492-513 (+ :651-658, :830-832)
 *   analyze_ndvi_statistics(ndvi_array)       process-ndvi.py:50-73
 *   plt.hist(bins=50, range=(-1,1))           process-ndvi.py:97
 *   imshow(cmap, vmin=-1, vmax=1)             process-images.py:695, backend-process.py:43
 *
 * Each entry point below names the interface it replaces.  A maintainer binds
 * them with ctypes (the shipped binding is
 * lars_image_processing_amd/_ffi.py).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every function returns LARS_OK (0) or a negative lars_status; the text of
 *     the last failure on the calling thread is lars_last_error().
 *   - "host entry points" (lars_h_*) take HOST pointers, stage through the
 *     library's device workspace and return when the result is in the caller's
 *     buffer.  They are re-entrant: each calling thread owns a stream and a
 *     workspace.
 *   - "device entry points" (lars_d_*) take DEVICE pointers (lars_malloc) and
 *     enqueue on `stream` (NULL = the calling thread's library stream) without
 *     synchronising; this is what the batched tile path and bench.py use.
 *   - there is no CPU fallback anywhere: without a gfx950 device every compute
 *     entry point fails with LARS_ERR_NO_DEVICE.
 */
#ifndef LARS_HIP_H
#define LARS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LARS_ABI_VERSION 1
#define LARS_HIST_BINS 50          /* process-ndvi.py:97 */

typedef enum lars_status {
    LARS_OK = 0,
    LARS_ERR_INVALID = -1,         /* bad argument (shape, dtype, NULL, mask) */
    LARS_ERR_NO_DEVICE = -2,       /* no usable gfx950 GPU / HIP runtime failure at init */
    LARS_ERR_HIP = -3,             /* a HIP call failed; see lars_last_error() */
    LARS_ERR_OOM = -4,             /* device allocation failed */
    LARS_ERR_RCCL = -5,            /* RCCL missing or a collective failed */
    LARS_ERR_UNSUPPORTED = -6
} lars_status;

/* sample types of interleaved images */
#define LARS_U8  1
#define LARS_U16 2

/* index ids / mask bits (process-images.py:466-482) */
#define LARS_NDVI  0               /* (nir - red)   / (nir + red)   */
#define LARS_GNDVI 1               /* (nir - green) / (nir + green) */
#define LARS_NDWI  2               /* (green - nir) / (green + nir) */
#define LARS_MASK_NDVI  1u
#define LARS_MASK_GNDVI 2u
#define LARS_MASK_NDWI  4u
#define LARS_MASK_ALL   7u

/* flags of lars_fused_args.flags */
#define LARS_F_STATS 1u            /* fill stats[tile][index] (min/max/sum/sumsq/above/count) */
#define LARS_F_HIST  2u            /* also the 50-bin histogram (implies LARS_F_STATS) */
#define LARS_F_SUMSQ 4u            /* also the sum of squares, for a standard deviation (implies LARS_F_HIST) */
#define LARS_F_RAW   8u            /* lars_d_fused only: args->stats are running accumulators that the caller has opened with
                                    * lars_d_stats_begin and closes with lars_d_stats_end -- one pair around the launches of a
                                    * batch that is processed in chunks, instead of one pair of small kernels per launch */

/*
 * Order-independent statistics record of one index over one tile (or, after a
 * merge, over many tiles / ranks).  Everything analyze_index (process-images.py
 * :506-512) and analyze_ndvi_statistics (process-ndvi.py:60-71) report except the
 * median follows from it: mean = sum/count, min, max, coverage = above/count*100,
 * std = sqrt(sumsq/count - mean^2).
 *   sum / sumsq : sums of the float32 index values (fixed point 2^-32 accumulation,
 *                 order independent); sum is exact for uint8 tiles and correctly
 *                 rounded to double.  The fused kernel fills sumsq only with
 *                 LARS_F_SUMSQ (it is 0 otherwise).
 *   above       : samples with x > threshold, compared in the sample's own
 *                 precision (float32 against float32(0.2), process-images.py:511).
 *   hist        : numpy.histogram(x, bins=50, range=(-1, 1)) counts.
 */
typedef struct lars_stats {
    double   sum;
    double   sumsq;
    uint64_t count;
    uint64_t above;
    uint64_t nans;
    double   min;                  /* float32 samples widen exactly */
    double   max;
    double   threshold;
    uint32_t index_id;
    uint32_t reserved;
    uint64_t hist[LARS_HIST_BINS];
} lars_stats;

/* ------------------------------------------------------------------ runtime */
int         lars_abi_version(void);
const char *lars_last_error(void);
int         lars_device_count(int *count);
int         lars_set_device(int ordinal);           /* per calling thread; default 0 */
int         lars_get_device(int *ordinal);
int         lars_device_name(char *buf, size_t buflen);
int         lars_malloc(void **dptr, size_t bytes);
int         lars_free(void *dptr);
int         lars_memset(void *dptr, int value, size_t bytes, void *stream);
int         lars_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int         lars_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int         lars_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int         lars_stream_create(void **stream);
int         lars_stream_destroy(void *stream);
int         lars_synchronize(void *stream);         /* NULL = calling thread's library stream */
int         lars_mem_info(size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo of the bound device */
int         lars_shutdown(void);                    /* frees the calling thread's workspace */
/* HIP events on a given stream (bench.py times kernels with these) */
int         lars_event_create(void **event);
int         lars_event_destroy(void *event);
int         lars_event_record(void *event, void *stream);
int         lars_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on stop */
/* work enqueued on `stream` after this call waits for `event` (recorded on any stream of the device): how a caller
 * orders passes it overlaps on two streams (hipStreamWaitEvent) */
int         lars_stream_wait_event(void *stream, void *event);

/* --------------------------------------------------------- device entry points */

/* Per-tile, per-channel sample histograms: the pre-pass np.percentile needs
 * (process-images.py:437).  hist is [ntiles][3][nvalues] uint32, nvalues = 256
 * (LARS_U8) or 65536 (LARS_U16); it is zeroed by the call. */
int lars_d_channel_hist(const void *tiles, int64_t ntiles, int64_t npix, int channels,
                        int dtype, uint32_t *hist, void *stream);

/* Histograms -> percentiles (2, 98) -> white-balance table, all on device.
 * table is [ntiles] x lars_wb_table_bytes(dtype) with table[t][c][v] == fix_white_balance()
 * of sample value v in channel c of tile t (process-images.py:437-441: float64
 * arithmetic, clip, float32 store, truncating uint8 cast).  percentiles is
 * [ntiles][3][2] double (may be NULL).  rgn_variant != 0 selects process-rgn.py
 * :25-33 (extra inner clip; same table for integer input, kept for fidelity). */
int lars_d_wb_table(const uint32_t *hist, int64_t ntiles, int64_t npix, int dtype,
                    uint8_t *table, double *percentiles, int rgn_variant, void *stream);

/* Bytes of one tile's white-balance table: 768 for LARS_U8 ([3][256] uint8); for LARS_U16 a blob of
 * 196608 + 4096 bytes: the [3][65536] uint8 table, then its threshold form (uint32 T[3][260],
 * T[c][k] = smallest sample value with table >= k) and double {p2, 255/(p98-p2)}[3] used by the
 * fast uint16 kernel.  wb_table arguments are [ntiles] of these. */
size_t lars_wb_table_bytes(int dtype);

/* The whole white-balance pre-pass of a batch in one call (histogram pass(es) + tables), both
 * sample types.  For LARS_U16 the percentiles come from two radix levels (high-byte histograms,
 * then low-byte histograms of the bins that hold the order statistics) instead of a 65536-bin
 * histogram per channel -- usually in ONE full pass: a subsample predicts candidate bins, the pass
 * counts high bytes and the candidates' low bytes together, and only tiles whose exact bins were
 * not among their candidates are recounted.  Scratch comes from the calling thread's library workspace. */
int lars_d_wb_prepare(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                      uint8_t *table, double *percentiles, int rgn_variant, void *stream);

typedef struct lars_fused_args {
    const void    *tiles;          /* [ntiles][h*w][channels] interleaved R,G,NIR(,...) */
    int64_t        ntiles;
    int64_t        npix;           /* h*w */
    int32_t        channels;       /* >= 3 */
    int32_t        dtype;          /* LARS_U8 | LARS_U16 */
    const uint8_t *wb_table;       /* [ntiles] x lars_wb_table_bytes(dtype), or NULL: indices of the raw samples */
    uint32_t       index_mask;     /* LARS_MASK_* */
    uint32_t       flags;          /* LARS_F_* */
    float         *out_index[3];   /* [ntiles][npix] float32 per index, or NULL */
    uint8_t       *out_wb;         /* [ntiles][npix][channels] uint8 (channels >= 3 zero), or NULL */
    uint8_t       *out_rgba[3];    /* [ntiles][npix][4] colormapped index, or NULL */
    const uint8_t *cmap_lut[3];    /* [256][4] RGBA8 table per index (needed where out_rgba set) */
    lars_stats    *stats;          /* [ntiles][3] (entries of unrequested indices untouched), or NULL */
    void          *stream;
} lars_fused_args;

/* The hot path: one pass over interleaved tiles doing band de-interleave,
 * white-balance table lookup, NDVI/GNDVI/NDWI (process-images.py:456-490, IEEE
 * float32), optional float32 / uint8 / RGBA8 outputs and per-tile statistics. */
int lars_d_fused(const lars_fused_args *args);
/* Open / close the statistics records [ntiles][3] of the indices in index_mask for launches with LARS_F_RAW (begin: the
 * accumulators of lars_d_fused's own prologue; end: sums, count, minimum and maximum in their final form). */
int lars_d_stats_begin(lars_stats *stats, int64_t ntiles, uint32_t index_mask, void *stream);
int lars_d_stats_end(lars_stats *stats, int64_t ntiles, uint32_t index_mask, int64_t npix, void *stream);

/* float32 band planes -> index, backend-process.py:28-38 (literal formula with
 * the float32 epsilon add and the clip; any float input). */
int lars_d_index_planes_f32(const float *red, const float *green, const float *nir,
                            int64_t n, int index_id, float *out, void *stream);

/* interleaved image -> float64 NDVI, process-ndvi.py:18-31 */
int lars_d_ndvi_f64(const void *img, int64_t npix, int channels, int dtype,
                    double *out, void *stream);

/* Statistics of an arbitrary float32 / float64 array (process-images.py:506-512,
 * process-ndvi.py:60-71).  want_hist adds the 50-bin histogram. */
int lars_d_array_stats_f32(const float *x, int64_t n, float threshold, int want_hist,
                           lars_stats *out_dev, void *stream);
int lars_d_array_stats_f64(const double *x, int64_t n, double threshold, int want_hist,
                           lars_stats *out_dev, double *out_sumsqdev_dev, void *stream);

/* np.median by radix select on device: writes the two middle order statistics
 * ((n-1)/2 and n/2) to out_dev[0..1]. scratch must hold lars_select_scratch_bytes(). */
size_t lars_select_scratch_bytes(void);
int lars_d_median_pair_f32(const float *x, int64_t n, float *out_dev, void *scratch, void *stream);
int lars_d_median_pair_f64(const double *x, int64_t n, double *out_dev, void *scratch, void *stream);
/* Batched: `items` arrays of n float32 values, `stride` values apart (the index planes of a batch of
 * tiles); out_dev is [items][2]; scratch must hold items * lars_select_scratch_bytes(). */
int lars_d_median_pair_batch_f32(const float *x, int64_t n, int64_t items, int64_t stride, float *out_dev,
                                 void *scratch, void *stream);

/* One pass of the two-level select over the index values of a batch WITHOUT the planes in memory (exact batch /
 * global medians): the NDVI and GNDVI quotients of every pixel are recomputed from the
 * uint8 tiles.  Position of a value: t = float32 fma(x, 1023.5, 3071.5) in [2048, 4095]; its 23 mantissa bits are
 * 11 bits of bucket and 12 bits of fraction.  first == 1: each value is counted in its bucket, under track 0 (first == 3:
 * the same over a 1/16 subsample of the pixels -- the prediction of the window below).
 * first == 0: per stream s and track k, the values of bucket[2s+k] are counted in slot (fraction >> 2) (1024 slots);
 * when both tracks of BOTH streams share their bucket only track 0 is counted.  A slot holds one distinct value
 * (two different quotients of bytes are >= 1/(510 * 509) apart = 16 units of the fraction).
 * first == 2: ONE pass instead of those two where a predicted window holds the ranks.  Slots sigma(x) = round(x * 524032);
 * (int32) bucket[2s] is the first slot ws of stream s's window of 1920 slots (3.75 buckets); under track 0 of stream s
 * words 0..63 together count the values below the window, words 64..1983 its slots, words 1984..2047 the values above.
 * hist is uint64[2 streams][2 tracks][2048], accumulated with atomics (zero it first).  NDWI = -GNDVI shares
 * GNDVI's order statistics. */
int lars_d_quotient_select_hist(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                const uint8_t *wb_table, uint32_t streams /* bit 0 NDVI, bit 1 GNDVI (and NDWI) */, int first,
                                const uint32_t bucket[4], uint64_t *hist, void *stream);
/* np.median of the NDVI and GNDVI planes of EVERY tile without writing a plane: per-tile histograms, the picks and the
 * value look-up on the device.  Usually ONE full pass: a subsample predicts a window per tile and stream, a window pass
 * (first == 2 above, per tile) counts it, and only tiles whose ranks fall outside take the two classic passes -- to find
 * out which, the call waits for its stream once (one word comes back to the host; lars_set_tuning("selq_window", 0)
 * = always the two passes, no wait).  out_pairs is float[ntiles][2 streams: NDVI, GNDVI][2]: the two middle order
 * statistics (median = their float32 mean; NDWI's median is -GNDVI's).  scratch holds
 * lars_quotient_median_scratch_bytes(ntiles). */
size_t lars_quotient_median_scratch_bytes(int64_t ntiles);
/* The statistics of lars_d_fused (a->stats, LARS_F_HIST honoured; index_mask = one index or all three; no
 * output planes) AND those medians in one call: the statistics kernel also counts the select's first
 * pass, so the tiles are read twice (three times with the white-balance histogram pass).  Streams the mask does not ask for come back
 * as NaN pairs.  With plain LARS_F_STATS the select passes are usually not needed at all: a subsample predicts a window per
 * tile and stream, the statistics kernel counts the values below it and the slots inside it instead of the buckets, and only
 * tiles whose ranks fall outside their window take the two classic passes -- to find out which, this call waits for its
 * stream once (one word comes back to the host). */
int lars_d_stats_medians(const lars_fused_args *a, float *out_pairs, void *scratch);
int lars_d_quotient_median_pairs(const void *tiles, int64_t ntiles, int64_t npix, int channels, int dtype,
                                 const uint8_t *wb_table, uint32_t streams /* bit 0 NDVI, bit 1 GNDVI; others come back NaN */,
                                 float *out_pairs, void *scratch, void *stream);

/* The statistics of lars_d_fused -- and the exact medians of lars_d_stats_medians -- from ONE read of the tiles, with the
 * white balance's own percentile pre-pass inside (csrc/joint.hip): every figure analyze_index reports
 * (process-images.py:506-512: mean, median, min, max, coverage), the 50-bin histogram (process-ndvi.py:97) and
 * np.percentile(ch, (2, 98)) of fix_white_balance (process-images.py:437) is a function of how often each pair of raw
 * bytes (nir, red) resp. (nir, green) occurs.  A counting kernel builds those 2 x 65536 counts per tile in LDS (no table
 * look-up, no quotient per pixel), a second kernel derives channel histograms -> percentiles -> tables -> records ->
 * medians from them.  The records are bit-identical to lars_d_fused's -- except the optional sum of squares (LARS_F_SUMSQ:
 * count x value^2 per cell here, value^2 per pixel there; they agree to a few units of 2^-32) -- the medians to np.median.
 *   a             uint8 tiles with 3 channels (RGNir; 4-byte aligned) or 4 (RGBA, alpha ignored; 16-byte aligned), npix < 2^32
 *                 (the per-cell counts are uint32); index_mask any non-empty subset; LARS_F_HIST / LARS_F_SUMSQ honoured; no
 *                 output planes; a->stats [ntiles][3] receives final records (unrequested indices untouched);
 *                 a->wb_table, if not NULL, is an OUTPUT here: [ntiles][3][256] tables of the channels the mask needs
 *                 (NIR and red for NDVI, NIR and green for GNDVI / NDWI)
 *   white_balance 0: indices of the raw samples (calculate_index without fix_white_balance); 1: percentile white balance
 *   percentiles   [ntiles][3][2] double or NULL, hist [ntiles][3][256] or NULL: outputs, same channels as the tables.
 *                 With hist == NULL, white_balance != 0, both value streams in the mask and tiles of >= 2^20 pixels the call may
 *                 count a tile on WINDOWED tables (csrc/joint_win.hip): red and green clamped to a window around their
 *                 percentiles -- fix_white_balance maps everything at or below p2 to 0 and at or above p98 to 255
 *                 (process-images.py:438), so the records, percentiles, tables and medians are the same bits -- which lets both pair
 *                 tables of a tile chunk live in ONE workgroup's LDS: one reader per byte instead of two.  A subsample predicts
 *                 the windows, the exact percentiles check them, a tile whose window missed is counted again on full tables
 *                 (lars_set_tuning("joint_window", 0) = never windowed; 2 = windows that miss on purpose).  Where red and green
 *                 windows with NIR whole do not fit (more than 306 rows) NIR gets a window too -- it is white-balanced through
 *                 its own percentiles like the others -- and rows shrink to its width: three windows of up to about 196 values
 *                 each still share one workgroup.  Channel histograms cannot be had from clamped counts: asking for hist keeps
 *                 the full tables
 *   out_pairs     float[ntiles][2 streams: NDVI, GNDVI][2] or NULL: the two middle order statistics (median = their
 *                 float32 mean; NDWI's is -GNDVI's); a stream the mask does not need comes back as NaN
 *   scratch       scratch_bytes >= lars_joint_scratch_bytes(ntiles, npix, index_mask) bytes of device memory; its first word
 *                 is an error flag the launch leaves at 0 (1 = a workgroup's hand-over list overflowed: cannot happen while a
 *                 workgroup counts at most 2^24 pixels, which the chunking guarantees) */
size_t lars_joint_scratch_bytes(int64_t ntiles, int64_t npix, uint32_t index_mask);
int lars_d_stats_joint(const lars_fused_args *a, int white_balance, int rgn_variant, double *percentiles, uint32_t *hist,
                       float *out_pairs, void *scratch, size_t scratch_bytes);
/* How the last lars_d_stats_joint on `scratch` used windowed tables (csrc/joint_win.hip), once its stream has finished:
 * *windowed = tiles counted by one reader on windowed tables, *recounted = tiles among them whose window missed a percentile's
 * order statistic and which were counted again on full tables.  Both 0 after a call that did not qualify (see above). */
int lars_joint_window_report(const void *scratch, int64_t ntiles, int64_t *windowed, int64_t *recounted);
/* ... and by table form: counts[0] = tiles counted on full tables (two readers), counts[1] = on windowed red and green rows with NIR
 * whole, counts[2] = on three windows (recounted tiles are listed under the form they were first counted on). */
int lars_joint_window_modes(const void *scratch, int64_t ntiles, int64_t counts[3]);

/* classification mask (see lars_h_threshold_mask_f32); x 16-byte, out_mask 4-byte aligned */
int lars_d_threshold_mask_f32(const float *x, int64_t n, float threshold, uint8_t *out_mask, void *stream);

/* float32 index -> RGBA8: LUT[min(int((x + 1f) * 128f), 255)], the per-pixel
 * mapping of imshow(cmap, vmin=-1, vmax=1) (process-images.py:695). */
int lars_d_colormap_f32(const float *x, int64_t n, const uint8_t *lut_rgba, uint8_t *out_rgba,
                        void *stream);
/* ... and only the entry min(int((x + 1f) * 128f), 255) of every sample, one byte each (x 16-byte, out_entry 4-byte aligned). */
int lars_d_colormap_entry_f32(const float *x, int64_t n, uint8_t *out_entry, void *stream);

/* ---- registration and change detection ---- */
/* skimage.color.rgb2gray of a uint8 [npix][3] image (process-images.py:538-546) into the real parts of a
 * complex128 buffer [npix][2] (imaginary parts 0); channels == 1: the sample value itself. */
int lars_d_gray_c128(const uint8_t *img, int64_t npix, int channels, double *out_c128, void *stream);
/* skimage.registration.phase_cross_correlation(fixed, moving) with its defaults (process-images.py:549):
 * FFT cross-power spectrum / max(|.|, 100 eps) -> inverse FFT -> argmax |.| -> signed shift.  Both
 * [h][w] complex128 buffers are overwritten.  shift_dev receives {dy, dx}; scratch holds
 * lars_phase_scratch_bytes().  The FFTs run in rocFFT via libhipfft.so, loaded on first use. */
size_t lars_phase_scratch_bytes(void);
int lars_d_phase_correlation(double *fixed_c128, double *moving_c128, int64_t h, int64_t w, int64_t *shift_dev,
                             void *scratch, void *stream);
/* scipy.ndimage.shift(img, (dy, dx, 0), order=1, mode='reflect') for the integer shift in shift_dev
 * (process-images.py:557); out must not alias img. */
int lars_d_shift_reflect_u8(const uint8_t *img, int64_t h, int64_t w, int channels, const int64_t *shift_dev,
                            uint8_t *out, void *stream);
/* diff = late - early (process-images.py:923) */
int lars_d_diff_f32(const float *early, const float *late, int64_t n, float *out_diff, void *stream);
/* RGBA8 of cmap(Normalize(vmin, vmax)(x), bytes=True): the per-pixel mapping of
 * imshow(x, cmap, vmin, vmax) -- e.g. the change map's bwr, -0.5, 0.5 (process-images.py:956). */
int lars_d_colormap_norm_f32(const float *x, int64_t n, float vmin, float vmax, const uint8_t *lut_rgba,
                             uint8_t *out_rgba, void *stream);

/* Synthetic RGNir tiles generated in HBM (bench / tests): counter hash of
 * (seed, tile, word); profile 0 = uniform bytes, 1 = vegetation-like squeeze. */
int lars_d_synth_u8(uint8_t *tiles, int64_t ntiles, int64_t first_tile, int64_t npix, int channels,
                    uint32_t seed, int profile, void *stream);

/* Tuning knobs (per process): "fused_impl" 0 (auto)|1|2, "hist_impl" 1|2, "nt_stores" 0|1, "blocks_per_tile" 0 = automatic
 * (also the chunks per tile of lars_d_stats_joint), "joint_depth" 4|6 loads in flight per lane of the counting kernel,
 * "joint_window" 1 (windowed pair tables where they fit: lars_d_stats_joint)|0 (never)|2 (windows that miss on purpose: exercises the
 * recount; tiles of any size)|3 (as 1 for tiles of any size)|4 (three windows -- NIR as well -- wherever they fit, before two are tried; tiles
 * of any size)|5 (as 4 with NIR windows that miss on purpose), "joint_win_depth" 4|5|6|12|15 loads in flight per lane of the windowed counting kernel,
 * "u16_hist_impl" 5 (uint16 percentiles usually from ONE full pass: per channel and mark the count of the samples below a window
 * predicted from a subsample and the histogram inside it; a tile whose window missed takes the two radix passes)|1 (always the two
 * radix passes)|3 (windows that miss on purpose: exercises the fall-back),
 * "selq_window" 1 (one-pass medians)|0 (always two select passes)|2 (wrong windows: exercises the fallback), "selq_list_wgs"
 * workgroups per select pass over the tiles a window missed (0 = 2048).  Results never depend on them.  Read-only:
 * "last_fused_kernel" = the kernel family the last lars_d_fused launched (1 k_fused_u8c3, 2 k_fused_v2, 3 its uint16 form,
 * 4 k_fused_generic: one pixel per lane, 5 k_fused_u8c3 for RGBA uint8 tiles). */
int lars_set_tuning(const char *key, int value);
int lars_get_tuning(const char *key, int *value);

/* Build-time switches of the kernel sources this library was compiled with: 0 for the product (what `make` builds and the
 * package loads).  Laboratory builds (csrc/Makefile `lablayout`, `EXTRA=-D...`) report what they changed: the plane-layout knob
 * (one more kernel argument: other register counts), IEEE division instead of rcp + fma (same results, slower), vector instead of
 * scalar coverage counters (same results), another geometry of the statistics-only kernels.  Needs no device. */
#define LARS_BUILD_LAB_LAYOUT 1u
#define LARS_BUILD_IEEE_DIV 2u
#define LARS_BUILD_COUNT_MODE 4u
#define LARS_BUILD_STATS_GEOMETRY 8u
unsigned int lars_build_flags(void);

/* Device self-check: number of (num, den) pairs, 1 <= den <= max_den, |num| <= den,
 * for which the kernels' rcp+fma quotient differs from IEEE float32 division
 * (must be 0; tests run it for the uint8 and the uint16 operand ranges). */
int lars_d_quot_selfcheck(uint32_t max_den, uint64_t *mismatches, uint32_t first_bad[2]);

/* Fold n records (same index) into one: sums add, min/max fold, histograms add. */
int lars_stats_merge(const lars_stats *records, int64_t n, lars_stats *out);
/* The same fold on the device, per index over the tiles of a batch: tile_records [ntiles][3] (final form) -> out[3]
 * (entries of indices outside index_mask untouched), bit-identical to lars_stats_merge over tile_records[:, k] in tile
 * order.  The three records are what a rank hands to lars_comm_allreduce_stats(..., is_device = 1): nothing but 3 x 472
 * bytes leaves the device per step. */
int lars_d_stats_fold(const lars_stats *tile_records, int64_t ntiles, uint32_t index_mask, lars_stats *out, void *stream);

/* ----------------------------------------------------------- host entry points */

/* fix_white_balance(img_array) -- process-images.py:424-447 (variant 0),
 * backend-process.py:17-26 (variant 0), process-rgn.py:25-44 (variant 1).
 * img/out are host [h][w][channels]; out is uint8 with channels >= 3 zeroed. */
int lars_h_fix_white_balance(const void *img, int64_t h, int64_t w, int channels, int dtype,
                             int variant, uint8_t *out, double *percentiles /* [3][2] or NULL */);
/* The same function for any other sample type: process-images.py:431 casts whatever it is given with
 * astype(np.float32) before anything else, so the caller hands over that float32 image.  np.percentile's order
 * statistics come from an exact radix select, numpy's `_lerp` (float32 difference, float64 blend) and the float64
 * stretch / float32 store / truncating cast of :438-441 run on the device. */
int lars_h_fix_white_balance_f32(const float *img, int64_t h, int64_t w, int channels, uint8_t *out,
                                 double *percentiles /* [3][2] or NULL */);

/* calculate_index(img_array, index_type) -- process-images.py:449-490.
 * Several indices in one pass: out[k] is host [h][w] float32 or NULL. */
int lars_h_calculate_index(const void *img, int64_t h, int64_t w, int channels, int dtype,
                           uint32_t index_mask, float *const out[3],
                           lars_stats *stats /* [3] or NULL */, int want_hist);

/* calculate_index(red, green, nir, index_type) -- backend-process.py:28-38 */
int lars_h_calculate_index_planes(const float *red, const float *green, const float *nir,
                                  int64_t n, int index_id, float *out);

/* calculate_ndvi maths -- process-ndvi.py:18-31 (float64 out) */
int lars_h_ndvi_f64(const void *img, int64_t h, int64_t w, int channels, int dtype, double *out);

/* analyze_index / analyze_ndvi_statistics on a host array: stats + the two
 * middle order statistics (median = their mean) + sum of squared deviations
 * from the mean (np.std, float64 flavour). */
int lars_h_analyze_f32(const float *x, int64_t n, float threshold, int want_hist,
                       lars_stats *out, float median_pair[2]);
int lars_h_analyze_f64(const double *x, int64_t n, double threshold, int want_hist,
                       lars_stats *out, double median_pair[2], double *sumsqdev);

/* Whole reference pipeline for one host image in one upload:
 * white balance -> requested indices -> statistics (+ medians) -> optional
 * RGBA8 colormaps.  Any output pointer may be NULL.  out_rgba[k] with cmap_lut[k] == NULL (or cmap_lut == NULL) receives the
 * colormap ENTRY of every pixel instead -- [h][w] uint8, min(int((x + 1f) * 128f), 255): the pixels of a palette image whose
 * palette is the colormap (backend-process.py:40-47, process-images.py:690-695 per pixel) -- one byte per pixel over PCIe. */
int lars_h_process_image(const void *img, int64_t h, int64_t w, int channels, int dtype,
                         int apply_wb, uint32_t index_mask, int want_hist,
                         uint8_t *out_wb, float *const out_index[3],
                         lars_stats *stats /* [3] */, float *medians /* [3][2] */,
                         uint8_t *const out_rgba[3], const uint8_t *const cmap_lut[3]);

int lars_h_colormap_f32(const float *x, int64_t n, const uint8_t *lut_rgba, uint8_t *out_rgba);
/* classification mask: out_mask[i] = index[i] > threshold in float32 (the array np.mean averages for the
 * coverage figures, process-images.py:511 / :657); 1 byte per sample, 0 or 1 */
int lars_h_threshold_mask_f32(const float *x, int64_t n, float threshold, uint8_t *out_mask);

/* preprocess_large_image -- process-images.py:398-422: PIL.Image.resize((new_w, new_h), LANCZOS) of a
 * uint8 image with 1, 3 or 4 (RGBA, premultiplied-alpha path) channels, bit-identical to Pillow's
 * fixed-point resampler.  out is host [new_h][new_w][channels]. */
int lars_h_resize_lanczos_u8(const uint8_t *img, int64_t h, int64_t w, int channels, int64_t new_h, int64_t new_w,
                             uint8_t *out);

/* align_images -- process-images.py:515-565 for two uint8 images of the same shape ([h][w][3] or [h][w]):
 * out_aligned = moving registered onto fixed, shift = {dy, dx} (what phase_cross_correlation returns). */
int lars_h_align_images(const uint8_t *fixed, const uint8_t *moving, int64_t h, int64_t w, int channels,
                        uint8_t *out_aligned, double shift[2]);
/* create_change_detection_visualization's arithmetic -- process-images.py:885-923 and :956 -- in one upload:
 * optional white balance of either image, optional registration of the late image, the index of both,
 * diff = late - early and its colormap.  Any output pointer may be NULL. */
int lars_h_change_detection(const uint8_t *early, const uint8_t *late, int64_t h, int64_t w, int channels,
                            int wb_early, int wb_late, int align, int index_id,
                            float *out_early, float *out_late, float *out_diff,
                            uint8_t *out_rgba_diff, const uint8_t *lut_rgba, float vmin, float vmax,
                            uint8_t *out_aligned_late, double shift[2]);
int lars_h_colormap_norm_f32(const float *x, int64_t n, float vmin, float vmax, const uint8_t *lut_rgba,
                             uint8_t *out_rgba);

/* ------------------------------------------------------------------ image files (host only, no GPU work) */
/* TIFF 6.0 LZW (compression 5, MSB-first codes, early width change) of one strip / tile: decodes at most ndst bytes
 * into dst, *nout = bytes produced.  For lars_image_processing_amd/tiffio.py, which reads the multi-sample 16-bit
 * TIFFs that PIL.Image.open (backend-process.py:52) reduces to 8 bits. */
int lars_h_tiff_lzw_decode(const uint8_t *src, int64_t nsrc, uint8_t *dst, int64_t ndst, int64_t *nout);
/* Every strip / tile of an image in one call, shared by `threads` workers: chunk i = file[offsets[i], +counts[i])
 * decodes into dst + i * chunk_bytes (at most chunk_bytes bytes), produced[i] = bytes that came out. */
int lars_h_tiff_lzw_decode_chunks(const uint8_t *file, int64_t file_len, const uint64_t *offsets, const uint64_t *counts,
                                  int64_t nchunks, uint8_t *dst, int64_t chunk_bytes, int64_t *produced, int threads);

/* ------------------------------------------------------------------ multi-GPU */
/* One process per GPU.  RCCL (librccl.so) is loaded on first use.  unique_id is
 * LARS_COMM_ID_BYTES bytes produced by lars_comm_unique_id() on rank 0 and
 * handed to the other ranks by the launcher (file, socket, environment). */
#define LARS_COMM_ID_BYTES 128
/* LARS_OK if librccl loads with every symbol needed (a pre-flight check: a rank that cannot join must say so before the
 * others block in ncclCommInitRank) */
int lars_comm_available(void);
int lars_comm_unique_id(uint8_t *id_out);
int lars_comm_init(void **comm, int nranks, int rank, const uint8_t *unique_id);
/* number of ranks RCCL reports for the communicator (ncclCommCount): bench.py prints it as config.ranks_seen */
int lars_comm_count(void *comm, int *nranks_out);
int lars_comm_destroy(void *comm);
/* Global statistics: every rank contributes n records (device or host memory,
 * is_device says which); on return every rank holds the fold over all ranks in
 * rank order (one ncclAllGather over xGMI + a deterministic local fold). */
int lars_comm_allreduce_stats(void *comm, lars_stats *records, int64_t n, int is_device, void *stream);
/* max / sum of a double over ranks (bench timing, barriers) */
int lars_comm_allreduce_f64(void *comm, double *values_host, int64_t n, int op /*0 sum,1 max,2 min*/);
int lars_comm_barrier(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* LARS_HIP_H */
