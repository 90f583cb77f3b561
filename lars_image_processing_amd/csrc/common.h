// Internal helpers shared by the translation units of liblars_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "host_common.h"

namespace lars {

// ---- error plumbing (set_error / fail / LARS_TRY: host_common.h) -----------

#define LARS_HIP_TRY(expr)                                                          \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess)                                                       \
            return ::lars::fail(_e == hipErrorOutOfMemory ? LARS_ERR_OOM : LARS_ERR_HIP, \
                                "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                __FILE__, __LINE__);                                \
    } while (0)

// ---- per-thread context ---------------------------------------------------
struct ThreadCtx {
    int device = -1;           // -1: not initialised yet
    hipStream_t stream = nullptr;
    void *ws = nullptr;        // grow-only device workspace of the host entry points
    size_t ws_bytes = 0;
    void *scratch = nullptr;   // small device scratch of the device entry points
    size_t scratch_bytes = 0;
};

int ensure_ctx(ThreadCtx **out);                 // initialises device 0 on first use
int ws_reserve(ThreadCtx *c, size_t bytes);      // grow-only; synchronises when it grows
int scratch_reserve(ThreadCtx *c, size_t bytes);
inline hipStream_t pick_stream(ThreadCtx *c, void *stream) {
    return stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
}

// ---- device-side shared pieces ---------------------------------------------
// 2^32: the fixed-point scale of the statistics accumulators.
#define LARS_FX_SCALE 4294967296.0
#define LARS_FX_INV (1.0 / 4294967296.0)

// While a fused launch is in flight a lars_stats record is used as its own
// accumulator: sum / sumsq hold int64 fixed point, min / max hold order-
// preserving uint64 keys.  k_stats_finalize converts in place.
struct StatsAccView {
    unsigned long long sum_fx;
    unsigned long long sumsq_fx;
    unsigned long long count;
    unsigned long long above;
    unsigned long long nans;
    unsigned long long min_key;      // f64_key of the running minimum
    unsigned long long max_key;
    double threshold;
    unsigned int index_id;
    unsigned int reserved;
    unsigned long long hist[LARS_HIST_BINS];
};
static_assert(sizeof(StatsAccView) == sizeof(lars_stats), "accumulator view must alias lars_stats");

__host__ __device__ inline unsigned int f32_key(float x) {
    unsigned int b = __builtin_bit_cast(unsigned int, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key_f32(unsigned int k) {
    unsigned int b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, b);
}
__host__ __device__ inline unsigned long long f64_key(double x) {
    unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double key_f64(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __builtin_bit_cast(double, b);
}

// numpy.histogram(bins=50, range=(-1,1)) edges: linspace computes
// arange(51) * (2/50) + (-1) in float64, pins the last edge to 1, then casts to
// the sample dtype.
__host__ __device__ inline double hist_edge_f64(int i) {
    return i >= LARS_HIST_BINS ? 1.0 : (double)i * (2.0 / 50.0) + (-1.0);
}

// white-balance table blob of a uint16 tile: [3][65536] uint8 table | uint32 thr[3][260] | double par[3][2]
#define LARS_U16_THR_OFFSET 196608
#define LARS_U16_PAR_OFFSET (196608 + 3 * 260 * 4)
#define LARS_U16_BLOB_BYTES (196608 + 4096)

// launch geometry helpers (host)
int launch_check(const char *what);
void align_release();                            // drops the calling thread's cached FFT plan (lars_shutdown)

// tuning knobs (lars_set_tuning)
struct Tuning {
    int fused_impl = 0;        // 0: automatic, 1: first-generation kernels (fused.hip), 2: fused_v2.hip
    int hist_impl = 2;
    int nt_stores = 0;         // non-temporal stores for the float32 planes
    int blocks_per_tile = 0;   // 0 = automatic
    int selq_window = 1;       // one-pass medians (select_q.hip): 1 predicted window, 0 always two passes, 2 wrong windows (test)
    int selq_list_wgs = 0;     // workgroups per launch of the classic select passes over the tiles a window missed (0 = 2048)
    int last_fused_kernel = 0; // read-only: the kernel family lars_d_fused launched last -- 1 k_fused_u8c3, 2 k_fused_v2, 3 uint16, 4 generic, 5 RGBA uint8
    int u16_hist_impl = 5;     // uint16 percentiles (u16.hip): 5 one full pass on value windows (counts below + histograms inside), 1 always two
                               // radix passes, 3 test hook (value windows that miss: every tile is flagged and takes the two passes as well)
    int out_stride_planes = 0; // laboratory build (LARS_LAB_LAYOUT) only: k > 1 = the fused kernel steps k x npix from tile to tile in its index planes
    int joint_depth = 6;       // joint.hip: 12-byte loads in flight per lane of the counting kernel (4 | 6 | 8 | 12)
    int joint_win_depth = 15;  // joint_win.hip: loads in flight per lane of the windowed counting kernel: 5 | 15 (a sweep every 15 steps), 4 | 6 | 12 (every 12)
    int joint_window = 1;      // joint_win.hip: 1 windowed pair tables (one reader per tile chunk) where they fit, 0 never, 2 test hook (windows that miss)
};
Tuning &tuning();

struct FusedParams;
#ifndef LARS_V2_STATS_THREADS
#define LARS_V2_STATS_THREADS 1024     // statistics-only second-generation kernels (tools/kbench.py A/B: 512 | 1024)
#endif
#ifndef LARS_V2_STATS_WAVES
#define LARS_V2_STATS_WAVES 8          // waves per SIMD the statistics-only kernels are compiled for (64 VGPRs)
#endif
int fused_v2_threads(bool any_out);
void fused_v2_launch(unsigned mask, bool wb, int stats, bool nt, dim3 grid, hipStream_t s, const FusedParams &P);
void chan_hist_v2_launch(const uint8_t *tiles, long long npix, unsigned int *hist, dim3 grid, hipStream_t s, int channels = 3);
int selq_pass_launch(const uint8_t *tiles, const uint8_t *wb_table, long long ntiles, long long npix, int first,
                     const unsigned int bucket[4], unsigned long long *hist, hipStream_t s, unsigned streams);
size_t selq_tile_scratch_bytes(long long ntiles);
int selq_tile_medians_launch(const uint8_t *tiles, const uint8_t *wb_table, long long ntiles, long long npix, float *out_pairs,
                             void *scratch, hipStream_t s, bool first_pass_done, unsigned streams, bool windowed = false);
int selq_tile_prepare(void *scratch, long long ntiles, long long npix, hipStream_t s, unsigned int streams);
unsigned int *selq_tile_hist32(void *scratch, long long ntiles);
void fused_v2_sel_launch(unsigned mask, bool wb, int stats, dim3 grid, hipStream_t s, const FusedParams &P);
// fused.hip: grid sizing and the statistics records' init / finalize kernels, for the other translation units
int blocks_per_tile(long long work_items, long long ntiles, int threads = 256, long long target_total = 8192);
void stats_init_launch(lars_stats *stats, long long nrec, unsigned int mask, hipStream_t s);
void stats_finalize_launch(lars_stats *stats, long long nrec, unsigned int mask, long long npix, hipStream_t s);
int quot_check_launch(unsigned int max_den, unsigned long long *mismatches_dev, unsigned int *first_bad_dev, hipStream_t s);

}  // namespace lars
