// Device helpers shared by the second-generation kernels (fused_v2.hip) and the exact-median select
// (select_q.hip): exact quotient, the white-balance table lookup, the ring load pipeline, the select's bucket.
#pragma once
#include "common.h"
#include "device_common.h"

namespace lars {

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

// ---------------------------------------------------------------------------
// exact quotient for integer-valued operands: |num| <= den, 1 <= den < 2^24
// ---------------------------------------------------------------------------
__device__ inline float exact_quot(float num, float den)
{
    const float r = __builtin_amdgcn_rcpf(den);
    const float q0 = num * r;
    const float e = __builtin_fmaf(-q0, den, num);
    return __builtin_fmaf(e, r, q0);
}
// a+b == 0 (both samples 0) must give +0.0.  Instead of a max(den, 1) per quotient, the NIR value gets
// a tiny epsilon once per pixel pair (one packed add shared by NDVI and GNDVI): float32(n + 1e-10) == n
// for every n >= 1 and the epsilon is absorbed again by any other sample >= 1, so the denominator is
// unchanged unless both samples are 0, where it becomes 1e-10 and the quotient (+0) * 1e10 = +0.0 --
// the same device the reference uses (process-images.py:464).
#define LARS_DEN_EPS 1e-10f

// two quotients per instruction: v_pk_mul_f32 / v_pk_fma_f32 (the VALU issues one wave64
// instruction per 4 cycles, packed or not -- rocprofv3: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.05 quad-cycles)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// den must be >= 1 or the tiny positive stand-in for a zero sum (see LARS_DEN_EPS)
__device__ inline f32x2 exact_quot2(f32x2 num, f32x2 den)
{
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(den.x);
    r.y = __builtin_amdgcn_rcpf(den.y);
    const f32x2 q0 = num * r;
    const f32x2 e = __builtin_elementwise_fma(-q0, den, num);
    return __builtin_elementwise_fma(e, r, q0);
}
// (a-b)/(a+b) with +0.0 where a+b == 0
__device__ inline float norm_diff_fast(float a, float b)
{
    const float s = a + b;
    const float d = a - b;
    return exact_quot(d, fmaxf(s, 1.0f));
}


#define V2_TABLE_BYTES 65536
// histogram bin of a quotient of bytes from the mantissa of fma(x, 25, 25.5001) + 2^23 (hist_pos2, fused_v2.hip)
#define V2_HIST_MAGIC_C 25.5001f

// byte k of a dword -> float in one VALU instruction.  Kept opaque (asm) so that hipcc does not
// turn "float(a) +/- float(b)" into integer SDWA adds plus conversions (5 instructions per pixel
// instead of 2 conversions + packed add/sub).
__device__ inline float cvt_ubyte(unsigned int w, int k)
{
    float f;
    switch (k) {
    case 0: asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w)); break;
    case 1: asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w)); break;
    case 2: asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w)); break;
    default: asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w)); break;
    }
    return f;
}

template <bool WB>
__device__ inline float sample(unsigned int word, int byte, int ch, unsigned int lane_off4, const char *s_tab)
{
    if (!WB) return cvt_ubyte(word, byte);
    // address = sample << 8 | lane*4 in one v_perm_b32 (S0 = word: selectors 4..7, S1 = lane_off4: 0..3)
    const unsigned int sel = 0x0c0c0000u | ((4u + (unsigned)byte) << 8);
    const unsigned int addr = __builtin_amdgcn_perm(word, lane_off4, sel);
    const unsigned int entry = *reinterpret_cast<const unsigned int *>(s_tab + addr);
    return cvt_ubyte(entry, ch);
}
template <bool WB>
__device__ inline unsigned int sample_entry(unsigned int word, int byte, unsigned int lane_off4, const char *s_tab)
{
    const unsigned int sel = 0x0c0c0000u | ((4u + (unsigned)byte) << 8);
    const unsigned int addr = __builtin_amdgcn_perm(word, lane_off4, sel);
    return *reinterpret_cast<const unsigned int *>(s_tab + addr);
}

// Exact-median select on uint8 quotients (select_q.hip; its first pass also lives in the statistics kernel).
// Position of x in [-1, 1]: t = fma(x, 1023.5, 3071.5) in [2048, 4095], one binade, so the 23 mantissa bits of t are
// 11 bits of bucket (floor(t) - 2048) followed by 12 bits of fraction (units of 2^-12).  Two quotients of bytes that differ
// are at least 1/(510 * 509) apart = 16.1 such units, 15 after the two roundings: inside one bucket the fraction, taken in
// groups of four units (1024 slots), still tells any two of them apart.
#define SELQ_BINS 2048
#define SELQ_SLOTS 1024
#define SELQ_T_BITS 0x45000000u                         /* float bits of 2048.0 */
// Predicted window of the one-pass median (select_q.hip): the statistics kernel counts, per stream, the values below the
// window and the slots inside it; the median is exact from those two whenever its rank falls inside.
// Slots: sigma(x) = round(x * 524032) -- 512 slots per bucket (8 of the bucket's units of 2^-12; two different quotients
// of bytes are 2.01 slots apart, so a slot still holds ONE distinct value).  The window is slots [ws, ws + 1920) = 3.75
// buckets, for any integer ws.  LDS row of a stream: 64 "below" words | 1920 slots | 64 "above" words = the 2048 words a
// bucket row takes otherwise, so the kernel keeps its 80 KiB (two blocks per CU).
// The word of x in that row comes out of ONE fma: u = fma(x, 524032, 1.5 * 2^23 - (ws - 64)) lies in [2^23, 2^24), where a
// float's mantissa IS its integer value, so bits(u) - bits(1.5 * 2^23) = sigma(x) - ws + 64 (selq_window_word).
#define SELQ_WIN_SLOTS 1920
#define SELQ_WIN_PER_BUCKET 512
#define SELQ_WIN_SCALE 524032.0f                           /* 1023.5 * 512 */
#define SELQ_WIN_MAGIC 12582912.0f                         /* 1.5 * 2^23 */
#define SELQ_WIN_MAGIC_BITS 0x4B400000u
#define SELQ_WIN_BOTTOM (-524032)                          /* sigma(-1): the lowest window start */
__device__ inline float selq_window_bias(int ws) { return (float)(12582912 - (ws - 64)); }      // exact: an integer below 2^24
__device__ inline int selq_window_word(float x, float bias)
{
    return (int)(__builtin_bit_cast(unsigned int, __builtin_fmaf(x, SELQ_WIN_SCALE, bias)) - SELQ_WIN_MAGIC_BITS);
}
__device__ inline float selq_t(float x) { return __builtin_fmaf(x, 1023.5f, 3071.5f); }
__device__ inline f32x2 selq_t2(f32x2 x)
{
    const f32x2 k = {1023.5f, 1023.5f}, c = {3071.5f, 3071.5f};
    return __builtin_elementwise_fma(x, k, c);
}
__device__ inline unsigned int selq_bucket_of(float t) { return (__builtin_bit_cast(unsigned int, t) >> 12) & 0x7FFu; }
__device__ inline void selq_add_bucket(float t, unsigned int row)      // row: LDS byte address of the stream's 2048-word row
{
    const unsigned int addr = (selq_bucket_of(t) << 2) + row;
    asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
}

// One value against a stream's window, branch-free: u (see above) as an integer is bits(1.5 * 2^23) + the value's word in
// the row if it lies inside the window (64 .. 64 + SELQ_WIN_SLOTS - 1); anything smaller is clamped onto the lane's own
// "below" word, anything larger onto its "above" word (values within 64 slots of the window land on a neighbour's word:
// still the right group).  After the fma (packed, two values at a time): one v_med3_i32, one shift-and-add whose constant
// takes bits(1.5 * 2^23) << 2 off again (mod 2^32), one LDS atomic.  lo / hi: the lane's two words + SELQ_WIN_MAGIC_BITS.
__device__ inline void selq_window_add(float u, unsigned int row_rel, int lo, int hi)
{
    int idx;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(idx) : "v"(__builtin_bit_cast(int, u)), "v"(lo), "v"(hi));
    const unsigned int addr = ((unsigned)idx << 2) + row_rel;        // row_rel = row - (SELQ_WIN_MAGIC_BITS << 2), mod 2^32
    asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
}

// The fused kernel's load pipeline as a reusable loop: four 12-byte buffer loads in flight per lane, slot k is
// consumed and refilled in place.  f(q, w0, w1, w2) sees every quad of the tile exactly once.
// `every` > 1: only every `every`-th grid stride is visited (a subsample in chunks of gridDim.x * NTHR quads).
template <int NTHR, typename F>
__device__ inline void for_each_quad_ring(const uint8_t *base, long long nquads, F &&f, int every = 1)
{
    const long long stride = (long long)gridDim.x * NTHR * every;
    const long long q0 = (long long)blockIdx.x * NTHR + threadIdx.x;
    const long long niter = (nquads + stride - 1) / stride;          // same for every lane of the grid
    if (niter <= 0) return;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(nquads * 12), 0x00020000);
    const unsigned int voff = (unsigned int)q0 * 12u;
    const unsigned int step_b = (unsigned int)stride * 12u;
    u32x3 w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, (unsigned)k * step_b, 0);
    long long it = 0;
    unsigned int soff = 4u * step_b;
    for (; it + 4 <= niter - 1; it += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f(q0 + (it + k) * stride, w[k].x, w[k].y, w[k].z);
            __builtin_amdgcn_sched_barrier(0);
            w[k] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff + (unsigned)k * step_b, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        soff += 4u * step_b;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long qq = q0 + (it + k) * stride;
        if (it + k < niter && qq < nquads) f(qq, w[k].x, w[k].y, w[k].z);
    }
}

}  // namespace lars
