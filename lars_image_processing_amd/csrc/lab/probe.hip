// Roofline probes: what this device sustains for plain streaming reads / copies
// with the access shapes the hot path uses.  Reported by bench.py next to the
// kernel numbers ("verify the peak with a STREAM-like kernel").
#include "lab.h"

namespace lars {

template <int WORDS, int UNROLL>
__global__ __launch_bounds__(256) void k_probe_read(const unsigned int *__restrict__ src, long long nvec,
                                                    unsigned int *__restrict__ sink)
{
    // nvec vectors of WORDS dwords; lane-contiguous
    unsigned int acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
        unsigned int v[UNROLL][WORDS];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned int *p = src + (i + u * stride) * WORDS;
            if (WORDS == 4) {
                const uint4 t = *reinterpret_cast<const uint4 *>(p);
                v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
            } else {
#pragma unroll
                for (int w = 0; w < WORDS; ++w) v[u][w] = p[w];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int w = 0; w < WORDS; ++w) acc ^= v[u][w];
    }
    for (; i < nvec; i += stride)
        for (int w = 0; w < WORDS; ++w) acc ^= src[i * WORDS + w];
    if (acc == 0x12345678u) sink[0] = acc;     // practically never; keeps the loads alive
}

__global__ __launch_bounds__(256) void k_probe_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long nvec)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_probe_write(uint4 *__restrict__ dst, long long nvec)
{
    const long long stride = (long long)gridDim.x * 256;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) dst[i] = v;
}

typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));

// Writes (or reads, READ = true) from the workgroups of ONE XCD only (workgroup i of a 1-D grid runs on XCD i % 8; xcd = 8:
// all of them): does it matter which XCD writes to a given piece of memory?  The participating workgroups cover the range.
template <bool READ>
__global__ __launch_bounds__(256) void k_probe_xcd(uint4 *__restrict__ dst, long long nvec, int xcd, unsigned int *sink)
{
    const unsigned int me = blockIdx.x & 7u;
    if (xcd < 8 && (int)me != xcd) return;
    const long long nwg = xcd < 8 ? gridDim.x / 8 : gridDim.x, wg = xcd < 8 ? blockIdx.x / 8 : blockIdx.x;
    const long long stride = nwg * 256;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3u, 4u);
    unsigned int acc = 0;
    for (long long i = wg * 256 + threadIdx.x; i < nvec; i += stride) {
        if (READ) { const uint4 r = dst[i]; acc ^= r.x ^ r.y ^ r.z ^ r.w; }
        else dst[i] = v;
    }
    if (READ && acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_probe_write_nt(pu32x4 *__restrict__ dst, long long nvec)
{
    const long long stride = (long long)gridDim.x * 256;
    const pu32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride)
        __builtin_nontemporal_store(v, dst + i);
}

// the fused kernel's traffic mix: 12 bytes read, 3 x 16 bytes written per lane and step
template <bool NT>
__global__ __launch_bounds__(256) void k_probe_mix(const unsigned int *__restrict__ src, pu32x4 *__restrict__ d0,
                                                   pu32x4 *__restrict__ d1, pu32x4 *__restrict__ d2, long long nquads)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += stride) {
        const unsigned int a = src[i * 3], b = src[i * 3 + 1], c = src[i * 3 + 2];
        const pu32x4 v0 = {a, b, c, a ^ b}, v1 = {b, c, a, b ^ c}, v2 = {c, a, b, c ^ a};
        if (NT) {
            __builtin_nontemporal_store(v0, d0 + i);
            __builtin_nontemporal_store(v1, d1 + i);
            __builtin_nontemporal_store(v2, d2 + i);
        } else {
            d0[i] = v0; d1[i] = v1; d2[i] = v2;
        }
    }
}

// the NDVI-plane mix: 12 bytes read, 16 bytes written per lane and step
__global__ __launch_bounds__(256) void k_probe_mix1(const unsigned int *__restrict__ src, pu32x4 *__restrict__ d0, long long nquads)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += stride) {
        const unsigned int a = src[i * 3], b = src[i * 3 + 1], c = src[i * 3 + 2];
        const pu32x4 v0 = {a, b, c, a ^ b};
        d0[i] = v0;
    }
}

// Two passes over the tiles in ONE launch, interleaved tile by tile in dispatch order: per tile t, `nh` read-only blocks that
// sweep tile t + 1 (the histogram pass's traffic) and then `nf` blocks that read tile t and write three planes (the fused pass's
// traffic).  Does the second read of a tile come out of the Infinity Cache when its first read was one tile period earlier?
// mode 0: as described; 1: the read-only blocks sweep a tile half a batch away (same traffic, the second read is cold);
// 2: no read-only blocks; 3: only the read-only blocks; 4: the read-only blocks sweep tile t itself (first read right before the second).  4096 x 4096 tiles, output ring of 64 tile slots.
__global__ __launch_bounds__(256) void k_probe_two_pass(const unsigned int *__restrict__ src, pu32x4 *__restrict__ dst, int ntiles, int nh,
                                                        int nf, int mode, unsigned int *__restrict__ sink)
{
    const long long QT = 4194304;                                // quads per tile
    const int per = nh + nf;
    const int t = (int)(blockIdx.x / (unsigned)per), j = (int)(blockIdx.x % (unsigned)per);
    const int tid = threadIdx.x;
    if (j < nh) {
        if (mode == 2) return;
        long long u = t + 1 < ntiles ? t + 1 : 0;
        if (mode == 1) u = (u + ntiles / 2) % ntiles;
        if (mode == 4) u = t;                                       // the sweep of a tile right before its own plane-writing blocks
        const unsigned int *p = src + u * QT * 3;
        const long long lo = QT * j / nh, hi = QT * (j + 1) / nh;
        unsigned int acc = 0;
        long long i = lo + tid;
        for (; i + 768 < hi; i += 1024) {
            unsigned int w[4][3];
#pragma unroll
            for (int k = 0; k < 4; ++k) { w[k][0] = p[(i + 256 * k) * 3]; w[k][1] = p[(i + 256 * k) * 3 + 1]; w[k][2] = p[(i + 256 * k) * 3 + 2]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= w[k][0] ^ w[k][1] ^ w[k][2];
        }
        for (; i < hi; i += 256) acc ^= p[i * 3] ^ p[i * 3 + 1] ^ p[i * 3 + 2];
        if (acc == 0x12345677u) sink[0] = acc;
    } else {
        if (mode == 3) return;
        const int jj = j - nh;
        const unsigned int *p = src + (long long)t * QT * 3;
        pu32x4 *d0 = dst + (long long)(t & 63) * 3 * QT, *d1 = d0 + QT, *d2 = d1 + QT;
        const long long lo = QT * jj / nf, hi = QT * (jj + 1) / nf;
        for (long long i = lo + tid; i < hi; i += 256) {
            const unsigned int a = p[i * 3], b = p[i * 3 + 1], c = p[i * 3 + 2];
            const pu32x4 v0 = {a, b, c, a ^ b}, v1 = {b, c, a, b ^ c}, v2 = {c, a, b, c ^ a};
            d0[i] = v0; d1[i] = v1; d2[i] = v2;
        }
    }
}

// ---------------------------------------------------------------------------
// Round 2: more shapes of the plane-writing kernel's 1 : 4 mix (per 4 pixels 12 B read, 3 x 16 B written).
// All move the same bytes as k_probe_mix; they differ in which lane touches which address when.
// ---------------------------------------------------------------------------
typedef unsigned int pu32x3 __attribute__((ext_vector_type(3)));

// lane-contiguous 16 pixels: 3 x 16 B loads of 48 contiguous bytes, 4 x 16 B stores of 64 contiguous bytes per plane
__global__ __launch_bounds__(256) void k_probe_lane16(const pu32x4 *__restrict__ src, pu32x4 *__restrict__ d0,
                                                      pu32x4 *__restrict__ d1, pu32x4 *__restrict__ d2, long long ngroups)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += stride) {
        const pu32x4 a = src[3 * i], b = src[3 * i + 1], c = src[3 * i + 2];
        d0[4 * i] = a; d0[4 * i + 1] = b; d0[4 * i + 2] = c; d0[4 * i + 3] = a ^ b;
        d1[4 * i] = b; d1[4 * i + 1] = c; d1[4 * i + 2] = a; d1[4 * i + 3] = b ^ c;
        d2[4 * i] = c; d2[4 * i + 1] = a; d2[4 * i + 2] = b; d2[4 * i + 3] = c ^ a;
    }
}

// wave-contiguous 1024 pixels per step.  LOAD16: 3 coalesced 16 B loads per lane (1 KiB per instruction);
// else 4 coalesced 12 B loads (768 B per instruction).  Stores: 4 x 1 KiB per plane, PLANE_MAJOR = one plane at a
// time (4 KiB bursts), else plane-interleaved.  SLAB: consecutive steps of a wave are consecutive in memory
// (each workgroup owns one contiguous slab) instead of grid-strided.
template <bool LOAD16, bool PLANE_MAJOR, bool SLAB>
__global__ __launch_bounds__(256) void k_probe_wave1k(const unsigned int *__restrict__ src, pu32x4 *__restrict__ d0,
                                                      pu32x4 *__restrict__ d1, pu32x4 *__restrict__ d2, long long nsteps)
{
    const unsigned int lane = threadIdx.x & 63u;
    const long long nwaves = (long long)gridDim.x * 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long per = (nsteps + nwaves - 1) / nwaves;
    long long st = SLAB ? wave * per : wave;
    const long long end = SLAB ? (st + per < nsteps ? st + per : nsteps) : nsteps;
    const long long inc = SLAB ? 1 : nwaves;
    for (; st < end; st += inc) {
        pu32x4 v[4];
        if (LOAD16) {
            const pu32x4 *p = reinterpret_cast<const pu32x4 *>(src) + st * 192 + lane;
            v[0] = p[0]; v[1] = p[64]; v[2] = p[128]; v[3] = v[0] ^ v[1];
        } else {
            // three dwords per lane (a 3-vector type would be padded to 16 bytes: never index memory with it)
            const unsigned int *p = src + (st * 256 + lane) * 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int a = p[192 * j], b = p[192 * j + 1], c = p[192 * j + 2];
                v[j] = (pu32x4){a, b, c, a ^ b};
            }
        }
        pu32x4 *o0 = d0 + st * 256 + lane, *o1 = d1 + st * 256 + lane, *o2 = d2 + st * 256 + lane;
        if (PLANE_MAJOR) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o0[64 * j] = v[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) o1[64 * j] = v[(j + 1) & 3];
#pragma unroll
            for (int j = 0; j < 4; ++j) o2[64 * j] = v[(j + 2) & 3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { o0[64 * j] = v[j]; o1[64 * j] = v[(j + 1) & 3]; o2[64 * j] = v[(j + 2) & 3]; }
        }
    }
}

// read-only counterpart of the wave-run shapes: a wave reads RUN x 768 contiguous bytes per step (12 B per lane and load)
template <int RUN>
__global__ __launch_bounds__(256) void k_probe_read_run(const unsigned int *__restrict__ src, long long nsteps, unsigned int *__restrict__ sink)
{
    const unsigned int lane = threadIdx.x & 63u;
    const long long nwaves = (long long)gridDim.x * 4;
    unsigned int acc = 0;
    for (long long st = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); st < nsteps; st += nwaves) {
        const unsigned int *p = src + (st * (64 * RUN) + lane) * 3;
        unsigned int v[RUN][3];
#pragma unroll
        for (int j = 0; j < RUN; ++j) { v[j][0] = p[192 * j]; v[j][1] = p[192 * j + 1]; v[j][2] = p[192 * j + 2]; }
#pragma unroll
        for (int j = 0; j < RUN; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the plain mix with each workgroup on its own contiguous slab of quads instead of a grid stride
__global__ __launch_bounds__(256) void k_probe_mix_slab(const unsigned int *__restrict__ src, pu32x4 *__restrict__ d0,
                                                        pu32x4 *__restrict__ d1, pu32x4 *__restrict__ d2, long long nquads)
{
    const long long per = ((nquads + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = lo + per < nquads ? lo + per : nquads;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const unsigned int a = src[i * 3], b = src[i * 3 + 1], c = src[i * 3 + 2];
        d0[i] = (pu32x4){a, b, c, a ^ b}; d1[i] = (pu32x4){b, c, a, b ^ c}; d2[i] = (pu32x4){c, a, b, c ^ a};
    }
}

// read phase / write phase: N grid-strided 12 B loads, then the 3 N stores (N = 4 or 8)
template <int N, bool PLANE_MAJOR>
__global__ __launch_bounds__(256) void k_probe_mix_phase(const unsigned int *__restrict__ src, pu32x4 *__restrict__ d0,
                                                         pu32x4 *__restrict__ d1, pu32x4 *__restrict__ d2, long long nquads)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (N - 1) * stride < nquads; i += N * stride) {
        pu32x3 t[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const unsigned int *q = src + (i + j * stride) * 3;
            t[j] = (pu32x3){q[0], q[1], q[2]};
        }
        if (PLANE_MAJOR) {
#pragma unroll
            for (int j = 0; j < N; ++j) d0[i + j * stride] = (pu32x4){t[j].x, t[j].y, t[j].z, t[j].x ^ t[j].y};
#pragma unroll
            for (int j = 0; j < N; ++j) d1[i + j * stride] = (pu32x4){t[j].y, t[j].z, t[j].x, t[j].y ^ t[j].z};
#pragma unroll
            for (int j = 0; j < N; ++j) d2[i + j * stride] = (pu32x4){t[j].z, t[j].x, t[j].y, t[j].z ^ t[j].x};
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                d0[i + j * stride] = (pu32x4){t[j].x, t[j].y, t[j].z, t[j].x ^ t[j].y};
                d1[i + j * stride] = (pu32x4){t[j].y, t[j].z, t[j].x, t[j].y ^ t[j].z};
                d2[i + j * stride] = (pu32x4){t[j].z, t[j].x, t[j].y, t[j].z ^ t[j].x};
            }
        }
    }
    for (; i < nquads; i += stride) {
        const unsigned int a = src[i * 3], b = src[i * 3 + 1], c = src[i * 3 + 2];
        d0[i] = (pu32x4){a, b, c, a ^ b}; d1[i] = (pu32x4){b, c, a, b ^ c}; d2[i] = (pu32x4){c, a, b, c ^ a};
    }
}

// three planes written, nothing read: the write side of the mix on its own (48 B per lane and step)
__global__ __launch_bounds__(256) void k_probe_write3(pu32x4 *__restrict__ d0, pu32x4 *__restrict__ d1, pu32x4 *__restrict__ d2,
                                                      long long nquads)
{
    const long long stride = (long long)gridDim.x * 256;
    const pu32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += stride) { d0[i] = v; d1[i] = v; d2[i] = v; }
}

// Infinity Cache (MALL) probes.  k_probe_resweep: `reps` read-only sweeps over the same `nvec` 12-byte vectors inside one
// launch (no barrier between sweeps: every vector is read again after about one sweep's worth of other reads).
// WRITE: each sweep also writes 4 x as many bytes as it reads to a destination that is never read (3 planes of 16-byte
// vectors, a different range every sweep) -- does the write stream evict the input from the cache?
template <bool WRITE, bool NT>
__global__ __launch_bounds__(256) void k_probe_resweep(const unsigned int *__restrict__ src, long long nquads, int reps,
                                                       pu32x4 *__restrict__ dst, long long dst_quads, unsigned int *__restrict__ sink)
{
    unsigned int acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    for (int r = 0; r < reps; ++r) {
        pu32x4 *d0 = dst + (long long)r * 3 * nquads % (dst_quads > 0 ? dst_quads : 1);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += stride) {
            const unsigned int a = src[i * 3], b = src[i * 3 + 1], c = src[i * 3 + 2];
            acc ^= a ^ b ^ c;
            if (WRITE) {
                const pu32x4 v0 = {a, b, c, a ^ b}, v1 = {b, c, a, b ^ c}, v2 = {c, a, b, c ^ a};
                if (NT) {
                    __builtin_nontemporal_store(v0, d0 + i);
                    __builtin_nontemporal_store(v1, d0 + nquads + i);
                    __builtin_nontemporal_store(v2, d0 + 2 * nquads + i);
                } else {
                    d0[i] = v0; d0[nquads + i] = v1; d0[2 * nquads + i] = v2;
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---------------------------------------------------------------------------
// instruction-issue probes: 16 independent chains of ONE instruction per loop
// trip, every lane of every wave.  tools/probe.py turns the time into cycles
// per wave64 instruction per SIMD (what the hot loops' VALU budget is made of).
// ---------------------------------------------------------------------------
typedef float pf32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k_probe_issue(int iters, unsigned int *__restrict__ sink)
{
    __shared__ unsigned int s_lds[64 * 64];
    float a[16];
    pf32x2 p[16];
    double d[16];
    unsigned int u[16];
    unsigned int cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        a[j] = 1.0f + (float)((threadIdx.x + j) & 7) * 0.125f;
        p[j] = (pf32x2){a[j], 0.5f * a[j]};
        d[j] = (double)a[j];
        u[j] = threadIdx.x * 2654435761u + j;
    }
    if (OP == 10) {
        // conflict-free pointer chase: row r, lane column l -> byte address of (r + 1, l)
        for (int i = threadIdx.x; i < 64 * 64; i += 256) s_lds[i] = ((((i >> 6) + 1) & 63) << 8) | ((i & 63) << 2);
#pragma unroll
        for (int j = 0; j < 16; ++j) u[j] = (j << 8) | ((threadIdx.x & 63u) << 2);
        __syncthreads();
    }
    const float one = 1.0f, thr = 0.2f;
    const double dOne = 1.0, dA = 0.0;
    double dt[4] = {0, 0, 0, 0};
    const unsigned int lane_off4 = (threadIdx.x & 63u) << 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(one));
            else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[j]) : "v"(p[(j + 1) & 15]));
            else if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
            else if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 15]));
            else if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(a[j]));
            else if (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(lane_off4), "s"(0x0c0c0500u));
            else if (OP == 6) {
                unsigned int c;
                asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_bcnt1_i32_b64 %0, vcc" : "=s"(c) : "s"(thr), "v"(a[j]) : "vcc", "scc");
                cnt += c;
            }
            else if (OP == 7) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(a[(j + 1) & 15]), "v"(a[(j + 2) & 15]));
            else if (OP == 8) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[j]) : "v"(u[j]));
            else if (OP == 9) asm volatile("v_fma_mix_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(one));
            else if (OP == 10) {
                asm volatile("ds_read_b32 %0, %0" : "+v"(u[j]));
                if (j == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            else if (OP == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "s"(thr), "v"(a[j]) : "vcc");
            else if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[(j + 1) & 15]));
            else if (OP == 13) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[(j + 1) & 15]));
            else if (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(one));
            else if (OP == 15) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[j]) : "v"(d[(j + 1) & 15]));
            else if (OP == 16) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 15]));
            else if (OP == 17) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[j]) : : "vcc");
            else if (OP == 18) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 15]), "v"(lane_off4));
            else if (OP == 19) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(lane_off4));
            else if (OP == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 15]));
            else if (OP == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j]) : "v"(one));
            else if (OP == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(one));
            else if (OP == 23) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(u[j]));
            else if (OP == 24) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(lane_off4));
            else if (OP == 25) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[j]) : "v"(a[(j + 1) & 15]), "v"(a[(j + 2) & 15]));
            else if (OP == 26) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(a[(j + 1) & 15]), "v"(a[(j + 2) & 15]));
            else if (OP == 27) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[0]));
            else if (OP == 28) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[0]));
            else if (OP == 30) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[j]) : "s"(0x7F80u), "v"(lane_off4));
            else if (OP == 31) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(u[j]) : "v"(lane_off4));
            else if (OP == 32) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[j]));
            else if (OP == 33) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[j]));
            else if (OP == 34) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[j]) : "s"(0x7F80u));
            else if (OP == 35) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(u[j]) : "v"(7u));
            else if (OP == 36) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(u[j]) : "v"(lane_off4));
            else if (OP == 37) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 clamp" : "+v"(p[j]) : "v"(p[(j + 1) & 15]));
            else if (OP == 38) asm volatile("v_fma_f32 %0, %0, %1, %1 clamp" : "+v"(a[j]) : "v"(one));
            else if (OP == 39) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[j]) : "v"(lane_off4));
            else if (OP == 29) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(u[j]) : "s"(0x0c0c0500u));
            // does the matrix pipe take the float64 sum off the vector pipe?  40: the MFMA alone (acc += A x ones), 41: MFMA next to
            // the conversion that feeds it, 42: v_add_f64 next to the same conversion (compare 41 with 42 and with 4 alone)
            else if (OP == 40) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(dA), "v"(dOne));
            else if (OP == 41) {
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dt[j & 3]) : "v"(a[j]));
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(dt[j & 3]), "v"(dOne));
            }
            else if (OP == 42) {
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dt[j & 3]) : "v"(a[j]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(dt[j & 3]));
            }
        }
    }
    float fa = 0; double fd = 0; unsigned int fu = cnt;
#pragma unroll
    for (int j = 0; j < 16; ++j) { fa += a[j] + p[j].x + p[j].y; fd += d[j]; fu ^= u[j]; }
    if (fa == 123.456f && fd == 654.321 && fu == 77u) sink[0] = fu;
}

// shader clock against the 100 MHz constant-rate counter: out[0] = shader cycles, out[1] = 100 MHz ticks
__global__ void k_probe_clock(unsigned long long *out, int spin)
{
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long t0 = wall_clock64();
    float x = 1.0f;
    for (int i = 0; i < spin; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; out[2] = (unsigned long long)(x != 0.5f); }
}


// LDS atomics of the one-read statistics route in isolation (kinds 70..73): one workgroup of 1024 threads per CU with a 128 KiB
// table, four adds to random dwords per lane and step.  MODE 0: ds_add_u32 (no return), the product's form; 1: ds_add_rtn_u32, the
// returned low halves of the PREVIOUS step compared against a threshold (what a scan-free overflow check would cost); 2: MODE 0 plus
// the product's scan every 12 steps (two barriers around a sweep of the table); 3: ds_add_rtn_u32, returns unused but waited for.
template <int MODE>
__global__ __launch_bounds__(1024, 4) void k_probe_lds_atomics(int iters, unsigned int *sink)
{
    __shared__ __attribute__((aligned(16))) unsigned int s_tab[32768];
    __shared__ unsigned int s_hits;
    const int tid = threadIdx.x;
    uint4 *tab4 = reinterpret_cast<uint4 *>(s_tab);
    for (int i = tid; i < 8192; i += 1024) tab4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) s_hits = 0;
    __syncthreads();
    char *tab = reinterpret_cast<char *>(s_tab);
    unsigned int r = (unsigned)tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned int o0 = 0, o1 = 0, o2 = 0, o3 = 0, hits = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned int a[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r = r * 1664525u + 1013904223u;
            a[k] = (r >> 15) & 0x1FFFCu;
            v[k] = ((r & 0x100u) << 8) | 1u;
        }
        if (MODE == 1) {
            // the adds of the previous step have returned by now: did any of them push a low half to the threshold?
            const unsigned int m = (o0 | o1 | o2 | o3) & 0xC000u;
            if (m) hits += 1;
        }
        if (MODE == 1 || MODE == 3) {
            o0 = __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(tab + a[0]), v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            o1 = __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(tab + a[1]), v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            o2 = __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(tab + a[2]), v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            o3 = __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(tab + a[3]), v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(tab + a[k]), v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 2 && (it % 12) == 11) {
            __syncthreads();
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                const uint4 q = tab4[tid + i * 1024];
                if ((q.x | q.y | q.z | q.w) & 0xC000u) { atomicAdd(&s_hits, 1u); tab4[tid + i * 1024] = make_uint4(0u, 0u, 0u, 0u); }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    unsigned int acc = hits + o0 + o1 + o2 + o3;
    for (int i = tid; i < 32768; i += 1024) acc += s_tab[i];
    if (acc == 0x12345677u) sink[0] = acc + s_hits;
}

}  // namespace lars

using namespace lars;

template <int OP>
static void issue_launch(int iters, int blocks, unsigned int *sink, hipStream_t s)
{
    hipLaunchKernelGGL((k_probe_issue<OP>), dim3(blocks), dim3(256), 0, s, iters, sink);
}

// Shared readers (kind 30 + R, 40 + R): R workgroups of 1024 threads read the SAME chunk at the same moment -- block b = 8 R a + 8 r + x
// is reader r of chunk 8 a + x, so the readers of a chunk sit 8 apart in dispatch order (one XCD) -- with 128 KiB of LDS
// reserved so that one workgroup fills a CU, as the counting kernel of csrc/joint.hip does.  W dwords per lane and load (3 | 4),
// six loads in flight.  What a CU can take in from the XCD's L2 when the HBM side delivers each byte once.
template <int W, int D = 6>
__global__ __launch_bounds__(1024, 4) void k_probe_shared(const unsigned int *__restrict__ src, long long chunk_vecs, long long nchunks,
                                                          int readers, unsigned int *sink)
{
    __shared__ unsigned int s_pad[32768];
    const unsigned int b = blockIdx.x;
    const long long chunk = (long long)(b / (8u * readers)) * 8 + (b & 7u);
    if (chunk >= nchunks) return;
    if (threadIdx.x == 0) s_pad[b & 32767u] = b;
    // the chunk as a raw buffer: loads past its end return zeros (no access outside the allocation)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned int *>(src) + chunk * chunk_vecs * W, 0, (int)(chunk_vecs * W * 4), 0x00020000);
    typedef unsigned int v3 __attribute__((ext_vector_type(3)));
    typedef unsigned int v4 __attribute__((ext_vector_type(4)));
    const unsigned int voff = threadIdx.x * (W * 4u);
    constexpr unsigned int STEP_B = 1024u * W * 4u;
    auto load = [&](unsigned int soff) {
        v4 r;
        if constexpr (W == 3) { const v3 t = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff, 0); r = v4{t.x, t.y, t.z, 0u}; }
        else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        return r;
    };
    const long long nsteps = (chunk_vecs + 1023) / 1024;
    unsigned int acc = 0;
    v4 w[D];
#pragma unroll
    for (int k = 0; k < D; ++k) w[k] = load((unsigned)k * STEP_B);
    unsigned int soff = (unsigned)D * STEP_B;
    for (long long it = 0; it < nsteps; it += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            acc ^= w[k].x ^ w[k].y ^ w[k].z ^ w[k].w;
            __builtin_amdgcn_sched_barrier(0);
            w[k] = load(soff + (unsigned)k * STEP_B);
            __builtin_amdgcn_sched_barrier(0);
        }
        soff += (unsigned)D * STEP_B;
    }
    if (acc == 0x12345677u) sink[0] = acc + s_pad[1];
}

// kind: 0 read 16 B/lane, 1 read 12 B/lane, 2 copy 16 B/lane (bytes read + bytes written = 2*bytes), 3 write 16 B/lane,
// 4 non-temporal write, 5 / 6 the fused kernel's mix (12 B read + 48 B written per lane; plain / non-temporal stores),
// 7 the NDVI-plane mix (12 B read + 16 B written per lane); 8..19 round-2 shapes of the 12 B / 48 B mix (see the
// dispatch below; 19 writes the three planes without reading); 100 + op: instruction-issue probe (unroll = loop trips)
extern "C" int lars_d_probe(int kind, int unroll, int blocks, const void *src, void *dst, int64_t bytes, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (bytes <= 0 || blocks <= 0) return fail(LARS_ERR_INVALID, "lars_d_probe: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    LARS_TRY(scratch_reserve(c, 64));
    unsigned int *sink = static_cast<unsigned int *>(c->scratch);
    const unsigned int *p = static_cast<const unsigned int *>(src);
    if (kind >= 60 && kind <= 64) {
        // two passes interleaved tile by tile in one launch: unroll = read-only blocks per tile, blocks = plane-writing blocks per tile,
        // bytes = ntiles * 48 MiB of source; dst holds 64 tiles x 3 planes x 64 MiB
        const int ntiles = (int)(bytes / 50331648ll);
        if (ntiles < 2 || unroll < 1 || blocks < 1) return fail(LARS_ERR_INVALID, "lars_d_probe (two passes): arguments");
        const long long grid = (long long)ntiles * (unroll + blocks);
        if (grid > 0x7FFFFFFFll) return fail(LARS_ERR_INVALID, "lars_d_probe (two passes): grid");
        hipLaunchKernelGGL(k_probe_two_pass, dim3((unsigned)grid), dim3(256), 0, s, p, static_cast<pu32x4 *>(dst), ntiles, unroll, blocks,
                           kind - 60, sink);
        return launch_check("lars_d_probe (two passes)");
    }
    if (kind > 30 && kind < 50) {
        // 31..34: R = kind - 30 readers, 12 B per lane; 41..44: 16 B per lane; `blocks` = chunks
        const int wide = kind > 40, readers = kind - (wide ? 40 : 30);
        if (readers < 1 || readers > 4) return fail(LARS_ERR_INVALID, "lars_d_probe: 1..4 readers");
        const long long per = wide ? 16 : 12, chunk_vecs = bytes / per / blocks;
        const long long groups = (blocks + 7) / 8;
        const dim3 grid((unsigned)(groups * 8 * readers));
        // unroll = loads in flight per lane: 6 (default; 0 and 1 mean 6 too), 8, 12, 16
        const int depth = unroll <= 1 ? 6 : unroll;
#define LARS_SHARED(WW, DD) hipLaunchKernelGGL((k_probe_shared<WW, DD>), grid, dim3(1024), 0, s, p, chunk_vecs, (long long)blocks, readers, sink)
        if (wide) { if (depth == 6) LARS_SHARED(4, 6); else if (depth == 8) LARS_SHARED(4, 8); else if (depth == 12) LARS_SHARED(4, 12); else return fail(LARS_ERR_INVALID, "lars_d_probe: depth 6, 8 or 12 (16-byte loads)"); }
        else { if (depth == 6) LARS_SHARED(3, 6); else if (depth == 8) LARS_SHARED(3, 8); else if (depth == 12) LARS_SHARED(3, 12); else if (depth == 16) LARS_SHARED(3, 16); else return fail(LARS_ERR_INVALID, "lars_d_probe: depth 6, 8, 12 or 16"); }
#undef LARS_SHARED
        return launch_check("lars_d_probe (shared readers)");
    }
    if (kind >= 70 && kind <= 73) {
        // LDS atomics in isolation: unroll = steps (four adds per lane each), blocks = workgroups of 1024 threads
        if (kind == 70) hipLaunchKernelGGL((k_probe_lds_atomics<0>), dim3(blocks), dim3(1024), 0, s, unroll, sink);
        else if (kind == 71) hipLaunchKernelGGL((k_probe_lds_atomics<1>), dim3(blocks), dim3(1024), 0, s, unroll, sink);
        else if (kind == 72) hipLaunchKernelGGL((k_probe_lds_atomics<2>), dim3(blocks), dim3(1024), 0, s, unroll, sink);
        else hipLaunchKernelGGL((k_probe_lds_atomics<3>), dim3(blocks), dim3(1024), 0, s, unroll, sink);
        return launch_check("lars_d_probe (LDS atomics)");
    }
    if (kind == 0) {
        const long long n = bytes / 16;
        if (unroll >= 8) hipLaunchKernelGGL((k_probe_read<4, 8>), dim3(blocks), dim3(256), 0, s, p, n, sink);
        else if (unroll >= 4) hipLaunchKernelGGL((k_probe_read<4, 4>), dim3(blocks), dim3(256), 0, s, p, n, sink);
        else if (unroll >= 2) hipLaunchKernelGGL((k_probe_read<4, 2>), dim3(blocks), dim3(256), 0, s, p, n, sink);
        else hipLaunchKernelGGL((k_probe_read<4, 1>), dim3(blocks), dim3(256), 0, s, p, n, sink);
    } else if (kind == 1) {
        const long long n = bytes / 12;
        if (unroll >= 8) hipLaunchKernelGGL((k_probe_read<3, 8>), dim3(blocks), dim3(256), 0, s, p, n, sink);
        else if (unroll >= 4) hipLaunchKernelGGL((k_probe_read<3, 4>), dim3(blocks), dim3(256), 0, s, p, n, sink);
        else if (unroll >= 2) hipLaunchKernelGGL((k_probe_read<3, 2>), dim3(blocks), dim3(256), 0, s, p, n, sink);
        else hipLaunchKernelGGL((k_probe_read<3, 1>), dim3(blocks), dim3(256), 0, s, p, n, sink);
    } else if (kind == 2) {
        hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, s, static_cast<const uint4 *>(src),
                           static_cast<uint4 *>(dst), (long long)(bytes / 16));
    } else if (kind == 3) {
        hipLaunchKernelGGL(k_probe_write, dim3(blocks), dim3(256), 0, s, static_cast<uint4 *>(dst), (long long)(bytes / 16));
    } else if (kind == 4) {
        hipLaunchKernelGGL(k_probe_write_nt, dim3(blocks), dim3(256), 0, s, static_cast<pu32x4 *>(dst), (long long)(bytes / 16));
    } else if (kind == 5 || kind == 6) {
        // src: bytes/5 read as 12-byte quads; dst: three planes of 16-byte vectors (total traffic = bytes)
        const long long nquads = bytes / 60;
        pu32x4 *d = static_cast<pu32x4 *>(dst);
        if (kind == 5)
            hipLaunchKernelGGL((k_probe_mix<false>), dim3(blocks), dim3(256), 0, s, p, d, d + nquads, d + 2 * nquads, nquads);
        else
            hipLaunchKernelGGL((k_probe_mix<true>), dim3(blocks), dim3(256), 0, s, p, d, d + nquads, d + 2 * nquads, nquads);
    } else if (kind == 7) {
        const long long nquads = bytes / 28;               // total traffic = bytes
        hipLaunchKernelGGL(k_probe_mix1, dim3(blocks), dim3(256), 0, s, p, static_cast<pu32x4 *>(dst), nquads);
    } else if (kind >= 8 && kind <= 19) {
        // round-2 shapes of the 12 B-read / 48 B-written mix: total traffic = bytes (bytes / 5 read, 4 * bytes / 5 written)
        const long long nquads = bytes / 60 / 256 * 256;                   // whole 1024-pixel steps
        pu32x4 *d = static_cast<pu32x4 *>(dst);
        pu32x4 *d1 = d + nquads, *d2 = d + 2 * nquads;
        const dim3 g(blocks), b(256);
        switch (kind) {
        case 8: hipLaunchKernelGGL(k_probe_lane16, g, b, 0, s, reinterpret_cast<const pu32x4 *>(p), d, d1, d2, nquads / 4); break;
        case 9: hipLaunchKernelGGL((k_probe_wave1k<true, false, false>), g, b, 0, s, p, d, d1, d2, nquads / 256); break;
        case 10: hipLaunchKernelGGL((k_probe_wave1k<true, true, false>), g, b, 0, s, p, d, d1, d2, nquads / 256); break;
        case 11: hipLaunchKernelGGL((k_probe_wave1k<false, true, false>), g, b, 0, s, p, d, d1, d2, nquads / 256); break;
        case 12: hipLaunchKernelGGL((k_probe_wave1k<false, false, false>), g, b, 0, s, p, d, d1, d2, nquads / 256); break;
        case 13: hipLaunchKernelGGL((k_probe_wave1k<false, true, true>), g, b, 0, s, p, d, d1, d2, nquads / 256); break;
        case 14: hipLaunchKernelGGL(k_probe_mix_slab, g, b, 0, s, p, d, d1, d2, nquads); break;
        case 15: hipLaunchKernelGGL((k_probe_mix_phase<4, false>), g, b, 0, s, p, d, d1, d2, nquads); break;
        case 16: hipLaunchKernelGGL((k_probe_mix_phase<8, false>), g, b, 0, s, p, d, d1, d2, nquads); break;
        case 17: hipLaunchKernelGGL((k_probe_mix_phase<4, true>), g, b, 0, s, p, d, d1, d2, nquads); break;
        case 18: hipLaunchKernelGGL((k_probe_mix_phase<8, true>), g, b, 0, s, p, d, d1, d2, nquads); break;
        default: hipLaunchKernelGGL(k_probe_write3, g, b, 0, s, d, d1, d2, nquads); break;      // 19: 48 of 60 bytes move
        }
    } else if (kind >= 23 && kind <= 25) {
        // read-only wave runs: 23 = 4 x 768 B, 24 = 8 x 768 B, 25 = 1 x 768 B (the same loop without runs)
        const long long nquads = bytes / 12;
        if (kind == 23) hipLaunchKernelGGL((k_probe_read_run<4>), dim3(blocks), dim3(256), 0, s, p, nquads / 256, sink);
        else if (kind == 24) hipLaunchKernelGGL((k_probe_read_run<8>), dim3(blocks), dim3(256), 0, s, p, nquads / 512, sink);
        else hipLaunchKernelGGL((k_probe_read_run<1>), dim3(blocks), dim3(256), 0, s, p, nquads / 64, sink);
    } else if (kind >= 20 && kind <= 22) {
        // `bytes` = size of the re-read source range, `unroll` = sweeps; 21 / 22 also write 4 x bytes per sweep (plain /
        // non-temporal) into dst, which must hold at least 4 * bytes * min(unroll, 8) bytes (the sweeps rotate through it)
        const long long nquads = bytes / 12;
        const int reps = unroll < 1 ? 1 : unroll;
        const long long dst_quads = (long long)(reps < 8 ? reps : 8) * 3 * nquads;
        pu32x4 *d = static_cast<pu32x4 *>(dst);
        if (kind == 20) hipLaunchKernelGGL((k_probe_resweep<false, false>), dim3(blocks), dim3(256), 0, s, p, nquads, reps, d, 0ll, sink);
        else if (kind == 21) hipLaunchKernelGGL((k_probe_resweep<true, false>), dim3(blocks), dim3(256), 0, s, p, nquads, reps, d, dst_quads, sink);
        else hipLaunchKernelGGL((k_probe_resweep<true, true>), dim3(blocks), dim3(256), 0, s, p, nquads, reps, d, dst_quads, sink);
    } else if (kind == 26 || kind == 27) {
        // unroll = XCD (0..7) whose workgroups do the work, 8 = all; 26 writes dst, 27 reads it
        if (kind == 26) hipLaunchKernelGGL((k_probe_xcd<false>), dim3(blocks), dim3(256), 0, s, static_cast<uint4 *>(dst), (long long)(bytes / 16), unroll, sink);
        else hipLaunchKernelGGL((k_probe_xcd<true>), dim3(blocks), dim3(256), 0, s, static_cast<uint4 *>(dst), (long long)(bytes / 16), unroll, sink);
    } else if (kind == 99) {
        // shader clock: dst receives {shader cycles, 100 MHz ticks}; unroll = spin count; runs beside `blocks` busy blocks
        hipLaunchKernelGGL(k_probe_clock, dim3(blocks), dim3(256), 0, s, static_cast<unsigned long long *>(dst), unroll);
    } else if (kind >= 100 && kind < 143) {
        // instruction-issue probe: unroll = loop trips (16 instructions each), blocks of 4 waves
        typedef void (*fn_t)(int, int, unsigned int *, hipStream_t);
        static const fn_t table[43] = {issue_launch<0>, issue_launch<1>, issue_launch<2>, issue_launch<3>, issue_launch<4>,
                                       issue_launch<5>, issue_launch<6>, issue_launch<7>, issue_launch<8>, issue_launch<9>,
                                       issue_launch<10>, issue_launch<11>, issue_launch<12>, issue_launch<13>, issue_launch<14>,
                                       issue_launch<15>, issue_launch<16>, issue_launch<17>, issue_launch<18>, issue_launch<19>,
                                       issue_launch<20>, issue_launch<21>, issue_launch<22>, issue_launch<23>, issue_launch<24>,
                                       issue_launch<25>, issue_launch<26>, issue_launch<27>, issue_launch<28>, issue_launch<29>,
                                       issue_launch<30>, issue_launch<31>, issue_launch<32>, issue_launch<33>, issue_launch<34>,
                                       issue_launch<35>, issue_launch<36>, issue_launch<37>, issue_launch<38>, issue_launch<39>,
                                       issue_launch<40>, issue_launch<41>, issue_launch<42>};
        table[kind - 100](unroll, blocks, sink, s);
    } else {
        return fail(LARS_ERR_INVALID, "lars_d_probe: kind");
    }
    return launch_check("lars_d_probe");
}

extern "C" int lars_d_probe_mix3(const void *src, void *d0, void *d1, void *d2, int64_t nquads, int blocks, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (!src || !d0 || !d1 || !d2 || nquads <= 0 || blocks <= 0) return fail(LARS_ERR_INVALID, "lars_d_probe_mix3: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    hipLaunchKernelGGL((k_probe_mix<false>), dim3(blocks), dim3(256), 0, s, static_cast<const unsigned int *>(src), static_cast<pu32x4 *>(d0),
                       static_cast<pu32x4 *>(d1), static_cast<pu32x4 *>(d2), (long long)nquads);
    return launch_check("lars_d_probe_mix3");
}

