// Roofline probes: what this device sustains for plain streaming reads / copies
// with the access shapes the hot path uses.  Reported by bench.py next to the
// kernel numbers ("verify the peak with a STREAM-like kernel").
#include "common.h"

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

}  // namespace lars

using namespace lars;

// kind: 0 read 16 B/lane, 1 read 12 B/lane, 2 copy 16 B/lane (bytes read + bytes written = 2*bytes), 3 write 16 B/lane,
// 4 non-temporal write, 5 / 6 the fused kernel's mix (12 B read + 48 B written per lane; plain / non-temporal stores),
// 7 the NDVI-plane mix (12 B read + 16 B written per lane)
extern "C" int lars_d_probe(int kind, int unroll, int blocks, const void *src, void *dst, int64_t bytes, void *stream)
{
    ThreadCtx *c;
    LARS_TRY(ensure_ctx(&c));
    if (bytes <= 0 || blocks <= 0) return fail(LARS_ERR_INVALID, "lars_d_probe: bad arguments");
    hipStream_t s = pick_stream(c, stream);
    LARS_TRY(scratch_reserve(c, 64));
    unsigned int *sink = static_cast<unsigned int *>(c->scratch);
    const unsigned int *p = static_cast<const unsigned int *>(src);
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
    } else {
        return fail(LARS_ERR_INVALID, "lars_d_probe: kind");
    }
    return launch_check("lars_d_probe");
}
