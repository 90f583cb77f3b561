// Host-side decoders for the image files either side of the path (no GPU work): TIFF LZW strips / tiles.
// Pillow, the reference's only decoder (backend-process.py:52), reduces three-sample 16-bit TIFFs to 8 bits; the
// package's own loader (lars_image_processing_amd/tiffio.py) keeps the full depth and calls this for the LZW flavour.
#include <string.h>

#include "host_common.h"

using namespace lars;

// TIFF 6.0 section 13: codes packed MSB first, 9 to 12 bits wide, 256 = Clear, 257 = EndOfInformation, first free
// code 258; the width grows one code early (when the table holds 2^width - 1 entries).
extern "C" int lars_h_tiff_lzw_decode(const uint8_t *src, int64_t nsrc, uint8_t *dst, int64_t ndst, int64_t *nout)
{
    if (!src || !dst || nsrc < 0 || ndst < 0) return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode: bad arguments");
    enum { CLEAR = 256, EOI = 257, FIRST = 258, MAXCODES = 4096 };
    static thread_local unsigned short prefix[MAXCODES];
    static thread_local unsigned char suffix[MAXCODES], first[MAXCODES];
    static thread_local unsigned short length[MAXCODES];
    for (int i = 0; i < 256; ++i) { prefix[i] = 0xFFFF; suffix[i] = first[i] = (unsigned char)i; length[i] = 1; }
    if (nsrc >= 2 && src[0] == 0x00 && (src[1] & 1))
        return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode: old-style (LSB-first) LZW is not supported");
    unsigned long long acc = 0;
    int nbits = 0, width = 9, next = FIRST, prev = -1;
    int64_t ip = 0, op = 0;
    for (;;) {
        while (nbits < width && ip < nsrc) { acc = (acc << 8) | src[ip++]; nbits += 8; }
        if (nbits < width) break;                            // ran out of input without EOI: what was decoded stands
        const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1u));
        nbits -= width;
        if (code == EOI) break;
        if (code == CLEAR) { width = 9; next = FIRST; prev = -1; continue; }
        int entry_len;
        if (prev < 0) {
            if (code >= 256) return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode: corrupt stream (code %d after Clear)", code);
            entry_len = 1;
            if (op + 1 > ndst) { op = ndst; break; }
            dst[op++] = (unsigned char)code;
            prev = code;
            continue;
        }
        if (code < next) {
            entry_len = length[code];
        } else if (code == next && next < MAXCODES) {
            entry_len = length[prev] + 1;
        } else {
            return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode: corrupt stream (code %d, table holds %d)", code, next);
        }
        // new table entry = string(prev) + first byte of this string
        const unsigned char head = code < next ? first[code] : first[prev];
        if (next < MAXCODES) {
            prefix[next] = (unsigned short)prev;
            suffix[next] = head;
            first[next] = first[prev];
            length[next] = (unsigned short)(length[prev] + 1);
            ++next;
        }
        // write the string backwards from its end (clipped to the output)
        int64_t end = op + entry_len;
        int c = code;
        int64_t at = end;
        while (c != 0xFFFF && at > op) {
            --at;
            if (at < ndst) dst[at] = suffix[c];
            c = prefix[c];
        }
        op = end;
        if (op >= ndst) { op = ndst; break; }
        prev = code;
        if (next == (1 << width) - 1 && width < 12) ++width;
    }
    if (nout) *nout = op;
    return LARS_OK;
}

// All strips / tiles of an image in one call: chunk i = file[offsets[i] .. offsets[i] + counts[i]) decodes into
// dst + i * chunk_bytes (at most chunk_bytes bytes; produced[i] = how many came out).  `threads` workers share the chunks.
#include <atomic>
#include <thread>
#include <vector>

extern "C" int lars_h_tiff_lzw_decode_chunks(const uint8_t *file, int64_t file_len, const uint64_t *offsets, const uint64_t *counts,
                                             int64_t nchunks, uint8_t *dst, int64_t chunk_bytes, int64_t *produced, int threads)
{
    if (!file || !offsets || !counts || !dst || !produced || nchunks <= 0 || chunk_bytes <= 0 || file_len <= 0)
        return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode_chunks: bad arguments");
    for (int64_t i = 0; i < nchunks; ++i)
        if (offsets[i] > (uint64_t)file_len || counts[i] > (uint64_t)file_len - offsets[i])
            return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode_chunks: chunk %lld lies outside the file", (long long)i);
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > nchunks) threads = (int)nchunks;
    std::atomic<int64_t> next{0};
    std::atomic<int64_t> bad{-1};
    auto work = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= nchunks) return;
            int64_t n = 0;
            if (lars_h_tiff_lzw_decode(file + offsets[i], (int64_t)counts[i], dst + i * chunk_bytes, chunk_bytes, &n) != LARS_OK) {
                int64_t none = -1;
                bad.compare_exchange_strong(none, i);
                n = 0;
            }
            produced[i] = n;
        }
    };
    if (threads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (bad.load() >= 0) return fail(LARS_ERR_INVALID, "lars_h_tiff_lzw_decode_chunks: corrupt LZW data in chunk %lld", (long long)bad.load());
    return LARS_OK;
}
