// Driver of the host-only sanitizer build (`make asan`): replays cases from a file through the functions that see
// untrusted bytes, every buffer allocated at its exact size on the heap so that AddressSanitizer sees a one-byte
// overrun, and prints one line per case (status, bytes produced, FNV-1a of the output) for the test to compare with the
// shipped library's answers.  Case file: records of { u32 kind, u32 a, u32 b, u32 nbytes, bytes[nbytes] }:
//   kind 0  lars_h_tiff_lzw_decode(src = bytes, dst of a bytes)
//   kind 1  lars_h_tiff_lzw_decode_chunks: bytes = { u64 offsets[a], u64 counts[a], file... }, chunk_bytes = b, 3 threads
//   kind 2  lars_stats_merge over nbytes / sizeof(lars_stats) records (a = how many to pass; 0 must be refused)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "host_common.h"

static unsigned long long fnv(const void *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s cases.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    unsigned int head[4];
    long ncase = 0;
    while (fread(head, sizeof head, 1, f) == 1) {
        const unsigned int kind = head[0], a = head[1], b = head[2], nbytes = head[3];
        unsigned char *src = static_cast<unsigned char *>(malloc(nbytes ? nbytes : 1));
        if (nbytes && fread(src, 1, nbytes, f) != nbytes) { fprintf(stderr, "truncated case file\n"); return 2; }
        if (kind == 0) {
            unsigned char *dst = static_cast<unsigned char *>(malloc(a ? a : 1));
            memset(dst, 0, a ? a : 1);
            int64_t n = -1;
            const int rc = lars_h_tiff_lzw_decode(src, nbytes, dst, a, &n);
            printf("%ld lzw rc=%d n=%lld h=%016llx\n", ncase, rc, (long long)(rc == 0 ? n : -1), rc == 0 ? fnv(dst, (size_t)n) : 0ull);
            free(dst);
        } else if (kind == 1) {
            const size_t table = (size_t)a * 16;
            if (table > nbytes) { fprintf(stderr, "bad chunk case\n"); return 2; }
            std::vector<uint64_t> offsets(a), counts(a);
            memcpy(offsets.data(), src, (size_t)a * 8);
            memcpy(counts.data(), src + (size_t)a * 8, (size_t)a * 8);
            const size_t file_len = nbytes - table;
            unsigned char *file = static_cast<unsigned char *>(malloc(file_len ? file_len : 1));
            memcpy(file, src + table, file_len);
            unsigned char *dst = static_cast<unsigned char *>(calloc((size_t)a * b ? (size_t)a * b : 1, 1));
            std::vector<int64_t> produced(a ? a : 1, -1);
            const int rc = lars_h_tiff_lzw_decode_chunks(file, (int64_t)file_len, offsets.data(), counts.data(), a, dst, b,
                                                         produced.data(), 3);
            printf("%ld chunks rc=%d h=%016llx p=%016llx\n", ncase, rc, rc == 0 ? fnv(dst, (size_t)a * b) : 0ull,
                   rc == 0 ? fnv(produced.data(), (size_t)a * 8) : 0ull);
            free(dst);
            free(file);
        } else if (kind == 2) {
            lars_stats out;
            memset(&out, 0, sizeof out);
            const int rc = lars_stats_merge(reinterpret_cast<const lars_stats *>(src), (int64_t)a, &out);
            printf("%ld merge rc=%d h=%016llx\n", ncase, rc, rc == 0 ? fnv(&out, sizeof out) : 0ull);
        } else {
            fprintf(stderr, "unknown case kind %u\n", kind);
            return 2;
        }
        if (kind <= 1 && lars_last_error() == nullptr) return 3;
        free(src);
        ++ncase;
    }
    fclose(f);
    printf("done %ld cases\n", ncase);
    return 0;
}
