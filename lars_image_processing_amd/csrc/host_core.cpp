// The parts of liblars_hip.so that never touch HIP: the thread-local error message, the ABI version and the host-side
// fold of statistics records.  Builds with hipcc into the library and with plain g++ under AddressSanitizer /
// UBSan (`make asan`, exercised by tests/test_asan_cpu.py in the build container).
#include <stdarg.h>

#include <cmath>
#include <string>

#include "host_common.h"

namespace lars {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace lars

using namespace lars;

extern "C" {

int lars_abi_version(void) { return LARS_ABI_VERSION; }
const char *lars_last_error(void) { return g_last_error.c_str(); }

// Fold records of one index (tiles of a batch, or ranks): sums in the given
// order (deterministic), min/max fold, integer fields add.
int lars_stats_merge(const lars_stats *r, int64_t n, lars_stats *out)
{
    if (!r || !out || n <= 0) return fail(LARS_ERR_INVALID, "lars_stats_merge: bad arguments");
    lars_stats m = r[0];
    for (int64_t i = 1; i < n; ++i) {
        m.sum += r[i].sum;
        m.sumsq += r[i].sumsq;
        m.count += r[i].count;
        m.above += r[i].above;
        m.nans += r[i].nans;
        m.min = std::fmin(m.min, r[i].min);
        m.max = std::fmax(m.max, r[i].max);
        for (int b = 0; b < LARS_HIST_BINS; ++b) m.hist[b] += r[i].hist[b];
    }
    *out = m;
    return LARS_OK;
}

}  // extern "C"
