// Host-only plumbing shared by every translation unit of liblars_hip.so: status codes, the thread-local error message.
// No HIP here: host_core.cpp and tiff_codec.cpp build with a plain host compiler too (`make asan`: g++
// -fsanitize=address,undefined, the sanitizer target SURVEY.md section 5 asks for; never run on the GPU box).
#pragma once
#include <stdint.h>
#include <stdio.h>

#include "lars_hip.h"

namespace lars {

void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define LARS_TRY(expr)                \
    do {                              \
        int _s = (expr);              \
        if (_s != LARS_OK) return _s; \
    } while (0)

}  // namespace lars
