// Laboratory library (liblars_lab.so, `make lab`): experiments that are NOT part of the product -- streaming probes and
// allocation kinds.  It links against liblars_hip.so and is only loaded by tools/lab/ and tests/test_gpu_lab.py.
#pragma once
#include "../common.h"
#include "lars_lab.h"

namespace lars {
bool vmm_free(void *dptr);     // lab_alloc.cpp
}  // namespace lars
