// Laboratory library (liblars_lab.so, `make lab`): experiments that are NOT part of the product -- streaming probes, the
// persistent one-launch pipeline, allocation kinds.  It links
// against liblars_hip.so and is only loaded by tools/lab/ and tests/test_gpu_lab.py; the product never reads its knobs.
#pragma once
#include "../common.h"
#include "lars_lab.h"

namespace lars {
struct LabTuning {
    int pipe_steps = 0;        // pipeline.hip: wave-steps per work item (0 = 64)
    int pipe_cold = 0;         // pipeline.hip timing experiment: fused items read a far-away tile (results are wrong)
    int pipe_trace = 0;        // pipeline.hip: record item timestamps behind the scratch's sync words
    int pipe_head = 0;         // pipeline.hip: histogram items handed out before each fused item (0 = 2)
};
LabTuning &lab_tuning();
bool vmm_free(void *dptr);     // lab_alloc.cpp
}  // namespace lars
