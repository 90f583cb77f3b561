"""The plane-writing kernels sit next to a register cliff: the headline instantiation (three planes + statistics, runs of 4096 pixels)
needs 253 of the 256 registers that still allow TWO resident waves per SIMD.  Round 4 added three scalar kernel arguments for a
laboratory experiment, the count went to 264, one wave was left -- and every output arena, fast or slow, ran at the slow class's
level (2.88-3.05 instead of 2.50 ms per 64-tile launch; NOTES.md).  This test compiles csrc/fused.hip to assembly for gfx950 (no GPU
needed) and reads the counts back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# mangled-name fragment of k_fused_u8c3<PIX, MASK, WB, STATS, CH> -> most registers it may use
BUDGET = {
    "k_fused_u8c3IhLj7ELb1ELi1ELi3E": 256,       # uint8, three planes + statistics (BASELINE configs[1]): two waves per SIMD
    "k_fused_u8c3IhLj7ELb1ELi0ELi3E": 168,       # uint8, three planes, no statistics: three waves per SIMD (amdgpu_waves_per_eu)
    "k_fused_u8c3IhLj1ELb1ELi0ELi3E": 168,       # uint8, NDVI plane only
    "k_fused_u8c3ItLj1ELb1ELi1ELi3E": 256,       # uint16, NDVI + RGBA + statistics (configs[4])
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_plane_writing_kernels_keep_their_resident_waves(tmp_path):
    out = tmp_path / "fused.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", f"-I{ROOT}/include",
           "-Wno-pass-failed", "-S", "--cuda-device-only", f"{ROOT}/lars_image_processing_amd/csrc/fused.hip", "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text()
    counts = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        counts[m.group(1)] = int(m.group(2))
    assert counts, "no kernel metadata found in the assembly"
    for frag, budget in BUDGET.items():
        hits = {k: v for k, v in counts.items() if frag in k}
        assert hits, f"instantiation {frag} not found"
        for name, vgprs in hits.items():
            assert vgprs <= budget, f"{name}: {vgprs} registers > {budget}: it loses a resident wave per SIMD"
