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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_one_read_kernels_keep_their_residency(tmp_path):
    """The one-read statistics kernels live by what fits a CU (profiles/r05_finish_block_timeline.txt): k_joint_finish must stay within 64
    registers so that TWO of its 16-wave blocks share a CU (at 68 its launch ran in two rounds: 97 instead of 74 us per 256 tiles); the
    counting kernels within 128 (16 waves of one workgroup per CU) and within the CU's 160 KiB of LDS: the full-table kernel 128 KiB + its
    list, the windowed kernel 306 rows of 133 dwords."""
    for src in ("joint.hip", "joint_win.hip"):
        out = tmp_path / (src + ".s")
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", f"-I{ROOT}/include",
               "-Wno-pass-failed", "-S", "--cuda-device-only", f"{ROOT}/lars_image_processing_amd/csrc/{src}", "-o", str(out)]
        subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    meta = {}
    for src in ("joint.hip", "joint_win.hip"):
        text = (tmp_path / (src + ".s")).read_text()
        for m in re.finditer(r"- \.agpr_count:.*?\n((?:    .*\n)+)", text):
            block = m.group(1)
            name = re.search(r"\.name:\s+(\S+)", block)
            if name:
                meta[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\d+)", block)}
    finish = [v for k, v in meta.items() if "k_joint_finish" in k]
    assert len(finish) == 1 and finish[0]["vgpr_count"] <= 64 and finish[0]["group_segment_fixed_size"] <= 24 * 1024, finish
    assert finish[0]["private_segment_fixed_size"] <= 16, finish                       # a register or two spilled, not an array
    full = {k: v for k, v in meta.items() if "k_joint_countILi" in k}
    win = {k: v for k, v in meta.items() if "k_joint_count_winILi" in k}
    assert len(full) >= 4 and len(win) >= 5, (sorted(full), sorted(win))
    for name, v in {**full, **win}.items():
        assert v["vgpr_count"] <= 128 and v["private_segment_fixed_size"] == 0, (name, v)
        assert v["group_segment_fixed_size"] <= 160 * 1024, (name, v)
    assert all(v["group_segment_fixed_size"] >= 306 * 133 * 4 for v in win.values())
