"""Shared fixtures.  `gpu` marks tests that need a real MI355X (run via gpurun)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")
    # the shared library is git-ignored: build it when a fresh checkout has none (hipcc cross-compiles
    # gfx950 without a GPU; on the GPU box the .so travels with the snapshot)
    lib = os.path.join(ROOT, "lars_image_processing_amd", "liblars_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "lars_image_processing_amd", "csrc"), "-j4"])


@pytest.fixture(scope="session")
def golden():
    """Arrays the reference produced (tools/gen_golden.py), loaded without pickle."""
    with np.load(os.path.join(GOLDEN_DIR, "reference_outputs.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_dicts():
    with open(os.path.join(GOLDEN_DIR, "reference_dicts.json")) as fh:
        return json.load(fh)


def golden_case_names(kind=None):
    with np.load(os.path.join(GOLDEN_DIR, "reference_outputs.npz"), allow_pickle=False) as z:
        names = sorted({k.split("/")[0] for k in z.files if k.endswith("/input")})
    if kind:
        names = [n for n in names if n.startswith(kind)]
    return names


def golden_series():
    """The image_data dicts of tools/gen_golden.py's time series (tests/golden/timeframe_inputs.npz + the dates in
    reference_dicts.json), as the reference received them."""
    import datetime
    import json
    arrays = np.load(os.path.join(GOLDEN_DIR, "timeframe_inputs.npz"))
    with open(os.path.join(GOLDEN_DIR, "reference_dicts.json")) as fh:
        dicts = json.load(fh)["dicts"]
    series = []
    for i, (date, has_key) in enumerate(zip(dicts["timeframe/dates"], dicts["timeframe/has_corrected_key"])):
        d = {"metadata": {"upload_date": datetime.datetime.fromisoformat(date)}, "original": None, "array": arrays[f"img{i}/array"]}
        if has_key:
            d["corrected_array"] = arrays[f"img{i}/corrected_array"] if f"img{i}/corrected_array" in arrays.files else None
        series.append(d)
    return series
