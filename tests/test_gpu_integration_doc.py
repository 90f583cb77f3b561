"""The bare ctypes stub printed in INTEGRATION.md section 2 is executed as it stands (needs a MI355X): a maintainer who pastes it
gets the reference's results."""
import os
import re
import warnings

import numpy as np
import pytest

from conftest import ROOT
from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu


def test_the_stub_in_integration_md_runs_and_matches_the_oracle():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "C.CDLL(" in b]
    assert len(stub) == 1, "INTEGRATION.md section 2 holds one ctypes stub"
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                                            # the stub opens the library by its path inside the repository
    try:
        exec(compile(stub[0], "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(12)
    for img in (rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), rng.integers(0, 65536, (20, 31, 3), dtype=np.uint16)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_wb = orc.wb_app(img)
        got_wb = ns["fix_white_balance"](img)
        np.testing.assert_array_equal(got_wb, want_wb)
        for t in ("NDVI", "GNDVI", "NDWI"):
            want = orc.index_app(want_wb, t)
            got = ns["calculate_index"](want_wb, t)
            assert got.tobytes() == want.tobytes()
            ws, gs = orc.stats_app(want, t), ns["analyze_index"](want, t)
            assert list(gs) == list(ws)
            for k, v in ws.items():
                if k.startswith("Mean"):
                    assert abs(gs[k] - v) <= 1e-6 * max(abs(v), float(np.mean(np.abs(want))))
                else:
                    assert gs[k] == v, (t, k)
    assert ns["fix_white_balance"](None) is None and ns["analyze_index"](None, "NDVI") == {}
    with pytest.raises(ValueError, match="Unknown index type"):
        ns["calculate_index"](want_wb, "EVI")
