#!/usr/bin/env python3
"""Ad-hoc parity check of one large odd-shaped image through the host API (35 Mpix; ~15 s of oracle time)."""
import sys, time, warnings
sys.path.insert(0, '.')
import numpy as np
import lars_image_processing_amd as lars
from oracle import index_oracle as orc
rng = np.random.default_rng(77)
img = rng.integers(0, 256, (5001, 7001, 3), dtype=np.uint8)
img[:, :, 2] = np.clip(img[:, :, 2].astype(int) // 2 + 100, 0, 255).astype(np.uint8)
t = time.time(); res = lars.process_image(img, want_hist=True); print("gpu", time.time() - t)
t = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    wb = orc.wb_closed_form(img)
print("oracle wb", time.time() - t)
assert np.array_equal(res["corrected"], wb)
for tname in ("NDVI", "GNDVI", "NDWI"):
    want = orc.index_app(wb, tname)
    assert np.array_equal(res["indices"][tname]["index"].view(np.uint32), want.view(np.uint32)), tname
    ws = orc.stats_app(want, tname)
    gs = res["indices"][tname]["stats"]
    for k, v in ws.items():
        if k.startswith("Mean"):
            assert abs(gs[k] - v) <= 1e-6 * max(abs(v), float(np.mean(np.abs(want)))), (k, gs[k], v)
        else:
            assert gs[k] == v, (k, gs[k], v)
    assert np.array_equal(res["indices"][tname]["hist"], orc.hist50(want))
print("big odd image OK", img.shape)
