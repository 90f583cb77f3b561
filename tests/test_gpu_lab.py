"""The laboratory library (liblars_lab.so, include/lars_lab.h): experiments that are not part of the product stay
buildable -- the streaming probes and the allocation kinds (needs a MI355X)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "lab"))

@pytest.fixture(scope="module")
def lablib():
    import lablib as lab
    assert lab.available(), "build the laboratory library: make -C lars_image_processing_amd/csrc lab"
    lab.load()
    return lab


def test_streaming_probe_and_allocation_kinds(lablib):
    from lars_image_processing_amd import _ffi
    nbytes = 96 << 20
    src = _ffi.DeviceBuffer(nbytes)
    src.zero()
    dst = lablib.LabBuffer(nbytes, kind=0)
    lablib.probe(2, 1, 4096, src.ptr, dst.ptr, nbytes - nbytes % 960)         # copy, 16 bytes per lane
    lablib.probe(32, 1, 64, src.ptr, None, nbytes)                            # two readers per chunk, 12-byte loads
    lablib.probe(43, 1, 64, src.ptr, None, nbytes)                            # three readers, 16-byte loads
    _ffi.call("lars_synchronize", None)
    assert not dst.download(np.uint8, (1 << 20,)).any()                        # the copy of a zeroed buffer
    dst.free(); src.free()
