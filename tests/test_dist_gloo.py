"""N>1 path on CPU: world_size 2, 3 and 8 (the driver's node; one rank then owns no tile) under torch.distributed.run with the gloo backend."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_fold_matches_single_process(tmp_path, world):
    out = tmp_path / "result.json"
    env = dict(os.environ, LARS_RDZV_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out)]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["world"] == world and res["shards"][0][0] == 0 and res["shards"][-1][1] == 7
    assert set(res["summary"]) == {"NDVI", "GNDVI", "NDWI"}
    # the rendezvous file is per launch and lives in the directory we gave it
    assert any(name.startswith("lars_rccl_id_") for name in os.listdir(tmp_path))
