"""bench.py's own N-rank launcher, without a GPU: process management, the rendezvous file, and the loud failures."""
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

import bench
from lars_image_processing_amd import _ffi, dist


def test_gpus_n_without_devices_fails_loudly():
    if _ffi.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LARS_COMM")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr and not out.stdout.strip()


def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


RANK_OK = "import os; print('{\"rank\": %s}' % os.environ['RANK']) if os.environ['RANK'] == '0' else None"
RANK_FAIL = ("import os, sys, time\n"
             "r = int(os.environ['RANK'])\n"
             "if r == 1: sys.exit(7)\n"
             "time.sleep(120)\n")


def test_rank_processes_relay_rank0_stdout():
    out = bench.run_rank_processes(3, [sys.executable, "-c", RANK_OK], dict(os.environ, WORLD_SIZE="3"))
    assert out.strip() == '{"rank": 0}'


def test_one_failing_rank_ends_the_group():
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        bench.run_rank_processes(3, [sys.executable, "-c", RANK_FAIL], dict(os.environ, WORLD_SIZE="3"))
    assert "rank 1 exited with status 7" in str(e.value)
    assert time.time() - t0 < 60                      # the sleeping ranks were ended, not waited for


def test_rendezvous_file_is_private_and_tagged(tmp_path, monkeypatch):
    monkeypatch.setenv("LARS_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("LARS_RDZV_TOKEN", "tok-a")
    monkeypatch.setenv("WORLD_SIZE", "2")
    ident = os.urandom(_ffi.COMM_ID_BYTES)
    # a leftover at the name (another launch, or somebody else's file) is replaced, never written through
    path = dist._rendezvous_path()
    with open(path, "wb") as fh:
        fh.write(b"x" * 64)
    os.chmod(path, 0o666)
    assert dist.exchange_unique_id(0, 2, make_id=lambda: ident) == ident
    st = os.stat(path)
    assert (st.st_mode & 0o777) == 0o600
    assert dist.exchange_unique_id(1, 2, timeout_s=5) == ident
    # a reader of another launch (different token, same file name forced) does not accept it
    monkeypatch.setenv("LARS_RDZV_TOKEN", "tok-b")
    os.replace(path, dist._rendezvous_path())
    with pytest.raises(TimeoutError):
        dist.exchange_unique_id(1, 2, timeout_s=0.3)
    # neither a file that others may write, nor one from long before this process started
    monkeypatch.setenv("LARS_RDZV_TOKEN", "tok-a")
    os.replace(dist._rendezvous_path().replace("tok-a", "tok-b"), path)
    os.chmod(path, 0o622)
    with pytest.raises(TimeoutError):
        dist.exchange_unique_id(1, 2, timeout_s=0.3)
    os.chmod(path, 0o600)
    assert dist.exchange_unique_id(1, 2, timeout_s=5) == ident
    with pytest.raises(TimeoutError):
        dist.exchange_unique_id(1, 2, timeout_s=0.3, max_age_s=-3600)


def test_ranks_agree_on_the_transport_before_any_blocking_bootstrap(tmp_path, monkeypatch):
    """dist.agree: every rank learns whether ALL ranks can use librccl; one 'no' or one absent rank makes everybody say no."""
    import threading
    monkeypatch.setenv("LARS_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("LARS_RDZV_TOKEN", "vote")
    monkeypatch.setenv("WORLD_SIZE", "3")

    def vote(oks, timeout_s=20.0, absent=()):
        out = {}
        threads = [threading.Thread(target=lambda r=r: out.__setitem__(r, dist.agree(r, 3, oks[r], timeout_s)))
                   for r in range(3) if r not in absent]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dist.forget_agreement(3)
        return out

    assert vote([True, True, True]) == {0: True, 1: True, 2: True}
    assert vote([True, False, True]) == {0: False, 1: False, 2: False}
    assert vote([True, True, True], timeout_s=0.5, absent=(2,)) == {0: False, 1: False}
    assert not any(name.endswith((".pre0", ".pre1", ".pre2")) for name in os.listdir(tmp_path))
    # the second vote (did the bootstrap come up everywhere?) keeps its own markers: a 'no' there does not read a 'yes' of the first
    out = {}
    threads = [threading.Thread(target=lambda r=r: out.__setitem__(r, (dist.agree(r, 3, True, 20.0), dist.agree(r, 3, r != 1, 20.0, phase="up"))))
               for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert out == {r: (True, False) for r in range(3)}
    dist.forget_agreement(3)
    assert not [name for name in os.listdir(tmp_path) if ".pre" in name or ".up" in name]
    # Rank 0 is the LAST to vote and removes the markers as soon as it may: the others, still polling every 10 ms, must not
    # lose its marker -- release_agreement holds the deletion back behind a barrier of the agreed communicator.
    class ThreadComm:
        def __init__(self, barrier):
            self._b = barrier

        def barrier(self):
            self._b.wait(timeout=30)

    import time
    for _ in range(5):
        barrier, out = threading.Barrier(3), {}

        def rank_body(r):
            if r == 0:
                time.sleep(0.15)                                 # arrives last: sees every marker at its first scan
            ok = dist.agree(r, 3, True, 10.0, phase="up")
            dist.release_agreement(ThreadComm(barrier), r, 3)
            out[r] = ok
        threads = [threading.Thread(target=rank_body, args=(r,)) for r in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert out == {0: True, 1: True, 2: True}
        assert not [name for name in os.listdir(tmp_path) if ".pre" in name or ".up" in name]
    # the library's own pre-flight check answers without a GPU: librccl is part of the ROCm image
    assert _ffi.load().lars_comm_available() in (0, -6, -7, -8, -5, -4, -3)
