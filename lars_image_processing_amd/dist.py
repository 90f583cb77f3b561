"""One process per GPU; global statistics over RCCL (xGMI).

The tile path shards with no data-path collective.  The only exchange is the
fold of the per-index statistics records: ``Comm.allreduce_stats`` -- one
``ncclAllGather`` of the packed ``lars_stats`` records (3 x 472 B per rank) and a
deterministic local fold in rank order, done inside ``liblars_hip.so``
(csrc/comm.cpp).  It is latency-bound (a few KB), so ring-vs-tree and per-link
xGMI bandwidth do not matter; scaling is decided by per-GPU HBM streaming.

Rendezvous: the 128-byte RCCL unique id travels through a file in a directory
shared by the ranks of one node (``torch.distributed.run`` exports RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT and a run id; none of torch is
imported here).

``TorchComm`` is the same interface over ``torch.distributed`` (backend ``nccl`` = RCCL with the
exchange buffers in HBM, or ``gloo`` on the host): for callers that already run inside a torch
process group, as the fallback of ``bench.py`` when the direct RCCL bootstrap fails, and for the
world_size-2 CPU tests.  It moves the same bytes and applies the same rank-order fold
(``fold_gathered``); only the transport differs.
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
import time

import numpy as np

from . import _ffi
from ._ffi import STATS_DTYPE
from .batch import merge_records


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


def _rendezvous_path():
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    port = os.environ.get("MASTER_PORT", "0")
    # the launcher's pid keeps back-to-back runs on one port apart
    boss = os.environ.get("LARS_RDZV_TOKEN") or str(os.getppid())
    d = os.environ.get("LARS_RDZV_DIR", tempfile.gettempdir())
    return os.path.join(d, f"lars_rccl_id_{run}_{port}_{boss}")


def _rccl_unique_id():
    buf = (C.c_uint8 * _ffi.COMM_ID_BYTES)()
    _ffi.call("lars_comm_unique_id", buf)
    return bytes(buf)


def exchange_unique_id(rank, world, timeout_s=120.0, make_id=_rccl_unique_id):
    """Rank 0 creates the RCCL unique id and publishes it atomically; the others poll."""
    path = _rendezvous_path()
    if rank == 0:
        data = make_id()
        assert len(data) == _ffi.COMM_ID_BYTES
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(data)
        os.replace(tmp, path)
        return data
    deadline = time.time() + timeout_s
    while time.time() < deadline:
        try:
            with open(path, "rb") as fh:
                data = fh.read()
            if len(data) == _ffi.COMM_ID_BYTES:
                return data
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout_s}s")


class Comm:
    """RCCL communicator of this process (rank = one GPU)."""

    def __init__(self, rank, world, unique_id):
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        uid = (C.c_uint8 * _ffi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        _ffi.call("lars_comm_init", C.byref(self._h), self.world, self.rank, uid)

    @classmethod
    def from_env(cls):
        """Bind GPU LOCAL_RANK and join the communicator described by the launcher's environment."""
        rank, local_rank, world = env_rank_world()
        _ffi.call("lars_set_device", local_rank)
        uid = exchange_unique_id(rank, world)
        comm = cls(rank, world, uid)
        comm.barrier()
        if rank == 0:
            try:
                os.unlink(_rendezvous_path())
            except OSError:
                pass
        return comm

    def allreduce_stats(self, records):
        """records: structured array [n] of STATS_DTYPE (host).  Returns the fold over all ranks."""
        rec = np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1).copy()
        _ffi.call("lars_comm_allreduce_stats", self._h, _ffi.ptr(rec), rec.size, 0, None)
        return rec

    def allreduce_f64(self, values, op="sum"):
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()
        _ffi.call("lars_comm_allreduce_f64", self._h, _ffi.ptr(v), v.size, {"sum": 0, "max": 1, "min": 2}[op])
        return v

    def barrier(self):
        _ffi.call("lars_comm_barrier", self._h)

    def destroy(self):
        if self._h:
            _ffi.load().lars_comm_destroy(self._h)
            self._h = C.c_void_p()


class SingleProcessComm:
    """world_size == 1: no RCCL needed."""
    rank, world = 0, 1

    def allreduce_stats(self, records):
        return np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1).copy()

    def allreduce_f64(self, values, op="sum"):
        return np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()

    def barrier(self):
        pass

    def destroy(self):
        pass


class TorchComm:
    """``Comm``'s interface over an initialised ``torch.distributed`` process group.

    ``device`` = where the exchange buffers live: ``"cuda:<LOCAL_RANK>"`` for the nccl (RCCL) backend,
    ``"cpu"`` for gloo.  The statistics exchange is one ``all_gather`` of the packed records followed by the
    fold ``lars_comm_allreduce_stats`` applies (rank order); ``allreduce_f64`` is one ``all_reduce``.
    """

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as td
        if not td.is_initialized():
            raise RuntimeError("TorchComm needs an initialised torch.distributed process group")
        self._torch, self._td, self._group = torch, td, group
        self.rank, self.world = td.get_rank(group), td.get_world_size(group)
        if device is None:
            device = f"cuda:{env_rank_world()[1]}" if td.get_backend(group) == "nccl" else "cpu"
        self.device = torch.device(device)
        self._owns_group = False

    @classmethod
    def from_env(cls, backend="nccl"):
        """Join the launcher's process group (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
        import torch
        import torch.distributed as td
        rank, local_rank, world = env_rank_world()
        if backend == "nccl":
            _ffi.call("lars_set_device", local_rank)
            torch.cuda.set_device(local_rank)
        if not td.is_initialized():
            td.init_process_group(backend, rank=rank, world_size=world)
        comm = cls(f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        comm._owns_group = True
        comm.barrier()
        return comm

    def allreduce_stats(self, records):
        torch, td = self._torch, self._td
        rec = np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1)
        mine = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        td.all_gather(gathered, mine, group=self._group)
        per_rank = [g.cpu().numpy().view(STATS_DTYPE) for g in gathered]
        return fold_gathered(per_rank)

    def allreduce_f64(self, values, op="sum"):
        torch, td = self._torch, self._td
        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()).to(self.device)
        td.all_reduce(t, op={"sum": td.ReduceOp.SUM, "max": td.ReduceOp.MAX, "min": td.ReduceOp.MIN}[op],
                      group=self._group)
        return t.cpu().numpy()

    def barrier(self):
        if self.device.type == "cuda":
            # an all_reduce on the device + the copy back orders this rank after every other rank's launch
            self.allreduce_f64([0.0])
        else:
            self._td.barrier(group=self._group)

    def destroy(self):
        if self._owns_group and self._td.is_initialized():
            self._td.destroy_process_group()
        self._owns_group = False


def fold_gathered(per_rank_records):
    """The fold ``lars_comm_allreduce_stats`` applies after its all-gather:
    ``per_rank_records[r][i]`` -> record i folded over ranks in rank order."""
    per_rank = [np.ascontiguousarray(r, dtype=STATS_DTYPE).reshape(-1) for r in per_rank_records]
    n = per_rank[0].size
    out = np.zeros(n, dtype=STATS_DTYPE)
    for i in range(n):
        out[i] = merge_records(np.array([r[i] for r in per_rank], dtype=STATS_DTYPE))
    return out
