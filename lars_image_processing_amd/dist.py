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
import stat as _stat
import tempfile
import time

import numpy as np

from . import _ffi
from ._ffi import STATS_DTYPE
from .batch import merge_records


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


_PROCESS_START = time.time()
_MAGIC = b"LARSRDZV1"


def _rendezvous_path():
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    port = os.environ.get("MASTER_PORT", "0")
    # the launcher's pid keeps back-to-back runs on one port apart; bench.py's own launcher passes a random token
    boss = os.environ.get("LARS_RDZV_TOKEN") or str(os.getppid())
    d = os.environ.get("LARS_RDZV_DIR", tempfile.gettempdir())
    return os.path.join(d, f"lars_rccl_id_{run}_{port}_{boss}")


def _rccl_unique_id():
    buf = (C.c_uint8 * _ffi.COMM_ID_BYTES)()
    _ffi.call("lars_comm_unique_id", buf)
    return bytes(buf)


def _launch_tag():
    """What every rank of one launch knows and a file left behind by another launch does not (ranks are siblings)."""
    boss = os.environ.get("LARS_RDZV_TOKEN") or str(os.getppid())
    return "|".join([boss, os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"),
                     os.environ.get("WORLD_SIZE", "1")]).encode()


def exchange_unique_id(rank, world, timeout_s=120.0, make_id=_rccl_unique_id, max_age_s=900.0):
    """Rank 0 creates the RCCL unique id and publishes it atomically; the others poll.

    The file is created exclusively with mode 0600 (a file somebody else put at that name is removed first and never
    written through), carries a magic, the launch tag and rank 0's clock, and a reader only accepts a regular file it
    owns itself, not writable by others, with the right tag and younger than its own start by at most ``max_age_s``
    -- so neither a leftover of a crashed launch nor a file planted by another local user is taken for the id."""
    path = _rendezvous_path()
    tag = _launch_tag()
    if rank == 0:
        data = make_id()
        assert len(data) == _ffi.COMM_ID_BYTES
        payload = _MAGIC + len(tag).to_bytes(2, "little") + tag + int(time.time() * 1e3).to_bytes(8, "little") + data
        tmp = f"{path}.tmp{os.getpid()}"
        for leftover in (path, tmp):
            try:
                os.unlink(leftover)
            except FileNotFoundError:
                pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as fh:
            fh.write(payload)
        os.replace(tmp, path)
        return data
    deadline = time.time() + timeout_s
    want_len = len(_MAGIC) + 2 + len(tag) + 8 + _ffi.COMM_ID_BYTES
    while time.time() < deadline:
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        except (FileNotFoundError, OSError):
            time.sleep(0.01)
            continue
        with os.fdopen(fd, "rb") as fh:
            st = os.fstat(fh.fileno())
            blob = fh.read(want_len + 1)
        ok = (_stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o022) and len(blob) == want_len
              and blob.startswith(_MAGIC + len(tag).to_bytes(2, "little") + tag))
        if ok:
            stamp = int.from_bytes(blob[want_len - _ffi.COMM_ID_BYTES - 8:want_len - _ffi.COMM_ID_BYTES], "little") / 1e3
            if stamp >= _PROCESS_START - max_age_s:
                return blob[-_ffi.COMM_ID_BYTES:]
        time.sleep(0.01)
    raise TimeoutError(f"rank {rank}: no valid RCCL unique id at {path} after {timeout_s}s")


def agree(rank, world, ok, timeout_s=120.0, phase="pre"):
    """All ranks of one launch learn whether EVERY rank said ``ok`` -- before any of them enters a blocking collective
    bootstrap (``phase="pre"``), or after one (``phase="up"``: did it come up everywhere?).  Each rank drops a marker next
    to the rendezvous file (exclusive, mode 0600) and waits for the others'.  Returns True only if all ``world`` markers
    say ok; a rank that never shows up within ``timeout_s`` counts as not ok (every rank then times out alike and makes
    the same choice)."""
    base = _rendezvous_path()
    tag = _launch_tag() + b"|" + phase.encode()
    mine = f"{base}.{phase}{rank}"
    try:
        os.unlink(mine)
    except FileNotFoundError:
        pass
    fd = os.open(mine, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "wb") as fh:
        fh.write(_MAGIC + tag + (b"|ok" if ok else b"|no"))
    deadline = time.time() + timeout_s
    verdicts = {}
    while time.time() < deadline and len(verdicts) < world:
        for r in range(world):
            if r in verdicts:
                continue
            try:
                fd = os.open(f"{base}.{phase}{r}", os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            except OSError:
                continue
            with os.fdopen(fd, "rb") as fh:
                st = os.fstat(fh.fileno())
                blob = fh.read(4096)
            if (_stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o022)
                    and blob[:-3] == _MAGIC + tag and blob[-3:] in (b"|ok", b"|no")):
                verdicts[r] = blob[-3:] == b"|ok"
        if len(verdicts) < world:
            time.sleep(0.01)
    return len(verdicts) == world and all(verdicts.values())


def forget_agreement(world, phases=("pre", "up")):
    """Remove the markers of ``agree`` (rank 0, once every rank is past it -- e.g. after the communicator's first barrier)."""
    base = _rendezvous_path()
    for phase in phases:
        for r in range(world):
            try:
                os.unlink(f"{base}.{phase}{r}")
            except OSError:
                pass


def release_agreement(comm, rank, world, phases=("pre", "up")):
    """Every rank calls this once it is past its last ``agree``: the markers may only go when EVERY rank has left the
    polling loop (a rank that still waits for a marker somebody has deleted would time out and choose differently), so the
    ranks meet in a barrier of the communicator they agreed on, and only then does rank 0 remove the files."""
    comm.barrier()
    if rank == 0:
        forget_agreement(world, phases)


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

    def ranks_seen(self):
        """The rank count RCCL itself reports for this communicator (ncclCommCount)."""
        n = C.c_int(0)
        _ffi.call("lars_comm_count", self._h, C.byref(n))
        return int(n.value)

    def allreduce_stats(self, records):
        """records: structured array [n] of STATS_DTYPE (host).  Returns the fold over all ranks."""
        rec = np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1).copy()
        _ffi.call("lars_comm_allreduce_stats", self._h, _ffi.ptr(rec), rec.size, 0, None)
        return rec

    def allreduce_stats_device(self, records_dev, n=3, stream=None):
        """``records_dev``: a DeviceBuffer of ``n`` records (e.g. ``TileBatch.fold_stats``).  The all-gather reads and the
        fold over ranks lands in that device memory (``lars_comm_allreduce_stats(is_device=1)``); returns the folded records
        as a host array as well."""
        _ffi.call("lars_comm_allreduce_stats", self._h, C.c_void_p(records_dev.ptr), int(n), 1, stream)
        return records_dev.download(STATS_DTYPE, (int(n),))

    def allreduce_f64(self, values, op="sum"):
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()
        _ffi.call("lars_comm_allreduce_f64", self._h, _ffi.ptr(v), v.size, {"sum": 0, "max": 1, "min": 2}[op])
        return v

    def allgather_f64(self, values):
        """[world, len(values)]: every rank's vector (an all-reduce of a vector that is zero outside the rank's own row)."""
        return _allgather_by_sum(self, values)

    def barrier(self):
        _ffi.call("lars_comm_barrier", self._h)

    def destroy(self):
        if self._h:
            _ffi.load().lars_comm_destroy(self._h)
            self._h = C.c_void_p()


class SingleProcessComm:
    """world_size == 1: no RCCL needed."""
    rank, world = 0, 1

    def ranks_seen(self):
        return 1

    def allreduce_stats(self, records):
        return np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1).copy()

    def allreduce_stats_device(self, records_dev, n=3, stream=None):
        _ffi.call("lars_synchronize", stream)
        return records_dev.download(STATS_DTYPE, (int(n),))

    def allreduce_f64(self, values, op="sum"):
        return np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()

    def allgather_f64(self, values):
        return np.ascontiguousarray(values, dtype=np.float64).reshape(1, -1).copy()

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

    def ranks_seen(self):
        """Ranks that really answer: an all_reduce of ones over the group (not just the configured world size)."""
        return int(round(float(self.allreduce_f64([1.0], "sum")[0])))

    def allreduce_stats(self, records):
        torch, td = self._torch, self._td
        rec = np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1)
        mine = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        td.all_gather(gathered, mine, group=self._group)
        per_rank = [g.cpu().numpy().view(STATS_DTYPE) for g in gathered]
        return fold_gathered(per_rank)

    def allreduce_stats_device(self, records_dev, n=3, stream=None):
        """The device-resident records of this rank through the torch group: 3 x 472 bytes come to the host, the fold over
        ranks goes back into ``records_dev`` (as the RCCL path leaves it) and is returned."""
        _ffi.call("lars_synchronize", stream)
        out = self.allreduce_stats(records_dev.download(STATS_DTYPE, (int(n),)))
        records_dev.upload(out)
        return out

    def allgather_f64(self, values):
        return _allgather_by_sum(self, values)

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


def _allgather_by_sum(comm, values):
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    wide = np.zeros((comm.world, v.size), dtype=np.float64)
    wide[comm.rank] = v
    return np.asarray(comm.allreduce_f64(wide.reshape(-1), "sum")).reshape(comm.world, v.size)


def fold_gathered(per_rank_records):
    """The fold ``lars_comm_allreduce_stats`` applies after its all-gather:
    ``per_rank_records[r][i]`` -> record i folded over ranks in rank order."""
    per_rank = [np.ascontiguousarray(r, dtype=STATS_DTYPE).reshape(-1) for r in per_rank_records]
    n = per_rank[0].size
    out = np.zeros(n, dtype=STATS_DTYPE)
    for i in range(n):
        out[i] = merge_records(np.array([r[i] for r in per_rank], dtype=STATS_DTYPE))
    return out
