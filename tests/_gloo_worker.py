"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N CPU run (gloo).

Exercises everything of the N>1 path that is not a GPU kernel: the launcher
environment, the file rendezvous that carries the RCCL unique id, tile
sharding, the per-rank fold and the rank-order fold that
lars_comm_allreduce_stats applies after its all-gather.  Per-tile records come
from the oracle here (test infrastructure; on a GPU box they come from the
fused kernel).
"""
import json
import os
import sys
import warnings

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from lars_image_processing_amd import _ffi, batch, dist  # noqa: E402
from oracle import index_oracle as orc  # noqa: E402
from test_abi_cpu import to_records  # noqa: E402
from _select_stub import select_pass_on_planes  # noqa: E402

TYPES = ("NDVI", "GNDVI", "NDWI")


def tile_records(tile_ids, h, w):
    rec = np.zeros((len(tile_ids), 3), dtype=_ffi.STATS_DTYPE)
    for j, t in enumerate(tile_ids):
        img = orc.synth_tile_u8(77, t, h, w, profile="vegetation")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wb = orc.wb_app(img)
        for k, name in enumerate(TYPES):
            rec[j, k] = to_records([orc.tile_partials(orc.index_app(wb, name), name)], k)[0]
    return rec


def main():
    out_path = sys.argv[1]
    rank, local_rank, world = dist.env_rank_world()
    td.init_process_group("gloo", rank=rank, world_size=world)
    assert (td.get_rank(), td.get_world_size()) == (rank, world)

    # 1. the rendezvous that carries the RCCL unique id (a random stand-in id here)
    uid = dist.exchange_unique_id(rank, world, timeout_s=60, make_id=lambda: os.urandom(_ffi.COMM_ID_BYTES))
    t = torch.tensor(list(uid), dtype=torch.uint8)
    ref = t.clone()
    td.broadcast(ref, src=0)
    assert torch.equal(t, ref), "ranks disagree on the unique id"

    # 2. shard, process, fold locally, fold globally
    ntiles, h, w = 7, 40, 56
    lo, hi = batch.shard_range(ntiles, rank, world)
    mine = tile_records(list(range(lo, hi)), h, w)
    comm = dist.TorchComm()                     # dist.Comm's interface over the gloo group (RCCL on the GPU path)
    assert (comm.rank, comm.world, comm.device.type) == (rank, world, "cpu")
    local = batch.local_fold(mine) if hi > lo else np.zeros(3, dtype=_ffi.STATS_DTYPE)
    if hi == lo:                                  # a rank with no tiles contributes neutral records
        local["min"], local["max"] = np.inf, -np.inf
    glob = comm.allreduce_stats(local)
    comm.barrier()
    # the per-rank diagnostics of bench.py's line travel the same way: every rank's vector, in rank order, on every rank
    rows = comm.allgather_f64([rank, 10.0 * rank + 0.5, float(hi - lo)])
    assert rows.shape == (world, 3) and rows[:, 0].tolist() == list(range(world))
    assert rows[:, 1].tolist() == [10.0 * r + 0.5 for r in range(world)] and int(rows[:, 2].sum()) == ntiles

    # 3. every rank must hold the same bytes, equal to the single-process fold over all tiles
    everything = tile_records(list(range(ntiles)), h, w)
    want = batch.local_fold(everything)
    summary = {}
    for k, name in enumerate(TYPES):
        g, s = batch.summarize(glob[k]), batch.summarize(want[k])
        assert g["count"] == s["count"] == ntiles * h * w
        assert g["min"] == s["min"] and g["max"] == s["max"] and g["coverage"] == s["coverage"]
        assert np.array_equal(g["hist"], s["hist"])
        assert abs(g["mean"] - s["mean"]) <= 1e-15
        summary[name] = {"mean": g["mean"], "coverage": g["coverage"]}
    # 4. exact global medians: two select passes, histograms summed over ranks
    def planes_of(tile_ids):
        out = {name: [np.zeros(0, np.float32)] for name in TYPES}
        for t in tile_ids:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                wb = orc.wb_app(orc.synth_tile_u8(77, t, h, w, profile="vegetation"))
            for name in TYPES:
                out[name].append(orc.index_app(wb, name).ravel())
        return {name: np.concatenate(v) for name, v in out.items()}
    mine_planes, all_planes = planes_of(range(lo, hi)), planes_of(range(ntiles))
    values = batch.select_order_statistics(select_pass_on_planes([mine_planes["NDVI"], mine_planes["GNDVI"]]),
                                           mine_planes["NDVI"].size, comm)
    medians = batch.medians_from_pairs(values)
    for name in TYPES:
        assert medians[name] == float(np.median(all_planes[name])), name
        summary[name]["median"] = medians[name]
    blob = torch.from_numpy(glob.view(np.uint8).copy())
    ref = blob.clone()
    td.broadcast(ref, src=0)
    assert torch.equal(blob, ref), "global records differ between ranks"
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"world": world, "shards": [batch.shard_range(ntiles, r, world) for r in range(world)],
                       "summary": summary}, fh)
    td.destroy_process_group()


if __name__ == "__main__":
    main()
