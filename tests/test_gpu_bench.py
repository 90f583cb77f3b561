"""bench.py end to end on a small batch: the JSON line's contract, and both transports of the statistics fold
(the library's own RCCL communicator and dist.TorchComm over torch.distributed's nccl backend) with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *args):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731",
               LARS_RDZV_TOKEN=f"bench{os.getpid()}", **extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--tiles", "6", "--tile", "512", "--ring", "4", "--steps", "2",
           "--warmup", "1", "--no-probe", "--placement-trials", "2", *(a for a in args if a != "--all-modes"),
           *(() if "--all-modes" in args else ("--no-all-modes",))]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def check_line(line, tiles=6, tile=512):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["unit"] == "Mpix/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["higher_is_better"] is True
    assert abs(line["value"] - tiles * tile * tile / (line["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    # the whole step against the same peak (histogram pre-pass, tables, folds included): algorithmic bytes / ms_per_step
    bpp = roof["algorithmic_bytes_per_pixel"]
    whole = tiles * tile * tile * bpp / (line["ms_per_step"] * 1e-3) / 1e9 / roof["peak"]
    assert abs(roof["whole_step_frac"] - whole) <= 1e-9 * whole and roof["whole_step_frac"] <= roof["frac"]
    # traffic is a PMC figure with its provenance, or null with the reason -- never a silent None
    assert isinstance(roof["traffic_source"], str) and roof["traffic_source"]
    if roof["traffic"] is None:
        assert roof["traffic_source"].startswith("null: ")
    else:
        assert roof["traffic_source"].startswith("profiles/traffic.json@") and roof["traffic"] > 0
    assert line["config"]["ranks_seen"] == line["n_gpus"]
    # what the line is and what ran it: the library's own RCCL communicator or not, which BASELINE configuration, a product build
    cfg = line["config"]
    assert cfg["rccl_used"] is cfg["collective"].startswith("RCCL ncclAllGather") and cfg["build_flags"] == 0
    assert cfg["baseline_config"].startswith("configs[1] per GPU") and "configs[3] = --gpus 8 --tiles 2048" in cfg["baseline_config"]
    for key in ("arena_allocations", "arena_transient_bytes", "arena_bytes"):
        assert key in cfg, key
    for name in ("NDVI", "GNDVI", "NDWI"):
        assert line["global_stats"][name]["count"] == tiles * tile * tile
    # the self-check after the timed region: records (and ring planes) of sampled tiles against single-tile runs
    ver = line["verified"]
    assert ver["ok"] is True and ver["ok_on_every_rank"] is True and ver["records"] is True and ver["planes"] is True
    assert ver["global_stats_identical_across_modes"] is True and len(ver["record_tiles"]) >= 2
    # one entry per rank, gathered over the communicator; the slowest rank is the line's step time
    assert [r["rank"] for r in line["ranks"]] == list(range(line["n_gpus"]))
    for r in line["ranks"]:
        for key in ("ms_per_step", "fused_ms", "hist_ms", "avg_launch_ms", "arena_ms", "arena_search_ms", "arena_post_free_ms",
                    "first_step_launch_ms_min_max"):
            assert key in r, key
        assert 0 < r["first_step_launch_ms_min_max"][0] <= r["first_step_launch_ms_min_max"][1]
        assert 0 < r["ms_per_step"] <= line["ms_per_step"] * (1 + 1e-9) and r["fused_ms"] > 0
    assert abs(max(r["ms_per_step"] for r in line["ranks"]) - line["ms_per_step"]) <= 1e-6 * line["ms_per_step"]
    # the fused launches of the first timed step one by one, next to what the arena probe predicted for them
    assert len(roof["first_step_launch_ms"]) == roof["launches_per_step"] and min(roof["first_step_launch_ms"]) > 0
    assert abs(sum(roof["first_step_launch_ms"]) / roof["launches_per_step"] - roof["avg_launch_ms"]) < 0.5 * roof["avg_launch_ms"]
    if line["config"]["arena"] and line["config"]["arena"].get("rejected"):
        assert roof["arena_probe_vs_steps"]["post_free_ms"] == line["config"]["arena"]["post_free_ms"] > 0
    assert "lars_d_stats_fold" in line["config"]["statistics_fold"] and line["config"]["stats_route"] in ("joint", "classic")
    assert abs(line["passes_ms"]["rest_of_step"] - (line["ms_per_step"] - line["passes_ms"]["histogram+tables"] - line["passes_ms"]["fused"])) < 1e-9


def test_bench_line_single_process():
    line = run_bench({}, "--cpu-tiles", "1", "--cpu-workers", "2")
    check_line(line)
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "Mpix/s" and cpu["value"] > 0
    # best of 3, per-step breakdown, the job of the like-for-like GPU mode
    assert len(cpu["runs_s"]) == 3 and abs(min(cpu["runs_s"]) - 512 * 512 / cpu["value"] / 1e6) < 1e-6 * min(cpu["runs_s"])
    assert sorted(cpu["steps_s_per_tile"]) == sorted(["wb", "index_NDVI", "index_GNDVI", "index_NDWI", "stats_NDVI", "stats_GNDVI", "stats_NDWI"])
    assert abs(sum(cpu["steps_s_per_tile"].values()) - min(cpu["runs_s"])) < 1e-9 and cpu["same_job_as_gpu_mode"] == "wb3idx_out_stats_medians"
    assert cpu["pool"]["cores"] == 2 and "why_not_all_cpus" in cpu["pool"]
    assert line["config"]["collective"].startswith("none")


def test_bench_all_modes_and_their_self_check():
    """Every mode of the line on a small batch: the like-for-like mode (planes + statistics + medians), the one-read route
    next to the two-pass route for the statistics-only modes, and the self-check of the modes with medians."""
    line = run_bench({}, "--no-cpu-baseline", "--all-modes", "--no-u16-leg")
    check_line(line)
    modes = line["modes"]
    for name in ("wb3idx_out_stats_hist", "wb_ndvi_out_stats", "wb3idx_stats_only", "wb_ndvi_stats_only", "wb3idx_stats_medians",
                 "wb3idx_out_stats_medians", "wb3idx_stats_only_classic", "wb_ndvi_stats_only_classic", "wb3idx_stats_medians_classic"):
        assert name in modes and modes[name]["Mpix_s"] > 0 and 0 < modes[name]["whole_step_frac"] < 1, name
    assert "one read" in modes["wb_ndvi_stats_only"]["route"] and "histogram pass" in modes["wb_ndvi_stats_only_classic"]["route"]
    for name in ("wb3idx_stats_only", "wb3idx_stats_medians"):
        # the bench's own (vegetation) tiles go on windowed tables, none is counted again -- from 2^20 pixels per tile on (the
        # default 4096 x 4096: every tile; this test's 512 x 512: none)
        big = line["config"]["tile"][0] * line["config"]["tile"][1] >= 1 << 20
        assert modes[name]["tiles_on_windowed_tables"] == (line["config"]["tiles_per_gpu"] if big else 0) and modes[name]["tiles_recounted"] == 0
    assert modes["wb_ndvi_stats_only"]["tiles_on_windowed_tables"] == 0
    for name in ("wb3idx_stats_medians", "wb3idx_out_stats_medians"):
        assert line["verified"][name] == {"records": True, "planes": (True if name == "wb3idx_out_stats_medians" else None), "medians": True, "ok": True}
    assert len(line["verified"]["modes_compared"]) == 10
    # the statistics-only jobs once more on image-like content: both routes timed, their records identical, auto's choice named
    for content in ("smooth_content", "natural_content", "natural_mid_content"):
        for name in ("wb_ndvi_stats_only", "wb3idx_stats_only", "wb3idx_stats_medians"):
            leg = modes[content][name]
            assert leg["records_identical"] is True and leg["auto_route"] in ("one read", "per pixel"), (content, name)
            assert leg["ms_per_step"] == (leg["one_read_ms"] if leg["auto_route"] == "one read" else leg["per_pixel_ms"]) > 0
    assert "one-read statistics pass" in modes["wb_ndvi_out_stats"]["route"]
    for name in ("NDVI", "GNDVI", "NDWI"):
        assert "median" in line["global_stats"][name]


@pytest.mark.parametrize("transport", ["rccl", "torch", "auto"])
def test_bench_statistics_fold_transports_agree(transport):
    """One rank through each transport (auto = the pre-flight vote, which librccl wins on this image): same global
    statistics as the single-process fold."""
    base = run_bench({}, "--no-cpu-baseline")
    line = run_bench({"LARS_FORCE_RCCL": "1", **({} if transport == "auto" else {"LARS_COMM": transport})}, "--no-cpu-baseline")
    check_line(line)
    assert ("torch.distributed" in line["config"]["collective"]) == (transport == "torch")
    assert line["config"]["rccl_used"] is (transport != "torch") and base["config"]["rccl_used"] is False
    assert line["global_stats"] == base["global_stats"]


@pytest.mark.parametrize("ranks,all_modes", [(2, False), (3, True), (4, False)])
def test_ranks_on_one_gpu_equal_one_process_over_all_tiles(tmp_path, ranks, all_modes):
    """The N > 1 flow end to end on real kernels: `ranks` ranks (all on GPU 0, statistics exchanged over gloo) with
    12 / ranks tiles each must report the global statistics and medians of one process over the same 12 tiles (rank r
    owns a contiguous block of the same counter-hash sequence; sums are exact, so even the means are identical).
    With N > 1 the line times the headline mode only (no `modes`, no global medians) unless --all-modes is given."""
    per_rank = 12 // ranks
    port = 29000 + os.getpid() % 2000
    env = dict(os.environ, LARS_COMM="gloo", LARS_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    common = ["--tile", "512", "--ring", "4", "--steps", "2", "--warmup", "1", "--no-probe", "--placement-trials", "2", "--no-cpu-baseline",
              *(["--all-modes", "--no-u16-leg"] if all_modes else [])]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port + ranks), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--tiles", str(per_rank), *common]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                                      # rank 0 alone prints
    two = json.loads(lines[0])
    cmd1 = [sys.executable, os.path.join(ROOT, "bench.py"), "--tiles", "12", *common, *([] if all_modes else ["--no-all-modes"])]
    out1 = subprocess.run(cmd1, env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")},
                          capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out1.returncode == 0, out1.stderr[-2000:]
    one = json.loads([ln for ln in out1.stdout.splitlines() if ln.startswith("{")][0])
    assert two["n_gpus"] == ranks and two["scaling"] == "weak" and "gloo" in two["config"]["collective"]
    assert two["config"]["rccl_used"] is False and two["config"]["baseline_config"].startswith("configs[1] per GPU")   # a rehearsal says so
    assert [r["rank"] for r in two["ranks"]] == list(range(ranks)) and two["verified"]["ok_on_every_rank"] is True
    assert abs(max(r["ms_per_step"] for r in two["ranks"]) - two["ms_per_step"]) <= 1e-6 * two["ms_per_step"]
    assert two["config"]["tiles_per_gpu"] == per_rank and one["config"]["tiles_per_gpu"] == 12
    assert two["global_stats"] == one["global_stats"]
    assert ("modes" in two) == all_modes
    for name in ("NDVI", "GNDVI", "NDWI"):
        assert two["global_stats"][name]["count"] == 12 * 512 * 512
        assert ("median" in two["global_stats"][name]) == all_modes
        if all_modes:
            assert two["global_stats"][name]["median"] == one["global_stats"][name]["median"]
    # every rank reports its arena: the probe's figure after the rejected candidates were freed, and its first step's launches
    for r in two["ranks"]:
        assert r["arena_post_free_ms"] > 0 and r["arena_rejected"] == 1 and 0 < r["first_step_launch_ms_min_max"][0]
    assert abs(two["value"] - 12 * 512 * 512 / (two["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * two["value"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_plain_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (both on GPU 0,
    statistics over gloo) and rank 0's line says so; global statistics equal one process over the same 12 tiles."""
    common = ["--tile", "512", "--ring", "4", "--steps", "2", "--warmup", "1", "--no-probe", "--placement-trials", "0",
              "--no-cpu-baseline", "--no-all-modes"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiles", "6", *common],
                         env=_clean_env(LARS_COMM="gloo", LARS_DEVICE="0"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["ranks_seen"] == 2 and two["config"]["launcher"] == "self"
    out1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiles", "12", *common], env=_clean_env(),
                          capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out1.returncode == 0, out1.stderr[-2000:]
    one = json.loads([ln for ln in out1.stdout.splitlines() if ln.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["config"]["launcher"] == "none"
    assert two["global_stats"] == one["global_stats"]


def test_more_ranks_than_gpus_fails_loudly():
    from lars_image_processing_amd import _ffi
    if _ffi.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--tiles", "2", "--tile", "256"],
                         env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
