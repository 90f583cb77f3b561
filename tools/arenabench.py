#!/usr/bin/env python3
"""The headline configuration in N fresh processes, one after the other: does every process get a fast output arena?

    python tools/arenabench.py --runs 10 [--arena auto|plain] [-- extra bench.py flags]

Prints one row per process: roofline.frac, whole_step_frac, ms per fused launch and what the arena search did.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--arena", default="auto", help="auto, plain or slowest (bench.py --arena)")
    ap.add_argument("--trials", type=int, default=-1)
    ap.add_argument("rest", nargs="*")
    args = ap.parse_args()
    fracs = []
    for i in range(args.runs):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-all-modes", "--no-cpu-baseline",
               "--no-probe", "--no-verify", "--arena", args.arena, "--placement-trials", str(args.trials), *args.rest]
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900)
        if out.returncode != 0:
            print(f"run {i}: FAILED {out.stderr[-400:]}", flush=True)
            continue
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        a = line["config"]["arena"] or {}
        fracs.append(line["roofline"]["frac"])
        cands = " ".join(f"{x:.3f}" for x in a.get("candidate_ms", []))
        mallocs = " ".join(f"{x:.0f}" for x in a.get("malloc_ms", []))
        avg = line["roofline"]["avg_launch_ms"]
        post = a.get("post_free_ms") or 0.0
        first = line["roofline"].get("first_step_launch_ms") or [0.0]
        print(f"run {i:2d}: frac {line['roofline']['frac']:.4f}  whole step {line['roofline']['whole_step_frac']:.4f}  steps {avg:.3f} ms per launch  "
              f"probe chosen {a.get('chosen_ms') or 0:.3f}  after freeing {post:.3f} ({100 * (post / avg - 1) if post else 0:+.1f} % vs steps)  "
              f"first step {min(first):.3f}-{max(first):.3f}  search {a.get('search_ms') or 0:7.1f} ms  candidates [{cands}]  planes at {a.get('chosen_offsets_gib')} GiB  "
              f"hipMalloc ms [{mallocs}]  {a.get('kind')}",
              flush=True)
    if fracs:
        print(f"# {len(fracs)} processes: min {min(fracs):.4f}  median {sorted(fracs)[len(fracs) // 2]:.4f}  max {max(fracs):.4f}")


if __name__ == "__main__":
    main()
