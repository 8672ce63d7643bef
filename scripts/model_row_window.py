"""CPU model: colour-gradient rows per lit segment of the flat (training) batch's backward when a wave merges the
contributions of W consecutive steps before emitting them (W = 1: what backward_replay_direct_kernel does -- the lanes
of one step that sit in the same cell share a row), per wave and per block of four waves, with every segment lit.
Per-ray cell sequences from the oracle (rfo_trace_paths), the kernels' slot order (scripts/model_train_batch.py).

  python scripts/model_row_window.py [--blocks 60]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from oracle import oracle as O  # noqa: E402
from radfoam_amd import foam  # noqa: E402
import bench  # noqa: E402
from model_train_batch import ray_order  # noqa: E402


def rows(cells, n, window, group):
    """cells [256, cap], n [256]: lockstep replay (hop i of every ray at step i); rows emitted when `group` consecutive
    lanes (64 = a wave, 256 = the block) merge `window` consecutive steps."""
    total = 0
    segs = int(n.sum())
    steps = int(n.max())
    for g0 in range(0, 256, group):
        c = cells[g0:g0 + group]
        nn = n[g0:g0 + group]
        for s0 in range(0, steps, window):
            blk = c[:, s0:s0 + window]
            live = (np.arange(s0, min(s0 + window, c.shape[1]))[None, :] < nn[:, None])
            total += np.unique(blk[:, :live.shape[1]][live]).size
    return total, segs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=60)
    ap.add_argument("--cap", type=int, default=384)
    args = ap.parse_args()
    fm = foam.make_synthetic_foam(2_000_000, 3, 5, cache_dir=foam.default_cache_dir())
    rays, start = bench.training_batch(fm, 1_000_000, 105)
    order = ray_order(rays, start)
    nblocks = order.size // 256
    pick = np.sort(np.random.default_rng(0).choice(nblocks, size=args.blocks, replace=False))
    rr = order[(pick[:, None] * 256 + np.arange(256)[None, :]).reshape(-1)]
    cells, t1, n = O.trace_paths(3, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"],
                                 rays[rr], start[rr], cap=args.cap)
    n = np.minimum(n, args.cap).astype(np.int64)
    out = {}
    for group in (64, 256):
        for w in (1, 2, 3, 4, 6, 8, 16):
            tot = seg = 0
            for b in range(pick.size):
                sl = slice(b * 256, (b + 1) * 256)
                r, s = rows(cells[sl].astype(np.int64), n[sl], w, group)
                tot += r
                seg += s
            out[f"lanes{group}_window{w}"] = round(tot / seg, 3)
            print(group, w, round(tot / seg, 3), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "model_row_window.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
