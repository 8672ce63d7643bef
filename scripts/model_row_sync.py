"""CPU model: colour-gradient rows per lit segment of the flat batch's backward (rows merged over a window of W wave-steps, as
backward_replay_direct_kernel does) when the lanes of a wave are kept together IN DEPTH -- a lane takes its next hop only
while the segment it would cross starts within delta of the wave's laggard -- against the lockstep replay (hop i of every
ray at step i), and what that costs in wave-steps.  Every segment lit; cell sequences and exit depths from the oracle.

  python scripts/model_row_sync.py [--blocks 40]      -> gpurun_out/model_row_sync.json"""
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


def wave_rows(cells, t1, n, delta, window):
    """one wave: (rows emitted, segments, wave-steps)"""
    lanes = cells.shape[0]
    pos = np.zeros(lanes, dtype=np.int64)
    t0 = np.zeros(lanes)
    rows = steps = 0
    seen = set()
    in_window = 0
    alive = pos < n
    while alive.any():
        act = alive.copy()
        if delta is not None:
            act &= t0 <= t0[alive].min() + delta
        idx = np.nonzero(act)[0]
        for c in cells[idx, pos[idx]].tolist():
            seen.add(c)
        t0[idx] = np.maximum(t0[idx], t1[idx, pos[idx]])
        pos[idx] += 1
        steps += 1
        in_window += 1
        alive = pos < n
        if in_window == window or not alive.any():
            rows += len(seen)
            seen.clear()
            in_window = 0
    return rows, int(n.sum()), steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=40)
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
    cells = cells.astype(np.int64)
    t1 = np.where(np.isfinite(t1), t1, 1e30).astype(np.float64)
    spacing = (8.0 / 2_000_000) ** (1 / 3)
    out = {}
    for window in (4, 8):
        for delta in (None, 1.0, 2.0, 4.0, 8.0):
            rows = segs = steps = 0
            for w0 in range(0, cells.shape[0], 64):
                r, s, st = wave_rows(cells[w0:w0 + 64], t1[w0:w0 + 64], n[w0:w0 + 64],
                                     None if delta is None else delta * spacing, window)
                rows += r
                segs += s
                steps += st
            key = "window%d_%s" % (window, "lockstep" if delta is None else "delta%.0f" % delta)
            out[key] = dict(rows_per_segment=round(rows / segs, 3), wave_steps=steps)
            print(key, out[key], flush=True)
    base = out["window4_lockstep"]["wave_steps"]
    for k in out:
        out[k]["wave_steps_rel"] = round(out[k]["wave_steps"] / base, 3)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "model_row_sync.json"), "w"), indent=1)
    print(json.dumps({k: (v["rows_per_segment"], v["wave_steps_rel"]) for k, v in out.items()}))


if __name__ == "__main__":
    main()
