"""CPU model of a STAGED walk for the training-shaped batch: every K hops the live rays are compacted
(ballot + prefix sum in a kernel) and RE-BINNED by the cell they are in, so that the lanes of a wave / the waves of a
block sit in the same few cells again -- against today's single launch, in which a ray keeps the thread slot the sort by
(entry cell, direction) gave it and the lanes of a wave drift apart.

Input: the exact per-ray cell sequences of bench.py's training batch (1 M shuffled rays of 8 cameras through the 2 M-point
foam) from the oracle (rfo_trace_paths).  Both schedules are run in lockstep (all lanes of a wave take a hop together) and
the same things are counted:

  wave_fetches     sum over (wave, hop) of the distinct cells the wave's live lanes sit in = requests for colour rows /
                   cell records / face blocks when only lanes that are in the same cell in the same instruction share a
                   fetch (what the texture path coalesces; with every segment lit each is a 196-byte row + 176 bytes of
                   cell record and face blocks);
  block_fetches    sum over (block, stage) of the distinct cells the block's rays touch during the stage = requests if
                   the block held everything it touches during a stage in LDS (an upper bound on what a block-level table
                   can do; for today's schedule a "stage" is the whole walk);
  lane_steps       live lane-hops (the work), wave_steps (issue time: a wave steps while any lane lives)

  python scripts/model_staged_walk.py [--rays 1000000] [--stages 8 16 32 64] [--threads 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from radfoam_amd import foam  # noqa: E402
import bench  # noqa: E402
from scripts.model_train_batch import ray_order  # noqa: E402


def count_schedule(cells, n, slot_rays, h0, h1, stats):
    """Lockstep walk of hops [h0, h1) by the rays `slot_rays` (thread slot s = ray slot_rays[s]; 64 slots a wave, 256 a
    block).  Adds to stats."""
    m = slot_rays.size
    wave = (np.arange(m, dtype=np.int64) // 64)
    block = (np.arange(m, dtype=np.int64) // 256)
    seen_block = []
    for h in range(h0, h1):
        live = n[slot_rays] > h
        if not live.any():
            break
        cur = cells[slot_rays[live], h].astype(np.int64)
        w = wave[live]
        stats["lane_steps"] += int(live.sum())
        stats["wave_steps"] += int(np.unique(w).size)
        stats["wave_fetches"] += int(np.unique((w << 32) | cur).size)
        seen_block.append((block[live] << 32) | cur)
    if seen_block:
        stats["block_fetches"] += int(np.unique(np.concatenate(seen_block)).size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--cap", type=int, default=320)
    ap.add_argument("--stages", type=int, nargs="+", default=[8, 16, 32, 64])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "model_staged_walk.json"))
    args = ap.parse_args()
    fm = foam.make_synthetic_foam(args.points, 3, args.seed, cache_dir=foam.default_cache_dir())
    rays, start = bench.training_batch(fm, args.rays, args.seed + 100)
    t0 = time.time()
    cells, _t1, n = O.trace_paths(3, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"],
                                  rays, start, cap=args.cap, num_threads=args.threads)
    del _t1
    n = np.minimum(n, args.cap).astype(np.int64)
    print(f"paths of {args.rays} rays in {time.time() - t0:.0f} s; mean scans per ray {n.mean():.1f}, longest {n.max()}", flush=True)
    # direction code of every ray (the second half of the kernels' sort key): rays of a cell that point the same way stay
    # together longest
    order0 = ray_order(rays, start)
    dir_rank = np.empty(args.rays, dtype=np.int64)
    dir_rank[order0] = np.arange(args.rays)
    result = {"rays": args.rays, "points": args.points, "mean_scans_per_ray": float(n.mean()), "schedules": {}}

    def finish(name, stats, extra=None):
        rec = dict(stats)
        rec["lane_utilisation"] = round(rec["lane_steps"] / (64.0 * rec["wave_steps"]), 4)
        rec["wave_fetches_per_visit"] = round(rec["wave_fetches"] / rec["lane_steps"], 4)
        rec["block_fetches_per_visit"] = round(rec["block_fetches"] / rec["lane_steps"], 4)
        if extra:
            rec.update(extra)
        result["schedules"][name] = rec
        print(name, json.dumps(rec), flush=True)

    # today: one launch, slots in the order of (entry cell, direction)
    stats = dict(lane_steps=0, wave_steps=0, wave_fetches=0, block_fetches=0)
    t0 = time.time()
    # (block_fetches over the whole walk of a block: per block, so the blocks are processed in groups to bound memory)
    group = 256 * 512
    for b0 in range(0, args.rays, group):
        count_schedule(cells, n, order0[b0:b0 + group], 0, int(n.max()), stats)
    finish("today: one launch, slots sorted by (entry cell, direction)", stats, {"seconds": round(time.time() - t0)})

    for K in args.stages:
        for key_name in ("cell", "cell+direction"):
            stats = dict(lane_steps=0, wave_steps=0, wave_fetches=0, block_fetches=0)
            rebinned = 0
            stages = 0
            t0 = time.time()
            for h0 in range(0, int(n.max()), K):
                live = np.nonzero(n > h0)[0]
                if live.size == 0:
                    break
                cur = cells[live, h0].astype(np.int64)
                if key_name == "cell":
                    o = np.argsort(cur, kind="stable")
                else:
                    o = np.lexsort((dir_rank[live], cur))
                slot_rays = live[o]
                rebinned += int(live.size)
                stages += 1
                for b0 in range(0, slot_rays.size, group):
                    count_schedule(cells, n, slot_rays[b0:b0 + group], h0, h0 + K, stats)
            finish(f"staged: re-binned by {key_name} every {K} hops", stats,
                   {"stages": stages, "rays_re_binned_total": rebinned, "rays_re_binned_per_ray": round(rebinned / args.rays, 2),
                    "seconds": round(time.time() - t0)})
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(result, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
