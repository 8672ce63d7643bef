"""Timing of the GPU triangulation on the BASELINE foams (gpurun): kd-order, AABB tree, one Delaunay star per point
from scratch and with the previous lists as candidates (incremental), each checked against the cached Qhull CSR.
  python scripts/gpu_delaunay.py [points seed] ...      -> one JSON line per foam (also gpurun_out/delaunay.json)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radfoam_amd import foam, triangulation  # noqa: E402

args = [int(a) for a in sys.argv[1:]] or [500_000, 1, 2_000_000, 5]
out = []
for n, seed in zip(args[0::2], args[1::2]):
    fm = foam.make_synthetic_foam(n, 0, seed, cache_dir=foam.default_cache_dir())
    pts = torch.from_numpy(fm["points"]).cuda()

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        return r, best * 1e3

    shuffled = pts[torch.randperm(n, device="cuda", generator=torch.Generator("cuda").manual_seed(0))]
    (_, _), t_kd = timed(lambda: triangulation.kd_order(shuffled))
    tree, t_tree = timed(lambda: triangulation.build_aabb_tree(pts))
    (adj, off, stats), t_full = timed(lambda: triangulation.delaunay_adjacency(pts, tree), reps=2)
    ok = bool(np.array_equal(adj.cpu().numpy(), fm["point_adjacency"]) and
              np.array_equal(off.cpu().numpy(), fm["point_adjacency_offsets"]))
    # an optimiser step later: points moved by ~3 % of the spacing, old lists as candidates
    g = torch.Generator("cuda").manual_seed(1)
    moved = pts + 0.03 * (8.0 / n) ** (1 / 3) * torch.randn(pts.shape, device="cuda", generator=g)
    tree2 = triangulation.build_aabb_tree(moved)
    (adj2, off2, stats2), t_inc = timed(lambda: triangulation.delaunay_adjacency(moved, tree2, (adj, off)), reps=2)
    (adj3, off3, _), t_moved_full = timed(lambda: triangulation.delaunay_adjacency(moved, tree2), reps=1)
    same = bool(torch.equal(adj2.view(torch.int32), adj3.view(torch.int32)) and
                torch.equal(off2.view(torch.int32), off3.view(torch.int32)))
    line = dict(points=n, waves=os.environ.get("RF_DELAUNAY_WAVES", "default"), kd_order_ms=round(t_kd, 2),
                aabb_tree_ms=round(t_tree, 3), stars_ms=round(t_full, 1), equals_qhull=ok,
                incremental_ms=round(t_inc, 1), incremental_equals_full=same,
                changed_edges=int(adj2.numel() - (adj2.numel() == adj.numel() and
                                                  int((adj2.view(torch.int32) == adj.view(torch.int32)).sum()))),
                nodes_per_point=round(stats["tree_nodes_visited"] / n, 1),
                insertions_per_point=round(stats["insertions"] / n, 2), large_stars=stats["large_stars"],
                incremental_nodes_per_point=round(stats2["tree_nodes_visited"] / n, 1))
    print(json.dumps(line), flush=True)
    out.append(line)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/delaunay_%s.json" % os.environ.get("RF_DELAUNAY_WAVES", "default"), "w") as f:
    json.dump(out, f, indent=1)
