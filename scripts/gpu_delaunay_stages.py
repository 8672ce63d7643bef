"""Where the time of the first-pass star kernel goes (gpurun; A/B builds with -DRF_STAR_EXPERIMENT_STAGE=1 / 2 return early
from star_build, so their lists are WRONG on purpose and the binding's checks are expected to raise -- only the time counts).
  RADFOAM_HIP_LIB=radfoam_amd/libradfoam_hip_stage1.so python scripts/gpu_delaunay_stages.py [points seed]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radfoam_amd import foam, triangulation  # noqa: E402

n, seed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2_000_000, 5)
fm = foam.make_synthetic_foam(n, 0, seed, cache_dir=foam.default_cache_dir())
pts = torch.from_numpy(fm["points"]).cuda()
adj = torch.from_numpy(fm["point_adjacency"]).cuda()
off = torch.from_numpy(fm["point_adjacency_offsets"]).cuda()
g = torch.Generator("cuda").manual_seed(1)
moved = pts + 0.03 * (8.0 / n) ** (1 / 3) * torch.randn(pts.shape, device="cuda", generator=g)
tree, tree2 = triangulation.build_aabb_tree(pts), triangulation.build_aabb_tree(moved)


def timed(fn, reps=3):
    best, err = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        try:
            fn()
        except Exception as e:   # the early-return builds fail the symmetry check: expected
            err = type(e).__name__
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return round(best * 1e3, 1), err


print(json.dumps(dict(lib=os.path.basename(os.environ.get("RADFOAM_HIP_LIB", "libradfoam_hip.so")), points=n,
                      from_scratch_ms=timed(lambda: triangulation.delaunay_adjacency(pts, tree)),
                      incremental_ms=timed(lambda: triangulation.delaunay_adjacency(moved, tree2, (adj, off))))), flush=True)
