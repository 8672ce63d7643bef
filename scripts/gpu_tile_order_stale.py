"""How much a tile order learnt on ANOTHER camera is worth (Pipeline.tile_order_mode): the north-star frame traced with the
static dealing, with the order learnt on the same camera, and with orders learnt on the neighbouring cameras of bench.py's
orbit (45 degrees apart) and on the opposite one.  Forward + backward, median of --reps."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import radfoam  # noqa: E402
from radfoam_amd import foam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda:0")
fm = foam.make_synthetic_foam(2_000_000, 2, 5, cache_dir=foam.default_cache_dir())
p, a = torch.from_numpy(fm["points"]).to(dev), torch.from_numpy(fm["attributes"]).to(dev)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)


def frame(k):
    cam = bench.orbit_camera(1920, 1080, k)
    rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
    s = foam.nearest_point(fm["points"], cam["position"])
    return rays, torch.full(rays.shape[:-1], s, dtype=torch.int64).to(torch.uint32).to(dev)


g = torch.randn((1080, 1920, 4), generator=torch.Generator().manual_seed(1)).to(dev)
target_rays, target_start = frame(0)


def timed(pipe):
    st = {}

    def fwd():
        st["f"] = pipe.trace_forward(p, a, adj, off, target_rays, target_start)

    def bwd():
        pipe.trace_backward(p, a, adj, off, target_rays, target_start, st["f"]["rgba"], g)

    out = []
    for fn in (fwd, bwd):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append(round(float(np.median(ts)), 4))
    return out


res = {}
pipe = radfoam.create_pipeline(2)
pipe.record_trail = True
pipe.tile_order_mode = None
res["static"] = timed(pipe)
pipe.tile_order_mode = "auto"
res["learnt_on_this_camera"] = timed(pipe)
for k, name in ((1, "learnt_45_degrees_away"), (2, "learnt_90_degrees_away"), (4, "learnt_on_the_opposite_camera")):
    pipe = radfoam.create_pipeline(2)
    pipe.record_trail = True
    r, s = frame(k)
    pipe.trace_forward(p, a, adj, off, r, s)          # learns the orders of camera k
    learnt = pipe._tiles
    pipe._tile_cost_begin = lambda *a_, **k_: None    # ... and keeps them
    res[name] = timed(pipe)
    assert pipe._tiles is learnt
print(json.dumps({"forward_ms_backward_ms": res}))
