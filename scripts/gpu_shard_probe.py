"""How long the walk kernels take on a row block of the north-star frame as a function of the block's height and place:
the staircase behind the N-GPU strong-scaling step.  For every (first row, rows) the forward and the
backward of that block are timed with HIP events (median of --reps launches).  One JSON line per block.
usage: python scripts/gpu_shard_probe.py [--starts 0 464] [--rows 16 32 48 64 96 112 128 144 160 192 256]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radfoam  # noqa: E402
from radfoam_amd import foam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--starts", type=int, nargs="+", default=[0, 464])
ap.add_argument("--rows", type=int, nargs="+", default=[16, 32, 48, 64, 80, 96, 112, 128, 136, 144, 160, 192, 256])
ap.add_argument("--reps", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda:0")
fm = foam.make_synthetic_foam(2_000_000, 2, 5, cache_dir=foam.default_cache_dir())
cam = foam.default_camera(1920, 1080)
rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
start_idx = foam.nearest_point(fm["points"], cam["position"])
start = torch.full(rays.shape[:-1], start_idx, dtype=torch.int64).to(torch.uint32).to(dev)
p, a = torch.from_numpy(fm["points"]).to(dev), torch.from_numpy(fm["attributes"]).to(dev)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
g = torch.randn(rays.shape[:-1] + (4,), generator=torch.Generator().manual_seed(1234)).to(dev)
pipe = radfoam.create_pipeline(2)
pipe.record_trail = True


def timed(fn):
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
    return float(np.median(ts))


pipe.trace_forward(p, a, adj, off, rays, start)
for b in args.starts:
    for rows in args.rows:
        e = min(b + rows, 1080)
        rr, ss, gg = rays[b:e].contiguous(), start[b:e].contiguous(), g[b:e].contiguous()
        st = {}

        def fwd():
            st["f"] = pipe.trace_forward(p, a, adj, off, rr, ss)

        def bwd():
            st["b"] = pipe.trace_backward(p, a, adj, off, rr, ss, st["f"]["rgba"], gg)

        tf, tb = timed(fwd), timed(bwd)
        ni = st["f"]["num_intersections"].reshape(e - b, -1).to(torch.int64)
        wave_max = ni.view((e - b) // 8, 8, -1, 8).amax(dim=(1, 3)) if (e - b) % 8 == 0 else None
        print(json.dumps({"first_row": b, "rows": e - b, "blocks": ((e - b + 15) // 16) * 120, "forward_ms": round(tf, 4),
                          "backward_ms": round(tb, 4), "mean_steps": round(float(ni.float().mean()), 1),
                          "longest_ray": int(ni.max()),
                          "mean_wave_steps": None if wave_max is None else round(float(wave_max.float().mean()), 1)}), flush=True)
