"""Tile order (Pipeline.tile_order_mode) on a scene WITHOUT the benchmark frame's symmetry, in benchmark.py's call pattern:
every launch a new camera.

Scene: the north-star foam's points and triangulation with another density field -- two off-centre blobs and a thin slab,
empty elsewhere -- so that the cost map of a frame (the steps of every tile's longest ray) is lopsided and changes with
the camera.  Path: --cameras views on a tilted orbit, never the same one twice in a row.  Modes:

  static   the kernels' own dealing of tiles to the XCDs
  auto     the default: orders learnt from a forward, reused for the next launches over a frame of this shape and relearnt
           every tile_order_refresh (16) launches when the rays keep changing -- here the order is always another camera's
  prev     tile_order_refresh = 1: every launch learns, the next one (another camera) uses it

For each: trace_forward + trace_backward per camera (fp32, SH 2) and trace_benchmark per camera (fp16, SH 2), mean and
worst over the path.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` with --mode to get the HBM bytes
per launch beside the milliseconds (scripts/gpu_call.sh of the round did; profiles/README.md).
  python scripts/gpu_tile_order_asymmetric.py [--mode static|auto|prev|all] [--cameras 12]"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radfoam  # noqa: E402
from radfoam_amd import foam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="all")
ap.add_argument("--cameras", type=int, default=12)
ap.add_argument("--points", type=int, default=2_000_000)
ap.add_argument("--seed", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
fm = foam.make_synthetic_foam(args.points, 2, args.seed, cache_dir=foam.default_cache_dir())
pts = fm["points"]
att = fm["attributes"].copy()
sigma0 = float(np.median(att[att[:, -1] > 0, -1]))
blob = lambda c, r: np.exp(-((pts - np.asarray(c, np.float32)) ** 2).sum(1) / (2 * r * r))
dens = 2.0 * sigma0 * (blob((0.45, 0.2, 0.0), 0.18) + blob((-0.3, -0.35, 0.3), 0.14)) + \
    1.5 * sigma0 * (np.abs(pts[:, 1] + 0.6) < 0.04) * (np.abs(pts[:, 0]) < 0.7) * (np.abs(pts[:, 2]) < 0.7)
dens[dens < 0.02 * sigma0] = 0.0
att[:, -1] = dens.astype(np.float32)
p, a = torch.from_numpy(pts).to(dev), torch.from_numpy(att).to(dev)
a16 = a.to(torch.float16)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
W, H = 1920, 1080


def camera(k):
    az = 2 * math.pi * ((k * 5) % args.cameras) / args.cameras          # consecutive launches: 150 degrees apart
    el = 0.45 * math.sin(2.3 * k)
    pos = 3.0 * np.array([math.sin(az) * math.cos(el), math.sin(el), -math.cos(az) * math.cos(el)], np.float32)
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(np.array([0, 1, 0], np.float32), fwd)
    right /= np.linalg.norm(right)
    cam = foam.default_camera(W, H)
    cam.update(position=pos, forward=fwd.astype(np.float32), right=right.astype(np.float32),
               up=np.cross(fwd, right).astype(np.float32))
    return cam


cams = [camera(k) for k in range(args.cameras)]
frames = []
for cam in cams:
    rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
    s = foam.nearest_point(pts, cam["position"])
    frames.append((rays, torch.full(rays.shape[:-1], s, dtype=torch.int64).to(torch.uint32).to(dev),
                   {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()},
                   torch.tensor([s], dtype=torch.int64).to(torch.uint32).to(dev)))
g = torch.randn((H, W, 4), generator=torch.Generator().manual_seed(1)).to(dev)


def run(mode):
    pipe, rend = radfoam.create_pipeline(2), radfoam.create_pipeline(2, torch.float16)
    for q in (pipe, rend):
        q.record_trail = True
        q.tile_order_mode = None if mode == "static" else "auto"
        if mode == "prev":
            q.tile_order_refresh = 1
    diff = rend.build_adjacent_diff(p, adj, off)
    out8 = torch.zeros((H, W), dtype=torch.uint32, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t = {"forward": [], "backward": [], "render": []}
    for rnd in range(3):                     # round 0 warms up (and, for auto, learns on the first camera)
        for rays, start, cam, sp in frames:
            e = [ev() for _ in range(4)]
            e[0].record()
            f = pipe.trace_forward(p, a, adj, off, rays, start)
            e[1].record()
            pipe.trace_backward(p, a, adj, off, rays, start, f["rgba"], g)
            e[2].record()
            rend.trace_benchmark(p, a16, adj, off, diff, cam, sp, out8, weight_threshold=0.05)
            e[3].record()
            torch.cuda.synchronize()
            if rnd:
                for k, name in enumerate(("forward", "backward", "render")):
                    t[name].append(e[k].elapsed_time(e[k + 1]))
    hist = pipe.trace_forward(p, a, adj, off, frames[0][0], frames[0][1])["num_intersections"].float()
    return {k: {"mean_ms": round(float(np.mean(v)), 4), "worst_ms": round(float(np.max(v)), 4)} for k, v in t.items()} | \
        {"rows_mean_steps_top_to_bottom": [round(float(x), 1) for x in hist.mean(dim=(1, 2)).reshape(8, -1).mean(1)]}


modes = ("static", "auto", "prev") if args.mode == "all" else (args.mode,)
res = {m: run(m) for m in modes}
print(json.dumps({"scene": "two off-centre blobs + a slab on the north-star foam's points", "cameras": args.cameras,
                  "frame": [H, W], "result": res}))
