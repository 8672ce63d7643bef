"""The tile-cost prior (Pipeline.tile_prior: rf_build_cost_grid + rf_estimate_tile_cost) against the static dealing and
against a frame's own measured order, on rays that are new in every launch.

Scenes: the north-star foam (bench.py's camera path, 0.05 degrees per view) and -- --asymmetric -- the same points with the
lopsided density field of scripts/gpu_tile_order_asymmetric.py on a tilted orbit, consecutive cameras 150 degrees apart.
Per mode: mean forward / backward (fp32 SH 2) and render (fp16 SH 2, weight_threshold 0.05) milliseconds over the views,
plus how well the estimate ranks the tiles (Spearman correlation with the measured cost of the same view).
  python scripts/gpu_tile_prior.py [--asymmetric] [--views 12] [--rules xcd xcd:8 tail]"""
import argparse
import json
import math
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
ap.add_argument("--asymmetric", action="store_true")
ap.add_argument("--views", type=int, default=12)
ap.add_argument("--rules", nargs="+", default=["xcd", "xcd:8", "xcd:16", "tail"])
ap.add_argument("--res", type=int, nargs="+", default=[32])
ap.add_argument("--points", type=int, default=2_000_000)
ap.add_argument("--seed", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
W, H = 1920, 1080
fm = foam.make_synthetic_foam(args.points, 2, args.seed, cache_dir=foam.default_cache_dir())
pts, att = fm["points"], fm["attributes"].copy()
if args.asymmetric:
    sigma0 = float(np.median(att[att[:, -1] > 0, -1]))
    blob = lambda c, r: np.exp(-((pts - np.asarray(c, np.float32)) ** 2).sum(1) / (2 * r * r))
    dens = 2.0 * sigma0 * (blob((0.45, 0.2, 0.0), 0.18) + blob((-0.3, -0.35, 0.3), 0.14)) + \
        1.5 * sigma0 * (np.abs(pts[:, 1] + 0.6) < 0.04) * (np.abs(pts[:, 0]) < 0.7) * (np.abs(pts[:, 2]) < 0.7)
    dens[dens < 0.02 * sigma0] = 0.0
    att[:, -1] = dens.astype(np.float32)


def camera(k):
    if not args.asymmetric:
        return bench.view_camera(W, H, 0, k)
    az = 2 * math.pi * ((k * 5) % args.views) / args.views
    el = 0.45 * math.sin(2.3 * k)
    pos = 3.0 * np.array([math.sin(az) * math.cos(el), math.sin(el), -math.cos(az) * math.cos(el)], np.float32)
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(np.array([0, 1, 0], np.float32), fwd)
    right /= np.linalg.norm(right)
    cam = foam.default_camera(W, H)
    cam.update(position=pos, forward=fwd.astype(np.float32), right=right.astype(np.float32),
               up=np.cross(fwd, right).astype(np.float32))
    return cam


p, a = torch.from_numpy(pts).to(dev), torch.from_numpy(att).to(dev)
a16 = a.to(torch.float16)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
frames = []
for k in range(args.views):
    cam = camera(k)
    rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
    s = foam.nearest_point(pts, cam["position"])
    frames.append((rays, torch.full(rays.shape[:-1], s, dtype=torch.int64).to(torch.uint32).to(dev),
                   {k2: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k2, v in cam.items()},
                   torch.tensor([s], dtype=torch.int64).to(torch.uint32).to(dev)))
g = torch.randn((H, W, 4), generator=torch.Generator().manual_seed(1)).to(dev)
ev = lambda: torch.cuda.Event(enable_timing=True)


def measured_orders(rule="xcd"):
    """Per view: the block -> tile table its OWN measured tile costs give (forward and render), computed ahead of time."""
    from radfoam_amd.pipeline import tile_order
    pipe = radfoam.create_pipeline(2)
    default = pipe._default_tiles(H, W, dev)
    out = []
    for rays, start, cam, sp in frames:
        ni = pipe.trace_forward(p, a, adj, off, rays, start)["num_intersections"].reshape(H, W).to(torch.int32)
        pad = torch.nn.functional.pad(ni, (0, (-W) % 16, 0, (-H) % 16))
        cost = pad.view(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).amax(dim=(1, 3)).reshape(-1)
        out.append(tile_order(cost, default, rule).to(torch.int32).contiguous())
    return out


def run(mode, rule=None, res=32):
    """mode: static | prior | own (every view traced twice, the second pass timed: its own measured order, caches warm) |
    static-warm (every view twice under the static dealing, the second pass timed: what the caches alone are worth) |
    own-cold (the view's own measured order, computed ahead of time, on a view traced ONCE after other views: what the
    order alone is worth)."""
    pipe, rend = radfoam.create_pipeline(2), radfoam.create_pipeline(2, torch.float16)
    orders = measured_orders() if mode == "own-cold" else None
    for q in (pipe, rend):
        q.record_trail = True
        q.tile_prior = mode == "prior"
        q.tile_prior_rule = rule
        q.tile_prior_resolution = res
        if mode in ("static", "static-warm", "own-cold"):
            q.tile_order_mode = None if mode != "own-cold" else "auto"
        if mode != "auto":
            q.tile_order_coherence_degrees = 0.0      # (the modes of rounds 4-5; "auto" = the pipeline's defaults of round 6)
    diff = rend.build_adjacent_diff(p, adj, off)
    out8 = torch.zeros((H, W), dtype=torch.uint32, device=dev)
    t = {"forward": [], "backward": [], "render": []}
    for rnd in range(3):                                   # round 0 warms up
        for vi, (rays, start, cam, sp) in enumerate(frames):
            if orders is not None:
                pipe.experiment_tile_order = rend.experiment_tile_order = orders[vi]
            for rep in range(2 if mode in ("own", "static-warm") else 1):
                e = [ev() for _ in range(4)]
                e[0].record()
                f = pipe.trace_forward(p, a, adj, off, rays, start)
                e[1].record()
                pipe.trace_backward(p, a, adj, off, rays, start, f["rgba"], g)
                e[2].record()
                rend.trace_benchmark(p, a16, adj, off, diff, cam, sp, out8, weight_threshold=0.05)
                e[3].record()
                torch.cuda.synchronize()
                if rnd and rep == (1 if mode in ("own", "static-warm") else 0):
                    for k, name in enumerate(("forward", "backward", "render")):
                        t[name].append(e[k].elapsed_time(e[k + 1]))
    return {k: {"mean_ms": round(float(np.mean(v)), 4), "worst_ms": round(float(np.max(v)), 4)} for k, v in t.items()}


def spearman(x, y):
    rx, ry = torch.argsort(torch.argsort(x)).double(), torch.argsort(torch.argsort(y)).double()
    rx, ry = rx - rx.mean(), ry - ry.mean()
    return float((rx * ry).sum() / (rx.norm() * ry.norm()))


def rank_quality(res):
    pipe = radfoam.create_pipeline(2)
    pipe.tile_prior_resolution = res
    out = []
    for rays, start, cam, sp in frames[:4]:
        ni = pipe.trace_forward(p, a, adj, off, rays, start)["num_intersections"].reshape(H, W).float()
        pad = torch.nn.functional.pad(ni, (0, (-W) % 16, 0, (-H) % 16))
        measured = pad.view(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).amax(dim=(1, 3)).reshape(-1)
        est = pipe.estimate_tile_cost((p, a, adj, off), H, W, rays=rays).float()
        out.append({"spearman": round(spearman(est, measured), 4), "estimated_mean": round(float(est.mean()), 1),
                    "measured_mean": round(float(measured.mean()), 1), "estimated_max": float(est.max()),
                    "measured_max": float(measured.max())})
    return out


res = {"scene": "asymmetric density on the north-star points, tilted orbit" if args.asymmetric else
       "north-star frame, camera path of 0.05 degrees per view", "views": args.views, "rank": {}, "modes": {}}
for r in args.res:
    res["rank"][str(r)] = rank_quality(r)
res["modes"]["auto"] = run("auto")
res["modes"]["static"] = run("static")
res["modes"]["static-warm"] = run("static-warm")
res["modes"]["own-cold"] = run("own-cold")
res["modes"]["own"] = run("own")
for r in args.res:
    for rule in args.rules:
        res["modes"][f"prior res {r} rule {rule}"] = run("prior", rule, r)
print(json.dumps(res))
