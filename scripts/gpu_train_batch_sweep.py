"""Sweep of the scheduling knobs on the training-shaped batch (bench.py --workload train-batch: 1 M shuffled rays of 8
cameras, 2 M-point foam, SH 3) in ONE process: depth synchronisation (Pipeline.sync_cells, forward and backward
separately), backward scatter mode.  Every configuration is checked against the first one: rgba bit-equal (scheduling
must not change a result), gradients within 1e-5 relative L2 (sums in another order).
  python scripts/gpu_train_batch_sweep.py [--sh 3] [--configs "fwd:0,2,4,8;bwd4:0,2,4,8;bwd3:0,2,4,8"]  -> JSON lines
"""
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
ap.add_argument("--sh", type=int, default=3)
ap.add_argument("--points", type=int, default=2_000_000)
ap.add_argument("--rays", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--configs", default="fwd:0,1,2,4,8,16;bwd4:0,2,4,8;bwd3:0,2,4,8")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_batch_sweep.json"))
args = ap.parse_args()

dev = torch.device("cuda", 0)
fm = foam.make_synthetic_foam(args.points, args.sh, 5, cache_dir=foam.default_cache_dir())
rays_np, start_np = bench.training_batch(fm, args.rays, 105)
t = lambda a: torch.from_numpy(a).to(dev)
points, attributes = t(fm["points"]), t(fm["attributes"])
adjacency, offsets = t(fm["point_adjacency"]), t(fm["point_adjacency_offsets"])
rays, start = t(rays_np), t(start_np)
grad = torch.randn(rays.shape[:-1] + (4,), generator=torch.Generator().manual_seed(1234)).to(dev)

pipe = radfoam.create_pipeline(args.sh)
pipe.record_trail = True


def run(sync_fwd, sync_bwd, mode):
    pipe.sync_cells = float(sync_fwd)
    pipe.sync_cells_backward = float(sync_bwd)
    pipe.backward_mode = mode
    fe, be = [], []
    out = res = None
    for i in range(2 + args.steps):
        pipe._cache.invalidate_geometry()
        pipe.prepare_foam(points, attributes, adjacency, offsets)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = pipe.trace_forward(points, attributes, adjacency, offsets, rays, start)
        e1.record()
        res = pipe.trace_backward(points, attributes, adjacency, offsets, rays, start, out["rgba"], grad)
        e2.record()
        if i >= 2:
            fe.append((e0, e1))
            be.append((e1, e2))
    torch.cuda.synchronize()
    ms = lambda ev: float(np.mean([a.elapsed_time(b) for a, b in ev]))
    return ms(fe), ms(be), out, res


base = None
results = []
for group in args.configs.split(";"):
    kind, values = group.split(":")
    for v in values.split(","):
        k = float(v)
        if kind == "fwd":
            cfg = dict(sync_fwd=k, sync_bwd=0.0, mode=4)
        elif kind == "both":
            cfg = dict(sync_fwd=k, sync_bwd=k, mode=4)
        else:
            cfg = dict(sync_fwd=0.0, sync_bwd=k, mode=int(kind[3:]))
        f_ms, b_ms, out, res = run(**cfg)
        rec = dict(cfg, forward_ms=round(f_ms, 3), backward_ms=round(b_ms, 3),
                   mrays_per_s=round(args.rays / (f_ms + b_ms) / 1e3, 2))
        if base is None:
            base = (out["rgba"].clone(), res["points_grad"].clone(), res["attr_grad"].clone())
            rec["reference"] = True
        else:
            rec["rgba_bit_equal"] = bool(torch.equal(out["rgba"], base[0]))
            for name, got, ref in (("points_grad", res["points_grad"], base[1]), ("attr_grad", res["attr_grad"], base[2])):
                rec[name + "_rel_l2"] = float((got.double() - ref.double()).norm() / ref.double().norm())
        print(json.dumps(rec), flush=True)
        results.append(rec)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(results, open(args.out, "w"), indent=1)
