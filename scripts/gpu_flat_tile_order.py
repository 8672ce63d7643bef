"""Experiment: tile orders for the 256-slot groups of a sorted flat batch (Pipeline.tile_order_mode), on the training-shaped
batch of bench.py: static dealing, order learnt on the same batch, order learnt on ANOTHER random batch of the same cameras
(what a training loop would have: every step brings new rays)."""
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

dev = torch.device("cuda:0")
sh = 3
fm = foam.make_synthetic_foam(2_000_000, sh, 5, cache_dir=foam.default_cache_dir())
p, a = torch.from_numpy(fm["points"]).to(dev), torch.from_numpy(fm["attributes"]).to(dev)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)


def batch(seed):
    r, s = bench.training_batch(fm, 1_000_000, seed)
    return torch.from_numpy(r).to(dev), torch.from_numpy(s).to(dev)


rays, start = batch(105)
other = batch(106)
g = torch.randn(rays.shape[:-1] + (4,), generator=torch.Generator().manual_seed(1)).to(dev)


def timed(pipe, reps=6):
    st = {}

    def fwd():
        st["f"] = pipe.trace_forward(p, a, adj, off, rays, start)

    def bwd():
        pipe.trace_backward(p, a, adj, off, rays, start, st["f"]["rgba"], g)

    out = []
    for fn in (fwd, bwd):
        fn(); fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append(round(float(np.median(ts)), 4))
    return out


res = {}
for name, mode, learn_on in (("static", None, None), ("xcd_own", "xcd", None), ("tail512_own", "tail:512", None),
                             ("xcd_learnt_on_another_batch", "xcd", other), ("tail512_learnt_on_another_batch", "tail:512", other)):
    pipe = radfoam.create_pipeline(sh)
    pipe.record_trail = True
    pipe.tile_order_mode = mode
    if learn_on is not None:
        pipe.trace_forward(p, a, adj, off, learn_on[0], learn_on[1])
        pipe.tile_order_refresh = 1 << 30      # keep what the other batch taught
    res[name] = timed(pipe)
print(json.dumps({"forward_ms_backward_ms": res}))
