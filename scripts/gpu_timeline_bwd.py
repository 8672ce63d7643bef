"""Experiment: per-block start/end times of the cached replay backward kernel
(RF_EXPERIMENT_TIMELINE build)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radfoam
from radfoam_amd import foam
import bench

dev = torch.device("cuda", 0)
fm = foam.make_synthetic_foam(2_000_000, 2, 5, cache_dir=foam.default_cache_dir())
cam = bench.orbit_camera(1920, 1080, 0)
rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
start_idx = foam.nearest_point(fm["points"], cam["position"])
points = torch.from_numpy(fm["points"]).to(dev)
attributes = torch.from_numpy(fm["attributes"]).to(dev)
adjacency = torch.from_numpy(fm["point_adjacency"]).to(dev)
offsets = torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
start = torch.full(rays.shape[:-1], start_idx, dtype=torch.int64).to(torch.uint32).to(dev)
grad = torch.randn(rays.shape[:-1] + (4,), generator=torch.Generator().manual_seed(1234)).to(dev)
pipe = radfoam.create_pipeline(2, torch.float32)
pipe.record_trail = True   # backward is driven by hand on plain tensors
nblk = ((1920 + 15) // 16) * ((1080 + 15) // 16)
stats = torch.zeros(8 + 4 * nblk, dtype=torch.int64, device=dev)
orig = pipe._launch_opts
use = [False]
def patched(*a, **kw):
    o = orig(*a, **kw)
    if use[0]:
        o.stats = stats.data_ptr()
    return o
pipe._launch_opts = patched
for _ in range(2):
    out = pipe.trace_forward(points, attributes, adjacency, offsets, rays, start)
    use[0] = True
    pipe.trace_backward(points, attributes, adjacency, offsets, rays, start, out["rgba"], grad)
    use[0] = False
torch.cuda.synchronize()
head = stats.cpu().numpy()[:8] // 2   # two backward launches were recorded
if head[4]:   # RF_EXPERIMENT_COUNTERS build
    print("lit wave-steps %d  lit lanes %d (%.1f/step)  after the pre-merge %d (%.1f/step)  lock rounds %d (%.2f/step)" % (
        head[4], head[5], head[5] / head[4], head[6], head[6] / head[4], head[7], head[7] / head[4]))
    print("flush: rows %d values (global atomics) %d ; lit lanes: bypass %d cached %d" % tuple(head[:4]))
raw = stats.cpu().numpy()[8:].reshape(nblk, 4)
np.save(os.path.join(ROOT, "gpurun_out", "timeline_bwd.npy"), raw)
xcc = raw[:, 0] & 0xF
t0 = raw[:, 1].min()
s = (raw[:, 1] - t0) / 100.0
e = (raw[:, 2] - t0) / 100.0
print("blocks", nblk, "span us", e.max(), "mean block us", (e - s).mean(), "max", (e - s).max())
for x in range(8):
    m = xcc == x
    print("xcc", x, "blocks", int(m.sum()), "last end %.0f  sum block-us %.0f  iterations %d" % (
        e[m].max(), (e[m] - s[m]).sum(), raw[m, 3].sum()))
T = e.max()
for b in range(20):
    t = (b + 0.5) * T / 20
    print("t=%5.0f us resident blocks %d" % (t, int(((s <= t) & (e > t)).sum())))
# cost per iteration by block duration decile
d = e - s
itn = raw[:, 3].astype(np.float64)
order = np.argsort(itn)
for q in range(10):
    sel = order[q * nblk // 10:(q + 1) * nblk // 10]
    print("decile %d: iterations/block %.0f  us/block %.0f  ns/iteration %.0f" % (q, itn[sel].mean(), d[sel].mean(), 1e3 * d[sel].sum() / itn[sel].sum()))
