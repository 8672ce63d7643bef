"""Experiment: per-block start/end times of the forward kernel (RF_EXPERIMENT_TIMELINE build).
Prints how evenly the 8 XCDs finish and how occupancy decays towards the end of the launch."""
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
pipe = radfoam.create_pipeline(2, torch.float32)
pipe.record_trail = True   # backward is driven by hand on plain tensors
nblk = ((1920 + 15) // 16) * ((1080 + 15) // 16)
for _ in range(2):
    pipe.walk_statistics(points, attributes, adjacency, offsets, rays, start, extra_slots=4 * nblk)
raw = pipe.last_raw_statistics.numpy()[8:].reshape(nblk, 4)
np.save(os.path.join(ROOT, "gpurun_out", "timeline.npy"), raw)
xcc = raw[:, 0] & 0xF
t0 = raw[:, 1].min()
s = (raw[:, 1] - t0) / 100.0   # us (100 MHz)
e = (raw[:, 2] - t0) / 100.0
print("blocks", nblk, "span us", e.max(), "mean block us", (e - s).mean(), "max", (e - s).max())
for x in range(8):
    m = xcc == x
    print("xcc", x, "blocks", int(m.sum()), "first start %.0f last end %.0f  sum block-us %.0f  wave steps %d" % (
        s[m].min(), e[m].max(), (e[m] - s[m]).sum(), raw[m, 3].sum()))
# occupancy over time: number of resident blocks in 20 bins
T = e.max()
for b in range(20):
    t = (b + 0.5) * T / 20
    print("t=%5.0f us resident blocks %d" % (t, int(((s <= t) & (e > t)).sum())))
