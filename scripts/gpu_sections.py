"""Experiment (RF_EXPERIMENT_SECTIONS build): where the waves of the trail replay spend their time.
WORKLOAD=train-batch (default): the training-shaped batch of bench.py, backward mode 4; WORKLOAD=north-star: the 1080p
frame, backward mode 3.  Prints the wave clocks summed per section as shares of the walk."""
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
workload = os.environ.get("WORKLOAD", "train-batch")
sh = int(os.environ.get("SH", "3" if workload == "train-batch" else "2"))
fm = foam.make_synthetic_foam(2_000_000, sh, 5, cache_dir=foam.default_cache_dir())
if os.environ.get("EMPTY_DENSITY"):     # every segment lit: bench.py --empty-density
    fm = dict(fm)
    fm["attributes"] = fm["attributes"].copy()
    fm["attributes"][:, -1] = np.maximum(fm["attributes"][:, -1], np.float32(os.environ["EMPTY_DENSITY"]))
if workload == "train-batch":
    rays_np, start_np = bench.training_batch(fm, 1_000_000, 105)
else:
    cam = bench.orbit_camera(1920, 1080, 0)
    rays_np = foam.camera_rays(cam)
    start_np = np.full(rays_np.shape[:-1], foam.nearest_point(fm["points"], cam["position"]), dtype=np.uint32)
rays, start = torch.from_numpy(rays_np).to(dev), torch.from_numpy(start_np).to(dev)
p, a = torch.from_numpy(fm["points"]).to(dev), torch.from_numpy(fm["attributes"]).to(dev)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
g = torch.randn(rays.shape[:-1] + (4,), generator=torch.Generator().manual_seed(1234)).to(dev)
pipe = radfoam.create_pipeline(sh)
pipe.record_trail = True
for _ in range(3):
    f = pipe.trace_forward(p, a, adj, off, rays, start)
    pipe.trace_backward(p, a, adj, off, rays, start, f["rgba"], g)
torch.cuda.synchronize()
stats = torch.zeros(16, dtype=torch.int64, device=dev)
pipe.experiment_stats = stats
f = pipe.trace_forward(p, a, adj, off, rays, start)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
pipe.trace_backward(p, a, adj, off, rays, start, f["rgba"], g)
e1.record()
torch.cuda.synchronize()
s = stats.cpu().tolist()
tot = s[12]
third, fourth = ("tables", "colour_rows_out") if workload == "train-batch" else ("merge_and_cache_updates", "epoch_barriers_and_flush")
print(json.dumps({"workload": workload, "sh_degree": sh, "empty_density": os.environ.get("EMPTY_DENSITY"), "backward_ms": round(e0.elapsed_time(e1), 3), "wave_steps": s[13],
                  "clocks_per_wave_step": round(tot / max(s[13], 1), 1),
                  "share_wait_records_and_face_hit": round(s[8] / tot, 3), "share_segment_colour_row_and_math": round(s[9] / tot, 3),
                  "share_" + third: round(s[10] / tot, 3), "share_" + fourth: round(s[11] / tot, 3),
                  "share_other": round(1 - (s[8] + s[9] + s[10] + s[11]) / tot, 3)}))
