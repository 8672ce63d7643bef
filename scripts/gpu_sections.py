"""Experiment (RF_EXPERIMENT_SECTIONS build): where the waves of the flat-batch replay (backward mode 4) spend their time.
Runs the training-shaped batch of bench.py once and prints the wave clocks summed per section."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import radfoam  # noqa: E402
from radfoam_amd import foam  # noqa: E402

dev = torch.device("cuda:0")
sh = int(os.environ.get("SH", "3"))
fm = foam.make_synthetic_foam(2_000_000, sh, 5, cache_dir=foam.default_cache_dir())
rays_np, start_np = bench.training_batch(fm, 1_000_000, 105)
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
print(json.dumps({"backward_ms": round(e0.elapsed_time(e1), 3), "wave_steps": s[13], "clocks_per_wave_step": round(tot / max(s[13], 1), 1),
                  "share_wait_records_and_face_hit": round(s[8] / tot, 3), "share_segment_colour_row_and_math": round(s[9] / tot, 3),
                  "share_tables": round(s[10] / tot, 3), "share_colour_rows_out": round(s[11] / tot, 3),
                  "share_other": round(1 - (s[8] + s[9] + s[10] + s[11]) / tot, 3)}))
