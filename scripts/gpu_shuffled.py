"""Training-like incoherent batch (stand-in for BASELINE config 4, whose dataset is not available):
1,000,000 rays drawn at random from 8 synthetic cameras around the cached 2M-point foam, SH degree 3,
forward + backward, flat ray list (no image tiles).  Prints ms and Mrays/s."""
import json, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radfoam
from radfoam_amd import foam
import bench

dev = torch.device("cuda", 0)
d = int(os.environ.get("SH_DEGREE", "3"))
fm = foam.make_synthetic_foam(2_000_000, d, 5, cache_dir=foam.default_cache_dir())
pts = torch.from_numpy(fm["points"]).to(dev)
att = torch.from_numpy(fm["attributes"]).to(dev)
adj = torch.from_numpy(fm["point_adjacency"]).to(dev)
off = torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
rng = np.random.default_rng(0)
rays_all, start_all = [], []
for c in range(8):
    cam = bench.orbit_camera(960, 540, c)
    r = foam.camera_rays(cam).reshape(-1, 6)
    rays_all.append(r)
    start_all.append(np.full(r.shape[0], foam.nearest_point(fm["points"], cam["position"]), dtype=np.uint32))
rays_all = np.concatenate(rays_all); start_all = np.concatenate(start_all)
res = {}
for name, idx in (("shuffled", rng.permutation(rays_all.shape[0])[:1_000_000]),
                  ("same rays, camera order", np.sort(rng.permutation(rays_all.shape[0])[:1_000_000]))):
    rays = torch.from_numpy(rays_all[idx]).to(dev)
    start = torch.from_numpy(start_all[idx].astype(np.int64)).to(torch.uint32).to(dev)
    grad = torch.randn(rays.shape[0], 4, device=dev)
    pipe = radfoam.create_pipeline(d, torch.float32)
    pipe.record_trail = True   # backward is driven by hand on plain tensors
    ts = []
    for it in range(5):
        pipe._cache.invalidate_geometry()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipe.trace_forward(pts, att, adj, off, rays, start)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pipe.trace_backward(pts, att, adj, off, rays, start, out["rgba"], grad)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    f, b = np.median([t[0] for t in ts[1:]]), np.median([t[1] for t in ts[1:]])
    res[name] = {"forward_ms": round(f * 1e3, 2), "backward_ms": round(b * 1e3, 2),
                 "Mrays_per_s": round(rays.shape[0] / (f + b) / 1e6, 1),
                 "mean_cells_per_ray": round(float(out["num_intersections"].to(torch.int64).float().mean()), 1)}
print(json.dumps({"workload": f"2M-point foam, SH {d}, 1,000,000 rays from 8 cameras (960x540 each), flat list", "results": res}))
