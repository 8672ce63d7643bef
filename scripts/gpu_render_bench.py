"""Render-speed path (BASELINE config 3 stand-in; the trained mipnerf360 'room' checkpoint is not
available): trace_benchmark -- camera by value, caller's half4 table, fp16 attributes, SH degree 3,
RGBA8 output -- on the cached synthetic foams, 1557x1038 like benchmark.py's room frames."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radfoam
from radfoam_amd import foam

dev = torch.device("cuda", 0)
out = {}
for n, seed in ((500_000, 1), (2_000_000, 5)):
    if not os.path.exists(os.path.join(foam.default_cache_dir(), f"foam_n{n}_s{seed}.npz")):
        continue
    fm = foam.make_synthetic_foam(n, 3, seed, cache_dir=foam.default_cache_dir())
    pts = torch.from_numpy(fm["points"]).to(dev)
    att = torch.from_numpy(fm["attributes"]).to(dev).half()
    adj = torch.from_numpy(fm["point_adjacency"]).to(dev)
    off = torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
    pipe = radfoam.create_pipeline(3, torch.float16)
    diff = pipe.build_adjacent_diff(pts, adj, off)
    cam = foam.default_camera(1557, 1038)
    camera = {k: torch.from_numpy(np.asarray(cam[k], dtype=np.float32)) for k in ("position", "forward", "right", "up")}
    camera.update(fov=float(cam["fov"]), width=1557, height=1038, model="pinhole")
    start = torch.tensor([foam.nearest_point(fm["points"], cam["position"])], dtype=torch.int64).to(torch.uint32).to(dev)
    img = torch.zeros(1038 * 1557, dtype=torch.int32, device=dev).view(torch.uint32)
    for _ in range(3):
        pipe.trace_benchmark(pts, att, adj, off, diff, camera, start, img, weight_threshold=0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        pipe.trace_benchmark(pts, att, adj, off, diff, camera, start, img, weight_threshold=0.05)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    out[f"{n}_points"] = {"ms_per_frame": round(dt * 1e3, 3), "fps": round(1 / dt, 1),
                          "Mrays_per_s": round(1557 * 1038 / dt / 1e6, 1),
                          "opaque_fraction": float(((img.view(torch.int32) >> 24) & 0xFF).float().mean() / 255)}
print(json.dumps({"workload": "trace_benchmark, 1557x1038, SH 3, fp16 attributes, weight_threshold 0.05 (benchmark.py settings)", "results": out}))
