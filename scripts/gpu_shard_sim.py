"""What every rank of an N-GPU strong-scaling step would do, measured one rank after the other on ONE GPU
(gpurun has a single GPU; the collectives themselves cannot be measured here and are priced from byte counts).

For world in (2, 4, 8): the north-star frame's rows are cut into `world` contiguous blocks of equal measured
cost (dist.balanced_row_blocks on the full frame's num_intersections; also the even cut for comparison); for each
block the per-step device work of that rank is timed with HIP events -- geometry repack + forward + backward of
its rows, compaction of its gradient rows, and the scatter of all ranks' packed rows -- and its touched rows are
counted.  Prints one JSON document (kept under profiles/).
usage: python scripts/gpu_shard_sim.py [--points 2000000 --seed 5 --sh-degree 2]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import radfoam  # noqa: E402
from radfoam_amd import dist as rdist  # noqa: E402
from radfoam_amd import foam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=2_000_000)
ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--sh-degree", type=int, default=2)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--forward-only", action="store_true", help="BASELINE config 5: --points 4000000 --seed 4 --sh-degree 3 --width 3840 --height 2160 --forward-only")
ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--cuts", nargs="+", default=["balanced", "even"])
ap.add_argument("--forward-mode", type=int, default=0, help="Pipeline.forward_mode of the ranks (0 auto, 1, 2)")
args = ap.parse_args()
dev = torch.device("cuda:0")
d = args.sh_degree
H, Wd = args.height, args.width


def gpu_triangulation(raw):
    # no cached Qhull lists for this foam: the GPU triangulation (as in bench.py)
    from radfoam_amd import triangulation
    _, sorted_pts = triangulation.kd_order(torch.from_numpy(raw).to(dev))
    adj_, off_, _ = triangulation.delaunay_adjacency(sorted_pts)
    return sorted_pts.cpu().numpy(), off_.cpu().numpy(), adj_.cpu().numpy()


fm = foam.make_synthetic_foam(args.points, d, args.seed, cache_dir=foam.default_cache_dir(), triangulate=gpu_triangulation)
cam = foam.default_camera(Wd, H)
rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
start_idx = foam.nearest_point(fm["points"], cam["position"])
start = torch.full(rays.shape[:-1], start_idx, dtype=torch.int64).to(torch.uint32).to(dev)
p, a = torch.from_numpy(fm["points"]).to(dev), torch.from_numpy(fm["attributes"]).to(dev)
adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
g = torch.randn(rays.shape[:-1] + (4,), generator=torch.Generator().manual_seed(1234)).to(dev)
pipe = radfoam.create_pipeline(d)
pipe.record_trail = not args.forward_only
pipe.forward_mode = args.forward_mode
pipe.gradient_row_pitch = "dense"   # the exchange kernels read the reference's dense [N][A] rows (ShardedTracer sets this)
A = pipe.attribute_dim()
n = p.shape[0]
ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, reps=args.reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = ev(), ev()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


full = pipe.trace_forward(p, a, adj, off, rays, start)
# the cost model of ShardedTracer.rebalance: per 8-pixel segment the longest ray (a wave runs to its longest ray)
row_cost = full["num_intersections"].reshape(H, -1).to(torch.int64).view(H, -1, 8).amax(dim=2).sum(dim=1).tolist()
ex = rdist.SparseGradExchange()
pitch = ex._pitch(A)
out = {"workload": {"num_points": n, "sh_degree": d, "frame": [H, Wd], "seed": args.seed, "forward_only": args.forward_only}, "worlds": {}}
for world in args.worlds:
    for cut in (tuple(args.cuts) if world > 1 else ("even",)):
        bounds = rdist.balanced_row_blocks(row_cost, world, align=8) if cut == "balanced" else \
            [rdist.row_block(H, r, world)[0] for r in range(world)] + [H]
        ranks = []
        packed = []
        for r in range(world):
            b, e = bounds[r], bounds[r + 1]
            rr, ss, gg = rays[b:e].contiguous(), start[b:e].contiguous(), g[b:e].contiguous()
            state = {}

            def pack():
                pipe._cache.invalidate_geometry()
                pipe.prepare_foam(p, a, adj, off)

            def fwd():
                state["f"] = pipe.trace_forward(p, a, adj, off, rr, ss)

            def bwd():
                state["b"] = pipe.trace_backward(p, a, adj, off, rr, ss, state["f"]["rgba"], gg)

            t_pack, t_fwd = timed(pack), timed(fwd)
            if args.forward_only:
                ranks.append({"rows": [b, e], "rays": int((e - b) * Wd), "pack_ms": round(t_pack, 4), "forward_ms": round(t_fwd, 4),
                              "backward_ms": 0.0, "compact_ms": 0.0, "touched_rows": 0})
                continue
            t_bwd = timed(bwd)
            res = state["b"]
            send = torch.empty((n // 2, pitch), dtype=torch.float32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)

            def compact():
                cnt.zero_()
                ex._compact(res["points_grad"], res["attr_grad"], send, cnt)

            t_compact = timed(compact)
            k = int(cnt)
            packed.append(send[:k].clone())
            ranks.append({"rows": [b, e], "rays": int((e - b) * Wd), "pack_ms": round(t_pack, 4), "forward_ms": round(t_fwd, 4),
                          "backward_ms": round(t_bwd, 4), "compact_ms": round(t_compact, 4), "touched_rows": k})
            del send
        # the scatter every rank performs: zero own rows, add all ranks' rows in rank order
        scratch = torch.zeros(n * (3 + A), dtype=torch.float32, device=dev)
        pg, ag = scratch[: 3 * n].view(n, 3), scratch[3 * n:].view(n, A)

        def scatter():
            ex._scatter(packed[0], packed[0].shape[0], pg, ag, zero=True)
            for q in packed:
                ex._scatter(q, q.shape[0], pg, ag, zero=False)

        t_scatter = timed(scatter) if world > 1 and not args.forward_only else 0.0
        t_zero = timed(lambda: scratch.zero_())
        longest = max(x["touched_rows"] for x in ranks)
        # forward only: a static scene is packed once, not per frame
        device_ms = [(0.0 if args.forward_only else x["pack_ms"]) + x["forward_ms"] + x["backward_ms"] +
                     (x["compact_ms"] + t_scatter if world > 1 else 0.0) for x in ranks]
        out["worlds"][f"{world}_{cut}"] = {
            "bounds": bounds, "ranks": ranks, "scatter_all_ranks_ms": round(t_scatter, 4),
            "zero_fill_flat_grad_ms": round(t_zero, 4),
            "longest_list_rows": longest, "sum_list_rows": sum(x["touched_rows"] for x in ranks),
            "sparse_bytes_in_per_rank": int((world - 1) * longest * pitch * 4),
            "dense_allreduce_bytes_per_rank": int(2 * (world - 1) / world * n * (3 + A) * 4) if world > 1 else 0,
            "max_rank_device_ms_without_collectives": round(max(device_ms), 4),
            "mean_rank_device_ms_without_collectives": round(float(np.mean(device_ms)), 4),
        }
        del packed, scratch
        torch.cuda.empty_cache()
print(json.dumps(out))
