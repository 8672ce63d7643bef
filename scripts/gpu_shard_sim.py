"""What every rank of an N-GPU strong-scaling step would do, measured one rank after the other on ONE GPU
(gpurun has a single GPU; the collectives themselves cannot be measured here and are priced from byte counts).

For world in (2, 4, 8): the north-star frame's rows are cut into `world` contiguous blocks of equal measured
cost (dist.balanced_row_blocks on the full frame's num_intersections; also the even cut for comparison); for each
block the per-step device work of that rank is timed with HIP events -- geometry repack + forward + backward of
its rows, compaction of its gradient rows, and the scatter of all ranks' packed rows -- and its touched rows are
counted.  Prints one JSON document (kept under profiles/).
usage: python scripts/gpu_shard_sim.py [--points 2000000 --seed 5 --sh-degree 2]

--batch: the data-parallel TRAINING step instead (BASELINE config 4; radfoam_amd.dist.DataParallelPipeline): the
`train-batch-lit` workload of bench.py -- 1,000,000 shuffled rays of 8 cameras through the 2 M-point foam at SH 3 with every
segment lit -- split by index over the ranks the way radfoam.BatchFetcher(rank=, world_size=) does (rank r takes rays
[r B / W, (r + 1) B / W) of the shuffled batch: a 1/W sample of every camera).  Per rank, one after the other on this GPU:
geometry repack, trace_forward, trace_backward of its share (flat-batch kernels, the launch shapes a rank would really
see: 125,000 rays = 489 blocks on 256 CUs at W = 8); once: the replicated Adam step over the scene's parameters and the
non-finite scrub of render.py:98-99.  The all-reduce of the flat [points_grad | attr_grad] buffer cannot be measured on
one GPU: it is PRICED from its bytes at the xGMI figure of the task statement (7 links x 153 GB/s per GPU, direct
reduce-scatter + all-gather over all links, 75 % of peak assumed).  Everything this mode prints is a SIMULATION."""
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
ap.add_argument("--batch", action="store_true", help="the data-parallel training step on the train-batch-lit workload (see above)")
ap.add_argument("--rays", type=int, default=1_000_000)
ap.add_argument("--quantiles", type=int, default=0)
ap.add_argument("--shards", choices=["index", "sorted"], default="index",
                help="--batch: a rank's share = a 1/W slice of the shuffled batch (what BatchFetcher(rank=, world_size=) serves) "
                     "or of the batch in the kernels' coherent order")
args = ap.parse_args()
if args.batch and args.sh_degree == 2:
    args.sh_degree = 3          # the training batch of bench.py is SH 3
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


def simulate_training_batch():
    import bench
    att = fm["attributes"].copy()
    att[:, -1] = np.maximum(att[:, -1], np.float32(4.5e-6))        # every segment lit, as the scene's softplus makes it
    p, a = torch.from_numpy(fm["points"]).to(dev), torch.from_numpy(att).to(dev)
    adj, off = torch.from_numpy(fm["point_adjacency"]).to(dev), torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
    r_np, s_np = bench.training_batch(fm, args.rays, args.seed + 100)
    rays, start = torch.from_numpy(r_np).to(dev), torch.from_numpy(s_np).to(dev)
    gen = torch.Generator().manual_seed(1234)
    g = torch.randn((args.rays, 4), generator=gen).to(dev)
    nq = args.quantiles
    q = dg = None
    if nq:
        q = torch.rand((args.rays, nq), generator=gen).sort(dim=-1, descending=True).values.to(dev)
        dg = torch.randn((args.rays, nq), generator=gen).to(dev)
    pipe = radfoam.create_pipeline(d)
    pipe.record_trail = True
    A = pipe.attribute_dim()
    n = p.shape[0]
    pitch = pipe._gradient_pitch()
    flat_bytes = (((n * 3 + 15) // 16 * 16) + n * pitch) * 4
    dense_bytes = n * (3 + A) * 4
    link_GBps, links, eff = 153.0, 7, 0.75
    out = {"simulation": True, "workload": {"num_points": n, "sh_degree": d, "rays": args.rays, "quantiles": nq,
                                            "every_segment_lit": True, "seed": args.seed},
           "xgmi_assumption": {"links_per_gpu": links, "GBps_per_link": link_GBps, "efficiency": eff,
                               "algorithm": "direct reduce-scatter + all-gather over all links (1/W of the buffer per peer and phase)"},
           "flat_grad_bytes": {"rows_on_64_byte_lines": flat_bytes, "dense_rows": dense_bytes}, "worlds": {}}
    # once: what every rank repeats whatever W is
    params = [torch.zeros((n, 3), device=dev, requires_grad=True), torch.zeros((n, 1), device=dev, requires_grad=True),
              torch.zeros((n, 3), device=dev, requires_grad=True), torch.zeros((n, A - 4), device=dev, requires_grad=True)]
    opt = torch.optim.Adam(params, eps=1e-15)
    for t in params:
        t.grad = torch.randn_like(t)
    t_adam = timed(opt.step)
    del opt, params
    flat = torch.randn(flat_bytes // 4, device=dev)
    t_scrub = timed(lambda: flat.masked_fill_(~flat.isfinite(), 0))
    del flat
    torch.cuda.empty_cache()
    out["replicated_per_rank_ms"] = {"adam_step": round(t_adam, 4), "nonfinite_scrub_of_the_flat_buffer": round(t_scrub, 4)}
    # --shards sorted: rank r takes a contiguous 1/W of the batch in the kernels' own coherent order (entry cell, then
    # direction: rf_build_ray_order over the WHOLE batch, which every rank would compute for itself) instead of a 1/W of
    # the batch as it was shuffled -- its rays are then neighbours (one patch of one camera's directions) instead of a
    # 1/W-dense sample of every camera
    order = None
    if args.shards == "sorted":
        import ctypes as C
        from radfoam_amd import _lib
        lib = _lib.load()
        order = torch.empty(args.rays, dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(lib.rf_ray_order_workspace_bytes(args.rays)), 256), dtype=torch.uint8, device=dev)

        def sort_batch():
            _lib.check(lib.rf_build_ray_order(C.c_void_p(rays.data_ptr()), C.c_void_p(start.data_ptr()), args.rays,
                                              C.c_void_p(order.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

        out["replicated_per_rank_ms"]["ray_order_of_the_whole_batch"] = round(timed(sort_batch), 4)
        order = order.to(torch.int64)
        start_i = start.view(torch.int32)        # (uint32 tensors cannot be indexed on the device: the words as int32)
        t_take = timed(lambda: (rays[order[: args.rays // 8]], start_i[order[: args.rays // 8]]))
        out["replicated_per_rank_ms"]["gather_of_a_rank_share_at_world_8"] = round(t_take, 4)
    out["shards"] = args.shards
    for world in args.worlds:
        per = args.rays // world
        ranks = []
        for r in range(world):
            sl = slice(r * per, (r + 1) * per)
            if order is not None:
                sl = order[sl]
            rr, ss, gg = rays[sl].contiguous(), start.view(torch.int32)[sl].contiguous().view(torch.uint32), \
                (g[sl] * (1.0 / world)).contiguous()
            qq = None if q is None else q[sl].contiguous()
            dd = None if dg is None else (dg[sl] * (1.0 / world)).contiguous()
            state = {}

            def pack():
                pipe._cache.invalidate_geometry()
                pipe.prepare_foam(p, a, adj, off)

            def fwd():
                state["f"] = pipe.trace_forward(p, a, adj, off, rr, ss, depth_quantiles=qq)

            def bwd():
                state["b"] = pipe.trace_backward(p, a, adj, off, rr, ss, state["f"]["rgba"], gg, qq,
                                                 state["f"].get("depth_indices"), dd)

            t_pack, t_fwd, t_bwd = timed(pack), timed(fwd), timed(bwd)
            touched = int(((state["b"]["points_grad"] != 0).any(dim=1) | (state["b"]["attr_grad"] != 0).any(dim=1)).sum())
            ranks.append({"rays": per, "blocks": (per + 255) // 256, "pack_ms": round(t_pack, 4), "forward_ms": round(t_fwd, 4),
                          "backward_ms": round(t_bwd, 4), "touched_rows": touched,
                          "replayed_trail": bool(pipe.last_backward_replayed)})
            state.clear()
        device_ms = [x["pack_ms"] + x["forward_ms"] + x["backward_ms"] for x in ranks]
        price = lambda nbytes: 0.0 if world == 1 else 2.0 * (nbytes / world) / (link_GBps * 1e9 * eff) * 1e3
        rec = {"ranks": ranks, "slowest_rank_tracer_ms": round(max(device_ms), 4),
               "mean_rank_tracer_ms": round(float(np.mean(device_ms)), 4),
               "touched_share_of_points": round(max(x["touched_rows"] for x in ranks) / n, 4),
               "all_reduce_priced_ms": {"rows_on_64_byte_lines": round(price(flat_bytes), 4), "dense_rows": round(price(dense_bytes), 4)}}
        rec["step_ms_tracer_plus_exchange_plus_adam"] = round(max(device_ms) + price(flat_bytes) + t_adam + t_scrub, 4)
        out["worlds"][str(world)] = rec
    base = out["worlds"].get("1")
    if base:
        for w, rec in out["worlds"].items():
            rec["speedup_tracer_only"] = round(base["slowest_rank_tracer_ms"] / rec["slowest_rank_tracer_ms"], 3)
            rec["speedup_step"] = round(base["step_ms_tracer_plus_exchange_plus_adam"] / rec["step_ms_tracer_plus_exchange_plus_adam"], 3)
    print(json.dumps(out))


if args.batch:
    simulate_training_batch()
    sys.exit(0)
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
