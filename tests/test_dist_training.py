"""CPU, gloo, world size 2: the data-parallel TRAINING step (BASELINE config 4: "full train.py loop, rays row-sharded
across 8xMI355X with RCCL grad all-reduce"; VERDICT r5 missing #1).

What runs on every rank is the reference's OWN model code -- radfoam_model/scene.py::RadFoamScene + render.py::TraceRays,
unmodified -- through train.py:176-216's loop body: shuffled batches from radfoam.BatchFetcher, two depth quantiles per
ray, SmoothL1 colour loss + opacity loss + quantile loss (means over the batch), loss.backward(), Adam, the learning-rate
schedule, update_triangulation(incremental=True).  The only additions are the three things radfoam_amd.dist provides:
``enable_data_parallel()`` (create_pipeline returns the DataParallelPipeline wrapper, shuffled fetchers serve the rank's
share of the reference's index sequence), ``assert_replicas_agree`` and ``replicated_inputs()``.

The HIP kernels cannot run here, so the wrapped pipeline is the CPU oracle behind the Pipeline interface (test
infrastructure, as in tests/test_dist.py); what is under test is everything around it: the sharded fetch, the exchange
inside trace_backward, the mean over ranks, the consistency of the replicas.

  * parameters after 3 Adam steps == the single-process run on the whole batches (same rays: the ranks' shares
    concatenated ARE the reference's batch), to fp32 summation order;
  * every rank holds the same bits in every parameter and in the rebuilt adjacency after every step;
  * shard="rows" (same rays on every rank, rows traced per rank, outputs gathered): forward outputs bit-identical to the
    single process on every rank.
"""
import importlib
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CANDIDATES = ["/root/reference", os.path.join(ROOT, "oracle", "_ref", "pyref")]
PYREF = next((p for p in _CANDIDATES if os.path.isfile(os.path.join(p, "radfoam_model", "scene.py"))), None)

pytestmark = pytest.mark.skipif(PYREF is None, reason="the reference's radfoam_model is neither under /root/reference "
                                                      "nor under oracle/_ref/pyref (make -C oracle -f Makefile.ref pyref)")


class OraclePipeline:
    """Pipeline-shaped wrapper over the CPU oracle with the reference binding's signature and the flat gradient buffer of
    radfoam_amd.pipeline.Pipeline (test infrastructure only)."""

    def __init__(self, d):
        self.d = d
        self.backward_mode = 0          # a knob, to check that the wrapper forwards attribute writes

    def attribute_dim(self):
        return 1 + 3 * (self.d + 1) ** 2

    def trace_forward(self, points, attributes, adj, off, rays, start_point, depth_quantiles=None,
                      weight_threshold=None, max_intersections=None, return_contribution=False):
        from oracle import oracle as O
        npy = lambda t: None if t is None else t.detach().contiguous().numpy()
        out = O.trace_forward(self.d, npy(points), npy(attributes), npy(adj), npy(off), npy(rays),
                              npy(start_point.contiguous()), depth_quantiles=npy(depth_quantiles),
                              weight_threshold=weight_threshold, max_intersections=max_intersections,
                              return_contribution=return_contribution, num_threads=1)
        return {k: torch.from_numpy(v) for k, v in out.items()}

    def trace_backward(self, points, attributes, adj, off, rays, start_point, rgb_out, grad_in, depth_quantiles=None,
                       depth_indices=None, depth_grad_in=None, ray_error=None, weight_threshold=None,
                       max_intersections=None):
        from oracle import oracle as O
        npy = lambda t: None if t is None else t.detach().contiguous().numpy()
        out = O.trace_backward(self.d, npy(points), npy(attributes), npy(adj), npy(off), npy(rays),
                               npy(start_point.contiguous()), npy(rgb_out), npy(grad_in),
                               depth_quantiles=npy(depth_quantiles), depth_indices=npy(depth_indices),
                               depth_grad_in=npy(depth_grad_in), ray_error=npy(ray_error),
                               weight_threshold=weight_threshold, max_intersections=max_intersections, num_threads=1)
        n, a = points.shape[0], self.attribute_dim()
        flat = torch.zeros(n * (3 + a), dtype=torch.float32)
        flat[: 3 * n] = torch.from_numpy(out["points_grad"]).reshape(-1)
        flat[3 * n:] = torch.from_numpy(out["attr_grad"]).reshape(-1)
        res = {"points_grad": flat[: 3 * n].view(n, 3), "attr_grad": flat[3 * n:].view(n, a), "flat_grad": flat,
               "ray_grad": torch.zeros_like(rays)}
        if "point_error" in out:
            res["point_error"] = torch.from_numpy(out["point_error"])
        return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _reference_scene():
    if PYREF not in sys.path:
        sys.path.insert(0, PYREF)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if "plyfile" not in sys.modules:
        ply = types.ModuleType("plyfile")   # scene.py:5 imports it for save_ply only
        ply.PlyData = ply.PlyElement = object
        sys.modules["plyfile"] = ply
    return importlib.import_module("radfoam_model.scene")


def _views(device="cpu"):
    """Two small frames looking at the scene's initial cloud + targets: (rays [V,H,W,6], rgbs, alphas)."""
    from tests.test_reference_scene import _camera_rays
    rays = torch.stack([_camera_rays((0.0, 10.0, -160.0), 48, 32, device), _camera_rays((120.0, -20.0, 90.0), 48, 32, device)])
    g = torch.Generator().manual_seed(3)
    rgbs = torch.rand(rays.shape[:-1] + (3,), generator=g)
    alphas = (torch.rand(rays.shape[:-1] + (1,), generator=g) > 0.3).float()
    return rays, rgbs, alphas


def train_steps(scene_mod, steps, batch, shard, sh=1, n_init=1200, device="cpu"):
    """train.py:162-248's loop body on the reference's scene, `steps` iterations; data parallel when torch.distributed is
    initialised (the caller has enabled it).  Returns what the callers compare."""
    from types import SimpleNamespace

    import radfoam
    from radfoam_amd import dist as rdist
    from torch import nn

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    torch.manual_seed(7)                                    # every rank: the same initial scene, the same RNG stream
    margs = SimpleNamespace(sh_degree=sh, init_points=n_init, final_points=4 * n_init, activation_scale=1.0)
    oargs = SimpleNamespace(points_lr_init=2e-4, points_lr_final=5e-6, density_lr_init=1e-1, density_lr_final=1e-2,
                            attributes_lr_init=5e-3, attributes_lr_final=5e-4, sh_factor=0.1, freeze_points=18_000)
    model = scene_mod.RadFoamScene(margs, device=torch.device(device))
    with torch.no_grad():
        model.att_dc.copy_(0.5 * torch.randn_like(model.att_dc))
        model.att_sh.copy_(0.2 * torch.randn_like(model.att_sh))
        model.density.copy_(-0.25 + 0.1 * torch.randn_like(model.density))
    model.declare_optimizer(oargs, warmup=100, max_iterations=1000)
    rays, rgbs, alphas = (t.to(device) for t in _views())
    flat = lambda t: t.reshape(-1, t.shape[-1])
    # data_loader/__init__.py:113-127, unmodified call sites: after enable_data_parallel(shard_batches=True) these
    # shuffled fetchers serve the rank's share
    fr = radfoam.BatchFetcher(flat(rays), batch, shuffle=True)
    fc = radfoam.BatchFetcher(flat(rgbs), batch, shuffle=True)
    fa = radfoam.BatchFetcher(flat(alphas), batch, shuffle=True)
    local = batch // world if (world > 1 and shard == "caller") else batch
    rgb_loss = nn.SmoothL1Loss(reduction="none")
    qgen = torch.Generator().manual_seed(99)
    losses, first_outputs = [], None
    for i in range(steps):
        ray_batch, rgb_batch, alpha_batch = fr.next(), fc.next(), fa.next()
        assert ray_batch.shape == (local, 6)
        q_full = torch.rand(batch, 2, generator=qgen).sort(dim=-1, descending=True).values.to(device)      # train.py:176-180
        q = q_full[rank * local:(rank + 1) * local] if local != batch else q_full
        rgba, depth, _, nint, _ = model(ray_batch, depth_quantiles=q)
        if first_outputs is None:
            first_outputs = (rgba.detach().clone(), depth.detach().clone(), nint.clone())
        opacity = rgba[..., -1:]
        rgb_out = rgba[..., :3] + (1 - opacity)
        color_loss = rgb_loss(rgb_batch, rgb_out)
        opacity_loss = ((alpha_batch - opacity) ** 2).mean()
        valid = (depth > 0).all(dim=-1)
        quant_loss = ((depth[..., 0] - depth[..., 1]).abs() * valid).mean()
        loss = color_loss.mean() + opacity_loss + 1e-2 * quant_loss
        model.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        model.optimizer.step()
        model.update_learning_rate(i)
        losses.append(float(loss.detach()))
        if i == 1:
            model.update_triangulation(incremental=True)    # train.py:243-248: every rank rebuilds for itself
        rdist.assert_replicas_agree({"primal_points": model.primal_points, "density": model.density,
                                     "att_dc": model.att_dc, "att_sh": model.att_sh,
                                     "point_adjacency": model.point_adjacency,
                                     "point_adjacency_offsets": model.point_adjacency_offsets})
    params = {k: getattr(model, k).detach().clone() for k in ("primal_points", "density", "att_dc", "att_sh")}
    return {"params": params, "losses": losses, "first": first_outputs, "model": model, "views": (rays, rgbs, alphas)}


def _worker(rank, world, port, shard, exchange, out_dir):
    sys.path.insert(0, ROOT)
    import radfoam
    from radfoam_amd import dist as rdist
    from radfoam_amd import shims

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        scene_mod = _reference_scene()
        rdist.enable_data_parallel(shard_batches=(shard == "caller"), shard=shard, exchange=exchange)
        assert shims.BatchFetcher.default_shard == ((rank, world) if shard == "caller" else None)
        radfoam.create_pipeline = lambda d, dt="float32": rdist.wrap_pipeline(OraclePipeline(d))
        if shard == "rows":     # flat batches cut in the coherent order (normally from 16384 rays on)
            keep_init = rdist.DataParallelPipeline.__init__

            def small_batches_too(self, *a, **k):
                keep_init(self, *a, **k)
                self.coherent_min_rays = 64

            rdist.DataParallelPipeline.__init__ = small_batches_too
        res = train_steps(scene_mod, steps=3, batch=512, shard=shard)
        model = res["model"]
        pipe = model.pipeline
        assert isinstance(pipe, rdist.DataParallelPipeline) and pipe.last_exchange["world"] == world
        if shard == "rows":
            assert pipe._cut is not None and pipe._cut["rays"].shape == (256, 6)      # the sorted cut was taken
        assert pipe.last_exchange["exchange"] == ("dense" if exchange in ("auto", "dense") else pipe.last_exchange["exchange"])
        pipe.backward_mode = 4                                  # a knob lands on the wrapped pipeline
        assert pipe.inner.backward_mode == 4 and "backward_mode" not in pipe.__dict__

        # statistics: a whole view on every rank ("rows"), contribution + point_error through the reference's ErrorBox
        rays = res["views"][0][0]
        extra = {}
        with pipe.replicated_inputs():
            rgba, _, contrib, nint, box = model(rays, return_contribution=True)
            box.ray_error = torch.rand(rays.shape[:-1], generator=torch.Generator().manual_seed(5))
            model.optimizer.zero_grad(set_to_none=True)
            rgba.sum().backward()
            extra = {"rgba": rgba.detach(), "contribution": contrib.detach(), "num_intersections": nint,
                     "point_error": box.point_error.detach(), "points_grad": model.primal_points.grad.detach().clone()}
            assert pipe.last_exchange["exchange"] in ("sparse", "dense")
            rdist.assert_replicas_agree(extra)
        assert pipe.shard == shard
        torch.save({"params": res["params"], "losses": res["losses"], "first": res["first"], "extra": extra},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        rdist.disable_data_parallel()
        dist.destroy_process_group()


def _single_process():
    """The same steps in one process on the whole batches (what the reference does)."""
    import radfoam
    scene_mod = _reference_scene()
    keep = radfoam.create_pipeline
    radfoam.create_pipeline = lambda d, dt="float32": OraclePipeline(d)
    try:
        res = train_steps(scene_mod, steps=3, batch=512, shard="caller")
        model = res["model"]
        rays = res["views"][0][0]
        rgba, _, contrib, nint, box = model(rays, return_contribution=True)
        box.ray_error = torch.rand(rays.shape[:-1], generator=torch.Generator().manual_seed(5))
        model.optimizer.zero_grad(set_to_none=True)
        rgba.sum().backward()
        res["extra"] = {"rgba": rgba.detach(), "contribution": contrib.detach(), "num_intersections": nint,
                        "point_error": box.point_error.detach(), "points_grad": model.primal_points.grad.detach().clone()}
        return res
    finally:
        radfoam.create_pipeline = keep
        for name in [m for m in sys.modules if m.startswith("radfoam_model")]:
            sys.modules.pop(name, None)


@pytest.mark.parametrize("shard,exchange", [("caller", "auto"), ("rows", "auto"), ("caller", "sparse")])
def test_data_parallel_training_step_equals_the_single_process_run(shard, exchange, tmp_path):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard, exchange, str(tmp_path))) for r in range(world)]
    for pr in procs:
        pr.start()
    single = _single_process()
    for pr in procs:
        pr.join(timeout=300)
    for pr in procs:
        if pr.is_alive():
            pr.kill()
            pytest.fail("distributed worker hung")
        assert pr.exitcode == 0
    ranks = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]

    # the replicas: the same bits everywhere
    for k, v in ranks[0]["params"].items():
        assert torch.equal(v, ranks[1]["params"][k]), k
    # against the single process on the whole batches: the same parameters to summation order
    for k, want in single["params"].items():
        got = ranks[0]["params"][k]
        assert got.shape == want.shape
        moved = (want - 0.0).abs().max()
        assert float(moved) > 0
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6, msg=lambda m: f"{k}: {m}")
    # the parameters did move (Adam steps of lr 2e-4 .. 1e-1 on every group)
    if shard == "rows":
        # every rank saw the whole batch: the loss is the single process's, the gathered outputs are its outputs
        assert ranks[0]["losses"] == ranks[1]["losses"]
        np.testing.assert_allclose(ranks[0]["losses"], single["losses"], rtol=1e-6)
        for got, want in zip(ranks[0]["first"], single["first"]):
            assert torch.equal(got.view(torch.int32) if got.dtype == torch.uint32 else got,
                               want.view(torch.int32) if want.dtype == torch.uint32 else want)
    else:
        # the mean of the ranks' losses is the batch's loss; the ranks' outputs concatenated are the batch's outputs
        np.testing.assert_allclose(0.5 * (np.array(ranks[0]["losses"]) + np.array(ranks[1]["losses"])), single["losses"],
                                   rtol=1e-5)
        for i in range(3):
            got = torch.cat([ranks[0]["first"][i], ranks[1]["first"][i]])
            want = single["first"][i]
            assert torch.equal(got.view(torch.int32) if got.dtype == torch.uint32 else got,
                               want.view(torch.int32) if want.dtype == torch.uint32 else want)
    # a view traced with replicated inputs after the steps: what one process returns (the two runs' parameters differ in
    # the last bits by now, so "equal" is to rounding; the ranks themselves were held bit-equal inside the workers)
    torch.testing.assert_close(ranks[0]["extra"]["rgba"], single["extra"]["rgba"], rtol=1e-4, atol=1e-5)
    same_steps = (ranks[0]["extra"]["num_intersections"].view(torch.int32) ==
                  single["extra"]["num_intersections"].view(torch.int32)).float().mean()
    assert float(same_steps) > 0.99
    for key in ("contribution", "point_error", "points_grad"):
        a, b = ranks[0]["extra"][key].double(), single["extra"][key].double()
        assert float((a - b).norm() / b.norm()) < 1e-3, key
        assert float(single["extra"][key].abs().max()) > 0


def test_replica_checksums_see_a_single_word():
    from radfoam_amd import dist as rdist
    a = torch.arange(1000, dtype=torch.float32)
    b = a.clone()
    assert torch.equal(rdist.replica_checksums([a, a.to(torch.int32)]), rdist.replica_checksums([b, b.to(torch.int32)]))
    b[517] = float(np.nextafter(np.float32(517), np.float32(1e9)))
    assert not torch.equal(rdist.replica_checksums([a]), rdist.replica_checksums([b]))
    c = a.clone()
    c[[3, 4]] = c[[4, 3]]                                   # a swap: position-weighted, so it shows
    assert not torch.equal(rdist.replica_checksums([a]), rdist.replica_checksums([c]))
    u = torch.arange(64, dtype=torch.int64).to(torch.uint32)
    assert rdist.replica_checksums([u]).shape == (1,)
    assert rdist.assert_replicas_agree({"a": a}).shape == (1,)   # a single process: nothing to compare, no error


def test_batch_fetcher_shares_concatenate_to_the_references_batch():
    """CPU path of radfoam.BatchFetcher(rank=, world_size=): rank r's share is elements [r*B/W, (r+1)*B/W) of the reference's
    sequence (batch_fetcher.cpp:60-70), shuffled and sequential alike; uneven shares are refused."""
    import radfoam
    data = torch.arange(1000, dtype=torch.float32).reshape(-1, 1) * torch.ones(1, 3)
    for shuffle in (True, False):
        whole = radfoam.BatchFetcher(data, 96, shuffle)
        parts = [radfoam.BatchFetcher(data, 96, shuffle, rank=r, world_size=4) for r in range(4)]
        for _ in range(3):
            assert torch.equal(torch.cat([p.next() for p in parts]).cpu(), whole.next().cpu())
        assert parts[0].local_batch_size == 24
    with pytest.raises(RuntimeError, match="multiple of world_size"):
        radfoam.BatchFetcher(data, 97, True, rank=0, world_size=4)
    with pytest.raises(RuntimeError, match="go together"):
        radfoam.BatchFetcher(data, 96, True, rank=0)


def _loop_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import radfoam
    from examples import train_loop
    from radfoam_amd import dist as rdist
    from radfoam_amd import foam, shims

    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        _reference_scene()
        radfoam.create_pipeline = lambda d, dt="float32": rdist.wrap_pipeline(OraclePipeline(d))
        fm = foam.make_synthetic_foam(1500, 1, 13)
        env = {"torch": torch, "dev": torch.device("cpu")}
        its, detail = train_loop.run(None, env, fm, sh_degree=1, iterations=5, rays_per_batch=1536, cameras=2, width=48,
                                     height=32, densify_at=10 ** 9)
        assert detail["world_size"] == world and detail["rays_per_rank"] == 1536 // world
        assert shims.BatchFetcher.default_shard is None
        if world > 1:
            assert detail["last_exchange"]["exchange"] == "dense" and detail["calls_by_section"]["replica_check"] >= 2
            assert detail["calls_by_section"]["tracer_backward_kernels"] == 5
        torch.save({"losses": detail["loss_trace"], "its": its}, os.path.join(out_dir, f"loop{world}_{rank}.pt"))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_train_loop_example_runs_data_parallel_on_cpu_ranks(tmp_path):
    """examples/train_loop.py (what `bench.py --workload train-loop --gpus N` runs) under gloo at world size 2 against the
    same script in one process: the ranks' losses (each the mean over its half of the batch) average to the single
    process's loss at every recorded iteration -- the same rays, the same quantiles, the same parameters after every
    step -- and the replicas pass their checks after every rebuild."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    procs.append(ctx.Process(target=_loop_worker, args=(0, 1, 0, str(tmp_path))))
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
    for pr in procs:
        if pr.is_alive():
            pr.kill()
            pytest.fail("worker hung")
        assert pr.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"loop2_{r}.pt")) for r in range(2))
    one = torch.load(os.path.join(str(tmp_path), "loop1_0.pt"))
    assert [i for i, _ in r0["losses"]] == [i for i, _ in one["losses"]]
    mean = [0.5 * (a[1] + b[1]) for a, b in zip(r0["losses"], r1["losses"])]
    np.testing.assert_allclose(mean, [l for _, l in one["losses"]], rtol=2e-4)


def test_coherent_shards_partition_a_batch_by_camera_and_direction():
    """dist.coherent_shard: a rank's share of a flat batch that every rank holds whole is a contiguous slice of the batch
    sorted by camera (origin) and direction -- the shares partition the batch, each holds rays of as few cameras as its
    size allows, and the order is a pure function of the rays (every rank computes the same one)."""
    from radfoam_amd import dist as rdist
    g = torch.Generator().manual_seed(1)
    cams = torch.tensor([[0.0, 0.0, -3.0], [2.0, 1.0, 0.5], [-1.0, 2.0, 2.0], [0.5, -2.5, 1.0]])
    which = torch.randint(0, 4, (4096,), generator=g)
    rays = torch.cat([cams[which], torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=1)], dim=1)
    shares = [rdist.coherent_shard(rays, r, 8) for r in range(8)]
    assert sorted(torch.cat(shares).tolist()) == list(range(4096))
    assert all(s.numel() == 512 for s in shares)
    assert torch.equal(rdist.coherent_order(rays), rdist.coherent_order(rays.clone()))
    # 4 cameras of ~1024 rays over 8 shares of 512: a share spans at most 2 cameras (an index share holds all 4)
    assert max(len(set(which[s].tolist())) for s in shares) <= 3
    assert len(set(which[:512].tolist())) == 4
    # an explicit group key (the entry cell) takes the place of the origin hash
    o = rdist.coherent_order(rays, which.to(torch.int64).to(torch.uint32))
    assert (which[o][1:] >= which[o][:-1]).all()
    with pytest.raises(RuntimeError, match="multiple of the world size"):
        rdist.coherent_shard(rays[:4095], 0, 8)


def _gpu_worker(rank, world, port, shard, out_dir):
    """Two ranks on ONE GPU (gloo moves CUDA tensors; RCCL refuses two ranks per device): the real thing on every rank --
    HIP tracer, GPU triangulation, device-resident fetchers, the exchange kernels."""
    sys.path.insert(0, ROOT)
    import radfoam
    from radfoam_amd import dist as rdist
    from radfoam_amd.pipeline import Pipeline

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        scene_mod = _reference_scene()
        rdist.enable_data_parallel(shard_batches=(shard == "caller"), shard=shard)
        if shard == "rows":     # flat batches cut in the coherent order (rf_build_ray_order over the whole batch) already here
            keep_init = rdist.DataParallelPipeline.__init__

            def small_batches_too(self, *a, **k):
                keep_init(self, *a, **k)
                self.coherent_min_rays = 1024

            rdist.DataParallelPipeline.__init__ = small_batches_too
        res = train_steps(scene_mod, steps=3, batch=4096, shard=shard, sh=2, n_init=6000, device="cuda")
        if shard == "rows":
            assert res["model"].pipeline._cut is not None
        pipe = res["model"].pipeline
        assert isinstance(pipe, rdist.DataParallelPipeline) and isinstance(pipe.inner, Pipeline)
        assert pipe.last_exchange["world"] == world and pipe.last_exchange["exchange"] == "dense"
        rays = res["views"][0][0]
        model = res["model"]
        with pipe.replicated_inputs():          # a whole view on every rank: rows traced per rank, sparse row exchange
            rgba, _, contrib, nint, _ = model(rays, return_contribution=True)
            model.optimizer.zero_grad(set_to_none=True)
            rgba.sum().backward()
            assert pipe.last_exchange["exchange"] in ("sparse", "dense")
            rdist.assert_replicas_agree({"rgba": rgba.detach(), "contribution": contrib.detach(),
                                         "points_grad": model.primal_points.grad})
        torch.save({"params": {k: v.cpu() for k, v in res["params"].items()}, "losses": res["losses"],
                    "rgba": rgba.detach().cpu()}, os.path.join(out_dir, f"gpu{world}_{rank}.pt"))
    finally:
        rdist.disable_data_parallel()
        dist.destroy_process_group()


def _gpu_single(out_dir):
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    scene_mod = _reference_scene()
    res = train_steps(scene_mod, steps=3, batch=4096, shard="caller", sh=2, n_init=6000, device="cuda")
    model = res["model"]
    rgba = model(res["views"][0][0])[0]
    torch.save({"params": {k: v.cpu() for k, v in res["params"].items()}, "losses": res["losses"], "rgba": rgba.detach().cpu()},
               os.path.join(out_dir, "gpu1_0.pt"))


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["caller", "rows"])
def test_data_parallel_training_step_two_ranks_on_the_gpu(shard, tmp_path):
    """The data-parallel step with NOTHING replaced: two processes on cuda:0 under gloo, each with the reference's scene on
    the HIP tracer, the GPU triangulation (every rank rebuilds for itself: the replica check after the rebuild is a test of
    its determinism across processes), device-resident fetchers and the pitched exchange kernels -- against one process on
    the whole batches.  Same parameters to summation order (atomics), same bits on both ranks."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, shard, str(tmp_path))) for r in range(2)]
    procs.append(ctx.Process(target=_gpu_single, args=(str(tmp_path),)))
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=600)
    for pr in procs:
        if pr.is_alive():
            pr.kill()
            pytest.fail("worker hung")
        assert pr.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"gpu2_{r}.pt")) for r in range(2))
    one = torch.load(os.path.join(str(tmp_path), "gpu1_0.pt"))
    for k, v in r0["params"].items():
        assert torch.equal(v, r1["params"][k]), k
        a, b = v.double(), one["params"][k].double()
        assert float((a - b).norm() / b.norm()) < 1e-5, k
    assert torch.equal(r0["rgba"], r1["rgba"])
    torch.testing.assert_close(r0["rgba"], one["rgba"], rtol=1e-3, atol=1e-4)
    if shard == "rows":
        np.testing.assert_allclose(r0["losses"], one["losses"], rtol=1e-4)
    else:
        np.testing.assert_allclose(0.5 * (np.array(r0["losses"]) + np.array(r1["losses"])), one["losses"], rtol=1e-4)
