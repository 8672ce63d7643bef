"""The training loop the north star names (BASELINE config 4: "full train.py loop"), timed end to end.

What runs is the reference's OWN model code -- radfoam_model/scene.py::RadFoamScene and render.py::TraceRays, unmodified,
imported from /root/reference when it exists, else from the byte-for-byte copy oracle/Makefile.ref leaves under the
git-ignored oracle/_ref/pyref/ (it travels to the GPU box with the snapshot; it is the CALLER of this package, not a
checker, and nothing of it is committed) -- driven by a restatement of train.py:162-270's loop body against this
repository's ``radfoam``: 1 M-ray shuffled batches from radfoam.BatchFetcher (train.py:61, data_loader/__init__.py:
113-127), two random depth quantiles per ray (train.py:176-180), SmoothL1 colour loss + opacity loss + quantile loss
(:188-203), loss.backward(), the event.synchronize() + next batch of :209-212, Adam (scene.py:300), the learning-rate
schedule, update_triangulation(incremental=True) on the schedule of :243-248 (period 1, +2 per rebuild, 100 at most;
back to 1 after a densification), and ONE densification: collect_error_map over every training view (scene.py:497-548),
prune_and_densify, the full rebuild after it (:255-268).

No dataset is available here (Mip-NeRF 360 'bicycle' needs pycolmap + the images): the scene is the north-star foam
(2,000,000 seeded points, SH degree 3) loaded into the reference's scene object the way its load_pt does, the training
views are 8 synthetic 1080p cameras on an orbit, the targets renders of a DIFFERENT state of the same foam (densities
doubled inside a sphere and halved around it, a smooth colour field added) plus noise: the loss has somewhere to go.  The
densities go through the scene's own parameterisation -- activation_scale * softplus(raw, beta=10) -- so that, unlike the
`train-batch` workload whose empty shell has density exactly 0, EVERY cell a ray crosses is "lit" (4.5e-6 > 1e-6 for the
raw value -1 the reference gives its empty cells, scene.py:459): the colour row is read and a colour gradient row is
written for every segment, as in real training.

Timing: CUDA events on the current stream around every section of an iteration (they partition the GPU timeline of the
loop's single stream; host-side waits show up in the section the host was in) + the wall clock of the whole loop.

Data parallel (one process per GPU under torch.distributed.run; `bench.py --workload train-loop --gpus N`): the same loop,
the same reference files, on every rank.  radfoam_amd.dist.enable_data_parallel() makes the scene's create_pipeline return
the DataParallelPipeline wrapper; every rank fetches the whole batch of the reference's index sequence and keeps a coherent
1/N of it (dist.coherent_shard: the ranks' shares together are the batch one process would fetch); loss.backward() ends, inside
trace_backward, in gradients averaged over the ranks -- the gradient of the whole batch's mean losses -- so every rank
takes the identical Adam step and rebuilds the identical triangulation for itself (checked after every rebuild:
dist.assert_replicas_agree); the depth quantiles are drawn for the whole batch from the (identically seeded) device
generator and each rank keeps its rows; collect_error_map sees whole views on every rank and runs under
pipeline.replicated_inputs(): rows traced per rank, outputs gathered, gradients and statistics summed.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import time
import types
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CANDIDATES = ["/root/reference", os.path.join(ROOT, "oracle", "_ref", "pyref")]


def reference_model():
    """(scene module, where it came from) of the reference's radfoam_model, imported against this repo's radfoam."""
    where = next((p for p in _CANDIDATES if os.path.isfile(os.path.join(p, "radfoam_model", "scene.py"))), None)
    if where is None:
        raise RuntimeError("the reference's radfoam_model is neither under /root/reference nor under oracle/_ref/pyref "
                           "(make -C oracle -f Makefile.ref pyref)")
    if where not in sys.path:
        sys.path.insert(0, where)
    if "plyfile" not in sys.modules:                 # scene.py:5 imports it for save_ply only
        ply = types.ModuleType("plyfile")
        ply.PlyData = ply.PlyElement = object
        sys.modules["plyfile"] = ply
    import radfoam  # noqa: F401  (the drop-in package the reference's files import)
    scene = importlib.import_module("radfoam_model.scene")
    return scene, where


class _Sections:
    """Stopwatch: ``with sec("name"):`` accumulates milliseconds per name -- CUDA events on the current stream (the GPU
    timeline) on a device, the host clock on CPU tensors (the dry run of tests/test_reference_scene.py)."""

    def __init__(self, torch, on_gpu=True):
        self.torch = torch
        self.on_gpu = on_gpu
        self.pending = []
        self.ms = {}
        self.calls = {}
        self.prefix = ""            # set while a phase runs whose nested calls must not be booked to the iterations

    def _mark(self):
        if not self.on_gpu:
            return time.perf_counter()
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def __call__(self, name):
        outer = self

        class _Ctx:
            def __enter__(self):
                self.a = outer._mark()

            def __exit__(self, *exc):
                outer.pending.append((name, self.a, outer._mark()))
                return False

        return _Ctx()

    def wrap(self, obj, attr, name):
        """Time every call of obj.attr under `name` (an instance attribute shadows the method)."""
        fn = getattr(obj, attr)
        outer = self

        def timed(*a, **k):
            with outer(outer.prefix + name):
                return fn(*a, **k)

        setattr(obj, attr, timed)
        return fn

    def collect(self):
        if self.on_gpu:
            self.torch.cuda.synchronize()
        for name, a, b in self.pending:
            dt = a.elapsed_time(b) if self.on_gpu else (b - a) * 1e3
            self.ms[name] = self.ms.get(name, 0.0) + dt
            self.calls[name] = self.calls.get(name, 0) + 1
        self.pending = []


def inverse_softplus(y, beta=10.0):
    """raw with softplus(raw, beta) = y (y > 0), in float64 on the host"""
    y = np.asarray(y, np.float64)
    return np.log(np.expm1(np.maximum(y * beta, 1e-12))) / beta


def build_scene(torch, dev, fm, sh_degree, iterations, densify_from, grow=1.3):
    """The reference's RadFoamScene holding the synthetic foam `fm` (the state load_pt would restore)."""
    import radfoam
    from torch import nn
    scene_mod, where = reference_model()
    n = int(fm["points"].shape[0])
    margs = SimpleNamespace(sh_degree=sh_degree, init_points=4096, final_points=int(grow * n), activation_scale=1.0)
    model = scene_mod.RadFoamScene(margs, device=dev)                      # random_initialize on 4096 points
    pts = torch.from_numpy(fm["points"]).to(dev)
    att = torch.from_numpy(fm["attributes"]).to(dev)
    s = fm["attributes"][:, -1]
    raw = np.where(s > 1e-6, inverse_softplus(s), -1.0).astype(np.float32)  # -1: what the reference gives empty cells
    model.triangulation = radfoam.Triangulation(pts)
    perm = model.triangulation.permutation().to(torch.long)
    model.primal_points = nn.Parameter(pts[perm].contiguous())
    model.density = nn.Parameter(torch.from_numpy(raw).to(dev)[perm].reshape(-1, 1).contiguous())
    model.att_dc = nn.Parameter(att[perm, :3].contiguous())
    model.att_sh = nn.Parameter(att[perm, 3:-1].contiguous())
    model.num_init_points = n
    model.update_triangulation(rebuild=False)
    oargs = SimpleNamespace(points_lr_init=2e-4, points_lr_final=5e-6, density_lr_init=1e-1, density_lr_final=1e-2,
                            attributes_lr_init=5e-3, attributes_lr_final=5e-4, sh_factor=0.1, freeze_points=18_000)
    model.declare_optimizer(oargs, warmup=densify_from, max_iterations=max(iterations, 20_000))
    return model, where


def training_views(torch, dev, model, cameras, width, height, noise=0.02, seed=11):
    """(rays [V,H,W,6], rgbs [V,H,W,3], alphas [V,H,W,1]) on the device: `cameras` orbit views of a TARGET scene that
    differs from the state the loop starts in, so that there is something to learn (the round-4 targets
    were a render of the initial state itself, the loss was flat): the same points, but the density doubled inside a
    sphere of radius 0.45 and halved in the shell around it, and a smooth colour field (a few sine waves of the
    position, amplitude 0.35) added to the DC colour.  The model is put back into its initial state afterwards; the loop
    then moves densities, colours and points towards the target, and the densification's error map has real structure
    (error where the density changed)."""
    import bench
    from radfoam_amd import foam
    rays = torch.stack([torch.from_numpy(foam.camera_rays(bench.orbit_camera(width, height, k))) for k in range(cameras)]).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    rgbs, alphas = [], []
    with torch.no_grad():
        saved = {k: getattr(model, k).detach().clone() for k in ("density", "att_dc")}
        x = model.primal_points.detach()
        r = x.norm(dim=1, keepdim=True)
        raw = model.density.detach()
        act = torch.nn.functional.softplus(raw, beta=10)
        want = torch.where(r < 0.45, 2.0 * act, torch.where(r < 0.8, 0.5 * act, act))
        lit = act > 1e-3
        w10 = (want * 10).clamp(min=1e-6)                     # inverse softplus (beta 10); expm1 overflows fp32 past 88
        model.density.copy_(torch.where(lit, torch.where(w10 > 20, w10, torch.log(torch.expm1(w10.clamp(max=20)))) / 10, raw))
        field = torch.stack([torch.sin(5.0 * x[:, 0] + 1.0), torch.sin(4.0 * x[:, 1] - 0.5) * torch.cos(3.0 * x[:, 2]),
                             torch.cos(6.0 * x[:, 2] + 0.3)], dim=1)
        model.att_dc.add_(0.35 * field / 0.28209479177387814)       # SH DC basis value: +-0.35 in colour
        for k in range(cameras):
            out = model(rays[k])[0]
            rgb = out[..., :3] + (1.0 - out[..., 3:])
            rgbs.append((rgb + noise * torch.randn(rgb.shape, generator=gen).to(dev)).clamp(0, 1))
            alphas.append(out[..., 3:].clone())
        for k, v in saved.items():
            getattr(model, k).copy_(v)
        if hasattr(model, "pipeline") and hasattr(model.pipeline, "invalidate"):
            model.pipeline.invalidate()        # parameter data was written in place: nothing of the target state may stay cached
    return rays, torch.stack(rgbs), torch.stack(alphas)


def _check_replicas(rdist, model):
    """Every rank must hold the same scene: parameters and the lists its own rebuild produced."""
    rdist.assert_replicas_agree({"primal_points": model.primal_points, "density": model.density, "att_dc": model.att_dc,
                                 "att_sh": model.att_sh, "point_adjacency": model.point_adjacency,
                                 "point_adjacency_offsets": model.point_adjacency_offsets})


def run(args, env, fm, sh_degree=3, iterations=300, rays_per_batch=1_000_000, cameras=8, width=1920, height=1080,
        densify_at=150, densify_factor=1.15, quantile_weight=1e-4, white_background=True):
    """The loop; returns (iterations per second, detail dict)."""
    import gc

    import radfoam
    torch, dev = env["torch"], env["dev"]
    from torch import nn
    import torch.distributed as tdist
    from radfoam_amd import dist as rdist
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    rank = tdist.get_rank() if tdist.is_initialized() else 0
    if rays_per_batch % world:
        raise RuntimeError(f"rays_per_batch {rays_per_batch} is not a multiple of the world size {world}")
    local_rays = rays_per_batch // world
    torch.manual_seed(20240 + sh_degree)                    # every rank: the same host and device random streams
    if world > 1:
        # before the scene is built: create_pipeline wraps.  The fetchers stay whole: every rank fetches the whole batch of
        # the reference's index sequence from its resident copy of the training set (0.09 ms) and keeps a COHERENT 1/W of
        # it (dist.coherent_shard: a contiguous slice of the batch sorted by camera and direction, 0.2 ms) -- a rank's
        # tracer is 1.5x faster on that than on its 1/W of the shuffled order (BatchFetcher(rank=, world_size=), which
        # serves the latter without fetching the rest, is the alternative when the training set is not resident)
        rdist.enable_data_parallel(shard_batches=False)
    densify_from = densify_at
    model, where = build_scene(torch, dev, fm, sh_degree, iterations, densify_from)
    n0 = int(model.primal_points.shape[0])
    rays, rgbs, alphas = training_views(torch, dev, model, cameras, width, height)
    handler = SimpleNamespace(rays=rays, rgbs=rgbs)                        # what collect_error_map reads
    flat = lambda t: t.reshape(-1, t.shape[-1])

    def get_iter():                                                        # data_loader/__init__.py:113-127
        fr = radfoam.BatchFetcher(flat(rays), rays_per_batch, shuffle=True)
        fc = radfoam.BatchFetcher(flat(rgbs), rays_per_batch, shuffle=True)
        fa = radfoam.BatchFetcher(flat(alphas), rays_per_batch, shuffle=True)
        while True:
            yield fr.next(), fc.next(), fa.next()

    on_gpu = torch.device(dev).type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    sec = _Sections(torch, on_gpu)
    pipe = model.pipeline
    if world > 1:
        assert isinstance(pipe, rdist.DataParallelPipeline)
        # inside the wrapper: the kernels' share and the exchange's share of trace_backward are booked apart
        sec.wrap(pipe.inner, "trace_backward", "tracer_backward_kernels")
    sec.wrap(pipe, "trace_forward", "tracer_forward")
    sec.wrap(pipe, "trace_backward", "tracer_backward")
    sec.wrap(model, "get_trace_data", "get_trace_data")
    sec.wrap(model, "get_starting_point", "get_starting_point")
    rgb_loss = nn.SmoothL1Loss(reduction="none")
    data_iterator = get_iter()

    def next_batch():
        """(rays, colours, alphas, indices of this rank's share in the whole batch or None)"""
        r, c, a = next(data_iterator)
        if world == 1:
            return r, c, a, None
        idx = rdist.coherent_shard(r, rank, world)
        return r[idx], c[idx], a[idx], idx

    with sec("batch_fetch"):
        ray_batch, rgb_batch, alpha_batch, share = next_batch()

    period, since_update, since_dens, next_dens = 1, 1, 0, 1
    rebuilds = {"incremental": 0, "full": 0}
    densified = None
    losses = []
    sync()
    t_loop = time.perf_counter()
    for i in range(iterations):
        with sec("depth_quantiles"):
            if world > 1:       # the whole batch's quantiles from the shared stream; this rank's rows of them
                depth_quantiles = torch.rand(rays_per_batch, 2, device=dev)[share].sort(dim=-1, descending=True).values
            else:
                depth_quantiles = torch.rand(*ray_batch.shape[:-1], 2, device=dev).sort(dim=-1, descending=True).values
        with sec("model_forward"):      # get_trace_data + get_starting_point + tracer_forward (timed inside as well)
            rgba_output, depth, _, _, _ = model(ray_batch, depth_quantiles=depth_quantiles)
        with sec("loss"):
            opacity = rgba_output[..., -1:]
            rgb_output = rgba_output[..., :3] + (1 - opacity) if white_background else rgba_output[..., :3]
            color_loss = rgb_loss(rgb_batch, rgb_output)
            opacity_loss = ((alpha_batch - opacity) ** 2).mean()
            valid_depth_mask = (depth > 0).all(dim=-1)
            quant_loss = ((depth[..., 0] - depth[..., 1]).abs() * valid_depth_mask).mean()
            w_depth = quantile_weight * min(2 * i / max(iterations, 1), 1)
            loss = color_loss.mean() + opacity_loss + w_depth * quant_loss
        model.optimizer.zero_grad(set_to_none=True)
        event = torch.cuda.Event() if on_gpu else None
        if on_gpu:
            event.record()
        with sec("loss_backward"):      # autograd: loss graph + tracer_backward (timed inside) + softplus / cat
            loss.backward()
        if on_gpu:
            event.synchronize()         # train.py:209-211 (hides the data loading behind the backward pass)
        with sec("batch_fetch"):
            ray_batch, rgb_batch, alpha_batch, share = next_batch()
        with sec("optimizer_step"):
            model.optimizer.step()
            model.update_learning_rate(i)
        if i % 50 == 0 or i == iterations - 1:
            losses.append((i, float(loss.detach())))
        if since_update >= period:
            with sec("update_triangulation_incremental"):
                model.update_triangulation(incremental=True)
            rebuilds["incremental"] += 1
            if world > 1:
                with sec("replica_check"):
                    _check_replicas(rdist, model)
            since_update = 0
            if period < 100:
                period += 2
        since_update += 1
        if i + 1 >= densify_from:
            since_dens += 1
        if densified is None and since_dens == next_dens and model.primal_points.shape[0] < 0.9 * model.num_final_points:
            sec.prefix = "densification:"       # its tracer calls are not the iterations'
            with sec("collect_error_map"):
                if world > 1:       # whole views on every rank: rows traced per rank, outputs gathered, sums exchanged
                    with pipe.replicated_inputs():
                        point_error, point_contribution = model.collect_error_map(handler, white_background)
                else:
                    point_error, point_contribution = model.collect_error_map(handler, white_background)
            with sec("prune_and_densify"):
                model.prune_and_densify(point_error, point_contribution, densify_factor)
            with sec("update_triangulation_full"):
                model.update_triangulation(incremental=False)
            if world > 1:
                _check_replicas(rdist, model)
            sec.prefix = ""
            rebuilds["full"] += 1
            period = 1
            gc.collect()
            since_dens = 0
            densified = {"iteration": i, "points_before": n0, "points_after": int(model.primal_points.shape[0])}
            next_dens = 10 ** 9         # one densification per run
    sync()
    wall = time.perf_counter() - t_loop
    sec.collect()
    if world > 1:
        _check_replicas(rdist, model)
        rdist.disable_data_parallel()

    ms = {k: round(v, 2) for k, v in sec.ms.items()}
    per_it = lambda k: round(sec.ms.get(k, 0.0) / iterations, 3)
    inner = ("tracer_forward", "get_trace_data", "get_starting_point")
    detail = {
        "reference_model_from": where, "iterations": iterations, "rays_per_batch": rays_per_batch,
        "world_size": world, "rays_per_rank": local_rays,
        "last_exchange": getattr(pipe, "last_exchange", None),
        "points": n0, "sh_degree": sh_degree, "training_views": f"{cameras} x {height}x{width}",
        "wall_seconds": round(wall, 2), "wall_ms_per_iteration": round(wall / iterations * 1e3, 2),
        "rebuilds": rebuilds, "densification": densified, "loss_trace": losses,
        "loss_first": losses[0][1] if losses else None, "loss_last": losses[-1][1] if losses else None,
        "gpu_ms_total_by_section": ms, "calls_by_section": dict(sec.calls),
        # one iteration, averaged over the run (ms of the loop's stream): the split VERDICT r3 #3 asks for
        "ms_per_iteration": {
            "tracer_forward": per_it("tracer_forward"), "tracer_backward": per_it("tracer_backward"),
            # world > 1: trace_backward = the kernels + the gradient exchange inside it
            "tracer_backward_kernels": per_it("tracer_backward_kernels") if world > 1 else per_it("tracer_backward"),
            "gradient_exchange": round((sec.ms.get("tracer_backward", 0.0) - sec.ms.get("tracer_backward_kernels", 0.0))
                                       / iterations, 3) if world > 1 else 0.0,
            "replica_check": per_it("replica_check"),
            "get_trace_data": per_it("get_trace_data"), "get_starting_point": per_it("get_starting_point"),
            "model_forward_other": round((sec.ms.get("model_forward", 0.0) - sum(sec.ms.get(k, 0.0) for k in inner))
                                         / iterations, 3),
            "depth_quantiles": per_it("depth_quantiles"), "loss": per_it("loss"),
            "autograd_backward_other": round((sec.ms.get("loss_backward", 0.0) - sec.ms.get("tracer_backward", 0.0))
                                             / iterations, 3),
            "batch_fetch": per_it("batch_fetch"), "optimizer_step": per_it("optimizer_step"),
            "update_triangulation_incremental": per_it("update_triangulation_incremental"),
            "densification_total": round(sum(sec.ms.get(k, 0.0) for k in
                                             ("collect_error_map", "prune_and_densify", "update_triangulation_full"))
                                         / iterations, 3),
        },
        "ms_per_call": {k: round(sec.ms[k] / sec.calls[k], 3) for k in sec.ms},
    }
    return iterations / wall, detail
