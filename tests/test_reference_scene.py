"""The reference's OWN Python model -- radfoam_model/scene.py::RadFoamScene and render.py::TraceRays, unmodified --
driven against this repository's ``radfoam`` package (missing #1 / next #2).

Where the files come from: /root/reference/radfoam_model when that exists (the build container), otherwise the
byte-for-byte copy oracle/Makefile.ref puts under the git-ignored oracle/_ref/pyref/ so that it travels to the GPU box
with a gpurun snapshot (nothing of it is committed; the product never imports it).  ``plyfile`` (save_ply only) is
stubbed; everything else the two files import is present.

What is driven, in the order train.py does it (train.py:93-107,162-270): construction (random_initialize ->
radfoam.Triangulation -> permutation -> build_aabb_tree), declare_optimizer, forward with and without an explicit
start_point (torch.unique + radfoam.nn + the uint32 <-> long round trips of scene.py:224-234; the broadcast_to,
non-contiguous uint32 start_point of :250), depth quantiles + return_contribution, loss.backward() through the
reference's TraceRays, an Adam step, update_triangulation(incremental=True), collect_error_map (BatchFetcher,
ErrorBox.ray_error -> point_error), prune_and_densify (radfoam.farthest_neighbor), the full rebuild after it,
save_pt / load_pt.  The rendered rgba is compared with the CPU oracle on the scene's own trace data (bit for bit),
the parameter gradients with the oracle's backward pushed through the same torch graph.

  * ``-m gpu``: the real thing -- HIP tracer, GPU triangulation, GPU nn / farthest_neighbor -- on cuda:0;
  * CPU (no GPU here): the same script with the tracer replaced by the oracle wrapped in the Pipeline interface, which
    checks the script itself and the CPU shims (Qhull Triangulation, torch nn / farthest_neighbor / BatchFetcher)
    against the reference's callers.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CANDIDATES = ["/root/reference", os.path.join(ROOT, "oracle", "_ref", "pyref")]
PYREF = next((p for p in _CANDIDATES if os.path.isfile(os.path.join(p, "radfoam_model", "scene.py"))), None)

pytestmark = pytest.mark.skipif(PYREF is None, reason="the reference's radfoam_model is neither under /root/reference "
                                                      "nor under oracle/_ref/pyref (make -C oracle -f Makefile.ref pyref)")


@pytest.fixture
def ref_model(monkeypatch):
    """(scene module, render module) of the reference, imported fresh against this repository's radfoam."""
    monkeypatch.syspath_prepend(PYREF)
    ply = types.ModuleType("plyfile")   # scene.py:5 imports it for save_ply only
    ply.PlyData = ply.PlyElement = object
    monkeypatch.setitem(sys.modules, "plyfile", ply)
    for name in [m for m in sys.modules if m.startswith("radfoam_model")]:
        monkeypatch.delitem(sys.modules, name)
    import radfoam
    render = importlib.import_module("radfoam_model.render")
    scene = importlib.import_module("radfoam_model.scene")
    assert scene.radfoam is radfoam
    assert os.path.realpath(scene.__file__).startswith(os.path.realpath(PYREF))
    yield scene, render
    for name in [m for m in sys.modules if m.startswith("radfoam_model")]:
        sys.modules.pop(name, None)


class _OraclePipeline:
    """Pipeline-shaped wrapper over the CPU oracle with the reference binding's signature (test infrastructure)."""

    def __init__(self, sh_degree, attr_dtype=torch.float32):
        self.d = sh_degree

    def trace_forward(self, points, attributes, adj, off, rays, start_point, depth_quantiles=None,
                      weight_threshold=None, max_intersections=None, return_contribution=False):
        from oracle import oracle as O
        npy = lambda t: None if t is None else t.detach().contiguous().numpy()
        out = O.trace_forward(self.d, npy(points), npy(attributes), npy(adj), npy(off), npy(rays),
                              npy(start_point.contiguous()), depth_quantiles=npy(depth_quantiles),
                              weight_threshold=weight_threshold, max_intersections=max_intersections,
                              return_contribution=return_contribution)
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
                               weight_threshold=weight_threshold, max_intersections=max_intersections)
        return {k: torch.from_numpy(v) for k, v in out.items()}


def _camera_rays(position, width, height, device):
    """[H, W, 6] rays of a pinhole camera at `position` looking at the origin."""
    from radfoam_amd import foam
    cam = foam.default_camera(width, height)
    pos = np.asarray(position, dtype=np.float32)
    fwd = -pos / np.linalg.norm(pos)
    up0 = np.array([0.0, 1.0, 0.0], dtype=np.float32)
    right = np.cross(up0, fwd)
    right /= np.linalg.norm(right)
    cam.update(position=pos, forward=fwd.astype(np.float32), right=right.astype(np.float32),
               up=np.cross(fwd, right).astype(np.float32), fov=0.9)
    return torch.from_numpy(foam.camera_rays(cam)).to(device)


def _oracle_forward(scene_obj, rays, start_point, q=None, contribution=False):
    from oracle import oracle as O
    pts, att, adj, off = (t.detach().cpu().numpy() for t in scene_obj.get_trace_data())
    return O.trace_forward(scene_obj.sh_degree, pts, att, adj.astype(np.uint32), off.astype(np.uint32),
                           rays.cpu().numpy(), start_point.cpu().numpy().astype(np.uint32),
                           depth_quantiles=None if q is None else q.cpu().numpy(), return_contribution=contribution)


def _drive(scene_mod, device, n_init=6000, check_oracle=True):
    """The script described in the module docstring; returns a dict of things the callers assert on."""
    from types import SimpleNamespace

    from oracle import oracle as O
    import radfoam

    dev = torch.device(device)
    torch.manual_seed(7)
    sh = 2
    margs = SimpleNamespace(sh_degree=sh, init_points=n_init, final_points=4 * n_init, activation_scale=1.0)
    oargs = SimpleNamespace(points_lr_init=2e-4, points_lr_final=5e-6, density_lr_init=1e-1, density_lr_final=1e-2,
                            attributes_lr_init=5e-3, attributes_lr_final=5e-4, sh_factor=0.1, freeze_points=18_000)
    model = scene_mod.RadFoamScene(margs, device=dev)
    n = model.primal_points.shape[0]
    assert n == n_init and model.point_adjacency.dtype == torch.uint32
    assert model.point_adjacency_offsets.numel() == n + 1 and model.aabb_tree.shape[1:] == (2, 3)
    with torch.no_grad():   # something to see: random colours, densities that make the cloud semi-opaque
        model.att_dc.copy_(0.5 * torch.randn_like(model.att_dc))
        model.att_sh.copy_(0.2 * torch.randn_like(model.att_sh))
        model.density.copy_(-0.25 + 0.1 * torch.randn_like(model.density))
    model.declare_optimizer(oargs, warmup=100, max_iterations=1000)
    out = {}

    # ---- 1. a frame, start point looked up by the scene (torch.unique -> radfoam.nn -> uint32) -----------------
    rays = _camera_rays((0.0, 10.0, -160.0), 64, 48, dev)
    rgba, depth, contrib, nint, box = model(rays)
    assert depth is None and contrib is None
    assert rgba.shape == (48, 64, 4) and nint.dtype == torch.uint32
    start = model.get_starting_point(rays, model.primal_points, model.aabb_tree)
    assert start.dtype == torch.uint32 and start.shape == rays.shape[:-1]
    d2 = ((model.primal_points.detach() - rays[0, 0, :3]) ** 2).sum(-1)
    assert int(start.reshape(-1)[0].item()) == int(d2.argmin())
    if check_oracle:
        ref = _oracle_forward(model, rays, start)
        assert np.array_equal(rgba.detach().cpu().numpy().view(np.uint32), ref["rgba"].view(np.uint32))
        assert np.array_equal(nint.cpu().numpy().view(np.uint32), ref["num_intersections"])
        assert float(ref["rgba"][..., 3].max()) > 0.5   # the frame does hit the cloud

    # ---- 2. explicit (0-dim, broadcast) start point, depth quantiles, contribution; backward; Adam ---------------
    q = torch.rand(*rays.shape[:-1], 2, device=dev).sort(dim=-1, descending=True).values
    sp0 = start.reshape(-1)[0]
    rgba2, depth2, contrib2, nint2, box2 = model(rays, start_point=sp0, depth_quantiles=q, return_contribution=True)
    assert depth2.shape == rays.shape[:-1] + (2,) and contrib2.shape == (n, 1)
    assert torch.equal(rgba2.detach(), rgba.detach())
    w = torch.randn(rgba2.shape, device=dev)
    wd = 0.01 * torch.randn(depth2.shape, device=dev)
    loss = (rgba2 * w).sum() + (depth2 * wd).sum()
    model.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    for p in (model.primal_points, model.att_dc, model.att_sh, model.density):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all())
    assert float(model.primal_points.grad.abs().max()) > 0 and float(model.att_sh.grad.abs().max()) > 0
    if check_oracle:
        ref2 = _oracle_forward(model, rays, start, q, contribution=True)
        assert np.array_equal(depth2.detach().cpu().numpy().view(np.uint32), ref2["depth"].view(np.uint32))
        np.testing.assert_allclose(contrib2.detach().cpu().numpy(), ref2["contribution"], rtol=1e-4, atol=1e-6)
        pts, att, adj, off = (t.detach().cpu().numpy() for t in model.get_trace_data())
        rb = O.trace_backward(sh, pts, att, adj.astype(np.uint32), off.astype(np.uint32), rays.cpu().numpy(),
                              start.cpu().numpy().astype(np.uint32), ref2["rgba"], w.cpu().numpy(),
                              depth_quantiles=q.cpu().numpy(), depth_indices=ref2["depth_indices"],
                              depth_grad_in=wd.cpu().numpy())
        pg = np.where(np.isfinite(rb["points_grad"]), rb["points_grad"], 0.0)
        ag = np.where(np.isfinite(rb["attr_grad"]), rb["attr_grad"], 0.0)
        rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30)
        assert rel(model.primal_points.grad.cpu().numpy(), pg) < 1e-4
        assert rel(model.att_dc.grad.cpu().numpy(), ag[:, :3]) < 1e-4
        assert rel(model.att_sh.grad.cpu().numpy(), ag[:, 3:-1]) < 1e-4
        # density: through activation_scale * softplus(x, beta=10)
        dsp = torch.sigmoid(10.0 * model.density.detach()).cpu().numpy()
        assert rel(model.density.grad.cpu().numpy(), ag[:, -1:] * dsp) < 1e-4
    before = model.primal_points.detach().clone()
    model.optimizer.step()
    model.update_learning_rate(1)
    assert not torch.equal(before, model.primal_points.detach())

    # ---- 3. the triangulation follows the points (train.py:243-248) ---------------------------------------------
    model.update_triangulation(incremental=True)
    assert model.point_adjacency_offsets.numel() == model.primal_points.shape[0] + 1
    rgba3 = model(rays)[0]
    if check_oracle:
        start3 = model.get_starting_point(rays, model.primal_points, model.aabb_tree)
        ref3 = _oracle_forward(model, rays, start3)
        assert np.array_equal(rgba3.detach().cpu().numpy().view(np.uint32), ref3["rgba"].view(np.uint32))

    # ---- 4. a shuffled flat batch from two cameras (train.py:61: what training feeds the tracer) -----------------
    r2 = torch.cat([rays.reshape(-1, 6), _camera_rays((120.0, -20.0, 90.0), 160, 120, dev).reshape(-1, 6)])
    r2 = r2[torch.randperm(r2.shape[0], device=dev)].contiguous()
    rgba4 = model(r2)[0]
    assert rgba4.shape == (r2.shape[0], 4) and bool(torch.isfinite(rgba4).all())
    if check_oracle:
        start4 = model.get_starting_point(r2, model.primal_points, model.aabb_tree)
        ref4 = _oracle_forward(model, r2, start4)
        assert np.array_equal(rgba4.detach().cpu().numpy().view(np.uint32), ref4["rgba"].view(np.uint32))

    # ---- 5. densification statistics, pruning and densification, full rebuild (train.py:255-268) -----------------
    if dev.type == "cuda":   # collect_error_map moves its rays with .cuda() (scene.py:502)
        frames = torch.stack([_camera_rays((0.0, 10.0, -160.0), 32, 32, "cpu"),
                              _camera_rays((150.0, 0.0, 30.0), 32, 32, "cpu")])
        handler = SimpleNamespace(rays=frames, rgbs=torch.rand(2, 32, 32, 3))
        point_error, point_contribution = model.collect_error_map(handler, white_bkg=True)
    else:
        rg, _, point_contribution, _, ebox = model(rays, start_point=sp0, return_contribution=True)
        ebox.ray_error = torch.rand(rays.shape[:-1], device=dev)
        model.optimizer.zero_grad(set_to_none=True)
        rg.sum().backward()
        point_error = ebox.point_error
        assert point_error is not None and point_error.shape == (model.primal_points.shape[0], 1)
        model.optimizer.zero_grad(set_to_none=True)
    assert point_error.shape == (n, 1) and point_contribution.shape == (n, 1)
    assert float(point_error.detach().sum()) > 0 and float(point_contribution.detach().max()) > 0
    model.prune_and_densify(point_error, point_contribution, upsample_factor=1.15)
    n2 = model.primal_points.shape[0]
    assert n < n2 <= int(1.15 * n) + 1
    model.update_triangulation(incremental=False)
    assert model.point_adjacency_offsets.numel() == n2 + 1 and model.att_sh.shape[0] == n2
    rgba5, _, _, _, _ = model(rays)
    model.optimizer.zero_grad(set_to_none=True)
    (rgba5 * w).sum().backward()
    model.optimizer.step()
    assert bool(torch.isfinite(model.primal_points).all())
    out["points_after_densify"] = n2

    # ---- 6. checkpoint round trip (scene.py:614-656) -------------------------------------------------------------
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "model.pt")
        model.update_triangulation(incremental=True)
        want = model(rays)[0].detach()
        model.save_pt(path)
        again = scene_mod.RadFoamScene(SimpleNamespace(sh_degree=sh, init_points=64, final_points=128,
                                                       activation_scale=1.0), device=dev)
        again.load_pt(path)
        assert again.point_adjacency.dtype == torch.uint32 and again.primal_points.shape[0] == n2
        with torch.no_grad():
            got = again(rays)[0]
        assert torch.equal(got, want)
    return out


def test_reference_scene_runs_on_the_cpu_shims(ref_model, monkeypatch):
    """CPU dry run of the script: the reference's RadFoamScene against this package's CPU shims, with the tracer
    replaced by the oracle behind the Pipeline interface (the product has no CPU tracer).  The rgba comparison with
    the oracle is skipped: it would compare the oracle with itself."""
    import radfoam
    scene, _ = ref_model
    monkeypatch.setattr(radfoam, "create_pipeline", lambda d, dt="float32": _OraclePipeline(d, dt))
    res = _drive(scene, "cpu", n_init=1500, check_oracle=False)
    assert res["points_after_densify"] > 1500


@pytest.mark.gpu
def test_reference_scene_and_trace_rays_on_the_gpu(ref_model):
    """The reference's unmodified RadFoamScene / TraceRays on cuda:0 through the HIP tracer, the GPU triangulation
    and the GPU scene ops; rgba / depths bit-equal to the oracle on the scene's own trace data, parameter gradients
    within 1e-4 of the oracle's backward."""
    scene, render = ref_model
    import radfoam
    from radfoam_amd.pipeline import Pipeline
    assert isinstance(radfoam.create_pipeline(2), Pipeline)   # the HIP pipeline, not a stand-in
    res = _drive(scene, "cuda", n_init=6000, check_oracle=True)
    assert res["points_after_densify"] > 6000


@pytest.mark.gpu
def test_no_grad_renders_of_the_reference_scene_record_no_trail(ref_model):
    """RadFoamScene passes its nn.Parameter points even under torch.no_grad(); an evaluation render must
    neither allocate nor write the hop trail (trail_steps * 4 B per ray), a training forward must."""
    from types import SimpleNamespace
    scene, _ = ref_model
    model = scene.RadFoamScene(SimpleNamespace(sh_degree=1, init_points=2000, final_points=4000,
                                               activation_scale=1.0), device=torch.device("cuda"))
    rays = _camera_rays((0.0, 10.0, -160.0), 64, 48, "cuda")
    with torch.no_grad():
        model(rays)
    assert model.pipeline._trail is None
    model(rays)
    assert model.pipeline._trail is not None


def _loop_case(device, points, sh, **kw):
    """examples/train_loop.py (what `bench.py --workload train-loop` runs) on a small foam."""
    from examples import train_loop
    from radfoam_amd import foam
    fm = foam.make_synthetic_foam(points, sh, 13)
    env = {"torch": torch, "dev": torch.device(device)}
    its, detail = train_loop.run(None, env, fm, sh_degree=sh, **kw)
    per = detail["ms_per_iteration"]
    assert its > 0 and detail["iterations"] == kw["iterations"]
    assert per["tracer_forward"] > 0 and per["tracer_backward"] > 0 and per["optimizer_step"] > 0
    assert detail["calls_by_section"]["tracer_forward"] == kw["iterations"]          # densification booked apart
    assert detail["calls_by_section"]["tracer_backward"] == kw["iterations"]
    # train.py:243-248: rebuilds after iterations 0, 3, 8, 15, ... (period 1, +2 per rebuild)
    want = sum(1 for k in range(1, 100) if k * k <= kw["iterations"])
    assert detail["rebuilds"]["incremental"] >= want
    assert all(np.isfinite(l) for _, l in detail["loss_trace"])
    return detail


def test_train_loop_script_dry_run_on_the_cpu_shims(ref_model, monkeypatch):
    """CPU: the loop's own logic (fetchers, quantiles, losses, rebuild schedule, section bookkeeping) with the tracer
    replaced by the oracle behind the Pipeline interface; no densification (collect_error_map needs .cuda())."""
    import radfoam
    monkeypatch.setattr(radfoam, "create_pipeline", lambda d, dt="float32": _OraclePipeline(d, dt))
    d = _loop_case("cpu", 1500, 1, iterations=5, rays_per_batch=1500, cameras=2, width=48, height=32,
                   densify_at=10 ** 9)
    assert d["densification"] is None and d["rebuilds"]["full"] == 0


@pytest.mark.gpu
def test_train_loop_smoke_on_the_gpu(ref_model):
    """-m gpu smoke of `bench.py --workload train-loop`: the reference's RadFoamScene through 24
    iterations of the loop with the rebuild schedule and one densification, HIP tracer + GPU triangulation + the
    device-resident BatchFetcher."""
    d = _loop_case("cuda", 20_000, 3, iterations=24, rays_per_batch=30_000, cameras=3, width=160, height=120,
                   densify_at=12)
    dens = d["densification"]
    # prune_and_densify adds 15 % and prunes what contributes nothing (the synthetic foam's empty shell): the count changes
    assert dens is not None and dens["points_after"] != dens["points_before"] and d["rebuilds"]["full"] == 1
    assert 32 < dens["points_after"] <= int(1.15 * dens["points_before"]) + 1
    assert d["calls_by_section"]["densification:tracer_forward"] == 3                 # one per training view


@pytest.mark.gpu
def test_train_loop_has_something_to_learn(ref_model):
    """the loop's targets are renders of ANOTHER state of the foam (densities doubled inside a sphere and
    halved around it, a smooth colour field added), so the reference's own losses, Adam and schedules must bring the loss
    down -- by more than a third within 80 iterations on this small scene."""
    d = _loop_case("cuda", 20_000, 3, iterations=80, rays_per_batch=30_000, cameras=3, width=160, height=120,
                   densify_at=10 ** 9)
    assert d["loss_last"] < 0.66 * d["loss_first"], d["loss_trace"]
