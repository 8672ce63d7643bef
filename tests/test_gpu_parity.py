"""-m gpu: the HIP path (through the radfoam boundary -> C-ABI) against the CPU oracle on the
same seeded inputs.

Tolerances (north star: 1e-4 RGB, 1e-3 relative gradient):
  * forward outputs of fp32 pipelines -- rgba, depth, depth_indices, num_intersections -- must be
    BIT-IDENTICAL to the oracle (both evaluate the same pinned fp32 arithmetic);
  * scatter outputs (contribution, points_grad, attr_grad, point_error) are sums whose order
    differs (atomics): rtol 1e-3 per element (helpers.grad_close), and 1e-5 relative L2.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _pipeline(d, dtype=torch.float32):
    """Pipeline with the hop trail always recorded: these tests drive trace_forward / trace_backward by
    hand on plain tensors, where the default ("auto": record when points or attributes require grad)
    would leave every backward on the re-walk kernel.  The default itself is covered by
    test_trail_is_recorded_only_when_a_backward_can_follow."""
    import radfoam

    pipe = radfoam.create_pipeline(d, dtype)
    pipe.record_trail = True
    return pipe


def _run_forward(pipe, fm, rays, start, attr_dtype=None, **kw):
    p, a, adj, off = H.to_torch_foam(fm, DEV, attr_dtype)
    r = torch.from_numpy(rays).to(DEV)
    if np.ndim(start) == 0:
        s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    else:
        s = torch.from_numpy(np.asarray(start, dtype=np.uint32)).to(DEV)
    if "depth_quantiles" in kw and kw["depth_quantiles"] is not None:
        kw["depth_quantiles"] = torch.from_numpy(kw["depth_quantiles"]).to(DEV)
    out = pipe.trace_forward(p, a, adj, off, r, s, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}, (p, a, adj, off, r, s)


@pytest.mark.parametrize("forward_mode", [1, 2, 4, 5])
@pytest.mark.parametrize("d", [0, 1, 2, 3])
def test_forward_image_bit_exact(foam_factory, d, forward_mode):
    """forward_mode: 1 = face blocks requested one at a time (what large image launches run), 2 = the first six of a
    cell together (flat batches and small launches: what mode 0 picks for every other test of this file), 4 = persistent
    waves that refill their dead lanes from a queue (experiment: ballot / prefix-sum compaction of live rays), 5 = the
    eager instance behind a block-level LDS table of cell records and face blocks (experiment)."""
    fm = foam_factory(6000, d, 11)
    cam, rays, start = H.camera_setup(fm, 96, 64)
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], rays, start)
    pipe = _pipeline(d)
    pipe.forward_mode = forward_mode
    got, _ = _run_forward(pipe, fm, rays, start)
    assert got["rgba"].shape == (64, 96, 4) and got["num_intersections"].shape == (64, 96, 1)
    np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
    np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    assert ref["rgba"][..., 3].max() > 0.5  # the scene is actually hit


@pytest.mark.parametrize("forward_mode", [1, 2, 5])
@pytest.mark.parametrize("d", [0, 2, 3])
def test_forward_flat_rays_quantiles_contribution(foam_factory, d, forward_mode):
    fm = foam_factory(6000, d, 12)
    rays, starts = H.random_rays(fm, 5000, seed=3)
    rng = np.random.default_rng(5)
    q = np.sort(rng.uniform(0.02, 0.98, size=(5000, 2)).astype(np.float32), axis=1)[:, ::-1].copy()
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], rays, starts, depth_quantiles=q,
                          return_contribution=True, num_threads=1)
    pipe = _pipeline(d)
    pipe.forward_mode = forward_mode
    got, _ = _run_forward(pipe, fm, rays, starts, depth_quantiles=q, return_contribution=True)
    np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
    np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    np.testing.assert_array_equal(got["depth_indices"].numpy().view(np.uint32), ref["depth_indices"])
    np.testing.assert_array_equal(got["depth"].numpy().view(np.uint32), ref["depth"].view(np.uint32))
    assert (ref["depth_indices"] == 0xFFFFFFFF).any() and (ref["depth_indices"] != 0xFFFFFFFF).any()
    ok, rel, worst = H.grad_close(got["contribution"].numpy(), ref["contribution"])
    assert ok and rel < 1e-5, (rel, worst)
    # invariant: sum of weights == sum of alpha
    assert abs(float(got["contribution"].double().sum()) - float(got["rgba"][..., 3].double().sum())) < 1e-2


@pytest.mark.parametrize("forward_mode", [1, 2, 5])
@pytest.mark.parametrize("d", [0, 1, 2, 3])
def test_forward_half_attributes(foam_factory, d, forward_mode):
    fm = foam_factory(6000, d, 13)
    fm16 = dict(fm)
    fm16["attributes"] = fm["attributes"].astype(np.float16)
    cam, rays, start = H.camera_setup(fm, 64, 48)
    ref = O.trace_forward(d, fm16["points"], fm16["attributes"], fm16["point_adjacency"],
                          fm16["point_adjacency_offsets"], rays, start, return_contribution=True)
    pipe = _pipeline(d, torch.float16)
    pipe.forward_mode = forward_mode
    got, _ = _run_forward(pipe, fm16, rays, start, return_contribution=True)
    assert got["rgba"].dtype == torch.float16 and got["contribution"].dtype == torch.float16
    np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
    np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint16), ref["rgba"].view(np.uint16))
    np.testing.assert_allclose(got["contribution"].float().numpy(), ref["contribution"].astype(np.float32),
                               rtol=2e-3, atol=1e-3)


def test_forward_settings_and_edge_cases(foam_factory):
    d = 1
    fm = foam_factory(6000, d, 14)
    cam, rays, start = H.camera_setup(fm, 40, 30)
    pipe = _pipeline(d)
    for thr, mi in [(0.05, None), (None, 7), (0.3, 3), (None, 0)]:
        ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                              fm["point_adjacency_offsets"], rays, start, weight_threshold=thr,
                              max_intersections=mi)
        got, _ = _run_forward(pipe, fm, rays, start, weight_threshold=thr, max_intersections=mi)
        np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
        np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    # rays pointing away from the foam from outside: leave through the first (unbounded) cell
    away = rays.copy()
    away[..., 3:] *= -1.0
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], away, start)
    got, _ = _run_forward(pipe, fm, away, start)
    np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
    np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    assert float(got["rgba"].abs().max()) == 0.0
    # empty batch
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    out = pipe.trace_forward(p, a, adj, off, torch.zeros((0, 6), device=DEV),
                             torch.zeros((0,), dtype=torch.uint32, device=DEV))
    assert out["rgba"].shape == (0, 4)
    # launch options outside their range are refused by the C-ABI, not guessed at
    for knob in ("forward_mode", "backward_mode"):
        bad = _pipeline(d)
        setattr(bad, knob, 7)
        with pytest.raises(RuntimeError, match=knob):
            r, s = torch.from_numpy(rays).to(DEV), torch.full(rays.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
            f = pipe.trace_forward(p, a, adj, off, r, s)
            if knob == "forward_mode":
                bad.trace_forward(p, a, adj, off, r, s)
            else:
                bad.trace_backward(p, a, adj, off, r, s, f["rgba"], torch.ones_like(f["rgba"]))


def _backward_case(foam_factory, d, seed, image, quantiles, with_error, n_points=5000):
    fm = foam_factory(n_points, d, seed)
    rng = np.random.default_rng(seed)
    if image:
        cam, rays, start = H.camera_setup(fm, 80, 56)
        starts = np.full(rays.shape[:-1], start, dtype=np.uint32)
    else:
        rays, starts = H.random_rays(fm, 4000, seed=seed + 1)
    batch = rays.shape[:-1]
    q = dg = None
    if quantiles:
        q = np.sort(rng.uniform(0.02, 0.98, size=batch + (2,)).astype(np.float32), axis=-1)[..., ::-1].copy()
        dg = rng.normal(size=batch + (2,)).astype(np.float32)
    g = rng.normal(size=batch + (4,)).astype(np.float32)
    err = rng.uniform(0, 1, size=batch).astype(np.float32) if with_error else None
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    fwd = O.trace_forward(*args, rays, starts, depth_quantiles=q)
    ref = O.trace_backward(*args, rays, starts, fwd["rgba"], g, depth_quantiles=q,
                           depth_indices=fwd.get("depth_indices"), depth_grad_in=dg, ray_error=err,
                           num_threads=1)
    return fm, rays, starts, q, dg, g, err, fwd, ref


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("trail", ["rewalk", "replay", "short"])
@pytest.mark.parametrize("d,image,quantiles,with_error", [
    (0, True, False, False), (1, False, True, True), (2, True, True, False), (2, False, False, True),
    (3, True, False, False), (3, False, True, True),
])
def test_backward_parity(foam_factory, d, image, quantiles, with_error, mode, trail):
    """trail: 'rewalk' = backward re-scans every cell (no forward call on this pipeline);
    'replay' = backward replays the hop trail recorded by trace_forward; 'short' = the trail holds
    only 5 hops per ray, the rest is re-scanned."""
    fm, rays, starts, q, dg, g, err, fwd, ref = _backward_case(foam_factory, d, 20 + d, image, quantiles, with_error)
    pipe = _pipeline(d)
    pipe.backward_mode = mode
    pipe.forward_mode = 1 + mode % 2    # the trail comes from either forward instance (both write the same one)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: None if x is None else torch.from_numpy(x).to(DEV)
    tr, ts, tq = t(rays), t(starts), t(q)
    if trail == "rewalk":
        pipe.record_trail = False
    else:
        if trail == "short":
            pipe.trail_steps = 5
        f = pipe.trace_forward(p, a, adj, off, tr, ts, depth_quantiles=tq)
        np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
        assert pipe._trail is not None
    out = pipe.trace_backward(p, a, adj, off, tr, ts, t(fwd["rgba"]), t(g), tq,
                              t(fwd.get("depth_indices")), t(dg), t(err))
    torch.cuda.synchronize()
    assert out["points_grad"].shape == (fm["points"].shape[0], 3)
    assert out["attr_grad"].shape == fm["attributes"].shape
    assert out["ray_grad"].shape == rays.shape
    for key in ["points_grad", "attr_grad"] + (["point_error"] if with_error else []):
        ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
        assert ok and rel < 1e-5, (key, rel, worst)
        assert np.abs(ref[key]).max() > 0


@pytest.mark.parametrize("width,height", [(96, 40), (70, 24), (33, 8), (64, 72)])
def test_short_and_ragged_images(foam_factory, width, height):
    """Frames whose height or width is not a multiple of the 16x16 block (partly empty blocks and waves; (33, 8) is below
    the 16-row minimum and goes down the flat-batch path): forward bit for bit in both forward modes, gradients by trail
    replay and by re-walk -- the shapes a frame cut into row blocks over several GPUs produces."""
    d = 2
    fm = foam_factory(5000, d, 31)
    cam, rays, start = H.camera_setup(fm, width, height)
    starts = np.full(rays.shape[:-1], start, dtype=np.uint32)
    rng = np.random.default_rng(7)
    g = rng.normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    fwd = O.trace_forward(*args, rays, starts)
    ref = O.trace_backward(*args, rays, starts, fwd["rgba"], g, num_threads=1)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: torch.from_numpy(x).to(DEV)
    for mode, record, forward_mode in ((3, True, 1), (2, False, 2)):
        pipe = _pipeline(d)
        pipe.backward_mode = mode
        pipe.record_trail = record
        pipe.forward_mode = forward_mode
        f = pipe.trace_forward(p, a, adj, off, t(rays), t(starts))
        np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
        np.testing.assert_array_equal(f["num_intersections"].cpu().numpy().view(np.uint32), fwd["num_intersections"])
        out = pipe.trace_backward(p, a, adj, off, t(rays), t(starts), f["rgba"], t(g))
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
            assert ok and rel < 1e-5, (mode, key, rel, worst)


@pytest.mark.parametrize("rule", ["auto", "xcd", "tail:7", "global", "reversed"])
def test_tile_order_does_not_change_results(foam_factory, rule):
    """The blocks of an image launch take the tiles in an order learnt from the previous frame's hop counts
    (Pipeline.tile_order_mode, rf_launch_opts.tile_order); any order -- the static one backwards included -- gives the same
    forward bit for bit and the same gradients."""
    d = 2
    fm = foam_factory(6000, d, 41)
    cam, rays, start = H.camera_setup(fm, 200, 136)
    starts = np.full(rays.shape[:-1], start, dtype=np.uint32)
    g = np.random.default_rng(9).normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    fwd = O.trace_forward(*args, rays, starts)
    ref = O.trace_backward(*args, rays, starts, fwd["rgba"], g, num_threads=1)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: torch.from_numpy(x).to(DEV)
    tr, ts, tg = t(rays), t(starts), t(g)
    pipe = _pipeline(d)
    pipe.tile_order_mode = "auto" if rule == "reversed" else rule
    pipe.trace_forward(p, a, adj, off, tr, ts)                  # the frame the orders are learnt from
    tiles = pipe._tiles
    assert tiles is not None and tiles["shape"] == (136, 200)
    nt = 13 * 9
    for which in ("forward", "backward"):
        o = tiles[which].cpu().numpy().astype(np.int64)
        assert sorted(o[o < nt].tolist()) == list(range(nt))
    if rule == "reversed":
        tiles["forward"] = torch.flip(tiles["default"], dims=(0,)).to(torch.int32).contiguous()
        tiles["backward"] = tiles["forward"]
    f = pipe.trace_forward(p, a, adj, off, tr, ts)
    np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
    np.testing.assert_array_equal(f["num_intersections"].cpu().numpy().view(np.uint32), fwd["num_intersections"])
    out = pipe.trace_backward(p, a, adj, off, tr, ts, f["rgba"], tg)
    for key in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
        assert ok and rel < 1e-5, (rule, key, rel, worst)


def test_tile_order_of_a_sorted_flat_batch(foam_factory):
    """Flat batches large enough to be traced in the sorted order also learn an order for their 256-slot groups; the second
    batch (other rays, same size) is traced in the order the first one taught, with the same results as ever."""
    d = 1
    fm = foam_factory(6000, d, 43)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    pipe = _pipeline(d)
    pipe.reorder_min_rays = 1024
    t = lambda x: torch.from_numpy(x).to(DEV)
    for seed in (5, 6):
        rays, starts = H.random_rays(fm, 20000, seed=seed)
        g = np.random.default_rng(seed).normal(size=(20000, 4)).astype(np.float32)
        fwd = O.trace_forward(*args, rays, starts)
        ref = O.trace_backward(*args, rays, starts, fwd["rgba"], g, num_threads=1)
        tr, ts = t(rays), t(starts)
        f = pipe.trace_forward(p, a, adj, off, tr, ts)
        np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
        out = pipe.trace_backward(p, a, adj, off, tr, ts, f["rgba"], t(g))
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
            assert ok and rel < 1e-5, (seed, key, rel, worst)
        tiles = pipe._tiles
        assert tiles is not None and tiles["shape"] == ("flat", 20000)
        o = tiles["forward"].cpu().numpy().astype(np.int64)
        nt = (20000 + 255) // 256
        assert sorted(o[o < nt].tolist()) == list(range(nt))


@pytest.mark.parametrize("mode", [2, 3])
def test_backward_with_zero_weight_threshold(foam_factory, mode):
    """weight_threshold = 0 on an opaque foam: the transmittance in the compositing denominators decays
    through the subnormal range to 0, where the short divide sequence alone is not IEEE;
    gradients must still match the oracle's (plain '/')."""
    d = 1
    fm = dict(foam_factory(5000, d, 26))
    fm["attributes"] = fm["attributes"].copy()
    fm["attributes"][:, -1] *= 60.0
    cam, rays, start = H.camera_setup(fm, 48, 40)
    starts = np.full(rays.shape[:-1], start, dtype=np.uint32)
    g = np.random.default_rng(3).normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    fwd = O.trace_forward(*args, rays, starts, weight_threshold=0.0)
    ref = O.trace_backward(*args, rays, starts, fwd["rgba"], g, weight_threshold=0.0, num_threads=1)
    assert fwd["rgba"][..., 3].max() == 1.0          # saturated: T reached exactly 0 somewhere
    pipe = _pipeline(d)
    pipe.backward_mode = mode
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: torch.from_numpy(x).to(DEV)
    f = pipe.trace_forward(p, a, adj, off, t(rays), t(starts), weight_threshold=0.0)
    np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
    np.testing.assert_array_equal(f["num_intersections"].cpu().numpy().view(np.uint32).reshape(-1),
                                  fwd["num_intersections"].reshape(-1))
    out = pipe.trace_backward(p, a, adj, off, t(rays), t(starts), f["rgba"], t(g), weight_threshold=0.0)
    for key in ("points_grad", "attr_grad"):
        got, want = out[key].cpu().numpy(), ref[key]
        # the reference's own arithmetic yields inf / NaN here (0 * inf, inf - inf in the saturated tail; zeroed
        # afterwards by render.py:98-99): they must sit in (nearly) the same places -- sums that overflow in a
        # different order may differ in a handful -- and everything finite in both must agree
        fin_w, fin_g = np.isfinite(want), np.isfinite(got)
        assert (fin_w != fin_g).mean() < 2e-3, (key, (fin_w != fin_g).mean())
        assert 0.5 < fin_w.mean() < 1.0
        both = fin_w & fin_g
        ok, rel, worst = H.grad_close(np.where(both, got, 0.0), np.where(both, want, 0.0))
        assert ok and rel < 1e-5, (key, rel, worst)


def test_autograd_operator_matches_oracle(foam_factory):
    """TraceRays (radfoam_amd/render.py) end to end, incl. the non-finite scrub."""
    from radfoam_amd.render import TraceRays

    d = 2
    fm, rays, starts, q, dg, g, err, fwd, ref = _backward_case(foam_factory, d, 31, True, True, False)
    pipe = _pipeline(d)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    p.requires_grad_(True)
    a.requires_grad_(True)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.from_numpy(starts).to(DEV)
    rgba, depth, contrib, nint, box = TraceRays.apply(pipe, p, a, adj, off, r, s, torch.from_numpy(q).to(DEV), False)
    np.testing.assert_array_equal(rgba.detach().cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
    loss = (rgba * torch.from_numpy(g).to(DEV)).sum() + (depth * torch.from_numpy(dg).to(DEV)).sum()
    loss.backward()
    for got, key in [(p.grad, "points_grad"), (a.grad, "attr_grad")]:
        refg = np.where(np.isfinite(ref[key]), ref[key], 0.0)
        ok, rel, worst = H.grad_close(got.cpu().numpy(), refg)
        assert ok and rel < 1e-5, (key, rel, worst)


@pytest.mark.parametrize("d,half", [(0, False), (2, True), (3, True), (1, False)])
def test_benchmark_path(foam_factory, d, half):
    fm = foam_factory(6000, d, 40 + d)
    if half:
        fm = dict(fm)
        fm["attributes"] = fm["attributes"].astype(np.float16)
    cam, _, start = H.camera_setup(fm, 100, 60)
    diff = O.build_adjacent_diff(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    ref = O.trace_benchmark(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                            fm["point_adjacency_offsets"], diff, cam, start, weight_threshold=0.05)
    pipe = _pipeline(d, torch.float16 if half else torch.float32)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    # the table exactly as benchmark.py builds it, without padding
    dtab = pipe.build_adjacent_diff(p, adj, off)
    np.testing.assert_array_equal(dtab.cpu().numpy().view(np.uint16), diff)
    out = torch.zeros((60, 100), dtype=torch.uint32, device=DEV)
    camera = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    sp = torch.tensor([int(start)], dtype=torch.int64).to(torch.uint32).to(DEV)
    pipe.trace_benchmark(p, a, adj, off, dtab, camera, sp, out, weight_threshold=0.05)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref)
    # the second frame of the same camera is walked in the tile order the first one taught (Pipeline.tile_order_mode)
    tiles = pipe._tiles
    assert tiles is not None and tiles["shape"] == (60, 100) and int(tiles["forward"].numel()) % 8 == 0
    out.zero_()
    pipe.trace_benchmark(p, a, adj, off, dtab, camera, sp, out, weight_threshold=0.05)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref)
    assert pipe._tiles is tiles      # same camera: nothing recomputed
    # fisheye: device sin/cos/atan2 differ from glibc by ulps -> channels within 1 LSB
    cam_f = dict(cam)
    cam_f["model"] = "fisheye"
    cam_f["fov"] = 1.2
    ref_f = O.trace_benchmark(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                              fm["point_adjacency_offsets"], diff, cam_f, start, weight_threshold=0.05)
    camera["model"], camera["fov"] = "fisheye", 1.2
    pipe.trace_benchmark(p, a, adj, off, dtab, camera, sp, out, weight_threshold=0.05)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    gb = np.stack([(got >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(np.int32)
    rb = np.stack([(ref_f >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(np.int32)
    assert np.abs(gb - rb).max() <= 1 and (np.abs(gb - rb) > 0).mean() < 0.02


def test_foam_cache_invalidation(foam_factory):
    """The packed-foam cache must notice in-place updates and new tensors."""
    d = 0
    fm = foam_factory(6000, d, 50)
    cam, rays, start = H.camera_setup(fm, 32, 32)
    pipe = _pipeline(d)
    got1, (p, a, adj, off, r, s) = _run_forward(pipe, fm, rays, start)
    with torch.no_grad():
        a[:, -1] *= 0.5  # in-place: bumps _version
    out2 = pipe.trace_forward(p, a, adj, off, r, s)
    fm2 = dict(fm)
    fm2["attributes"] = fm["attributes"].copy()
    fm2["attributes"][:, -1] *= 0.5
    ref2 = O.trace_forward(d, fm2["points"], fm2["attributes"], fm2["point_adjacency"],
                           fm2["point_adjacency_offsets"], rays, start)
    np.testing.assert_array_equal(out2["rgba"].cpu().numpy().view(np.uint32), ref2["rgba"].view(np.uint32))
    assert not np.array_equal(got1["rgba"].numpy(), ref2["rgba"])


@pytest.mark.parametrize("d", [1, 2])
def test_geometry_only_repack_after_an_optimiser_step(foam_factory, d):
    """points / attributes updated in place with the triangulation unchanged: the pipeline repacks
    only cells and face offsets (foam_prepared = 2) and must give what a fresh full pack gives."""
    fm = foam_factory(5000, d, 60 + d)
    cam, rays, start = H.camera_setup(fm, 48, 40)
    pipe = _pipeline(d)
    _, (p, a, adj, off, r, s) = _run_forward(pipe, fm, rays, start)
    rng = np.random.default_rng(3)
    dp = (rng.normal(0, 2e-3, fm["points"].shape)).astype(np.float32)
    da = (rng.normal(0, 5e-2, fm["attributes"].shape)).astype(np.float32)
    da[:, -1] = 0.0
    with torch.no_grad():
        p += torch.from_numpy(dp).to(DEV)
        a += torch.from_numpy(da).to(DEV)
    opts = pipe._launch_opts(p, a, adj, off, r.shape)
    assert opts.foam_prepared == 2
    pipe._cache.invalidate_geometry()
    out = pipe.trace_forward(p, a, adj, off, r, s)
    pts2, att2 = fm["points"] + dp, fm["attributes"] + da
    ref = O.trace_forward(d, pts2, att2, fm["point_adjacency"], fm["point_adjacency_offsets"], rays, start)
    np.testing.assert_array_equal(out["rgba"].cpu().numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    np.testing.assert_array_equal(out["num_intersections"].cpu().numpy().view(np.uint32), ref["num_intersections"])
    # backward through the same (geometry-repacked) workspace and trail
    g = torch.from_numpy(rng.normal(0, 1, out["rgba"].shape).astype(np.float32)).to(DEV)
    res = pipe.trace_backward(p, a, adj, off, r, s, out["rgba"], g)
    refb = O.trace_backward(d, pts2, att2, fm["point_adjacency"], fm["point_adjacency_offsets"], rays, start,
                            ref["rgba"], g.cpu().numpy())
    for key in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(res[key].cpu().numpy(), refb[key])
        assert ok and rel < 1e-5, (key, rel, worst)
        assert np.abs(refb[key]).max() > 0
    # a new adjacency tensor (a rebuilt triangulation) is packed from scratch
    assert pipe._launch_opts(p, a, adj.clone(), off, r.shape).foam_prepared == 0


def test_large_image_properties(foam_factory):
    """Bigger workload (not oracle-sized): size-independent properties."""
    d = 2
    fm = foam_factory(60000, d, 60)
    cam, rays, start = H.camera_setup(fm, 640, 360)
    pipe = _pipeline(d)
    got, (p, a, adj, off, r, s) = _run_forward(pipe, fm, rays, start, return_contribution=True)
    rgba = got["rgba"].double()
    assert torch.isfinite(rgba).all() and float(rgba[..., 3].max()) <= 1.0 and float(rgba.min()) >= 0.0
    # sum of per-cell weights == sum of per-ray alpha
    assert abs(float(got["contribution"].double().sum()) - float(rgba[..., 3].sum())) < 1e-3 * float(rgba[..., 3].sum())
    assert int(got["num_intersections"].to(torch.int64).max()) <= 1025
    # forward is deterministic and independent of the ray->lane mapping (image tiles vs flat list)
    flat = pipe.trace_forward(p, a, adj, off, r.reshape(-1, 6), s.reshape(-1))
    assert torch.equal(flat["rgba"].cpu().reshape(360, 640, 4), got["rgba"])
    # a sampled sub-image agrees bit-for-bit with the oracle
    sub = rays[::9, ::16]
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], sub, start)
    np.testing.assert_array_equal(got["rgba"].numpy()[::9, ::16].view(np.uint32), ref["rgba"].view(np.uint32))
    # backward is linear in the incoming gradient
    g1 = torch.randn(360, 640, 4, device=DEV)
    g2 = torch.randn(360, 640, 4, device=DEV)
    rg = got["rgba"].to(DEV)
    b1 = pipe.trace_backward(p, a, adj, off, r, s, rg, g1)
    b2 = pipe.trace_backward(p, a, adj, off, r, s, rg, g2)
    b12 = pipe.trace_backward(p, a, adj, off, r, s, rg, g1 + g2)
    for k in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close((b1[k] + b2[k]).cpu().numpy(), b12[k].cpu().numpy())
        assert rel < 1e-4, (k, rel)


def test_baseline_config_1():
    """BASELINE.json configs[0] as a parity case: 8,192-point foam (seed 0), 256x256 frame, SH degree 0,
    forward only.  SURVEY 8(d) describes it as the reference's CPU-runnable plumbing case; the product has no CPU
    path, so the same inputs go through the HIP path and must equal the CPU oracle bit for bit."""
    from radfoam_amd import foam
    d = 0
    fm = foam.make_synthetic_foam(8192, d, 0)
    cam, rays, start = H.camera_setup(fm, 256, 256)
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], rays, start, return_contribution=True)
    got, _ = _run_forward(_pipeline(d), fm, rays, start, return_contribution=True)
    np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32).reshape(-1),
                                  ref["num_intersections"].reshape(-1))
    ok, rel, worst = H.grad_close(got["contribution"].numpy(), ref["contribution"])
    assert ok and rel < 1e-5, (rel, worst)
    assert ref["rgba"][..., 3].max() > 0.5


def _full_size_cases():
    """BASELINE.json configs at their full size: config 2 (500k points, seed 1) always -- Qhull on 500k
    points takes about half a minute when the foam is not cached; the 2M-point north-star foam (seed 5)
    only when its cached triangulation travelled with the repository."""
    import os
    from radfoam_amd import foam
    have_2m = os.path.exists(os.path.join(foam.default_cache_dir(), "foam_n2000000_s5.npz"))
    return [
        pytest.param(500_000, 1, id="config2-500k"),
        pytest.param(2_000_000, 5, id="north-star-2M", marks=pytest.mark.skipif(
            not have_2m, reason="the cached triangulation .foam_cache/foam_n2000000_s5.npz is not here "
                                "(Qhull on 2M points takes minutes; python -m radfoam_amd.foam 2000000 5 builds it)")),
    ]


@pytest.mark.parametrize("n_points,seed", _full_size_cases())
def test_full_size_frame_properties(n_points, seed):
    """1080x1920 frame, SH degree 2, forward+backward, on the BASELINE foams: properties that do not need the
    oracle at full size, plus the oracle on a sub-sampled grid of the same rays."""
    from radfoam_amd import foam
    d = 2
    fm = foam.make_synthetic_foam(n_points, d, seed, cache_dir=foam.default_cache_dir())
    cam, rays, start = H.camera_setup(fm, 1920, 1080)
    pipe = _pipeline(d)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    out = pipe.trace_forward(p, a, adj, off, r, s, return_contribution=True)
    rgba = out["rgba"]
    assert torch.isfinite(rgba).all() and float(rgba[..., 3].max()) <= 1.0 and float(rgba.min()) >= 0.0
    alpha_sum = float(rgba[..., 3].double().sum())
    assert alpha_sum > 0.2 * 1920 * 1080 * 0.1          # the foam is actually seen
    assert abs(float(out["contribution"].double().sum()) - alpha_sum) < 1e-3 * alpha_sum
    assert int(out["num_intersections"].to(torch.int64).max()) <= 1025
    # every 40th row / column against the oracle, bit for bit
    sub = np.ascontiguousarray(rays[::40, ::40])
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], sub, start)
    np.testing.assert_array_equal(rgba.cpu().numpy()[::40, ::40].view(np.uint32), ref["rgba"].view(np.uint32))
    np.testing.assert_array_equal(out["num_intersections"].cpu().numpy()[::40, ::40].view(np.uint32),
                                  ref["num_intersections"])
    # backward: replaying the trail == walking again, and linear in the incoming gradient
    gen = torch.Generator(device="cpu").manual_seed(7)
    g1 = torch.randn(1080, 1920, 4, generator=gen).to(DEV)
    g2 = torch.randn(1080, 1920, 4, generator=gen).to(DEV)
    b1 = pipe.trace_backward(p, a, adj, off, r, s, rgba, g1)
    pg1, ag1 = b1["points_grad"].clone(), b1["attr_grad"].clone()
    b2 = pipe.trace_backward(p, a, adj, off, r, s, rgba, g2)
    pg2, ag2 = b2["points_grad"].clone(), b2["attr_grad"].clone()
    b12 = pipe.trace_backward(p, a, adj, off, r, s, rgba, g1 + g2)
    for x, y in ((pg1 + pg2, b12["points_grad"]), (ag1 + ag2, b12["attr_grad"])):
        rel = float((x - y).double().norm() / y.double().norm())
        assert rel < 1e-4, rel
    pipe2 = _pipeline(d)
    pipe2.record_trail = False                            # re-walk, wave-reduced atomics
    w1 = pipe2.trace_backward(p, a, adj, off, r, s, rgba, g1)
    for x, y in ((pg1, w1["points_grad"]), (ag1, w1["attr_grad"])):
        rel = float((x - y).double().norm() / y.double().norm())
        assert rel < 1e-5, rel
    assert torch.isfinite(pg1).all() and torch.isfinite(ag1).all() and float(ag1.abs().max()) > 0


def test_full_frame_gradients_against_the_oracle():
    """BASELINE config 2 at full size (500k points, 1080x1920, SH 2): the gradients of the WHOLE frame
    against the oracle's -- every ray, the block cache under its real load (evictions, bypasses, the
    re-walk launch for rays longer than the trail) -- not only the 5k-point cases above."""
    from radfoam_amd import foam
    d = 2
    fm = foam.make_synthetic_foam(500_000, d, 1, cache_dir=foam.default_cache_dir())
    cam, rays, start = H.camera_setup(fm, 1920, 1080)
    gen = np.random.default_rng(17)
    g = gen.normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    diff = O.build_adjacent_diff(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"], pad=32)
    ref_f = O.trace_forward(*args, rays, start, diff=diff)
    ref_b = O.trace_backward(*args, rays, start, ref_f["rgba"], g, diff=diff)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    tg = torch.from_numpy(g).to(DEV)
    for mode, trail_steps in ((0, 256), (0, 64), (4, 256)):   # default; short trail -> replay + re-walk; direct rows
        pipe = _pipeline(d)
        pipe.backward_mode = mode
        pipe.trail_steps = trail_steps
        out = pipe.trace_forward(p, a, adj, off, r, s)
        np.testing.assert_array_equal(out["rgba"].cpu().numpy().view(np.uint32), ref_f["rgba"].view(np.uint32))
        np.testing.assert_array_equal(out["num_intersections"].cpu().numpy().view(np.uint32)[..., 0],
                                      ref_f["num_intersections"].reshape(1080, 1920))
        res = pipe.trace_backward(p, a, adj, off, r, s, out["rgba"], tg)
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(res[key].cpu().numpy(), ref_b[key])
            assert ok and rel < 1e-5, (mode, trail_steps, key, rel, worst)
            assert np.abs(ref_b[key]).max() > 0
        del pipe, out, res
        torch.cuda.empty_cache()


def test_empty_batch_does_not_poison_the_foam_cache(foam_factory):
    """An empty call on fresh tensors packs nothing and must not leave a cache entry that a later
    non-empty call on the same tensors would trust."""
    d = 1
    fm = foam_factory(5000, d, 91)
    cam, rays, start = H.camera_setup(fm, 40, 32)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], rays, start)
    g = np.random.default_rng(2).normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    refb = O.trace_backward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                            fm["point_adjacency_offsets"], rays, start, ref["rgba"], g)
    e_r = torch.zeros((0, 6), device=DEV)
    e_s = torch.zeros((0,), dtype=torch.uint32, device=DEV)
    for first in ("forward", "backward", "statistics"):
        pipe = _pipeline(d)
        # poison: whatever workspace the allocator hands out next holds garbage
        junk = torch.full((64 << 20,), 0x7F, dtype=torch.uint8, device=DEV)
        del junk
        if first == "forward":
            out0 = pipe.trace_forward(p, a, adj, off, e_r, e_s, return_contribution=True)
            assert out0["rgba"].shape == (0, 4) and out0["contribution"].shape == (5000, 1)
        elif first == "backward":
            b0 = pipe.trace_backward(p, a, adj, off, e_r, e_s, torch.zeros((0, 4), device=DEV),
                                     torch.zeros((0, 4), device=DEV))
            assert b0["points_grad"].shape == (5000, 3) and float(b0["attr_grad"].abs().max()) == 0.0
        else:
            st = pipe.walk_statistics(p, a, adj, off, e_r, e_s)
            assert st["cells_scanned"] == 0
        out = pipe.trace_forward(p, a, adj, off, r, s)
        np.testing.assert_array_equal(out["rgba"].cpu().numpy().view(np.uint32), ref["rgba"].view(np.uint32))
        res = pipe.trace_backward(p, a, adj, off, r, s, out["rgba"], torch.from_numpy(g).to(DEV))
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(res[key].cpu().numpy(), refb[key])
            assert ok and rel < 1e-5, (first, key, rel, worst)


def test_trail_is_recorded_only_when_a_backward_can_follow(foam_factory):
    """Default record_trail="auto": no trail for plain (evaluation) tensors; a trail when points or
    attributes require grad -- which is what TraceRays.forward sees, the reference's included."""
    import radfoam
    from radfoam_amd.render import TraceRays

    d = 0
    fm = foam_factory(5000, d, 92)
    cam, rays, start = H.camera_setup(fm, 32, 32)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    pipe = radfoam.create_pipeline(d)
    assert pipe.record_trail == "auto"
    with torch.no_grad():
        ev = pipe.trace_forward(p, a, adj, off, r, s)
    assert pipe._trail is None
    pg, ag = p.clone().requires_grad_(True), a.clone().requires_grad_(True)
    rgba, _, _, _, _ = TraceRays.apply(pipe, pg, ag, adj, off, r, s, None, False)
    assert pipe._trail is not None and torch.equal(rgba.detach(), ev["rgba"])
    rgba.sum().backward()
    assert pg.grad is not None and float(ag.grad.abs().max()) > 0
    # inference tensors have no version counter: traced uncached instead of failing
    with torch.inference_mode():
        pi, ai = p.clone(), a.clone()
        out = pipe.trace_forward(pi, ai, adj, off, r, s)
        out2 = pipe.trace_forward(pi, ai, adj, off, r, s)
    assert torch.equal(out["rgba"], ev["rgba"]) and torch.equal(out2["rgba"], ev["rgba"])
    assert pipe._cache.key is None or not pipe._cache.lookup((pi, ai, adj, off, None))


def test_invalidate_after_a_write_autograd_cannot_see(foam_factory):
    """`.data` writes do not bump _version (the documented hazard of the foam cache): the stale
    result is what an uninformed cache gives, Pipeline.invalidate() / radfoam.invalidate_caches()
    are the documented remedies."""
    import radfoam

    d = 0
    fm = foam_factory(5000, d, 93)
    cam, rays, start = H.camera_setup(fm, 32, 32)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    fm2 = dict(fm)
    fm2["attributes"] = fm["attributes"].copy()
    fm2["attributes"][:, -1] *= 0.25
    ref2 = O.trace_forward(d, fm2["points"], fm2["attributes"], fm2["point_adjacency"],
                           fm2["point_adjacency_offsets"], rays, start)
    for remedy in ("pipeline", "global"):
        pa = a.clone()
        pipe = _pipeline(d)
        pipe.trace_forward(p, pa, adj, off, r, s)
        v = pa._version
        pa.data[:, -1] *= 0.25
        assert pa._version == v
        if remedy == "pipeline":
            pipe.invalidate()
        else:
            radfoam.invalidate_caches()
        out = pipe.trace_forward(p, pa, adj, off, r, s)
        np.testing.assert_array_equal(out["rgba"].cpu().numpy().view(np.uint32), ref2["rgba"].view(np.uint32))


import glob as _glob
import os as _os

_GOLDEN = sorted(_glob.glob(_os.path.join(_os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", _GOLDEN, ids=[_os.path.basename(p)[:-4] for p in _GOLDEN])
def test_hip_matches_reference_source_goldens(path):
    """The HIP path against outputs of the reference's own kernel source (tests/golden, generated by
    tests/golden/make_golden.py): 1e-5 abs on rgba (north star 1e-4), integer outputs equal, gradients
    within the 1e-3 relative bound."""
    from tests.test_reference_source import check_against_golden, golden_camera

    z = dict(np.load(path))
    d = int(z["sh_degree"])
    half = z["attributes"].dtype == np.float16
    pipe = _pipeline(d, torch.float16 if half else torch.float32)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    p, a, adj, off = t(z["points"]), t(z["attributes"]), t(z["point_adjacency"]), t(z["point_adjacency_offsets"])
    rays, start, q = t(z["rays"]), t(z["start_point"]), t(z.get("depth_quantiles"))
    f = pipe.trace_forward(p, a, adj, off, rays, start, depth_quantiles=q, return_contribution=True)
    b = pipe.trace_backward(p, a, adj, off, rays, start, t(z["ref_rgba"]), t(z["grad_rgba"]), q,
                            t(z.get("ref_depth_indices")), t(z.get("depth_grad")), t(z["ray_error"]))
    diff = pipe.build_adjacent_diff(p, adj, off)
    bench = None
    if "ref_benchmark_rgba8" in z:
        cam = golden_camera(z)
        out = torch.zeros((cam["height"], cam["width"]), dtype=torch.uint32, device=DEV)
        camera = {k: (torch.from_numpy(np.asarray(v)) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
        sp = start.reshape(-1)[:1].contiguous()
        pipe.trace_benchmark(p, a, adj, off, diff, camera, sp, out, weight_threshold=0.05)
        bench = out.cpu().numpy().view(np.uint32)
    torch.cuda.synchronize()
    npy = lambda d_: {k: v.cpu().numpy() for k, v in d_.items() if torch.is_tensor(v)}
    fwd, bwd = npy(f), npy(b)
    for k in ("num_intersections", "depth_indices"):
        if k in fwd:
            fwd[k] = fwd[k].view(np.uint32)
    check_against_golden(z, fwd, bwd, diff.cpu().numpy(), bench, half)


def test_kernel_evaluates_the_references_scan_on_a_frame_with_contested_cells(foam_factory):
    """The kernels find a cell's exit by a cross-multiplied tournament and divide only the winner; cells where two exits
    lie within three floats of each other (the certificate of rf_kernels.hip, "the face scan") go through the dividing
    scan.  On 64,000 rays x ~45 cells the CPU checker's mirror of that evaluation counts the contested cells (there must
    be some, or this test exercises nothing) and the kernel must equal the reference's evaluation -- every face divided,
    running minimum of rounded quotients, (P + o/2) - O, tracing_utils.cuh:43-67 -- bit for bit, in every scheduling
    mode, on every ray."""
    d = 2
    fm = foam_factory(30000, d, 23)
    cam, rays, start = H.camera_setup(fm, 320, 200)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    ref = O.trace_forward(*args, rays, start)
    with O.scan_mode("filtered") as m:
        mirror = O.trace_forward(*args, rays, start)
        contested = m.contested
    assert contested >= 20, contested                       # observed: ~250 of 2.9e6 cell scans
    np.testing.assert_array_equal(mirror["rgba"].view(np.uint32), ref["rgba"].view(np.uint32))
    for mode in (0, 1, 2, 3, 4, 5):      # filtered: blocks / eager / persistent / cached instances; 3: every face divided
        pipe = _pipeline(d)
        pipe.forward_mode = mode
        got, _ = _run_forward(pipe, fm, rays, start)
        np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
        np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    assert float(ref["rgba"][..., 3].max()) > 0.9


def test_trail_capacity_follows_the_longest_ray(foam_factory):
    """Rays with more hops than the trail holds are re-walked by a second launch as long as its longest ray; the
    pipeline therefore sizes the next trail for the longest ray of the batch it just traced (read back asynchronously).
    The gradients do not depend on where the trail ends."""
    d = 1
    fm = foam_factory(6000, d, 81)
    cam, rays, start = H.camera_setup(fm, 64, 48)
    g = np.random.default_rng(3).normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    fwd = O.trace_forward(*args, rays, start)
    ref = O.trace_backward(*args, rays, start, fwd["rgba"], g)
    longest = int(fwd["num_intersections"].max())
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    pipe = _pipeline(d)
    pipe.trail_steps = 4
    for attempt in range(2):
        f = pipe.trace_forward(p, a, adj, off, r, s)
        assert pipe._trail["cap"] == (4 if attempt == 0 else pipe.trail_steps)
        out = pipe.trace_backward(p, a, adj, off, r, s, f["rgba"], torch.from_numpy(g).to(DEV))
        torch.cuda.synchronize()
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
            assert ok and rel < 1e-5, (attempt, key, rel, worst)
    assert longest - 1 <= pipe.trail_steps <= 2 * longest + 32   # hops = scans - 1 at most; grown with a margin
    pipe.trail_steps_limit = 8
    pipe.trail_steps = 4
    pipe.trace_forward(p, a, adj, off, r, s)
    torch.cuda.synchronize()
    pipe.trace_forward(p, a, adj, off, r, s)
    assert pipe.trail_steps == 8                                    # never beyond the limit


def test_shuffled_batch_is_traced_in_a_coherent_order(foam_factory):
    """Flat batches above Pipeline.reorder_min_rays are traced in the order rf_build_ray_order computes;
    every output stays indexed by the caller's ray index and nothing else changes."""
    d = 1
    fm = foam_factory(6000, d, 77)
    rays, starts = H.random_rays(fm, 30_000, seed=5)
    rng = np.random.default_rng(9)
    q = np.sort(rng.uniform(0.05, 0.95, size=(30_000, 2)).astype(np.float32), axis=1)[:, ::-1].copy()
    g = rng.normal(size=(30_000, 4)).astype(np.float32)
    dg = rng.normal(size=(30_000, 2)).astype(np.float32)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: torch.from_numpy(x).to(DEV)
    tr, ts, tq = t(rays), t(starts), t(q)
    outs = []
    for reorder in (True, False):
        pipe = _pipeline(d)
        pipe.reorder_rays = reorder
        f = pipe.trace_forward(p, a, adj, off, tr, ts, depth_quantiles=tq, return_contribution=True)
        assert (pipe._order is not None) == reorder
        b = pipe.trace_backward(p, a, adj, off, tr, ts, f["rgba"], t(g), tq, f["depth_indices"], t(dg))
        outs.append((f, b))
    (f1, b1), (f0, b0) = outs
    for k in ("rgba", "depth", "depth_indices", "num_intersections"):
        assert torch.equal(f1[k], f0[k]), k
    ok, rel, worst = H.grad_close(f1["contribution"].cpu().numpy(), f0["contribution"].cpu().numpy())
    assert ok and rel < 1e-5, ("contribution", rel, worst)
    for k in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(b1[k].cpu().numpy(), b0[k].cpu().numpy())
        assert ok and rel < 1e-5, (k, rel, worst)
    # the order is a permutation, grouped by entry cell
    order = pipe_order = outs[0][0]  # noqa: F841  (kept for readability)
    pipe = _pipeline(d)
    pipe.trace_forward(p, a, adj, off, tr, ts)
    perm = pipe._order["order"].cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(perm), np.arange(30_000))
    s_sorted = starts[perm].astype(np.int64)
    assert (np.diff(s_sorted) >= 0).all()
    # and against the oracle, in the caller's order
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], rays, starts, depth_quantiles=q)
    np.testing.assert_array_equal(f1["rgba"].cpu().numpy().view(np.uint32), ref["rgba"].view(np.uint32))


def test_degenerate_image_shape_is_a_flat_batch(foam_factory):
    """[B, 1, 6] rays: batch shape is preserved in the outputs, the tile mapping is not used."""
    d = 2
    fm = foam_factory(5000, d, 31)
    rays, starts = H.random_rays(fm, 3000, seed=2)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV).reshape(3000, 1, 6)
    s = torch.from_numpy(starts).to(DEV).reshape(3000, 1)
    pipe = _pipeline(d)
    assert pipe._launch_opts(p, a, adj, off, r.shape).image_width == 0
    out = pipe.trace_forward(p, a, adj, off, r, s)
    assert out["rgba"].shape == (3000, 1, 4) and out["num_intersections"].shape == (3000, 1, 1)
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                          fm["point_adjacency_offsets"], rays, starts)
    np.testing.assert_array_equal(out["rgba"].cpu().numpy().reshape(3000, 4).view(np.uint32), ref["rgba"].view(np.uint32))
    g = np.random.default_rng(1).normal(size=(3000, 4)).astype(np.float32)
    b = pipe.trace_backward(p, a, adj, off, r, s, out["rgba"], torch.from_numpy(g).to(DEV).reshape(3000, 1, 4))
    refb = O.trace_backward(d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"],
                            rays, starts, ref["rgba"], g)
    for k in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(b[k].cpu().numpy(), refb[k])
        assert ok and rel < 1e-5, (k, rel, worst)


def test_ray_order_handles_degenerate_directions(foam_factory):
    """rf_build_ray_order: zero / non-finite directions get a key like any other ray, the result is a
    permutation, and tracing such a batch matches the unordered run bit for bit."""
    d = 0
    fm = foam_factory(5000, d, 33)
    rays, starts = H.random_rays(fm, 20_000, seed=4)
    rays[::97, 3:] = 0.0                      # no direction
    rays[5::101, 3] = np.inf                  # non-finite
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r, s = torch.from_numpy(rays).to(DEV), torch.from_numpy(starts).to(DEV)
    outs = []
    for reorder in (True, False):
        pipe = _pipeline(d)
        pipe.reorder_rays = reorder
        pipe.reorder_min_rays = 1
        outs.append(pipe.trace_forward(p, a, adj, off, r, s))
        if reorder:
            perm = pipe._order["order"].cpu().numpy().astype(np.int64)
            assert np.array_equal(np.sort(perm), np.arange(20_000))
    good = torch.isfinite(outs[1]["rgba"]).all(dim=-1)
    assert torch.equal(outs[0]["rgba"][good], outs[1]["rgba"][good])
    assert torch.equal(outs[0]["num_intersections"], outs[1]["num_intersections"])


@pytest.mark.parametrize("d,image,quantiles", [(0, True, False), (2, True, True), (3, False, True), (1, False, False)])
def test_dividing_scan_instance(foam_factory, d, image, quantiles):
    """rf_launch_opts.forward_mode = 3 (Pipeline.strict_reference_scan): every face of every cell divided, the way the
    reference writes its scan (tracing_utils.cuh:43-67) -- in trace_forward, trace_backward (replay of a trail recorded
    under it, short trail + re-walk, no trail) and trace_benchmark: BIT-IDENTICAL to the oracle, like the default
    (filtered) instances, and a trail recorded under either mode replays under the other."""
    fm, rays, starts, q, dg, g, err, fwd, ref = _backward_case(foam_factory, d, 60 + d, image, quantiles, False,
                                                               n_points=7000)
    diff = O.build_adjacent_diff(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    cam, _, cstart = H.camera_setup(fm, 100, 60)
    bench_ref = O.trace_benchmark(d, fm["points"], fm["attributes"], fm["point_adjacency"],
                                  fm["point_adjacency_offsets"], diff, cam, cstart, weight_threshold=0.05)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: None if x is None else torch.from_numpy(x).to(DEV)
    tr, ts, tq = t(rays), t(starts), t(q)
    for trail in ("replay", "short", "rewalk"):
        pipe = _pipeline(d)
        pipe.strict_reference_scan = True
        assert pipe.forward_mode == 3
        if trail == "rewalk":
            pipe.record_trail = False
        elif trail == "short":
            pipe.trail_steps = 5
        f = pipe.trace_forward(p, a, adj, off, tr, ts, depth_quantiles=tq)
        np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
        np.testing.assert_array_equal(f["num_intersections"].cpu().numpy().view(np.uint32), fwd["num_intersections"])
        if quantiles:
            np.testing.assert_array_equal(f["depth"].cpu().numpy().view(np.uint32), fwd["depth"].view(np.uint32))
            np.testing.assert_array_equal(f["depth_indices"].cpu().numpy().view(np.uint32), fwd["depth_indices"])
        out = pipe.trace_backward(p, a, adj, off, tr, ts, f["rgba"], t(g), tq, f.get("depth_indices"), t(dg))
        torch.cuda.synchronize()
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
            assert ok and rel < 1e-5, (trail, key, rel, worst)
    # one function, two evaluations: the trail of a default forward is replayed by a backward in mode 3
    pipe = _pipeline(d)
    f = pipe.trace_forward(p, a, adj, off, tr, ts, depth_quantiles=tq)
    np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
    pipe.strict_reference_scan = True
    out = pipe.trace_backward(p, a, adj, off, tr, ts, t(fwd["rgba"]), t(g), tq, t(fwd.get("depth_indices")), t(dg))
    assert pipe.last_backward_replayed
    for key in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
        assert ok and rel < 1e-5, ("default trail, mode-3 backward", key, rel, worst)
    # the render path
    pipe = _pipeline(d)
    pipe.strict_reference_scan = True
    dtab = pipe.build_adjacent_diff(p, adj, off)
    img = torch.zeros((60, 100), dtype=torch.uint32, device=DEV)
    camera = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    sp = torch.tensor([int(cstart)], dtype=torch.int64).to(torch.uint32).to(DEV)
    pipe.trace_benchmark(p, a, adj, off, dtab, camera, sp, img, weight_threshold=0.05)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(img.cpu().numpy().view(np.uint32), bench_ref)


@pytest.mark.parametrize("d,image", [(1, True), (2, False), (3, False), (3, True)])
def test_gradient_row_pitch_does_not_change_the_gradients(foam_factory, d, image):
    """Pipeline.gradient_row_pitch / rf_launch_opts.attr_grad_pitch: "auto" accumulates attr_grad in rows on 64-byte lines
    (16 / 32 / 64 floats for A = 13 / 28 / 49) and returns a [N, A] VIEW of them, "dense" is the reference's contiguous
    [N, A], an int any pitch >= A -- same gradients (vs the oracle) whichever; the padding columns stay zero; a pitch below
    A is refused."""
    fm, rays, starts, q, dg, g, err, fwd, ref = _backward_case(foam_factory, d, 70 + d, image, False, False)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: None if x is None else torch.from_numpy(x).to(DEV)
    A = fm["attributes"].shape[1]
    for pitch in ("auto", "dense", A + 3):
        pipe = _pipeline(d)
        pipe.gradient_row_pitch = pitch
        f = pipe.trace_forward(p, a, adj, off, t(rays), t(starts))
        out = pipe.trace_backward(p, a, adj, off, t(rays), t(starts), f["rgba"], t(g))
        torch.cuda.synchronize()
        ag = out["attr_grad"]
        assert ag.shape == (fm["points"].shape[0], A)
        want_pitch = {"auto": {4: 4, 13: 16, 28: 32, 49: 64}[A], "dense": A}.get(pitch, pitch)
        assert ag.stride() == (want_pitch, 1) and ag.is_contiguous() == (want_pitch == A)
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
            assert ok and rel < 1e-5, (pitch, key, rel, worst)
        if want_pitch != A:      # what lies between the rows was never written
            n = fm["points"].shape[0]
            rows = out["flat_grad"][-n * want_pitch:].view(n, want_pitch)
            assert rows.data_ptr() == ag.data_ptr() and float(rows[:, A:].abs().max()) == 0.0
            assert ag.data_ptr() % 64 == 0
    bad = _pipeline(d)
    bad.gradient_row_pitch = A - 1
    with pytest.raises(RuntimeError, match="gradient_row_pitch"):
        bad.trace_backward(p, a, adj, off, t(rays), t(starts), t(fwd["rgba"]), t(g))


@pytest.mark.parametrize("forward_mode", [4, 5])
@pytest.mark.parametrize("d,image", [(2, True), (3, False), (1, False)])
def test_persistent_forward_feeds_the_replay(foam_factory, d, image, forward_mode):
    """forward_mode 4 (persistent waves refilled from a queue by ballot + prefix count): rgba / num_intersections /
    contribution as the oracle's, and the trail it records -- slots are those of the ordinary launch, whichever lane walked
    the ray -- replayed by trace_backward (both backward paths) gives the oracle's gradients.  forward_mode 5 (cell
    records and face blocks served from a block-level LDS table, filled and read without locks by the block's four
    waves): the same, on a sorted batch whose 256-slot groups do re-visit cells."""
    fm, rays, starts, q, dg, g, err, fwd, ref = _backward_case(foam_factory, d, 80 + d, image, False, False,
                                                               n_points=7000)
    if not image:      # large enough for the sorted order (Pipeline.reorder_min_rays)
        rays, starts = H.random_rays(fm, 20_000, seed=91)
        rng = np.random.default_rng(5)
        g = rng.normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
        args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
        fwd = O.trace_forward(*args, rays, starts, return_contribution=True)
        ref = O.trace_backward(*args, rays, starts, fwd["rgba"], g, num_threads=1)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    t = lambda x: None if x is None else torch.from_numpy(x).to(DEV)
    pipe = _pipeline(d)
    pipe.forward_mode = forward_mode
    f = pipe.trace_forward(p, a, adj, off, t(rays), t(starts), return_contribution=not image)
    np.testing.assert_array_equal(f["rgba"].cpu().numpy().view(np.uint32), fwd["rgba"].view(np.uint32))
    np.testing.assert_array_equal(f["num_intersections"].cpu().numpy().view(np.uint32), fwd["num_intersections"])
    if not image:
        np.testing.assert_allclose(f["contribution"].cpu().numpy(), fwd["contribution"], rtol=1e-4, atol=1e-6)
    assert pipe._trail is not None
    out = pipe.trace_backward(p, a, adj, off, t(rays), t(starts), f["rgba"], t(g))
    torch.cuda.synchronize()
    for key in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(out[key].cpu().numpy(), ref[key])
        assert ok and rel < 1e-5, (key, rel, worst)


def test_backward_of_a_strided_ray_view_replays_the_trail(foam_factory):
    """collect_error_map (scene.py:518-537) traces a strided view of a frame's rays through TraceRays with a ray_error box:
    forward and backward each copy the view, and the backward must still find the ray order and the hop trail of its
    forward (keys on the caller's view) -- same gradients and point_error as for a contiguous copy of the same rays."""
    d = 2
    fm = foam_factory(7000, d, 61)
    cam, rays, start = H.camera_setup(fm, 192, 128)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    full = torch.from_numpy(rays).to(DEV)[None]                      # [1, H, W, 6]
    view = full[:, 1::2, 0::2, :]
    assert not view.is_contiguous()
    st = torch.full(view.shape[:-1], int(start.reshape(-1)[0]), dtype=torch.int64, device=DEV).to(torch.uint32)
    rng = np.random.default_rng(3)
    g = torch.from_numpy(rng.normal(size=tuple(view.shape[:-1]) + (4,)).astype(np.float32)).to(DEV)
    err = torch.from_numpy(rng.uniform(size=tuple(view.shape[:-1]) + (1,)).astype(np.float32)).to(DEV)
    outs = []
    for r in (view, view.contiguous()):
        pipe = _pipeline(d)
        pipe.record_trail = True
        pipe.reorder_min_rays = 1024
        f = pipe.trace_forward(p, a, adj, off, r, st)
        b = pipe.trace_backward(p, a, adj, off, r, st, f["rgba"], g, ray_error=err)
        assert pipe.last_backward_replayed is True
        outs.append((f, b))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0]["rgba"], outs[1][0]["rgba"])
    for key in ("points_grad", "attr_grad", "point_error"):
        ok, rel, worst = H.grad_close(outs[0][1][key].cpu().numpy(), outs[1][1][key].cpu().numpy())
        assert ok and rel < 1e-5, (key, rel, worst)


def test_writes_autograd_cannot_see_need_cache_foam_off_or_invalidate(foam_factory):
    """the Pipeline keeps the packed foam between calls, keyed on tensor identity + `_version`
    -- a deliberate deviation from the reference's stateless binding.  A write behind autograd's back (`param.data.add_`)
    changes neither, so a caching pipeline goes on tracing the OLD foam until `invalidate()`; with `cache_foam = False`
    (the reference's behaviour: repack on every call) every call sees the memory as it is.  Forward outputs and gradients
    against the oracle on the moved points, both ways."""
    d = 1
    fm = foam_factory(5000, d, 31)
    cam, rays, start = H.camera_setup(fm, 64, 48)
    g = np.random.default_rng(2).normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    shift = (np.random.default_rng(3).normal(size=fm["points"].shape) * 2e-4).astype(np.float32)   # the lists stay Delaunay-valid enough to walk
    moved = dict(fm)
    moved["points"] = fm["points"] + shift
    args0 = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    args1 = (d, moved["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    ref0 = O.trace_forward(*args0, rays, start)
    ref1 = O.trace_forward(*args1, rays, start)
    bwd1 = O.trace_backward(*args1, rays, start, ref1["rgba"], g)
    assert not np.array_equal(ref0["rgba"], ref1["rgba"])
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    r = torch.from_numpy(rays).to(DEV)
    s = torch.full(r.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32).to(DEV)
    same = lambda out, ref: np.array_equal(out["rgba"].cpu().numpy().view(np.uint32), ref["rgba"].view(np.uint32))

    def check_moved(pipe):
        f = pipe.trace_forward(p, a, adj, off, r, s)
        assert same(f, ref1)
        out = pipe.trace_backward(p, a, adj, off, r, s, f["rgba"], torch.from_numpy(g).to(DEV))
        for key in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(out[key].cpu().numpy(), bwd1[key])
            assert ok and rel < 1e-5, (key, rel, worst)

    # 1. the reference's behaviour: nothing cached
    pipe = _pipeline(d)
    pipe.cache_foam = False
    assert same(pipe.trace_forward(p, a, adj, off, r, s), ref0)
    version = p._version
    p.data.add_(torch.from_numpy(shift).to(DEV))
    assert p._version == version                      # the write is invisible to the version counter
    check_moved(pipe)
    # 2. the default: the stale foam until invalidate()
    p.data.sub_(torch.from_numpy(shift).to(DEV))
    p.data.copy_(torch.from_numpy(fm["points"]).to(DEV))
    pipe = _pipeline(d)
    assert pipe.cache_foam
    assert same(pipe.trace_forward(p, a, adj, off, r, s), ref0)
    p.data.copy_(torch.from_numpy(moved["points"]).to(DEV))
    assert same(pipe.trace_forward(p, a, adj, off, r, s), ref0)     # documented: the packed copy of the old points
    pipe.invalidate()
    check_moved(pipe)


@pytest.mark.parametrize("forward_mode", [1, 2, 3, 5])
def test_subnormal_exit_distances_and_underflowing_foams(forward_mode):
    """the certificate's proof assumes normal quotients; the kernels do not leave that to the caller
    (scan_end's fail-safe: a winning quotient with a zero exponent field goes to the dividing scan).  Rays that start within
    a few subnormals of a bisector -- where, without the fail-safe, the filtered evaluation picks other exits than the
    reference (tests/test_oracle.py shows it on the CPU mirror) -- and whole foams scaled by 2^-12 / 2^-17 / 2^-60 (fp16
    offsets normal / subnormal with heavy ties / all underflown to zero): every scheduling mode equals the oracle's LITERAL
    reference scan bit for bit."""
    from radfoam_amd import foam
    from tests.test_oracle import _bisectors_through_the_origin

    pts, adj, off = _bisectors_through_the_origin()
    att = np.zeros((8, 4), dtype=np.float32)
    att[:, :3] = 0.3
    att[:, 3] = 2.0
    rng = np.random.default_rng(3)
    n = 20480                      # > reorder_min_rays: the flat batch takes the sorted order and (mode 5) the cell table
    rays = np.zeros((n, 6), dtype=np.float32)
    rays[:, 0] = -(rng.integers(1, 2000, n).astype(np.uint32).view(np.float32))
    rays[: n // 2, 0] = -(rng.uniform(1.0, 2.0, n // 2) * 2.0 ** rng.integers(-140, -120, n // 2)).astype(np.float32)
    rays[:, 1:3] = rng.uniform(-0.05, 0.05, (n, 2)).astype(np.float32)
    d = np.stack([np.ones(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], axis=1)
    rays[:, 3:] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    start = np.zeros(n, dtype=np.uint32)
    fm = {"points": pts, "attributes": att, "point_adjacency": adj, "point_adjacency_offsets": off}
    want = O.trace_forward(0, pts, att, adj, off, rays, start)
    pipe = _pipeline(0)
    pipe.forward_mode = forward_mode
    got, _ = _run_forward(pipe, fm, rays, start)
    np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), want["rgba"].view(np.uint32))
    np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32).reshape(-1), want["num_intersections"].reshape(-1))
    assert float(want["rgba"][:, 3].max()) > 0

    base = foam.make_synthetic_foam(3000, 1, 2)
    cam = foam.default_camera(96, 64)
    r0 = foam.camera_rays(cam)
    pipe = _pipeline(1)
    pipe.forward_mode = forward_mode
    for e in (-12, -17, -60):
        s = np.float32(2.0 ** e)
        fs = dict(base)
        fs["points"] = base["points"] * s
        a = base["attributes"].copy()
        a[:, -1] = np.minimum(a[:, -1] / s, np.float32(3e38))
        fs["attributes"] = a
        r = r0.copy()
        r[..., :3] *= s
        st = np.uint32(foam.nearest_point(base["points"], cam["position"]))
        want = O.trace_forward(1, fs["points"], a, fs["point_adjacency"], fs["point_adjacency_offsets"], r, st)
        got, _ = _run_forward(pipe, fs, r, st)
        np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), want["rgba"].view(np.uint32), err_msg=str(e))
        np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), want["num_intersections"], err_msg=str(e))


def test_tile_prior_ranks_tiles_and_does_not_change_results(foam_factory):
    """rays the pipeline has not traced before take their block order from a cost prior (a coarse grid of
    the foam marched by five rays per tile: rf_build_cost_grid / rf_estimate_tile_cost) instead of the static dealing.  The
    prior only orders blocks: forward outputs with and without it are the same bits (and the oracle's), gradients within the
    usual bars; its estimate ranks the tiles as a trace of the same rays does (Spearman > 0.6 on this frame) -- through
    trace_forward's ray tensor and through trace_benchmark's camera alike."""
    import radfoam

    d = 2
    fm = foam_factory(20000, d, 17)
    cam, rays, start = H.camera_setup(fm, 320, 208, position=(0.4, 0.3, -2.6))
    ref = O.trace_forward(d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"], rays, start)
    outs = {}
    for prior in (True, False):
        pipe = _pipeline(d)
        pipe.tile_prior = prior
        got, (p, a, adj, off, r, s) = _run_forward(pipe, fm, rays, start)
        assert (pipe._prior_keep is not None) == prior          # new rays: the launch did / did not get a prior order
        g = torch.randn(r.shape[:-1] + (4,), generator=torch.Generator().manual_seed(2)).to(DEV)
        bwd = pipe.trace_backward(p, a, adj, off, r, s, got["rgba"].to(DEV), g)
        outs[prior] = (got, {k: bwd[k].cpu() for k in ("points_grad", "attr_grad")})
        np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
        np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
        if prior:
            order = pipe._prior_keep.cpu().numpy()
            tiles = ((208 + 15) // 16) * ((320 + 15) // 16)
            assert sorted(order[order < tiles].tolist()) == list(range(tiles))      # every tile exactly once
            est = pipe.estimate_tile_cost((p, a, adj, off), 208, 320, rays=r).float().cpu()
            ni = got["num_intersections"].reshape(208, 320).float()
            measured = ni.view(13, 16, 20, 16).amax(dim=(1, 3)).reshape(-1)
            rank = lambda x: torch.argsort(torch.argsort(x)).double()
            rx, ry = rank(est) - rank(est).mean(), rank(measured) - rank(measured).mean()
            assert float((rx * ry).sum() / (rx.norm() * ry.norm())) > 0.6
    for k in ("points_grad", "attr_grad"):
        ok, rel, worst = H.grad_close(outs[True][1][k].numpy(), outs[False][1][k].numpy())
        assert ok and rel < 1e-5, (k, rel, worst)
    # the render path: the camera instead of a ray tensor
    a16 = torch.from_numpy(fm["attributes"].astype(np.float16)).to(DEV)
    words = {}
    for prior in (True, False):
        rend = radfoam.create_pipeline(d, torch.float16)
        rend.tile_prior = prior
        p, _, adj, off = H.to_torch_foam(fm, DEV)
        diff = rend.build_adjacent_diff(p, adj, off)
        out8 = torch.zeros((208, 320), dtype=torch.uint32, device=DEV)
        camt = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
        rend.trace_benchmark(p, a16, adj, off, diff, camt, torch.tensor([int(start)], dtype=torch.int64).to(torch.uint32).to(DEV),
                             out8, weight_threshold=0.05)
        words[prior] = out8.cpu().numpy().view(np.uint32).copy()
        assert (rend._prior_keep is not None) == prior
    np.testing.assert_array_equal(words[True], words[False])
    assert words[True].any()


def test_device_built_tile_orders_equal_the_host_rules():
    """rf_build_tile_orders (one launch, per-XCD bitonic sort in LDS) against radfoam_amd.pipeline.tile_order (the torch
    restatement rounds 4-5 ran after every learning forward): the same table entry for entry, for the rules "auto" uses and
    their parameters, on frames whose last round of blocks is padding and on a flat batch -- ties, a cost map with many
    equal entries and tiles beyond the histogram's last bin included."""
    import ctypes as C
    from radfoam_amd import _lib
    from radfoam_amd.pipeline import Pipeline, tile_order

    lib = _lib.load()
    pipe = _pipeline(0)
    gen = torch.Generator().manual_seed(4)
    for (h, w) in ((1080, 1920), (136, 200), (2160, 3840), ("flat", 1_000_000)):
        default = pipe._default_tiles(h, w, torch.device(DEV))
        nt = (w + 255) // 256 if h == "flat" else ((h + 15) // 16) * ((w + 15) // 16)
        for kind in ("spread", "ties", "huge"):
            cost = torch.randint(1, 300, (nt,), generator=gen, dtype=torch.int32)
            if kind == "ties":
                cost = (cost // 40) * 40
            if kind == "huge":
                cost[::7] += 6000
            cost = cost.to(DEV)
            for rules in (("xcd", "tail"), ("xcd:8", "tail:7"), ("tail:%d" % max(8, nt // 8),) * 2):
                a, b = pipe._build_orders(cost, default, h, w, rules)
                for got, rule in ((a, rules[0]), (b, rules[1])):
                    want = tile_order(cost, default, rule)
                    # tiles past the end keep SOME value >= nt in both (which one is immaterial: the block owns no rays)
                    g64, w64 = got.to(torch.int64).clamp(max=nt), want.clamp(max=nt)
                    assert torch.equal(g64, w64), (h, w, kind, rule)
    with pytest.raises(RuntimeError, match="rule must be"):
        out = torch.empty(64, dtype=torch.int32, device=DEV)
        _lib.check(lib.rf_build_tile_orders(C.c_void_p(out.data_ptr()), 256 * 64, 0, 0, 9, 1, C.c_void_p(out.data_ptr()), 0, 0,
                                            None, None))


def test_a_camera_path_borrows_the_previous_frames_order_and_another_camera_does_not(foam_factory):
    """Pipeline.tile_order_coherence_degrees: the next frame of a camera path (0.05 degrees on) takes
    the forward order the previous frame's trace measured -- decided on the device by rf_gate_tile_order from five sample
    rays, no synchronisation --, a camera 40 degrees away fails the test and runs under the static dealing; either way the
    outputs are the oracle's, bit for bit.  Same through trace_benchmark's camera."""
    import bench
    import radfoam
    from radfoam_amd import foam

    d = 1
    fm = foam_factory(20000, d, 19)
    p, a, adj, off = H.to_torch_foam(fm, DEV)
    pipe = _pipeline(d)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    W_, H_ = 320, 208

    def view(k):
        cam = bench.view_camera(W_, H_, 0, k)
        rays = foam.camera_rays(cam)
        return cam, rays, np.uint32(foam.nearest_point(fm["points"], cam["position"]))

    verdicts = []
    for k in (0, 1, 2, 800):                # 0.05 degrees per view; view 800 is 40 degrees on
        cam, rays, start = view(k)
        ref = O.trace_forward(*args, rays, start)
        got, _ = _run_forward(pipe, fm, rays, start)
        np.testing.assert_array_equal(got["rgba"].numpy().view(np.uint32), ref["rgba"].view(np.uint32))
        np.testing.assert_array_equal(got["num_intersections"].numpy().view(np.uint32), ref["num_intersections"])
        t = pipe._tiles
        if k:
            order = pipe._prior_keep
            assert order is not None
            coherent = int(t["verdict"].item())
            verdicts.append(coherent)
            default = pipe._default_tiles(H_, W_, torch.device(DEV)).to(torch.int32)
            assert torch.equal(order, prev_forward if coherent else default)
        prev_forward = t["forward"].clone()
    assert verdicts == [1, 1, 0]
    pipe.tile_order_coherence_degrees = 0.0     # off: new rays run under the static dealing, no gate
    cam, rays, start = view(3)
    _run_forward(pipe, fm, rays, start)
    assert pipe._prior_keep is None
    # the render path: the camera instead of a ray tensor (learns every tile_order_refresh_render launches)
    rend = radfoam.create_pipeline(d, torch.float16)
    rend.tile_order_refresh_render = 1
    a16 = a.to(torch.float16)
    diff = rend.build_adjacent_diff(p, adj, off)
    hdiff = O.build_adjacent_diff(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    verdicts = []
    for k in (0, 1, 800):
        cam, rays, start = view(k)
        out8 = torch.zeros((H_, W_), dtype=torch.uint32, device=DEV)
        camt = {kk: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for kk, v in cam.items()}
        rend.trace_benchmark(p, a16, adj, off, diff, camt, torch.tensor([int(start)], dtype=torch.int64).to(torch.uint32).to(DEV),
                             out8, weight_threshold=0.05)
        want = O.trace_benchmark(d, fm["points"], fm["attributes"].astype(np.float16), fm["point_adjacency"],
                                 fm["point_adjacency_offsets"], hdiff, cam, start, weight_threshold=0.05)
        np.testing.assert_array_equal(out8.cpu().numpy().view(np.uint32), want)
        if k:
            verdicts.append(int(rend._tiles["verdict"].item()))
    assert verdicts == [1, 0]
