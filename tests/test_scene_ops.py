"""Scene-side operators: attribute packing, entry-cell lookup, farthest neighbour.

CPU part: the numpy checker (oracle/scene_ops_ref.py) is pinned to torch evaluating the reference's
own expression (scene.py:202-217).  GPU part: the HIP kernels, through the C-ABI, against the checker.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import scene_ops_ref as R
from radfoam_amd import foam


def _scene_params(n, degree, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    att_dc = rng.normal(0, 0.3, (n, 3)).astype(np.float32)
    att_sh = rng.normal(0, 0.3, (n, 3 * ((degree + 1) ** 2 - 1))).astype(np.float32)
    # raw densities covering all three softplus regimes: exp underflow, the knee, the linear branch
    density = (rng.normal(0, 1.0, (n, 1)) * spread).astype(np.float32)
    density[: min(n, 8), 0] = [-30.0, -9.0, -2.0, -0.1, 0.0, 0.3, 1.9999, 2.5][: min(n, 8)]
    return att_dc, att_sh, density


def _reference_expression(att_dc, att_sh, density, scale, dtype):
    """RadFoamScene.get_trace_data, scene.py:202-217, verbatim in torch."""
    primal_density = scale * F.softplus(density, beta=10)
    primal_attributes = torch.cat([att_dc, att_sh], dim=-1)
    return torch.cat([primal_attributes, primal_density], dim=-1).to(dtype)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_checker_matches_reference_expression(degree):
    dc, sh, dn = _scene_params(257, degree, 3 + degree, spread=1.5)
    scale = 1.7
    want = _reference_expression(torch.from_numpy(dc), torch.from_numpy(sh), torch.from_numpy(dn), scale,
                                 torch.float32).numpy()
    got = R.pack_attributes(dc, sh, dn, scale)
    assert got.shape == (257, 1 + 3 * (degree + 1) ** 2)
    np.testing.assert_array_equal(got[:, :-1], want[:, :-1])
    np.testing.assert_allclose(got[:, -1], want[:, -1], rtol=3e-7, atol=1e-30)
    # backward: autograd of the same expression
    t = [torch.from_numpy(a.copy()).requires_grad_(True) for a in (dc, sh, dn)]
    g = np.random.default_rng(9).normal(0, 1, want.shape).astype(np.float32)
    _reference_expression(t[0], t[1], t[2], scale, torch.float32).backward(torch.from_numpy(g))
    d_dc, d_sh, d_dn = R.pack_attributes_backward(dn, scale, g)
    np.testing.assert_array_equal(d_dc, t[0].grad.numpy())
    np.testing.assert_array_equal(d_sh, t[1].grad.numpy())
    np.testing.assert_allclose(d_dn, t[2].grad.numpy(), rtol=3e-6, atol=1e-30)


def test_checker_nearest_and_farthest_small():
    fm = foam.make_synthetic_foam(600, 0, 21)
    p = fm["points"]
    q = np.array([[0.1, -0.2, 0.3], [5.0, 5.0, 5.0], p[17]], dtype=np.float32)
    idx = R.nearest_point(p, q)
    d = ((p.astype(np.float64)[None] - q.astype(np.float64)[:, None]) ** 2).sum(-1)
    np.testing.assert_array_equal(idx, d.argmin(1).astype(np.uint32))
    assert idx[2] == 17
    far, radius = R.farthest_neighbor(p, fm["point_adjacency"], fm["point_adjacency_offsets"])
    from radfoam_amd import shims
    far_t, radius_t = shims.farthest_neighbor(torch.from_numpy(p), torch.from_numpy(fm["point_adjacency"]),
                                              torch.from_numpy(fm["point_adjacency_offsets"]))
    np.testing.assert_array_equal(far, far_t.numpy())
    np.testing.assert_allclose(radius, radius_t.numpy(), rtol=2e-6)


def test_checker_adjacency_from_tets_matches_qhull_neighbours():
    from scipy.spatial import Delaunay
    pts = np.random.default_rng(4).uniform(-1, 1, (500, 3))
    tri = Delaunay(pts)
    adj, off = R.adjacency_from_tets(tri.simplices, 500)
    indptr, indices = tri.vertex_neighbor_vertices
    np.testing.assert_array_equal(off, indptr.astype(np.uint32))
    for i in range(500):
        np.testing.assert_array_equal(adj[off[i]:off[i + 1]], np.sort(indices[indptr[i]:indptr[i + 1]]))
    # the same CSR the synthetic foams (and so every tracer test) are built with
    fm = foam.make_synthetic_foam(700, 0, 5)
    assert (np.diff(fm["point_adjacency_offsets"].astype(np.int64)) > 0).all()


# ------------------------------------------------------------------------------------------------ GPU

def _cuda(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_pack_attributes_forward(degree, dtype):
    import radfoam
    dc, sh, dn = _scene_params(10_007, degree, 40 + degree, spread=1.5)
    scale = 0.83
    out = radfoam.pack_attributes(_cuda(dc), _cuda(sh), _cuda(dn), scale, dtype)
    assert out.dtype == dtype and out.shape == (10_007, 1 + 3 * (degree + 1) ** 2)
    want = R.pack_attributes(dc, sh, dn, scale)
    got = out.float().cpu().numpy()
    if dtype == torch.float32:
        np.testing.assert_array_equal(got[:, :-1], want[:, :-1])               # pure copies
        np.testing.assert_allclose(got[:, -1], want[:, -1], rtol=4e-7, atol=1e-30)   # <= 2 ulp softplus
    else:
        want16 = want.astype(np.float16).astype(np.float32)
        np.testing.assert_array_equal(got[:, :-1], want16[:, :-1])
        np.testing.assert_allclose(got[:, -1], want16[:, -1], rtol=1e-3)       # one half ulp at rounding ties
    # and against torch evaluating the reference expression on the device
    ref = _reference_expression(_cuda(dc), _cuda(sh), _cuda(dn), scale, dtype).float().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-3 if dtype == torch.float16 else 1e-6, atol=1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("degree", [0, 2, 3])
def test_pack_attributes_backward(degree):
    import radfoam
    dc, sh, dn = _scene_params(5_003, degree, 50 + degree, spread=1.5)
    scale = 1.3
    t = [_cuda(a).requires_grad_(True) for a in (dc, sh, dn)]
    out = radfoam.pack_attributes(t[0], t[1], t[2], scale, torch.float32)
    g = np.random.default_rng(5).normal(0, 1, tuple(out.shape)).astype(np.float32)
    out.backward(_cuda(g))
    d_dc, d_sh, d_dn = R.pack_attributes_backward(dn, scale, g)
    np.testing.assert_array_equal(t[0].grad.cpu().numpy(), d_dc)
    np.testing.assert_array_equal(t[1].grad.cpu().numpy(), d_sh)
    np.testing.assert_allclose(t[2].grad.cpu().numpy(), d_dn, rtol=3e-6, atol=1e-30)


@pytest.mark.gpu
def test_pack_attributes_feeds_the_tracer_and_backpropagates():
    """get_trace_data -> TraceRays -> loss.backward through the fused packing, against the torch expression."""
    import radfoam
    from radfoam_amd.render import TraceRays
    fm = foam.make_synthetic_foam(3000, 1, 31)
    n = fm["points"].shape[0]
    dc, sh, dn = _scene_params(n, 1, 7, spread=0.4)
    dn += 0.2
    cam = foam.default_camera(40, 30)
    rays = _cuda(foam.camera_rays(cam))
    start = torch.full(rays.shape[:-1], foam.nearest_point(fm["points"], cam["position"]),
                       dtype=torch.int64).to(torch.uint32).cuda()
    pts, adj, off = _cuda(fm["points"]), _cuda(fm["point_adjacency"]), _cuda(fm["point_adjacency_offsets"])
    pipe = radfoam.create_pipeline(1, torch.float32)
    grads = []
    for fused in (True, False):
        t = [_cuda(a).requires_grad_(True) for a in (dc, sh, dn)]
        attrs = radfoam.pack_attributes(t[0], t[1], t[2], 2.0) if fused else \
            _reference_expression(t[0], t[1], t[2], 2.0, torch.float32)
        rgba, *_ = TraceRays.apply(pipe, pts, attrs, adj, off, rays, start, None, False)
        rgba[..., :3].square().sum().backward()
        grads.append([x.grad.cpu().numpy() for x in t])
    for a, b in zip(*grads):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-6)


@pytest.mark.gpu
def test_pack_attributes_argument_errors():
    import radfoam
    dc, sh, dn = _scene_params(16, 2, 1)
    with pytest.raises(RuntimeError):
        radfoam.pack_attributes(torch.from_numpy(dc), _cuda(sh), _cuda(dn))          # CPU tensor
    with pytest.raises(RuntimeError):
        radfoam.pack_attributes(_cuda(dc), _cuda(sh[:, :5].copy()), _cuda(dn))       # not an SH width
    with pytest.raises(RuntimeError):
        radfoam.pack_attributes(_cuda(dc), _cuda(sh), _cuda(dn), 1.0, torch.bfloat16)
    empty = radfoam.pack_attributes(_cuda(dc[:0]), _cuda(sh[:0]), _cuda(dn[:0]))
    assert empty.shape == (0, 28)


@pytest.mark.gpu
def test_nearest_point_matches_checker():
    import radfoam
    fm = foam.make_synthetic_foam(200_000, 0, 11, cache_dir=foam.default_cache_dir())
    p = fm["points"]
    rng = np.random.default_rng(2)
    q = np.concatenate([rng.uniform(-1, 1, (13, 3)), [[0.0, 0.0, -3.0], [40.0, -7.0, 2.0]], p[[5, 199_999]]]
                       ).astype(np.float32)
    got = radfoam.nn(_cuda(p), radfoam.build_aabb_tree(_cuda(p)), _cuda(q))
    assert got.dtype == torch.uint32 and got.shape == (17,)
    want = R.nearest_point(p, q)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert got[-2] == 5 and got[-1] == 199_999
    # batch shape is preserved, an empty batch is fine
    assert radfoam.nn(_cuda(p), None, _cuda(q[:6].reshape(2, 3, 3))).shape == (2, 3)
    assert radfoam.nn(_cuda(p), None, _cuda(q[:0])).shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 33, 1000, 200_000])
def test_nearest_point_through_the_tree_equals_the_brute_force_search(n):
    """radfoam.nn with many queries walks the caller's AABB tree (the reference's route, aabb_tree.cu:343-415), a lane per
    query; the answer is the brute-force kernel's, index for index -- queries inside the cloud, far outside it, on
    points, and duplicated points (lowest index wins)."""
    import radfoam
    from radfoam_amd import scene_ops
    rng = np.random.default_rng(40 + n)
    p = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    if n >= 33:
        p[7] = p[21]                                  # an exact tie
    p = np.ascontiguousarray(p[foam.kd_order(p)])     # the tree is over kd-ordered points
    q = np.concatenate([rng.uniform(-1, 1, (4000, 3)), rng.normal(0, 30, (500, 3)), p[rng.integers(0, n, 300)],
                        [[1e30, 0, 0], [0, -1e-30, 0]]]).astype(np.float32)
    pc, qc = _cuda(p), _cuda(q)
    tree = radfoam.build_aabb_tree(pc)
    brute = scene_ops.nearest_point(pc, qc)
    walked = scene_ops.nearest_point_tree(pc, tree, qc)
    assert walked.dtype == torch.uint32 and torch.equal(walked.view(torch.int32), brute.view(torch.int32))
    assert torch.equal(radfoam.nn(pc, tree, qc).view(torch.int32), brute.view(torch.int32))     # > 4096 queries: the tree
    assert torch.equal(radfoam.nn(pc, tree, qc[:5]).view(torch.int32), brute[:5].view(torch.int32))
    if n == 1000:
        want = R.nearest_point(p, q[:200])
        np.testing.assert_array_equal(walked[:200].cpu().numpy(), want)
    with pytest.raises(RuntimeError, match="aabb_tree must have shape"):
        scene_ops.nearest_point_tree(pc, tree[:1].repeat(3, 1, 1), qc)


@pytest.mark.gpu
def test_nearest_point_ties_take_the_lowest_index():
    import radfoam
    p = np.zeros((5000, 3), dtype=np.float32)
    p[:, 0] = np.arange(5000) % 7          # many exact duplicates
    got = radfoam.nn(_cuda(p), None, _cuda(np.array([[3.2, 0, 0], [100.0, 0, 0]], dtype=np.float32)))
    assert got.cpu().tolist() == [3, 6]


@pytest.mark.gpu
def test_farthest_neighbor_matches_checker():
    import radfoam
    fm = foam.make_synthetic_foam(4000, 0, 13)
    far, radius = radfoam.farthest_neighbor(_cuda(fm["points"]), _cuda(fm["point_adjacency"]),
                                            _cuda(fm["point_adjacency_offsets"]))
    want_far, want_radius = R.farthest_neighbor(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    np.testing.assert_array_equal(far.cpu().numpy(), want_far)
    np.testing.assert_allclose(radius.cpu().numpy(), want_radius, rtol=1e-6)


@pytest.mark.gpu
def test_adjacency_from_tets_matches_checker():
    from scipy.spatial import Delaunay
    from radfoam_amd import scene_ops
    pts = np.random.default_rng(8).uniform(-1, 1, (20_000, 3))
    tets = Delaunay(pts).simplices.astype(np.int32)
    adj, off = scene_ops.adjacency_from_tets(torch.from_numpy(tets).cuda(), 20_000)
    want_adj, want_off = R.adjacency_from_tets(tets, 20_000)
    assert adj.dtype == torch.uint32 and off.dtype == torch.uint32
    np.testing.assert_array_equal(off.cpu().numpy(), want_off)
    np.testing.assert_array_equal(adj.cpu().numpy(), want_adj)
    # junk tets are ignored, unreferenced points get empty ranges, no tets at all is fine
    bad = np.concatenate([tets[:100], [[0, 0, 1, 2], [5, 6, 7, 99_999_999 % 2**31]]]).astype(np.int32)
    adj2, off2 = scene_ops.adjacency_from_tets(torch.from_numpy(bad).cuda(), 20_010)
    w_adj2, w_off2 = R.adjacency_from_tets(bad, 20_010)
    np.testing.assert_array_equal(off2.cpu().numpy(), w_off2)
    np.testing.assert_array_equal(adj2.cpu().numpy(), w_adj2)
    adj3, off3 = scene_ops.adjacency_from_tets(torch.zeros((0, 4), dtype=torch.int32).cuda(), 7)
    assert adj3.numel() == 0 and off3.cpu().tolist() == [0] * 8


@pytest.mark.gpu
def test_triangulation_on_gpu_builds_the_same_csr():
    import radfoam
    pts = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (3000, 3)).astype(np.float32))
    cpu = radfoam.Triangulation(pts)
    gpu = radfoam.Triangulation(pts.cuda())
    assert gpu.point_adjacency().is_cuda
    np.testing.assert_array_equal(gpu.point_adjacency().cpu().numpy(), cpu.point_adjacency().numpy())
    np.testing.assert_array_equal(gpu.point_adjacency_offsets().cpu().numpy(), cpu.point_adjacency_offsets().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("shuffle", [True, False])
def test_batch_fetcher_on_the_device_follows_the_references_index_sequence(shuffle):
    """radfoam.BatchFetcher with a GPU present keeps the array in HBM and gathers a batch with one kernel
    (rf_fetch_batch); the rows it returns are those of the reference's index sequence (batch_fetcher.cpp:60-70,
    random.h:13-57 -- restated in numpy by the shim's own CPU path), for rays (24-byte rows), colours (12) and alphas
    (4) alike, so that the three fetchers of train.py stay aligned; the batch counter carries over the wrap of
    b * batch_size + j."""
    import radfoam
    from radfoam_amd import shims

    rng = np.random.default_rng(5)
    n, bs = 100_003, 4096
    for width in (6, 3, 1):
        data = torch.from_numpy(rng.normal(size=(n, width)).astype(np.float32))
        f = radfoam.BatchFetcher(data, bs, shuffle)
        assert f._native and f.data.is_cuda
        ref = shims.BatchFetcher.__new__(shims.BatchFetcher)      # the numpy index path of the same class
        ref.data, ref.batch_size, ref.shuffle, ref.batch_idx = data, bs, shuffle, 0
        ref.rank, ref.world_size, ref.local_batch_size = 0, 1, bs
        for b in range(3):
            want = data[torch.from_numpy(ref._indices())]
            ref.batch_idx += 1
            got = f.next()
            assert got.shape == (bs, width) and got.is_cuda
            assert torch.equal(got.cpu(), want), (width, b)
    # an image-shaped array with batch_size 1, as collect_error_map / test_render use it (scene.py:505-512)
    imgs = torch.from_numpy(rng.normal(size=(5, 8, 12, 6)).astype(np.float32))
    f = radfoam.BatchFetcher(imgs, 1, False)
    for b in range(7):
        assert torch.equal(f.next().cpu(), imgs[b % 5][None])
    # far into a run: batch index times batch size beyond 2^32
    f = radfoam.BatchFetcher(data, bs, shuffle)
    ref.data, ref.batch_idx = data, (1 << 32) // bs + 7
    f.batch_idx = ref.batch_idx
    assert torch.equal(f.next().cpu(), data[torch.from_numpy(ref._indices())])
    # data-parallel training: the ranks' shares (rf_fetch_batch_range), concatenated in rank order, are the batch
    for world in (2, 8):
        whole = radfoam.BatchFetcher(data, bs, shuffle)
        parts = [radfoam.BatchFetcher(data, bs, shuffle, rank=r, world_size=world) for r in range(world)]
        for b in range(2):
            got = torch.cat([p.next() for p in parts])
            assert got.shape[0] == bs and torch.equal(got, whole.next()), (world, b)
