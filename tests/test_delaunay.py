"""The GPU triangulation: rf_kd_order, rf_build_aabb_tree, rf_delaunay_adjacency and the
``radfoam.Triangulation`` protocol over them.

Oracle: Qhull (scipy.spatial.Delaunay).  The Delaunay triangulation of points in general position is unique, so
the neighbour lists must be EQUAL, entry for entry (lists ascending, as the reference's find_adjacency emits them,
src/delaunay/delaunay.cu:190-229).  The star algorithm (radfoam_amd/csrc/rf_star.hpp) is plain C++ that the HIP
kernels run one lane per point; tests/host_harness compiles the same header for the host, so its logic -- link
surgery, tree search, filtered and exact predicates -- is checked here without a GPU as well (``-m "not gpu"``),
and the predicates against Python rationals.
"""
from fractions import Fraction

import numpy as np
import pytest

from radfoam_amd import foam
from tests.host_harness import star_host as S


def _kd(points):
    pts = np.asarray(points, dtype=np.float32)
    return np.ascontiguousarray(pts[foam.kd_order(pts)])


def _clustered(rng, per=3000):
    return np.concatenate([rng.normal(0, s, size=(per, 3)) + rng.uniform(-1, 1, 3)
                           for s in (1.0, 0.1, 0.01, 0.001, 0.3)]).astype(np.float32)


# ---- host build of the star code --------------------------------------------------------------------------------------

def _det(m):
    if len(m) == 1:
        return m[0][0]
    return sum((-1) ** j * m[0][j] * _det([r[:j] + r[j + 1:] for r in m[1:]]) for j in range(len(m)))


def _sgn(x):
    return (x > 0) - (x < 0)


def test_predicates_against_rationals():
    """orient / insphere signs (fp64 filter -> 384-bit integers) equal the signs of the exact determinants,
    including exactly coplanar / cospherical inputs (sign 0) and a 2^40 spread of magnitudes."""
    lib = S.lib()
    rng = np.random.default_rng(0)
    sphere = [(0, 3, 4), (3, 4, 0), (4, 0, 3), (5, 0, 0), (0, -5, 0), (-3, 0, 4), (0, 0, -5), (4, 3, 0), (-4, -3, 0)]
    zeros = 0
    for trial in range(1500):
        kind = trial % 5
        p = rng.uniform(-1, 1, size=(5, 3)).astype(np.float32)
        if kind == 1:
            p = rng.integers(-3, 4, size=(5, 3)).astype(np.float32)
        elif kind == 2:
            p[:, 2] = (p[:, 0] * 0.5 + p[:, 1] * 0.25).astype(np.float32)
            p[4, 2] += np.float32(1e-7) * (trial % 3 - 1)
        elif kind == 3:
            p *= np.float32(2.0) ** rng.integers(-20, 20, size=(5, 3)).astype(np.float32)
        elif kind == 4:
            p = np.array([sphere[i] for i in rng.choice(len(sphere), 5, replace=False)], dtype=np.float32) + np.float32(0.5)
        f = [[Fraction(float(x)) for x in row] for row in p]
        a = [[f[v][k] - f[0][k] for k in range(3)] for v in range(1, 5)]
        o = _sgn(_det(a[:3]))
        i4 = _sgn(_det([r + [sum(x * x for x in r)] for r in a]))
        zeros += (o == 0) + (i4 == 0)
        flat = np.ascontiguousarray(p.reshape(-1))
        got = (lib.star_host_exact_orient(flat.ctypes.data), lib.star_host_orient_sign(flat.ctypes.data),
               lib.star_host_exact_insphere(flat.ctypes.data), lib.star_host_insphere_sign(flat.ctypes.data))
        assert got == (o, o, i4, i4), (kind, p)
    assert zeros > 100   # the degenerate cases were really exercised


@pytest.mark.parametrize("sweep", [False, True], ids=["queries", "sweep"])
@pytest.mark.parametrize("name", ["uniform", "clustered", "sheet", "offset"])
def test_host_stars_equal_qhull(name, sweep):
    """Both ways of certifying a star: a tree walk per triangle (what the kernels ship: RF_STAR_SWEEP=0) and the sweep --
    one range query per star that offers every point within reach of its balls to the link (rf_star.hpp: star_sweep;
    built, equal to Qhull, measured out on the GPU and compiled out there)."""
    rng = np.random.default_rng(3)
    if name == "uniform":
        pts = rng.uniform(-1, 1, size=(12000, 3))
    elif name == "clustered":
        pts = _clustered(rng, 2000)
    elif name == "sheet":   # a thin sheet in a sparse volume: long hull facets, stars with > 100 neighbours
        pts = np.concatenate([np.c_[rng.uniform(-1, 1, size=(4000, 2)), 1e-3 * rng.normal(size=4000)],
                              rng.uniform(-1, 1, size=(1000, 3))])
    else:                   # large coordinates, small spacing
        pts = rng.uniform(-1, 1, size=(5000, 3)) + 1000.0
    pts = _kd(pts)
    off0, adj0 = foam.delaunay_csr(pts)
    off, adj, info = S.delaunay(pts, sweep=sweep)
    assert info["bad"] == 0
    assert np.array_equal(off, off0) and np.array_equal(adj, adj0)


def test_host_sweep_certifies_with_a_fraction_of_the_tree_walks():
    """The sweep's point: the same lists from a fifth of the tree nodes (on the GPU the nodes are not what bounds the
    launch -- DESIGN.md section 4.6 -- which is why it is compiled out there)."""
    rng = np.random.default_rng(12)
    pts = _kd(rng.uniform(-1, 1, size=(20000, 3)))
    off0, adj0 = foam.delaunay_csr(pts)
    moved = (pts + rng.normal(0, 0.03 * (8.0 / 20000) ** (1 / 3), size=pts.shape)).astype(np.float32)
    off1, adj1 = foam.delaunay_csr(moved)
    nodes = {}
    for sweep in (False, True):
        off, adj, info = S.delaunay(moved, old=(off0, adj0), sweep=sweep)
        assert info["bad"] == 0
        assert np.array_equal(off, off1) and np.array_equal(adj, adj1)
        nodes[sweep] = info["visited"].mean()
    assert nodes[True] < 0.4 * nodes[False]


@pytest.mark.parametrize("n", [33, 63, 65, 641, 642, 643, 1025, 2050])
def test_host_short_last_block_is_seeded_from_the_last_64_points(n):
    """with n % 64 in 1..3 the last kd-block has fewer than three other points, the first-pass
    seeds of its stars could not span a tetrahedron and the build failed as 'ambiguous triangulation' whatever the
    perturbation (4.7 % of all point counts).  The seed window of a short last block is the last 64 points now."""
    rng = np.random.default_rng(100 + n)
    pts = _kd(rng.uniform(-1, 1, size=(n, 3)))
    off0, adj0 = foam.delaunay_csr(pts)
    off, adj, info = S.delaunay(pts)
    assert info["bad"] == 0 and (info["status"] == 0).all()
    assert np.array_equal(off, off0) and np.array_equal(adj, adj0)
    # the incremental path with nothing usable in the seed lists falls back to the same window
    empty = (np.zeros(n + 1, dtype=np.uint32), np.zeros(0, dtype=np.uint32))
    off, adj, info = S.delaunay(pts, old=empty)
    assert info["bad"] == 0
    assert np.array_equal(off, off0) and np.array_equal(adj, adj0)


def _hub_cloud(rng):
    """A point at the centre of an empty shell of 3000 points (a floater inside a densely sampled surface): more
    than 2000 Delaunay neighbours."""
    shell = rng.normal(size=(3000, 3))
    shell /= np.linalg.norm(shell, axis=1, keepdims=True)
    outside = rng.uniform(-3, 3, size=(4000, 3))
    outside = outside[np.linalg.norm(outside, axis=1) > 1.5]
    return _kd(np.concatenate([shell * (1 + 1e-3 * rng.normal(size=(3000, 1))), np.zeros((1, 3)), outside]))


def _shell_cloud(rng, n=20000):
    """Every point on the rim: a fifth of the stars run out of budget in the first pass."""
    shell = rng.normal(size=(n, 3))
    shell /= np.linalg.norm(shell, axis=1, keepdims=True)
    return _kd(shell * (1 + 1e-3 * rng.normal(size=(n, 1))))


def test_host_hub_goes_through_the_third_instance():
    pts = _hub_cloud(np.random.default_rng(0))
    off0, adj0 = foam.delaunay_csr(pts)
    assert np.diff(off0.astype(np.int64)).max() > 2000
    _, _, info = S.delaunay(pts, stride=250)        # two instances only: the hub does not fit
    assert info["bad"] == 1 and (info["status"] == 1).sum() == 1
    off, adj, info = S.delaunay(pts, stride=4096)   # + Star<4096, 8188, uint16_t>: room for 4095 neighbours
    assert info["bad"] == 0
    assert np.array_equal(off, off0) and np.array_equal(adj, adj0)


@pytest.mark.parametrize("tree_knn", [0, 24])
def test_host_owner_build_equals_qhull(tree_knn):
    """The experimental two-pass build (every tetrahedron certified once: rf_star.hpp star_certify_owned / star_close,
    star records, the tree k-NN seeding) through the host harness: same lists, a third of the tree work."""
    rng = np.random.default_rng(8)
    for pts in (_kd(rng.uniform(-1, 1, size=(15000, 3))), _kd(_clustered(rng, 2000)),
                _kd(np.concatenate([np.c_[rng.uniform(-1, 1, size=(4000, 2)), 1e-3 * rng.normal(size=4000)],
                                    rng.uniform(-1, 1, size=(1000, 3))]))):
        off0, adj0 = foam.delaunay_csr(pts)
        _, _, base = S.delaunay(pts, sweep=False)   # the per-triangle certification this experiment was measured against
        off, adj, info = S.delaunay_owner(pts, tree_knn=tree_knn)
        assert info["bad"] == 0
        assert np.array_equal(off, off0) and np.array_equal(adj, adj0)
        assert info["closed_by_owner"] > 5   # certificates were really shared
        total = info["nodes_pass1"] + info["nodes_pass2"] + info["nodes_knn"]
        assert total < 0.85 * base["visited"].mean()


def test_host_incremental_seeds_give_the_same_lists():
    rng = np.random.default_rng(5)
    pts = _kd(rng.uniform(-1, 1, size=(8000, 3)))
    off, adj, _ = S.delaunay(pts)
    moved = (pts + rng.normal(0, 2e-3, size=pts.shape)).astype(np.float32)   # ~ 5 % of the point spacing
    off0, adj0 = foam.delaunay_csr(moved)
    for sweep in (False, True):
        off1, adj1, info = S.delaunay(moved, old=(off, adj), sweep=sweep)
        assert info["bad"] == 0
        assert np.array_equal(off1, off0) and np.array_equal(adj1, adj0)
    assert not np.array_equal(adj, adj0)   # the move did change the triangulation


def test_host_failures_are_reported_not_hidden():
    rng = np.random.default_rng(6)
    dup = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    dup[17] = dup[2000]
    _, _, info = S.delaunay(_kd(dup))
    assert (info["status"] == 4).sum() == 2              # both copies see each other
    # an integer grid is cospherical everywhere: the strict predicates must still terminate, and whatever diagonals
    # a star picks, the six axis neighbours of an interior point are Delaunay edges in every valid triangulation
    grid = _kd(np.stack(np.meshgrid(*[np.arange(10)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32))
    off, adj, info = S.delaunay(grid)
    index = {tuple(p): i for i, p in enumerate(grid.astype(np.int64))}
    for i, p in enumerate(grid.astype(np.int64)):
        if info["status"][i] != 0 or p.min() == 0 or p.max() == 9:
            continue
        row = set(adj[off[i]:off[i + 1]].tolist())
        for axis in range(3):
            for step in (-1, 1):
                q = p.copy()
                q[axis] += step
                assert index[tuple(q)] in row


@pytest.mark.parametrize("n", [33, 1000, 4096, 5001])
def test_cpu_aabb_tree_is_the_reference_tree(n):
    """radfoam.build_aabb_tree on CPU tensors (torch pooling) = the numpy restatement of build_aabb_tree
    (src/aabb_tree/aabb_tree.cu:192-283) the GPU kernel is checked against."""
    import torch
    import radfoam
    pts = _kd(np.random.default_rng(n).normal(size=(n, 3)))
    tree = radfoam.build_aabb_tree(torch.from_numpy(pts)).numpy()
    ref = S.aabb_tree(pts)
    assert tree.shape == ref.shape and np.array_equal(tree[:-1].view(np.uint32), ref[:-1].view(np.uint32))


def test_host_degenerate_clouds_terminate_and_say_so():
    """Inputs a training run can stumble into (coplanar / collinear / identical points, exact lattices, a unit sphere of
    normalised floats, denormal-sized coordinates): the star loop must end -- a lane that does not would hang the GPU --
    and every failure must surface as a star status or an unmatched edge, never as a silently wrong list."""
    rng = np.random.default_rng(0)
    n = 2500
    sphere = rng.normal(size=(n, 3))
    sphere /= np.linalg.norm(sphere, axis=1, keepdims=True)
    lattice = np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3)
    clouds = {
        "plane": (np.c_[rng.uniform(-1, 1, size=(n, 2)), np.zeros(n)], "all_fail"),
        "line": (np.c_[rng.uniform(-1, 1, size=n), np.zeros(n), np.zeros(n)], "all_fail"),
        "identical": (np.ones((200, 3)), "all_fail"),
        "duplicates": (np.repeat(rng.uniform(-1, 1, size=(n // 4, 3)), 4, axis=0), "all_fail"),
        "unit sphere": (sphere, "exact"),
        "tiny": (1e-30 * rng.uniform(-1, 1, size=(n, 3)), "exact"),
        "lattice + jitter": (lattice + 1e-6 * rng.normal(size=lattice.shape), "flagged_or_exact"),
    }
    for name, (cloud, expect) in clouds.items():
        pts = _kd(cloud)
        off, adj, info = S.delaunay(pts, stride=4096)
        rows = np.repeat(np.arange(len(off) - 1, dtype=np.int64), np.diff(off.astype(np.int64)))
        forward = rows * (1 << 32) | adj.astype(np.int64)
        backward = adj.astype(np.int64) * (1 << 32) | rows
        unmatched = np.setdiff1d(forward, backward).size
        if expect == "all_fail":
            assert info["bad"] == len(pts), name
            continue
        exact = False
        if info["bad"] == 0 and unmatched == 0:
            off0, adj0 = foam.delaunay_csr(pts)
            exact = np.array_equal(off, off0) and np.array_equal(adj, adj0)
        if expect == "exact":
            assert exact, name
        else:
            assert exact or info["bad"] > 0 or unmatched > 0, name


# ---- GPU ----------------------------------------------------------------------------------------------------------------

def _t(a, dev="cuda"):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [33, 1000, 4096, 5001])
def test_aabb_tree_bit_equal(n):
    import radfoam
    rng = np.random.default_rng(n)
    pts = _kd(rng.normal(size=(n, 3)))
    tree = radfoam.build_aabb_tree(_t(pts)).cpu().numpy()
    ref = S.aabb_tree(pts)
    assert tree.shape == ref.shape
    assert np.array_equal(tree[:-1].view(np.uint32), ref[:-1].view(np.uint32))   # the last entry is never written


@pytest.mark.gpu
@pytest.mark.parametrize("n", [100, 4097, 50000])
def test_kd_order_matches_the_reference_order(n):
    from radfoam_amd import triangulation
    rng = np.random.default_rng(n)
    pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    pts[::7, 1] = pts[(np.arange(0, n, 7) + 3) % n, 1]   # ties on one axis: the sorts are stable
    perm, out = triangulation.kd_order(_t(pts))
    ref = foam.kd_order(pts)
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), ref)
    assert np.array_equal(out.cpu().numpy(), pts[ref])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["uniform", "clustered", "sheet"])
def test_gpu_stars_equal_qhull(name):
    from radfoam_amd import triangulation
    rng = np.random.default_rng(11)
    if name == "uniform":
        pts = rng.uniform(-1, 1, size=(60000, 3))
    elif name == "clustered":
        pts = _clustered(rng, 6000)
    else:
        pts = np.concatenate([np.c_[rng.uniform(-1, 1, size=(8000, 2)), 1e-3 * rng.normal(size=8000)],
                              rng.uniform(-1, 1, size=(2000, 3))])
    pts = _kd(pts)
    off0, adj0 = foam.delaunay_csr(pts)
    adj, off, stats = triangulation.delaunay_adjacency(_t(pts))
    assert np.array_equal(off.cpu().numpy(), off0) and np.array_equal(adj.cpu().numpy(), adj0)
    assert stats["asymmetric_edges"] == 0 and stats["failed_stars"] == 0
    if name == "sheet":
        assert stats["large_stars"] > 0   # the large instance was exercised


@pytest.mark.gpu
@pytest.mark.parametrize("n", [65, 641, 642, 643, 1025, 100003])
def test_gpu_short_last_block(n):
    """n % 64 in 1..3: from scratch, incrementally with the previous lists, and incrementally with
    empty seed lists, through radfoam.Triangulation as the reference's scene drives it."""
    import torch
    import radfoam
    from radfoam_amd import triangulation
    rng = np.random.default_rng(200 + n)
    pts = _kd(rng.uniform(-1, 1, size=(n, 3)))
    off0, adj0 = foam.delaunay_csr(pts)
    adj, off, stats = triangulation.delaunay_adjacency(_t(pts))
    assert stats["failed_stars"] == 0 and stats["asymmetric_edges"] == 0
    assert np.array_equal(off.cpu().numpy(), off0) and np.array_equal(adj.cpu().numpy(), adj0)
    empty = (torch.zeros(0, dtype=torch.int32, device="cuda").view(torch.uint32),
             torch.zeros(n + 1, dtype=torch.int32, device="cuda").view(torch.uint32))
    adj, off, stats = triangulation.delaunay_adjacency(_t(pts), seed=empty)
    assert np.array_equal(off.cpu().numpy(), off0) and np.array_equal(adj.cpu().numpy(), adj0)
    tri = radfoam.Triangulation(_t(pts))
    moved = (pts + rng.normal(0, 1e-4, size=pts.shape)).astype(np.float32)
    assert tri.rebuild(_t(moved), incremental=True) is False
    off1, adj1 = foam.delaunay_csr(moved)
    assert np.array_equal(tri.point_adjacency().cpu().numpy(), adj1)
    assert np.array_equal(tri.point_adjacency_offsets().cpu().numpy(), off1)


@pytest.mark.gpu
def test_gpu_hub_and_shell():
    """The two clouds the default sizes are not made for: a hub with > 2000 neighbours (third pass, a block per star,
    shared insertions) and a sphere shell whose rim stars outnumber the default second-pass rows (the call asks for a
    larger workspace and the binding comes back with one)."""
    from radfoam_amd import triangulation
    pts = _hub_cloud(np.random.default_rng(0))
    off0, adj0 = foam.delaunay_csr(pts)
    adj, off, stats = triangulation.delaunay_adjacency(_t(pts))
    assert int(np.diff(off.cpu().numpy().astype(np.int64)).max()) > 2000
    assert np.array_equal(off.cpu().numpy(), off0) and np.array_equal(adj.cpu().numpy(), adj0)
    pts = _shell_cloud(np.random.default_rng(2))
    off0, adj0 = foam.delaunay_csr(pts)
    adj, off, stats = triangulation.delaunay_adjacency(_t(pts))
    assert stats["large_stars"] > 1024   # more than the default workspace has rows for
    assert np.array_equal(off.cpu().numpy(), off0) and np.array_equal(adj.cpu().numpy(), adj0)


@pytest.mark.gpu
@pytest.mark.parametrize("waves,group", [(4, 1), (4, 4), (8, 4), (16, 16)])
def test_gpu_second_pass_configurations(waves, group):
    """The second pass's other instantiations (waves per rim star x waves per query; the default is 8 x 8): same lists.
    The knobs are read once per process, so each configuration gets a process of its own."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "from tests.test_delaunay import _shell_cloud, _t\n"
        "from radfoam_amd import foam, triangulation\n"
        "pts = _shell_cloud(np.random.default_rng(2), 6000)\n"
        "off0, adj0 = foam.delaunay_csr(pts)\n"
        "adj, off, stats = triangulation.delaunay_adjacency(_t(pts))\n"
        "assert stats['large_stars'] > 100\n"
        "assert np.array_equal(off.cpu().numpy(), off0) and np.array_equal(adj.cpu().numpy(), adj0)\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, RF_DELAUNAY_COOP_WAVES=str(waves), RF_DELAUNAY_COOP_GROUP=str(group))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-2000:]


@pytest.mark.gpu
def test_gpu_stars_on_the_cached_foams():
    """Whole BASELINE foams: the lists equal the cached Qhull CSR (500 k always; 2 M when its cache is here)."""
    import os
    import torch
    from radfoam_amd import triangulation
    cases = [(500_000, 1)]
    if os.path.exists(os.path.join(foam.default_cache_dir(), "foam_n2000000_s5.npz")):
        cases.append((2_000_000, 5))
    for n, seed in cases:
        fm = foam.make_synthetic_foam(n, 0, seed, cache_dir=foam.default_cache_dir())
        pts = _t(fm["points"])
        torch.cuda.synchronize()
        adj, off, stats = triangulation.delaunay_adjacency(pts)
        assert np.array_equal(off.cpu().numpy(), fm["point_adjacency_offsets"]), n
        assert np.array_equal(adj.cpu().numpy(), fm["point_adjacency"]), n
        assert stats["asymmetric_edges"] == 0 and stats["failed_stars"] == 0
        assert stats["large_stars"] > 0   # the rim of the cloud went through the second pass
        # an optimiser step later (3 % of the spacing): the previous lists as candidates give the lists a fresh build
        # of the moved points gives (the second pass must see every hull candidate for that: stars that outgrew the
        # small instance included)
        g = torch.Generator("cuda").manual_seed(n)
        moved = pts + 0.03 * (8.0 / n) ** (1 / 3) * torch.randn(pts.shape, device="cuda", generator=g)
        adj1, off1, _ = triangulation.delaunay_adjacency(moved, seed=(adj, off))
        adj2, off2, _ = triangulation.delaunay_adjacency(moved)
        assert torch.equal(adj1.view(torch.int32), adj2.view(torch.int32))
        assert torch.equal(off1.view(torch.int32), off2.view(torch.int32))
        assert not torch.equal(off1.view(torch.int32), off.view(torch.int32))


@pytest.mark.gpu
def test_triangulation_protocol():
    """radfoam.Triangulation as RadFoamScene uses it (scene.py:65-72,160-200): permutation, incremental rebuild
    without a new permutation, full rebuild with one, errors the caller retries on."""
    import torch
    import radfoam
    rng = np.random.default_rng(21)
    raw = rng.uniform(-1, 1, size=(20000, 3)).astype(np.float32)
    tri = radfoam.Triangulation(_t(raw))
    perm = tri.permutation()
    assert perm.dtype == torch.uint32 and tri.point_adjacency().dtype == torch.uint32
    assert tri.point_adjacency_offsets().dtype == torch.uint32
    order = foam.kd_order(raw)
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), order)
    pts = raw[order]
    off0, adj0 = foam.delaunay_csr(pts)
    assert np.array_equal(tri.point_adjacency().cpu().numpy(), adj0)
    assert np.array_equal(tri.point_adjacency_offsets().cpu().numpy(), off0)
    # an optimiser step later: same order, previous lists as candidates
    moved = (pts + rng.normal(0, 1e-3, size=pts.shape)).astype(np.float32)
    assert tri.rebuild(_t(moved), incremental=True) is False
    off1, adj1 = foam.delaunay_csr(moved)
    assert np.array_equal(tri.point_adjacency().cpu().numpy(), adj1)
    assert np.array_equal(tri.point_adjacency_offsets().cpu().numpy(), off1)
    # a different point count cannot be incremental (delaunay.cu:293): full rebuild, new permutation
    more = np.concatenate([moved, rng.uniform(-1, 1, size=(500, 3)).astype(np.float32)])
    assert tri.rebuild(_t(more), incremental=True) is True
    order2 = tri.permutation().cpu().numpy().astype(np.int64)
    off2, adj2 = foam.delaunay_csr(more[order2])
    assert np.array_equal(tri.point_adjacency().cpu().numpy(), adj2)
    # failures surface as the reference's exception
    more[5] = more[777]
    with pytest.raises(radfoam.TriangulationFailedError, match="duplicate points found"):
        tri.rebuild(_t(more))
    with pytest.raises(RuntimeError, match="less than 32 points"):
        radfoam.Triangulation(_t(more[:20]))
    bad = more.copy()
    bad[3, 1] = np.nan
    with pytest.raises(radfoam.TriangulationFailedError):
        tri.rebuild(_t(bad))


@pytest.mark.gpu
def test_traced_image_through_a_gpu_built_foam(foam_factory):
    """End to end: a foam triangulated on the GPU traces to the same image, bit for bit, as the Qhull-built one."""
    import torch
    import radfoam
    fm = foam.make_synthetic_foam(20000, 2, 9)
    tri = radfoam.Triangulation(_t(fm["points"]))   # points are already in kd-order: identity permutation
    assert np.array_equal(tri.permutation().cpu().numpy().astype(np.int64), np.arange(20000))
    cam = foam.default_camera(96, 64)
    rays = _t(foam.camera_rays(cam))
    start = foam.nearest_point(fm["points"], cam["position"])
    s = torch.full(rays.shape[:-1], start, dtype=torch.int64).to(torch.uint32).to("cuda")
    pipe = radfoam.create_pipeline(2)
    a = pipe.trace_forward(_t(fm["points"]), _t(fm["attributes"]), tri.point_adjacency(),
                           tri.point_adjacency_offsets(), rays, s)
    b = pipe.trace_forward(_t(fm["points"]), _t(fm["attributes"]), _t(fm["point_adjacency"]),
                           _t(fm["point_adjacency_offsets"]), rays, s)
    assert torch.equal(a["rgba"], b["rgba"]) and torch.equal(a["num_intersections"].view(torch.int32),
                                                             b["num_intersections"].view(torch.int32))
