"""CPU: the oracle validated against things that do not depend on it.

The reference pins nothing (no tests, no golden vectors: SURVEY.md section 4), so the oracle is
checked by (i) closed-form cases, (ii) invariants of the algorithm, (iii) a float64 twin,
(iv) central finite differences of the twin's forward against its backward, and (v) the exact
form of each documented quirk of the reference's backward (strict vs clean).
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from radfoam_amd import foam as foam_mod
from tests import helpers as H


def test_half_conversion_matches_ieee():
    L = O.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, 4000), rng.normal(0, 1e-5, 4000), rng.normal(0, 2e4, 2000),
                        [0.0, -0.0, 65504, 65519.9, 65520, 1e-8, 6e-8, 2.9802322e-8, 5.96e-8, 6.1e-5, np.inf, -np.inf]])
    x = x.astype(np.float32)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    got = np.array([L.rfo_float_to_half(float(v)) for v in x], dtype=np.uint16)
    np.testing.assert_array_equal(got, ref)
    hs = np.arange(0, 65536, 7, dtype=np.uint16)
    back = np.array([L.rfo_half_to_float(int(h)) for h in hs], dtype=np.float32)
    refb = hs.view(np.float16).astype(np.float32)
    assert ((back == refb) | (np.isnan(back) & np.isnan(refb))).all()


def test_exp_log_within_one_ulp():
    L = O.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([-rng.uniform(0, 20, 8000), rng.uniform(-87, 88, 4000)]).astype(np.float32)
    e = np.array([L.rfo_expf(float(v)) for v in xs], dtype=np.float32).astype(np.float64)
    true = np.exp(xs.astype(np.float64))
    ulp = np.spacing(true.astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(e - true) / ulp) < 1.0
    xs = np.concatenate([rng.uniform(0.5, 4, 8000), 10 ** rng.uniform(-37, 38, 4000)]).astype(np.float32)
    l = np.array([L.rfo_logf(float(v)) for v in xs], dtype=np.float32).astype(np.float64)
    true = np.log(xs.astype(np.float64))
    ulp = np.spacing(np.abs(true).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(l - true) / np.maximum(ulp, 1e-45)) < 1.0
    assert L.rfo_expf(0.0) == 1.0 and L.rfo_expf(-1000.0) == 0.0 and np.isinf(L.rfo_expf(100.0))
    assert L.rfo_logf(1.0) == 0.0 and np.isneginf(L.rfo_logf(0.0)) and np.isnan(L.rfo_logf(-1.0))


def _two_point_slab(s0=0.7, s1=1.3):
    """Two sites on the z axis: cells are the half-spaces z<0 and z>0."""
    pts = np.array([[0, 0, -0.5], [0, 0, 0.5]], dtype=np.float32)
    adj = np.array([1, 0], dtype=np.uint32)
    off = np.array([0, 1, 2], dtype=np.uint32)
    attrs = np.array([[0.2, -0.1, 0.3, s0], [-0.3, 0.4, 0.1, s1]], dtype=np.float32)
    return pts, attrs, adj, off


def test_closed_form_two_cells():
    pts, attrs, adj, off = _two_point_slab()
    ray = np.array([[0, 0, -2.0, 0, 0, 3.0]], dtype=np.float32)  # un-normalised direction
    out = O.trace_forward(0, pts, attrs, adj, off, ray, np.uint32(0), return_contribution=True)
    # cell 0 spans t in [0,2]; cell 1 is unbounded along the ray: no exit face -> the walk stops
    # before compositing it (tracing_utils.cuh:69-71)
    c0 = 0.28209479177387814
    rgb0 = np.maximum(0.5 + c0 * attrs[0, :3].astype(np.float64), 0)
    a0 = 1 - np.exp(-0.7 * 2.0)
    np.testing.assert_allclose(out["rgba"][0, :3], a0 * rgb0, rtol=2e-6)
    np.testing.assert_allclose(out["rgba"][0, 3], a0, rtol=2e-6)
    assert out["num_intersections"][0, 0] == 2
    np.testing.assert_allclose(out["contribution"][:, 0], [a0, 0.0], rtol=2e-6)
    # looking away: exits through the first cell immediately
    out = O.trace_forward(0, pts, attrs, adj, off, ray * np.array([1, 1, 1, 1, 1, -1], np.float32), np.uint32(0))
    assert out["num_intersections"][0, 0] == 1 and not out["rgba"].any()
    # depth quantile inside the first slab: T(t)=exp(-s t) = q  ->  t = ln(1/q)/s
    q = np.array([[0.6, 0.1]], dtype=np.float32)
    out = O.trace_forward(0, pts, attrs, adj, off, ray, np.uint32(0), depth_quantiles=q)
    np.testing.assert_allclose(out["depth"][0, 0], np.log(1 / 0.6) / 0.7, rtol=2e-6)
    assert out["depth_indices"][0, 0] == 0
    assert out["depth"][0, 1] == -1.0 and out["depth_indices"][0, 1] == 0xFFFFFFFF  # T never drops below 0.1


def test_step_cap_and_threshold():
    fm = H.__dict__["foam_mod"].make_synthetic_foam(2000, 0, 3)
    cam, rays, start = H.camera_setup(fm, 16, 12)
    args = (0, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    full = O.trace_forward(*args, rays, start)
    for cap in (0, 1, 5):
        out = O.trace_forward(*args, rays, start, max_intersections=cap)
        assert (out["num_intersections"] <= cap + 1).all()
        assert (out["num_intersections"] == np.minimum(full["num_intersections"], cap + 1)).all()
    hi = O.trace_forward(*args, rays, start, weight_threshold=0.5)
    assert (hi["num_intersections"] <= full["num_intersections"]).all()
    assert (1 - hi["rgba"][..., 3] >= 0).all()


@pytest.mark.parametrize("d", [0, 1, 2, 3])
def test_invariants_and_twin(foam_factory, d):
    fm = foam_factory(3000, d, 7)
    cam, rays, start = H.camera_setup(fm, 24, 16)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    out = O.trace_forward(*args, rays, start, return_contribution=True, want_stats=True)
    assert abs(out["contribution"].astype(np.float64).sum() - out["rgba"][..., 3].astype(np.float64).sum()) < 1e-3
    st = out["stats"]
    assert st["cells_scanned"] == int(out["num_intersections"].sum())  # no ray hits the step cap here
    assert st["hops"] >= st["segments"] >= st["segments_lit"] > 0
    # threads do not change per-ray outputs
    one = O.trace_forward(*args, rays, start, num_threads=1)
    np.testing.assert_array_equal(one["rgba"].view(np.uint32), out["rgba"].view(np.uint32))
    # float64 twin with exact face offsets: same image up to fp16 face rounding
    a64 = (d, fm["points"].astype(np.float64), fm["attributes"].astype(np.float64), fm["point_adjacency"],
           fm["point_adjacency_offsets"])
    f64 = O.trace_forward_f64(*a64, rays.astype(np.float64), start)
    assert np.abs(f64["rgba"] - out["rgba"]).max() < 5e-3
    assert (f64["num_intersections"] == out["num_intersections"]).mean() > 0.95


def _dense_inside_scene(foam_factory, d):
    """Camera INSIDE a foam dense enough that rays saturate: exercises the reference's
    first-cell and dropped-tail behaviours."""
    fm = foam_factory(3000, d, 9)
    attrs = fm["attributes"].astype(np.float64).copy()
    attrs[:, -1] *= 12.0
    pts = fm["points"].astype(np.float64)
    cam, rays, _ = H.camera_setup(fm, 12, 8, position=(0.03, -0.02, 0.05))
    start = np.uint32(foam_mod.nearest_point(fm["points"], cam["position"]))
    return fm, pts, attrs, rays.astype(np.float64), start


@pytest.mark.parametrize("use_q", [False, True])
def test_backward_against_finite_differences(foam_factory, use_q):
    d = 1
    fm, pts, attrs, rays, start = _dense_inside_scene(foam_factory, d)
    adj, off = fm["point_adjacency"], fm["point_adjacency_offsets"]
    rng = np.random.default_rng(3)
    g = rng.normal(size=rays.shape[:-1] + (4,))
    q = np.sort(rng.uniform(0.05, 0.95, size=rays.shape[:-1] + (2,)), axis=-1)[..., ::-1].copy() if use_q else None
    dg = rng.normal(size=rays.shape[:-1] + (2,)) if use_q else None

    def loss(p, a):
        out = O.trace_forward_f64(d, p, a, adj, off, rays, start, depth_quantiles=q)
        val = (out["rgba"] * g).sum()
        if use_q:
            val += (np.where(out["depth_indices"] != O.NONE, out["depth"], 0.0) * dg).sum()
        return val

    f = O.trace_forward_f64(d, pts, attrs, adj, off, rays, start, depth_quantiles=q)
    assert (f["rgba"][..., 3] > 0.999).mean() > 0.5  # rays do saturate
    clean = O.trace_backward_f64(d, pts, attrs, adj, off, rays, start, f["rgba"], g, depth_quantiles=q,
                                 depth_indices=f.get("depth_indices"), depth_grad_in=dg, strict=False)
    # tolerance 3e-4: the reference's "+1e-6" regularisers (pipeline.cu:246,251) bias the analytic
    # gradient by ~1e-6/(1-alpha) per cell, visible in this deliberately dense scene (the thin-foam
    # variant of this check agrees to 1e-7)
    lit = np.where(attrs[:, -1] > 0)[0]
    rows = lit[np.argsort(-np.abs(clean["attr_grad"][lit]).sum(1))[:3]]
    for i in rows:
        for c in (0, 4, attrs.shape[1] - 1):
            h = 1e-6
            ap, am = attrs.copy(), attrs.copy()
            ap[i, c] += h
            am[i, c] -= h
            fd = (loss(pts, ap) - loss(pts, am)) / (2 * h)
            assert abs(fd - clean["attr_grad"][i, c]) <= 3e-4 * max(1.0, abs(fd)), (i, c, fd, clean["attr_grad"][i, c])
    rows = np.argsort(-np.abs(clean["points_grad"]).sum(1))[:3]
    for i in rows:
        for c in range(3):
            h = 1e-7
            pp, pm = pts.copy(), pts.copy()
            pp[i, c] += h
            pm[i, c] -= h
            fd = (loss(pp, attrs) - loss(pm, attrs)) / (2 * h)
            assert abs(fd - clean["points_grad"][i, c]) <= 3e-4 * max(1.0, abs(fd)), (i, c, fd, clean["points_grad"][i, c])


def test_reference_quirks_have_their_documented_form(foam_factory):
    """strict (the reference's behaviour) vs clean differ exactly as SURVEY.md Appendix A.4 says."""
    d = 0
    fm, pts, attrs, rays, start = _dense_inside_scene(foam_factory, d)
    adj, off = fm["point_adjacency"], fm["point_adjacency_offsets"]
    rng = np.random.default_rng(4)
    g = rng.normal(size=rays.shape[:-1] + (4,))
    f = O.trace_forward_f64(d, pts, attrs, adj, off, rays, start)
    strict = O.trace_backward_f64(d, pts, attrs, adj, off, rays, start, f["rgba"], g, strict=True)
    clean = O.trace_backward_f64(d, pts, attrs, adj, off, rays, start, f["rgba"], g, strict=False)
    # attribute gradients are not affected by any quirk when no depth quantiles are used
    np.testing.assert_array_equal(strict["attr_grad"], clean["attr_grad"])
    # point gradients are: dropped tails + the phantom first-cell term
    dpg = strict["points_grad"] - clean["points_grad"]
    assert np.abs(dpg).max() > 1e-3
    # single ray: strict misses the last visited cell's and its exit neighbour's accumulators and
    # adds a term on the start cell only
    r1 = rays[4:5, 6:7]
    g1 = g[4:5, 6:7]
    f1 = O.trace_forward_f64(d, pts, attrs, adj, off, r1, start)
    s1 = O.trace_backward_f64(d, pts, attrs, adj, off, r1, start, f1["rgba"], g1, strict=True)
    c1 = O.trace_backward_f64(d, pts, attrs, adj, off, r1, start, f1["rgba"], g1, strict=False)
    rows = np.where(np.abs(s1["points_grad"] - c1["points_grad"]).sum(1) > 0)[0]
    assert 1 <= len(rows) <= 3
    assert int(start) in rows  # phantom dt0/dP against the world origin on the first segment
    # fp32 strict oracle agrees with the strict twin to fp16-face-table accuracy
    a32 = (d, fm["points"], attrs.astype(np.float32), adj, off)
    f32 = O.trace_forward(*a32, rays.astype(np.float32), start)
    b32 = O.trace_backward(*a32, rays.astype(np.float32), start, f32["rgba"], g.astype(np.float32))
    rel = np.linalg.norm(b32["attr_grad"] - strict["attr_grad"]) / np.linalg.norm(strict["attr_grad"])
    assert rel < 0.05


def test_benchmark_path_equals_forward(foam_factory):
    """trace_benchmark == trace_forward on cast_ray's rays, quantised (pipeline.cu:483-543)."""
    d = 2
    fm = foam_factory(3000, d, 5)
    cam, _, start = H.camera_setup(fm, 40, 24)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    diff = O.build_adjacent_diff(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    img = O.trace_benchmark(*args, diff, cam, start, weight_threshold=0.05)
    rays = O.cast_rays(cam)
    fwd = O.trace_forward(*args, rays, start, weight_threshold=0.05, diff=diff)
    rgb = np.clip(fwd["rgba"][..., :3], 0, 1)
    exp = (rgb * np.float32(255.0)).astype(np.int32)
    got = np.stack([(img >> s) & 0xFF for s in (0, 8, 16)], -1).astype(np.int32)
    np.testing.assert_array_equal(got, exp)
    assert ((img >> 24) == 255).all()
    # the table equals torch-style (q - p).half() as benchmark.py builds it
    rows = np.repeat(np.arange(3000), np.diff(fm["point_adjacency_offsets"].astype(np.int64)))
    with np.errstate(over="ignore"):
        exp_tab = (fm["points"][fm["point_adjacency"]] - fm["points"][rows]).astype(np.float16)
    np.testing.assert_array_equal(diff[:, :3], exp_tab.view(np.uint16))
    assert not diff[:, 3].any()


def test_filtered_evaluation_of_the_scan_equals_the_references_bit_for_bit(foam_factory):
    """The HIP kernels do not divide every face: they run a tournament on cross-multiplied products, certify it by the
    smallest bit distance of any two compared products, and hand contested cells (distance <= 3) to the dividing scan
    (radfoam_amd/csrc/rf_kernels.hip, "the face scan", with the derivation of the bound).  The oracle mirrors that
    evaluation step for step -- padded lists with scaled copies included -- as ``scan_mode("filtered")``; it must be
    the SAME FUNCTION as the reference's evaluation (every face divided, running minimum of rounded quotients,
    (P + o/2) - O; tracing_utils.cuh:43-67): 57,600 rays x ~60 cells x ~17 faces = 6e7 face tests, forward outputs and
    gradients bit-equal, and the contested cells counted (about one in 10^4 at this foam's cell size)."""
    fm = foam_factory(60000, 2, 7)
    cam, rays, start = H.camera_setup(fm, 320, 180)
    args = (2, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    g = np.random.default_rng(4).normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    a = O.trace_forward(*args, rays, start)
    ab = O.trace_backward(*args, rays, start, a["rgba"], g, num_threads=1)
    assert O.lib().rfo_get_scan_mode() == 0
    with O.scan_mode("filtered") as m:
        assert O.lib().rfo_get_scan_mode() == 1
        b = O.trace_forward(*args, rays, start)
        contested = m.contested
        bb = O.trace_backward(*args, rays, start, b["rgba"], g, num_threads=1)
    assert O.lib().rfo_get_scan_mode() == 0
    cells = int(a["num_intersections"].sum())
    assert cells > 3_000_000
    assert 50 <= contested <= cells // 2000, (contested, cells)     # observed 447 of 3.47e6
    for k in ("rgba", "num_intersections"):
        np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
    for k in ("points_grad", "attr_grad"):      # one thread: the same sums in the same order
        np.testing.assert_array_equal(ab[k].view(np.uint32), bb[k].view(np.uint32))


@pytest.mark.parametrize("case", ["grid", "axis_rays", "origin_on_bisector", "tiny_direction", "duplicate_points"])
def test_filtered_evaluation_on_degenerate_geometry(case):
    """Exact ties, zero products, exactly axis-aligned rays on a lattice of points (whole faces with o.d = 0), a ray
    origin exactly on a bisector (num = 0), direction components of 1e-30 and coincident neighbours: everywhere the
    certificate must send the doubtful cells to the dividing scan, so the filtered evaluation still equals the
    reference's."""
    from radfoam_amd import foam
    rng = np.random.default_rng(11)
    if case == "duplicate_points":
        pts = rng.uniform(-1, 1, size=(400, 3)).astype(np.float32)
        pts[200:300] = pts[100:200] + np.float32(1e-4)         # pairs closer than fp16 resolves relative to the others
    else:
        g = np.arange(-3, 4, dtype=np.float32) * np.float32(0.25)
        pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        if case != "grid":
            pts = pts + (rng.uniform(-1, 1, size=pts.shape) * 1e-3).astype(np.float32) * (case == "tiny_direction")
    pts = np.ascontiguousarray(pts.astype(np.float32)[foam.kd_order(pts.astype(np.float32))])
    offsets, adjacency = foam.delaunay_csr(pts)
    attrs = rng.normal(0.0, 0.3, size=(len(pts), 13)).astype(np.float32)
    attrs[:, 12] = rng.uniform(0.5, 3.0, size=len(pts)).astype(np.float32)
    fm = {"points": pts, "attributes": attrs, "point_adjacency": adjacency, "point_adjacency_offsets": offsets}
    n = 4096
    o = np.zeros((n, 3), np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    if case in ("grid", "axis_rays"):
        d[: n // 2] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 2)] * rng.choice([-1, 1], n // 2)[:, None]
        o[:] = pts[rng.integers(0, len(pts), n)] * (case == "axis_rays")
    if case == "origin_on_bisector":
        i = rng.integers(0, len(pts), n)
        nb = fm["point_adjacency"][fm["point_adjacency_offsets"][i].astype(np.int64)].astype(np.int64)
        o[:] = (pts[i] + pts[nb]) * np.float32(0.5)
    if case == "tiny_direction":
        d[:, 0] = np.float32(1e-30)
        d[: n // 2, 1] = 0
    rays = np.concatenate([o, d], -1).astype(np.float32)
    from scipy.spatial import cKDTree
    start = cKDTree(pts.astype(np.float64)).query(o.astype(np.float64))[1].astype(np.uint32)
    args = (1, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    a = O.trace_forward(*args, rays, start)
    with O.scan_mode("filtered") as m:
        b = O.trace_forward(*args, rays, start)
        contested = m.contested
    np.testing.assert_array_equal(a["num_intersections"], b["num_intersections"])
    np.testing.assert_array_equal(a["rgba"].view(np.uint32), b["rgba"].view(np.uint32))
    if case in ("grid", "axis_rays"):
        assert contested > 0                                   # a lattice is nothing but ties


def test_the_certificate_threshold_is_not_decorative(tmp_path):
    """The filtered evaluation hands a cell to the dividing scan when two compared products are at most 3 floats apart
    (proved sufficient in rf_kernels.hip, "the face scan").  Built with the threshold at 0 (only EQUAL products are
    contested) the mirror must disagree with the reference's evaluation on some rays of a 230,400-ray frame -- otherwise the
    tests above could not tell a certificate from none; at the shipped threshold it agrees on all of them."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(O.__file__))
    lib0 = str(tmp_path / "liboracle_tie0.so")
    subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                    "-DRFO_TIE_ULPS=0u", os.path.join(here, "rf_oracle.c"), "-o", lib0, "-shared", "-lm"], check=True)
    script = r"""
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import oracle as O
if sys.argv[1] != "shipped":
    O._LIB_PATH = sys.argv[1]; O.build = lambda *a, **k: None
import helpers as H
from radfoam_amd import foam
fm = foam.make_synthetic_foam(60000, 2, 7)
cam, rays, start = H.camera_setup(fm, 640, 360)
args = (2, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
a = O.trace_forward(*args, rays, start)
with O.scan_mode("filtered") as m:
    b = O.trace_forward(*args, rays, start)
    c = m.contested
print(json.dumps({"contested": c, "rays_differing": int((a["num_intersections"] != b["num_intersections"]).sum())}))
""" % (os.path.dirname(here), os.path.dirname(os.path.abspath(__file__)))
    import json
    out = {}
    for name in (lib0, "shipped"):
        r = subprocess.run([sys.executable, "-c", script, name], capture_output=True, text=True, check=True)
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out[lib0]["rays_differing"] >= 3, out                 # observed 19
    assert out["shipped"]["rays_differing"] == 0 and out["shipped"]["contested"] > out[lib0]["contested"], out


def test_certificate_on_adversarial_pairs_of_quotients():
    """The statement the certificate rests on, tested where it is tight instead of where frames happen to go: for
    fp32 (num_a, dp_a), (num_b, dp_b) with dp > 0, let p1 = RN(num_a * dp_b), p2 = RN(num_b * dp_a) and D the distance of
    their bit patterns.  If D >= 4 then RN(num_a / dp_a) and RN(num_b / dp_b) are STRICTLY ordered the way p1 and p2
    are (rf_kernels.hip "the face scan": no tie, so no lowest-index rule to get wrong, and the divided scan would have
    picked the same face).  4e7 pairs constructed so that the two quotients are 0-12 ulp apart, across 60 binades of
    every operand, both signs of num, products on either side of a power of two (where the ulp halves).  The converse
    is also shown: among pairs at D <= 3 the products DO misorder or tie rounded quotients -- the fallback is needed --
    and the bound is within two of tight: a certificate at D >= 2 lets wrong orders through (observed: D >= 1 222,634,
    D >= 2 2,855, D >= 3 none of 1.9e7 pairs, D >= 4 -- shipped, proved -- none)."""
    rng = np.random.default_rng(2024)
    n = 4_000_000
    violations = 0
    certified = 0
    needed = 0
    loose = 0
    with np.errstate(over="ignore", under="ignore"):
        for rnd in range(10):
            def operand(sign=False):
                m = rng.uniform(1.0, 2.0, n)
                if rnd % 2:                                # half the rounds: mantissas crowded at the binade's ends
                    m = np.where(rng.random(n) < 0.5, 1.0 + rng.uniform(0, 2.0 ** -20, n), 2.0 - rng.uniform(0, 2.0 ** -20, n))
                x = m * 2.0 ** rng.integers(-30, 31, n)
                return (x * (rng.choice([-1.0, 1.0], n) if sign else 1.0)).astype(np.float32)
            na, da, db = operand(True), operand(), operand()
            # num_b so that num_b/db is within a few ulp of num_a/da: exact target in float64, then a step of k floats
            target = (na.astype(np.float64) / da.astype(np.float64) * db.astype(np.float64)).astype(np.float32)
            k = rng.integers(-12, 13, n).astype(np.int32)
            bits = target.view(np.int32) + np.where(target < 0, -k, k)
            nb = bits.view(np.float32)
            ok = np.isfinite(nb) & (nb != 0) & (np.sign(nb) == np.sign(na))
            na, da, db, nb = na[ok], da[ok], db[ok], nb[ok]
            # RN of an fp32 x fp32 product: exact in float64 (48 bits), one rounding to fp32
            p1 = (na.astype(np.float64) * db.astype(np.float64)).astype(np.float32)
            p2 = (nb.astype(np.float64) * da.astype(np.float64)).astype(np.float32)
            dist = np.abs(p1.view(np.uint32).astype(np.int64) - p2.view(np.uint32).astype(np.int64))
            qa, qb = na / da, nb / db                   # IEEE fp32 divisions: what the reference compares
            sure = dist >= 4
            certified += int(sure.sum())
            wrong = sure & (((p1 < p2) != (qa < qb)) | (qa == qb))
            violations += int(wrong.sum())
            needed += int((~sure & (((p1 < p2) != (qa < qb)) | (qa == qb))).sum())
            loose += int(((dist >= 2) & (((p1 < p2) != (qa < qb)) | (qa == qb))).sum())
    assert certified > 20_000_000, certified
    assert violations == 0, violations
    assert needed > 100_000, needed                     # below the threshold the products alone do get it wrong
    assert loose > 1000, loose                          # ... and two floats of distance are not enough


def test_certificate_needs_normal_quotients_and_the_fail_safe_catches_the_rest():
    """the certificate's proof assumes the compared quotients are NORMAL floats.  With subnormal
    quotients it does fail -- two exits a few percent apart round to the same subnormal float, the reference keeps the first,
    the products order them strictly (D >= 4) -- and the kernels' fail-safe (scan_end: a winning quotient with a zero
    exponent field sends the cell to the dividing scan; mirrored by the oracle's filtered mode) catches every such pair:
    among pairs certified at D >= 4, every wrong order or tie has min(|qa|, |qb|) below 2^-126."""
    rng = np.random.default_rng(7)
    n = 2_000_000
    wrong_total = wrong_caught = 0
    with np.errstate(over="ignore", under="ignore"):
        for rnd in range(4):
            # numerators tiny (a ray origin within 1e-40 of a bisector through the coordinate origin), denominators ordinary
            na = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-149, -120, n)).astype(np.float32)
            da = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-3, 4, n)).astype(np.float32)
            db = (rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-3, 4, n)).astype(np.float32)
            rel = 1.0 + rng.uniform(-0.3, 0.3, n)
            nb = (na.astype(np.float64) / da.astype(np.float64) * db.astype(np.float64) * rel).astype(np.float32)
            ok = (na != 0) & (nb != 0)
            na, da, db, nb = na[ok], da[ok], db[ok], nb[ok]
            p1 = (na.astype(np.float64) * db.astype(np.float64)).astype(np.float32)
            p2 = (nb.astype(np.float64) * da.astype(np.float64)).astype(np.float32)
            dist = np.abs(p1.view(np.uint32).astype(np.int64) - p2.view(np.uint32).astype(np.int64))
            qa, qb = na / da, nb / db
            sure = dist >= 4
            wrong = sure & (((p1 < p2) != (qa < qb)) | (qa == qb))
            tiny = (np.minimum(np.abs(qa), np.abs(qb)).view(np.uint32) & 0x7F800000) == 0
            wrong_total += int(wrong.sum())
            wrong_caught += int((wrong & tiny).sum())
    assert wrong_total > 1000, wrong_total          # the assumption is needed ...
    assert wrong_caught == wrong_total              # ... and a zero exponent field of the smaller quotient finds every case


def _bisectors_through_the_origin(shift=0.0):
    """A hand-built cell whose exits are at SUBNORMAL distances: the cell's point at (-1/2, 0, 0) (+ shift in x), neighbours
    mirrored through planes that pass within 1e-40 of the ray origin."""
    # cell 0 in the middle, six neighbours; bisector of 0 and 1 is the plane x = 0 exactly (points -a and +a)
    pts = np.array([[-0.5, 0.0, 0.0], [0.5, 0.0, 0.0], [-0.5, 1.0, 0.0], [-0.5, -1.0, 0.0], [-0.5, 0.0, 1.0],
                    [-0.5, 0.0, -1.0], [-1.5, 0.0, 0.0], [0.5, 0.25, 0.0]], dtype=np.float32)
    pts[:, 0] += np.float32(shift)
    adj, off = [], [0]
    nbrs = {0: [1, 2, 3, 4, 5, 6, 7], 1: [0, 7], 2: [0], 3: [0], 4: [0], 5: [0], 6: [0], 7: [0, 1]}
    for i in range(8):
        adj += nbrs[i]
        off.append(len(adj))
    return pts, np.array(adj, dtype=np.uint32), np.array(off, dtype=np.uint32)


def test_filtered_evaluation_with_subnormal_exit_distances():
    """Rays that start within a few subnormals of a bisector (t = 1e-41 .. 1e-38 to the first exit) and foams scaled down
    until their fp16 offsets and exit distances underflow: the filtered evaluation (with its fail-safe) equals the
    reference's loop bit for bit, and the fail-safe does fire."""
    from oracle import oracle as O
    pts, adj, off = _bisectors_through_the_origin()
    att = np.zeros((8, 4), dtype=np.float32)
    att[:, :3] = 0.3
    att[:, 3] = 2.0
    rng = np.random.default_rng(3)
    n = 4096
    rays = np.zeros((n, 6), dtype=np.float32)
    # origins at subnormal / tiny negative x (inside cell 0, a hair from the plane x = 0), directions mostly +x
    rays[:, 0] = -(rng.integers(1, 2000, n).astype(np.uint32).view(np.float32))       # -k * 2^-149
    rays[: n // 2, 0] = -(rng.uniform(1.0, 2.0, n // 2) * 2.0 ** rng.integers(-140, -120, n // 2)).astype(np.float32)
    rays[:, 1:3] = rng.uniform(-0.05, 0.05, (n, 2)).astype(np.float32)
    d = np.stack([np.ones(n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], axis=1)
    rays[:, 3:] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    start = np.zeros(n, dtype=np.uint32)
    want = O.trace_forward(0, pts, att, adj, off, rays, start)
    with O.scan_mode("filtered") as mode:
        got = O.trace_forward(0, pts, att, adj, off, rays, start)
        fired = mode.contested
    assert np.array_equal(got["rgba"].view(np.uint32), want["rgba"].view(np.uint32))
    assert np.array_equal(got["num_intersections"], want["num_intersections"])
    assert fired > 0
    assert float(want["rgba"][:, 3].max()) > 0          # the rays do cross into the neighbour and composite there
    # whole foams scaled down: 2^-12 (fp16 offsets still normal), 2^-17 (subnormal fp16 offsets: heavy ties), 2^-60 (every
    # offset underflows to zero: no exits anywhere, as in the reference)
    from radfoam_amd import foam
    fm = foam.make_synthetic_foam(3000, 1, 2)
    cam = foam.default_camera(48, 32)
    r0 = foam.camera_rays(cam).reshape(-1, 6)
    for e in (-12, -17, -60):
        s = np.float32(2.0 ** e)
        p = fm["points"] * s
        r = r0.copy()
        r[:, :3] *= s
        st = np.full(r.shape[0], foam.nearest_point(fm["points"], cam["position"]), dtype=np.uint32)
        a = fm["attributes"].copy()
        a[:, -1] = np.minimum(a[:, -1] / s, np.float32(3e38))                         # optical depths as before the scaling
        args = (1, p, a, fm["point_adjacency"], fm["point_adjacency_offsets"], r, st)
        want = O.trace_forward(*args)
        with O.scan_mode("filtered"):
            got = O.trace_forward(*args)
        assert np.array_equal(got["rgba"].view(np.uint32), want["rgba"].view(np.uint32)), e
        assert np.array_equal(got["num_intersections"], want["num_intersections"]), e
