"""CPU: the oracle pinned against the reference's OWN kernel source.

tests/golden/*.npz hold outputs of /root/reference's forward / backward / benchmark /
prefetch_adjacent_diff kernels, compiled for the CPU against oracle/ref_shim by
oracle/Makefile.ref and run by tests/golden/make_golden.py.  That build does every vector
operation as plain scalar C without FMA contraction and uses glibc's expf/logf, so it agrees
with any other faithful evaluation of the same formulas to a few ulp per operation, not bit for
bit.  Tolerances here: rgba/depth 1e-5 absolute (north star: 1e-4), integer outputs equal,
gradients within the north star's 1e-3: relative L2 < 1e-3 (observed: 2e-6 .. 2e-4, the larger
values on foams whose unbounded hull cells carry density gradients of +-4000 that cancel) and the
per-element bound of helpers.grad_close.
Where /root/reference is present the live library is exercised as well.
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def check_against_golden(z, fwd, bwd, diff, bench=None, half=False, grad_rel=1e-3):
    """Shared by the CPU (oracle) and GPU (HIP) golden tests."""
    atol = 2e-3 if half else 1e-5
    np.testing.assert_array_equal(fwd["num_intersections"], z["ref_num_intersections"])
    np.testing.assert_allclose(_f32(fwd["rgba"]), _f32(z["ref_rgba"]), rtol=0, atol=atol)
    if "ref_depth" in z:
        np.testing.assert_array_equal(fwd["depth_indices"], z["ref_depth_indices"])
        np.testing.assert_allclose(fwd["depth"], z["ref_depth"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(np.asarray(diff).view(np.uint16).reshape(-1, 4), z["ref_adjacent_diff"])
    if half:
        # fp16 scatter outputs: the reference rounds to half after every atomic add, the oracle and
        # the HIP path accumulate in fp32 and round once (documented deviation): loose comparison
        for key, ref in (("contribution", "ref_contribution"), ("attr_grad", "ref_attr_grad")):
            got, want = _f32(fwd[key] if key == "contribution" else bwd[key]), _f32(z[ref])
            assert np.linalg.norm(got - want) <= 0.05 * max(np.linalg.norm(want), 1e-6), key
    else:
        ok, rel, worst = H.grad_close(fwd["contribution"], z["ref_contribution"])
        assert ok and rel < grad_rel, ("contribution", rel, worst)
        for key in ("points_grad", "attr_grad", "point_error"):
            ok, rel, worst = H.grad_close(bwd[key], z["ref_" + key])
            assert ok and rel < grad_rel, (key, rel, worst)
    ok, rel, worst = H.grad_close(bwd["points_grad"], z["ref_points_grad"], rtol=5e-3 if half else 1e-3)
    assert ok, ("points_grad", rel, worst)
    if bench is not None:
        got = np.stack([(bench >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(np.int32)
        ref = np.stack([(z["ref_benchmark_rgba8"] >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(np.int32)
        assert np.abs(got - ref).max() <= 1 and (got != ref).mean() < 0.01


def golden_camera(z):
    from radfoam_amd import foam

    w, h = (int(v) for v in z["camera_wh"])
    cam = foam.default_camera(w, h)
    cam["position"] = z["camera_position"]
    cam["fov"] = float(z["camera_fov"])
    return cam


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_source_goldens(path):
    z = dict(np.load(path))
    d = int(z["sh_degree"])
    half = z["attributes"].dtype == np.float16
    args = (d, z["points"], z["attributes"], z["point_adjacency"], z["point_adjacency_offsets"])
    q = z.get("depth_quantiles")
    fwd = O.trace_forward(*args, z["rays"], z["start_point"], depth_quantiles=q, return_contribution=True,
                          num_threads=1)
    bwd = O.trace_backward(*args, z["rays"], z["start_point"], z["ref_rgba"], z["grad_rgba"], depth_quantiles=q,
                           depth_indices=z.get("ref_depth_indices"), depth_grad_in=z.get("depth_grad"),
                           ray_error=z["ray_error"], num_threads=1)
    diff = O.build_adjacent_diff(z["points"], z["point_adjacency"], z["point_adjacency_offsets"])
    bench = None
    if "ref_benchmark_rgba8" in z:
        bench = O.trace_benchmark(*args, diff, golden_camera(z), int(z["start_point"].reshape(-1)[0]),
                                  weight_threshold=0.05)
    check_against_golden(z, fwd, bwd, diff, bench, half)


def test_goldens_cover_the_path():
    assert len(GOLDEN) >= 6
    names = " ".join(GOLDEN)
    for must in ("d0", "d1", "d2", "d3", "half", "quantiles", "dense_inside"):
        assert must in names
    z = np.load([p for p in GOLDEN if "dense_inside" in p][0])
    assert float(np.asarray(z["ref_rgba"], np.float32)[..., 3].max()) > 0.999  # saturating rays are covered
    idx = np.concatenate([np.load(p)["ref_depth_indices"].reshape(-1) for p in GOLDEN if "quantiles" in p])
    assert (idx == 0xFFFFFFFF).any() and (idx != 0xFFFFFFFF).any()  # reached and unreached quantiles


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/tracing"), reason="reference sources not on this box")
def test_live_reference_source_agrees_with_oracle(foam_factory):
    """Fresh inputs (not in the goldens) through the live oracle/_ref library."""
    from oracle import refsrc as Rf

    Rf.build()
    d = 2
    fm = foam_factory(2500, d, 17)
    fm = dict(fm)
    fm["attributes"] = fm["attributes"].copy()
    fm["attributes"][:, -1] *= 8.0   # rays saturate: threshold exits and the dropped-tail quirk are exercised
    rays, starts = H.random_rays(fm, 600, 5)
    rng = np.random.default_rng(2)
    q = np.sort(rng.uniform(0.02, 0.98, size=(600, 3)).astype(np.float32), axis=-1)[:, ::-1].copy()
    g = rng.normal(size=(600, 4)).astype(np.float32)
    dg = rng.normal(size=(600, 3)).astype(np.float32)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    for thr, cap in ((1e-3, 1024), (0.05, 9)):
        ro = Rf.trace_forward(*args, rays, starts, depth_quantiles=q, weight_threshold=thr, max_intersections=cap)
        oo = O.trace_forward(*args, rays, starts, depth_quantiles=q, weight_threshold=thr, max_intersections=cap)
        np.testing.assert_array_equal(oo["num_intersections"], ro["num_intersections"])
        np.testing.assert_array_equal(oo["depth_indices"], ro["depth_indices"])
        np.testing.assert_allclose(oo["rgba"], ro["rgba"], rtol=0, atol=1e-5)
        rb = Rf.trace_backward(*args, rays, starts, ro["rgba"], g, depth_quantiles=q,
                               depth_indices=ro["depth_indices"], depth_grad_in=dg, weight_threshold=thr,
                               max_intersections=cap)
        ob = O.trace_backward(*args, rays, starts, ro["rgba"], g, depth_quantiles=q,
                              depth_indices=ro["depth_indices"], depth_grad_in=dg, weight_threshold=thr,
                              max_intersections=cap, num_threads=1)
        for k in ("points_grad", "attr_grad"):
            ok, rel, worst = H.grad_close(ob[k], rb[k])
            assert ok and rel < 1e-4, (k, rel, worst)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/tracing"), reason="reference sources not on this box")
def test_pinned_arithmetic_takes_the_reference_sources_paths_on_a_large_frame():
    """57,600 rays x ~60 cells x ~17 faces = 6e7 face tests of a 60k-point foam: the oracle evaluates the reference's
    scan (every face's rounded quotient, (P + o/2) - O) with its FMAs spelled out; how often does that pick another exit
    face than the reference's source text compiled without any contraction?  Only at exact-tie scale, from the FMAs
    alone: a handful of rays insert or skip a zero-length segment (measured: 7 rays), and rgba agrees to 1e-5
    everywhere.  Against the CONTRACTED build of the same source (what nvcc's default does) fewer still."""
    from oracle import refsrc as Rf
    from radfoam_amd import foam

    Rf.build()
    d = 2
    fm = foam.make_synthetic_foam(60000, d, 7)
    cam, rays, start = H.camera_setup(fm, 320, 180)
    args = (d, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])
    ro = Rf.trace_forward(*args, rays, start)
    oo = O.trace_forward(*args, rays, start)
    differ = int((ro["num_intersections"] != oo["num_intersections"]).sum())
    assert differ <= 15, differ                       # observed 7
    assert float(np.abs(ro["rgba"] - oo["rgba"]).max()) < 2e-5
    assert float(ro["rgba"][..., 3].max()) > 0.9 and float(ro["num_intersections"].mean()) > 40


_ENVELOPE_RECORDS = {}


def _envelope_case(name):
    from oracle import parity_envelope as PE
    from oracle import refsrc as Rf
    from radfoam_amd import foam

    if not Rf.available():
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref on this box")
    n, d, seed = PE.CONFIGS[name]
    cached = os.path.exists(os.path.join(foam.default_cache_dir(), f"foam_n{n}_s{seed}.npz"))
    if not cached and n > 500_000 and not os.environ.get("RF_TEST_LARGE"):
        pytest.skip(f"the {n}-point foam is not cached (Qhull takes minutes): python -m radfoam_amd.foam {n} {seed}")
    fm, d = PE.load_foam(name)
    return PE, PE.measure(fm, d)


@pytest.mark.parametrize("name", ["c2", "north-star"])
def test_oracle_is_inside_the_references_own_envelope_at_baseline_scale(name):
    """every 6th row and column of the BASELINE frames (57,600 rays; 500 k points and the 2 M
    north-star foam) through the reference SOURCE compiled without and with FMA contraction, and through the oracle
    (== the HIP kernels bit for bit), forward and backward.  The bar and the reasoning are in
    oracle/parity_envelope.py; the committed record is profiles/r05/parity_baseline_scale.json."""
    PE, rec = _envelope_case(name)
    assert PE.check(rec) == [], rec
    own, o, of = rec["ref_fma_vs_ref"], rec["oracle_vs_ref"], rec["oracle_vs_ref_fma"]
    # the sample is big enough to see the effect at all: the reference's two builds do disagree on some rays
    assert own["rays_on_another_path"] >= 10
    # what the oracle is closest to is the contracted build (it spells the same FMAs out): same-path gradients
    assert of["same_path_points_grad_rel_l2"] < 2e-4 and of["same_path_attr_grad_rel_l2"] < 1e-5
    # colours: at most a couple of tie rays leave the north star's 1e-4, as between the reference's own builds
    assert max(o["rays_drgba_gt_1e-4"], of["rays_drgba_gt_1e-4"]) <= 3
    # the north star's tolerances taken literally: share of rays / of gradient elements inside them
    for pair in (o, of):
        assert pair["frac_rays_within_1e-4_rgba"] >= 0.9999
        assert pair["points_grad_frac_elements_within_1e-3"] >= 0.9995 and pair["attr_grad_frac_elements_within_1e-3"] >= 0.9999
    # overall: within 2x the reference's own distance between its builds (PE.check), per frame ...
    assert min(o["points_grad_rel_l2"], of["points_grad_rel_l2"]) <= 2.0 * max(own["points_grad_rel_l2"], 1e-4)
    _ENVELOPE_RECORDS[name] = rec
    if len(_ENVELOPE_RECORDS) == 2:     # ... and not further than the reference from itself over the two frames together
        ratios, bad = PE.check_frames(list(_ENVELOPE_RECORDS.values()))
        assert bad == [], (ratios, bad)
