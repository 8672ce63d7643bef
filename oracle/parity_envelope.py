"""Parity against the reference SOURCE at the BASELINE configurations.  TEST INFRASTRUCTURE ONLY.

The north star states its tolerance (1e-4 RGB, 1e-3 gradients) against the reference renderer.  What can run here is
the reference's kernel source text compiled for the CPU (oracle/_ref/libref.so, oracle/Makefile.ref) -- and that text
does not define its fp32 results to the last bit: nvcc contracts a*b+c into FMAs by default, this build does not.  So
the same source is compiled twice (libref.so: no contraction; libref_fma.so: -ffp-contract=fast -mfma, every a*b+c the
compiler sees fused) and three evaluations are compared on every STRIDE-th row and column of a BASELINE frame:

    oracle   oracle/rf_oracle.c: the reference's scan (every face's rounded quotient, tracing_utils.cuh:43-67) in the
             pinned fp32 arithmetic == the HIP kernels, bit for bit (tests/test_gpu_parity.py, bench.py's whole-frame
             check; the kernels' filtered evaluation of that scan is the same function, tests/test_oracle.py)
    ref      the reference source, no contraction
    ref_fma  the reference source, contracted

For each pair: rays whose walk takes another path (different num_intersections -- a near-tie between two exit faces
decided the other way: a cell inserted or skipped), rays with |d rgba| above 1e-4 / 1e-5, and the relative L2 distance
of points_grad / attr_grad: overall, and split by linearity (a gradient is a sum over rays) into the part carried by
the rays of the pair that take the same path and the part carried by the flipped rays (a second backward over just
those).  The pair (ref_fma, ref) is the reference's OWN envelope: what its two legitimate builds disagree about.

What the numbers say (profiles/r05/parity_baseline_scale.json): 0.05-0.2 % of the rays pass a Voronoi edge so closely
that the two candidate exits agree to an ulp, and any change of rounding decides them the other way.  Such a ray keeps
its colour (1e-5 typically, 2e-4 at worst) but deposits its point gradient in other cells, so the OVERALL gradient
distance is the gradient of the one or two heaviest flipped rays (per-ray norms are heavy-tailed: median 10, max 85 on
config 2, against 8.3e3 for the whole sample) -- 1.2e-3 between the reference's own two builds, above the north
star's 1e-3 by itself.  A factor between two such figures compares two outliers, not two implementations.  The bar
(tests/test_reference_source.py) is therefore: no more flipped rays than 1.5x the reference's own; no more rays
beyond 1e-4 in rgba than 1.5x its own (floor 2), none beyond 3e-4; on the rays that take the same path, gradients
within 2e-4 of the nearer of the two builds (the oracle spells its FMAs out, so it sits with the contracted build: 3e-7
on attr_grad where the uncontracted build is 1.9e-3 away from both); overall gradients within 2x the reference's own
distance, and -- the north star's words taken literally -- at least 99.99 % of the rays within 1e-4 in rgba and as
large a share of the gradient elements within 1e-3 as between the reference's own builds (check() below has the list).
(Rounds 1-4 shipped a cross-multiplied tournament WITHOUT a certificate as the default scan -- "canonical" -- and
allowed it 3x; it is gone, and so is the allowance.)

    python -m oracle.parity_envelope [c2] [north-star] [--stride 6] [--out profiles/r05/parity_baseline_scale.json]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

CONFIGS = {
    # name: (points, sh degree, foam seed)   -- BASELINE.json configs[1] and the north-star point
    "c2": (500_000, 2, 1),
    "north-star": (2_000_000, 2, 5),
}


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _elements_within(got, ref, rtol=1e-3):
    """The north star's gradient tolerance taken literally, element by element (the bound of tests/helpers.grad_close:
    rtol*|ref| + rtol*rms(ref)): (elements either side touches, those of them inside the bound)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    touched = (got != 0) | (ref != 0)
    nz = ref[ref != 0]
    rms = float(np.sqrt(np.mean(nz ** 2))) if nz.size else 0.0
    inside = np.abs(got - ref) <= rtol * np.abs(ref) + rtol * rms
    return int(touched.sum()), int((inside & touched).sum())


def _pair(fa, fb, ba, bb, flipped_a, flipped_b):
    d = np.abs(np.asarray(fa["rgba"], np.float64) - np.asarray(fb["rgba"], np.float64)).max(axis=-1)
    rec = {
        "rays_on_another_path": int((fa["num_intersections"] != fb["num_intersections"]).sum()),
        "rays_drgba_gt_1e-4": int((d > 1e-4).sum()),
        "rays_drgba_gt_1e-5": int((d > 1e-5).sum()),
        "max_drgba": float(d.max()),
        # the north star's colour tolerance, literally: rays whose four channels all agree to 1e-4
        "rays": int(d.size),
        "frac_rays_within_1e-4_rgba": float((d <= 1e-4).mean()),
    }
    for k in ("points_grad", "attr_grad"):
        ref = np.asarray(bb[k], np.float64)
        total = np.asarray(ba[k], np.float64) - ref
        flipped = np.asarray(flipped_a[k], np.float64) - np.asarray(flipped_b[k], np.float64)
        scale = max(np.linalg.norm(ref), 1e-300)
        rec[k + "_rel_l2"] = float(np.linalg.norm(total) / scale)
        rec["same_path_" + k + "_rel_l2"] = float(np.linalg.norm(total - flipped) / scale)   # linearity
        rec["flipped_rays_" + k + "_rel_l2"] = float(np.linalg.norm(flipped) / scale)
        # ... and its gradient tolerance, literally: per element, not a norm that two rays dominate
        n, ok = _elements_within(ba[k], bb[k])
        rec[k + "_elements_touched"] = n
        rec[k + "_frac_elements_within_1e-3"] = float(ok / max(n, 1))
    return rec


def measure(fm, sh_degree, width=1920, height=1080, stride=6, grad_seed=11):
    """The three evaluations on rows/columns 0, stride, 2*stride, ... of the width x height frame of the SURVEY 8(d)
    camera; returns the record described in the module docstring."""
    from oracle import oracle as O
    from oracle import refsrc as Rf
    from radfoam_amd import foam

    cam = foam.default_camera(width, height)
    rays = np.ascontiguousarray(foam.camera_rays(cam)[::stride, ::stride])
    start = np.uint32(foam.nearest_point(fm["points"], cam["position"]))
    rng = np.random.default_rng(grad_seed)
    g = rng.normal(size=rays.shape[:-1] + (4,)).astype(np.float32)
    args = (sh_degree, fm["points"], fm["attributes"], fm["point_adjacency"], fm["point_adjacency_offsets"])

    t = time.time()

    def run(name, r, gg, rgba=None):
        """(forward, backward) of evaluation `name` over rays r"""
        def both(fwd_fn, bwd_fn):
            f = fwd_fn(*args, r, start)
            return f, bwd_fn(*args, r, start, f["rgba"] if rgba is None else rgba, gg)
        if name in ("ref", "ref_fma"):
            with Rf.variant("fma" if name == "ref_fma" else "plain"):
                return both(Rf.trace_forward, Rf.trace_backward)
        return both(O.trace_forward, O.trace_backward)

    names = ["ref", "ref_fma", "oracle"]
    fwd, bwd = {}, {}
    for name in names:
        fwd[name], bwd[name] = run(name, rays, g)

    def pair(a, b):
        flipped = (fwd[a]["num_intersections"] != fwd[b]["num_intersections"]).reshape(-1)
        r, gg = rays.reshape(-1, 6)[flipped], g.reshape(-1, 4)[flipped]
        if r.shape[0]:
            _, fa = run(a, r, gg)
            _, fb = run(b, r, gg)
        else:
            fa = fb = {k: np.zeros_like(bwd[a][k]) for k in ("points_grad", "attr_grad")}
        return _pair(fwd[a], fwd[b], bwd[a], bwd[b], fa, fb)

    rec = {
        "points": int(fm["points"].shape[0]), "sh_degree": sh_degree, "frame": [height, width], "stride": stride,
        "rays": int(rays.shape[0] * rays.shape[1]),
        "mean_cells_per_ray": float(fwd["ref"]["num_intersections"].mean()),
        "oracle_vs_ref": pair("oracle", "ref"),
        "oracle_vs_ref_fma": pair("oracle", "ref_fma"),
        "ref_fma_vs_ref": pair("ref_fma", "ref"),
    }
    rec["seconds"] = round(time.time() - t, 1)
    return rec


def check(rec):
    """The bar (module docstring): the oracle is inside the reference's own envelope.  Returns the list of violations
    (empty = pass).

    Stated literally first: the share of rays whose rgba agrees with the reference source to the north star's 1e-4, and
    the share of gradient ELEMENTS inside its 1e-3 (tests/helpers.grad_close's bound) -- at least the reference's own
    share between its two builds, less one part in 10^4.  Then the norms: overall gradient distance to the nearer build
    at most 2x the reference's own distance between its builds."""
    bad = []
    own = rec["ref_fma_vs_ref"]
    pairs = ["oracle_vs_ref", "oracle_vs_ref_fma"]
    for other in pairs:
        o = rec[other]
        if o["rays_on_another_path"] > 1.5 * max(own["rays_on_another_path"], 8):
            bad.append((other, "rays_on_another_path", o["rays_on_another_path"], own["rays_on_another_path"]))
        if o["rays_drgba_gt_1e-4"] > 1.5 * max(own["rays_drgba_gt_1e-4"], 2):
            bad.append((other, "rays_drgba_gt_1e-4", o["rays_drgba_gt_1e-4"], own["rays_drgba_gt_1e-4"]))
        if not o["max_drgba"] < 3e-4:
            bad.append((other, "max_drgba", o["max_drgba"], 3e-4))
        if not o["frac_rays_within_1e-4_rgba"] >= 0.9999:
            bad.append((other, "frac_rays_within_1e-4_rgba", o["frac_rays_within_1e-4_rgba"], 0.9999))
        for k in ("points_grad", "attr_grad"):
            key = k + "_frac_elements_within_1e-3"
            if not o[key] >= own[key] - 1e-4:
                bad.append((other, key, o[key], own[key]))
    for k in ("points_grad", "attr_grad"):
        same = min(rec["oracle_vs_ref"]["same_path_" + k + "_rel_l2"],
                   rec["oracle_vs_ref_fma"]["same_path_" + k + "_rel_l2"])
        if not same < 2e-4:
            bad.append(("oracle_vs_nearer_build", "same_path_" + k + "_rel_l2", same, 2e-4))
        floor = max(own[k + "_rel_l2"], 1e-4)    # attr_grad: the two builds agree to 1e-5 on some frames
        overall = min(rec["oracle_vs_ref"][k + "_rel_l2"], rec["oracle_vs_ref_fma"][k + "_rel_l2"])
        if not overall <= 2.0 * floor:
            bad.append(("oracle_vs_nearer_build", k + "_rel_l2", overall, 2.0 * floor))
    return bad


def check_frames(records):
    """Across frames (a list of measure() records): (the oracle's overall points_grad distance to the nearer build) / (the
    reference's own distance between its builds) per frame -- one frame's ratio compares two outliers, so the geometric
    mean over frames is held to 1.0: the scan is not systematically further from the reference than the reference is
    from itself.  Returns (ratios, violations)."""
    ratios = []
    for rec in records:
        own = max(rec["ref_fma_vs_ref"]["points_grad_rel_l2"], 1e-4)
        ratios.append(min(rec["oracle_vs_ref"]["points_grad_rel_l2"], rec["oracle_vs_ref_fma"]["points_grad_rel_l2"]) / own)
    gm = float(np.exp(np.mean(np.log(np.maximum(ratios, 1e-12))))) if ratios else 0.0
    return ratios, ([("oracle_over_own_geomean", gm, 1.0)] if gm > 1.0 else [])


def load_foam(name, build_if_missing=True):
    from radfoam_amd import foam

    n, d, seed = CONFIGS[name]
    path = os.path.join(foam.default_cache_dir(), f"foam_n{n}_s{seed}.npz")
    if not os.path.exists(path) and not build_if_missing:
        return None, d
    return foam.make_synthetic_foam(n, d, seed, cache_dir=foam.default_cache_dir()), d


def main(argv):
    names = [a for a in argv if a in CONFIGS] or list(CONFIGS)
    stride = int(argv[argv.index("--stride") + 1]) if "--stride" in argv else 6
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    result = {}
    for name in names:
        fm, d = load_foam(name)
        rec = measure(fm, d, stride=stride)
        rec["violations"] = check(rec)
        result[name] = rec
        print(name, json.dumps(rec, indent=1))
    ratios, bad = check_frames(list(result.values()))
    print("oracle / reference's own overall points_grad distance per frame:", [round(r, 3) for r in ratios], bad)
    if out:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            json.dump(result, f, indent=1)
    return 0 if all(not r["violations"] for r in result.values()) and not bad else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
