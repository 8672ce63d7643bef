#!/usr/bin/env python
"""Generates tests/golden/*.npz: inputs + outputs of the REFERENCE's own tracer kernels.

The outputs come from oracle/_ref/libref.so, i.e. the source text of
/root/reference/src/tracing/{pipeline.cu kernels, tracing_utils.cuh, sh_utils.cuh, camera.h}
compiled for the CPU against oracle/ref_shim (no Eigen / CUDA here; see oracle/ref_driver.cpp for
what that does and does not pin).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

The fixtures are small (a few thousand points, a few hundred rays) so that they can live in git.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import refsrc as Rf  # noqa: E402
from radfoam_amd import foam  # noqa: E402
from tests import helpers as H  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, points, sh degree, half attrs, image (w,h) or 0 for random rays, quantiles, density scale, camera inside
    ("d0_image", 1500, 0, False, (24, 16), False, 1.0, False),
    ("d1_rays_quantiles", 1500, 1, False, 0, True, 1.0, False),
    ("d2_image_quantiles", 2000, 2, False, (24, 16), True, 1.0, False),
    ("d3_rays", 1500, 3, False, 0, False, 1.0, False),
    ("d2_dense_inside", 2000, 2, False, (16, 12), True, 10.0, True),   # saturating rays, first-cell quirk
    ("d3_half_image", 1500, 3, True, (24, 16), False, 1.0, False),
    ("d0_half_rays", 1500, 0, True, 0, False, 1.0, False),
]


def main():
    Rf.build()
    for name, n, d, half, image, quant, dens, inside in CASES:
        seed = abs(hash(name)) % 1000 if False else sum(map(ord, name))
        fm = foam.make_synthetic_foam(n, d, seed)
        attrs = fm["attributes"].copy()
        attrs[:, -1] *= dens
        if half:
            attrs = attrs.astype(np.float16)
        rng = np.random.default_rng(seed + 1)
        if image:
            pos = (0.03, -0.02, 0.05) if inside else (0.0, 0.0, -3.0)
            cam, rays, start = H.camera_setup(fm, image[0], image[1], position=pos)
            starts = np.full(rays.shape[:-1], start, dtype=np.uint32)
        else:
            cam = None
            rays, starts = H.random_rays(fm, 300, seed + 2)
        batch = rays.shape[:-1]
        adt = np.float16 if half else np.float32
        q = dg = None
        if quant:
            q = np.sort(rng.uniform(0.02, 0.98, size=batch + (2,)).astype(np.float32), axis=-1)[..., ::-1].copy()
            dg = rng.normal(size=batch + (2,)).astype(np.float32)
        g = rng.normal(size=batch + (4,)).astype(adt)
        err = rng.uniform(0, 1, size=batch).astype(adt)
        foam_args = (d, fm["points"], attrs, fm["point_adjacency"], fm["point_adjacency_offsets"])
        fwd = Rf.trace_forward(*foam_args, rays, starts, depth_quantiles=q, return_contribution=True)
        bwd = Rf.trace_backward(*foam_args, rays, starts, fwd["rgba"], g, depth_quantiles=q,
                                depth_indices=fwd.get("depth_indices"), depth_grad_in=dg, ray_error=err)
        out = dict(
            sh_degree=d, points=fm["points"], attributes=attrs, point_adjacency=fm["point_adjacency"],
            point_adjacency_offsets=fm["point_adjacency_offsets"], rays=rays, start_point=starts,
            grad_rgba=g, ray_error=err,
            ref_rgba=fwd["rgba"], ref_num_intersections=fwd["num_intersections"],
            ref_contribution=fwd["contribution"], ref_points_grad=bwd["points_grad"],
            ref_attr_grad=bwd["attr_grad"], ref_point_error=bwd["point_error"],
            ref_adjacent_diff=Rf.build_adjacent_diff(fm["points"], fm["point_adjacency"], fm["point_adjacency_offsets"]),
        )
        if quant:
            out.update(depth_quantiles=q, depth_grad=dg, ref_depth=fwd["depth"], ref_depth_indices=fwd["depth_indices"])
        if cam is not None:
            diff = out["ref_adjacent_diff"]
            out["ref_benchmark_rgba8"] = Rf.trace_benchmark(*foam_args, diff, cam, start, weight_threshold=0.05)
            out["camera_position"] = np.asarray(cam["position"], dtype=np.float32)
            out["camera_fov"] = np.float32(cam["fov"])
            out["camera_wh"] = np.array([cam["width"], cam["height"]], dtype=np.int32)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB; mean steps", float(fwd["num_intersections"].mean()),
              "max alpha", float(np.asarray(fwd["rgba"], dtype=np.float32)[..., 3].max()))


if __name__ == "__main__":
    main()
