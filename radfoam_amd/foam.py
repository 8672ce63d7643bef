"""Synthetic foam fixtures (CPU, numpy/scipy) for tests and bench.

The hot path consumes a foam as four arrays (reference: radfoam_model/scene.py:208-217):
``points[N,3] f32``, ``attributes[N,A] f32|f16`` (row = ``[3B SH colour | density]``),
``point_adjacency[E] u32`` and ``point_adjacency_offsets[N+1] u32`` (CSR of the Delaunay
graph, both directions present).  The reference builds the CSR on the GPU
(src/delaunay/delaunay.cu:273-370); the north star keeps that build on the CPU, so here it
comes from Qhull (``scipy.spatial.Delaunay``).  Points are put in the reference's kd-order
(src/aabb_tree/aabb_tree.cu:62-190) first, which is what gives neighbouring cells
neighbouring indices.

Nothing in this file touches the GPU; it is input generation only.
"""
from __future__ import annotations

import math
import os

import numpy as np

SH_DIM = {0: 1, 1: 4, 2: 9, 3: 16}


def attribute_dim(sh_degree: int) -> int:
    """A = 1 + 3*(d+1)^2 (reference: src/tracing/pipeline.cu:768-770)."""
    return 1 + 3 * (sh_degree + 1) ** 2


def pow2_round_up(x: int) -> int:
    return 1 if x <= 1 else 1 << ((x - 1).bit_length())


def kd_order(points: np.ndarray) -> np.ndarray:
    """Permutation that puts ``points`` in the reference's kd-order.

    Mirrors the semantics of sort_points (src/aabb_tree/aabb_tree.cu:62-190): with
    P = pow2_round_up(N), sort everything by x, then every consecutive P/2 segment by y,
    every P/4 segment by z, ... cycling the axis, down to segments of 2.
    """
    n = points.shape[0]
    perm = np.arange(n, dtype=np.int64)
    seg = pow2_round_up(n)
    dim = 0
    idx = np.arange(n, dtype=np.int64)
    while seg > 1:
        seg_id = idx // seg
        key = points[perm, dim]
        order = np.lexsort((key, seg_id))
        perm = perm[order]
        seg >>= 1
        dim = (dim + 1) % 3
    return perm


def delaunay_csr(points: np.ndarray):
    """Delaunay neighbour graph as CSR (offsets u32[N+1], adjacency u32[E]).

    Neighbour lists are sorted ascending (what the reference's stable merge sort of edge
    pairs yields, src/delaunay/delaunay.cu:190-205).
    """
    from scipy.spatial import Delaunay

    tri = Delaunay(points.astype(np.float64))
    indptr, indices = tri.vertex_neighbor_vertices
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    n = points.shape[0]
    if indptr.shape[0] != n + 1:
        raise RuntimeError("Qhull dropped points (degenerate input)")
    # sort each row ascending: sort by (row, value)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    order = np.lexsort((indices, rows))
    indices = indices[order]
    return indptr.astype(np.uint32), indices.astype(np.uint32)


def nearest_point(points: np.ndarray, query: np.ndarray) -> int:
    """Exact nearest neighbour (entry cell of a camera; reference: radfoam.nn)."""
    d = points.astype(np.float64) - np.asarray(query, dtype=np.float64)[None, :]
    return int(np.argmin(np.einsum("ij,ij->i", d, d)))


def make_synthetic_foam(n_points: int, sh_degree: int, seed: int, *, shell_radius: float = 0.8,
                        optical_depth_per_cell: float = 0.15, cache_dir: str | None = None, triangulate=None):
    """Seeded synthetic foam, SURVEY.md section 8(d) recipe.

    points ~ U[-1,1]^3 in kd-order; density (already activated) = sigma0*U[0.5,1.5] with
    sigma0 = optical_depth_per_cell*1.455*(N/8)^(1/3), and exactly 0 outside
    ``shell_radius`` (empty shell, so the unbounded hull cells do not saturate alpha);
    SH coefficients ~ N(0, 0.3^2).  Returns a dict of numpy arrays (all float32 / uint32).

    ``cache_dir``: if given, the slow Qhull step (kd-ordered points + CSR, which do not
    depend on sh_degree) is cached there as an .npz.
    ``triangulate``: optional ``points -> (kd-ordered points, offsets, adjacency)`` used instead of
    kd_order + Qhull when nothing is cached (bench.py passes the GPU triangulation, whose lists the GPU
    tests hold equal to Qhull's on these very foams); the result is then not written to the cache, which
    stays a Qhull artefact.  ``foam["csr_source"]`` says where the lists came from.
    """
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.0, 1.0, size=(n_points, 3)).astype(np.float32)
    path = None if cache_dir is None else os.path.join(cache_dir, f"foam_n{n_points}_s{seed}.npz")
    source = "qhull"
    if path is not None and os.path.exists(path):
        z = np.load(path)
        pts, offsets, adjacency = z["points"], z["offsets"], z["adjacency"]
        source = "qhull (cached)"
    elif triangulate is not None:
        pts, offsets, adjacency = triangulate(pts)
        source = "gpu"
    else:
        pts = np.ascontiguousarray(pts[kd_order(pts)])
        offsets, adjacency = delaunay_csr(pts)
        if path is not None:
            try:
                os.makedirs(cache_dir, exist_ok=True)
                # compressed: the caches travel with every gpurun snapshot (512 MiB limit); zlib takes the CSR of
                # a kd-ordered foam to under a third
                np.savez_compressed(path, points=pts, offsets=offsets, adjacency=adjacency)
            except OSError:
                pass
    a = attribute_dim(sh_degree)
    attrs = np.empty((n_points, a), dtype=np.float32)
    attrs[:, : a - 1] = rng.normal(0.0, 0.3, size=(n_points, a - 1)).astype(np.float32)
    sigma0 = optical_depth_per_cell * 1.455 * (n_points / 8.0) ** (1.0 / 3.0)
    dens = (sigma0 * rng.uniform(0.5, 1.5, size=n_points)).astype(np.float32)
    dens[np.linalg.norm(pts.astype(np.float64), axis=1) > shell_radius] = 0.0
    attrs[:, a - 1] = dens
    return {
        "points": pts,
        "attributes": attrs,
        "point_adjacency": adjacency,
        "point_adjacency_offsets": offsets,
        "sh_degree": sh_degree,
        "seed": seed,
        "csr_source": source,
    }


def default_cache_dir() -> str:
    """Git-ignored foam cache that still travels with a gpurun snapshot."""
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".foam_cache")


def default_camera(width: int, height: int):
    """Pinhole camera of SURVEY.md 8(d): at (0,0,-3) looking +z, up +y, vfov 2*atan(0.8/3)."""
    return {
        "position": np.array([0.0, 0.0, -3.0], dtype=np.float32),
        "forward": np.array([0.0, 0.0, 1.0], dtype=np.float32),
        "right": np.array([1.0, 0.0, 0.0], dtype=np.float32),
        "up": np.array([0.0, 1.0, 0.0], dtype=np.float32),
        "fov": float(2.0 * math.atan(0.8 / 3.0)),
        "width": int(width),
        "height": int(height),
        "model": "pinhole",
    }


def camera_rays(camera: dict) -> np.ndarray:
    """Rays [H,W,6] f32 with the reference's pixel convention (src/tracing/camera.h:56-85):
    x=i/W, y=j/H, u=(2x-1)*W/H, v=1-2y, dir=normalize(fwd/tan(fov/2)+u*right+v*up).

    Input generation only (both the HIP path and the oracle consume the same array), so
    float32 rounding details of this function do not matter for parity.
    """
    w, h = camera["width"], camera["height"]
    i = np.arange(w, dtype=np.float32)[None, :]
    j = np.arange(h, dtype=np.float32)[:, None]
    x = i / np.float32(w)
    y = j / np.float32(h)
    aspect = np.float32(w) / np.float32(h)
    u = (np.float32(2.0) * x - np.float32(1.0)) * aspect
    v = np.float32(1.0) - np.float32(2.0) * y
    if camera.get("model", "pinhole") != "pinhole":
        raise ValueError("camera_rays only generates pinhole rays")
    wgt = np.float32(1.0 / math.tan(camera["fov"] * 0.5))
    f = np.asarray(camera["forward"], dtype=np.float32)
    r = np.asarray(camera["right"], dtype=np.float32)
    up = np.asarray(camera["up"], dtype=np.float32)
    d = (wgt * f)[None, None, :] + u[..., None] * r[None, None, :] + v[..., None] * up[None, None, :]
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.empty((h, w, 6), dtype=np.float32)
    rays[..., :3] = np.asarray(camera["position"], dtype=np.float32)[None, None, :]
    rays[..., 3:] = d.astype(np.float32)
    return rays


def to_reference_pt_dict(foam: dict) -> dict:
    """Foam in the reference checkpoint layout (radfoam_model/scene.py:614-630).

    ``density`` there is the raw pre-softplus parameter; synthetic foams carry activated
    densities, so the raw value is recovered with the inverse of
    activation_scale*softplus(x, beta=10) for activation_scale=1.
    """
    import torch

    a = foam["attributes"].astype(np.float64)
    dens = a[:, -1]
    raw = np.where(dens > 0, np.log(np.expm1(np.minimum(dens * 10.0, 700.0))) / 10.0, -10.0)
    return {
        "xyz": torch.from_numpy(foam["points"].copy()),
        "density": torch.from_numpy(raw.astype(np.float32))[:, None],
        "color_dc": torch.from_numpy(foam["attributes"][:, :3].copy()),
        "color_sh": torch.from_numpy(foam["attributes"][:, 3:-1].copy()),
        "adjacency": torch.from_numpy(foam["point_adjacency"].astype(np.int64)),
        "adjacency_offsets": torch.from_numpy(foam["point_adjacency_offsets"].astype(np.int64)),
    }


if __name__ == "__main__":   # python -m radfoam_amd.foam N SEED: build and cache a foam's triangulation
    import sys
    import time

    _n, _seed = int(sys.argv[1]), int(sys.argv[2])
    _t = time.time()
    _fm = make_synthetic_foam(_n, 0, _seed, cache_dir=default_cache_dir())
    print(f"foam n={_n} seed={_seed}: {_fm['point_adjacency'].shape[0]} adjacency entries, "
          f"{time.time() - _t:.1f}s, cached under {default_cache_dir()}")
