"""``radfoam.Triangulation`` / ``build_aabb_tree`` on the GPU.

Reference: torch_bindings/triangulation_bindings.cpp:15-140,225-237 over src/delaunay/delaunay.cu:230-396 and
src/aabb_tree/aabb_tree.cu.  The reference grows one global tetrahedral mesh (growth iterations over sorted tet /
face tables); what its callers consume is ``permutation()``, ``point_adjacency()`` and
``point_adjacency_offsets()`` (radfoam_model/scene.py:65-72,160-200).  Here those come from
``rf_kd_order`` -> ``rf_build_aabb_tree`` -> ``rf_delaunay_adjacency`` (radfoam_amd/csrc/rf_delaunay.hip,
rf_star.hpp): every point builds its own Delaunay star with exact predicates, certified against the tree.

Limits (rf_star.hpp): at most 4095 Delaunay neighbours per point and 64 points with more than 249 (beyond that a
RuntimeError, not the retryable TriangulationFailedError); the exact predicates snap coordinates more than 2^38 times
finer than the coarsest coordinate of the same predicate to its grid -- inputs like that are reported as an ambiguous
triangulation (unmatched edges), not triangulated wrongly.

No CPU path in this module: it needs the HIP library and CUDA (HIP) tensors.  The tetrahedra themselves
(``tets()``, ``tet_adjacency()``, ``vert_to_tet()`` -- only the reference's viewer reads them) are not produced
on the GPU; those three getters triangulate once more with Qhull on the host when asked.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


class TriangulationFailedError(RuntimeError):
    """radfoam::TriangulationFailedError (src/delaunay/delaunay.h:9-13; registered at
    triangulation_bindings.cpp:222).  RadFoamScene.update_triangulation catches it, perturbs the points and
    retries (radfoam_model/scene.py:160-186)."""


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def pow2_round_up(x: int) -> int:
    return 1 if x <= 1 else 1 << ((x - 1).bit_length())


def _check_points(points: torch.Tensor):
    if points.dim() != 2 or points.size(-1) != 3:
        raise RuntimeError("points must have shape [N, 3]")
    if not points.is_cuda or points.dtype != torch.float32:
        raise RuntimeError("points must be a float32 CUDA tensor")


def kd_order(points: torch.Tensor):
    """(permutation uint32[N], points[permutation]) in the reference's kd-order (sort_points,
    src/aabb_tree/aabb_tree.cu:62-190)."""
    _check_points(points)
    p = points.detach().contiguous()
    n, dev = p.size(0), p.device
    lib = _lib.load()
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    out = torch.empty_like(p)
    ws = torch.empty(max(int(lib.rf_kd_order_workspace_bytes(n)), 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rf_kd_order(_ptr(p), n, _ptr(perm), _ptr(out), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc)
    return perm.view(torch.uint32), out


def build_aabb_tree(points: torch.Tensor) -> torch.Tensor:
    """``radfoam.build_aabb_tree`` (triangulation_bindings.cpp:117-140): [pow2_round_up(N), 2, 3] float32, the
    reference's node layout, bit-identical boxes.  The points must already be in kd-order."""
    if points.size(-1) != 3:
        raise RuntimeError("points must have 3 as the last dimension")
    if points.dim() != 2:
        raise RuntimeError("points must have 2 dimensions")
    _check_points(points)
    p = points.detach().contiguous()
    n, dev = p.size(0), p.device
    tree = torch.zeros((pow2_round_up(n), 2, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().rf_build_aabb_tree(_ptr(p), n, _ptr(tree), _stream(dev))
    _lib.check(rc)
    return tree


def delaunay_adjacency(points: torch.Tensor, tree: torch.Tensor | None = None, seed=None):
    """(point_adjacency uint32[E], point_adjacency_offsets uint32[N+1], info) of the Delaunay triangulation of
    ``points`` (kd-ordered, N >= 32).  ``seed`` = (adjacency, offsets) of an earlier triangulation of the same
    number of points makes the search cheaper (incremental rebuild) without changing the result.  Raises
    TriangulationFailedError for duplicate points and for cospherical neighbourhoods the stars disagree on."""
    _check_points(points)
    p = points.detach().contiguous()
    n, dev = p.size(0), p.device
    if n < 32:
        raise RuntimeError("Delaunay triangulation does not support less than 32 points")
    lib = _lib.load()
    if tree is None:
        tree = build_aabb_tree(p)
    seed_adj = seed_off = None
    if seed is not None:
        seed_adj, seed_off = seed[0].contiguous(), seed[1].contiguous()
        if seed_off.numel() != n + 1 or seed_adj.dtype != torch.uint32 or seed_off.dtype != torch.uint32:
            raise RuntimeError("seed lists must be uint32 tensors of a triangulation of the same number of points")
        if seed_adj.numel() == 0:   # nothing to seed from (an empty tensor has no address to hand over): from scratch
            seed_adj = seed_off = None
    ws = torch.empty(max(int(lib.rf_delaunay_workspace_bytes(n)), 256), dtype=torch.uint8, device=dev)
    capacity = 20 * n   # the reference gives up beyond 20 tetrahedra per point (delaunay.cu:352); E ~ 15.5 N
    info = (C.c_uint32 * 12)()
    while True:
        adj = torch.empty(capacity, dtype=torch.int32, device=dev)
        off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.rf_delaunay_adjacency(_ptr(p), n, _ptr(tree), _ptr(seed_adj), _ptr(seed_off), _ptr(adj), capacity,
                                           _ptr(off), info, _ptr(ws), ws.numel(), _stream(dev))
        if rc == _lib.RF_ERR_WORKSPACE and info[2] > 0:
            # more rim stars than the default workspace has rows for (a cloud that is mostly surface): come back larger
            need = int(lib.rf_delaunay_workspace_bytes_for(n, int(info[2])))
            if need > ws.numel():
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                continue
        _lib.check(rc)
        if info[0] <= capacity:
            break
        capacity = int(info[0])
    stats = dict(adjacency_size=int(info[0]), failed_stars=int(info[1]), large_stars=int(info[2]),
                 duplicate_points=int(info[3]), asymmetric_edges=int(info[4]),
                 tree_nodes_visited=int(info[5]) | (int(info[6]) << 32), insertions=int(info[7]),
                 degenerate_stars=int(info[8]), broken_stars=int(info[9]), oversized_stars=int(info[10]),
                 hull_candidates=int(info[11]))
    if stats["duplicate_points"]:
        raise TriangulationFailedError("duplicate points found")
    if stats["oversized_stars"]:
        # Not a TriangulationFailedError: perturbing the points (what RadFoamScene.update_triangulation does about
        # that one, scene.py:160-186) cannot shrink a hub, so a retry loop would only end in "aborted triangulation
        # after 25 attempts".  The reference's global mesh has no per-vertex bound; this implementation does.
        raise RuntimeError(
            "Delaunay triangulation: %d point(s) have more than 4095 Delaunay neighbours or more than 64 points have "
            "more than 249 (a point inside a large empty shell of points); this build keeps at most 4095 neighbours "
            "per point and 64 such hubs per triangulation -- remove or move those points" % stats["oversized_stars"])
    if stats["failed_stars"] or stats["asymmetric_edges"]:
        raise TriangulationFailedError(
            "ambiguous triangulation (%d stars failed: %d without a non-coplanar start, %d inconsistent, %d with more "
            "than 4095 neighbours; %d unmatched edges)" % (stats["failed_stars"], stats["degenerate_stars"],
                                                          stats["broken_stars"], stats["oversized_stars"],
                                                          stats["asymmetric_edges"]))
    e = stats["adjacency_size"]
    return adj[:e].clone().view(torch.uint32), off.view(torch.uint32), stats


class Triangulation:
    """The reference's protocol (triangulation_bindings.cpp:225-237): construction puts the points in kd-order and
    triangulates them in that order; ``permutation()`` tells the caller how to reorder its own arrays;
    ``rebuild(points, incremental)`` expects points already in that order and returns whether a new permutation
    has to be applied (the reference returns ``sorted``: true whenever it re-sorted, delaunay.cu:368)."""

    def __init__(self, points: torch.Tensor):
        _check_points(points)
        self._n = -1
        self._adjacency = self._offsets = None
        self._host = None
        self.stats = {}
        self.rebuild(points, incremental=False)

    def rebuild(self, points: torch.Tensor, incremental: bool = False) -> bool:
        _check_points(points)
        if points.size(0) < 32:
            raise RuntimeError("Delaunay triangulation does not support less than 32 points")
        pts = points.detach().contiguous()
        if not bool(torch.isfinite(pts).all()):
            raise TriangulationFailedError("points contain non-finite values")
        if incremental and pts.size(0) == self._n and self._adjacency is not None:
            # delaunay.cu:293-311: no re-sort; the previous mesh is repaired.  Here: the previous neighbour lists
            # are the first candidates of every star.
            sorted_pts, seed, needs_permute = pts, (self._adjacency, self._offsets), False
        else:
            self._perm, sorted_pts = kd_order(pts)
            seed, needs_permute = None, True
        tree = build_aabb_tree(sorted_pts)
        adj, off, self.stats = delaunay_adjacency(sorted_pts, tree, seed)
        self._n = pts.size(0)
        self._points = sorted_pts.clone() if sorted_pts is pts else sorted_pts   # not the caller's (mutable) storage
        self._adjacency, self._offsets = adj, off
        self._host = None
        # whatever the tracer packed from the old lists is stale, even if a caller hands the new ones out at the old
        # addresses (the reference's from_blob getters do)
        from .pipeline import invalidate_caches
        invalidate_caches()
        return needs_permute

    def permutation(self):
        return self._perm

    def point_adjacency(self):
        return self._adjacency

    def point_adjacency_offsets(self):
        return self._offsets

    # the tetrahedra: not on the hot path (viewer only); Qhull on the host, on demand
    def _host_mesh(self):
        if self._host is None:
            from scipy.spatial import Delaunay
            self._host = Delaunay(self._points.cpu().numpy().astype(np.float64))
        return self._host

    def tets(self):
        return torch.from_numpy(self._host_mesh().simplices.astype(np.uint32)).to(self._points.device)

    def tet_adjacency(self):
        t = self._host_mesh().neighbors.astype(np.int64).astype(np.uint32)
        return torch.from_numpy(np.ascontiguousarray(t)).to(self._points.device)

    def vert_to_tet(self):
        t = self._host_mesh().vertex_to_simplex.astype(np.uint32)
        return torch.from_numpy(np.ascontiguousarray(t)).to(self._points.device)
