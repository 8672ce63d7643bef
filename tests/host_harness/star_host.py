"""ctypes loader of the HOST build of radfoam_amd/csrc/rf_star.hpp (test harness; see star_host.cpp) and a numpy
restatement of the reference's build_aabb_tree (src/aabb_tree/aabb_tree.cu:192-283) used as its checker."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstar_host.so")
_SRC = [os.path.join(_HERE, "star_host.cpp"), os.path.join(_HERE, "..", "..", "radfoam_amd", "csrc", "rf_star.hpp"),
        os.path.join(_HERE, "experiments", "star_owner.hpp")]


def build():
    if os.path.exists(_SO) and all(os.path.getmtime(s) <= os.path.getmtime(_SO) for s in _SRC):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-o", _SO,
                    _SRC[0]], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for name in ("star_host_exact_orient", "star_host_exact_insphere", "star_host_orient_sign",
                     "star_host_insphere_sign"):
            getattr(_lib, name).restype = C.c_int
            getattr(_lib, name).argtypes = [C.c_void_p]
        _lib.star_host_delaunay.restype = C.c_int
        _lib.star_host_delaunay.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.star_host_delaunay_owner.restype = C.c_int
        _lib.star_host_delaunay_owner.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def pow2_round_up(x: int) -> int:
    return 1 if x <= 1 else 1 << ((x - 1).bit_length())


def aabb_tree(points: np.ndarray) -> np.ndarray:
    """[pow2(N), 2, 3] float32: level d (2^d nodes) at node index 2^depth - 2^(d+1); the deepest stored level pairs
    the points, indices past N repeat the last point (aabb_tree.cu:206-219); the last entry is never written."""
    n = points.shape[0]
    p2 = pow2_round_up(n)
    depth = p2.bit_length() - 1
    pad = np.concatenate([points, np.repeat(points[-1:], p2 - n, axis=0)], axis=0).astype(np.float32)
    tree = np.zeros((p2, 2, 3), dtype=np.float32)
    lo = np.minimum(pad[0::2], pad[1::2])
    hi = np.maximum(pad[0::2], pad[1::2])
    for d in range(depth - 1, -1, -1):
        start = p2 - (1 << (d + 1))
        tree[start:start + (1 << d), 0] = lo
        tree[start:start + (1 << d), 1] = hi
        if d:
            lo = np.minimum(lo[0::2], lo[1::2])
            hi = np.maximum(hi[0::2], hi[1::2])
    return tree


def delaunay(points: np.ndarray, knn: int = 12, old=None, stride: int = 250, ghost_budget: int = 512,
             sweep: bool = False):
    """(offsets, adjacency, info) through the host build of the star code; sweep=False: every triangle certified by a
    query of its own (rf_star.hpp: star_sweep compiled in but switched off)."""
    lib().star_host_set_sweep(1 if sweep else 0)
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n = pts.shape[0]
    tree = aabb_tree(pts)
    depth = pow2_round_up(n).bit_length() - 1
    rows = np.zeros((n, stride), dtype=np.uint32)
    degree = np.zeros(n, dtype=np.uint32)
    hull = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.int32)
    visited = np.zeros(n, dtype=np.uint32)
    inserted = np.zeros(n, dtype=np.uint32)
    oa = oo = None
    if old is not None:
        oo = np.ascontiguousarray(old[0], dtype=np.uint32)
        oa = np.ascontiguousarray(old[1], dtype=np.uint32)
    bad = lib().star_host_delaunay(pts.ctypes.data, n, tree.ctypes.data, depth, knn,
                                   None if oa is None else oa.ctypes.data, None if oo is None else oo.ctypes.data,
                                   rows.ctypes.data, stride, degree.ctypes.data, hull.ctypes.data,
                                   status.ctypes.data, visited.ctypes.data, inserted.ctypes.data, ghost_budget)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(degree, out=offsets[1:])
    mask = np.arange(stride)[None, :] < degree[:, None]
    adjacency = rows[mask]
    return offsets.astype(np.uint32), adjacency, dict(bad=bad, status=status, visited=visited, inserted=inserted,
                                                       hull=hull, degree=degree)


def delaunay_owner(points: np.ndarray, tree_knn: int = 0, knn: int = 12, budget: int = 512, stride: int = 250,
                   old=None):
    """(offsets, adjacency, info) through the experimental two-pass build (every tetrahedron certified once)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n = pts.shape[0]
    tree = aabb_tree(pts)
    depth = pow2_round_up(n).bit_length() - 1
    rows = np.zeros((n, stride), dtype=np.uint32)
    degree = np.zeros(n, dtype=np.uint32)
    hull = np.zeros(n, dtype=np.uint8)
    status = np.zeros(n, dtype=np.int32)
    stats = np.zeros(6)
    oa = oo = None
    if old is not None:
        oo = np.ascontiguousarray(old[0], dtype=np.uint32)
        oa = np.ascontiguousarray(old[1], dtype=np.uint32)
    bad = lib().star_host_delaunay_owner(pts.ctypes.data, n, tree.ctypes.data, depth, knn, tree_knn, budget,
                                         None if oa is None else oa.ctypes.data, None if oo is None else oo.ctypes.data,
                                         rows.ctypes.data, stride, degree.ctypes.data, hull.ctypes.data,
                                         status.ctypes.data, stats.ctypes.data)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(degree, out=offsets[1:])
    adjacency = rows[np.arange(stride)[None, :] < degree[:, None]]
    return offsets.astype(np.uint32), adjacency, dict(
        bad=bad, status=status, nodes_pass1=stats[0], nodes_pass2=stats[1], nodes_knn=stats[2], insertions=stats[3],
        closed_by_owner=stats[4], redone=int(stats[5]))
