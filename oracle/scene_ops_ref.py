"""numpy restatements of the scene-side operators (TEST INFRASTRUCTURE ONLY, like the rest of
oracle/): the checker for radfoam_amd/csrc/rf_scene_ops.hip.

  pack_attributes(_backward)  radfoam_model/scene.py:202-217 (get_primal_density, get_primal_attributes,
                              get_trace_data): cat[att_dc, att_sh, scale * softplus(density, beta=10)]
  nearest_point               radfoam.nn semantics (triangulation_bindings.cpp:142-181): nearest by distance
  farthest_neighbor           src/delaunay/triangulation_ops.cu:9-44
  adjacency_from_tets         find_adjacency, src/delaunay/delaunay.cu:140-229

Pinned in tests/test_scene_ops.py against torch evaluating the reference's own expression
(F.softplus(beta=10), torch.cat, autograd) on the CPU.
"""
from __future__ import annotations

import numpy as np

BETA = 10.0        # scene.py:203
THRESHOLD = 20.0   # torch.nn.functional.softplus default


def _z(x):
    """beta * x as the fp32 product both torch and the kernel form before exponentiating (its rounding
    error, amplified by |z|, is part of the reference's fp32 result), widened to float64."""
    return (np.asarray(x, dtype=np.float32) * np.float32(BETA)).astype(np.float64)


def softplus(x):
    z = _z(x)
    x = np.asarray(x, dtype=np.float64)
    return np.where(z > THRESHOLD, x, np.log1p(np.exp(np.minimum(z, THRESHOLD))) / BETA)


def softplus_grad(x):
    z = _z(x)
    ez = np.exp(np.minimum(z, THRESHOLD))
    return np.where(z > THRESHOLD, 1.0, ez / (ez + 1.0))


def pack_attributes(att_dc, att_sh, density, activation_scale=1.0, dtype=np.float32):
    n = att_dc.shape[0]
    dens = (activation_scale * softplus(density.reshape(n, 1))).astype(np.float32)
    out = np.concatenate([att_dc.astype(np.float32), att_sh.astype(np.float32).reshape(n, -1), dens], axis=1)
    return out.astype(dtype)


def pack_attributes_backward(density, activation_scale, attr_grad):
    g = np.asarray(attr_grad, dtype=np.float64)
    a = g.shape[1]
    d_dc = g[:, :3].astype(np.float32)
    d_sh = g[:, 3:a - 1].astype(np.float32)
    d_dn = (g[:, a - 1:] * activation_scale * softplus_grad(density.reshape(-1, 1))).astype(np.float32)
    return d_dc, d_sh, d_dn.reshape(density.shape)


def _fma32(a, b, c):
    """float32 fused multiply-add through float64 (the product of two floats is exact there)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def squared_distance(points, q):
    """fp32, in the order of the kernel: fma(dx,dx, fma(dy,dy, dz*dz))."""
    d = points.astype(np.float32) - np.asarray(q, dtype=np.float32)[None, :]
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    return _fma32(dx, dx, _fma32(dy, dy, (dz * dz).astype(np.float32)))


def nearest_point(points, queries):
    q = np.asarray(queries, dtype=np.float32).reshape(-1, 3)
    out = np.empty(q.shape[0], dtype=np.uint32)
    for i in range(q.shape[0]):
        out[i] = int(np.argmin(squared_distance(points, q[i])))   # argmin: first minimum
    return out.reshape(np.asarray(queries).shape[:-1])


def farthest_neighbor(points, adjacency, offsets):
    p = points.astype(np.float32)
    n = p.shape[0]
    off = offsets.astype(np.int64)
    adj = adjacency.astype(np.int64)
    counts = np.diff(off)
    owner = np.repeat(np.arange(n), counts)
    d = p[adj] - p[owner]
    d2 = _fma32(d[:, 0], d[:, 0], _fma32(d[:, 1], d[:, 1], (d[:, 2] * d[:, 2]).astype(np.float32)))
    dist = np.sqrt(d2.astype(np.float32)).astype(np.float32)
    idx = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    radius = np.zeros(n, dtype=np.float32)
    for i in range(n):
        b, e = off[i], off[i + 1]
        s = np.float32(0.0)
        best = np.float32(0.0)
        for f in range(b, e):
            s = np.float32(np.float64(s) + 0.5 * np.float64(dist[f]))
            if dist[f] > best:
                best = dist[f]
                idx[i] = adj[f]
        radius[i] = s / np.float32(e - b) if e > b else np.float32(np.nan)
    return idx, radius


def adjacency_from_tets(tets, num_points):
    """CSR (adjacency, offsets) with ascending neighbour lists: the unique directed edges of all
    tetrahedra, sorted by (source, target).  Tets with an index >= num_points or a repeated vertex
    are ignored, as rf_build_adjacency does."""
    t = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    pairs = []
    for a in range(4):
        for b in range(4):
            if a != b:
                ok = (t[:, a] < num_points) & (t[:, b] < num_points) & (t[:, a] != t[:, b])
                pairs.append(np.stack([t[ok, a], t[ok, b]], axis=1))
    e = np.unique(np.concatenate(pairs, axis=0), axis=0) if pairs else np.zeros((0, 2), np.int64)
    offsets = np.searchsorted(e[:, 0], np.arange(num_points + 1), side="left").astype(np.uint32)
    return e[:, 1].astype(np.uint32), offsets
