"""HIP versions of the per-iteration work RadFoamScene does around the tracer.

  pack_attributes    RadFoamScene.get_trace_data (radfoam_model/scene.py:202-217):
                     cat[att_dc, att_sh, activation_scale * softplus(density, beta=10)].to(attr_dtype)
                     as one kernel, differentiable (torch.autograd.Function)
  nearest_point      radfoam.nn for camera positions (triangulation_bindings.cpp:142-181)
  farthest_neighbor  radfoam.farthest_neighbor (triangulation_bindings.cpp:183-217)
  adjacency_from_tets  find_adjacency (src/delaunay/delaunay.cu:140-229): tetrahedra -> CSR adjacency

All three take CUDA (HIP) tensors and run behind the C-ABI of include/radfoam_hip.h; there is no
CPU path in this module (radfoam_amd/shims.py keeps torch restatements for CPU tensors, which the
reference also serves on the CPU: nn_cpu, aabb_tree.cu:417-478).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_ATTR_TYPES = {torch.float32: _lib.RF_ATTR_FLOAT32, torch.float16: _lib.RF_ATTR_FLOAT16}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _sh_degree_of(att_sh_cols: int) -> int:
    # A = 1 + 3 (d+1)^2 ; att_sh holds 3 ((d+1)^2 - 1) columns
    for d in range(4):
        if att_sh_cols == 3 * ((d + 1) ** 2 - 1):
            return d
    raise RuntimeError(f"att_sh has {att_sh_cols} columns: not 3*((d+1)^2-1) for an SH degree d in 0..3")


def _check_f32_cuda(name, t):
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 CUDA tensor")


class _PackAttributes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att_dc, att_sh, density, activation_scale, attr_dtype):
        for name, t in (("att_dc", att_dc), ("att_sh", att_sh), ("density", density)):
            _check_f32_cuda(name, t)
        n = att_dc.size(0)
        if (att_dc.dim() != 2 or att_dc.size(1) != 3 or att_sh.dim() != 2 or att_sh.size(0) != n
                or density.numel() != n):
            raise RuntimeError("expected att_dc [N,3], att_sh [N,3((d+1)^2-1)], density [N,1]")
        if attr_dtype not in _ATTR_TYPES:
            raise RuntimeError("Unsupported attribute dtype")
        degree = _sh_degree_of(att_sh.size(1))
        a = 4 + att_sh.size(1)
        dc, sh, dn = att_dc.contiguous(), att_sh.contiguous(), density.contiguous()
        out = torch.empty((n, a), dtype=attr_dtype, device=att_dc.device)
        with torch.cuda.device(att_dc.device):
            rc = _lib.load().rf_pack_attributes(degree, _ATTR_TYPES[attr_dtype], n, _ptr(dc),
                                                _ptr(sh) if sh.numel() else None, _ptr(dn),
                                                float(activation_scale), _ptr(out), _stream(att_dc.device))
        _lib.check(rc)
        ctx.save_for_backward(dn)
        ctx.degree, ctx.scale, ctx.sh_cols = degree, float(activation_scale), att_sh.size(1)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (dn,) = ctx.saved_tensors
        n = dn.numel()
        g = grad_out.to(torch.float32).contiguous()
        d_dc = torch.empty((n, 3), dtype=torch.float32, device=g.device)
        d_sh = torch.empty((n, ctx.sh_cols), dtype=torch.float32, device=g.device)
        d_dn = torch.empty_like(dn)
        with torch.cuda.device(g.device):
            rc = _lib.load().rf_pack_attributes_backward(ctx.degree, n, _ptr(dn), ctx.scale, _ptr(g),
                                                         _ptr(d_dc), _ptr(d_sh) if ctx.sh_cols else None,
                                                         _ptr(d_dn), _stream(g.device))
        _lib.check(rc)
        return d_dc, d_sh, d_dn, None, None


def pack_attributes(att_dc: torch.Tensor, att_sh: torch.Tensor, density: torch.Tensor,
                    activation_scale: float = 1.0, attr_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """``torch.cat([att_dc, att_sh, activation_scale * F.softplus(density, beta=10)], -1).to(attr_dtype)``
    (scene.py:202-217) in one kernel; gradients flow to all three inputs."""
    return _PackAttributes.apply(att_dc, att_sh, density, activation_scale, attr_dtype)


def nearest_point(points: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    """Index (uint32, shape queries.shape[:-1]) of the point nearest to each query: exact, by
    squared fp32 distance, lowest index among ties.  O(N * Q): meant for camera positions."""
    _check_f32_cuda("points", points)
    _check_f32_cuda("queries", queries)
    if points.dim() != 2 or points.size(-1) != 3 or queries.size(-1) != 3:
        raise RuntimeError("points must be [N,3] and queries [...,3]")
    p, q = points.detach().contiguous(), queries.detach().reshape(-1, 3).contiguous()
    out = torch.empty(q.size(0), dtype=torch.int32, device=p.device)
    scratch = torch.empty(max(q.size(0), 1), dtype=torch.int64, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.load().rf_nearest_point(_ptr(p), p.size(0), _ptr(q), q.size(0), _ptr(out), _ptr(scratch),
                                          _stream(p.device))
    _lib.check(rc)
    return out.view(torch.uint32).reshape(queries.shape[:-1])


def nearest_point_tree(points: torch.Tensor, tree: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    """The same answer as ``nearest_point`` through the AABB tree of the (kd-ordered) points: O(log N) boxes per
    query, a lane per query (rf_nearest_point_tree; reference: nn_kernel, src/aabb_tree/aabb_tree.cu:343-415)."""
    _check_f32_cuda("points", points)
    _check_f32_cuda("queries", queries)
    _check_f32_cuda("aabb_tree", tree)
    if points.dim() != 2 or points.size(-1) != 3 or queries.size(-1) != 3:
        raise RuntimeError("points must be [N,3] and queries [...,3]")
    n = points.size(0)
    pow2 = 1 if n <= 1 else 1 << ((n - 1).bit_length())
    if tuple(tree.shape) != (pow2, 2, 3):
        raise RuntimeError("aabb_tree must have shape [pow2_round_up(num_points), 2, 3]")
    p, q, t = points.detach().contiguous(), queries.detach().reshape(-1, 3).contiguous(), tree.detach().contiguous()
    out = torch.empty(q.size(0), dtype=torch.int32, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.load().rf_nearest_point_tree(_ptr(p), n, _ptr(t), _ptr(q), q.size(0), _ptr(out), _stream(p.device))
    _lib.check(rc)
    return out.view(torch.uint32).reshape(queries.shape[:-1])


def farthest_neighbor(points: torch.Tensor, point_adjacency: torch.Tensor,
                      point_adjacency_offsets: torch.Tensor):
    """(uint32 index of the farthest Delaunay neighbour, fp32 mean half-distance to the neighbours)."""
    _check_f32_cuda("points", points)
    if point_adjacency.dtype != torch.uint32 or point_adjacency_offsets.dtype != torch.uint32:
        raise RuntimeError("point_adjacency and point_adjacency_offsets must have uint32 dtype")
    p = points.detach().contiguous()
    adj, off = point_adjacency.contiguous(), point_adjacency_offsets.contiguous()
    n = p.size(0)
    idx = torch.empty(n, dtype=torch.int32, device=p.device)
    radius = torch.empty(n, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.load().rf_farthest_neighbor(_ptr(p), n, _ptr(adj), _ptr(off), _ptr(idx), _ptr(radius),
                                              _stream(p.device))
    _lib.check(rc)
    return idx.view(torch.uint32), radius


def adjacency_from_tets(tets: torch.Tensor, num_points: int):
    """(point_adjacency uint32[E], point_adjacency_offsets uint32[N+1]) of the triangulation whose
    tetrahedra are ``tets`` (uint32 or int32/int64 [T,4], CUDA): every point's Delaunay neighbours in
    ascending order, as the reference's find_adjacency produces them.  Synchronises once to learn E
    (the reference does the same, delaunay.cu:180-184)."""
    if not tets.is_cuda or tets.dim() != 2 or tets.size(1) != 4:
        raise RuntimeError("tets must be a [T,4] CUDA tensor")
    if tets.dtype in (torch.int64, torch.int32):
        t = tets.to(torch.int32).contiguous()
    elif tets.dtype == torch.uint32:
        t = tets.contiguous()
    else:
        raise RuntimeError("tets must have an integer dtype")
    dev, nt = t.device, t.size(0)
    lib = _lib.load()
    adj = torch.empty(max(12 * nt, 1), dtype=torch.int32, device=dev)
    off = torch.empty(int(num_points) + 1, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(lib.rf_adjacency_workspace_bytes(nt)), 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rf_build_adjacency(_ptr(t), nt, int(num_points), _ptr(adj), _ptr(off), _ptr(count), _ptr(ws),
                                    ws.numel(), _stream(dev))
    _lib.check(rc)
    e = int(count.item())
    return adj[:e].clone().view(torch.uint32), off.view(torch.uint32)
