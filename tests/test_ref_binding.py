"""Seam B through the reference's own virtual interface.

oracle/_ref/libhip_pipeline.so = tests/ref_binding/hip_pipeline.cpp -- a complete ``radfoam::Pipeline`` subclass
over include/radfoam_hip.h plus ``radfoam::create_pipeline`` -- compiled against /root/reference/src/tracing/
pipeline.h (oracle/Makefile.ref; built where /root/reference exists, the .so travels to the GPU box), and a C
harness that calls Pipeline's virtuals.  -m gpu: what comes out of that interface equals what the ctypes path
(radfoam_amd/pipeline.py) returns for the same inputs.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libhip_pipeline.so")

needs_lib = pytest.mark.skipif(
    not os.path.exists(LIB),
    reason="oracle/_ref/libhip_pipeline.so is not built (needs /root/reference: make -C oracle -f Makefile.ref)")

_P, _U32, _F = C.c_void_p, C.c_uint32, C.c_float


def _load():
    lib = C.CDLL(LIB)
    lib.rb_last_error.restype = C.c_char_p
    lib.rb_create.restype = _P
    lib.rb_create.argtypes = [C.c_int, C.c_int, _P]
    lib.rb_destroy.argtypes = [_P]
    lib.rb_attribute_dim.restype = _U32
    lib.rb_attribute_dim.argtypes = [_P]
    lib.rb_attribute_type.argtypes = [_P]
    lib.rb_trace_visualization_is_rejected.argtypes = [_P]
    lib.rb_trace_forward.argtypes = [_P, _F, _U32, _U32, _P, _P, _U32, _P, _P, _U32, _P, _P, _U32, _P, _P, _P, _P, _P, _P]
    lib.rb_trace_backward.argtypes = [_P, _F, _U32, _U32, _P, _P, _U32, _P, _P, _U32, _P, _P, _U32, _P, _P, _P, _P, _P,
                                      _P, _P, _P, _P, _P]
    lib.rb_trace_benchmark.argtypes = [_P, _F, _U32, _U32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U32, _U32, C.c_int,
                                       _P, _P]
    lib.rb_prefetch_adjacent_diff.argtypes = [_P, _U32, _U32, _P, _P, _P, _P]
    return lib


@needs_lib
def test_binding_compiles_against_the_reference_header_and_mirrors_create_pipeline():
    """not gpu: the library exists (= hip_pipeline.cpp compiled against the reference's pipeline.h and linked
    against the C-ABI), and create_pipeline keeps the reference's dispatch rules (pipeline.cu:776-805)."""
    lib = _load()
    f16, f32, f64 = lib.rb_scalar_float16(), lib.rb_scalar_float32(), lib.rb_scalar_float64()
    for d, a in [(0, 4), (1, 13), (2, 28), (3, 49)]:
        for st in (f32, f16):
            h = lib.rb_create(d, st, None)
            assert h, lib.rb_last_error()
            assert lib.rb_attribute_dim(h) == a and lib.rb_attribute_type(h) == st
            assert lib.rb_trace_visualization_is_rejected(h) == 1
            lib.rb_destroy(h)
    assert not lib.rb_create(4, f32, None) and lib.rb_last_error() == b"Unsupported SH degree"
    assert not lib.rb_create(-1, f16, None) and lib.rb_last_error() == b"Unsupported SH degree"
    assert not lib.rb_create(1, f64, None) and lib.rb_last_error() == b"Unsupported attribute type"


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


@needs_lib
@pytest.mark.gpu
@pytest.mark.parametrize("d,half", [(0, False), (2, False), (3, True), (1, True)])
def test_reference_interface_matches_the_ctypes_path(foam_factory, d, half):
    import radfoam

    dev = "cuda:0"
    lib = _load()
    dtype = torch.float16 if half else torch.float32
    fm = foam_factory(6000, d, 70 + d)
    rays_np, starts_np = H.random_rays(fm, 5000, seed=8)
    rng = np.random.default_rng(4)
    q_np = np.sort(rng.uniform(0.02, 0.98, size=(5000, 2)).astype(np.float32), axis=1)[:, ::-1].copy()
    p, a, adj, off = H.to_torch_foam(fm, dev, dtype)
    n, e, A = p.shape[0], adj.numel(), a.shape[1]
    r, s, q = torch.from_numpy(rays_np).to(dev), torch.from_numpy(starts_np).to(dev), torch.from_numpy(q_np).to(dev)
    g = torch.from_numpy(rng.normal(size=(5000, 4)).astype(np.float32)).to(dev).to(dtype)
    dg = torch.from_numpy(rng.normal(size=(5000, 2)).astype(np.float32)).to(dev)
    err = torch.from_numpy(rng.uniform(0, 1, size=5000).astype(np.float32)).to(dev).to(dtype)

    pipe = radfoam.create_pipeline(d, dtype)          # the ctypes path (no trail: plain tensors, default "auto")
    ref_f = pipe.trace_forward(p, a, adj, off, r, s, depth_quantiles=q, return_contribution=True)
    ref_b = pipe.trace_backward(p, a, adj, off, r, s, ref_f["rgba"], g, q, ref_f["depth_indices"], dg, err)

    h = lib.rb_create(d, lib.rb_scalar_float16() if half else lib.rb_scalar_float32(),
                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert h, lib.rb_last_error()
    try:
        # ---- trace_forward through Pipeline::trace_forward -------------------------------------------
        rgba = torch.empty((5000, 4), dtype=dtype, device=dev)
        depth = torch.zeros((5000, 2), dtype=torch.float32, device=dev)
        didx = torch.zeros((5000, 2), dtype=torch.uint32, device=dev)
        nint = torch.zeros((5000,), dtype=torch.uint32, device=dev)
        contrib = torch.zeros((n,), dtype=dtype, device=dev)
        rc = lib.rb_trace_forward(h, 1e-3, 1024, n, _ptr(p), _ptr(a), e, _ptr(adj), _ptr(off), 5000, _ptr(r), _ptr(s),
                                  2, _ptr(q), _ptr(rgba), _ptr(depth), _ptr(didx), _ptr(nint), _ptr(contrib))
        assert rc == 0, lib.rb_last_error()
        torch.cuda.synchronize()
        assert torch.equal(rgba, ref_f["rgba"])
        assert torch.equal(depth, ref_f["depth"])
        assert torch.equal(didx.view(torch.int32), ref_f["depth_indices"].view(torch.int32))
        assert torch.equal(nint.view(torch.int32), ref_f["num_intersections"].reshape(-1).view(torch.int32))
        ok, rel, worst = H.grad_close(contrib.float().cpu().numpy(), ref_f["contribution"].float().reshape(-1).cpu().numpy(),
                                      rtol=2e-3 if half else 1e-3)
        assert ok, (rel, worst)
        # ---- trace_backward ---------------------------------------------------------------------------
        pg = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        ag = torch.zeros((n, A), dtype=dtype, device=dev)
        pe = torch.zeros((n,), dtype=dtype, device=dev)
        rg = torch.zeros_like(r)
        rc = lib.rb_trace_backward(h, 1e-3, 1024, n, _ptr(p), _ptr(a), e, _ptr(adj), _ptr(off), 5000, _ptr(r), _ptr(s),
                                   2, _ptr(q), _ptr(didx), _ptr(rgba), _ptr(g), _ptr(dg), _ptr(err), _ptr(rg), _ptr(pg),
                                   _ptr(ag), _ptr(pe))
        assert rc == 0, lib.rb_last_error()
        torch.cuda.synchronize()
        for got, key in ((pg, "points_grad"), (ag, "attr_grad"), (pe, "point_error")):
            ok, rel, worst = H.grad_close(got.float().cpu().numpy().reshape(-1),
                                          ref_b[key].float().cpu().numpy().reshape(-1), rtol=2e-3 if half else 1e-3)
            assert ok and rel < (2e-3 if half else 1e-5), (key, rel, worst)
            assert float(ref_b[key].float().abs().max()) > 0
        # ---- prefetch_adjacent_diff + trace_benchmark (no adjacency size in the reference's signature) ------
        diff = torch.zeros((e, 4), dtype=torch.float16, device=dev)
        assert lib.rb_prefetch_adjacent_diff(_ptr(p), n, e, _ptr(adj), _ptr(off), _ptr(diff), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(diff, pipe.build_adjacent_diff(p, adj, off))
        cam, _, start = H.camera_setup(fm, 96, 64)
        sp = torch.tensor([int(start)], dtype=torch.int64).to(torch.uint32).to(dev)
        want = torch.zeros((64, 96), dtype=torch.uint32, device=dev)
        camera = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
        pipe.trace_benchmark(p, a, adj, off, diff, camera, sp, want, weight_threshold=0.05)
        got = torch.zeros((64, 96), dtype=torch.uint32, device=dev)
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        rc = lib.rb_trace_benchmark(h, 0.05, 1024, n, _ptr(p), _ptr(a), _ptr(adj), _ptr(off), _ptr(diff),
                                    f3(cam["position"]), f3(cam["forward"]), f3(cam["right"]), f3(cam["up"]),
                                    cam["fov"], 96, 64, 0, _ptr(sp), _ptr(got))
        assert rc == 0, lib.rb_last_error()
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)) and int((got.view(torch.int32) != 0).sum()) > 0
        # an argument error surfaces as the exception text (-> Python RuntimeError in the reference's binding)
        rc = lib.rb_trace_forward(h, 1e-3, 1024, n, None, _ptr(a), e, _ptr(adj), _ptr(off), 5000, _ptr(r), _ptr(s),
                                  0, None, _ptr(rgba), None, None, None, None)
        assert rc == -1 and b"null pointer" in lib.rb_last_error()
    finally:
        lib.rb_destroy(h)


# ---- the triangulation through radfoam::Triangulation (tests/ref_binding/hip_triangulation.cpp) -----------------------

def _load_triangulation():
    lib = C.CDLL(LIB)
    lib.tb_last_error.restype = C.c_char_p
    lib.tb_init.argtypes = [_P]
    lib.tb_create.restype = _P
    lib.tb_create.argtypes = [_P, _U32]
    lib.tb_destroy.argtypes = [_P]
    lib.tb_rebuild.argtypes = [_P, _P, _U32, C.c_int]
    for name in ("tb_num_points", "tb_point_adjacency_size", "tb_num_tets"):
        getattr(lib, name).restype = _U32
        getattr(lib, name).argtypes = [_P]
    for name in ("tb_permutation", "tb_point_adjacency", "tb_point_adjacency_offsets"):
        getattr(lib, name).restype = _P
        getattr(lib, name).argtypes = [_P]
    return lib


@needs_lib
def test_triangulation_binding_compiles_against_the_reference_header():
    """not gpu: hip_triangulation.cpp -- a radfoam::Triangulation subclass over rf_kd_order / rf_build_aabb_tree /
    rf_delaunay_adjacency -- compiled against the reference's src/delaunay/delaunay.h and defines the factory the
    reference declares there (Triangulation::create_triangulation, delaunay.h:41-42)."""
    import subprocess
    _load_triangulation()   # every tb_* entry point resolves
    syms = subprocess.run(["nm", "-DC", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    assert "radfoam::Triangulation::create_triangulation(void const*, unsigned int)" in syms


@needs_lib
@pytest.mark.gpu
def test_triangulation_binding_matches_the_python_path():
    """What comes out of the reference's virtual interface (permutation, point_adjacency, offsets; rebuild with and
    without `incremental`; the failure type) equals radfoam.Triangulation's."""
    import radfoam
    lib = _load_triangulation()
    lib.tb_init(None)
    rng = np.random.default_rng(31)
    raw = torch.from_numpy(rng.uniform(-1, 1, size=(30000, 3)).astype(np.float32)).cuda()
    tri = radfoam.Triangulation(raw)
    h = lib.tb_create(raw.data_ptr(), raw.shape[0])
    assert h, lib.tb_last_error()

    def view(ptr, count):
        out = torch.empty(count, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(4 * count), 3)
        return out.view(torch.uint32)

    n = raw.shape[0]
    assert lib.tb_num_points(h) == n and lib.tb_num_tets(h) == 0
    e = lib.tb_point_adjacency_size(h)
    assert e == tri.point_adjacency().numel()
    assert torch.equal(view(lib.tb_permutation(h), n).view(torch.int32), tri.permutation().view(torch.int32))
    assert torch.equal(view(lib.tb_point_adjacency(h), e).view(torch.int32), tri.point_adjacency().view(torch.int32))
    assert torch.equal(view(lib.tb_point_adjacency_offsets(h), n + 1).view(torch.int32),
                       tri.point_adjacency_offsets().view(torch.int32))
    sorted_pts = raw[tri.permutation().to(torch.long)]
    moved = (sorted_pts + 1e-3 * torch.randn_like(sorted_pts)).contiguous()
    assert lib.tb_rebuild(h, moved.data_ptr(), n, 1) == 0            # incremental: no re-sort
    assert tri.rebuild(moved, incremental=True) is False
    e = lib.tb_point_adjacency_size(h)
    assert torch.equal(view(lib.tb_point_adjacency(h), e).view(torch.int32), tri.point_adjacency().view(torch.int32))
    dup = moved.clone()
    dup[5] = dup[777]
    assert lib.tb_rebuild(h, dup.data_ptr(), n, 0) == -1
    assert lib.tb_last_error_is_triangulation_failed() == 1 and b"duplicate points found" in lib.tb_last_error()
    lib.tb_destroy(h)
