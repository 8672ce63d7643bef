"""ctypes front-end of oracle/_ref/libref.so: the REFERENCE's own tracer kernels (source text of
/root/reference/src/tracing, untouched) compiled for the CPU against the stand-ins in
oracle/ref_shim/.  TEST INFRASTRUCTURE ONLY; exists only where /root/reference does (the build
container).  Used to validate oracle/rf_oracle.c and to generate tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref.so")
#: the same source text compiled with -ffp-contract=fast -mfma: every a*b+c contracted, as nvcc does by default
LIB_PATH_FMA = os.path.join(_HERE, "_ref", "libref_fma.so")
REFERENCE = "/root/reference"
_lib = None
_libs = {}


class variant:
    """``with refsrc.variant("fma"):`` -- calls inside go to libref_fma.so (the reference source with FMA
    contraction) instead of libref.so (no contraction).  The gap between the two builds is the reference's own
    floating-point indeterminacy (oracle/parity_envelope.py)."""

    def __init__(self, name):
        self.path = {"plain": LIB_PATH, "fma": LIB_PATH_FMA}[name]

    def __enter__(self):
        global _lib
        self.prev = _lib
        lib()   # liboracle.so + the build
        if self.path not in _libs:
            _libs[self.path] = C.CDLL(self.path)
        _lib = _libs[self.path]
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def available() -> bool:
    return (os.path.exists(LIB_PATH) and os.path.exists(LIB_PATH_FMA)) or \
        os.path.isdir(os.path.join(REFERENCE, "src", "tracing"))


def build() -> str:
    from . import oracle as O

    O.build()
    if not os.path.isdir(os.path.join(REFERENCE, "src", "tracing")):
        raise RuntimeError("reference sources not present: oracle/_ref cannot be built here")
    subprocess.run(["make", "-C", _HERE, "-f", "Makefile.ref"], check=True, capture_output=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not (os.path.exists(LIB_PATH) and os.path.exists(LIB_PATH_FMA)):
            build()
        from . import oracle as O

        O.lib()  # liboracle.so provides the half conversions the shim uses
        _lib = _libs.setdefault(LIB_PATH, C.CDLL(LIB_PATH))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def trace_forward(sh_degree, points, attributes, adjacency, offsets, rays, start_point, depth_quantiles=None,
                  weight_threshold=1e-3, max_intersections=1024, return_contribution=False):
    half = attributes.dtype == np.float16
    adt = np.float16 if half else np.float32
    points, attributes = _c(points, np.float32), _c(attributes, adt)
    adjacency, offsets = _c(adjacency, np.uint32), _c(offsets, np.uint32)
    rays = _c(rays, np.float32)
    batch = rays.shape[:-1]
    r = int(np.prod(batch))
    start = _c(np.broadcast_to(start_point, batch), np.uint32)
    q = _c(depth_quantiles, np.float32)
    nq = 0 if q is None else q.shape[-1]
    n = points.shape[0]
    rgba = np.zeros(batch + (4,), dtype=adt)
    nint = np.zeros(batch + (1,), dtype=np.uint32)
    depth = np.zeros(batch + (nq,), dtype=np.float32) if q is not None else None
    didx = np.zeros(batch + (nq,), dtype=np.uint32) if q is not None else None
    contrib = np.zeros((n, 1), dtype=adt) if return_contribution else None
    lib().rfref_trace_forward(C.c_int(sh_degree), C.c_int(int(half)), C.c_float(weight_threshold),
                              C.c_uint32(max_intersections), C.c_uint32(n), _p(points), _p(attributes),
                              C.c_uint32(adjacency.shape[0]), _p(adjacency), _p(offsets), C.c_uint32(r), _p(rays),
                              _p(start), C.c_uint32(nq), _p(q), _p(rgba), _p(depth), _p(didx), _p(nint), _p(contrib))
    out = {"rgba": rgba, "num_intersections": nint}
    if q is not None:
        out["depth"], out["depth_indices"] = depth, didx
    if return_contribution:
        out["contribution"] = contrib
    return out


def trace_backward(sh_degree, points, attributes, adjacency, offsets, rays, start_point, rgb_out, grad_in,
                   depth_quantiles=None, depth_indices=None, depth_grad_in=None, ray_error=None,
                   weight_threshold=1e-3, max_intersections=1024):
    half = attributes.dtype == np.float16
    adt = np.float16 if half else np.float32
    points, attributes = _c(points, np.float32), _c(attributes, adt)
    adjacency, offsets = _c(adjacency, np.uint32), _c(offsets, np.uint32)
    rays = _c(rays, np.float32)
    batch = rays.shape[:-1]
    r = int(np.prod(batch))
    start = _c(np.broadcast_to(start_point, batch), np.uint32)
    q, di, dg = _c(depth_quantiles, np.float32), _c(depth_indices, np.uint32), _c(depth_grad_in, np.float32)
    nq = 0 if q is None else q.shape[-1]
    n = points.shape[0]
    a = 1 + 3 * (sh_degree + 1) ** 2
    rgb_out, grad_in = _c(rgb_out, adt), _c(grad_in, adt)
    err = _c(ray_error, adt)
    pg = np.zeros((n, 3), dtype=np.float32)
    ag = np.zeros((n, a), dtype=adt)
    pe = np.zeros((n, 1), dtype=adt) if err is not None else None
    lib().rfref_trace_backward(C.c_int(sh_degree), C.c_int(int(half)), C.c_float(weight_threshold),
                               C.c_uint32(max_intersections), C.c_uint32(n), _p(points), _p(attributes),
                               C.c_uint32(adjacency.shape[0]), _p(adjacency), _p(offsets), C.c_uint32(r), _p(rays),
                               _p(start), C.c_uint32(nq), _p(q), _p(di), _p(rgb_out), _p(grad_in), _p(dg), _p(err),
                               _p(pg), _p(ag), _p(pe))
    out = {"points_grad": pg, "attr_grad": ag}
    if pe is not None:
        out["point_error"] = pe
    return out


def build_adjacent_diff(points, adjacency, offsets):
    points, adjacency, offsets = _c(points, np.float32), _c(adjacency, np.uint32), _c(offsets, np.uint32)
    e = adjacency.shape[0]
    diff = np.zeros((e + 32, 4), dtype=np.uint16)
    lib().rfref_prefetch_adjacent_diff(_p(points), C.c_uint32(points.shape[0]), C.c_uint32(e), _p(adjacency),
                                       _p(offsets), _p(diff))
    return diff[:e]


def trace_benchmark(sh_degree, points, attributes, adjacency, offsets, adjacent_diff, camera, start_point,
                    weight_threshold=1e-3, max_intersections=1024):
    half = attributes.dtype == np.float16
    adt = np.float16 if half else np.float32
    points, attributes = _c(points, np.float32), _c(attributes, adt)
    adjacency, offsets = _c(adjacency, np.uint32), _c(offsets, np.uint32)
    diff = np.ascontiguousarray(adjacent_diff).view(np.uint16)
    w, h = int(camera["width"]), int(camera["height"])
    out = np.zeros((h, w), dtype=np.uint32)
    v = lambda k: _c(np.asarray(camera[k], dtype=np.float32).reshape(3), np.float32)
    pos, fwd, right, up = v("position"), v("forward"), v("right"), v("up")
    lib().rfref_trace_benchmark(C.c_int(sh_degree), C.c_int(int(half)), C.c_float(weight_threshold),
                                C.c_uint32(max_intersections), C.c_uint32(points.shape[0]), _p(points),
                                _p(attributes), C.c_uint32(adjacency.shape[0]), _p(adjacency), _p(offsets),
                                _p(diff), _p(pos), _p(fwd), _p(right), _p(up), C.c_float(float(camera["fov"])),
                                C.c_uint32(w), C.c_uint32(h), C.c_int(int(camera.get("model", "pinhole") == "fisheye")),
                                C.c_uint32(int(start_point)), _p(out))
    return out
