"""ctypes front-end of the CPU oracle (oracle/rf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never
from radfoam_amd/.  numpy in, numpy out; the argument order follows the reference's
Pipeline::trace_* (src/tracing/pipeline.h:58-131).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

NONE = 0xFFFFFFFF


class Settings(C.Structure):
    _fields_ = [("weight_threshold", C.c_float), ("max_intersections", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("cells_scanned", C.c_uint64),
        ("faces_scanned", C.c_uint64),
        ("hops", C.c_uint64),
        ("segments", C.c_uint64),
        ("segments_lit", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("forward", C.c_float * 3),
        ("right", C.c_float * 3),
        ("up", C.c_float * 3),
        ("fov", C.c_float),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("model", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc, a few seconds)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rf_oracle.c", "rf_oracle_body.inc", "Makefile"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.rfo_float_to_half.restype = C.c_uint16
        _lib.rfo_float_to_half.argtypes = [C.c_float]
        _lib.rfo_half_to_float.restype = C.c_float
        _lib.rfo_half_to_float.argtypes = [C.c_uint16]
        _lib.rfo_expf.restype = C.c_float
        _lib.rfo_expf.argtypes = [C.c_float]
        _lib.rfo_logf.restype = C.c_float
        _lib.rfo_logf.argtypes = [C.c_float]
        _lib.rfo_max_threads.restype = C.c_int
    return _lib


class scan_mode:
    """``with O.scan_mode("filtered") as m:`` -- the oracle EVALUATES a cell's exit the way the HIP kernels do
    (cross-multiplied tournament over the padded list, bit-distance certificate, dividing scan for contested cells)
    instead of the way the reference writes it (``"reference"``, the default: every face divided, running minimum of the
    rounded quotients).  Both compute the same function; ``m.contested`` is the number of cells the filtered evaluation
    handed to the dividing scan while the block was active."""

    def __init__(self, mode):
        self.mode = {"reference": 0, "filtered": 1}[mode]

    def __enter__(self):
        L = lib()
        L.rfo_get_scan_contested.restype = C.c_uint64
        self.prev = L.rfo_get_scan_mode()
        L.rfo_set_scan_mode(self.mode)
        L.rfo_reset_scan_contested()
        return self

    @property
    def contested(self):
        return int(lib().rfo_get_scan_contested())

    def __exit__(self, *exc):
        self.final_contested = self.contested
        lib().rfo_set_scan_mode(self.prev)
        return False


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _settings(weight_threshold, max_intersections):
    return Settings(1e-3 if weight_threshold is None else float(weight_threshold),
                    1024 if max_intersections is None else int(max_intersections))


def attr_dim(sh_degree):
    return 1 + 3 * (sh_degree + 1) ** 2


def make_camera(camera: dict) -> Camera:
    cam = Camera()
    for k in ("position", "forward", "right", "up"):
        v = np.asarray(camera[k], dtype=np.float32).reshape(3)
        setattr(cam, k, (C.c_float * 3)(*[float(x) for x in v]))
    cam.fov = float(camera["fov"])
    cam.width = int(camera["width"])
    cam.height = int(camera["height"])
    cam.model = {"pinhole": 0, "fisheye": 1}[camera.get("model", "pinhole")]
    return cam


def build_adjacent_diff(points, adjacency, offsets, pad: int = 0):
    """half4 table [E+pad,4] as uint16 (view as float16 with .view(np.float16))."""
    points = _c(points, np.float32)
    adjacency = _c(adjacency, np.uint32)
    offsets = _c(offsets, np.uint32)
    e = adjacency.shape[0]
    diff = np.zeros((e + pad, 4), dtype=np.uint16)
    lib().rfo_build_adjacent_diff(_p(points), C.c_uint32(points.shape[0]), _p(adjacency), _p(offsets), _p(diff))
    return diff


def trace_forward(sh_degree, points, attributes, adjacency, offsets, rays, start_point,
                  depth_quantiles=None, weight_threshold=None, max_intersections=None,
                  return_contribution=False, diff=None, num_threads=0, want_stats=False):
    attr_half = attributes.dtype == np.float16
    adt = np.float16 if attr_half else np.float32
    points = _c(points, np.float32)
    attributes = _c(attributes, adt)
    adjacency = _c(adjacency, np.uint32)
    offsets = _c(offsets, np.uint32)
    rays = _c(rays, np.float32)
    batch = rays.shape[:-1]
    r = int(np.prod(batch)) if len(batch) else 1
    start_point = _c(np.broadcast_to(start_point, batch), np.uint32)
    n = points.shape[0]
    nq = 0
    q = None
    if depth_quantiles is not None:
        q = _c(depth_quantiles, np.float32)
        nq = q.shape[-1]
    rgba = np.empty(batch + (4,), dtype=adt)
    nint = np.empty(batch + (1,), dtype=np.uint32)
    depth = np.zeros(batch + (nq,), dtype=np.float32) if q is not None else None
    didx = np.zeros(batch + (nq,), dtype=np.uint32) if q is not None else None
    contrib = np.zeros((n, 1), dtype=adt) if return_contribution else None
    diff = _c(diff, np.uint16) if diff is not None else None
    stats = Stats()
    lib().rfo_trace_forward(
        C.c_int(sh_degree), C.c_int(int(attr_half)), _settings(weight_threshold, max_intersections),
        C.c_uint32(n), _p(points), _p(attributes), C.c_uint32(adjacency.shape[0]), _p(adjacency),
        _p(offsets), _p(diff), C.c_uint32(r), _p(rays), _p(start_point), C.c_uint32(nq), _p(q),
        _p(rgba), _p(depth), _p(didx), _p(nint), _p(contrib), C.c_int(num_threads),
        C.byref(stats) if want_stats else None)
    out = {"rgba": rgba, "num_intersections": nint}
    if q is not None:
        out["depth"] = depth
        out["depth_indices"] = didx
    if return_contribution:
        out["contribution"] = contrib
    if want_stats:
        out["stats"] = stats.as_dict()
    return out


def trace_paths(sh_degree, points, attributes, adjacency, offsets, rays, start_point, cap=256,
                weight_threshold=None, max_intersections=None, num_threads=0):
    """(cells uint32[R, cap], t1 float32[R, cap], n uint32[R]): the cells every ray scans and where it leaves them
    (fp32 attributes only).  For scheduling studies of the kernels (scripts/model_*.py), not a parity output."""
    points, attributes = _c(points, np.float32), _c(attributes, np.float32)
    adjacency, offsets = _c(adjacency, np.uint32), _c(offsets, np.uint32)
    rays = _c(rays, np.float32).reshape(-1, 6)
    r = rays.shape[0]
    start = _c(np.broadcast_to(start_point, (r,)), np.uint32)
    diff = build_adjacent_diff(points, adjacency, offsets, pad=1)
    cells = np.full((r, cap), NONE, dtype=np.uint32)
    t1 = np.full((r, cap), np.inf, dtype=np.float32)
    n = np.zeros(r, dtype=np.uint32)
    lib().rfo_trace_paths(C.c_int(sh_degree), _settings(weight_threshold, max_intersections),
                          C.c_uint32(points.shape[0]), _p(points), _p(attributes), C.c_uint32(adjacency.shape[0]),
                          _p(adjacency), _p(offsets), _p(diff), C.c_uint32(r), _p(rays), _p(start), C.c_uint32(cap),
                          _p(cells), _p(t1), _p(n), C.c_int(num_threads))
    return cells, t1, n


def trace_backward(sh_degree, points, attributes, adjacency, offsets, rays, start_point,
                   rgb_out, grad_in, depth_quantiles=None, depth_indices=None, depth_grad_in=None,
                   ray_error=None, weight_threshold=None, max_intersections=None, diff=None,
                   strict=True, num_threads=0):
    attr_half = attributes.dtype == np.float16
    adt = np.float16 if attr_half else np.float32
    points = _c(points, np.float32)
    attributes = _c(attributes, adt)
    adjacency = _c(adjacency, np.uint32)
    offsets = _c(offsets, np.uint32)
    rays = _c(rays, np.float32)
    batch = rays.shape[:-1]
    r = int(np.prod(batch)) if len(batch) else 1
    start_point = _c(np.broadcast_to(start_point, batch), np.uint32)
    n = points.shape[0]
    a = attr_dim(sh_degree)
    rgb_out = _c(rgb_out, adt)
    grad_in = _c(grad_in, adt)
    nq = 0
    q = di = dg = None
    if depth_quantiles is not None:
        q = _c(depth_quantiles, np.float32)
        nq = q.shape[-1]
        di = _c(depth_indices, np.uint32)
        dg = _c(depth_grad_in, np.float32)
    re = _c(ray_error, adt) if ray_error is not None else None
    pg = np.zeros((n, 3), dtype=np.float32)
    ag = np.zeros((n, a), dtype=adt)
    pe = np.zeros((n, 1), dtype=adt) if re is not None else None
    diff = _c(diff, np.uint16) if diff is not None else None
    lib().rfo_trace_backward(
        C.c_int(sh_degree), C.c_int(int(attr_half)), _settings(weight_threshold, max_intersections),
        C.c_uint32(n), _p(points), _p(attributes), C.c_uint32(adjacency.shape[0]), _p(adjacency),
        _p(offsets), _p(diff), C.c_uint32(r), _p(rays), _p(start_point), C.c_uint32(nq), _p(q),
        _p(di), _p(rgb_out), _p(grad_in), _p(dg), _p(re), _p(pg), _p(ag), _p(pe),
        C.c_int(int(bool(strict))), C.c_int(num_threads))
    out = {"points_grad": pg, "attr_grad": ag}
    if pe is not None:
        out["point_error"] = pe
    return out


def cast_rays(camera: dict) -> np.ndarray:
    cam = make_camera(camera)
    rays = np.empty((cam.height, cam.width, 6), dtype=np.float32)
    lib().rfo_cast_rays(C.byref(cam), _p(rays))
    return rays


def trace_benchmark(sh_degree, points, attributes, adjacency, offsets, adjacent_diff, camera,
                    start_point, weight_threshold=None, max_intersections=None, num_threads=0):
    attr_half = attributes.dtype == np.float16
    adt = np.float16 if attr_half else np.float32
    points = _c(points, np.float32)
    attributes = _c(attributes, adt)
    adjacency = _c(adjacency, np.uint32)
    offsets = _c(offsets, np.uint32)
    diff = np.ascontiguousarray(adjacent_diff).view(np.uint16)
    # the reference over-reads up to chunk-1 entries past the table; results do not depend on them
    cam = make_camera(camera)
    out = np.empty((cam.height, cam.width), dtype=np.uint32)
    lib().rfo_trace_benchmark(
        C.c_int(sh_degree), C.c_int(int(attr_half)), _settings(weight_threshold, max_intersections),
        C.c_uint32(points.shape[0]), _p(points), _p(attributes), _p(adjacency), _p(offsets), _p(diff),
        C.byref(cam), C.c_uint32(int(start_point)), _p(out), C.c_int(num_threads))
    return out


# ---- float64 twin ----------------------------------------------------------------------

def trace_forward_f64(sh_degree, points, attributes, adjacency, offsets, rays, start_point,
                      depth_quantiles=None, weight_threshold=None, max_intersections=None,
                      return_contribution=False):
    points = _c(points, np.float64)
    attributes = _c(attributes, np.float64)
    adjacency = _c(adjacency, np.uint32)
    offsets = _c(offsets, np.uint32)
    rays = _c(rays, np.float64)
    batch = rays.shape[:-1]
    r = int(np.prod(batch)) if len(batch) else 1
    start_point = _c(np.broadcast_to(start_point, batch), np.uint32)
    n = points.shape[0]
    nq = 0
    q = None
    if depth_quantiles is not None:
        q = _c(depth_quantiles, np.float64)
        nq = q.shape[-1]
    rgba = np.empty(batch + (4,), dtype=np.float64)
    nint = np.empty(batch + (1,), dtype=np.uint32)
    depth = np.zeros(batch + (nq,), dtype=np.float64) if q is not None else None
    didx = np.zeros(batch + (nq,), dtype=np.uint32) if q is not None else None
    contrib = np.zeros((n, 1), dtype=np.float64) if return_contribution else None
    lib().rfo_trace_forward_f64(
        C.c_int(sh_degree), _settings(weight_threshold, max_intersections), C.c_uint32(n),
        _p(points), _p(attributes), _p(adjacency), _p(offsets), C.c_uint32(r), _p(rays),
        _p(start_point), C.c_uint32(nq), _p(q), _p(rgba), _p(depth), _p(didx), _p(nint), _p(contrib))
    out = {"rgba": rgba, "num_intersections": nint}
    if q is not None:
        out["depth"] = depth
        out["depth_indices"] = didx
    if return_contribution:
        out["contribution"] = contrib
    return out


def trace_backward_f64(sh_degree, points, attributes, adjacency, offsets, rays, start_point,
                       rgb_out, grad_in, depth_quantiles=None, depth_indices=None,
                       depth_grad_in=None, ray_error=None, weight_threshold=None,
                       max_intersections=None, strict=True):
    points = _c(points, np.float64)
    attributes = _c(attributes, np.float64)
    adjacency = _c(adjacency, np.uint32)
    offsets = _c(offsets, np.uint32)
    rays = _c(rays, np.float64)
    batch = rays.shape[:-1]
    r = int(np.prod(batch)) if len(batch) else 1
    start_point = _c(np.broadcast_to(start_point, batch), np.uint32)
    n = points.shape[0]
    a = attr_dim(sh_degree)
    rgb_out = _c(rgb_out, np.float64)
    grad_in = _c(grad_in, np.float64)
    nq = 0
    q = di = dg = None
    if depth_quantiles is not None:
        q = _c(depth_quantiles, np.float64)
        nq = q.shape[-1]
        di = _c(depth_indices, np.uint32)
        dg = _c(depth_grad_in, np.float64)
    re = _c(ray_error, np.float64) if ray_error is not None else None
    pg = np.zeros((n, 3), dtype=np.float64)
    ag = np.zeros((n, a), dtype=np.float64)
    pe = np.zeros((n, 1), dtype=np.float64) if re is not None else None
    lib().rfo_trace_backward_f64(
        C.c_int(sh_degree), _settings(weight_threshold, max_intersections), C.c_uint32(n),
        _p(points), _p(attributes), _p(adjacency), _p(offsets), C.c_uint32(r), _p(rays),
        _p(start_point), C.c_uint32(nq), _p(q), _p(di), _p(rgb_out), _p(grad_in), _p(dg), _p(re),
        _p(pg), _p(ag), _p(pe), C.c_int(int(bool(strict))))
    out = {"points_grad": pg, "attr_grad": ag}
    if pe is not None:
        out["point_error"] = pe
    return out
