"""ctypes binding of libradfoam_hip.so (the C-ABI declared in include/radfoam_hip.h).

There is no CPU fallback: if the shared library is missing or a symbol cannot be resolved,
loading raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``radfoam_amd.build.build_hip()``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RADFOAM_HIP_LIB selects another build of the same C-ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("RADFOAM_HIP_LIB") or os.path.join(_HERE, "libradfoam_hip.so")

RF_OK = 0
RF_ERR_WORKSPACE = -2
RF_ATTR_FLOAT32 = 0
RF_ATTR_FLOAT16 = 1


class TraceSettings(C.Structure):
    """rf_trace_settings (radfoam::TraceSettings, src/tracing/pipeline.h:10-13)."""

    _fields_ = [("weight_threshold", C.c_float), ("max_intersections", C.c_uint32)]


class Camera(C.Structure):
    """rf_camera (radfoam::Camera, src/tracing/camera.h:17-26)."""

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


class LaunchOpts(C.Structure):
    """rf_launch_opts."""

    _fields_ = [
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("foam_prepared", C.c_uint32),
        ("image_width", C.c_uint32),
        ("image_height", C.c_uint32),
        ("backward_mode", C.c_uint32),
        ("stats", C.c_void_p),
        ("trail", C.c_void_p),
        ("trail_hops", C.c_void_p),
        ("trail_cap", C.c_uint32),
        ("trail_slots", C.c_uint32),
        ("ray_order", C.c_void_p),
        ("visit_marks", C.c_void_p),
        ("forward_mode", C.c_uint32),
        ("tile_order", C.c_void_p),
        ("tile_cost", C.c_void_p),
        ("attr_grad_pitch", C.c_uint32),
    ]


# every symbol include/radfoam_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_U32 = C.c_uint32
_INT = C.c_int
SYMBOLS = {
    "rf_last_error": (C.c_char_p, []),
    "rf_attribute_dim": (_U32, [_INT]),
    "rf_trail_slots": (_U32, [_U32, _U32, _U32]),
    "rf_launch_blocks": (_U32, [_U32, _U32, _U32, _P]),
    "rf_workspace_bytes": (C.c_size_t, [_U32, _U32, _INT, _INT]),
    "rf_build_adjacent_diff": (_INT, [_P, _U32, _U32, _P, _P, _P, _P]),
    "rf_prepare_foam": (_INT, [_INT, _INT, _U32, _P, _P, _U32, _P, _P, _P, _P, C.c_size_t, _P]),
    "rf_prepare_foam_geometry": (_INT, [_INT, _INT, _U32, _P, _P, _U32, _P, C.c_size_t, _P]),
    "rf_trace_forward": (_INT, [_INT, _INT, C.POINTER(TraceSettings), _U32, _P, _P, _U32, _P, _P, _U32,
                                _P, _P, _U32, _P, _P, _P, _P, _P, _P, C.POINTER(LaunchOpts), _P]),
    "rf_trace_backward": (_INT, [_INT, _INT, C.POINTER(TraceSettings), _U32, _P, _P, _U32, _P, _P, _U32,
                                 _P, _P, _U32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 C.POINTER(LaunchOpts), _P]),
    "rf_pack_attributes": (_INT, [_INT, _INT, _U32, _P, _P, _P, C.c_float, _P, _P]),
    "rf_pack_attributes_backward": (_INT, [_INT, _U32, _P, C.c_float, _P, _P, _P, _P, _P]),
    "rf_nearest_point": (_INT, [_P, _U32, _P, _U32, _P, _P, _P]),
    "rf_nearest_point_tree": (_INT, [_P, _U32, _P, _P, _U32, _P, _P]),
    "rf_farthest_neighbor": (_INT, [_P, _U32, _P, _P, _P, _P, _P]),
    "rf_fetch_batch": (_INT, [_P, C.c_uint64, _U32, _U32, _U32, _INT, _P, _P]),
    "rf_fetch_batch_range": (_INT, [_P, C.c_uint64, _U32, C.c_uint64, _U32, _INT, _P, _P]),
    "rf_ray_order_workspace_bytes": (C.c_size_t, [_U32]),
    "rf_build_ray_order": (_INT, [_P, _P, _U32, _P, _P, C.c_size_t, _P]),
    "rf_adjacency_workspace_bytes": (C.c_size_t, [_U32]),
    "rf_build_adjacency": (_INT, [_P, _U32, _U32, _P, _P, _P, _P, C.c_size_t, _P]),
    "rf_build_aabb_tree": (_INT, [_P, _U32, _P, _P]),
    "rf_kd_order_workspace_bytes": (C.c_size_t, [_U32]),
    "rf_kd_order": (_INT, [_P, _U32, _P, _P, _P, C.c_size_t, _P]),
    "rf_delaunay_workspace_bytes": (C.c_size_t, [_U32]),
    "rf_delaunay_workspace_bytes_for": (C.c_size_t, [_U32, _U32]),
    "rf_delaunay_adjacency": (_INT, [_P, _U32, _P, _P, _P, _P, _U32, _P, _P, _P, C.c_size_t, _P]),
    "rf_adjacency_size": (_INT, [_U32, _P, C.POINTER(_U32), _P]),
    "rf_cast_accumulator": (_INT, [_P, C.c_size_t, _INT, _P, _P]),
    "rf_grad_row_pitch": (_U32, [_U32]),
    "rf_compact_grad_rows": (_INT, [_P, _P, _U32, _U32, _U32, _P, _P, _P]),
    "rf_scatter_grad_rows": (_INT, [_P, _U32, _U32, _U32, _INT, _P, _P, _P]),
    "rf_compact_grad_rows_pitched": (_INT, [_P, _P, _U32, _U32, _U32, _U32, _P, _P, _P]),
    "rf_scatter_grad_rows_pitched": (_INT, [_P, _U32, _U32, _U32, _U32, _INT, _P, _P, _P]),
    "rf_cost_grid_bytes": (C.c_size_t, [_U32]),
    "rf_build_cost_grid": (_INT, [_P, _P, _INT, _U32, _U32, _U32, _P, C.c_size_t, _P]),
    "rf_estimate_tile_cost": (_INT, [_P, _U32, _P, C.POINTER(Camera), _U32, _U32, C.c_float, _U32, _P, _P]),
    "rf_build_tile_orders": (_INT, [_P, _U32, _U32, _U32, _U32, _U32, _P, _U32, _U32, _P, _P]),
    "rf_tile_order_reference": (_INT, [_P, C.POINTER(Camera), _P, _P, _U32, _U32, _P, _P]),
    "rf_gate_tile_order": (_INT, [_P, C.POINTER(Camera), _P, _U32, _U32, C.c_float, _P, _P, _P, _P]),
    "rf_trace_benchmark": (_INT, [_INT, _INT, C.POINTER(TraceSettings), _U32, _P, _P, _U32, _P, _P, _P,
                                  C.POINTER(Camera), _P, _P, C.POINTER(LaunchOpts), _P]),
}

_lib = None


def load():
    """Load libradfoam_hip.so and bind every declared symbol.  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"radfoam_amd: {LIB_PATH} is not built. The tracer has no CPU fallback; build the HIP "
            "library first (python -c 'import __graft_entry__ as g; g.build()').")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().rf_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int):
    """Reference errors surface as RuntimeError (pybind translation of std::runtime_error)."""
    if rc != RF_OK:
        raise RuntimeError(last_error() or f"radfoam_hip error {rc}")
