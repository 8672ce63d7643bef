"""radfoam_amd -- MI355X-native differentiable Voronoi ray tracer (drop-in for radfoam's tracing path).

Public surface = the reference's ``radfoam`` module (torch_bindings/radfoam/__init__.py.in:29
re-exporting torch_bindings): ``create_pipeline`` and ``Pipeline.trace_forward /
trace_backward / trace_benchmark`` run hand-written HIP kernels for gfx950 behind the C-ABI of
include/radfoam_hip.h; everything else the reference module exports is a torch/scipy shim
(radfoam_amd/shims.py).  ``import radfoam`` resolves to this package through the alias
package ``radfoam/`` at the repo root.
"""
from .pipeline import Pipeline, create_pipeline, invalidate_caches
from .scene_ops import pack_attributes
from .shims import (BatchFetcher, Triangulation, TriangulationFailedError, Viewer, build_aabb_tree,
                    farthest_neighbor, nn, run_with_viewer)

__all__ = [
    "Pipeline", "create_pipeline", "Triangulation", "TriangulationFailedError", "build_aabb_tree",
    "nn", "farthest_neighbor", "BatchFetcher", "Viewer", "run_with_viewer", "pack_attributes",
    "invalidate_caches",
]
