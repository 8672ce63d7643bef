"""Host-side mirror of the reference's pipeline bindings.

``create_pipeline`` / ``Pipeline.trace_forward`` / ``trace_backward`` / ``trace_benchmark`` have the
names, argument order, defaults, validation messages and returned dict keys of
/root/reference/torch_bindings/pipeline_bindings.cpp (cited per method), so that
radfoam_model/render.py::TraceRays and scene.py call them unchanged.  The work itself is done
by the HIP library behind the C-ABI of include/radfoam_hip.h; torch only provides device
memory and the current stream.  There is no CPU path: CPU tensors are rejected with the
reference's "... must be on CUDA device" errors.

Differences from the reference, all documented in DESIGN.md:
  * the packed foam (cell records + face table, what the reference rebuilds inside every
    trace_forward AND trace_backward call, pipeline.cu:613-620,667-674) is cached on the
    Pipeline while the input tensors are unchanged (same storage, same ``_version``);
  * fp16 pipelines accumulate ``contribution`` / ``attr_grad`` / ``point_error`` in fp32 and
    round once at the end;
  * ``ray_grad`` is returned zero-filled (the reference returns uninitialised memory).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_DEFAULT_WEIGHT_THRESHOLD = 0.001  # default_trace_settings(), src/tracing/pipeline.h:15-20
_DEFAULT_MAX_INTERSECTIONS = 1024


def _parse_attr_dtype(attr_dtype):
    """dtype_to_scalar_type(py::object), torch_bindings/bindings.h:24-42."""
    s = str(attr_dtype)
    if s in ("float32", "torch.float32"):
        return torch.float32
    if s in ("float16", "torch.float16"):
        return torch.float16
    if s in ("float64", "torch.float64"):
        # parsed by the binding, rejected by create_pipeline (pipeline.cu:802-804)
        raise RuntimeError("Unsupported attribute type")
    raise RuntimeError(f"unsupported dtype '{s}'")


def _dtype_name(dt) -> str:
    return {torch.float32: "float32", torch.float16: "float16", torch.float64: "float64"}.get(dt, str(dt))


def _c10_name(dt) -> str:
    """c10::toString(ScalarType) for the dtypes that can show up in the messages."""
    return {
        torch.float32: "Float", torch.float64: "Double", torch.float16: "Half",
        torch.bfloat16: "BFloat16", torch.int32: "Int", torch.int64: "Long",
        torch.uint32: "UInt32", torch.uint8: "Byte", torch.int16: "Short", torch.bool: "Bool",
    }.get(dt, str(dt))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


#: bumped by invalidate_caches(): every packed-foam / trail / ray-order cache entry made before the
#: bump stops matching.  radfoam.Triangulation.rebuild() bumps it, because a rebuild may rewrite the
#: adjacency buffers in place (the reference's getters are from_blob views of buffers that
#: Triangulation::rebuild overwrites, triangulation_bindings.cpp:225-237).
_EPOCH = [0]


def invalidate_caches():
    """Forget every cached packed foam / hop trail / ray order of every Pipeline.

    Call it after writing to a tensor behind autograd's back -- ``param.data.add_()``, a raw pointer,
    numpy / DLPack aliases -- i.e. whenever ``tensor._version`` does not see the write.  Ordinary
    in-place torch ops and optimiser steps bump ``_version`` and need nothing."""
    _EPOCH[0] += 1


class _Uncacheable:
    """Key of a tensor whose writes cannot be tracked (inference tensors have no version counter):
    never equal to anything, itself included, so lookups on it always miss."""

    def __eq__(self, other):
        return False

    def __ne__(self, other):
        return True

    __hash__ = None


def _tensor_key(t):
    """(storage address, version counter, shape, dtype, device) of a tensor, or an unmatchable key
    when torch cannot report a version (tensors created under torch.inference_mode())."""
    if t is None:
        return None
    try:
        ver = t._version
    except RuntimeError:
        return _Uncacheable()
    return (t.data_ptr(), ver, tuple(t.shape), t.dtype, t.device, _EPOCH[0])


def _source_key(t, t_c):
    """Key of a per-ray input as the CALLER holds it.  ``t_c = t.contiguous()`` is a fresh copy whenever ``t`` is a strided
    view (RadFoamScene.collect_error_map traces ``rays[:, d0::2, d1::2]``): keyed on the copy, the trace_backward that
    follows -- autograd hands it the same view, which is copied again -- would find neither the ray order nor the hop trail
    of its forward and re-walk every ray (70 ms instead of 8 for a 960x540 view of the 2 M-point scene).  So a view is keyed
    on its own storage address, version, shape AND strides."""
    if t is None or t_c is t:
        return _tensor_key(t_c)
    k = _tensor_key(t)
    return k if isinstance(k, _Uncacheable) else k + (tuple(t.stride()), int(t.storage_offset()))


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _FoamCache:
    """One packed foam per Pipeline, keyed on the identity + version of the input tensors
    (points, attributes, adjacency, offsets and, for trace_benchmark, the caller's half table).

    Two levels: ``lookup`` -- everything unchanged, the workspace is used as it is; ``lookup_topology``
    -- adjacency and offsets unchanged (the triangulation was not rebuilt) but points / attributes
    were updated, e.g. by an optimiser step: the padded offsets and links in the workspace are still
    right and only cells and face offsets are repacked (rf_launch_opts.foam_prepared = 2).

    The entry keeps the input tensors alive, so their storage cannot be handed to a different
    tensor while the entry could still be matched.
    """

    def __init__(self):
        self.key = None
        self.topo_key = None
        self.refs = None
        self.workspace = None

    @staticmethod
    def _key(tensors):
        return tuple(_tensor_key(t) for t in tensors)

    def lookup(self, tensors):
        return self.key is not None and self.key == self._key(tensors)

    def lookup_topology(self, tensors):
        """tensors = (points, attributes, adjacency, offsets, ext_diff)"""
        if self.topo_key is None or tensors[4] is not None:
            return False
        n_key = (tuple(tensors[0].shape), tensors[0].device, tuple(tensors[1].shape), tensors[1].dtype)
        return self.topo_key == (self._key(tensors[2:4]), n_key)

    def store(self, tensors):
        self.key = self._key(tensors)
        self.refs = tuple(tensors)
        if tensors[4] is None:
            n_key = (tuple(tensors[0].shape), tensors[0].device, tuple(tensors[1].shape), tensors[1].dtype)
            self.topo_key = (self._key(tensors[2:4]), n_key)
        else:   # packed from a caller's half table: not reusable by the differentiable path
            self.topo_key = None

    def invalidate_geometry(self):
        """Forget the points / attributes the workspace was packed from (keeps the topology level)."""
        self.key = None

    def clear(self):
        self.key = None
        self.topo_key = None
        self.refs = None


def tile_order(cost, default, rule):
    """The tile every block of an image launch walks, from the static assignment `default` (rf_launch_blocks: block b ->
    tile, values >= the number of tiles = none; block b runs on XCD b % 8) and a cost per tile (its longest ray).

      "xcd[:n]"       every XCD keeps its tiles and takes them longest first (in classes of n steps, static order within);
      "tail[:count]"  the static order, except that the `count` (2048) cheapest tiles of the frame come last, the longest
                      of them first: what is still running when the launch drains is short, and the bulk of the launch
                      keeps the strips of the static dealing;
      "global"        all tiles longest first, whatever the XCD;
      "chunk:n"       runs of n consecutive tiles per XCD, in tile order (experiment: L2 sharing of flat batches).
    Returns int64 [len(default)]: a permutation of `default`."""
    nt = int(cost.numel())
    if rule == "global":
        order = torch.full_like(default, nt)
        order[:nt] = torch.argsort(cost, descending=True, stable=True)
        return order
    per_xcd = default.view(-1, 8)                                      # [position in the XCD's sequence, XCD]
    valid = default < nt
    c = torch.where(valid, cost[default.clamp(max=nt - 1)], torch.full_like(default, -1, dtype=cost.dtype)).view(-1, 8)
    if rule == "xcd":
        idx = torch.sort(c, dim=0, descending=True, stable=True).indices
    elif rule.startswith("xcd:"):
        # longest first in classes of <n> steps; tiles of a class keep the static order (neighbours stay together)
        q = max(1, int(rule.split(":")[1]))
        idx = torch.sort(torch.div(c, q, rounding_mode="floor"), dim=0, descending=True, stable=True).indices
    elif rule.startswith("tail"):
        count = int(rule.split(":")[1]) if ":" in rule else 2048
        count = max(1, min(count, nt))
        limit = torch.kthvalue(cost.to(torch.float32), count).values.to(cost.dtype)
        pos = torch.arange(c.shape[0], device=c.device, dtype=torch.int64).view(-1, 1).expand_as(c)
        big = int(c.shape[0]) + 1
        cheap = (c >= 0) & (c <= limit)
        key = torch.where(cheap, big + (int(2 ** 24) - c.to(torch.int64)), pos)
        key = torch.where(c < 0, torch.full_like(key, big + int(2 ** 25)), key)
        idx = torch.sort(key, dim=0, stable=True).indices
    elif rule.startswith("chunk:"):
        # every XCD takes runs of <n> consecutive tiles (flat batches: 256-slot groups of the sorted order, i.e. one
        # compact patch of directions of one camera per run): what is resident on an XCD at a time shares its L2
        n = max(1, int(rule.split(":")[1]))
        rows = int(per_xcd.shape[0])
        pos = torch.arange(rows, device=default.device, dtype=torch.int64).view(-1, 1)
        x = torch.arange(8, device=default.device, dtype=torch.int64).view(1, -1)
        rnd = torch.div(pos, n, rounding_mode="floor")
        width = torch.clamp(rows - rnd * n, max=n)                   # the last round of an XCD may be shorter
        tiles = rnd * (8 * n) + x * width + (pos - rnd * n)
        return torch.where(tiles < nt, tiles, torch.full_like(tiles, nt)).reshape(-1)
    else:
        raise ValueError(f"unknown tile order rule {rule!r}")
    return torch.gather(per_xcd, 0, idx).reshape(-1)


class Pipeline:
    """radfoam::Pipeline as seen from Python (pipeline_bindings.cpp:626-667)."""

    def __init__(self, sh_degree: int, attr_dtype):
        if not isinstance(sh_degree, int) or isinstance(sh_degree, bool):
            raise TypeError("create_pipeline(): sh_degree must be an int")
        self._attr_dtype = _parse_attr_dtype(attr_dtype)
        if sh_degree < 0 or sh_degree > 3:
            raise RuntimeError("Unsupported SH degree")  # pipeline.cu:787,800
        self._lib = _lib.load()  # raises if the HIP library is not built: no fallback
        self._sh_degree = sh_degree
        self._attr_type = _lib.RF_ATTR_FLOAT16 if self._attr_dtype == torch.float16 else _lib.RF_ATTR_FLOAT32
        self._attr_dim = int(self._lib.rf_attribute_dim(sh_degree))
        self._cache = _FoamCache()
        #: reuse the packed foam between calls while the inputs are unchanged
        self.cache_foam = True
        #: 0 auto, 1 per-lane atomics, 2 wave pre-reduced atomics, 3 block cache, 4 direct row atomics
        #: (rf_launch_opts.backward_mode)
        self.backward_mode = 0
        #: how the scans of a launch are scheduled (rf_launch_opts.forward_mode); every mode returns the same results bit
        #: for bit -- the reference's scan, tracing_utils.cuh:43-67.  0 auto, 1 face blocks requested one at a time, 2 the
        #: first six of a cell together (auto picks by launch shape); 3 = every face divided, the way the reference
        #: writes it (the independent instance the filtered scan is tested against; ``strict_reference_scan`` sets it,
        #: 40 % slower); 4 = experiment: persistent waves refilling dead lanes from a queue (ballot + prefix count; slower,
        #: DESIGN.md 4.1); 5 = mode 2 behind a block-level LDS table of cell records and face blocks (auto picks it for
        #: sorted flat batches)
        self.forward_mode = 0
        #: set by trace_backward: True when it replayed the hop trail of its trace_forward, False when it walked again
        self.last_backward_replayed = None
        #: layout of the attr_grad accumulator: "auto" = rows on 64-byte lines at a pitch of 16 / 32 / 64 floats (fewer
        #: atomic line requests per gradient row; trace_backward then returns attr_grad as a [N, A] view of the padded
        #: rows), "dense" = the reference's contiguous [N, A] (radfoam_amd.dist's sparse exchange needs it), or an int
        self.gradient_row_pitch = "auto"
        #: experiment builds only (scripts/): int64 device tensor handed to rf_trace_backward as rf_launch_opts.stats
        self.experiment_stats = None
        #: image-shaped batches: the order in which the blocks of a launch take the 16x16 tiles (rf_launch_opts.tile_order),
        #: from the hop counts of the previous forward over a frame of this shape -- any order gives the same results, the
        #: order decides what is still running when a launch drains ("Work distribution"):
        #:   "auto" (default)  forward / render: for the very rays the order was learnt on (a frame traced again) every XCD
        #:                     keeps the tiles the static dealing gives it but takes them longest first (launches of at
        #:                     most 16384 blocks; larger ones as the backward), for new rays the static dealing; backward:
        #:                     the static order with the cheapest 2048 tiles last, learnt by the forward of the same rays;
        #:                     sorted flat batches (their 256-slot groups): the cheapest eighth last in both launches;
        #:   None / "static"   the static dealing of the kernels (strips of a quarter row per XCD, from both ends of the
        #:                     frame towards its middle);
        #:   "xcd", "tail", "tail:<count>", "global"   one rule for both launches (experiments).
        self.tile_order_mode = "auto"
        #: launches over a frame shape between two learnings of its tile orders when the rays keep changing (an order
        #: learnt on another camera of the same scene is worth as much as the frame's own: scripts/gpu_tile_order_stale.py)
        self.tile_order_refresh = 16
        #: a frame the pipeline has NOT traced may take the forward / render order learnt on the previous frame of its
        #: shape when the two show nearly the same picture -- a camera path (viewer, fly-through, bench.py's 0.05 degrees
        #: per step): five sample rays of the two frames are compared ON THE DEVICE (rf_gate_tile_order: angle between
        #: the directions + shift of the origin relative to its distance from the entry cell's point, every sample within
        #: this many degrees; about one 16-pixel tile of a 1080p frame) and the launch gets the learnt order or the static
        #: dealing accordingly, without a synchronisation.  Another camera of a data set fails the test and runs under the
        #: static dealing as before (an order learnt on other rays is worse than it: profiles/r04/d_tile_order_*).  0: off.
        self.tile_order_coherence_degrees = 0.5
        #: renders (no backward follows) of rays that keep changing learn their order every this many launches
        self.tile_order_refresh_render = 4
        #: image-shaped launches over rays the pipeline has NOT traced before (another camera: benchmark.py:95-139, a new
        #: training view): under "auto" the forward / render takes its block order from a cost PRIOR instead of the static
        #: dealing -- a coarse grid of the foam (cells per unit length, mean density per voxel: rf_build_cost_grid, rebuilt
        #: with the triangulation and every tile_prior_refresh geometry changes) marched by five rays per tile
        #: (rf_estimate_tile_cost), longest estimate first per XCD.  Needs no previous trace of these rays; results do
        #: not depend on it.  False: the static dealing for new rays (rounds 4-5).
        self.tile_prior = False
        self.tile_prior_resolution = 32
        self.tile_prior_refresh = 64
        #: the rule applied to the estimated costs (see tile_order()); None = "xcd:8" for launches of at most 16384
        #: blocks (classes of 8 estimated steps, static order within a class), "tail" above
        self.tile_prior_rule = None
        #: experiments only (scripts/gpu_tile_prior.py): an int32 device tensor that the next image-shaped forward / render
        #: launches take as their block -> tile table, whatever the mode says
        self.experiment_tile_order = None
        self._prior = None          # {"topo": key, "points": key, "grid": tensor, "age": geometry changes since it was built}
        self._prior_keep = None     # the order handed to the launch in flight
        self._defaults = {}         # (height, width, device) -> the static block -> tile table
        self._tiles = None          # the tile orders used last (one entry of _tile_sets)
        self._tile_sets = {}        # frame shape -> its tile orders (a few shapes: training batches, evaluation frames)
        #: trace_forward records the cell every hop enters so that a trace_backward call on the same
        #: inputs replays it instead of re-scanning every cell (rf_launch_opts.trail).  Costs
        #: trail_steps * 4 bytes per ray of HBM (2.1 GB for a 1080p frame at 256 steps), so:
        #:   "auto" (default)  record only when a backward can follow (see _wants_trail: the computed
        #:                     attributes require grad, or the points do and grad mode is on);
        #:                     evaluation / no-grad renders -- RadFoamScene's included, which passes its
        #:                     nn.Parameter points regardless -- neither allocate nor write a trail;
        #:   True / False      always / never (callers that drive trace_backward by hand, like
        #:                     bench.py, set True).
        self.record_trail = "auto"
        #: (what "auto" cannot see from inside another operator's forward -- a caller that optimises ONLY the points or the
        #: rays through an autograd.Function: grad mode is off there and the attributes do not require grad -- that operator
        #: passes as trace_forward's ``record_trail`` argument; radfoam_amd/render.py does)
        #: hops recorded per ray; rays that take more are left to a second launch that walks them again by scanning.
        #: That launch is as long as its longest ray (hundreds of dependent scans): 130 of 1 M rays of the training-shaped
        #: batch take 257-269 hops and cost 1.4 ms of an 8 ms backward.  So the capacity follows the data: every
        #: trace_forward leaves the largest hop count of its batch in pinned host memory (asynchronously, no stream
        #: synchronisation), and the next one that finds it there sizes its trail for it -- within trail_steps_limit AND
        #: within a memory budget (the trail is trail_steps * slots * 4 bytes: one ray that runs into the default
        #: max_intersections = 1024 would otherwise take a 1080p trail from 2.1 to 9.5 GB), and down again when the
        #: batches get shorter (trail_shrink_after consecutive probes that need less than 3/4 of it; never below
        #: trail_steps_floor).  Any capacity is correct: what does not fit is re-walked.  When the probe lands relative to
        #: the next call is a matter of timing, so the capacity of a given step is not reproducible; results are.
        self.trail_steps = 256
        self.trail_steps_floor = 256
        self.trail_steps_limit = 2048
        #: bytes the trail of one batch may take: None = trail_memory_fraction of the device's total memory
        self.trail_memory_limit = None
        self.trail_memory_fraction = 0.04
        self.trail_shrink_after = 8
        self._trail_short_probes = 0
        self._hops_probe = None
        #: drop the trail (and its memory) once a trace_backward has replayed it; off by default because
        #: the allocation is reused by the next trace_forward of the same shape (a training loop), and a
        #: second backward over the same forward (tests, linearity checks) may replay it again
        self.free_trail_after_backward = False
        self._trail = None
        #: flat ray batches (anything but [H, W, 6] images) are traced in a coherent order -- sorted by
        #: entry cell and direction (rf_build_ray_order) -- so that shuffled training batches
        #: (train.py:61) keep the locality the kernels rely on; results do not depend on it
        self.reorder_rays = True
        #: batches smaller than this are traced as they come
        self.reorder_min_rays = 16384
        self._order = None

    @property
    def strict_reference_scan(self) -> bool:
        """True: every scan divides every face and keeps a running minimum of the rounded quotients, the way
        tracing_utils.cuh:43-67 writes it (forward_mode 3) -- same results as the default filtered scan, slower."""
        return int(self.forward_mode) == 3

    @strict_reference_scan.setter
    def strict_reference_scan(self, on):
        self.forward_mode = 3 if on else 0

    def _gradient_pitch(self) -> int:
        """Floats between two rows of the attr_grad accumulator (rf_launch_opts.attr_grad_pitch)."""
        a = self._attr_dim
        if self.gradient_row_pitch == "dense":
            return a
        if self.gradient_row_pitch == "auto":
            return {4: 4, 13: 16, 28: 32, 49: 64}[a]
        pitch = int(self.gradient_row_pitch)
        if pitch < a:
            raise RuntimeError("gradient_row_pitch must be at least the attribute dimension")
        return pitch

    def invalidate(self):
        """Forget the cached packed foam, hop trail and ray order of this Pipeline (and free the trail).
        Needed only after writes ``tensor._version`` cannot see -- see ``invalidate_caches``."""
        self._cache.clear()
        self._trail = None
        self._order = None
        self._prior = None

    #: trace_forward takes the keyword-only ``record_trail`` (radfoam_amd.render.TraceRays passes it)
    accepts_record_trail = True

    def _wants_trail(self, points, attributes, override=None) -> bool:
        """"auto": will a trace_backward follow this forward?  Inside an autograd.Function.forward grad mode is off and
        the inputs keep their requires_grad flags whether or not the caller runs under torch.no_grad(), so the flags
        of a leaf say nothing: RadFoamScene hands its nn.Parameter points to every render, evaluation included.  What
        does tell is a COMPUTED input: the scene's attributes are cat(...softplus(density)...), which require grad
        exactly when the graph is being recorded.  So: attributes.requires_grad, or -- for callers outside a Function,
        where grad mode is the caller's -- points.requires_grad with grad mode on.  A wrong "no" only costs speed
        (trace_backward re-walks instead of replaying), never correctness."""
        if self.record_trail == "auto":
            if override is not None:                # an operator that knows whether a backward follows said so
                return bool(override)
            return bool(attributes.requires_grad or (points.requires_grad and torch.is_grad_enabled()))
        return bool(self.record_trail)

    # -- introspection (Pipeline::attribute_dim / attribute_type, pipeline.cu:768-774) ----------
    def attribute_dim(self) -> int:
        return self._attr_dim

    def attribute_type(self):
        return self._attr_dtype

    @property
    def sh_degree(self) -> int:
        return self._sh_degree

    # -- validation ----------------------------------------------------------------------------
    def _validate_scene_data(self, points, attributes, point_adjacency, point_adjacency_offsets):
        """validate_scene_data, pipeline_bindings.cpp:8-71."""
        if points.size(-1) != 3:
            raise RuntimeError(f"points had dimension {points.size(-1)} along axis -1, expected 3")
        if points.dtype != torch.float32:
            raise RuntimeError(f"points had dtype {_c10_name(points.dtype)}, expected float32")
        if not points.is_cuda:
            raise RuntimeError("points must be on CUDA device")
        num_points = points.numel() // 3
        if attributes.size(-1) != self._attr_dim:
            raise RuntimeError(
                f"attributes had dimension {attributes.size(-1)} along axis -1, expected {self._attr_dim}")
        if attributes.numel() // self._attr_dim != num_points:
            raise RuntimeError("attributes must have the same number of rows as points")
        if attributes.dtype != self._attr_dtype:
            raise RuntimeError(
                f"attributes had dtype {_c10_name(attributes.dtype)}, expected {_dtype_name(self._attr_dtype)}")
        if not attributes.is_cuda:
            raise RuntimeError("attributes must be on CUDA device")
        if point_adjacency_offsets.dtype != torch.uint32:
            raise RuntimeError("point_adjacency_offsets must have uint32 dtype")
        if not point_adjacency_offsets.is_cuda:
            raise RuntimeError("point_adjacency_offsets must be on CUDA device")
        if point_adjacency_offsets.numel() != num_points + 1:
            raise RuntimeError("point_adjacency_offsets must have num_points + 1 elements")
        if point_adjacency.dtype != torch.uint32:
            raise RuntimeError("point_adjacency must have uint32 dtype")
        if not point_adjacency.is_cuda:
            raise RuntimeError("point_adjacency must be on CUDA device")

    @staticmethod
    def _validate_rays(rays, start_point):
        """pipeline_bindings.cpp:139-159 / 304-324."""
        if rays.size(-1) != 6:
            raise RuntimeError("rays must have 6 as the last dimension")
        if rays.dtype != torch.float32:
            raise RuntimeError("rays must have float32 dtype")
        if not rays.is_cuda:
            raise RuntimeError("rays must be on CUDA device")
        num_rays = rays.numel() // 6
        if start_point.numel() != num_rays:
            raise RuntimeError("start_point must have the same batch size as rays")
        if start_point.dtype != torch.uint32:
            raise RuntimeError("start_point must have uint32 dtype")
        if not start_point.is_cuda:
            raise RuntimeError("start_point must be on CUDA device")
        return num_rays

    @staticmethod
    def _settings(weight_threshold, max_intersections):
        s = _lib.TraceSettings(_DEFAULT_WEIGHT_THRESHOLD, _DEFAULT_MAX_INTERSECTIONS)
        if weight_threshold is not None:
            s.weight_threshold = float(weight_threshold)
        if max_intersections is not None:
            mi = int(max_intersections)
            if mi < 0 or mi > 0xFFFFFFFF:
                raise RuntimeError("max_intersections out of range for uint32")
            s.max_intersections = mi
        return s

    # -- foam packing ---------------------------------------------------------------------------
    def _launch_opts(self, points, attributes, adjacency, offsets, rays_shape, ext_diff=None, launch="forward",
                     image=None):
        """Workspace + rf_launch_opts for this call (foam_prepared set on a cache hit)."""
        n = points.numel() // 3
        e = adjacency.numel()
        nbytes = int(self._lib.rf_workspace_bytes(n, e, self._sh_degree, self._attr_type))
        tensors = (points, attributes, adjacency, offsets, ext_diff)
        hit = self.cache_foam and self._cache.lookup(tensors)
        topo = (not hit) and self.cache_foam and self._cache.lookup_topology(tensors)
        ws = self._cache.workspace
        if ws is None or ws.numel() < nbytes or ws.device != points.device:
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=points.device)
            self._cache.workspace = ws
            hit = topo = False
        opts = _lib.LaunchOpts()
        opts.workspace = ws.data_ptr()
        opts.workspace_bytes = ws.numel()
        opts.foam_prepared = 1 if hit else (2 if topo else 0)
        opts.backward_mode = int(self.backward_mode)
        opts.forward_mode = int(self.forward_mode)
        # rays given as an image [H, W, 6]: let a wave own an 8x8 pixel tile (a degenerate "image" such
        # as [B, 1, 6] is a flat batch: tiles of it would be mostly empty)
        if len(rays_shape) == 3 and rays_shape[0] >= 16 and rays_shape[1] >= 16:
            opts.image_height, opts.image_width = int(rays_shape[0]), int(rays_shape[1])
        # The workspace is about to be (re)packed by the C call: until that call has succeeded the cache
        # must not claim it (a failed or never-issued launch would leave an unpacked workspace behind a
        # valid key).  _foam_done() records it afterwards.
        # image: the frame of a trace_benchmark call, whose camera carries the shape (image_width / _height stay 0)
        shape = (opts.image_height, opts.image_width) if opts.image_width else image
        if shape is None and len(rays_shape) >= 1:       # a flat batch: its 256-slot groups, in the traced (sorted) order
            batch_rays = 1
            for d in rays_shape[:-1]:
                batch_rays *= int(d)
            shape = ("flat", batch_rays)
        t = self._tile_sets.get(tuple(shape)) if shape else None
        if t is not None and self.tile_order_mode not in (None, "static") and t["mode"] == self.tile_order_mode and \
                t[launch].device == points.device:
            opts.tile_order = t[launch].data_ptr()
            self._tiles = t
        opts._pending_foam = None
        if not hit:
            opts._pending_foam = tensors if self.cache_foam else ()
            self._cache.key = None
            if opts.foam_prepared == 0:
                self._cache.topo_key = None
        return opts

    def _foam_done(self, opts):
        """The C call that packs the workspace described by `opts` has been issued successfully."""
        pending = opts._pending_foam
        if pending is None:
            return
        if pending:
            self._cache.store(pending)
        else:
            self._cache.clear()
        opts._pending_foam = None

    def prepare_foam(self, points, attributes, point_adjacency, point_adjacency_offsets):
        """Pack the foam now (rf_prepare_foam) so that the next trace_* call on the same tensors
        finds it cached.  Optional: trace_forward / trace_backward pack on demand."""
        points_c, attributes_c = points.contiguous(), attributes.contiguous()
        adjacency_c, offsets_c = point_adjacency.contiguous(), point_adjacency_offsets.contiguous()
        self._validate_scene_data(points, attributes, point_adjacency, point_adjacency_offsets)
        opts = self._launch_opts(points_c, attributes_c, adjacency_c, offsets_c, ())
        if opts.foam_prepared == 2:
            with torch.cuda.device(points_c.device):
                rc = self._lib.rf_prepare_foam_geometry(
                    self._sh_degree, self._attr_type, points_c.size(0), _ptr(points_c), _ptr(attributes_c),
                    adjacency_c.numel(), opts.workspace, opts.workspace_bytes, _stream_ptr(points_c.device))
            _lib.check(rc)
            self._foam_done(opts)
        elif not opts.foam_prepared:
            with torch.cuda.device(points_c.device):
                rc = self._lib.rf_prepare_foam(
                    self._sh_degree, self._attr_type, points_c.size(0), _ptr(points_c), _ptr(attributes_c),
                    adjacency_c.numel(), _ptr(adjacency_c), _ptr(offsets_c), None, opts.workspace,
                    opts.workspace_bytes, _stream_ptr(points_c.device))
            _lib.check(rc)
            self._foam_done(opts)

    # -- hop trail -------------------------------------------------------------------------------
    @staticmethod
    def _tkey(t):
        return _tensor_key(t)

    def _trail_key(self, foam, ray_keys, settings):
        """``ray_keys``: _source_key of rays, start_point and depth_quantiles."""
        return (tuple(self._tkey(t) for t in foam),) + tuple(ray_keys) + (
            float(settings.weight_threshold), int(settings.max_intersections))

    def _ray_order(self, opts, rays_c, start_c, num_rays, key=None):
        """Set opts.ray_order for a flat batch: the permutation rf_build_ray_order computes, cached on
        the identity + version of (rays, start_point) so that trace_backward reuses trace_forward's.  ``key``: the
        callers' (_source_key(rays), _source_key(start_point)) when the contiguous copies are not what they hold."""
        if not self.reorder_rays or opts.image_width or num_rays < self.reorder_min_rays:
            return
        if key is None:
            key = (self._tkey(rays_c), self._tkey(start_c))
        cached = self._order
        if cached is None or cached["key"] != key:
            dev = rays_c.device
            order = torch.empty(num_rays, dtype=torch.int32, device=dev)
            nbytes = int(self._lib.rf_ray_order_workspace_bytes(num_rays))
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = self._lib.rf_build_ray_order(_ptr(rays_c), _ptr(start_c), num_rays, _ptr(order), _ptr(ws),
                                                  ws.numel(), _stream_ptr(dev))
            _lib.check(rc)
            cached = self._order = {"key": key, "order": order, "refs": (rays_c, start_c)}
        opts.ray_order = cached["order"].data_ptr()

    def _probe_hops(self, hops):
        """Largest hop count of the batch just traced -> pinned host memory, without waiting for it."""
        dev = hops.device
        pr = self._hops_probe
        if pr is None or pr["device"] != dev:
            pr = self._hops_probe = {"host": torch.zeros((), dtype=torch.int32).pin_memory(), "event": None,
                                     "device": dev, "fresh": False}
        if pr["event"] is not None and not pr["event"].query():
            return                      # the previous probe has not landed yet: one in flight is enough
        # on the device (and its current stream) the hops were written on, whatever the caller's current device is
        with torch.cuda.device(dev):
            pr["host"].copy_(hops.max(), non_blocking=True)
            pr["event"] = torch.cuda.Event()
            pr["event"].record(torch.cuda.current_stream(dev))
        pr["fresh"] = True

    def _trail_budget_steps(self, slots, dev) -> int:
        """Hops per ray that fit the trail's memory budget for a launch of `slots` thread slots."""
        limit = self.trail_memory_limit
        if limit is None:
            limit = int(self.trail_memory_fraction * torch.cuda.get_device_properties(dev).total_memory)
        return max(1, int(limit) // (4 * max(int(slots), 1)))

    def _fit_trail_steps(self, longest: int) -> int:
        return (int(longest) * 9 // 8 + 31) // 32 * 32

    def _grow_trail_steps(self):
        """Follow the longest ray of the last probed batch: up at once, down after trail_shrink_after short probes."""
        pr = self._hops_probe
        if pr is None or pr["event"] is None or not pr["fresh"] or not pr["event"].query():
            return
        pr["fresh"] = False             # one decision per probe
        want = self._fit_trail_steps(int(pr["host"]))
        cur = int(self.trail_steps)
        if want > cur:
            self.trail_steps = min(want, int(self.trail_steps_limit))
            self._trail_short_probes = 0
        elif cur > int(self.trail_steps_floor) and 4 * want < 3 * cur:
            self._trail_short_probes += 1
            if self._trail_short_probes >= int(self.trail_shrink_after):
                self.trail_steps = max(want, int(self.trail_steps_floor))
                self._trail_short_probes = 0
        else:
            self._trail_short_probes = 0

    def _new_trail(self, opts, num_rays, dev):
        slots = int(self._lib.rf_trail_slots(num_rays, opts.image_width, opts.image_height))
        self._grow_trail_steps()
        cap = max(1, min(int(self.trail_steps), self._trail_budget_steps(slots, dev)))
        old = self._trail
        if old is not None and old["trail"].shape == (cap, slots) and old["trail"].device == dev:
            trail, hops = old["trail"], old["hops"]     # reuse the allocation
        else:
            self._trail = None
            trail = torch.empty((cap, slots), dtype=torch.int32, device=dev)
            hops = torch.empty((slots,), dtype=torch.int32, device=dev)
        opts.trail = trail.data_ptr()
        opts.trail_hops = hops.data_ptr()
        opts.trail_cap = cap
        opts.trail_slots = slots
        return trail, hops

    # -- trace_forward ---------------------------------------------------------------------------
    def trace_forward(self, points, attributes, point_adjacency, point_adjacency_offsets, rays,
                      start_point, depth_quantiles=None, weight_threshold=None,
                      max_intersections=None, return_contribution=False, *, record_trail=None):
        """pipeline_bindings.cpp:107-265 (kernel: src/tracing/pipeline.cu:14-130).  ``record_trail`` (keyword only, not
        in the reference): True / False from a caller that knows whether a trace_backward of these rays will follow
        (overrides the "auto" heuristic of ``Pipeline.record_trail`` for this call); None = the heuristic."""
        points_c = points.contiguous()
        attributes_c = attributes.contiguous()
        adjacency_c = point_adjacency.contiguous()
        offsets_c = point_adjacency_offsets.contiguous()
        rays_c = rays.contiguous()
        start_c = start_point.contiguous()
        self._validate_scene_data(points, attributes, point_adjacency, point_adjacency_offsets)
        num_points = points_c.size(0)
        num_rays = self._validate_rays(rays_c, start_c)
        dev = rays_c.device

        nq = 0
        quantiles_c = None
        if depth_quantiles is not None:
            quantiles_c = depth_quantiles.contiguous()
            nq = quantiles_c.size(-1)
            if quantiles_c.dtype != torch.float32:
                raise RuntimeError("depth_quantiles must have float32 dtype")
            if not quantiles_c.is_cuda:
                raise RuntimeError("depth_quantiles must be on CUDA device")
            if nq == 0 or quantiles_c.numel() // nq != num_rays:
                raise RuntimeError("depth_quantiles must have the same batch size as rays")
        settings = self._settings(weight_threshold, max_intersections)

        batch = tuple(rays_c.shape[:-1])
        rgba = torch.empty(batch + (4,), dtype=self._attr_dtype, device=dev)
        num_intersections = torch.empty(batch + (1,), dtype=torch.uint32, device=dev)
        contribution = None
        if return_contribution:
            contribution = torch.zeros((num_points, 1), dtype=torch.float32, device=dev)
        depth = depth_indices = None
        if quantiles_c is not None:
            depth = torch.zeros(batch + (nq,), dtype=torch.float32, device=dev)
            depth_indices = torch.zeros(batch + (nq,), dtype=torch.uint32, device=dev)

        if num_rays == 0:
            # nothing is launched, so nothing may be claimed: the foam cache is left exactly as it was
            # (rf_trace_forward would pack the workspace even for an empty batch; skipping the call
            # keeps "the cache key describes the workspace" independent of that)
            self._trail = None
            out = {"rgba": rgba}
            if quantiles_c is not None:
                out["depth"], out["depth_indices"] = depth, depth_indices
            if return_contribution:
                out["contribution"] = contribution.to(self._attr_dtype)
            out["num_intersections"] = num_intersections
            return out

        opts = self._launch_opts(points_c, attributes_c, adjacency_c, offsets_c, rays_c.shape)
        ray_keys = (_source_key(rays, rays_c), _source_key(start_point, start_c), _source_key(depth_quantiles, quantiles_c))
        self._ray_order(opts, rays_c, start_c, num_rays, key=ray_keys[:2])
        trail = None
        if self._wants_trail(points, attributes, record_trail):
            trail = self._new_trail(opts, num_rays, dev)
        tiles_pending = None
        if opts.image_width:
            tiles_pending = self._tile_cost_begin(
                opts, opts.image_height, opts.image_width, ray_keys[:2], dev, backward_follows=trail is not None,
                prior={"rays": rays_c, "camera": None, "foam": (points_c, attributes_c, adjacency_c, offsets_c),
                       "settings": settings, "start": start_c})
        elif opts.ray_order:
            tiles_pending = self._tile_cost_begin(opts, "flat", num_rays, ray_keys[:2], dev)
        with torch.cuda.device(dev):
            rc = self._lib.rf_trace_forward(
                self._sh_degree, self._attr_type, C.byref(settings), num_points, _ptr(points_c),
                _ptr(attributes_c), adjacency_c.numel(), _ptr(adjacency_c), _ptr(offsets_c), num_rays,
                _ptr(rays_c), _ptr(start_c), nq, _ptr(quantiles_c), _ptr(rgba), _ptr(depth),
                _ptr(depth_indices), _ptr(num_intersections), _ptr(contribution), C.byref(opts),
                _stream_ptr(dev))
        _lib.check(rc)
        self._foam_done(opts)
        if trail is not None:
            self._probe_hops(trail[1])
            foam = (points_c, attributes_c, adjacency_c, offsets_c)
            self._trail = {
                "key": self._trail_key(foam, ray_keys, settings),
                # keep the keyed storages from being recycled (the caller's views as well as their contiguous copies)
                "refs": foam + (rays_c, start_c, quantiles_c, rays, start_point, depth_quantiles),
                "trail": trail[0], "hops": trail[1], "cap": opts.trail_cap, "slots": opts.trail_slots,
                "order": opts.ray_order,   # the slot -> ray mapping the trail was recorded under
            }
        else:
            self._trail = None

        self._tile_cost_end(tiles_pending)

        out = {"rgba": rgba}
        if quantiles_c is not None:
            out["depth"] = depth
            out["depth_indices"] = depth_indices
        if return_contribution:
            out["contribution"] = contribution.to(self._attr_dtype)
        out["num_intersections"] = num_intersections
        return out

    def _default_tiles(self, height, width, dev):
        """The static block -> tile table of a launch shape (rf_launch_blocks), as an int64 device tensor."""
        k = (height, width, dev)
        d = self._defaults.get(k)
        if d is None:
            dims = (width, 0, 0) if height == "flat" else (height * width, width, height)
            nb = int(self._lib.rf_launch_blocks(*dims, None))
            host = (C.c_uint32 * nb)()
            self._lib.rf_launch_blocks(*dims, host)
            d = torch.tensor(list(host), dtype=torch.int64, device=dev)
            while len(self._defaults) >= 16:
                self._defaults.pop(next(iter(self._defaults)))
            self._defaults[k] = d
        return d

    def _prior_grid(self, foam):
        """The coarse cost grid of this foam (rf_build_cost_grid), rebuilt when the adjacency changes (a triangulation
        rebuild: points were added, pruned or reordered) or after tile_prior_refresh changes of the geometry."""
        points_c, attributes_c, adjacency_c, offsets_c = foam
        topo = (_tensor_key(adjacency_c), _tensor_key(offsets_c), tuple(points_c.shape), points_c.device,
                int(self.tile_prior_resolution))
        pkey = (_tensor_key(points_c), _tensor_key(attributes_c))
        pr = self._prior
        if pr is not None and pr["topo"] == topo:
            if pr["points"] == pkey:
                return pr["grid"]
            pr["age"] += 1
            pr["points"] = pkey
            if pr["age"] < int(self.tile_prior_refresh):
                return pr["grid"]
        res = int(self.tile_prior_resolution)
        nbytes = int(self._lib.rf_cost_grid_bytes(res))
        grid = pr["grid"] if (pr is not None and pr["grid"].numel() == nbytes and pr["grid"].device == points_c.device) \
            else torch.empty(nbytes, dtype=torch.uint8, device=points_c.device)
        with torch.cuda.device(points_c.device):
            rc = self._lib.rf_build_cost_grid(_ptr(points_c), _ptr(attributes_c), self._attr_type, self._attr_dim,
                                              points_c.size(0), res, _ptr(grid), nbytes, _stream_ptr(points_c.device))
        _lib.check(rc)
        self._prior = {"topo": topo, "points": pkey, "grid": grid, "age": 0, "refs": (adjacency_c, offsets_c)}
        return grid

    def estimate_tile_cost(self, foam, height, width, rays=None, camera=None, settings=None):
        """int32 [tiles]: estimated steps of every 16x16 tile's longest ray (rf_estimate_tile_cost) -- what a forward
        over these rays would report as rf_launch_opts.tile_cost, without tracing anything."""
        dev = foam[0].device
        grid = self._prior_grid(foam)
        settings = settings or self._settings(None, None)
        tiles = ((height + 15) // 16) * ((width + 15) // 16)
        cost = torch.empty(tiles, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = self._lib.rf_estimate_tile_cost(_ptr(grid), int(self.tile_prior_resolution), _ptr(rays),
                                                 C.byref(camera) if camera is not None else None, int(width), int(height),
                                                 float(settings.weight_threshold), int(settings.max_intersections),
                                                 _ptr(cost), _stream_ptr(dev))
        _lib.check(rc)
        return cost

    def _prior_tile_order(self, height, width, dev, prior):
        cost = self.estimate_tile_cost(prior["foam"], height, width, rays=prior["rays"], camera=prior["camera"],
                                       settings=prior["settings"])
        default = self._default_tiles(height, width, dev)
        rule = self.tile_prior_rule or ("xcd:8" if default.numel() <= 16384 else "tail")
        return tile_order(cost, default, rule).to(torch.int32).contiguous()

    def _tile_cost_begin(self, opts, height, width, key, dev, backward_follows=False, prior=None):
        """Decide what this launch does about tile orders.  Returns what _tile_cost_end needs when the launch is to report
        the cost of its tiles (rf_launch_opts.tile_cost), else None.

        "auto" (measured on an asymmetric scene with a new camera every launch, profiles/r04/b_tile_order_asymmetric_*):
        an order learnt on OTHER rays makes the forward and the render 2-6 % slower than the static dealing (the longest
        tiles of another camera are not this camera's), while the backward's rule -- the cheapest tiles last -- still gains
        2-6 % with it.  So a forward / render launch takes a learnt order only for the very rays it was learnt on (a frame
        that is traced again: benchmark loops, evaluation of a fixed view), and learns -- for the trace_backward that
        follows, and for a possible repeat -- whenever the rays are new and a backward will follow; without a backward
        (renders) it learns every tile_order_refresh launches at most."""
        mode = self.tile_order_mode
        if mode in (None, "static") or (height != "flat" and (height < 16 or width < 16)):
            return None
        t = self._tile_sets.get((height, width))
        known = t is not None and t["mode"] == mode and t["default"].device == dev
        same = known and t["key"] == key
        if known:
            t["age"] += 1
        if mode == "auto" and not same and height != "flat":
            # rays this pipeline has not traced: another camera's measured order does not transfer (the static dealing is
            # better than it) -- unless the frame is the next one of a camera path (tile_order_coherence_degrees: decided
            # on the device); else the order a cost prior of THESE rays gives (tile_prior, off by default: measured out)
            opts.tile_order = None
            self._prior_keep = None
            if known and t.get("ref") is not None and float(self.tile_order_coherence_degrees) > 0.0 and prior is not None:
                order = torch.empty(t["forward"].numel(), dtype=torch.int32, device=dev)
                cam = prior["camera"]
                with torch.cuda.device(dev):
                    rc = self._lib.rf_gate_tile_order(
                        _ptr(prior["rays"]), C.byref(cam) if cam is not None else None, _ptr(t["ref"]), int(width),
                        int(height), float(self.tile_order_coherence_degrees) * 0.017453292519943295, _ptr(t["forward"]),
                        _ptr(order), _ptr(t["verdict"]), _stream_ptr(dev))
                _lib.check(rc)
                opts.tile_order = order.data_ptr()
                self._prior_keep = order
            elif self.tile_prior and prior is not None:
                order = self._prior_tile_order(height, width, dev, prior)
                opts.tile_order = order.data_ptr()
                self._prior_keep = order
        if self.experiment_tile_order is not None and height != "flat":
            opts.tile_order = self.experiment_tile_order.data_ptr()
        refresh = int(self.tile_order_refresh if (backward_follows or height == "flat") else self.tile_order_refresh_render)
        if known and (same or (t["age"] < refresh and not (mode == "auto" and backward_follows and height != "flat"))):
            return None
        tiles = (width + 255) // 256 if height == "flat" else ((height + 15) // 16) * ((width + 15) // 16)
        cost = torch.zeros(tiles, dtype=torch.int32, device=dev)
        opts.tile_cost = cost.data_ptr()
        return (height, width, key, cost, prior)

    def _tile_cost_end(self, pending):
        """Tile orders for the next launches over a frame of this shape (all on the device, no synchronisation): a tile
        costs what its longest ray took.  Until the rays change again the orders are reused -- and used for other rays of
        the same frame shape meanwhile: any order is correct."""
        if pending is None:
            return
        height, width, key, cost, prior = pending
        mode = self.tile_order_mode
        dev = cost.device
        t = self._tile_sets.get((height, width))
        if t is None or t["default"].device != dev:
            t = {"shape": (height, width), "default": self._default_tiles(height, width, dev)}
        default = t["default"]
        if mode != "auto":
            rules = (mode, mode)
        elif height == "flat":
            # the 256-slot groups of a sorted batch: the cheapest eighth last, in both launches.  What one batch teaches
            # holds for the next batches of the same cameras (the sort key is camera, then direction): on the training
            # batch of bench.py forward + backward 11.35 ms with the static order, 10.88 with its own, 11.02 with
            # another batch's (scripts/gpu_flat_tile_order.py); longest-first within the XCDs does not transfer
            rules = ("tail:%d" % max(8, int(cost.numel()) // 8),) * 2
        else:
            rules = ("xcd" if default.numel() <= 16384 else "tail", "tail")
        orders = self._build_orders(cost, default, height, width, rules)
        t.update(key=key, mode=mode, age=0, forward=orders[0], backward=orders[1])
        # what a later frame is compared with before it may borrow this frame's order (rf_gate_tile_order)
        if height != "flat" and prior is not None and prior.get("start") is not None:
            ref = t.get("ref")
            if ref is None or ref.device != dev:
                ref = torch.empty(40, dtype=torch.float32, device=dev)
                t["verdict"] = torch.zeros(1, dtype=torch.int32, device=dev)
            cam = prior["camera"]
            with torch.cuda.device(dev):
                rc = self._lib.rf_tile_order_reference(
                    _ptr(prior["rays"]), C.byref(cam) if cam is not None else None, _ptr(prior["start"]),
                    _ptr(prior["foam"][0]), int(width), int(height), _ptr(ref), _stream_ptr(dev))
            _lib.check(rc)
            t["ref"] = ref
        else:
            t["ref"] = None
        self._tile_sets.pop((height, width), None)
        while len(self._tile_sets) >= 8:                    # a handful of frame shapes at most
            self._tile_sets.pop(next(iter(self._tile_sets)))
        self._tile_sets[(height, width)] = t
        self._tiles = t

    @staticmethod
    def _device_rule(rule):
        """(rule id, parameter) of rf_build_tile_orders for a tile_order() rule it implements, else None."""
        if rule == "xcd":
            return 1, 1
        if rule.startswith("xcd:"):
            return 1, max(1, int(rule.split(":")[1]))
        if rule == "tail":
            return 2, 2048
        if rule.startswith("tail:"):
            return 2, max(1, int(rule.split(":")[1]))
        return None

    def _build_orders(self, cost, default, height, width, rules):
        """The two block -> tile tables (forward, backward) of `rules` from a cost map: one launch of rf_build_tile_orders
        (rounds 4-5: two dozen torch operations, 0.2-0.25 ms on the launch stream after every forward that learnt); the
        torch path remains for the experiment rules ("global", "chunk:n") and for launches beyond 65536 blocks."""
        dev = cost.device
        nb = int(default.numel())
        dr = [self._device_rule(r) for r in rules]
        if all(r is not None for r in dr) and nb <= 65536:
            a = torch.empty(nb, dtype=torch.int32, device=dev)
            b = a if rules[1] == rules[0] else torch.empty(nb, dtype=torch.int32, device=dev)
            flat = height == "flat"
            num_rays = int(width) if flat else int(height) * int(width)
            with torch.cuda.device(dev):
                rc = self._lib.rf_build_tile_orders(
                    _ptr(cost), num_rays, 0 if flat else int(width), 0 if flat else int(height), dr[0][0], dr[0][1], _ptr(a),
                    dr[1][0], dr[1][1], None if b is a else _ptr(b), _stream_ptr(dev))
            _lib.check(rc)
            return a, b
        made = {rule: tile_order(cost, default, rule).to(torch.int32).contiguous() for rule in set(rules)}
        return made[rules[0]], made[rules[1]]

    # -- trace_backward --------------------------------------------------------------------------
    def trace_backward(self, points, attributes, point_adjacency, point_adjacency_offsets, rays,
                       start_point, rgb_out, grad_in, depth_quantiles=None, depth_indices=None,
                       depth_grad_in=None, ray_error=None, weight_threshold=None,
                       max_intersections=None):
        """pipeline_bindings.cpp:267-497 (kernel: src/tracing/pipeline.cu:132-343)."""
        points_c = points.contiguous()
        attributes_c = attributes.contiguous()
        adjacency_c = point_adjacency.contiguous()
        offsets_c = point_adjacency_offsets.contiguous()
        rays_c = rays.contiguous()
        start_c = start_point.contiguous()
        self._validate_scene_data(points, attributes, point_adjacency, point_adjacency_offsets)
        num_points = points_c.size(0)
        num_rays = self._validate_rays(rays_c, start_c)
        dev = rays_c.device

        grad_c = grad_in.contiguous()
        if grad_c.size(-1) != 4:
            raise RuntimeError("rgb_grad_in must have 4 as the last dimension")
        if grad_c.dtype != self._attr_dtype:
            raise RuntimeError(
                f"rgb_grad_in had dtype {_c10_name(grad_c.dtype)}, expected {_dtype_name(self._attr_dtype)}")
        if not grad_c.is_cuda:
            raise RuntimeError("rgb_grad_in must be on CUDA device")
        if grad_c.numel() // 4 != num_rays:
            raise RuntimeError("rgb_grad_in must have the same batch size as rays")
        # the reference passes rgb_out.data_ptr() unchecked (pipeline_bindings.cpp:479); a wrong
        # dtype/size there is undefined behaviour, here it is an error
        rgb_out_c = rgb_out.contiguous()
        if rgb_out_c.dtype != self._attr_dtype or rgb_out_c.numel() != num_rays * 4 or not rgb_out_c.is_cuda:
            raise RuntimeError("rgb_out must be the rgba returned by trace_forward for these rays")

        nq = 0
        quantiles_c = indices_c = depth_grad_c = None
        if depth_quantiles is not None:
            quantiles_c = depth_quantiles.contiguous()
            nq = quantiles_c.size(-1)
            if quantiles_c.dtype != torch.float32:
                raise RuntimeError("depth_quantiles must have float32 dtype")
            if not quantiles_c.is_cuda:
                raise RuntimeError("depth_quantiles must be on CUDA device")
            if quantiles_c.numel() != num_rays * nq:
                raise RuntimeError("depth_quantiles must have the same batch size as rays")
            if depth_grad_in is None:
                raise RuntimeError("depth_grad must be provided if depth_quantiles is provided")
            if depth_indices is None:
                raise RuntimeError("depth_indices must be provided if depth_quantiles is provided")
            indices_c = depth_indices.contiguous()
            if indices_c.dtype != torch.uint32:
                raise RuntimeError("depth_indices must have uint32 dtype")
            if not indices_c.is_cuda:
                raise RuntimeError("depth_indices must be on CUDA device")
            if indices_c.numel() != num_rays * nq:
                raise RuntimeError("depth_indices must have the same batch size as rays")
            depth_grad_c = depth_grad_in.contiguous()
            if depth_grad_c.size(-1) != nq:
                raise RuntimeError(
                    "depth_grad must have the same number of depth quantiles as depth_quantiles")
            if depth_grad_c.dtype != torch.float32:
                raise RuntimeError(f"depth_grad had dtype {_c10_name(depth_grad_c.dtype)}, expected float32")
            if not depth_grad_c.is_cuda:
                raise RuntimeError("depth_grad must be on CUDA device")
            if depth_grad_c.numel() != num_rays * nq:
                raise RuntimeError("depth_grad must have the same batch size as rays")

        ray_error_c = point_error = None
        if ray_error is not None:
            ray_error_c = ray_error.contiguous()
            if ray_error_c.dtype != self._attr_dtype:
                raise RuntimeError(
                    f"ray_error had dtype {_c10_name(ray_error_c.dtype)}, expected {_dtype_name(self._attr_dtype)}")
            if not ray_error_c.is_cuda:
                raise RuntimeError("ray_error must be on CUDA device")
            if ray_error_c.numel() != num_rays:
                raise RuntimeError("ray_error must have the same batch size as rays")
            point_error = torch.zeros((num_points, 1), dtype=torch.float32, device=dev)
        settings = self._settings(weight_threshold, max_intersections)

        # one flat fp32 buffer [points_grad | attr_grad] so a data-parallel caller can all-reduce
        # both with a single collective (radfoam_amd/dist.py).  The attr_grad rows sit on 64-byte lines at a pitch of
        # 16 / 32 / 64 floats (gradient_row_pitch "auto"): a row leaves the kernels as one atomic instruction and the
        # memory side works per (instruction, line) -- see rf_launch_opts.attr_grad_pitch; the returned attr_grad is then a
        # [N, A] VIEW of [N, pitch] rows (same values; .contiguous() gives the reference's dense layout).
        a = self._attr_dim
        pitch = self._gradient_pitch()
        head = (num_points * 3 + 15) // 16 * 16 if pitch != a else num_points * 3
        flat = torch.zeros(head + num_points * pitch, dtype=torch.float32, device=dev)
        points_grad = flat[: num_points * 3].view(num_points, 3)
        attr_grad = flat[head:].view(num_points, pitch)[:, :a]
        ray_grad = torch.zeros_like(rays_c)

        out = {
            "points_grad": points_grad,
            "attr_grad": attr_grad,
            "ray_grad": ray_grad,
            # extra key (not in the reference): the flat fp32 [points_grad | attr_grad] buffer the
            # two views above alias, for a single gradient all-reduce (radfoam_amd/dist.py)
            "flat_grad": flat,
        }
        if num_rays == 0:   # as in trace_forward: no launch, no claim on the foam cache
            if self._attr_dtype != torch.float32:
                out["attr_grad"] = attr_grad.to(self._attr_dtype)
            if ray_error is not None:
                out["point_error"] = point_error.to(self._attr_dtype)
            return out

        opts = self._launch_opts(points_c, attributes_c, adjacency_c, offsets_c, rays_c.shape, launch="backward")
        ray_keys = (_source_key(rays, rays_c), _source_key(start_point, start_c), _source_key(depth_quantiles, quantiles_c))
        self._ray_order(opts, rays_c, start_c, num_rays, key=ray_keys[:2])
        tr = self._trail
        self.last_backward_replayed = False
        if tr is not None and tr["order"] == opts.ray_order and \
                tr["key"] == self._trail_key((points_c, attributes_c, adjacency_c, offsets_c), ray_keys, settings):
            self.last_backward_replayed = True
            opts.trail = tr["trail"].data_ptr()
            opts.trail_hops = tr["hops"].data_ptr()
            opts.trail_cap = tr["cap"]
            opts.trail_slots = tr["slots"]
        if self.experiment_stats is not None:
            opts.stats = self.experiment_stats.data_ptr()
        opts.attr_grad_pitch = pitch
        with torch.cuda.device(dev):
            rc = self._lib.rf_trace_backward(
                self._sh_degree, self._attr_type, C.byref(settings), num_points, _ptr(points_c),
                _ptr(attributes_c), adjacency_c.numel(), _ptr(adjacency_c), _ptr(offsets_c), num_rays,
                _ptr(rays_c), _ptr(start_c), nq, _ptr(quantiles_c), _ptr(indices_c), _ptr(rgb_out_c),
                _ptr(grad_c), _ptr(depth_grad_c), _ptr(ray_error_c), _ptr(ray_grad), _ptr(points_grad),
                _ptr(attr_grad), _ptr(point_error), C.byref(opts), _stream_ptr(dev))
        _lib.check(rc)
        self._foam_done(opts)
        if self.free_trail_after_backward:
            self._trail = None

        if self._attr_dtype != torch.float32:
            out["attr_grad"] = attr_grad.to(self._attr_dtype)
        if ray_error is not None:
            out["point_error"] = point_error.to(self._attr_dtype)
        return out

    # -- trace_benchmark -------------------------------------------------------------------------
    def trace_benchmark(self, points, attributes, point_adjacency, point_adjacency_offsets,
                        adjacent_diff, camera, start_point, output_rgba, weight_threshold=None,
                        max_intersections=None):
        """pipeline_bindings.cpp:499-585 (kernel: src/tracing/pipeline.cu:472-544)."""
        points_c = points.contiguous()
        attributes_c = attributes.contiguous()
        adjacency_c = point_adjacency.contiguous()
        offsets_c = point_adjacency_offsets.contiguous()
        diff_c = adjacent_diff.contiguous()
        self._validate_scene_data(points, attributes, point_adjacency, point_adjacency_offsets)
        num_points = points_c.size(0)

        cam = _lib.Camera()
        for key in ("position", "forward", "up", "right"):
            v = torch.as_tensor(camera[key]).detach().to("cpu", torch.float32).reshape(-1)
            if v.numel() < 3:
                raise RuntimeError(f"camera['{key}'] must have 3 elements")
            setattr(cam, key, (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2])))
        cam.fov = float(camera["fov"])
        cam.width = int(camera["width"])
        cam.height = int(camera["height"])
        model = camera["model"]
        if model == "pinhole":
            cam.model = 0
        elif model == "fisheye":
            cam.model = 1
        else:
            raise RuntimeError("Invalid camera model")

        if start_point.numel() != 1:
            raise RuntimeError("start_point must have a single element")
        if start_point.dtype != torch.uint32:
            raise RuntimeError("start_point must have uint32 dtype")
        if not start_point.is_cuda:
            raise RuntimeError("start_point must be on CUDA device")
        if output_rgba.numel() != cam.width * cam.height:
            raise RuntimeError("output_rgba must have width * height elements")
        if output_rgba.dtype != torch.uint32:
            raise RuntimeError("output_rgba must have uint32 dtype")
        if not output_rgba.is_cuda:
            raise RuntimeError("output_rgba must be on CUDA device")
        # unchecked in the reference (reinterpret_cast of data_ptr, pipeline_bindings.cpp:581)
        if diff_c.dtype != torch.float16 or not diff_c.is_cuda or diff_c.numel() < 4 * adjacency_c.numel():
            raise RuntimeError("adjacent_diff must be a float16 CUDA tensor with 4 values per adjacency entry")
        if not output_rgba.is_contiguous():
            raise RuntimeError("output_rgba must be contiguous (it is written in place)")
        settings = self._settings(weight_threshold, max_intersections)

        opts = self._launch_opts(points_c, attributes_c, adjacency_c, offsets_c, (), ext_diff=diff_c,
                                 image=(cam.height, cam.width))
        dev = points_c.device
        cam_key = ("camera", bytes(cam), self._tkey(start_point))
        tiles_pending = self._tile_cost_begin(
            opts, cam.height, cam.width, cam_key, dev,
            prior={"rays": None, "camera": cam, "foam": (points_c, attributes_c, adjacency_c, offsets_c), "settings": settings,
                   "start": start_point})
        with torch.cuda.device(dev):
            rc = self._lib.rf_trace_benchmark(
                self._sh_degree, self._attr_type, C.byref(settings), num_points, _ptr(points_c),
                _ptr(attributes_c), adjacency_c.numel(), _ptr(adjacency_c), _ptr(offsets_c), _ptr(diff_c),
                C.byref(cam), _ptr(start_point), _ptr(output_rgba), C.byref(opts), _stream_ptr(dev))
        _lib.check(rc)
        self._foam_done(opts)
        self._tile_cost_end(tiles_pending)
        return None

    # -- extras (no reference counterpart) -------------------------------------------------------
    def walk_statistics(self, points, attributes, point_adjacency, point_adjacency_offsets, rays,
                        start_point, weight_threshold=None, max_intersections=None, extra_slots=0,
                        visit_marks=False):
        """Exact walk counters of one forward pass (cells/faces scanned, hops, segments, lit
        segments) -- the inputs of the algorithmic-bytes figure in bench.py.
        ``visit_marks``: also return ``"visited"``, a bool [N] device tensor of the cells any ray scanned
        (the distinct cells behind bench.py's compulsory-traffic floor)."""
        points_c, attributes_c = points.contiguous(), attributes.contiguous()
        adjacency_c, offsets_c = point_adjacency.contiguous(), point_adjacency_offsets.contiguous()
        rays_c, start_c = rays.contiguous(), start_point.contiguous()
        self._validate_scene_data(points, attributes, point_adjacency, point_adjacency_offsets)
        num_rays = self._validate_rays(rays_c, start_c)
        dev = rays_c.device
        settings = self._settings(weight_threshold, max_intersections)
        stats = torch.zeros(8 + int(extra_slots), dtype=torch.int64, device=dev)
        rgba = torch.empty(tuple(rays_c.shape[:-1]) + (4,), dtype=self._attr_dtype, device=dev)
        opts = self._launch_opts(points_c, attributes_c, adjacency_c, offsets_c, rays_c.shape)
        # the order trace_forward would use: the counters that depend on it (wave_steps, hence lane utilisation) then
        # describe the launch that is timed, not the batch as the caller shuffled it
        self._ray_order(opts, rays_c, start_c, num_rays)
        opts.stats = stats.data_ptr()
        marks = None
        if visit_marks:
            marks = torch.zeros(points_c.size(0), dtype=torch.uint8, device=dev)
            opts.visit_marks = marks.data_ptr()
        with torch.cuda.device(dev):
            rc = self._lib.rf_trace_forward(
                self._sh_degree, self._attr_type, C.byref(settings), points_c.size(0), _ptr(points_c),
                _ptr(attributes_c), adjacency_c.numel(), _ptr(adjacency_c), _ptr(offsets_c), num_rays,
                _ptr(rays_c), _ptr(start_c), 0, None, _ptr(rgba), None, None, None, None,
                C.byref(opts), _stream_ptr(dev))
        _lib.check(rc)
        self._foam_done(opts)
        s = stats[:8].cpu().tolist()
        if extra_slots:
            self.last_raw_statistics = stats.cpu()   # experiment builds (scripts/) append records
        out = {"cells_scanned": s[0], "faces_scanned": s[1], "hops": s[2], "segments": s[3],
               "segments_lit": s[4], "num_rays": num_rays, "wave_steps": s[6]}
        if marks is not None:
            out["visited"] = marks.bool()
        return out

    def build_adjacent_diff(self, points, point_adjacency, point_adjacency_offsets):
        """half4 neighbour-offset table [E,4] (prefetch_adjacent_diff, pipeline.cu:546-586; the
        table benchmark.py:44-54 builds in torch)."""
        points_c = points.contiguous()
        adjacency_c, offsets_c = point_adjacency.contiguous(), point_adjacency_offsets.contiguous()
        if points_c.dtype != torch.float32 or not points_c.is_cuda:
            raise RuntimeError("points must be a float32 CUDA tensor")
        diff = torch.empty((adjacency_c.numel(), 4), dtype=torch.float16, device=points_c.device)
        with torch.cuda.device(points_c.device):
            rc = self._lib.rf_build_adjacent_diff(
                _ptr(points_c), points_c.size(0), adjacency_c.numel(), _ptr(adjacency_c), _ptr(offsets_c),
                _ptr(diff), _stream_ptr(points_c.device))
        _lib.check(rc)
        return diff


def create_pipeline(sh_degree, attr_dtype="float32") -> Pipeline:
    """radfoam.create_pipeline (pipeline_bindings.cpp:587-590,669-672; pipeline.cu:776-805)."""
    from . import dist as _dist      # (after radfoam_amd.dist.enable_data_parallel(): the data-parallel wrapper)
    return _dist.wrap_pipeline(Pipeline(sh_degree, attr_dtype))
