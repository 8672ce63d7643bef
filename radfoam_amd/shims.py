"""Names of the reference's ``radfoam`` module that sit beside the hot path.

SURVEY.md section 8(b)/(f): the north star keeps the Delaunay build and the entry-cell lookup
off the GPU "for now"; these are plain torch/scipy restatements so that the reference's
radfoam_model/scene.py and data_loader import and run against this package.  They are NOT
part of the measured path and make no performance claim.

  Triangulation / TriangulationFailedError   torch_bindings/triangulation_bindings.cpp:225-237,222
  build_aabb_tree                            :117-140
  nn                                         :142-181
  farthest_neighbor                          :183-217  (kernel: src/delaunay/triangulation_ops.cu:9-44)
  BatchFetcher                               torch_bindings/torch_bindings.cpp:32-63,77-83
  run_with_viewer / Viewer                   pipeline_bindings.cpp:592-624 (unsupported: no display)
"""
from __future__ import annotations

import numpy as np
import torch

from . import foam as _foam


from .triangulation import TriangulationFailedError  # noqa: E402  (one class for both triangulations)


class Triangulation:
    """``radfoam.Triangulation(points)``.  CUDA float32 points get the GPU triangulation
    (radfoam_amd/triangulation.py: kd-order, AABB tree and one Delaunay star per point in HIP); CPU tensors get
    this class, a Qhull stand-in with the same protocol: construction kd-sorts the points and triangulates them
    in that order; ``permutation()`` tells the caller how to reorder its own arrays; ``rebuild(points)`` expects
    points already in that order and returns whether a new permutation has to be applied.
    """

    def __new__(cls, points: torch.Tensor):
        if cls is Triangulation and isinstance(points, torch.Tensor) and points.is_cuda:
            from .triangulation import Triangulation as GpuTriangulation
            return GpuTriangulation(points)   # not an instance of cls: __init__ below is not run
        return super().__new__(cls)

    def __init__(self, points: torch.Tensor):
        if points.dim() != 2 or points.size(-1) != 3:
            raise RuntimeError("points must have shape [N, 3]")
        self._device = points.device
        pts = points.detach().to("cpu", torch.float32).numpy()
        if not np.isfinite(pts).all():
            raise TriangulationFailedError("points contain non-finite values")
        self._perm = _foam.kd_order(pts)
        self._triangulate(np.ascontiguousarray(pts[self._perm]))

    def _triangulate(self, pts_sorted: np.ndarray):
        from scipy.spatial import Delaunay, QhullError

        try:
            tri = Delaunay(pts_sorted.astype(np.float64))
        except QhullError as exc:  # degenerate input
            raise TriangulationFailedError(str(exc)) from exc
        indptr, indices = tri.vertex_neighbor_vertices
        n = pts_sorted.shape[0]
        if len(indptr) != n + 1 or (np.diff(indptr) == 0).any():
            raise TriangulationFailedError("triangulation dropped points (duplicates or degenerate input)")
        self._tets = np.ascontiguousarray(tri.simplices.astype(np.uint32))
        if self._device.type == "cuda":
            # tetrahedra -> CSR on the GPU (find_adjacency, delaunay.cu:140-229)
            from . import scene_ops
            adj, off = scene_ops.adjacency_from_tets(torch.from_numpy(self._tets.view(np.int32)).to(self._device), n)
            self._adjacency = adj.cpu().numpy()
            self._offsets = off.cpu().numpy()
        else:
            rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
            order = np.lexsort((np.asarray(indices, dtype=np.int64), rows))
            self._adjacency = np.asarray(indices, dtype=np.int64)[order].astype(np.uint32)
            self._offsets = np.asarray(indptr, dtype=np.int64).astype(np.uint32)
        self._tet_adjacency = np.ascontiguousarray(tri.neighbors.astype(np.int64).astype(np.uint32))
        self._vert_to_tet = np.ascontiguousarray(tri.vertex_to_simplex.astype(np.uint32))

    def _t(self, a):
        return torch.from_numpy(a).to(self._device)

    def permutation(self):
        return self._t(self._perm.astype(np.uint32))

    def rebuild(self, points: torch.Tensor, incremental: bool = False) -> bool:
        del incremental  # always a full rebuild
        pts = points.detach().to("cpu", torch.float32).numpy()
        if not np.isfinite(pts).all():
            raise TriangulationFailedError("points contain non-finite values")
        perm = _foam.kd_order(pts)
        needs_permute = not np.array_equal(perm, np.arange(len(perm)))
        self._perm = perm
        self._triangulate(np.ascontiguousarray(pts[perm]))
        # a rebuilt triangulation invalidates whatever the tracer packed from the old adjacency, even if a
        # caller (like the reference's from_blob getters) hands the new lists out at the old addresses
        from .pipeline import invalidate_caches
        invalidate_caches()
        return bool(needs_permute)

    def point_adjacency(self):
        return self._t(self._adjacency)

    def point_adjacency_offsets(self):
        return self._t(self._offsets)

    def tets(self):
        return self._t(self._tets)

    def tet_adjacency(self):
        return self._t(self._tet_adjacency)

    def vert_to_tet(self):
        return self._t(self._vert_to_tet)


def build_aabb_tree(points: torch.Tensor) -> torch.Tensor:
    """Tensor of the reference's shape [pow2_round_up(N), 2, 3].  CUDA float32 points: the reference's
    tree itself (bit-identical boxes; the GPU triangulation searches it).  CPU tensors: the same tree, by torch
    pooling (only ``nn`` would consume it, and this package's ``nn`` is an exact brute-force search).
    """
    if points.size(-1) != 3:
        raise RuntimeError("points must have 3 as the last dimension")
    if points.dim() != 2:
        raise RuntimeError("points must have 2 dimensions")
    if points.is_cuda and points.dtype == torch.float32:
        from . import triangulation
        return triangulation.build_aabb_tree(points)   # the reference's tree, HIP (rf_build_aabb_tree)
    # CPU tensors: the same tree (build_leaves_kernel / build_tree_kernel, aabb_tree.cu:192-283) by pooling: level d
    # (2^d nodes) starts at node 2^depth - 2^(d+1); the deepest level pairs the points, indices past N repeat the last
    # point; the last entry is never written (left zero)
    n = points.size(0)
    p2 = _foam.pow2_round_up(n)
    tree = torch.zeros((p2, 2, 3), dtype=points.dtype, device=points.device)
    if p2 < 2:
        return tree
    pad = torch.cat([points.detach(), points.detach()[-1:].expand(p2 - n, 3)], dim=0)
    lo = torch.minimum(pad[0::2], pad[1::2])
    hi = torch.maximum(pad[0::2], pad[1::2])
    depth = p2.bit_length() - 1
    for d in range(depth - 1, -1, -1):
        start = p2 - (1 << (d + 1))
        tree[start:start + (1 << d), 0] = lo
        tree[start:start + (1 << d), 1] = hi
        if d:
            lo = torch.minimum(lo[0::2], lo[1::2])
            hi = torch.maximum(hi[0::2], hi[1::2])
    return tree


def nn(points: torch.Tensor, tree: torch.Tensor, queries: torch.Tensor) -> torch.Tensor:
    """Index of the nearest point for every query (exact), uint32, shape queries.shape[:-1].
    float32 CUDA tensors go to the HIP kernels -- a brute-force pass for a handful of queries, the tree walk
    (scene_ops.nearest_point_tree over the caller's ``tree`` of the kd-ordered points) for many, same answer --; CPU
    tensors, which the reference also serves (nn_cpu, aabb_tree.cu:417-478), to the torch restatement below."""
    if points.dtype != queries.dtype:
        raise RuntimeError("points and queries must have the same dtype")
    if points.is_cuda and points.dtype == torch.float32:
        from . import scene_ops
        q = queries.to(points.device)
        n = points.size(0)
        # Camera positions (up to a few thousand): one streaming pass over the points per query, exact whatever the
        # state of `tree` -- RadFoamScene keeps the tree of its last triangulation update while the optimiser moves the
        # points, and a walk through stale boxes may miss the true nearest point.  Bulk queries (per-ray origins, test
        # sets): the walk through the caller's tree, the reference's own route (aabb_tree.cu:343-415); the tree must
        # describe these points, as the reference requires.
        many = q.numel() // 3 > 4096
        if many and isinstance(tree, torch.Tensor) and tree.is_cuda and tree.dtype == torch.float32 \
                and tuple(tree.shape) == (_foam.pow2_round_up(n), 2, 3):
            return scene_ops.nearest_point_tree(points, tree, q).to(queries.device)
        return scene_ops.nearest_point(points, q).to(queries.device)
    q = queries.reshape(-1, 3).to(points.device)
    out = torch.empty(q.size(0), dtype=torch.int64, device=points.device)
    p = points.detach().double()
    chunk = max(1, (1 << 24) // max(1, p.size(0)))
    for i in range(0, q.size(0), chunk):
        d = (p[None, :, :] - q[i:i + chunk, None, :].double()).square().sum(-1)
        out[i:i + chunk] = d.argmin(dim=1)
    return out.to(torch.uint32).reshape(queries.shape[:-1]).to(queries.device)


def farthest_neighbor(points: torch.Tensor, point_adjacency: torch.Tensor,
                      point_adjacency_offsets: torch.Tensor):
    """(index of the farthest Delaunay neighbour, mean half-distance to the neighbours)."""
    if points.is_cuda and points.dtype == torch.float32 and point_adjacency.dtype == torch.uint32:
        from . import scene_ops
        return scene_ops.farthest_neighbor(points, point_adjacency, point_adjacency_offsets)
    n = points.size(0)
    off = point_adjacency_offsets.to(torch.int64)
    adj = point_adjacency.to(torch.int64)
    counts = off[1:] - off[:-1]
    owner = torch.repeat_interleave(torch.arange(n, device=points.device), counts)
    dist = (points[adj] - points[owner]).norm(dim=-1)
    radius = torch.zeros(n, dtype=torch.float32, device=points.device)
    radius.index_add_(0, owner, 0.5 * dist)
    radius = radius / counts.to(torch.float32)
    maxd = torch.full((n,), -1.0, dtype=dist.dtype, device=points.device)
    maxd = maxd.scatter_reduce(0, owner, dist, reduce="amax", include_self=True)
    is_max = dist == maxd[owner]
    # first neighbour attaining the maximum (the reference keeps the first strict improvement)
    pos = torch.arange(adj.numel(), device=points.device)
    big = torch.full((n,), adj.numel(), dtype=torch.int64, device=points.device)
    first = big.scatter_reduce(0, owner[is_max], pos[is_max], reduce="amin", include_self=True)
    idx = adj[first.clamp(max=max(adj.numel() - 1, 0))]
    return idx.to(torch.uint32), radius


def _mix(x: np.ndarray) -> np.ndarray:
    """hash-prospector mixer, src/utils/random.h:13-22 (uint32 arithmetic)."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(17)
    x = (x * np.uint32(0xED5AD4BB)).astype(np.uint32)
    x ^= x >> np.uint32(11)
    x = (x * np.uint32(0xAC4C1B51)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x31848BAB)).astype(np.uint32)
    x ^= x >> np.uint32(14)
    return x


class BatchFetcher:
    """``BatchFetcher(data, batch_size, shuffle).next()`` -> device tensor [batch_size, ...].

    Reproduces the reference's index sequence (src/utils/batch_fetcher.cpp:60-70): element j of
    batch b is ``randint(make_rng(b*batch_size + j), 0, n)`` when shuffling, else
    ``(b*batch_size + j) % n`` -- so parallel fetchers over rays / rgbs / alphas stay aligned.

    The reference gathers rows of a host array on a worker thread and uploads them, four batches ahead.  Here, with a
    GPU present, the array is moved to the device ONCE (``device_resident_limit`` bytes at most, default 64 GB of the
    288: a Mip-NeRF-360 training set is a few GB of rays, colours and alphas) and a batch is one gather kernel on the
    current stream (rf_fetch_batch: indices and gather fused, no host work per batch).  Arrays above the limit, rows
    that are not a multiple of 4 bytes, and boxes without a GPU fall back to a synchronous torch gather with the same
    indices.
    """

    device_resident_limit = 64 << 30

    #: (rank, world_size) every ``shuffle=True`` fetcher constructed WITHOUT explicit ``rank`` / ``world_size`` serves;
    #: None = the whole batch (the reference's behaviour).  radfoam_amd.dist.enable_data_parallel() sets it so that the
    #: reference's unmodified data_loader (``get_iter``, data_loader/__init__.py:113-127) hands every rank its share of
    #: the training batches; sequential fetchers (``shuffle=False``: collect_error_map's views, test renders) are never
    #: sharded by default.
    default_shard = None

    def __init__(self, data: torch.Tensor, batch_size: int, shuffle: bool, rank: int | None = None,
                 world_size: int | None = None):
        """``rank`` / ``world_size`` (keyword extensions; the reference is single-GPU): serve elements
        [rank * batch_size / world_size, (rank + 1) * batch_size / world_size) of every batch of the reference's index
        sequence -- the ranks' shares, concatenated in rank order, ARE the batch one process would fetch, so a
        data-parallel run sees the same rays as the single-process run (radfoam_amd/dist.py)."""
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.batch_idx = 0
        if rank is None and world_size is None and self.shuffle and BatchFetcher.default_shard is not None:
            rank, world_size = BatchFetcher.default_shard
        if (rank is None) != (world_size is None):
            raise RuntimeError("BatchFetcher: rank and world_size go together")
        self.rank, self.world_size = (0, 1) if rank is None else (int(rank), int(world_size))
        if self.world_size < 1 or not (0 <= self.rank < self.world_size):
            raise RuntimeError("BatchFetcher: invalid rank / world_size")
        if self.batch_size % self.world_size:
            # equal shares only: the mean of the ranks' mean losses is then the batch's mean loss (train.py:188-204)
            raise RuntimeError(f"BatchFetcher: batch_size {self.batch_size} is not a multiple of world_size {self.world_size}")
        #: elements next() returns: batch_size / world_size
        self.local_batch_size = self.batch_size // self.world_size
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        if data.size(0) > 0xFFFFFFFF:
            raise RuntimeError("Too many elements")            # batch_fetcher.cpp:49
        self._row_bytes = data[0].numel() * data.element_size() if data.size(0) else 0
        nbytes = data.numel() * data.element_size()
        self._native = (self.device.type == "cuda" and data.size(0) > 0 and self._row_bytes % 4 == 0 and self._row_bytes > 0
                        and (data.is_cuda or nbytes <= self.device_resident_limit))
        if self._native and not data.is_cuda:
            # the copy must FIT: at most half of what the device has free right now (three fetchers -- rays, colours,
            # alphas -- are made one after the other, and the scene needs its share); a box whose memory is taken falls back
            # to the host gather instead of failing in the constructor
            try:
                free, _total = torch.cuda.mem_get_info(self.device)
                if nbytes > free // 2:
                    self._native = False
                else:
                    data = data.to(self.device).contiguous()
            except (torch.cuda.OutOfMemoryError, RuntimeError):
                self._native = False
        elif self._native:
            data = data.contiguous()
        if self._native:
            self.device = data.device
        self.data = data

    def _first(self) -> int:
        """Position in the reference's sequence of the first element next() returns."""
        return self.batch_idx * self.batch_size + self.rank * self.local_batch_size

    def _indices(self) -> np.ndarray:
        n = self.data.size(0)
        base = np.arange(self.local_batch_size, dtype=np.uint64) + np.uint64(self._first())
        if not self.shuffle:
            return (base % np.uint64(n)).astype(np.int64)
        seed = (base & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        with np.errstate(over="ignore"):
            bits = _mix(seed ^ np.uint32(0x2815DB5B))
        x = bits // np.uint32(0xFFFFFFFF // n)
        return np.minimum(x, np.uint32(n - 1)).astype(np.int64)

    def next(self) -> torch.Tensor:
        if self._native:
            import ctypes as C

            from . import _lib
            out = torch.empty((self.local_batch_size,) + tuple(self.data.shape[1:]), dtype=self.data.dtype,
                              device=self.device)
            with torch.cuda.device(self.device):
                # (batch_idx wraps at 2^32 like the reference's uint32 counter; the sequence position is 64-bit)
                first = (self.batch_idx & 0xFFFFFFFF) * self.batch_size + self.rank * self.local_batch_size
                rc = _lib.load().rf_fetch_batch_range(
                    C.c_void_p(self.data.data_ptr()), self.data.size(0), self._row_bytes, first,
                    self.local_batch_size, int(self.shuffle), C.c_void_p(out.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            _lib.check(rc)
            self.batch_idx += 1
            return out
        idx = torch.from_numpy(self._indices())
        self.batch_idx += 1
        return self.data[idx.to(self.data.device)].to(self.device)


class Viewer:
    def __init__(self, *args, **kwargs):
        raise RuntimeError("radfoam_amd: the interactive viewer is not supported (no display / GL)")


def run_with_viewer(*args, **kwargs):
    raise RuntimeError("radfoam_amd: the interactive viewer is not supported (no display / GL)")
