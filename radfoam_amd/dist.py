"""Ray-row data parallelism over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is a single-GPU program (no NCCL / torch.distributed anywhere in it, SURVEY.md
section 2); this is the one parallel axis the hot path offers: rays are independent, the foam
is read-only during a step.  The foam (points, attributes, CSR, packed tables: <= 1.6 GB at the
largest BASELINE config) is replicated on every GPU; a batch of rays shaped [H, W, 6] (or
[R, 6]) is split into contiguous row blocks, rank g tracing rows [g*H/G, (g+1)*H/G).

  * forward needs no exchange; ``gather_rows`` assembles the image where one is wanted;
  * backward produces per-rank partial ``points_grad`` / ``attr_grad``; they alias ONE flat
    fp32 buffer (Pipeline.trace_backward()["flat_grad"]), so a step costs a single SUM
    all-reduce (N*(3+A)*4 bytes: 248 MB for the 2M-point SH-2 foam).  On the fully connected
    xGMI mesh RCCL runs this as reduce-scatter + all-gather over all 7 links of every GPU;
  * ``contribution`` / ``point_error`` ([N,1]) are summed the same way when requested;
  * the row blocks of ONE image touch nearly disjoint wedges of the foam, so each rank's dense buffer
    is almost all zeros: ``SparseGradExchange`` all-gathers only the rows that hold anything (a few MB
    per rank) and adds them in rank order -- bit-identical sums on every rank at a fraction of the
    dense all-reduce's xGMI traffic (DESIGN.md section 7 has the arithmetic);
  * rays through different parts of a frame walk very different numbers of cells, so equal row counts
    are not equal work: ``balanced_row_blocks`` cuts the rows by measured cost (the previous step's
    ``num_intersections``).

Backend "nccl" (= RCCL on ROCm) on GPUs; the same code runs on "gloo" with CPU tensors, which
is how tests/test_dist.py covers it without a GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def row_block(num_rows: int, rank: int, world_size: int):
    """[begin, end) of the rows rank `rank` owns; blocks differ by at most one row."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("invalid rank / world_size")
    base, extra = divmod(num_rows, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def balanced_row_blocks(row_cost, world_size: int, align: int = 8):
    """Cut ``len(row_cost)`` rows into ``world_size`` contiguous blocks of (nearly) equal total cost.

    ``row_cost[y]`` is any non-negative measure of the work of image row y -- bench.py and
    ``ShardedTracer.rebalance`` use the cells the row's rays visited in the previous step.  Block
    boundaries are multiples of ``align`` rows (a wave owns an 8x8 pixel tile; every block still
    gets at least one such band).  Returns ``world_size + 1`` ascending boundaries, first 0, last the row count.
    Contiguous blocks, not interleaved bands: the cells a block's rays cross are then a compact wedge of
    the foam, which is what keeps the sparse gradient exchange small."""
    cost = [float(c) for c in row_cost]
    rows = len(cost)
    if world_size <= 0:
        raise ValueError("invalid world_size")
    align = max(1, int(align))
    bands = (rows + align - 1) // align
    if bands < world_size:          # too few rows to give everyone an aligned band: plain even split
        return [row_block(rows, r, world_size)[0] for r in range(world_size)] + [rows]
    band_cost = [sum(cost[b * align:(b + 1) * align]) for b in range(bands)]
    total = sum(band_cost)
    if not total > 0.0:
        band_cost = [1.0] * bands
        total = float(bands)
    bounds = [0]
    acc, b = 0.0, 0
    for r in range(1, world_size):
        target = total * r / world_size
        # advance while taking the next band brings the running cost closer to the target; always leave
        # enough bands for the blocks still to be cut and take at least one
        lo = bounds[-1] + 1
        hi = bands - (world_size - r)
        while b < hi and (b < lo or abs(acc + band_cost[b] - target) <= abs(acc - target)):
            acc += band_cost[b]
            b += 1
        bounds.append(b)
    bounds.append(bands)
    return [min(x * align, rows) for x in bounds]


def shard_rows(tensor: torch.Tensor, rank: int | None = None, world_size: int | None = None, dim: int = 0,
               bounds=None):
    """This rank's row block of a per-ray tensor (rays, start_point, depth_quantiles, targets).
    ``bounds``: boundaries from ``balanced_row_blocks`` instead of the even split."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if bounds is not None:
        if len(bounds) != world_size + 1 or bounds[-1] != tensor.size(dim):
            raise ValueError("row bounds do not match the tensor / world size")
        b, e = bounds[rank], bounds[rank + 1]
    else:
        b, e = row_block(tensor.size(dim), rank, world_size)
    return tensor.narrow(dim, b, e - b)


def all_reduce_gradients(backward_result: dict, group=None, async_op: bool = False):
    """SUM the partial gradients of trace_backward across ranks, in place.

    Uses the flat [points_grad | attr_grad] buffer when the pipeline provides it (one
    collective); falls back to two collectives otherwise (fp16 pipelines return a converted
    attr_grad).  Returns the work handle(s) when async_op is set.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    handles = []
    flat = backward_result.get("flat_grad")
    attr = backward_result["attr_grad"]
    if flat is not None and attr.dtype == flat.dtype and \
            attr.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr():     # attr_grad is a view of flat
        handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    else:
        handles.append(dist.all_reduce(backward_result["points_grad"], op=dist.ReduceOp.SUM, group=group,
                                       async_op=async_op))
        handles.append(dist.all_reduce(attr, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    pe = backward_result.get("point_error")
    if pe is not None:
        handles.append(dist.all_reduce(pe, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    return handles if async_op else None


def all_reduce_statistic(t: torch.Tensor, group=None):
    """SUM a per-point statistic ([N,1] contribution / point_error) across ranks, in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def gather_rows(local: torch.Tensor, num_rows: int, group=None, dim: int = 0, bounds=None) -> torch.Tensor:
    """Concatenate the ranks' row blocks (e.g. rgba [h_g, W, 4]) into the full [H, W, 4] tensor on
    every rank.  Row blocks may differ in height, so shorter blocks are padded for the gather."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if bounds is not None:
        sizes = [(bounds[r], bounds[r + 1]) for r in range(world)]
    else:
        sizes = [row_block(num_rows, r, world) for r in range(world)]
    longest = max(e - b for b, e in sizes)
    pad_shape = list(local.shape)
    pad_shape[dim] = longest
    padded = local.new_zeros(pad_shape)
    padded.narrow(dim, 0, local.size(dim)).copy_(local)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p.narrow(dim, 0, e - b) for p, (b, e) in zip(parts, sizes)], dim=dim)


class SparseGradExchange:
    """SUM of the ranks' partial gradients by exchanging only the rows that hold anything.

    ``reduce(result)`` takes what ``Pipeline.trace_backward`` returned (the flat fp32
    ``[points_grad | attr_grad]`` buffer and its two views) and leaves, in place and on every rank, the sum
    over ranks -- what ``all_reduce_gradients`` computes with a dense all-reduce -- in three moves:

      1. compact: one pass over the dense buffer lists the cells with a non-zero gradient and packs their
         rows ``{cell, points_grad[3], attr_grad[A]}`` (rf_compact_grad_rows);
      2. all-gather the row counts (one small collective, read back on the host: the only synchronisation)
         and then the packed rows, padded to the longest list (one collective, a few MB per rank);
      3. every rank zeroes its own listed rows and adds all ranks' rows IN RANK ORDER
         (rf_scatter_grad_rows): the same additions in the same order everywhere, so the ranks' sums --
         and the parameters an identical optimiser step derives from them -- are bit-identical, which a
         ring all-reduce does not promise.

    Traffic per rank: (world - 1) * longest list * (4 + A) * 4 bytes in, against 2 * (world - 1) / world *
    N * (3 + A) * 4 for the dense all-reduce.  It pays when the ranks' rays cross mostly different cells:
    the row blocks of one image (each rank touches ~1/world of the frame's cells); for a shuffled training
    batch every rank touches everything and the dense all-reduce is the better tool -- ``reduce`` falls
    back to it by itself when the lists cover more than ``dense_fraction`` of the points.

    CUDA tensors go through the HIP kernels; CPU tensors (the gloo tests) through the same steps written
    with torch indexing."""

    def __init__(self, group=None, dense_fraction: float = 0.25, growth: float = 1.25):
        self.group = group
        self.dense_fraction = float(dense_fraction)
        self.growth = float(growth)
        self._send = None
        self._recv = None
        #: rows sent by each rank in the last reduce() (host ints), None when it fell back to dense
        self.last_counts = None

    # -- the two device steps, HIP for CUDA tensors / torch indexing for CPU tensors ---------------
    @staticmethod
    def _pitch(a: int) -> int:
        return (1 + 3 + a + 3) // 4 * 4

    @staticmethod
    def _row_pitch(attr_grad) -> int:
        """Floats between two rows of ``attr_grad``: A for the reference's dense [N, A], 16 / 32 / 64 for the [N, A] view
        of rows on 64-byte lines that Pipeline.trace_backward returns by default (gradient_row_pitch "auto")."""
        n, a = attr_grad.shape
        if attr_grad.stride(1) != 1 or (n > 1 and attr_grad.stride(0) < a):
            raise RuntimeError("SparseGradExchange needs attr_grad rows of contiguous floats (a [N, A] tensor or a "
                               "[N, A] view of padded rows), got strides %r" % (tuple(attr_grad.stride()),))
        return int(attr_grad.stride(0)) if n > 1 else a

    def _compact(self, points_grad, attr_grad, send, count):
        n, a = attr_grad.shape
        pitch = self._row_pitch(attr_grad)
        if points_grad.is_cuda:
            from . import _lib
            lib = _lib.load()
            with torch.cuda.device(points_grad.device):
                rc = lib.rf_compact_grad_rows_pitched(points_grad.data_ptr(), attr_grad.data_ptr(), n, a, pitch,
                                                      send.shape[0], count.data_ptr(), send.data_ptr(),
                                                      torch.cuda.current_stream(points_grad.device).cuda_stream)
            _lib.check(rc)
            return
        touched = ((points_grad != 0).any(dim=1) | (attr_grad != 0).any(dim=1)).nonzero().reshape(-1)
        k = int(touched.numel())
        count.fill_(k)
        k = min(k, send.shape[0])
        idx = touched[:k]
        send[:k, 0] = idx.to(torch.int32).view(torch.float32)
        send[:k, 1:4] = points_grad[idx]
        send[:k, 4:4 + a] = attr_grad[idx]
        send[:k, 4 + a:] = 0

    @classmethod
    def _scatter(cls, rows, k, points_grad, attr_grad, zero):
        if k == 0:
            return
        n, a = attr_grad.shape
        if points_grad.is_cuda:
            from . import _lib
            lib = _lib.load()
            with torch.cuda.device(points_grad.device):
                rc = lib.rf_scatter_grad_rows_pitched(rows.data_ptr(), k, n, a, cls._row_pitch(attr_grad),
                                                      1 if zero else 0, points_grad.data_ptr(), attr_grad.data_ptr(),
                                                      torch.cuda.current_stream(points_grad.device).cuda_stream)
            _lib.check(rc)
            return
        idx = rows[:k, 0].contiguous().view(torch.int32).to(torch.int64)
        if zero:
            points_grad[idx] = 0
            attr_grad[idx] = 0
        else:
            points_grad[idx] += rows[:k, 1:4]
            attr_grad[idx] += rows[:k, 4:4 + a]

    def _buffers(self, cap, world, pitch, like):
        if self._send is None or self._send.shape != (cap, pitch) or self._send.device != like.device:
            self._send = torch.empty((cap, pitch), dtype=torch.float32, device=like.device)
            self._recv = torch.empty((world * cap, pitch), dtype=torch.float32, device=like.device)
        return self._send, self._recv

    def reduce(self, backward_result: dict):
        """In place; returns ``backward_result``."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return backward_result
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        pg, ag = backward_result["points_grad"], backward_result["attr_grad"]
        flat = backward_result.get("flat_grad")
        # the rows must be the fp32 accumulator itself (a view into flat_grad, whatever its row pitch), not a converted
        # copy: fp16 pipelines return a half attr_grad next to the fp32 flat buffer
        inside = flat is not None and ag.dtype == torch.float32 and pg.dtype == torch.float32 and \
            ag.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() and \
            pg.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() and ag.dim() == 2 and ag.stride(1) == 1
        if not inside:
            all_reduce_gradients(backward_result, group=self.group)
            self.last_counts = None
            return backward_result
        n, a = ag.shape
        pitch = self._pitch(a)
        cap = self._send.shape[0] if self._send is not None else max(1024, n // (4 * world))
        count = torch.zeros(1, dtype=torch.int32, device=pg.device)
        counts_dev = torch.empty(world, dtype=torch.int32, device=pg.device)
        while True:
            send, recv = self._buffers(cap, world, pitch, pg)
            count.zero_()
            self._compact(pg, ag, send, count)
            dist.all_gather_into_tensor(counts_dev, count, group=self.group)
            counts = [int(c) for c in counts_dev.tolist()]      # the one host synchronisation of a step
            longest = max(counts)
            if longest <= cap:
                break
            cap = int(longest * self.growth) + 1                # every rank sees the same counts: same decision
        if sum(counts) > self.dense_fraction * n * world or longest == 0:
            if longest:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            self.last_counts = None if longest else counts
            pe = backward_result.get("point_error")
            if pe is not None:
                dist.all_reduce(pe, op=dist.ReduceOp.SUM, group=self.group)
            return backward_result
        dist.all_gather_into_tensor(recv[: world * longest], send[:longest], group=self.group)
        got = recv[: world * longest].view(world, longest, pitch)
        self._scatter(send, counts[rank], pg, ag, zero=True)
        for r in range(world):
            self._scatter(got[r], counts[r], pg, ag, zero=False)
        pe = backward_result.get("point_error")
        if pe is not None:
            dist.all_reduce(pe, op=dist.ReduceOp.SUM, group=self.group)
        self.last_counts = counts
        return backward_result


class ShardedTracer:
    """Row-sharded forward/backward around a Pipeline: every rank calls it with the FULL ray
    tensors and receives its own rows' outputs; ``backward`` returns gradients already summed
    over ranks (identical on all ranks, ready for an identical optimiser step).

    ``exchange``: "sparse" (SparseGradExchange, default: rows of one image touch nearly disjoint cells)
    or "dense" (one SUM all-reduce of the flat buffer).  ``rebalance(num_intersections_local)`` re-cuts
    the row blocks by the cost the last forward measured."""

    def __init__(self, pipeline, group=None, exchange: str = "sparse"):
        if exchange not in ("sparse", "dense"):
            raise ValueError("exchange must be 'sparse' or 'dense'")
        self.pipeline = pipeline
        self.group = group
        self.bounds = None
        self.sparse = SparseGradExchange(group) if exchange == "sparse" else None
        self._pitch_decided = False

    def _decide_row_pitch(self):
        """The dense exchange all-reduces the flat buffer: padding columns (13 -> 16, 28 -> 32, 49 -> 64 floats per row)
        would be up to 29 % more xGMI bytes per step, so a multi-rank dense tracer asks the pipeline for dense rows; the
        sparse exchange sends packed rows whatever the pitch.  Decided at the first backward, when the world size is
        known for sure -- a tracer may be built before init_process_group."""
        if self._pitch_decided or not dist.is_initialized():
            return
        self._pitch_decided = True
        if self.sparse is None and hasattr(self.pipeline, "gradient_row_pitch") and self._world() > 1:
            self.pipeline.gradient_row_pitch = "dense"

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def _shard(self, t):
        return None if t is None else shard_rows(t, self._rank(), self._world(), bounds=self.bounds)

    def rows(self, num_rows: int):
        """[begin, end) of this rank's rows of an image with ``num_rows`` rows."""
        if self.bounds is not None:
            return self.bounds[self._rank()], self.bounds[self._rank() + 1]
        return row_block(num_rows, self._rank(), self._world())

    def rebalance(self, num_intersections_local: torch.Tensor, num_rows: int, align: int = 8):
        """Re-cut the row blocks so that every rank gets the same number of visited cells, from the
        ``num_intersections`` ([h_local, W, 1]) the last forward returned for this rank's rows.  One small
        all-gather; every rank computes the same boundaries.  Returns them."""
        world = self._world()
        if world == 1:
            self.bounds = None
            return [0, num_rows]
        # A wave walks an 8x8 pixel tile until its LONGEST ray ends, so the work of a row is not the sum of its
        # rays' steps but, per 8-pixel segment, the longest of them (rows of one tile are alike, so summing the rows
        # of a band gives 8 x the tiles' maxima): measured on the north-star frame, cuts by summed steps left the
        # ranks 10 % apart at 8 GPUs -- the edge blocks had fewer rows but just as many wave-steps
        ni = num_intersections_local.reshape(num_intersections_local.shape[0], -1).to(torch.int64)
        pad = (-ni.shape[1]) % 8
        if pad:
            ni = torch.nn.functional.pad(ni, (0, pad))
        cost_local = ni.view(ni.shape[0], -1, 8).amax(dim=2).sum(dim=1)
        b, e = self.rows(num_rows)
        full = torch.zeros(num_rows, dtype=torch.int64, device=cost_local.device)
        full[b:e] = cost_local
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        self.bounds = balanced_row_blocks(full.tolist(), world, align=align)
        return self.bounds

    def forward(self, points, attributes, adjacency, offsets, rays, start_point, depth_quantiles=None, **kw):
        return self.pipeline.trace_forward(points, attributes, adjacency, offsets, self._shard(rays),
                                           self._shard(start_point), depth_quantiles=self._shard(depth_quantiles),
                                           **kw)

    def backward(self, points, attributes, adjacency, offsets, rays, start_point, rgba_local, grad_local,
                 depth_quantiles=None, depth_indices_local=None, depth_grad_local=None, **kw):
        self._decide_row_pitch()
        res = self.pipeline.trace_backward(points, attributes, adjacency, offsets, self._shard(rays),
                                           self._shard(start_point), rgba_local, grad_local,
                                           self._shard(depth_quantiles), depth_indices_local, depth_grad_local, **kw)
        if self.sparse is not None:
            self.sparse.reduce(res)
        else:
            all_reduce_gradients(res, group=self.group)
        return res


# ---------------------------------------------------------------------------------------------------------------------
# Data-parallel TRAINING (BASELINE config 4: "full train.py loop, rays row-sharded across 8xMI355X with RCCL grad
# all-reduce").  The reference has no distributed mode at all; what follows lets its unmodified
# RadFoamScene / TraceRays / train.py loop body run one process per GPU with identical parameters on every rank.


def replica_checksums(tensors) -> torch.Tensor:
    """One int64 per tensor: a position-weighted wrap-around sum of its 32-bit words (any dtype whose element size is a
    multiple of 4 bytes; bools / bytes are widened).  Equal tensors give equal sums; a single differing word changes it."""
    sums = []
    for t in tensors:
        t = t.detach().contiguous()
        if t.element_size() % 4:
            t = t.to(torch.int32)
        w = t.view(torch.int32).reshape(-1).to(torch.int64)
        k = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
        sums.append(((w + 0x9E3779B1) * (2 * k + 1)).sum() + w.numel())
    return torch.stack(sums) if sums else torch.zeros(0, dtype=torch.int64)


def assert_replicas_agree(named_tensors: dict, group=None):
    """Raise unless every rank holds the same bits in every named tensor (parameters after an optimiser step, the
    adjacency lists after a rebuild: the replicated foam must stay replicated or the ranks' gradients describe different
    scenes).  One small all-gather of checksums.  Returns the checksums (int64 [len(named_tensors)])."""
    names = list(named_tensors)
    mine = replica_checksums([named_tensors[k] for k in names])
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return mine
    world = dist.get_world_size(group)
    shapes = torch.tensor([named_tensors[k].numel() for k in names], dtype=torch.int64, device=mine.device)
    both = torch.cat([mine, shapes])
    parts = [torch.empty_like(both) for _ in range(world)]
    dist.all_gather(parts, both, group=group)
    host = [p.cpu() for p in parts]
    for r in range(1, world):
        if not torch.equal(host[r], host[0]):
            bad = [names[i % len(names)] for i in (host[r] != host[0]).nonzero().reshape(-1).tolist()]
            raise RuntimeError(f"data-parallel replicas diverged: rank {r} differs from rank 0 in {sorted(set(bad))}")
    return mine


def origin_group_key(rays: torch.Tensor) -> torch.Tensor:
    """uint32 per ray, equal for rays with the same origin (one camera): a hash of the origin's three floats.  Stands in
    for the entry cell as the high half of the kernels' ray-order key when the entry cells are not known yet."""
    w = rays.detach().reshape(-1, 6)[:, :3].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    h = (w[:, 0] * 0x9E3779B1 + ((w[:, 1] << 11) | (w[:, 1] >> 21)) * 0x85EBCA77 + ((w[:, 2] << 22) | (w[:, 2] >> 10)) * 0xC2B2AE3D)
    h = (h ^ (h >> 29)) & 0xFFFFFFFF
    return torch.where(h >= 0x80000000, h - 0x100000000, h).to(torch.int32).view(torch.uint32)


def coherent_order(rays: torch.Tensor, group=None) -> torch.Tensor:
    """int64 [R]: the rays of a flat batch in a coherent order -- by ``group`` (uint32 per ray: the entry cell, or
    origin_group_key(rays) when that is not known yet), then by direction (Morton code on the octahedral map): what
    rf_build_ray_order computes for the kernels' own slot order.  A pure function of its inputs, so every rank that holds
    the same batch computes the same order.  CUDA tensors: the HIP sort (0.2 ms for 10^6 rays); CPU tensors (the gloo
    tests): a torch restatement of the same key."""
    r = rays.detach().reshape(-1, 6).contiguous()
    n = r.shape[0]
    g = origin_group_key(r) if group is None else group.detach().reshape(-1).contiguous()
    if g.dtype != torch.uint32:
        g = g.to(torch.int64).to(torch.uint32) if g.dtype != torch.int32 else g.view(torch.uint32)
    if r.is_cuda and n > 0:
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        order = torch.empty(n, dtype=torch.int32, device=r.device)
        ws = torch.empty(max(int(lib.rf_ray_order_workspace_bytes(n)), 256), dtype=torch.uint8, device=r.device)
        with torch.cuda.device(r.device):
            rc = lib.rf_build_ray_order(C.c_void_p(r.data_ptr()), C.c_void_p(g.data_ptr()), n, C.c_void_p(order.data_ptr()),
                                        C.c_void_p(ws.data_ptr()), ws.numel(),
                                        C.c_void_p(torch.cuda.current_stream(r.device).cuda_stream))
        _lib.check(rc)
        return order.to(torch.int64)
    d = r[:, 3:6].double()
    l1 = d.abs().sum(dim=1).clamp(min=1e-300)
    px, py = d[:, 0] / l1, d[:, 1] / l1
    neg = d[:, 2] < 0
    ox = (1 - py.abs()) * torch.where(px >= 0, 1.0, -1.0)
    oy = (1 - px.abs()) * torch.where(py >= 0, 1.0, -1.0)
    px, py = torch.where(neg, ox, px), torch.where(neg, oy, py)
    u = ((px * 0.5 + 0.5) * 65535.0).clamp(0, 65535).to(torch.int64)
    v = ((py * 0.5 + 0.5) * 65535.0).clamp(0, 65535).to(torch.int64)

    def spread(x):
        x = (x | (x << 8)) & 0x00FF00FF
        x = (x | (x << 4)) & 0x0F0F0F0F
        x = (x | (x << 2)) & 0x33333333
        return (x | (x << 1)) & 0x55555555

    key = (g.view(torch.int32).to(torch.int64) & 0xFFFFFFFF) * (1 << 32) + (spread(u) | (spread(v) << 1))
    # (the high half may exceed 2^31: compare as unsigned by sorting on the two halves)
    hi, lo = key >> 32, key & 0xFFFFFFFF
    o = torch.argsort(lo, stable=True)
    return o[torch.argsort(hi[o], stable=True)]


def coherent_shard(rays: torch.Tensor, rank: int | None = None, world_size: int | None = None, group=None,
                   process_group=None) -> torch.Tensor:
    """int64 indices of THIS rank's share of a flat batch that every rank holds whole: a contiguous 1/W of
    coherent_order(rays, group) -- one patch of one camera's directions instead of a 1/W-dense sample of every camera.
    Why: a rank's 125,000 rays of a shuffled 10^6-ray batch (W = 8) are 11 pixels apart; the walk's lanes then share
    nothing and the launch is 1.5x slower than for a coherent eighth (profiles/r06/*shard_simulation_training_batch*:
    forward + backward 6.26 against 4.13 ms per rank).  The batch size must be a multiple of the world size."""
    if world_size is None:
        world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(process_group) if dist.is_initialized() else 0
    n = rays.reshape(-1, 6).shape[0]
    if n % world_size:
        raise RuntimeError(f"coherent_shard: {n} rays are not a multiple of the world size {world_size}")
    per = n // world_size
    return coherent_order(rays, group)[rank * per:(rank + 1) * per]


class DataParallelPipeline:
    """A ``radfoam.Pipeline`` that keeps the ranks of a data-parallel job consistent: same methods, same arguments, same
    returned dicts as the pipeline it wraps (pipeline_bindings.cpp:626-667), plus the exchanges one process per GPU needs.
    The reference's ``TraceRays`` (radfoam_model/render.py:10-122) calls it like any pipeline, so its ``loss.backward()``
    ends in gradients already combined over the ranks and an identical Adam step follows on every rank.

    Two ways to divide a step's rays (``shard``; ``replicated_inputs()`` switches to "rows" for a block of calls):

      "caller"  the caller hands every rank ITS share (``BatchFetcher(..., rank=, world_size=)``: the ranks' batches
                concatenated are the batch one process would fetch).  Outputs are the share's; ``trace_backward``
                exchanges the flat ``[points_grad | attr_grad]`` buffer and -- ``reduce="mean"`` -- divides by the world
                size: train.py's losses are means over the batch (train.py:188-204), so the mean of the ranks'
                gradients IS the gradient of the whole batch's loss.  ``contribution`` / ``point_error`` are per-point
                SUMS over rays and are summed over ranks.
      "rows"    every rank is handed the SAME rays (collect_error_map's views, evaluation renders, or a train.py whose
                data loader was left alone): each rank traces its contiguous block of leading-dimension rows, forward
                outputs are all-gathered, so every rank returns what a single process would -- bit for bit in rgba /
                depth / num_intersections -- and the backward takes its block of the caller's full upstream gradient and
                SUMS over ranks (the shares are disjoint parts of one loss).  No rank-dependent value reaches the
                caller: loss, statistics and whatever the scene derives from them stay identical on every rank.
                A FLAT batch ([R, 6], at least ``coherent_min_rays`` rays, R a multiple of the world size) is cut in the
                kernels' coherent order instead of the caller's (coherent_shard: a contiguous 1/W of the rays sorted by
                entry cell and direction), outputs go back to the caller's positions.

    ``exchange``: "dense" = one SUM all-reduce of the flat buffer (RCCL: reduce-scatter + all-gather over the xGMI mesh);
    "sparse" = SparseGradExchange (packed non-zero rows, added in rank order); "auto" = sparse for image-shaped rays
    (a frame's row blocks touch nearly disjoint cells), dense for flat batches (a shuffled batch touches everything).
    Either way every rank ends with the SAME bits, which the replicated optimiser needs.  The row pitch of the gradient
    accumulator is left to the pipeline (rows on 64-byte lines: 29 % more bytes on the wire than dense rows at SH 3, but
    the all-lit backward is 2x slower without them).
    """

    def __init__(self, pipeline, group=None, shard: str = "caller", reduce: str = "mean", exchange: str = "auto"):
        if shard not in ("caller", "rows"):
            raise ValueError("shard must be 'caller' or 'rows'")
        if reduce not in ("mean", "sum"):
            raise ValueError("reduce must be 'mean' or 'sum'")
        if exchange not in ("auto", "dense", "sparse"):
            raise ValueError("exchange must be 'auto', 'dense' or 'sparse'")
        self.inner = pipeline
        self.group = group
        self.shard = shard
        self.reduce = reduce
        self.exchange = exchange
        self._sparse = SparseGradExchange(group)
        #: what the last trace_backward did: {"exchange": "dense" | "sparse" | "none", "world": W, "rows": [..] | None}
        self.last_exchange = None
        #: "rows": flat batches of at least this many rays are cut in the coherent order (see the class docstring)
        self.coherent_min_rays = 16384
        self._cut = None            # the last flat "rows" cut: the caller's tensors -> this rank's gathered share

    # everything that is not a trace call (attribute_dim, knobs such as backward_mode, invalidate, ...) is the inner
    # pipeline's; knobs set on the wrapper land on the inner pipeline too
    def __getattr__(self, name):
        return getattr(self.__dict__["inner"], name)

    def __setattr__(self, name, value):
        own = name.startswith("_") or name in self.__dict__ or hasattr(type(self), name) or callable(value) or \
            "inner" not in self.__dict__ or not hasattr(self.__dict__["inner"], name)
        if own:     # the wrapper's own state, and methods patched onto it (a profiler timing trace_forward)
            object.__setattr__(self, name, value)
        else:       # a knob of the wrapped pipeline (backward_mode, record_trail, tile_order_mode, ...)
            setattr(self.__dict__["inner"], name, value)

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    class _Replicated:
        def __init__(self, outer):
            self.outer = outer

        def __enter__(self):
            self.was = self.outer.shard
            self.outer.shard = "rows"
            return self.outer

        def __exit__(self, *exc):
            self.outer.shard = self.was
            return False

    def replicated_inputs(self):
        """``with pipeline.replicated_inputs():`` -- the calls inside receive the same rays on every rank ("rows")."""
        return DataParallelPipeline._Replicated(self)

    # -- forward --------------------------------------------------------------------------------------------------------
    def trace_forward(self, points, attributes, point_adjacency, point_adjacency_offsets, rays, start_point,
                      depth_quantiles=None, weight_threshold=None, max_intersections=None, return_contribution=False,
                      **extra):
        world = self._world()
        kw = dict(depth_quantiles=depth_quantiles, weight_threshold=weight_threshold,
                  max_intersections=max_intersections, return_contribution=return_contribution, **extra)
        if world == 1:
            return self.inner.trace_forward(points, attributes, point_adjacency, point_adjacency_offsets, rays,
                                            start_point, **kw)
        if self.shard == "caller":
            out = self.inner.trace_forward(points, attributes, point_adjacency, point_adjacency_offsets, rays,
                                           start_point, **kw)
            if return_contribution:
                all_reduce_statistic(out["contribution"], group=self.group)
            return out
        # "rows": the same rays everywhere; this rank traces its share, everyone gets everything
        rank = self._rank()
        lead = rays.shape[:-1]
        if len(lead) == 0:
            raise RuntimeError("rays must have a batch dimension")
        start_b = torch.broadcast_to(start_point, lead)
        cs = self._coherent_cut(rays, start_b, depth_quantiles, world, rank)
        if cs is not None:
            kw["depth_quantiles"] = cs["quantiles"]
            local = self.inner.trace_forward(points, attributes, point_adjacency, point_adjacency_offsets, cs["rays"],
                                             cs["start"], **kw)
        else:
            cut = lambda t: None if t is None else shard_rows(t, rank, world)
            kw["depth_quantiles"] = cut(depth_quantiles)
            local = self.inner.trace_forward(points, attributes, point_adjacency, point_adjacency_offsets, cut(rays),
                                             cut(start_b), **kw)
        out = {}
        for key in ("rgba", "depth", "depth_indices", "num_intersections"):
            if key in local:
                t = local[key]
                view32 = t.dtype == torch.uint32            # collectives on uint32 are thin: move the words as int32
                g = gather_rows(t.view(torch.int32) if view32 else t, lead[0], group=self.group)
                if cs is not None:                           # the ranks' shares in sorted order -> the caller's positions
                    g = torch.empty_like(g).index_copy_(0, cs["order"], g)
                out[key] = g.view(torch.uint32) if view32 else g
        if return_contribution:
            out["contribution"] = all_reduce_statistic(local["contribution"], group=self.group)
        return out

    def _coherent_cut(self, rays, start_b, depth_quantiles, world, rank):
        """This rank's share of a flat batch in the coherent order, or None when the batch is cut by leading rows (images,
        small batches, sizes that do not divide).  The gathered share is kept (keyed on the caller's tensors) so that the
        trace_backward of the same batch hands the inner pipeline the SAME tensors -- it replays the forward's hop trail
        only for the tensors it recorded it on."""
        if rays.dim() != 2 or rays.shape[0] < int(self.coherent_min_rays) or rays.shape[0] % world:
            return None
        from .pipeline import _source_key
        key = (_source_key(rays, rays.contiguous()), _source_key(start_b, start_b),
               None if depth_quantiles is None else _source_key(depth_quantiles, depth_quantiles.contiguous()), world, rank)
        c = self._cut
        if c is not None and c["key"] == key:
            return c
        start_c = start_b.contiguous()
        order = coherent_order(rays, start_c)
        per = rays.shape[0] // world
        mine = order[rank * per:(rank + 1) * per]
        take32 = lambda t: t.view(torch.int32)[mine].contiguous().view(torch.uint32) if t.dtype == torch.uint32 else t[mine].contiguous()
        self._cut = {"key": key, "order": order, "mine": mine, "rays": rays[mine].contiguous(), "start": take32(start_c),
                     "quantiles": None if depth_quantiles is None else depth_quantiles[mine].contiguous(),
                     "refs": (rays, start_b, depth_quantiles)}
        return self._cut

    # -- backward -------------------------------------------------------------------------------------------------------
    def _exchange(self, res, image_shaped):
        world = self._world()
        how = self.exchange if self.exchange != "auto" else ("sparse" if image_shaped else "dense")
        rows = None
        if how == "sparse":
            self._sparse.reduce(res)           # (point_error summed inside)
            rows = self._sparse.last_counts
            if rows is None:
                how = "dense"                  # the lists were not sparse: reduce() fell back to the all-reduce
        else:
            all_reduce_gradients(res, group=self.group)
        self.last_exchange = {"exchange": how, "world": world, "rows": rows}

    def trace_backward(self, points, attributes, point_adjacency, point_adjacency_offsets, rays, start_point, rgb_out,
                       grad_in, depth_quantiles=None, depth_indices=None, depth_grad_in=None, ray_error=None,
                       weight_threshold=None, max_intersections=None):
        world = self._world()
        if world == 1:
            self.last_exchange = {"exchange": "none", "world": 1, "rows": None}
            return self.inner.trace_backward(points, attributes, point_adjacency, point_adjacency_offsets, rays,
                                             start_point, rgb_out, grad_in, depth_quantiles, depth_indices, depth_grad_in,
                                             ray_error, weight_threshold, max_intersections)
        rank = self._rank()
        full_rays = rays
        if self.shard == "rows":
            lead = rays.shape[:-1]
            start_b = torch.broadcast_to(start_point, lead)
            cs = self._coherent_cut(rays, start_b, depth_quantiles, world, rank)
            if cs is not None:
                mine = cs["mine"]
                cut = lambda t: None if t is None else (
                    t.view(torch.int32)[mine].contiguous().view(torch.uint32) if t.dtype == torch.uint32 else t[mine].contiguous())
                rays, start_point, depth_quantiles = cs["rays"], cs["start"], cs["quantiles"]
            else:
                cut = lambda t: None if t is None else shard_rows(t, rank, world)
                rays, start_point, depth_quantiles = cut(rays), cut(start_b), cut(depth_quantiles)
            # (this rank's rows of the gathered forward outputs are its own forward's outputs, bit for bit)
            rgb_out, grad_in = cut(rgb_out), cut(grad_in)
            depth_indices, depth_grad_in = cut(depth_indices), cut(depth_grad_in)
            ray_error = None if ray_error is None else cut(torch.broadcast_to(ray_error, lead))
        if self.reduce == "mean" and self.shard == "caller":
            # the mean over ranks = the sum of the ranks' gradients of loss / W: the backward is linear in the upstream
            # gradients, so the 1 / W goes onto them (a few MB) instead of onto the summed flat buffer (half a GB at 2 M
            # points and SH 3: a pass of its own); exact for power-of-two worlds.  point_error does not depend on them.
            scale = 1.0 / world
            grad_in = grad_in * scale
            depth_grad_in = None if depth_grad_in is None else depth_grad_in * scale
        res = self.inner.trace_backward(points, attributes, point_adjacency, point_adjacency_offsets, rays, start_point,
                                        rgb_out, grad_in, depth_quantiles, depth_indices, depth_grad_in, ray_error,
                                        weight_threshold, max_intersections)
        self._exchange(res, image_shaped=full_rays.dim() == 3)
        if self.shard == "rows":
            res["ray_grad"] = torch.zeros_like(full_rays)      # the reference never writes it (pipeline_bindings.cpp:455)
        return res

    def trace_benchmark(self, *args, **kwargs):
        """The render path writes the caller's frame in place: every rank renders it whole (no exchange)."""
        return self.inner.trace_benchmark(*args, **kwargs)


_DATA_PARALLEL = {"on": False, "kw": {}}


def enable_data_parallel(shard_batches: bool = True, **pipeline_kw):
    """After ``init_process_group``: from now on ``radfoam.create_pipeline`` returns a DataParallelPipeline
    (``pipeline_kw``: its ``group`` / ``shard`` / ``reduce`` / ``exchange``) and -- ``shard_batches`` -- every
    ``radfoam.BatchFetcher(..., shuffle=True)`` serves this rank's share of the reference's index sequence.  That is all an
    unmodified train.py needs to run one process per GPU: its scene calls create_pipeline (scene.py:59), its data loader
    builds the shuffled fetchers (data_loader/__init__.py:113-127), its collect_error_map / test renders use sequential
    fetchers, which stay whole -- wrap those calls in ``model.pipeline.replicated_inputs()``."""
    from . import shims
    if not dist.is_initialized():
        raise RuntimeError("enable_data_parallel: call torch.distributed.init_process_group first")
    group = pipeline_kw.get("group")
    _DATA_PARALLEL["on"] = True
    _DATA_PARALLEL["kw"] = dict(pipeline_kw)
    shims.BatchFetcher.default_shard = (dist.get_rank(group), dist.get_world_size(group)) if shard_batches else None


def disable_data_parallel():
    from . import shims
    _DATA_PARALLEL["on"] = False
    _DATA_PARALLEL["kw"] = {}
    shims.BatchFetcher.default_shard = None


def wrap_pipeline(pipeline):
    """What create_pipeline returns: the pipeline itself, or -- after enable_data_parallel() -- its data-parallel wrapper."""
    if _DATA_PARALLEL["on"] and not isinstance(pipeline, DataParallelPipeline):
        return DataParallelPipeline(pipeline, **_DATA_PARALLEL["kw"])
    return pipeline
