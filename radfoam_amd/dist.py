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
  * ``contribution`` / ``point_error`` ([N,1]) are summed the same way when requested.

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


def shard_rows(tensor: torch.Tensor, rank: int | None = None, world_size: int | None = None, dim: int = 0):
    """This rank's row block of a per-ray tensor (rays, start_point, depth_quantiles, targets)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
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
    if flat is not None and attr.dtype == flat.dtype and attr.data_ptr() == flat.data_ptr() + \
            backward_result["points_grad"].numel() * flat.element_size():
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


def gather_rows(local: torch.Tensor, num_rows: int, group=None, dim: int = 0) -> torch.Tensor:
    """Concatenate the ranks' row blocks (e.g. rgba [h_g, W, 4]) into the full [H, W, 4] tensor on
    every rank.  Row blocks may differ by one row, so shorter blocks are padded for the gather."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [row_block(num_rows, r, world) for r in range(world)]
    longest = max(e - b for b, e in sizes)
    pad_shape = list(local.shape)
    pad_shape[dim] = longest
    padded = local.new_zeros(pad_shape)
    padded.narrow(dim, 0, local.size(dim)).copy_(local)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p.narrow(dim, 0, e - b) for p, (b, e) in zip(parts, sizes)], dim=dim)


class ShardedTracer:
    """Row-sharded forward/backward around a Pipeline: every rank calls it with the FULL ray
    tensors and receives its own rows' outputs; ``backward`` returns gradients already summed
    over ranks (identical on all ranks, ready for an identical optimiser step)."""

    def __init__(self, pipeline, group=None):
        self.pipeline = pipeline
        self.group = group

    def _shard(self, t):
        return None if t is None else shard_rows(t, dist.get_rank(self.group) if dist.is_initialized() else 0,
                                                 dist.get_world_size(self.group) if dist.is_initialized() else 1)

    def forward(self, points, attributes, adjacency, offsets, rays, start_point, depth_quantiles=None, **kw):
        return self.pipeline.trace_forward(points, attributes, adjacency, offsets, self._shard(rays),
                                           self._shard(start_point), depth_quantiles=self._shard(depth_quantiles),
                                           **kw)

    def backward(self, points, attributes, adjacency, offsets, rays, start_point, rgba_local, grad_local,
                 depth_quantiles=None, depth_indices_local=None, depth_grad_local=None, **kw):
        res = self.pipeline.trace_backward(points, attributes, adjacency, offsets, self._shard(rays),
                                           self._shard(start_point), rgba_local, grad_local,
                                           self._shard(depth_quantiles), depth_indices_local, depth_grad_local, **kw)
        all_reduce_gradients(res, group=self.group)
        return res
