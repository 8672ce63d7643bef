"""CPU, world_size 2, gloo: the row-sharded data-parallel path (radfoam_amd/dist.py).

The HIP kernels cannot run here, so the ranks trace their row blocks with the CPU oracle wrapped
in the Pipeline interface; what is under test is the sharding, the single flat-buffer SUM
all-reduce and the image gather -- the same code bench.py --gpus N runs over RCCL.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OraclePipeline:
    """Pipeline-shaped wrapper over the CPU oracle (test infrastructure only)."""

    def __init__(self, d):
        self.d = d

    def attribute_dim(self):
        return 1 + 3 * (self.d + 1) ** 2

    def trace_forward(self, points, attributes, adj, off, rays, start, depth_quantiles=None, **kw):
        from oracle import oracle as O

        q = None if depth_quantiles is None else depth_quantiles.numpy()
        out = O.trace_forward(self.d, points.numpy(), attributes.numpy(), adj.numpy(), off.numpy(), rays.numpy(),
                              start.numpy(), depth_quantiles=q, num_threads=1)
        return {k: torch.from_numpy(v) for k, v in out.items()}

    def trace_backward(self, points, attributes, adj, off, rays, start, rgba, grad, depth_quantiles=None,
                       depth_indices=None, depth_grad_in=None, **kw):
        from oracle import oracle as O

        npy = lambda t: None if t is None else t.numpy()
        out = O.trace_backward(self.d, points.numpy(), attributes.numpy(), adj.numpy(), off.numpy(), rays.numpy(),
                               start.numpy(), rgba.numpy(), grad.numpy(), depth_quantiles=npy(depth_quantiles),
                               depth_indices=npy(depth_indices), depth_grad_in=npy(depth_grad_in), num_threads=1)
        n, a = points.shape[0], self.attribute_dim()
        flat = torch.zeros(n * (3 + a), dtype=torch.float32)
        flat[: 3 * n] = torch.from_numpy(out["points_grad"]).reshape(-1)
        flat[3 * n:] = torch.from_numpy(out["attr_grad"]).reshape(-1)
        return {"points_grad": flat[: 3 * n].view(n, 3), "attr_grad": flat[3 * n:].view(n, a), "flat_grad": flat}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, d, results):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from radfoam_amd import dist as rdist
    from radfoam_amd import foam
    from tests import helpers as H

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fm = foam.make_synthetic_foam(1500, d, 4)
        cam, rays_np, start = H.camera_setup(fm, 20, 13)   # 13 rows: uneven split 7 + 6 (or 5 + 4 + 4)
        t = torch.from_numpy
        p, a, adj, off = t(fm["points"]), t(fm["attributes"]), t(fm["point_adjacency"]), t(fm["point_adjacency_offsets"])
        rays = t(rays_np)
        starts = torch.full(rays.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32)
        g = torch.from_numpy(np.random.default_rng(0).normal(size=rays.shape[:-1] + (4,)).astype(np.float32))

        pipe = OraclePipeline(d)
        tracer = rdist.ShardedTracer(pipe, exchange="dense")
        fwd = tracer.forward(p, a, adj, off, rays, starts)
        b, e = rdist.row_block(13, rank, world)
        assert fwd["rgba"].shape == (e - b, 20, 4)
        image = rdist.gather_rows(fwd["rgba"], 13)
        bwd = tracer.backward(p, a, adj, off, rays, starts, fwd["rgba"], rdist.shard_rows(g, rank, world))

        # the sparse exchange gives the dense all-reduce's sums: exactly for two ranks (one addition per
        # element, commutative), and the same bits on every rank for any world size
        sparse = rdist.ShardedTracer(pipe, exchange="sparse")
        sparse.sparse.dense_fraction = 10.0          # this toy foam is touched almost everywhere: no fallback
        bwd_s = sparse.backward(p, a, adj, off, rays, starts, fwd["rgba"], rdist.shard_rows(g, rank, world))
        assert sparse.sparse.last_counts is not None and len(sparse.sparse.last_counts) == world
        assert 0 < max(sparse.sparse.last_counts) <= p.shape[0]
        if world == 2:
            assert torch.equal(bwd_s["flat_grad"], bwd["flat_grad"])
        else:
            assert torch.allclose(bwd_s["flat_grad"], bwd["flat_grad"], rtol=1e-5, atol=1e-6)
        same = [torch.empty_like(bwd_s["flat_grad"]) for _ in range(world)]
        dist.all_gather(same, bwd_s["flat_grad"])
        assert all(torch.equal(same[0], x) for x in same[1:]), "ranks disagree on the summed gradients"
        # a second call reuses the buffers; a tiny initial capacity exercises the grow-and-retry path
        small = rdist.SparseGradExchange(dense_fraction=10.0)
        small._buffers(3, world, small._pitch(pipe.attribute_dim()), p)
        res2 = pipe.trace_backward(p, a, adj, off, sparse._shard(rays), sparse._shard(starts), fwd["rgba"],
                                   rdist.shard_rows(g, rank, world))
        small.reduce(res2)
        assert torch.equal(res2["flat_grad"], bwd_s["flat_grad"])
        # fallback to the dense all-reduce when the lists are not sparse
        dense_fb = rdist.SparseGradExchange(dense_fraction=0.0)
        res3 = pipe.trace_backward(p, a, adj, off, sparse._shard(rays), sparse._shard(starts), fwd["rgba"],
                                   rdist.shard_rows(g, rank, world))
        dense_fb.reduce(res3)
        assert dense_fb.last_counts is None and torch.allclose(res3["flat_grad"], bwd["flat_grad"], rtol=1e-5, atol=1e-6)

        # cost-balanced row blocks: same image, same gradients, boundaries agreed by all ranks
        bounds = sparse.rebalance(fwd["num_intersections"], 13, align=1)
        assert bounds[0] == 0 and bounds[-1] == 13 and all(x < y for x, y in zip(bounds, bounds[1:]))
        fwd_b = sparse.forward(p, a, adj, off, rays, starts)
        assert fwd_b["rgba"].shape[0] == bounds[rank + 1] - bounds[rank]
        image_b = rdist.gather_rows(fwd_b["rgba"], 13, bounds=bounds)
        assert torch.equal(image_b, image)
        bwd_b = sparse.backward(p, a, adj, off, rays, starts, fwd_b["rgba"],
                                rdist.shard_rows(g, rank, world, bounds=bounds))
        assert torch.allclose(bwd_b["flat_grad"], bwd["flat_grad"], rtol=1e-5, atol=1e-6)

        full_f = pipe.trace_forward(p, a, adj, off, rays, starts)
        full_b = pipe.trace_backward(p, a, adj, off, rays, starts, full_f["rgba"], g)
        assert torch.equal(image, full_f["rgba"]), "gathered image differs from the single-process image"
        for k in ("points_grad", "attr_grad"):
            err = (bwd[k] - full_b[k]).norm() / full_b[k].norm()
            assert err < 1e-5, (k, float(err))
            assert full_b[k].abs().max() > 0
        # both views alias the reduced flat buffer
        assert bwd["flat_grad"].data_ptr() == bwd["points_grad"].data_ptr()
        results[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("d,world", [(0, 2), (2, 2), (1, 3)])
def test_row_sharded_forward_backward_gloo(d, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, d, results)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=240)
    for pr in procs:
        if pr.is_alive():
            pr.kill()
            pytest.fail("distributed worker hung")
        assert pr.exitcode == 0
    assert dict(results) == {r: 1 for r in range(world)}


def test_row_blocks_partition():
    from radfoam_amd import dist as rdist

    for rows, world in [(1080, 8), (13, 2), (5, 8), (7, 3)]:
        blocks = [rdist.row_block(rows, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == rows
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        sizes = [e - b for b, e in blocks]
        assert max(sizes) - min(sizes) <= 1
    x = torch.arange(26).reshape(13, 2)
    assert torch.equal(torch.cat([rdist.shard_rows(x, r, 2) for r in range(2)]), x)


def test_balanced_row_blocks():
    import numpy as np
    from radfoam_amd import dist as rdist

    rows = np.arange(1080)
    cost = 100.0 + 60.0 * np.abs(rows - 540) / 540.0          # frame edges cost more than the centre
    for world in (2, 4, 8):
        even = [cost[slice(*rdist.row_block(1080, r, world))].sum() for r in range(world)]
        b = rdist.balanced_row_blocks(cost, world, align=8)
        assert b[0] == 0 and b[-1] == 1080 and len(b) == world + 1
        assert all(x % 8 == 0 for x in b[:-1]) and all(x < y for x, y in zip(b, b[1:]))
        bal = [cost[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(bal) / np.mean(bal) <= 1.07                 # within one 8-row band per boundary
        if world == 8:                                        # where the even split is visibly unbalanced
            assert max(even) / np.mean(even) > 1.15 and max(bal) / np.mean(bal) < max(even) / np.mean(even)
    # degenerate inputs: zero cost -> even bands; fewer bands than ranks -> plain even split
    assert rdist.balanced_row_blocks([0.0] * 64, 4, align=8) == [0, 16, 32, 48, 64]
    assert rdist.balanced_row_blocks([1.0] * 5, 8, align=8) == [0, 1, 2, 3, 4, 5, 5, 5, 5]
    # all the cost in one row: every block still gets at least one band
    spike = [0.0] * 64
    spike[3] = 1.0
    b = rdist.balanced_row_blocks(spike, 4, align=8)
    assert b[0] == 0 and b[-1] == 64 and all(x < y for x, y in zip(b, b[1:]))


def test_sparse_exchange_takes_gradient_rows_of_any_pitch():
    """Pipeline.trace_backward accumulates attr_grad in rows padded to 64-byte lines by default and returns a
    [N, A] VIEW of them.  The exchange reads the row pitch off the view (rf_compact_grad_rows_pitched /
    rf_scatter_grad_rows_pitched on the GPU) instead of depending on N's divisibility: the packed rows are the same
    whether the rows are dense or padded, for N % 16 == 0 and N % 16 != 0; the dense all-reduce is the one that asks for
    dense rows (when there is more than one rank)."""
    import torch

    from radfoam_amd import dist as rdist

    class _Pipe:
        gradient_row_pitch = "auto"

    pipe = _Pipe()
    rdist.ShardedTracer(pipe, exchange="sparse")
    assert pipe.gradient_row_pitch == "auto"
    rdist.ShardedTracer(pipe, exchange="dense")             # a single process: nothing to exchange, nothing to change
    assert pipe.gradient_row_pitch == "auto"
    ex = rdist.SparseGradExchange()
    for n in (32, 37):
        a = 13
        g = torch.Generator().manual_seed(n)
        dense = torch.randn(n, a, generator=g) * (torch.rand(n, 1, generator=g) < 0.4)
        pg = torch.randn(n, 3, generator=g) * (dense.abs().sum(1, keepdim=True) > 0)
        padded = torch.zeros(n, 16)
        padded[:, :a] = dense
        view = padded[:, :a]
        assert not view.is_contiguous() and ex._row_pitch(view) == 16 and ex._row_pitch(dense) == a
        packed = []
        for rows in (dense, view):
            send = torch.zeros(n, ex._pitch(a))
            count = torch.zeros(1, dtype=torch.int32)
            ex._compact(pg, rows, send, count)
            k = int(count)
            packed.append(send[:k].clone())
            out_pg, out = torch.zeros(n, 3), torch.zeros(n, 16)
            ex._scatter(send, k, out_pg, out[:, :a], zero=False)
            assert torch.equal(out[:, :a], dense) and torch.equal(out_pg, pg) and not out[:, a:].any()
        assert torch.equal(packed[0], packed[1])
    with pytest.raises(RuntimeError, match="contiguous floats"):
        ex._row_pitch(torch.zeros(8, 26)[:, ::2])
