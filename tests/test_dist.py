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
        cam, rays_np, start = H.camera_setup(fm, 20, 13)   # 13 rows: uneven split 7 + 6
        t = torch.from_numpy
        p, a, adj, off = t(fm["points"]), t(fm["attributes"]), t(fm["point_adjacency"]), t(fm["point_adjacency_offsets"])
        rays = t(rays_np)
        starts = torch.full(rays.shape[:-1], int(start), dtype=torch.int64).to(torch.uint32)
        g = torch.from_numpy(np.random.default_rng(0).normal(size=rays.shape[:-1] + (4,)).astype(np.float32))

        pipe = OraclePipeline(d)
        tracer = rdist.ShardedTracer(pipe)
        fwd = tracer.forward(p, a, adj, off, rays, starts)
        b, e = rdist.row_block(13, rank, world)
        assert fwd["rgba"].shape == (e - b, 20, 4)
        image = rdist.gather_rows(fwd["rgba"], 13)
        bwd = tracer.backward(p, a, adj, off, rays, starts, fwd["rgba"], rdist.shard_rows(g, rank, world))

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


@pytest.mark.parametrize("d", [0, 2])
def test_row_sharded_forward_backward_gloo(d):
    world = 2
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
    assert dict(results) == {0: 1, 1: 1}


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
