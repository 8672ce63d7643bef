"""Probe: do gloo collectives take CUDA (HIP) tensors on this image?  Two ranks on ONE GPU (RCCL refuses that)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    out = {}
    for name, fn in {
        "all_reduce": lambda: dist.all_reduce(torch.ones(8, device=dev) * (rank + 1)),
        "all_gather": lambda: dist.all_gather([torch.empty(4, device=dev) for _ in range(world)], torch.ones(4, device=dev)),
        "all_gather_into_tensor": lambda: dist.all_gather_into_tensor(torch.empty(8, device=dev), torch.ones(4, device=dev)),
        "all_reduce_int32": lambda: dist.all_reduce(torch.ones(8, device=dev, dtype=torch.int32)),
        "all_reduce_int64_max": lambda: dist.all_reduce(torch.ones(8, device=dev, dtype=torch.int64), op=dist.ReduceOp.MAX),
    }.items():
        try:
            fn()
            torch.cuda.synchronize()
            out[name] = "ok"
        except Exception as exc:  # noqa: BLE001
            out[name] = repr(exc)[:120]
    if rank == 0:
        print(out, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
