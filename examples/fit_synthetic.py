"""Minimal optimisation loop over the drop-in operators (what train.py does, without datasets):
render a target image from a hidden foam, then fit the colour coefficients and densities of a second
foam with the same triangulation to it by Adam -- packing (radfoam.pack_attributes), forward and backward
(radfoam_amd.render.TraceRays, the reference's autograd signature) all run the HIP kernels.

    python examples/fit_synthetic.py [--points 20000] [--steps 60]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import radfoam  # noqa: E402
from radfoam_amd import foam  # noqa: E402
from radfoam_amd.render import TraceRays  # noqa: E402


def inverse_softplus(y, beta=10.0):
    return torch.log(torch.expm1((y * beta).clamp(min=1e-6))) / beta


def fit(num_points=20000, sh_degree=1, width=160, height=120, steps=60, lr=5e-2, seed=0, device="cuda:0",
        log=print):
    fm = foam.make_synthetic_foam(num_points, sh_degree, seed)
    dev = torch.device(device)
    pts = torch.from_numpy(fm["points"]).to(dev)
    adj = torch.from_numpy(fm["point_adjacency"]).to(dev)
    off = torch.from_numpy(fm["point_adjacency_offsets"]).to(dev)
    target_attr = torch.from_numpy(fm["attributes"]).to(dev)
    cam = foam.default_camera(width, height)
    rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
    start = radfoam.nn(pts, radfoam.build_aabb_tree(pts), torch.from_numpy(cam["position"]).to(dev)[None])
    start = torch.broadcast_to(start, rays.shape[:-1]).contiguous()
    pipe = radfoam.create_pipeline(sh_degree, torch.float32)

    with torch.no_grad():
        target = pipe.trace_forward(pts, target_attr, adj, off, rays, start)["rgba"]

    # the learner starts from grey, semi-transparent cells inside the ball the target occupies
    n = pts.size(0)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    att_dc = (0.05 * torch.randn(n, 3, generator=g)).to(dev).requires_grad_(True)
    att_sh = torch.zeros(n, 3 * ((sh_degree + 1) ** 2 - 1), device=dev, requires_grad=True)
    inside = (pts.norm(dim=1, keepdim=True) <= 0.8).float()
    mean_density = target_attr[:, -1:].mean().clamp(min=1e-3)
    density = (inverse_softplus(0.5 * mean_density * inside + 1e-4)).clone().requires_grad_(True)
    opt = torch.optim.Adam([att_dc, att_sh, density], lr=lr)

    losses = []
    for it in range(steps):
        opt.zero_grad(set_to_none=True)
        attributes = radfoam.pack_attributes(att_dc, att_sh, density, 1.0, torch.float32)
        rgba, _, _, _, _ = TraceRays.apply(pipe, pts, attributes, adj, off, rays, start, None, False)
        loss = (rgba - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if it % 10 == 0 or it == steps - 1:
            log(f"step {it:4d}  mse {losses[-1]:.6f}")
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    ls = fit(num_points=a.points, steps=a.steps)
    print(f"mse {ls[0]:.6f} -> {ls[-1]:.6f}")
