"""Optimisation loop WITH moving points (what train.py:160-248 does between two densifications, without datasets):
the learner's point positions, colours and densities are all fitted to a target image by Adam, and every
``rebuild_every`` steps the triangulation follows the points -- ``Triangulation.rebuild(points, incremental=True)``
as RadFoamScene.update_triangulation calls it (radfoam_model/scene.py:160-200) -- on the GPU
(radfoam_amd/triangulation.py: one Delaunay star per point, the previous neighbour lists as candidates).

    python examples/fit_points.py [--points 20000] [--steps 60] [--rebuild-every 5]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import radfoam  # noqa: E402
from radfoam_amd import foam  # noqa: E402
from radfoam_amd.render import TraceRays  # noqa: E402
from examples.fit_synthetic import inverse_softplus  # noqa: E402


def fit(num_points=20000, sh_degree=1, width=160, height=120, steps=60, rebuild_every=5, lr=2e-2, point_lr=2e-4,
        seed=0, device="cuda:0", log=print):
    fm = foam.make_synthetic_foam(num_points, sh_degree, seed)
    dev = torch.device(device)
    target_pts = torch.from_numpy(fm["points"]).to(dev)
    target_attr = torch.from_numpy(fm["attributes"]).to(dev)
    cam = foam.default_camera(width, height)
    rays = torch.from_numpy(foam.camera_rays(cam)).to(dev)
    cam_pos = torch.from_numpy(cam["position"]).to(dev)[None]
    pipe = radfoam.create_pipeline(sh_degree, torch.float32)

    tri = radfoam.Triangulation(target_pts)   # the foam's points are in kd-order already: identity permutation
    adj, off = tri.point_adjacency(), tri.point_adjacency_offsets()
    start = torch.broadcast_to(radfoam.nn(target_pts, radfoam.build_aabb_tree(target_pts), cam_pos),
                               rays.shape[:-1]).contiguous()
    with torch.no_grad():
        target = pipe.trace_forward(target_pts, target_attr, adj, off, rays, start)["rgba"]

    # the learner: the same cells, jittered by a fifth of the spacing, grey and half as dense
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    n = target_pts.size(0)
    spacing = (8.0 / n) ** (1.0 / 3.0)
    pts = (target_pts + 0.2 * spacing * torch.randn(n, 3, generator=g).to(dev)).requires_grad_(True)
    att_dc = (0.05 * torch.randn(n, 3, generator=g)).to(dev).requires_grad_(True)
    att_sh = torch.zeros(n, 3 * ((sh_degree + 1) ** 2 - 1), device=dev, requires_grad=True)
    inside = (target_pts.norm(dim=1, keepdim=True) <= 0.8).float()
    density = inverse_softplus(0.5 * target_attr[:, -1:].mean().clamp(min=1e-3) * inside + 1e-4).clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [att_dc, att_sh, density], "lr": lr}, {"params": [pts], "lr": point_lr}])

    tri.rebuild(pts.detach(), incremental=True)   # same point count, same order: the jittered points' own lists
    adj, off = tri.point_adjacency(), tri.point_adjacency_offsets()
    losses, rebuild_ms = [], []
    for it in range(steps):
        opt.zero_grad(set_to_none=True)
        attributes = radfoam.pack_attributes(att_dc, att_sh, density, 1.0, torch.float32)
        rgba, _, _, _, _ = TraceRays.apply(pipe, pts, attributes, adj, off, rays, start, None, False)
        loss = (rgba - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if (it + 1) % rebuild_every == 0:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            needs_permute = tri.rebuild(pts.detach(), incremental=True)
            torch.cuda.synchronize(dev)
            rebuild_ms.append((time.perf_counter() - t0) * 1e3)
            assert needs_permute is False   # incremental: no re-sort (delaunay.cu:293-311)
            adj, off = tri.point_adjacency(), tri.point_adjacency_offsets()
            start = torch.broadcast_to(radfoam.nn(pts.detach(), radfoam.build_aabb_tree(pts.detach()), cam_pos),
                                       rays.shape[:-1]).contiguous()
        if it % 10 == 0 or it == steps - 1:
            log(f"step {it:4d}  mse {losses[-1]:.6f}")
    return dict(losses=losses, rebuild_ms=rebuild_ms, points=pts.detach(), adjacency=adj, offsets=off)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rebuild-every", type=int, default=5)
    a = ap.parse_args()
    r = fit(num_points=a.points, steps=a.steps, rebuild_every=a.rebuild_every)
    print(f"mse {r['losses'][0]:.6f} -> {r['losses'][-1]:.6f}; {len(r['rebuild_ms'])} rebuilds, "
          f"{sum(r['rebuild_ms']) / max(len(r['rebuild_ms']), 1):.1f} ms each")
