"""Autograd operator over the pipeline: same call signature and outputs as the reference's
``radfoam_model.render.TraceRays`` (radfoam_model/render.py:10-122), so scene code written
against the reference can use either.

``TraceRays.apply(pipeline, points, attributes, point_adjacency, point_adjacency_offsets, rays,
start_point, depth_quantiles, return_contribution)`` ->
``(rgba, depth | None, contribution | None, num_intersections, errbox)``.
Gradients flow to ``points`` and ``attributes`` only; non-finite gradient entries are zeroed
after accumulation (render.py:98-99).  Setting ``errbox.ray_error`` before calling backward
makes it deposit ``errbox.point_error`` (the densification statistic, scene.py:497-548).
"""
from __future__ import annotations

import torch


class ErrorBox:
    def __init__(self):
        self.ray_error = None
        self.point_error = None


class TraceRays(torch.autograd.Function):
    @classmethod
    def apply(cls, pipeline, points, attributes, *rest):
        """Whether a backward can follow is known HERE and nowhere below: inside forward() grad mode is off and
        ctx.needs_input_grad only repeats the inputs' requires_grad flags (an nn.Parameter keeps its flag under
        torch.no_grad()), so Pipeline._wants_trail's "auto" cannot see a caller that optimises only the points -- or only
        the rays (a camera).  The caller's grad mode and the differentiable inputs decide; the answer travels to
        trace_forward as an ARGUMENT (record_trail), nothing on the shared pipeline object is touched: one Pipeline may
        serve several threads or streams."""
        rays = rest[2] if len(rest) > 2 else None
        wanted = bool(torch.is_grad_enabled() and (points.requires_grad or attributes.requires_grad or
                                                   (torch.is_tensor(rays) and rays.requires_grad)))
        return super().apply(pipeline, points, attributes, *rest, wanted)

    @staticmethod
    def forward(ctx, pipeline, points, attributes, point_adjacency, point_adjacency_offsets, rays,
                start_point, depth_quantiles, return_contribution, record_trail=None):
        # (a pipeline object with the reference binding's exact signature is served too: the hint is an extension)
        extra = {"record_trail": record_trail} if getattr(pipeline, "accepts_record_trail", False) else {}
        out = pipeline.trace_forward(points, attributes, point_adjacency, point_adjacency_offsets, rays,
                                     start_point, depth_quantiles=depth_quantiles,
                                     return_contribution=return_contribution, **extra)
        box = ErrorBox()
        ctx.pipeline = pipeline
        ctx.box = box
        ctx.foam = (points, attributes, point_adjacency, point_adjacency_offsets)
        ctx.ray_args = (rays, start_point, depth_quantiles)
        ctx.fwd = (out["rgba"], out.get("depth_indices"))
        return out["rgba"], out.get("depth"), out.get("contribution"), out["num_intersections"], box

    @staticmethod
    def backward(ctx, grad_rgba, grad_depth, _grad_contribution, _grad_count, _grad_box):
        rays, start_point, depth_quantiles = ctx.ray_args
        rgba, depth_indices = ctx.fwd
        res = ctx.pipeline.trace_backward(*ctx.foam, rays, start_point, rgba, grad_rgba, depth_quantiles,
                                          depth_indices, grad_depth, ctx.box.ray_error)
        ctx.box.point_error = res.get("point_error")
        points_grad, attr_grad = res["points_grad"], res["attr_grad"]
        points_grad.masked_fill_(~points_grad.isfinite(), 0)
        attr_grad.masked_fill_(~attr_grad.isfinite(), 0)
        ctx.foam = ctx.ray_args = ctx.fwd = ctx.pipeline = None
        return None, points_grad, attr_grad, None, None, None, None, None, None, None
