"""View-sharded data parallelism for the rasterizer: one process per GPU, one view per rank and step,
RCCL all-reduce of the per-Gaussian gradients over xGMI.

The reference trains on ONE view per iteration on one GPU and contains no torch.distributed call on
this path (reference train.py:73-149), so this is new functionality layered ABOVE
the drop-in op: every rank holds all P Gaussians (2M x (59+256) floats = 2.5 GB at config c4, trivial in
288 GB), renders its own view, and the gradients of the leaf inputs of the op — means3D 3, SH 48,
semantic feature C, opacity 1, scales 3, rotations 4 = (59 + C) floats per Gaussian — are summed.

xGMI on MI355X is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE
link, so the gradients are packed into a few large flat buckets (default 256 MiB) and handed to RCCL as
single collectives, which lets it use its direct/one-shot algorithms across all links; buckets are issued
asynchronously in reverse production order so the first ones overlap the tail of the backward pass.
Densification statistics need two more tiny reductions: SUM of the screen-space gradient norms / visibility
counts and MAX of the radii (reference train.py:132-133, scene/gaussian_model.py:436-438).

Backends: "nccl" (= RCCL on ROCm) on GPUs; "gloo" works for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

GRAD_KEYS = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")


def views_for_rank(num_views: int, rank: int, world: int, iteration: int = 0) -> List[int]:
    """Deterministic sharding: rank r renders views world*iteration + r (mod num_views) — every rank a
    different view of the same step, every view visited equally often."""
    if num_views <= 0:
        return []
    return [(world * iteration + rank) % num_views]


class GradBuckets:
    """Packs a fixed set of gradient tensors into flat fp32 buckets and all-reduces them."""

    def __init__(self, shapes: Dict[str, Sequence[int]], device, bucket_bytes: int = 256 << 20):
        self.layout = []  # (key, bucket, offset, numel, shape)
        self.buckets: List[torch.Tensor] = []
        cap = max(1, bucket_bytes // 4)
        cur, fill = [], 0
        sizes = []
        for k, shp in shapes.items():
            n = 1
            for s in shp:
                n *= int(s)
            if fill and fill + n > cap:
                sizes.append(fill)
                fill = 0
            self.layout.append((k, len(sizes), fill, n, tuple(shp)))
            fill += n
        sizes.append(fill)
        self.buckets = [torch.empty(s, dtype=torch.float32, device=device) for s in sizes]

    def views(self) -> Dict[str, torch.Tensor]:
        """Per-key views INTO the buckets (write gradients here to skip the pack copy)."""
        return {k: self.buckets[b][o:o + n].view(shp) for k, b, o, n, shp in self.layout}

    def pack(self, grads: Dict[str, torch.Tensor]) -> None:
        for k, b, o, n, _ in self.layout:
            self.buckets[b][o:o + n].copy_(grads[k].reshape(-1))

    def all_reduce(self, group=None, async_op: bool = False):
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in reversed(self.buckets)]
        if async_op:
            return works
        for w in works:
            w.wait()
        return None

    def unpack(self) -> Dict[str, torch.Tensor]:
        return self.views()


def all_reduce_gaussian_grads(grads: Dict[str, torch.Tensor], group=None, bucket_bytes: int = 256 << 20,
                              buckets: Optional[GradBuckets] = None,
                              direct_bytes: int = 32 << 20) -> Dict[str, torch.Tensor]:
    """Sum the per-Gaussian gradients over all ranks; returns tensors with the input shapes.

    Tensors of at least `direct_bytes` (SH and feature gradients: 192 MB / 4C MB per million Gaussians) are
    already collective-sized and are reduced IN PLACE without a pack copy; the small ones (means, opacity,
    scales, rotations) share one flat bucket so that they cost one collective instead of four."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    keys = [k for k in GRAD_KEYS if k in grads and grads[k] is not None]
    big = [k for k in keys if grads[k].numel() * 4 >= direct_bytes and grads[k].is_contiguous()]
    small = [k for k in keys if k not in big]
    works = [dist.all_reduce(grads[k], op=dist.ReduceOp.SUM, group=group, async_op=True) for k in big]
    out = dict(grads)
    if small:
        if buckets is None:
            buckets = GradBuckets({k: grads[k].shape for k in small}, grads[small[0]].device, bucket_bytes)
        buckets.pack({k: grads[k] for k in small})
        works += buckets.all_reduce(group, async_op=True)
    for w in works:
        w.wait()
    if small:
        out.update(buckets.unpack())
    return out


def reduce_densification_stats(grad_norm_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                               group=None) -> None:
    """In-place: SUM the accumulated view-space gradient norms and visibility counts, MAX the radii."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    w1 = dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w2 = dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w3 = dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group, async_op=True)
    for w in (w1, w2, w3):
        w.wait()


def dp_step(render_and_backward, leaves: Dict[str, torch.Tensor], view_ids: Iterable[int], group=None,
            buckets: Optional[GradBuckets] = None) -> Dict[str, torch.Tensor]:
    """One data-parallel step: `render_and_backward(view_id)` must run the op forward+backward for that view
    and accumulate into `leaves[k].grad`; afterwards the gradients of all ranks are summed."""
    for v in leaves.values():
        v.grad = None
    for vid in view_ids:
        render_and_backward(vid)
    grads = {k: leaves[k].grad for k in GRAD_KEYS if k in leaves and leaves[k].grad is not None}
    return all_reduce_gaussian_grads(grads, group=group, buckets=buckets)
