"""Fused feature-map loss: bilinear resize (align_corners=True) -> optional 1x1-conv decoder
-> L1 against the ground-truth feature map, forward AND backward in one pass over the data.

Replaces these lines of the reference's training loop (train.py:99-105; decoder: models/networks.py:107-119):

    feature_map = F.interpolate(feature_map.unsqueeze(0), size=(Hg, Wg), mode='bilinear', align_corners=True).squeeze(0)
    if dataset.speedup: feature_map = cnn_decoder(feature_map)
    Ll1_feature = l1_loss(feature_map, gt_feature_map)

with

    Ll1_feature = fused_feature_l1(feature_map, gt_feature_map, cnn_decoder.conv.weight, cnn_decoder.conv.bias)

The (C, H, W) rendered map is read once, the (4C, Hg, Wg) decoded map never exists in memory, and the gradient with
respect to the rendered map - what the rasterizer's backward consumes as dL_dout_feature - is produced by the same
call.  With `lowres_grad=True` that gradient is not even written out at the image's resolution (where 8 of 9 values are
zero when the ground truth is a third of the image): it stays at the LOSS's resolution and the rasterizer's backward
applies the transposed resize tile by tile (include/f3dgs.h: f3dgs_set_feature_grad_lowres).  The decoder is the one dense contraction next to the rasterizer: it runs on the fp32 matrix pipe
(v_mfma_f32_32x32x2_f32, exact fp32).  HIP only (csrc/feature_loss.hip behind include/f3dgs.h); no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from diff_gaussian_rasterization import _C


class _FusedFeatureL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature_map, gt, weight, bias, lowres):
        # lowres: 0 = dense gradient, otherwise the serial number of the rasterizer call that rendered feature_map
        loss, d_fm, d_w, d_b, gx = _C.feature_l1(feature_map, gt, weight, bias, not lowres)
        ctx.has_decoder = weight.numel() > 0
        ctx.lowres = lowres
        if lowres:
            # gx: dL/d(resized map), (Hg, Wg, C); the placeholder stands in for the dense gradient on autograd's side
            ctx.save_for_backward(gx, d_w, d_b, feature_map.new_zeros(()))
            ctx.fm_shape = tuple(feature_map.shape)
        else:
            ctx.save_for_backward(d_fm, d_w, d_b)
        return loss

    @staticmethod
    def backward(ctx, g):
        # the gradients were produced for dL/dloss = 1; the loss is a scalar, so they scale linearly
        if ctx.lowres:
            import diff_gaussian_rasterization as dgr
            gx, d_w, d_b, zero = ctx.saved_tensors
            # handed to the backward call of the rasterizer call that produced feature_map (matched by its serial number); autograd
            # itself carries a zero-stride placeholder, to which any other consumer's dense gradient is simply added
            dgr._offer_feature_grad_lowres(ctx.lowres, gx, g.reshape(()).to(torch.float32))
            d_fm = zero.expand(ctx.fm_shape)
        else:
            d_fm, d_w, d_b = ctx.saved_tensors
            d_fm = d_fm * g
        return d_fm, None, (d_w * g) if ctx.has_decoder else None, (d_b * g) if ctx.has_decoder else None, None


def fused_feature_l1(feature_map: torch.Tensor, gt_feature_map: torch.Tensor, weight: Optional[torch.Tensor] = None,
                     bias: Optional[torch.Tensor] = None, lowres_grad: bool = False) -> torch.Tensor:
    """mean |decode(resize(feature_map)) - gt|.  feature_map (C, H, W), gt (Cout, Hg, Wg), weight (Cout, C) or
    (Cout, C, 1, 1) and bias (Cout) of the 1x1 decoder, or None for no decoder (then Cout == C).

    lowres_grad=True: `feature_map` must be the feature map exactly as `GaussianRasterizer` returned it (not a view or a copy:
    the gradient reaches that call's backward beside autograd; ValueError otherwise).  Where the ground truth is larger than
    the image along an axis the dense path is taken.  The hand-over has ONE slot per render: the first lowres_grad loss on a
    feature map takes it, any further loss on the same map (multi-scale, a second ground truth) takes the dense path by
    itself - the sum of the gradients is the same.  Because the gradient travels beside autograd,
    `torch.autograd.grad(loss, feature_map)` / `feature_map.retain_grad()` see a zero placeholder for a lowres_grad loss, not
    the gradient: use lowres_grad=False where the feature map's own gradient is wanted."""
    e = torch.Tensor([])
    lowres = 0
    if lowres_grad and feature_map.requires_grad:       # (nothing to hand over where no gradient is asked for: evaluation)
        fn = feature_map.grad_fn
        lowres = getattr(feature_map, "_f3dgs_call", 0)
        if fn is None or "_RasterizeGaussians" not in fn.name() or not lowres:
            raise ValueError("lowres_grad=True needs the feature map as returned by GaussianRasterizer (its gradient is handed to "
                             "that call's backward at the loss's resolution); got a tensor produced by "
                             f"{fn.name() if fn is not None else 'no autograd node'}")
        if not (gt_feature_map.shape[-2] <= feature_map.shape[-2] and gt_feature_map.shape[-1] <= feature_map.shape[-1]):
            lowres = 0
        if lowres:
            import diff_gaussian_rasterization as dgr
            if not dgr._claim_feature_grad_lowres(lowres):      # a second loss on this render: dense path
                lowres = 0
    if weight is None:
        return _FusedFeatureL1.apply(feature_map, gt_feature_map, e, e, lowres)
    w2 = weight.reshape(weight.shape[0], -1)
    return _FusedFeatureL1.apply(feature_map, gt_feature_map, w2, bias if bias is not None else torch.zeros(
        weight.shape[0], device=weight.device, dtype=weight.dtype), lowres)


@torch.no_grad()
def fused_feature_decode(feature_map: torch.Tensor, size, weight: Optional[torch.Tensor] = None,
                         bias: Optional[torch.Tensor] = None, half: bool = False) -> torch.Tensor:
    """Forward only (the inference side of the same ops, render.py:169-171 / :137-139 / :294-296):

        feature_map = F.interpolate(feature_map.unsqueeze(0), size=size, mode='bilinear', align_corners=True).squeeze(0)
        if speedup: feature_map = cnn_decoder(feature_map)

    as one call; `half=True` returns fp16 as render.py stores the maps (`.half()`, :179).  Returns (Cout, Hg, Wg)."""
    e = torch.Tensor([])
    if weight is None:
        return _C.feature_decode(feature_map, int(size[0]), int(size[1]), e, e, bool(half))
    w2 = weight.reshape(weight.shape[0], -1)
    b = bias if bias is not None else torch.zeros(weight.shape[0], device=weight.device, dtype=weight.dtype)
    return _C.feature_decode(feature_map, int(size[0]), int(size[1]), w2, b, bool(half))
