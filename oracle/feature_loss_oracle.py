"""TEST INFRASTRUCTURE - the feature-map loss of the reference's training loop, stated with plain PyTorch ops exactly as
the reference writes it (train.py:99-105, models/networks.py:107-119, utils/loss_utils.py:17-18):

    feature_map = F.interpolate(feature_map.unsqueeze(0), size=gt.shape[1:], mode='bilinear', align_corners=True).squeeze(0)
    if speedup: feature_map = cnn_decoder(feature_map)          # nn.Conv2d(C, 4C, kernel_size=1)
    Ll1_feature = torch.abs(feature_map - gt_feature_map).mean()

`reference_feature_l1` returns the loss and, through autograd, the gradients the fused HIP op must reproduce.
Only tests may import this module."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def reference_feature_l1(feature_map, gt, weight=None, bias=None, dtype=torch.float64):
    fm = feature_map.detach().to(dtype).requires_grad_(True)
    w = weight.detach().to(dtype).requires_grad_(True) if weight is not None else None
    b = bias.detach().to(dtype).requires_grad_(True) if bias is not None else None
    x = F.interpolate(fm.unsqueeze(0), size=(gt.shape[1], gt.shape[2]), mode="bilinear", align_corners=True).squeeze(0)
    if w is not None:
        x = F.conv2d(x.unsqueeze(0), w.reshape(w.shape[0], w.shape[1], 1, 1), b).squeeze(0)
    loss = torch.abs(x - gt.to(dtype)).mean()
    loss.backward()
    return dict(loss=loss.detach(), d_feature_map=fm.grad, d_weight=None if w is None else w.grad,
                d_bias=None if b is None else b.grad, decoded=x.detach())
