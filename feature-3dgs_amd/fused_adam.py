"""`FusedAdam`: torch.optim.Adam for the Gaussian parameters with one HIP pass per tensor.

A `torch.optim.Optimizer` with torch.optim.Adam's param_groups and STATE LAYOUT (`step`, `exp_avg`, `exp_avg_sq`), so the
reference's densification code, which edits the optimizer state directly (scene/gaussian_model.py:285-382:
`replace_tensor_to_optimizer`, `_prune_optimizer`, `cat_tensors_to_optimizer`), works on it unchanged:

    self.optimizer = FusedAdam(l, lr=0.0, eps=1e-15)        # instead of torch.optim.Adam(l, lr=0.0, eps=1e-15)

No weight decay, no amsgrad (the reference uses neither).  HIP tensors only.
"""
from __future__ import annotations

import torch

from diff_gaussian_rasterization import _C


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, visibility=None):
        """`visibility` (optional, (P,) bool/uint8 HIP tensor, e.g. `radii > 0`): an EXTENSION beyond the reference -
        only those rows of every (P, ...) tensor are stepped, the others keep parameter and moments (torch.optim.Adam
        would still move a never-visible Gaussian by its decaying first moment).  Default: the reference's dense Adam."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                mask = visibility if visibility is not None and p.dim() >= 1 and p.shape[0] == visibility.shape[0] else None
                _C.adam_step(p, p.grad, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), b1, b2, group["eps"],
                             int(st["step"]), mask)
        return loss
