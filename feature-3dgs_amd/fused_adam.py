"""`FusedAdam`: torch.optim.Adam for the Gaussian parameters as ONE HIP launch over all tensors.

A `torch.optim.Optimizer` with torch.optim.Adam's param_groups and STATE LAYOUT (`step`, `exp_avg`, `exp_avg_sq`), so the
reference's densification code, which edits the optimizer state directly (scene/gaussian_model.py:285-382:
`replace_tensor_to_optimizer`, `_prune_optimizer`, `cat_tensors_to_optimizer`), works on it unchanged:

    self.optimizer = FusedAdam(l, lr=0.0, eps=1e-15)        # instead of torch.optim.Adam(l, lr=0.0, eps=1e-15)

No weight decay, no amsgrad (the reference uses neither).  HIP tensors only.
"""
from __future__ import annotations

import torch

from diff_gaussian_rasterization import _C

_C_MAX_TENSORS = 16     # include/f3dgs.h: F3DGS_ADAM_MAX_TENSORS


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, multi_tensor=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.multi_tensor = multi_tensor     # False: one launch per tensor (f3dgs_adam_step), as in round 2

    @torch.no_grad()
    def step(self, closure=None, visibility=None):
        """`visibility` (optional, (P,) bool/uint8 HIP tensor, e.g. `radii > 0`): an EXTENSION beyond the reference -
        only those rows of every (P, ...) tensor are stepped, the others keep parameter and moments (torch.optim.Adam
        would still move a never-visible Gaussian by its decaying first moment).  Default: the reference's dense Adam."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []      # (param, group) with a gradient, state initialised and stepped
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                work.append((p, group))
        # ONE launch over every tensor that shares betas / eps and (with a visibility mask) the row count - the seven
        # per-Gaussian tensors of the reference's model; whatever does not fit that takes the per-tensor entry point
        if self.multi_tensor and work:
            g0 = work[0][1]
            same = [pg for pg in work if pg[1]["betas"] == g0["betas"] and pg[1]["eps"] == g0["eps"] and pg[0].is_contiguous()
                    and (visibility is None or (pg[0].dim() >= 1 and pg[0].shape[0] == visibility.shape[0]))]
            same = same[:_C_MAX_TENSORS]
            if len(same) > 1:
                ps = [p for p, _ in same]
                _C.adam_step_multi(ps, [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                                   [self.state[p]["exp_avg_sq"] for p in ps], [float(g["lr"]) for _, g in same],
                                   g0["betas"][0], g0["betas"][1], g0["eps"], [int(self.state[p]["step"]) for p in ps], visibility)
                done = {id(p) for p in ps}
                work = [pg for pg in work if id(pg[0]) not in done]
        for p, group in work:
            b1, b2 = group["betas"]
            st = self.state[p]
            mask = visibility if visibility is not None and p.dim() >= 1 and p.shape[0] == visibility.shape[0] else None
            _C.adam_step(p, p.grad, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), b1, b2, group["eps"], int(st["step"]), mask)
        return loss
