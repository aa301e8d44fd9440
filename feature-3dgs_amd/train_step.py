"""Data-parallel training step around the drop-in op.

One call = the body of the reference's training loop between "pick a camera" and "optimizer step"
(reference train.py:84-149): render, loss, backward, densification statistics.  With one process (world size 1)
and one view per step it performs exactly the reference's updates:

    train.py:91        render_pkg = render(viewpoint_cam, gaussians, pipe, background)
    train.py:98-106    loss = (1-l) L1 + l (1 - ssim) + L1_feature        -> `loss_fn(render_pkg, camera)`
    train.py:108       loss.backward()
    train.py:132       gaussians.max_radii2D[vis] = max(gaussians.max_radii2D[vis], radii[vis])
    train.py:133       gaussians.add_densification_stats(viewspace_point_tensor, vis)
                       (scene/gaussian_model.py:436-438: xyz_gradient_accum[vis] += |grad_xy|, denom[vis] += 1)

With N processes (one per GPU, `torch.distributed` initialised) every rank renders ITS views of the same Gaussians;
afterwards every rank holds, for every parameter, the SUM of the gradients over all ranks' views (`dp.dp_step`), and
the densification statistics of all views (per-step deltas reduced with SUM / SUM / MAX), so the replicas stay
identical whatever they do next (optimizer step, densify_and_prune).  N views per step instead of one is a change of
the training recipe ABOVE the op - the op, its gradients and the statistics per view are the reference's.

`model` is duck-typed like the reference's `GaussianModel`: it needs `max_radii2D`, `xyz_gradient_accum`, `denom`
(tensors of length P) and the parameter tensors named in `param_names` (default: the reference's `_xyz`,
`_features_dc`, `_features_rest`, `_opacity`, `_scaling`, `_rotation`, `_semantic_feature`, scene/gaussian_model.py:41-50).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, NamedTuple, Optional, Sequence

import torch

import dp

REFERENCE_PARAM_NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantic_feature")


class StepResult(NamedTuple):
    loss: torch.Tensor                    # sum over this rank's views (detached)
    views: int                            # views rendered by this rank
    grads: Dict[str, torch.Tensor]        # name -> summed gradient (the tensors in `param.grad`)


def densification_deltas(render_pkg: dict, P: int, device) -> tuple:
    """The per-view increments of scene/gaussian_model.py:436-438 and train.py:132, as dense tensors."""
    vis = render_pkg["visibility_filter"]
    vsp = render_pkg["viewspace_points"]
    g = vsp.grad if vsp.grad is not None else torch.zeros_like(vsp)
    norm = torch.zeros(P, 1, device=device)
    norm[vis] = torch.norm(g[vis, :2], dim=-1, keepdim=True)
    count = torch.zeros(P, 1, device=device)
    count[vis] = 1.0
    radii = torch.zeros(P, device=device, dtype=torch.float32)
    radii[vis] = render_pkg["radii"][vis].float()
    return norm, count, radii


def dp_train_step(render: Callable, loss_fn: Callable, model, cameras: Iterable, pipe, background: torch.Tensor,
                  group=None, overlap: bool = True, param_names: Sequence[str] = REFERENCE_PARAM_NAMES,
                  feature_param: str = "_semantic_feature", densification_stats: bool = True,
                  sh_params: Sequence[str] = ("_features_dc", "_features_rest"), rows_chunks: int = 4,
                  reduce: bool = True) -> StepResult:
    """Render + loss + backward for this rank's `cameras`, then make gradients and densification statistics
    global.  `render(camera, model, pipe, background)` is the reference's `gaussian_renderer.render`;
    `loss_fn(render_pkg, camera)` returns the scalar loss of one view.  The optimizer step stays with the caller
    (train.py:146-151), exactly as in the reference.  `sh_params`: the parameters that receive nothing but the op's SH
    gradient (scene/gaussian_model.py:113-116: get_features = cat(_features_dc, _features_rest)); with `overlap` their
    all-reduce runs in `rows_chunks` row ranges inside the backward pass (empty: reduce them afterwards).
    `reduce=False`: the gradients stay LOCAL to the rank (the densification statistics are still made global) - for a sharded
    optimizer, whose reduce-scatter is the exchange: `dp.ShardedOptimizer(...).step(result.grads)` (half the gradient
    bytes of the all-reduce, optimizer work and state divided by the world size)."""
    params = {n: getattr(model, n) for n in param_names if getattr(model, n, None) is not None}
    cameras = list(cameras)
    P = next(iter(params.values())).shape[0]
    device = next(iter(params.values())).device
    pkgs, losses = [], []

    def render_and_backward(i):
        pkg = render(cameras[i], model, pipe, background)
        loss = loss_fn(pkg, cameras[i])
        loss.backward()
        pkgs.append(pkg)
        losses.append(loss.detach())

    def forward(i):
        pkg = render(cameras[i], model, pipe, background)
        loss = loss_fn(pkg, cameras[i])
        pkgs.append(pkg)
        losses.append(loss.detach())
        return loss

    # the overlap needs the feature parameter to be the op's direct input (true for the reference's model:
    # get_semantic_feature returns the parameter itself, scene/gaussian_model.py:121-123)
    if len(cameras) > 1:
        # several views per rank: pipelined over two streams, the feature gradient accumulated in place across the views and
        # reduced from inside the last view's backward pass (dp.dp_step_views)
        grads = dp.dp_step_views(forward, lambda loss: loss.backward(), params, range(len(cameras)), group=group,
                                 overlap=overlap and reduce, feature_key=feature_param,
                                 accumulate=None if overlap else False, reduce=reduce)
    else:
        grads = dp.dp_step(render_and_backward, params, range(len(cameras)), group=group,
                           overlap=overlap and reduce and len(cameras) == 1, feature_key=feature_param,
                           rows_leaves={"sh": tuple(sh_params)} if sh_params else None, rows_chunks=rows_chunks, reduce=reduce)

    if densification_stats:
        with torch.no_grad():
            norm = torch.zeros(P, 1, device=device)
            count = torch.zeros(P, 1, device=device)
            radii = torch.zeros(P, device=device, dtype=torch.float32)
            for pkg in pkgs:
                n, c, r = densification_deltas(pkg, P, device)
                norm += n
                count += c
                radii = torch.maximum(radii, r)
            dp.reduce_densification_stats(norm, count, radii, group=group)
            model.xyz_gradient_accum += norm
            model.denom += count
            seen = count.squeeze(-1) > 0
            mr = model.max_radii2D
            mr[seen] = torch.maximum(mr[seen], radii[seen].to(mr.dtype))
    total = torch.stack(losses).sum() if losses else torch.zeros((), device=device)
    return StepResult(total, len(cameras), grads)


def _default_set_band(r0: int, r1: int) -> None:
    from diff_gaussian_rasterization import _C
    _C.set_tile_band(r0, r1)


def dp_train_step_bands(render: Callable, loss_fn: Callable, model, camera, pipe, background: torch.Tensor, group=None,
                        param_names: Sequence[str] = REFERENCE_PARAM_NAMES, densification_stats: bool = True,
                        image_keys: Sequence[str] = ("render", "feature_map", "depth"), reduce: bool = True,
                        set_band: Callable[[int, int], None] = _default_set_band) -> StepResult:
    """ONE view split over the ranks by tile rows ("alternative for single huge views"; include/f3dgs.h:
    f3dgs_set_tile_band) - for steps that have fewer views than GPUs.  Every rank renders the same `camera` restricted to its
    band of 16-pixel tile rows (`dp.band_rows`), the bands are gathered into whole images (`dp.gather_bands` on
    `render_pkg[k]` for k in `image_keys`: the images the loss reads), `loss_fn(render_pkg, camera)` sees the whole view on
    every rank - so window-based terms such as SSIM are exact at the band borders - and its backward pass reaches this rank's
    rows only; the gradients of the ranks then SUM to the whole view's gradient (`dp.all_reduce_gaussian_grads`), the radii MAX
    to the whole view's, and the densification statistics are those of ONE view: |sum of the screen-space gradients| and one
    visibility count per Gaussian seen in any band (scene/gaussian_model.py:436-438).  The loss returned is the whole view's, the
    same on every rank."""
    params = {n: getattr(model, n) for n in param_names if getattr(model, n, None) is not None}
    P = next(iter(params.values())).shape[0]
    device = next(iter(params.values())).device
    world = torch.distributed.get_world_size(group) if dp._active(group) else 1
    rank = torch.distributed.get_rank(group) if world > 1 else 0
    H = int(camera.image_height)
    r0, r1, _y0, _y1 = dp.band_rows(H, rank, world)
    for p in params.values():
        p.grad = None
    if world > 1:       # (1 << 20: "an empty band", the share of a rank beyond the last tile row)
        set_band(r0, r1) if r1 > r0 else set_band(1 << 20, 1 << 20)
    try:
        pkg = render(camera, model, pipe, background)
    finally:
        if world > 1:
            set_band(0, 0)
    whole = dict(pkg)
    for k in image_keys:
        if k in whole and isinstance(whole[k], torch.Tensor) and whole[k].dim() >= 2 and whole[k].numel():
            whole[k] = dp.gather_bands(whole[k], group=group)
    loss = loss_fn(whole, camera)
    loss.backward()
    grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in params.items()}
    for n, p in params.items():
        p.grad = grads[n]
    if reduce and world > 1:
        dp.all_reduce_gaussian_grads(grads, group=group)
    if densification_stats:
        with torch.no_grad():
            vsp = pkg["viewspace_points"]
            g2 = (vsp.grad if vsp.grad is not None else torch.zeros_like(vsp))[:, :2].contiguous()
            radii = pkg["radii"].float().clone()
            if world > 1:       # one view: its screen-space gradient is the SUM over the bands, its radii their MAX
                torch.distributed.all_reduce(g2, op=torch.distributed.ReduceOp.SUM, group=group)
                torch.distributed.all_reduce(radii, op=torch.distributed.ReduceOp.MAX, group=group)
            seen = radii > 0
            norm = torch.zeros(P, 1, device=device)
            norm[seen] = torch.norm(g2[seen], dim=-1, keepdim=True)
            model.xyz_gradient_accum += norm
            model.denom += seen.float().unsqueeze(-1)
            mr = model.max_radii2D
            mr[seen] = torch.maximum(mr[seen], radii[seen].to(mr.dtype))
    return StepResult(loss.detach(), 1, grads)
