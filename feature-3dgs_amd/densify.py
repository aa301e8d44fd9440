"""Densification on stable-shape buffers: the reference's `densify_and_prune` /
`prune_points` as a PLAN (which output row comes from which source row) plus ONE gather launch over every
per-Gaussian tensor and its Adam moments.

Drop-in for the two calls the reference's training loop makes on its model (train.py:135-141):

    gaussians.densify_and_prune(grad_threshold, 0.005, scene.cameras_extent, size_threshold)
        ->  densify.densify_and_prune(gaussians, grad_threshold, 0.005, scene.cameras_extent, size_threshold)
    gaussians.prune_points(mask)  ->  densify.prune_points(gaussians, mask)

`model` is duck-typed like the reference's `GaussianModel` (scene/gaussian_model.py): the seven parameters `_xyz`,
`_features_dc`, `_features_rest`, `_opacity`, `_scaling`, `_rotation`, `_semantic_feature`, the statistics
`xyz_gradient_accum`, `denom`, `max_radii2D`, `percent_dense`, and `optimizer` with one param group per tensor named
like the reference's (`xyz`, `f_dc`, ..., :168-176).  After the call the model is in the state the reference's methods
leave it in - same row order, same values, same optimizer-state edits (new `nn.Parameter` objects registered in the
groups, `exp_avg`/`exp_avg_sq` carried for kept rows and zero for new ones, `step` untouched; everything bit for bit
except the children's positions, which differ by the rounding of a 3-term dot product, see below):

  densify_and_clone (:398-413)   rows with |grad| >= t and max scale <= percent_dense * extent are appended as copies
  densify_and_split (:378-396)   rows (of the grown set; clones carry gradient 0) with grad >= t and max scale >
                                 percent_dense * extent are replaced by N children sampled from the row's Gaussian
                                 (torch.normal with the reference's call shapes, so a seeded run draws the same
                                 samples), scale divided by 0.8 N; the children are appended k-major (`repeat(N, 1)`)
  densify_and_prune (:415-431)   rows with opacity < min_opacity or (when max_screen_size) a world-space scale above
                                 0.1 * extent are dropped.  The screen-size test reads `max_radii2D`, which
                                 `densification_postfix` has just reset to zero for every row (:376) - it is kept
                                 here as the same (always false for a positive threshold) test.
  statistics                     xyz_gradient_accum, denom, max_radii2D come out as zeros (:374-376)

What differs is the execution: the reference re-materialises each of the 7 + 14 tensors four times (cat, cat, mask,
mask: ~120 launches and P-sized allocator churn per densification); here the masks are folded into one row plan and
every tensor is written once, by one launch (csrc/densify.hip), into capacity-sized ping-pong buffers (`RowPool`) that
are only reallocated when the point count outgrows them - the rasterizer's workspace and the allocator see stable
shapes between densifications.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from diff_gaussian_rasterization import _C

# optimizer group name -> model attribute (scene/gaussian_model.py:168-176, :318-324)
GROUPS: Tuple[Tuple[str, str], ...] = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"),
                                       ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation"),
                                       ("semantic_feature", "_semantic_feature"))
_COPY, _ZERO_NEW, _OVERRIDE_CHILD = 0, 1, 2     # include/f3dgs.h F3DGS_DENSIFY_*


class RowPool:
    """Two capacity-sized buffers per tensor: a gather reads the buffer the model currently lives in and writes the
    other one.  A buffer is reallocated (capacity = growth * rows) only when the row count outgrows it."""

    def __init__(self, growth: float = 1.5, min_rows: int = 1024):
        self.growth, self.min_rows = growth, min_rows
        self._bufs: Dict[str, List[Optional[torch.Tensor]]] = {}
        self.reallocations = 0

    def out(self, key: str, rows: int, like: torch.Tensor) -> torch.Tensor:
        """A (capacity, *like.shape[1:]) buffer with capacity >= rows that does not share storage with `like`."""
        pair = self._bufs.setdefault(key, [None, None])
        busy = like.untyped_storage().data_ptr() if like.numel() else None
        slot = 0
        for i, b in enumerate(pair):
            if b is None or b.untyped_storage().data_ptr() != busy:
                slot = i
                break
        b = pair[slot]
        if b is None or b.shape[0] < rows or b.shape[1:] != like.shape[1:] or b.device != like.device:
            cap = max(self.min_rows, int(rows * self.growth) + 1)
            b = torch.empty((cap,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
            pair[slot] = b
            self.reallocations += 1
        return b

    def capacity(self, key: str) -> int:
        return min((b.shape[0] for b in self._bufs.get(key, []) if b is not None), default=0)


def _pool_of(model, pool: Optional[RowPool]) -> RowPool:
    if pool is not None:
        return pool
    p = getattr(model, "_row_pool", None)
    if p is None:
        p = RowPool()
        model._row_pool = p
    return p


def _unit_rotation_matrices(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrices of the normalised quaternions (w, x, y, z) - what the reference's `build_rotation` returns
    (utils/general_utils.py:78-99)."""
    q = q / torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = (1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y))
    return torch.stack(rows, dim=-1).reshape(-1, 3, 3)


def _apply_plan(model, pool: RowPool, src_row: torch.Tensor, kind: torch.Tensor, override_row: torch.Tensor,
                overrides: Dict[str, torch.Tensor], stats: str) -> None:
    """One gather launch for all tensors, then the reference's bookkeeping: new Parameters in the optimizer groups,
    state re-keyed (scene/gaussian_model.py:300-316, :337-357).  `stats`: "zero" (densification_postfix) or
    "gather" (prune_points keeps the statistics of the surviving rows, :327-330)."""
    n_out = int(src_row.numel())
    opt = getattr(model, "optimizer", None)
    group_of = {g["name"]: g for g in opt.param_groups} if opt is not None else {}
    srcs, dsts, ovs, modes, placed = [], [], [], [], []
    empty = torch.empty(0, device=src_row.device)

    def add(key, tensor, mode, override=None):
        t = tensor.detach()
        t = t if t.is_contiguous() else t.contiguous()
        out = pool.out(key, n_out, t)
        # a tensor without columns (sh_degree = 0: `_features_rest` is (P, 0, 3)) or without source rows has nothing to move:
        # it only gets its correctly shaped (n_out, ...) view; the gather table holds the others
        if t.numel() > 0 and t.shape[0] > 0 and out[:1].numel() > 0:
            srcs.append(t); dsts.append(out); modes.append(mode); ovs.append(override if override is not None else empty)
        elif n_out and out[:1].numel() > 0:
            out[:n_out].zero_()
        return out

    for name, attr in GROUPS:
        p = getattr(model, attr)
        mode = _OVERRIDE_CHILD if name in overrides else _COPY
        out_p = add(attr, p, mode, overrides.get(name))
        out_m = out_v = None
        g = group_of.get(name)
        st = opt.state.get(g["params"][0], None) if g is not None else None
        if st is not None and "exp_avg" in st:
            out_m = add(attr + ".exp_avg", st["exp_avg"], _ZERO_NEW)
            out_v = add(attr + ".exp_avg_sq", st["exp_avg_sq"], _ZERO_NEW)
        placed.append((name, attr, p, out_p, st, out_m, out_v))
    stat_out = {}
    if stats == "gather":
        for attr in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(model, attr)
            if t.dtype != torch.float32:
                raise TypeError(f"{attr}: float32 expected, got {t.dtype}")
            stat_out[attr] = (add(attr, t.reshape(t.shape[0], -1), _COPY), t.shape[1:])
    if n_out and srcs:
        _C.densify_gather(src_row, kind, override_row, srcs, dsts, ovs, modes)

    for name, attr, p, out_p, st, out_m, out_v in placed:
        new_p = nn.Parameter(out_p[:n_out].requires_grad_(True))
        g = group_of.get(name)
        if g is not None:
            if st is not None:
                if out_m is not None:
                    st["exp_avg"], st["exp_avg_sq"] = out_m[:n_out], out_v[:n_out]
                del opt.state[g["params"][0]]
                g["params"][0] = new_p
                opt.state[new_p] = st
            else:
                g["params"][0] = new_p
        setattr(model, attr, new_p)
    dev = src_row.device
    if stats == "gather":
        for attr, (buf, tail) in stat_out.items():
            setattr(model, attr, buf[:n_out].reshape((n_out,) + tuple(tail)))
    else:
        for attr, tail in (("xyz_gradient_accum", (1,)), ("denom", (1,)), ("max_radii2D", ())):
            cur = getattr(model, attr, None)
            ok = isinstance(cur, torch.Tensor) and cur.dtype == torch.float32 and cur.device == dev and cur.dim() >= 1
            like = cur.reshape(cur.shape[0], 1) if ok else torch.empty((0, 1), device=dev)
            setattr(model, attr, pool.out(attr, n_out, like)[:n_out].zero_().reshape((n_out,) + tail))


@torch.no_grad()
def prune_points(model, mask: torch.Tensor, pool: Optional[RowPool] = None) -> None:
    """`GaussianModel.prune_points(mask)` (scene/gaussian_model.py:316-331): rows where `mask` is True are removed
    from every parameter, its Adam moments and the densification statistics."""
    pool = _pool_of(model, pool)
    rows = (~mask.reshape(-1)).nonzero(as_tuple=True)[0]
    src_row = rows.to(torch.int32)
    kind = torch.zeros(rows.numel(), dtype=torch.uint8, device=rows.device)
    _apply_plan(model, pool, src_row, kind, torch.zeros_like(src_row), {}, stats="gather")


@torch.no_grad()
def densify_and_prune(model, max_grad: float, min_opacity: float, extent: float, max_screen_size, N: int = 2,
                      pool: Optional[RowPool] = None, normal=torch.normal) -> Dict[str, int]:
    """`GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)` (:415-431).  Returns the
    row counts of the plan (`cloned`, `split`, `pruned`, `points`).  `normal(mean=, std=)` draws the split samples
    (tests substitute a recorded draw)."""
    pool = _pool_of(model, pool)
    dev = model._xyz.device
    P = model._xyz.shape[0]
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    scal = torch.exp(model._scaling)                       # get_scaling (:94-96)
    smax = torch.max(scal, dim=1).values
    dense = model.percent_dense * extent

    # densify_and_clone: appended copies, rows P .. P+Nc-1 of the grown set
    clone = torch.where(torch.norm(grads, dim=-1) >= max_grad, True, False) & (smax <= dense)
    clone_src = clone.nonzero(as_tuple=True)[0]
    src2 = torch.cat((torch.arange(P, device=dev), clone_src))          # source row of every row of the grown set
    # densify_and_split over the grown set: the clones' padded gradient is zero (:381-382)
    padded = torch.zeros(src2.numel(), device=dev)
    padded[:P] = grads.reshape(-1)
    split = torch.where(padded >= max_grad, True, False) & (smax[src2] > dense)
    sel = src2[split.nonzero(as_tuple=True)[0]]                         # source rows being split, in order
    stds = scal[sel].repeat(N, 1)
    means = torch.zeros((stds.size(0), 3), device=dev)
    samples = normal(mean=means, std=stds)                              # the reference's call, the reference's stream
    rots = _unit_rotation_matrices(model._rotation[sel]).repeat(N, 1, 1)
    # R s as three multiply-adds per row (the reference calls torch.bmm here: the same numbers up to the summation
    # order of a 3-term dot product, but a batched-GEMM launch over ~10^6 3x3 matrices takes 13 ms, this takes 0.1)
    child_xyz = (rots * samples[:, None, :]).sum(-1) + model._xyz[sel].repeat(N, 1)
    child_scaling = torch.log(scal[sel].repeat(N, 1) / (0.8 * N))       # scaling_inverse_activation = log
    child_src = sel.repeat(N)

    # final prune (:421-427) over [grown set without the split rows] ++ [children]
    low = (torch.sigmoid(model._opacity) < min_opacity).reshape(-1)
    drop_a, drop_c = low[src2], low[child_src]
    if max_screen_size:
        screen = bool(0.0 > max_screen_size)      # max_radii2D of every row is zero at this point (:376)
        drop_a = drop_a | (smax[src2] > 0.1 * extent) | screen
        drop_c = drop_c | (torch.exp(child_scaling).max(dim=1).values > 0.1 * extent) | screen
    rows_a = (~split & ~drop_a).nonzero(as_tuple=True)[0]
    rows_c = (~drop_c).nonzero(as_tuple=True)[0]

    src_row = torch.cat((src2[rows_a], child_src[rows_c])).to(torch.int32)
    kind = torch.cat(((rows_a >= P).to(torch.uint8), torch.full((rows_c.numel(),), 2, dtype=torch.uint8, device=dev)))
    override_row = torch.cat((torch.zeros_like(rows_a), rows_c)).to(torch.int32)
    _apply_plan(model, pool, src_row, kind, override_row, {"xyz": child_xyz.contiguous(), "scaling": child_scaling.contiguous()},
                stats="zero")
    n_out = int(src_row.numel())
    return {"cloned": int(clone_src.numel()), "split": int(sel.numel()),
            "pruned": int(src2.numel() - sel.numel() + child_src.numel() - n_out), "points": n_out}


@torch.no_grad()
def compact(model) -> None:
    """After a densification every parameter, Adam moment and statistic is a VIEW `buffer[:n]` of a capacity-sized RowPool
    buffer (that is what keeps the shapes stable).  Two consequences the reference's fresh exact-size tensors do not have:
    `torch.save(model.capture())` / `optimizer.state_dict()` serialise the whole underlying storage (~1.5x, with stale rows),
    and a tensor held across TWO densifications is overwritten when its buffer is reused.  `compact(model)` replaces every
    one of them by an exact-size clone (same values, optimizer state re-keyed) - call it before capturing a checkpoint, or
    before handing tensors to code that keeps them."""
    opt = getattr(model, "optimizer", None)
    group_of = {g.get("name"): g for g in opt.param_groups} if opt is not None else {}
    if opt is not None:
        # every parameter must be found in the optimizer BEFORE anything is replaced: with a missing or renamed group the
        # optimizer would keep stepping the old pool view while the model holds the new tensor - training of that parameter
        # would silently stop
        missing = [name for name, _attr in GROUPS if name not in group_of]
        if missing:
            raise KeyError(f"compact(): the optimizer has no param_group named {missing} (its groups: {sorted(map(str, group_of))})")
    for name, attr in GROUPS:
        p = getattr(model, attr)
        new_p = nn.Parameter(p.detach().clone().requires_grad_(True))
        if p.grad is not None:          # called between backward() and step(): the gradient moves along
            new_p.grad = p.grad.detach().clone()
        g = group_of.get(name)
        if g is not None:
            st = opt.state.pop(g["params"][0], None)
            g["params"][0] = new_p
            if st is not None:
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st:
                        st[k] = st[k].clone()
                opt.state[new_p] = st
        setattr(model, attr, new_p)
    for attr in ("xyz_gradient_accum", "denom", "max_radii2D"):
        t = getattr(model, attr, None)
        if isinstance(t, torch.Tensor):
            setattr(model, attr, t.clone())
