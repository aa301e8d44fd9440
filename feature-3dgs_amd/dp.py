"""View-sharded data parallelism for the rasterizer: one process per GPU, one view per rank and step,
RCCL reduction of the per-Gaussian gradients over xGMI.

The reference trains on ONE view per iteration on one GPU and contains no torch.distributed call on
this path (reference train.py:73-149), so this is new functionality layered ABOVE
the drop-in op: every rank holds all P Gaussians (2M x (59+256) floats = 2.5 GB at config c4, trivial in
288 GB), renders its own view, and the gradients of the leaf inputs of the op - means3D 3, SH 48,
semantic feature C, opacity 1, scales 3, rotations 4 = (59 + C) floats per Gaussian - are summed.

xGMI on MI355X is a point-to-point mesh (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by ONE
link, so the gradients travel as a few LARGE collectives (SH and feature gradients in place, the four
small tensors in one flat bucket), which lets RCCL use its direct algorithms across all links.

Four ways to exchange, all IN PLACE on the tensors handed in (so `leaf.grad` is what the optimiser reads):

  * `all_reduce_gaussian_grads(grads)`            after the backward pass, everything at once;
  * `FeatureGradOverlap`                          starts the all-reduce of the (largest) feature gradient as
                                                  soon as the blend backward has produced it, on a side
                                                  stream, while the op's last stage (preprocess backward)
                                                  still runs;
  * `RowsGradOverlap`                             that last stage runs in row chunks; the SH gradient (48 of the
                                                  59 remaining floats) is reduced chunk by chunk behind it;
  * `reduce_scatter_gaussian_grads(grads, ...)`   for a sharded optimiser: every rank receives only the sum
                                                  of ITS slice of Gaussians - half the bytes of an all-reduce;
                                                  `all_gather_params` redistributes the updated parameters.

Densification statistics need two more tiny reductions: SUM of the screen-space gradient norms / visibility
counts and MAX of the radii (reference train.py:132-133, scene/gaussian_model.py:436-438).

Backends: "nccl" (= RCCL on ROCm) on GPUs; "gloo" for the CPU tests.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

# the leaf inputs of the op under the names of `GaussianRasterizer.forward` (reference __init__.py:204)
GRAD_KEYS = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")


def _active(group) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def views_for_rank(num_views: int, rank: int, world: int, iteration: int = 0, views_per_iter: Optional[int] = None) -> List[int]:
    """Deterministic sharding: an iteration covers `views_per_iter` consecutive views (default: one per rank), rank r
    renders its contiguous share of them - every rank different views of the same step, every view visited equally
    often.  `views_per_iter` must be a multiple of the world size (every rank renders the same number: the collective
    schedule and the step time are the same everywhere)."""
    if num_views <= 0:
        return []
    V = world if views_per_iter is None else int(views_per_iter)
    if V <= 0 or V % world != 0:
        raise ValueError(f"views_per_iter = {V} is not a positive multiple of the world size {world}")
    per = V // world
    return [(V * iteration + rank * per + j) % num_views for j in range(per)]


class GradBuckets:
    """Packs a fixed set of gradient tensors into flat fp32 buckets and all-reduces them."""

    def __init__(self, shapes: Dict[str, Sequence[int]], device, bucket_bytes: int = 256 << 20):
        self.layout = []  # (key, bucket, offset, numel, shape)
        self.buckets: List[torch.Tensor] = []
        cap = max(1, bucket_bytes // 4)
        fill = 0
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
        """Per-key views INTO the buckets."""
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

    def unpack_into(self, grads: Dict[str, torch.Tensor]) -> None:
        """Copy the reduced values back into the caller's tensors (in place)."""
        for k, b, o, n, shp in self.layout:
            grads[k].copy_(self.buckets[b][o:o + n].view(shp))


_DEBUG = os.environ.get("F3DGS_DP_DEBUG", "0") not in ("", "0")     # cross-rank agreement checks (extra collectives)
# How a collective-sized gradient tensor is summed: "allreduce" (the library's all-reduce, whatever algorithm it picks) or "direct"
# (all-to-all of the slices + a local sum + all-gather, see all_reduce_direct).  `bench.py --comm-only` times both on the node.
EXCHANGE = os.environ.get("F3DGS_DP_EXCHANGE", "allreduce")


def all_reduce_direct(t: torch.Tensor, group=None, async_op: bool = False):
    """SUM `t` over all ranks in place WITHOUT a ring: every rank sends slice j of its tensor straight to rank j (one
    all-to-all: N - 1 point-to-point transfers per rank, one per peer), sums the N slices it received, and the summed slices are
    all-gathered back.  On a full mesh of point-to-point links (MI355X: seven xGMI links of ~153 GB/s per GPU) every transfer has
    a link of its own, so the exchange costs 2 (S / N) / link - 4.1 ms for c4's 2.52 GB on eight GPUs - where a ring all-reduce
    is bound by ONE link: 2 (N - 1) / N x S / link = 29 ms.  The library's all-reduce may or may not pick a
    direct algorithm; this one is direct by construction, out of collectives every backend has (tested on gloo at world size 2:
    the sum of two terms is order-free, so the result equals all_reduce bit for bit).  Costs one extra buffer of the tensor's size.
    Returns a handle with .wait() when `async_op` (the work runs on the collective stream of the backend as usual)."""
    world = dist.get_world_size(group)
    flat = t.reshape(-1)
    n = flat.numel()
    per = (n + world - 1) // world
    if per * world != n:
        send = torch.zeros(per * world, dtype=t.dtype, device=t.device)
        send[:n].copy_(flat)
    else:
        send = flat.contiguous()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)                    # recv[j * per : (j + 1) * per] = rank j's slice for me
    mine = recv.view(world, per).sum(dim=0)
    work = dist.all_gather_into_tensor(send, mine, group=group, async_op=async_op)

    def finish():
        if send.data_ptr() != flat.data_ptr() or not t.is_contiguous():
            t.copy_(send[:n].view(t.shape))

    class _Handle:
        def wait(self_inner):
            if work is not None:
                work.wait()
            finish()
    if async_op:
        return _Handle()
    finish()
    return None


def _sum_tensor(t: torch.Tensor, group=None, async_op: bool = False):
    """One collective-sized tensor, summed in place by the configured exchange."""
    if EXCHANGE == "direct":
        return all_reduce_direct(t, group=group, async_op=async_op)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _check(grads: Dict[str, torch.Tensor]) -> List[str]:
    keys = []
    for k, v in grads.items():
        if v is None:
            continue
        if not isinstance(v, torch.Tensor):
            raise TypeError(f"gradient '{k}' is a {type(v).__name__}, expected a tensor")
        if v.dtype != torch.float32:
            raise TypeError(f"gradient '{k}' is {v.dtype}; the op produces float32 gradients")
        keys.append(k)
    return keys


def all_reduce_gaussian_grads(grads: Dict[str, torch.Tensor], group=None, bucket_bytes: int = 256 << 20,
                              buckets: Optional[GradBuckets] = None, direct_bytes: int = 32 << 20,
                              skip: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """SUM every tensor of `grads` over all ranks, IN PLACE, and return the same dict.

    Any key is accepted (the reference's leaves are called `_xyz`, `_features_dc`, ...: pass
    `{name: p.grad for name, p in model.named_parameters()}` or the op's own input names alike); `None`
    values are ignored.  Tensors of at least `direct_bytes` (SH and feature gradients: 192 MB / 4C MB per
    million Gaussians) are already collective-sized and are reduced directly; the small ones (means, opacity,
    scales, rotations) share one flat bucket so that they cost one collective instead of four, and the reduced
    values are copied back into the caller's tensors.  `skip`: keys that were already reduced elsewhere
    (FeatureGradOverlap)."""
    if not _active(group):
        return grads
    skip = set(skip)
    keys = [k for k in _check(grads) if k not in skip]
    # which tensors go alone and which share the bucket is decided by SHAPE only: every rank must issue the same collectives
    # in the same order (a rank-local property such as contiguity must not change the schedule)
    big = [k for k in keys if grads[k].numel() * 4 >= direct_bytes]
    small = [k for k in keys if k not in big]
    staged = {k: grads[k].contiguous() for k in big if not grads[k].is_contiguous()}
    works = [_sum_tensor(staged.get(k, grads[k]), group=group, async_op=True) for k in big]
    if small:
        if buckets is None:
            buckets = GradBuckets({k: grads[k].shape for k in small}, grads[small[0]].device, bucket_bytes)
        buckets.pack({k: grads[k] for k in small})
        works += buckets.all_reduce(group, async_op=True)
    for w in works:
        w.wait()
    for k, t in staged.items():
        grads[k].copy_(t)
    if small:
        buckets.unpack_into(grads)
    return grads


class FeatureGradOverlap:
    """Overlaps the all-reduce of dL/dsemantic_feature with the tail of the op's backward pass.

    The feature gradient (C of the 59 + C floats per Gaussian: 2 GB of the 2.5 GB at config c4) is complete
    when the blend backward kernel has run; the op then still executes its preprocess backward.  The op calls
    the hook installed here at that point (`diff_gaussian_rasterization.set_feature_grad_hook`): it records an
    event on the op's stream, lets a side stream wait for it and starts the RCCL all-reduce there.
    The current stream is made to wait for the collective before the extension call returns to autograd, so
    whatever autograd does with the tensor afterwards (keep it as `leaf.grad` or copy it) sees the SUM.  Then reduce
    the remaining gradients, skipping the feature leaf:

        with FeatureGradOverlap(group) as ov:
            loss.backward()
            all_reduce_gaussian_grads(grads, skip=ov.reduced(("semantic_feature",)))
    """

    def __init__(self, group=None):
        self.group = group
        self._work = None
        self._tensor: Optional[torch.Tensor] = None
        self._stream = None
        self._installed = False
        self._count = 0

    def _hook(self, feature_grad: torch.Tensor) -> None:
        if not _active(self.group) or feature_grad.numel() == 0:
            return
        self._tensor = feature_grad
        if feature_grad.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=feature_grad.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(feature_grad.device))
            self._stream.wait_event(ready)
            with torch.cuda.stream(self._stream):
                self._work = _sum_tensor(feature_grad, group=self.group, async_op=True)
        else:
            self._work = _sum_tensor(feature_grad, group=self.group, async_op=True)
        self._count += 1

    def __enter__(self):
        import diff_gaussian_rasterization as dgr
        dgr.set_feature_grad_hook(self._hook, self.finish)
        self._installed = True
        return self

    def __exit__(self, *exc):
        import diff_gaussian_rasterization as dgr
        if self._installed:
            dgr.set_feature_grad_hook(None)
            self._installed = False
        return False

    def finish(self) -> None:
        """Join (called by the op when its backward call has returned): work queued on the current stream from here
        on sees the reduced tensor.  No host synchronisation."""
        if self._work is not None:
            self._work.wait()          # NCCL/RCCL: the current stream waits for the collective
            if self._stream is not None:
                torch.cuda.current_stream(self._tensor.device).wait_stream(self._stream)
            self._work = None
            self._tensor = None

    def reduced(self, feature_keys: Iterable[str]) -> List[str]:
        """`feature_keys` if a feature gradient was reduced by this object since it was entered, else nothing."""
        return list(feature_keys) if self._count > 0 else []


class RowsGradOverlap:
    """Overlaps the all-reduce of per-Gaussian op gradients - by default the SH gradient, 48 of the 59 non-feature
    floats per Gaussian - with the per-Gaussian stage of the op's backward pass.

    With the hook installed (`diff_gaussian_rasterization.set_grad_rows_hook`) that stage runs as `chunks` launches over
    consecutive row ranges; after each the op calls `_hook(row_begin, row_end, grads)`: an event is recorded on the op's
    stream, a side stream waits for it and starts the RCCL all-reduce of rows [row_begin, row_end) of every tensor named in
    `names` (row slices of the contiguous (P, ...) gradients: no staging copy).  The chunk boundaries depend on P and
    `chunks` only, so every rank issues the same collectives in the same order.  `finish()` (called by the op when its
    backward call has returned) makes the current stream wait for them, so autograd - which still has to carry the op-level
    gradient to the leaves (the SH gradient through the `cat` of features_dc / features_rest) - sees the SUM; those steps
    are linear in the gradient, the leaves therefore end up with the sum over ranks and are skipped by the final reduce:

        with RowsGradOverlap(group) as rv:
            loss.backward()
            all_reduce_gaussian_grads(grads, skip=rv.reduced({"sh": ("_features_dc", "_features_rest")}))

    Every rank's backward pass must reach the op (a rank without a view must not use this)."""

    def __init__(self, group=None, names: Sequence[str] = ("sh",), chunks: int = 4):
        self.group = group
        self.names = tuple(names)
        self.chunks = int(chunks)
        self._works = []
        self._stream = None
        self._device = None
        self._installed = False
        self._count = 0                      # chunks actually reduced
        self.fired = 0                       # hook invocations (the op was reached), reduced or not
        self.rows: List[tuple] = []          # (row_begin, row_end) of every call since entered

    def _hook(self, row_begin: int, row_end: int, grads: Dict[str, torch.Tensor]) -> None:
        self.rows.append((int(row_begin), int(row_end)))
        self.fired += 1
        if not _active(self.group):
            return
        parts = [grads[n][row_begin:row_end] for n in self.names if grads[n].numel() > 0]
        if not parts:
            return
        if parts[0].is_cuda:
            self._device = parts[0].device
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=self._device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self._device))
            self._stream.wait_event(ready)
            with torch.cuda.stream(self._stream):
                for t in parts:
                    self._works.append(_sum_tensor(t, group=self.group, async_op=True))
        else:
            for t in parts:
                self._works.append(_sum_tensor(t, group=self.group, async_op=True))
        self._count += 1

    def __enter__(self):
        import diff_gaussian_rasterization as dgr
        dgr.set_grad_rows_hook(self._hook, self.chunks, self.finish)
        self._installed = True
        return self

    def __exit__(self, *exc):
        import diff_gaussian_rasterization as dgr
        if self._installed:
            dgr.set_grad_rows_hook(None)
            self._installed = False
        return False

    def finish(self) -> None:
        """Join: work queued on the current stream from here on sees the reduced rows.  No host synchronisation."""
        for w in self._works:
            w.wait()
        if self._works and self._stream is not None:
            torch.cuda.current_stream(self._device).wait_stream(self._stream)
        self._works = []

    def reduced(self, leaves_of: Dict[str, Sequence[str]]) -> List[str]:
        """The leaf keys made final by this object since it was entered: `leaves_of[name]` for every reduced op gradient.
        Empty when the hook fired but the named gradients were empty (the op was fed precomputed colours: dL_dsh has zero
        width, a property of the call that is the same on every rank) - those leaves then belong to the final all-reduce."""
        if self._count == 0:
            return []
        return [leaf for n in self.names for leaf in leaves_of.get(n, ())]


def shard_range(P: int, rank: int, world: int):
    """Gaussians [lo, hi) owned by `rank` under the contiguous sharding used by the sharded-optimiser path."""
    per = (P + world - 1) // world
    return min(P, rank * per), min(P, (rank + 1) * per)


def reduce_scatter_gaussian_grads(grads: Dict[str, torch.Tensor], group=None) -> Dict[str, torch.Tensor]:
    """Sharded-optimiser exchange: returns, per key, the SUM over ranks of the rows [lo, hi) this rank owns
    (`shard_range`).  Moves (N-1)/N of the gradient bytes once instead of twice (all-reduce = reduce-scatter +
    all-gather): the second half is replaced by `all_gather_params` of the UPDATED parameters, which a training
    loop needs anyway after a sharded step and can overlap with the next view's preprocess.  With world size 1
    it returns the full tensors."""
    keys = _check(grads)
    if not _active(group):
        return {k: grads[k] for k in keys}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out, works = {}, []
    for k in keys:
        g = grads[k]
        P = g.shape[0]
        per = (P + world - 1) // world
        flat = g.reshape(P, -1)
        if per * world != P:   # pad the row dimension to a multiple of the world size
            pad = torch.zeros(per * world - P, flat.shape[1], dtype=g.dtype, device=g.device)
            flat = torch.cat([flat, pad], 0)
        flat = flat.contiguous()
        mine = torch.empty(per, flat.shape[1], dtype=g.dtype, device=g.device)
        works.append(dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        lo, hi = shard_range(P, rank, world)
        out[k] = (mine, hi - lo, g.shape[1:])
    for w in works:
        w.wait()
    return {k: m[:n].reshape((n,) + tuple(shp)) for k, (m, n, shp) in out.items()}


def all_gather_params(shards: Dict[str, torch.Tensor], full: Dict[str, torch.Tensor], group=None) -> None:
    """Inverse of the sharding above: every rank contributes its updated rows, `full[k]` receives all P rows."""
    if not _active(group):
        for k, s in shards.items():
            full[k].copy_(s)
        return
    world = dist.get_world_size(group)
    for k, s in shards.items():
        P = full[k].shape[0]
        per = (P + world - 1) // world
        row = full[k].reshape(P, -1).shape[1]
        send = torch.zeros(per, row, dtype=s.dtype, device=s.device)
        send[:s.shape[0]].copy_(s.reshape(s.shape[0], -1))
        recv = torch.empty(per * world, row, dtype=s.dtype, device=s.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        full[k].copy_(recv[:P].reshape(full[k].shape))


class ShardedOptimizer:
    """The byte-halving exchange made usable end to end: every rank owns the contiguous row range `shard_range(P, rank,
    world)` of every per-Gaussian parameter and keeps optimizer state (Adam's two moments: 2/3 of the training state) for
    those rows only.  One step =

        reduce-scatter of the ranks' gradients  ->  the summed gradient of MY rows        ((N-1)/N of the bytes, once)
        the optimizer's step on my rows         ->  any elementwise optimizer (`make_optimizer`: FusedAdam on HIP
                                                    tensors, torch.optim.Adam in the CPU tests)
        all-gather of the updated rows          ->  every rank holds all P updated rows   ((N-1)/N of the PARAMETER bytes)

    against all-reduce (2 (N-1)/N of the gradient bytes) + N identical full optimizer steps.  The gradient message is
    (59 + C) floats per Gaussian, the parameter message the same: the bytes on the wire are equal, but the all-gather
    carries parameters - it has no dependence on the NEXT step's backward pass and can run under the next forward's
    preprocess (`step(..., gather=False)` + `gather()`), and the optimizer's work and state shrink by N.
    An elementwise optimizer stepped on a row range gives those rows bit for bit what the full step gives them, so at
    world size 2 (a sum of two terms is order-free) the parameters equal the all-reduce path's EXACTLY
    (tests/test_dp_gloo.py::test_sharded_optimizer_equals_the_all_reduce_path).

    `params`: name -> full (P, ...) tensor (updated in place by `step`).  `make_optimizer(shards)`: name -> my rows as a
    leaf tensor; returns the optimizer over them (the caller chooses groups / learning rates, as
    scene/gaussian_model.py:163-178 does for the full tensors).  Densification changes P: call `full_state()` before
    it (all moments gathered, keyed like torch.optim.Adam's state) and build a new ShardedOptimizer after it
    (`load_full_state`)."""

    def __init__(self, params: Dict[str, torch.Tensor], make_optimizer: Callable[[Dict[str, torch.Tensor]], "torch.optim.Optimizer"],
                 group=None):
        self.group = group
        self.params = params
        self.world = dist.get_world_size(group) if _active(group) else 1
        self.rank = dist.get_rank(group) if _active(group) else 0
        self.P = next(iter(params.values())).shape[0]
        for k, v in params.items():
            if v.shape[0] != self.P:
                raise ValueError(f"parameter '{k}' has {v.shape[0]} rows, expected {self.P}")
        self.lo, self.hi = shard_range(self.P, self.rank, self.world)
        self.shards = {k: v.detach()[self.lo:self.hi].clone().requires_grad_(True) for k, v in params.items()}
        self.optimizer = make_optimizer(self.shards)
        self._pending = False

    def step(self, grads: Dict[str, torch.Tensor], gather: bool = True) -> None:
        """`grads`: name -> this rank's LOCAL (unreduced) gradient of the full tensor (e.g. `dp_step(..., reduce=False)`)."""
        local = {k: grads[k] for k in self.shards}
        mine = reduce_scatter_gaussian_grads(local, group=self.group)
        for k, p in self.shards.items():
            g = mine[k]
            p.grad = g[self.lo:self.hi] if self.world == 1 else g      # (world 1: the helper returns the full tensors)
        self.optimizer.step()
        self._pending = True
        if gather:
            self.gather()

    @torch.no_grad()
    def gather(self) -> None:
        """The updated rows of every rank -> the full parameters, in place."""
        if not self._pending:
            return
        if self.world == 1:
            for k, p in self.shards.items():
                self.params[k].data[self.lo:self.hi].copy_(p.detach())
        else:
            all_gather_params({k: p.detach() for k, p in self.shards.items()}, {k: self.params[k].data for k in self.shards}, group=self.group)
        self._pending = False

    @torch.no_grad()
    def refresh_from_params(self, names: Optional[Iterable[str]] = None) -> None:
        """The shards are CLONES of my rows: an in-place edit of the full parameters from outside the optimizer - the reference's
        `reset_opacity` / `replace_tensor_to_optimizer` (scene/gaussian_model.py:185-190, 284-297), a manual clamp - would be
        overwritten by the next `gather()`.  Call this after such an edit (on every rank): my rows of `params[name]` are copied
        into the shards again.  The optimizer state of the edited tensors is the caller's business, as it is in the reference
        (which zeroes the moments of a replaced tensor): `self.optimizer.state[self.shards[name]]`.  Row counts must not have
        changed - densification needs a new ShardedOptimizer (`full_state` / `load_full_state`)."""
        for k in (list(names) if names is not None else list(self.shards)):
            if self.params[k].shape[0] != self.P:
                raise ValueError(f"parameter '{k}' now has {self.params[k].shape[0]} rows, the sharding was built for {self.P}")
            self.shards[k].data.copy_(self.params[k].detach()[self.lo:self.hi])

    @torch.no_grad()
    def full_state(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """name -> {state key -> full (P, ...) tensor} (per-row state tensors gathered from all ranks; scalars such as Adam's
        `step` as they are): what an unsharded optimizer would hold, e.g. for the reference's densification code."""
        out = {}
        for k, p in self.shards.items():
            st = self.optimizer.state.get(p, {})
            full = {}
            # per-row state is what has the SHARD's shape (Adam: exp_avg, exp_avg_sq; `step` is a scalar); keys in sorted order:
            # every gather below is a collective and all ranks must issue the same sequence
            for name in sorted(st):
                val = st[name]
                if isinstance(val, torch.Tensor) and val.dim() >= 1 and tuple(val.shape) == tuple(p.shape):
                    dst = torch.zeros((self.P,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
                    all_gather_params({"x": val}, {"x": dst}, group=self.group) if self.world > 1 else dst[self.lo:self.hi].copy_(val)
                    full[name] = dst
                else:
                    full[name] = val.clone() if isinstance(val, torch.Tensor) else val
            out[k] = full
        return out

    @torch.no_grad()
    def load_full_state(self, state: Dict[str, Dict[str, torch.Tensor]]) -> None:
        """Inverse of `full_state` (after a densification built a new ShardedOptimizer over the new P rows)."""
        for k, p in self.shards.items():
            if k not in state:
                continue
            st = self.optimizer.state[p]
            for name, val in state[k].items():
                if isinstance(val, torch.Tensor) and val.dim() >= 1 and tuple(val.shape) == (self.P,) + tuple(p.shape[1:]):
                    st[name] = val[self.lo:self.hi].clone()
                else:
                    st[name] = val.clone() if isinstance(val, torch.Tensor) else val


def reduce_densification_stats(grad_norm_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                               group=None) -> None:
    """In-place: SUM the accumulated view-space gradient norms and visibility counts, MAX the radii."""
    if not _active(group):
        return
    w1 = dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w2 = dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w3 = dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group, async_op=True)
    for w in (w1, w2, w3):
        w.wait()


# ---- one view split over the ranks by tile rows ("alternative for single huge views") -------------------------
# View sharding needs as many views per step as GPUs.  Where there are fewer - a 4K evaluation render, the last views of an
# epoch - ONE view can be split instead: rank r lists and blends only tile rows band_rows(r) of the 16 x 16 grid
# (`_C.set_tile_band`, include/f3dgs.h), the bands are gathered into the whole image for the loss, every rank back-propagates
# the upstream gradient through ITS band, and the per-Gaussian gradients are summed by the same exchange a view-sharded step
# ends with.  Inside a band the images are bit-identical to the whole view's; the per-Gaussian stages (projection, depth sort)
# are repeated on every rank - they are P-proportional and small beside the blend (c5: 0.53 of 9.5 ms).

def band_rows(height: int, rank: int, world: int, tile: int = 16):
    """(tile_row_begin, tile_row_end, pixel_row_begin, pixel_row_end) of rank `rank`'s band: the tile rows of the image split
    into `world` consecutive bands of (nearly) equal height.  Ranks beyond the number of tile rows get an empty band."""
    gy = (height + tile - 1) // tile
    per, extra = divmod(gy, world)
    r0 = rank * per + min(rank, extra)
    r1 = r0 + per + (1 if rank < extra else 0)
    return r0, r1, min(height, r0 * tile), min(height, r1 * tile)


def gather_bands(image: torch.Tensor, group=None, tile: int = 16) -> torch.Tensor:
    """`image` (..., H, W): this rank's render of its band (anything outside the band is ignored).  Returns the whole image
    with every band taken from the rank that rendered it; differentiable - the gradient of the result flows back into this
    rank's band rows only (the other ranks' rows are constants here and are back-propagated there)."""
    if not _active(group):
        return image
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H = image.shape[-2]
    spans = [band_rows(H, r, world, tile)[2:] for r in range(world)]
    hmax = max(y1 - y0 for y0, y1 in spans)
    y0, y1 = spans[rank]
    slab = image.new_zeros(image.shape[:-2] + (hmax, image.shape[-1]))
    slab[..., : y1 - y0, :] = image[..., y0:y1, :].detach()
    slabs = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(slabs, slab, group=group)
    parts = []
    for r, (a, b) in enumerate(spans):
        parts.append(image[..., a:b, :] if r == rank else slabs[r][..., : b - a, :])
    return torch.cat(parts, dim=-2)


def dp_step(render_and_backward: Callable[[int], None], leaves: Dict[str, torch.Tensor], view_ids: Iterable[int],
            group=None, buckets: Optional[GradBuckets] = None, overlap: bool = False,
            feature_key: str = "semantic_feature", rows_leaves: Optional[Dict[str, Sequence[str]]] = None,
            rows_chunks: int = 4, reduce: bool = True) -> Dict[str, torch.Tensor]:
    """One data-parallel step: `render_and_backward(view_id)` must run the op forward+backward for that view
    and accumulate into `leaves[k].grad`; afterwards `leaves[k].grad` holds the sum over all ranks' views for
    EVERY leaf (any names).  `overlap=True` starts the all-reduce of `leaves[feature_key].grad` inside the backward
    pass (one view per rank and step; the leaf must be fed to the op directly; needs the HIP extension).
    `rows_leaves` (with overlap): op gradient name -> the leaves that receive nothing but that gradient, e.g.
    {"sh": ("_features_dc", "_features_rest")} for the reference's model; those op gradients are reduced in `rows_chunks`
    row ranges inside the per-Gaussian stage of the backward pass (RowsGradOverlap) and their leaves are not reduced again.
    `reduce=False`: no exchange - the leaves keep this rank's LOCAL sums (zeros for a leaf its views did not reach, so that
    every rank holds the same key list): the input of `ShardedOptimizer.step`, whose reduce-scatter is the exchange."""
    for v in leaves.values():
        v.grad = None
    view_ids = list(view_ids)

    def all_grads():
        # the SAME key list on every rank: a leaf this rank's view did not reach (no gradient) contributes zeros - the
        # collective schedule is derived from the leaves, never from which gradients happen to exist locally
        if _active(group):
            for v in leaves.values():
                if v.grad is None:
                    v.grad = torch.zeros_like(v)
        return {k: v.grad for k, v in leaves.items() if v.grad is not None}

    if not reduce:
        for vid in view_ids:
            render_and_backward(vid)
        return all_grads()
    if overlap and len(view_ids) == 1 and _active(group):
        import contextlib
        rows_leaves = {n: tuple(k for k in ks if k in leaves) for n, ks in (rows_leaves or {}).items()}
        rows_leaves = {n: ks for n, ks in rows_leaves.items() if ks}
        rv_cm = RowsGradOverlap(group, tuple(rows_leaves), rows_chunks) if rows_leaves else contextlib.nullcontext()
        with FeatureGradOverlap(group) as ov, rv_cm as rv:
            render_and_backward(view_ids[0])
            grads = all_grads()
            rows_done: List[str] = []
            if rows_leaves:
                rows_done = rv.reduced(rows_leaves)
                # hook fired, nothing to reduce (zero-width SH gradient: colours were precomputed - the same on every rank):
                # rows_done stays empty and the leaves go through the final all-reduce.  Only a hook that never fired means this
                # rank did not reach the op while its peers wait in the chunk all-reduces.
                if rv.fired == 0:
                    raise RuntimeError("dp_step: rows overlap was requested but this rank's backward pass never reached the op "
                                       "(its peers are waiting in the chunk all-reduces)")
            feat = leaves.get(feature_key)
            if feat is None or feat.numel() == 0:
                return all_reduce_gaussian_grads(grads, group=group, buckets=buckets, skip=rows_done)
            # exactly ONE all-reduce of the feature gradient per rank and step: started inside the backward pass where the
            # hook fired; a rank whose backward pass did not reach the op (or produced no feature gradient) issues the
            # matching one here, so that its peers' collective is not left without a partner
            if not ov.reduced((feature_key,)):
                _sum_tensor(grads[feature_key], group=group)
            if _DEBUG:
                fired = torch.tensor([float(bool(ov.reduced((feature_key,))))], device=grads[feature_key].device)
                lo = fired.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
                dist.all_reduce(fired, op=dist.ReduceOp.MAX, group=group)
                if float(lo) != float(fired):
                    raise RuntimeError("dp_step: the feature-gradient hook fired on some ranks only (F3DGS_DP_DEBUG)")
            return all_reduce_gaussian_grads(grads, group=group, buckets=buckets, skip=[feature_key] + rows_done)
    for vid in view_ids:
        render_and_backward(vid)
    return all_reduce_gaussian_grads(all_grads(), group=group, buckets=buckets)


_stream_pool: Dict[tuple, list] = {}


def _side_streams(device, n: int) -> list:
    key = (device.type, device.index)
    pool = _stream_pool.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def dp_step_views(forward: Callable[[int], object], backward: Callable[[object], None], leaves: Dict[str, torch.Tensor],
                  view_ids: Iterable[int], group=None, buckets: Optional[GradBuckets] = None, overlap: bool = True,
                  feature_key: str = "semantic_feature", n_streams: int = 2, accumulate: Optional[bool] = None,
                  reduce: bool = True) -> Dict[str, torch.Tensor]:
    """One data-parallel step over SEVERAL views per rank (a step of `views_per_iter` views on fewer GPUs than views;
    SURVEY.md 8(e): config c4 is "8 views per iteration").  `forward(view_id)` runs the op forward and the loss of one view
    and returns a handle; `backward(handle)` runs its backward pass, accumulating into `leaves[k].grad`.  Afterwards
    `leaves[k].grad` holds the sum over all views of all ranks, for every leaf.  Against `dp_step` with the same views:

      * pipelining (HIP devices, `n_streams` >= 2): the views alternate between two side streams and view v + 1's forward
        is enqueued right behind view v's backward - its preprocess and binning (launch- and HBM-bound, one host round
        trip for the instance count) run under view v's blend kernels (issue-bound) instead of behind them.  Backward passes
        stay ordered among themselves (they add into the same gradients);
      * in-place accumulation (`accumulate`, default: on where the extension is present and the feature leaf is fed to the op
        directly): the feature gradient - C of the 59 + C floats per Gaussian - of every view is ADDED by the blend backward
        into the leaf's `.grad` itself (`set_feature_grad_accumulator`): one zero-fill per step instead of one per view, no
        per-view (P, C) tensor, no per-view add;
      * overlap: the all-reduce of that accumulated feature gradient starts inside the LAST view's backward pass, as soon as
        its blend backward is enqueued (`FeatureGradOverlap`), under that view's per-Gaussian stage; the remaining leaves
        follow in the bucketed all-reduce.  (The row-chunked SH overlap of `dp_step` needs the op-level gradient to BE the
        step's gradient, i.e. one view per rank.)

    The collective schedule depends on the leaves and the view COUNT only - every rank must render the same number of views.
    `reduce=False`: no exchange, the leaves keep this rank's local sums (see dp_step)."""
    for v in leaves.values():
        v.grad = None
    view_ids = list(view_ids)
    V = len(view_ids)
    if not reduce:
        overlap = False
    first = next(iter(leaves.values()))
    dev = first.device
    on_gpu = dev.type == "cuda"
    feat = leaves.get(feature_key)
    have_feat = feat is not None and feat.numel() > 0
    auto = accumulate is None
    if auto:
        accumulate = on_gpu and have_feat and V > 0
    dgr = None
    if accumulate or (overlap and on_gpu):
        import diff_gaussian_rasterization as dgr        # (the HIP extension; CPU stand-ins of the op run without it)
    if accumulate:
        if not (have_feat and feat.is_contiguous()):
            raise ValueError("accumulate=True needs a contiguous feature leaf")
        feat.grad = torch.zeros_like(feat)
        # (`leaf=feat`: every backward call checks that the op's feature input IS this leaf - a model that feeds the op a
        # transformed / masked / copied feature tensor gets an error instead of a gradient that skipped its autograd chain)
        # `accumulate=None` (not asked for, switched on because it is usually right): a backward call whose feature input turns
        # out NOT to be the leaf takes the autograd path instead of raising (strict=False) - autograd then adds into the same
        # zero-initialised feat.grad through the model's own chain
        dgr.set_feature_grad_accumulator(feat.grad, feat, strict=not auto)
    active = _active(group)
    ov = None
    # The leaves were created on the caller's stream and their gradients arrive from the side streams below: intended (the
    # backward passes are ordered among themselves with events, and the caller's stream waits for the last of them) - the
    # warning about it is silenced for the duration of this step only.
    # (process-wide switch: its previous state is read where torch exposes it and put back afterwards; another thread that
    # runs backward passes meanwhile sees the warning silenced for that long)
    quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    warn_was = getattr(torch._C, "_warn_on_accumulate_grad_stream_mismatch", None)
    warn_was = bool(warn_was()) if callable(warn_was) else True
    pipelined = on_gpu and V > 1 and n_streams >= 2
    if not pipelined:
        quiet = None
    # the in-backward all-reduce of the LAST view's feature gradient is the step's exchange only if that tensor is the step's
    # sum: one view per rank, or the running sum of the accumulator
    use_ov = overlap and active and have_feat and (accumulate or V == 1)
    try:
        if quiet is not None:
            quiet(False)
        if pipelined:
            main = torch.cuda.current_stream(dev)
            pool = _side_streams(dev, n_streams)
            for st in pool:
                st.wait_stream(main)                     # leaves, zeroed accumulator: ready on `main`
            with torch.cuda.stream(pool[0]):
                handle = forward(view_ids[0])
            prev_done = None
            for i in range(V):
                st = pool[i % n_streams]
                last = i == V - 1
                with torch.cuda.stream(st):
                    if prev_done is not None:
                        st.wait_event(prev_done)         # backward passes add into the same gradients: one after the other
                    if last and use_ov and dgr is not None:
                        ov = FeatureGradOverlap(group)
                        with ov:
                            backward(handle)
                    else:
                        backward(handle)
                    prev_done = torch.cuda.Event()
                    prev_done.record(st)
                if not last:
                    with torch.cuda.stream(pool[(i + 1) % n_streams]):
                        handle = forward(view_ids[i + 1])        # enqueued behind backward(i): runs under it
            for st in pool:
                main.wait_stream(st)
        else:
            for i, vid in enumerate(view_ids):
                handle = forward(vid)
                if i == V - 1 and use_ov and on_gpu and dgr is not None:
                    ov = FeatureGradOverlap(group)
                    with ov:
                        backward(handle)
                else:
                    backward(handle)
    finally:
        if quiet is not None:
            quiet(warn_was)
        if accumulate:
            dgr.set_feature_grad_accumulator(None)
    # the SAME key list on every rank (see dp_step)
    if active:
        for v in leaves.values():
            if v.grad is None:
                v.grad = torch.zeros_like(v)
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    if not active or not reduce:
        return grads
    skip: List[str] = []
    if have_feat and overlap and on_gpu:
        # exactly ONE all-reduce of the feature gradient per rank and step (dp_step): inside the last backward pass where the
        # hook fired, here otherwise
        if ov is not None and ov.reduced((feature_key,)):
            skip = [feature_key]
        elif accumulate:
            _sum_tensor(grads[feature_key], group=group)
            skip = [feature_key]
    return all_reduce_gaussian_grads(grads, group=group, buckets=buckets, skip=skip)
