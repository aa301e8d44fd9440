"""diff_gaussian_rasterization — MI355X-native drop-in for the Feature-3DGS rasterizer op.

Public surface is the reference's (reference: submodules/diff-gaussian-rasterization-feature/
diff_gaussian_rasterization/__init__.py:21-44 `rasterize_gaussians`, :46-172 `_RasterizeGaussians`,
:174-186 `GaussianRasterizationSettings`, :188-238 `GaussianRasterizer`), so the reference's
`gaussian_renderer/__init__.py` and `train.py` import and call it unchanged:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    color, feature_map, radii, depth = GaussianRasterizer(settings)(means3D=..., means2D=..., ...)

The compute lives in hand-written HIP kernels for gfx950 behind a C ABI (include/f3dgs.h,
feature-3dgs_amd/csrc/); `_C` is the libtorch/pybind11 binding over that ABI.  There is NO CPU
fallback: importing this package without the built extension raises, and so does handing it
non-HIP tensors.
"""
from __future__ import annotations

import itertools
import threading
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

try:
    from . import _C  # noqa: F401  (built in-tree by feature-3dgs_amd/build.py)
except ImportError as exc:  # fail loudly — a silent fallback would void every parity claim
    raise ImportError(
        "diff_gaussian_rasterization._C (the gfx950 HIP extension) is not built. "
        "Run `python feature-3dgs_amd/build.py` (or `__graft_entry__.build()`)."
    ) from exc


_backward_done_hook = None
_rows_done_hook = None
_feature_ready_hook = None      # what set_feature_grad_hook installed (kept so that a bypassing backward call can put it back)


def set_feature_grad_hook(on_ready, on_done=None) -> None:
    """Data-parallel overlap (not in the reference): `on_ready(dL_dsemantic_feature)` runs inside the backward
    pass as soon as that tensor is final on the current stream (the per-Gaussian stage still follows);
    `on_done()` runs when the extension call has returned, before autograd sees the gradients.  `None` removes
    both.  See feature-3dgs_amd/dp.py: FeatureGradOverlap."""
    global _backward_done_hook, _feature_ready_hook
    _C.set_feature_grad_hook(on_ready)
    _feature_ready_hook = on_ready
    _backward_done_hook = on_done if on_ready is not None else None


def set_grad_rows_hook(on_rows, chunks: int = 4, on_done=None) -> None:
    """Data-parallel overlap (not in the reference): the per-Gaussian stage of the backward pass runs in `chunks` row
    ranges and `on_rows(row_begin, row_end, grads)` is called after each has been enqueued - `grads` maps the op's
    input names ("sh", "means3D", "scales", "rotations", "opacities", "colors_precomp", "means2D",
    "cov3Ds_precomp") to the gradient tensors, whose rows [row_begin, row_end) are final on the current stream from
    there on.  `on_done()` runs when the extension call has returned.  `None` removes both.  See dp.py: RowsGradOverlap."""
    global _rows_done_hook
    _C.set_grad_rows_hook(on_rows, chunks)
    _rows_done_hook = on_done if on_rows is not None else None


_accum_leaf = None       # (data_ptr, numel) of the leaf whose .grad the accumulator is, or None: not checked
_accum_buffer = None     # the buffer itself (a bypassing backward call restores it)
_accum_strict = True     # a backward call whose feature input is not the leaf: raise (True) or take the autograd path (False)
_accum_bypassed = 0      # backward calls that took the autograd path since the accumulator was set


def accumulator_bypassed() -> int:
    """Backward calls since the last set_feature_grad_accumulator(..., strict=False) whose `semantic_feature` input was not
    the accumulator's leaf and which therefore returned their feature gradient to autograd the normal way."""
    return _accum_bypassed


def set_feature_grad_accumulator(buffer: Optional[torch.Tensor], leaf: Optional[torch.Tensor] = None, strict: bool = True) -> None:
    """Several views per optimiser step (not in the reference): while `buffer` - a contiguous float32 tensor with the P x C
    elements of `semantic_feature`, normally the leaf's zero-initialised `.grad` - is set, every backward call ADDS its
    feature gradient into it (no per-view gradient tensor, zero-fill or add) and reports no gradient for `semantic_feature`
    to autograd.  The feature-gradient hook then sees the running sum.  `None` restores the default.  See dp.py: dp_step_views.

    The op's gradient goes straight into `buffer`, past the autograd chain: that is only right when the op's
    `semantic_feature` input IS the tensor `buffer` is the gradient of.  Pass that tensor as `leaf` and every backward call
    checks it (same storage, and an autograd leaf at forward time); a transformed, masked or copied feature tensor then
    raises instead of silently receiving nothing - or, with `strict=False` (what dp_step_views uses when it was not TOLD to
    accumulate), that call leaves the accumulator and the in-backward feature hook alone and hands its gradient to autograd,
    which carries it through the chain into the same `leaf.grad` (`accumulator_bypassed()` counts such calls)."""
    global _accum_leaf, _accum_buffer, _accum_strict, _accum_bypassed
    _accum_leaf = None if (buffer is None or leaf is None) else (leaf.data_ptr(), leaf.numel())
    _accum_buffer, _accum_strict, _accum_bypassed = buffer, bool(strict), 0
    _C.set_feature_grad_accumulator(buffer)


_lowres_offers = {}     # serial number of a rasterizer call -> (gx, scale): see feature_loss.py, lowres_grad=True
_lowres_claims = {}     # serial numbers of the rasterizer calls a lowres_grad loss has been attached to (at most ONE per call)
_call_serial = itertools.count(1)   # every forward call gets one; its feature_map output carries it as `_f3dgs_call`
_tls = threading.local()            # .serial: the one the forward call of THIS thread just took


def _claim_feature_grad_lowres(call_serial: int) -> bool:
    """The hand-over beside autograd has ONE slot per rasterizer call: the first lowres_grad loss on a feature map takes it
    (True); a second loss on the same map (multi-scale, two ground truths) gets False and must take the dense path, whose
    gradient autograd adds the normal way (the blend backward adds both)."""
    if call_serial in _lowres_claims:
        return False
    while len(_lowres_claims) >= 4096:
        _lowres_claims.pop(next(iter(_lowres_claims)))
    _lowres_claims[call_serial] = True
    return True


def _offer_feature_grad_lowres(call_serial: int, gx: torch.Tensor, scale: Optional[torch.Tensor]) -> None:
    """The fused feature loss leaves its gradient at the loss's resolution for the backward call of rasterizer call
    `call_serial` (the `_f3dgs_call` attribute of its feature_map output; feature_loss.fused_feature_l1, lowres_grad=True)."""
    if call_serial in _lowres_offers:
        # (cannot happen through fused_feature_l1, which lets one loss per render claim the slot; a silent overwrite would
        # lose the first loss's gradient without any error)
        raise RuntimeError("a low-resolution feature-map gradient is already waiting for this rasterizer call: at most one "
                           "lowres_grad loss per render")
    while len(_lowres_offers) >= 64:   # offers nobody came for (backward passes that raised) pin their loss scratch: drop the oldest
        _lowres_offers.pop(next(iter(_lowres_offers)))
    _lowres_offers[call_serial] = (gx, scale)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def cpu_deep_copy_tuple(input_tuple):
    """Host clones of every tensor in `input_tuple` (used for the debug snapshots)."""
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


def _call_with_snapshot(fn, args, debug: bool, dump_name: str, what: str):
    """Debug mode keeps a host copy of the arguments and writes it out if the extension throws
    (reference __init__.py:89-97 / :147-155)."""
    if not debug:
        return fn(*args)
    saved = cpu_deep_copy_tuple(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, semantic_feature, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, semantic_feature, opacities, scales, rotations, rs.scale_modifier,
                cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        (num_rendered, color, feature_map, depth, radii, geomBuffer, binningBuffer, imgBuffer) = _call_with_snapshot(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # undefined upstream gradients arrive as None instead of zero tensors: `radii` is an integer output, autograd would
        # otherwise fill a (P,) int32 zero tensor for it in front of every backward call (one launch for nothing)
        ctx.set_materialize_grads(False)
        ctx.call_serial = _tls.serial = next(_call_serial)
        ctx.feat_src = (semantic_feature.data_ptr(), semantic_feature.numel(), bool(semantic_feature.is_leaf))
        ctx.save_for_backward(colors_precomp, semantic_feature, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer)
        return color, feature_map, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_feature, _grad_radii, grad_depth):
        rs = ctx.raster_settings
        (colors_precomp, semantic_feature, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
         binningBuffer, imgBuffer) = ctx.saved_tensors
        H, W = rs.image_height, rs.image_width
        if grad_out_color is None:      # an output the loss did not use: its gradient is zero
            grad_out_color = means3D.new_zeros((3, H, W))
        if grad_out_feature is None:
            grad_out_feature = means3D.new_zeros((semantic_feature.shape[-1], H, W))
        if grad_depth is None:
            grad_depth = means3D.new_zeros((1, H, W))
        args = (rs.bg, means3D, radii, colors_precomp, semantic_feature, scales, rotations, rs.scale_modifier,
                cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color,
                grad_out_feature, grad_depth, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered,
                binningBuffer, imgBuffer, rs.debug)
        bypass = False
        if _accum_leaf is not None and semantic_feature.numel() != 0 and (
                ctx.feat_src[:2] != _accum_leaf or not ctx.feat_src[2]):
            if _accum_strict:
                raise RuntimeError("set_feature_grad_accumulator(buffer, leaf): this op's semantic_feature input is not that leaf "
                                   "(a transformed, masked or copied tensor?) - adding the op's gradient into leaf.grad would skip "
                                   "the autograd chain between them; use accumulate=False / remove the accumulator")
            # strict=False: this call goes the autograd way - no accumulator, and no in-backward reduction of what would only be
            # this view's gradient
            global _accum_bypassed
            bypass = True
            _accum_bypassed += 1
            _C.set_feature_grad_accumulator(None)
            if _feature_ready_hook is not None:
                _C.set_feature_grad_hook(None)
        offer = _lowres_offers.pop(ctx.call_serial, None)
        if offer is not None:       # this call's feature-map gradient (or part of it) waits at the loss's resolution
            _C.set_feature_grad_lowres(offer[0], offer[1])
        try:
            (grad_means2D, grad_colors_precomp, grad_semantic_feature, grad_opacities, grad_means3D, grad_cov3Ds_precomp,
             grad_sh, grad_scales, grad_rotations) = _call_with_snapshot(
                _C.rasterize_gaussians_backward, args, rs.debug, "snapshot_bw.dump", "backward")
        finally:
            if offer is not None:
                _C.set_feature_grad_lowres(None)
            if bypass:
                _C.set_feature_grad_accumulator(_accum_buffer)
                if _feature_ready_hook is not None:
                    _C.set_feature_grad_hook(_feature_ready_hook)
        if grad_semantic_feature.numel() == 0 and semantic_feature.numel() != 0:
            grad_semantic_feature = None        # accumulated into the buffer of set_feature_grad_accumulator
        if _backward_done_hook is not None:
            _backward_done_hook()
        if _rows_done_hook is not None:
            _rows_done_hook()
        # one gradient per forward input, in input order; raster_settings gets None
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_semantic_feature, grad_opacities,
                grad_scales, grad_rotations, grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, semantic_feature, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    out = _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, semantic_feature, opacities, scales,
                                    rotations, cov3Ds_precomp, raster_settings)
    out[1]._f3dgs_call = _tls.serial        # which call rendered this feature map (fused_feature_l1, lowres_grad=True)
    return out


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of points in front of the near plane of this camera."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs: Optional[torch.Tensor] = None,
                semantic_feature: Optional[torch.Tensor] = None, colors_precomp: Optional[torch.Tensor] = None,
                scales: Optional[torch.Tensor] = None, rotations: Optional[torch.Tensor] = None,
                cov3D_precomp: Optional[torch.Tensor] = None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])  # absent optionals travel as 0-element tensors (reference :214-224)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        if semantic_feature is None:  # RGB-only use: a (P,1,0) tensor keeps the C==0 path uniform
            semantic_feature = means3D.new_zeros((means3D.shape[0], 1, 0))
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, semantic_feature, opacities, scales,
                                   rotations, cov3D_precomp, self.raster_settings)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "cpu_deep_copy_tuple",
           "set_feature_grad_hook", "set_grad_rows_hook", "set_feature_grad_accumulator", "accumulator_bypassed"]
