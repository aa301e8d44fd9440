"""Helpers shared by the parity tests: run a synth scene through the HIP op / the oracle and compare."""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

def set_option(name: str, value: int) -> int:
    """Change a process-wide option of the product library (include/f3dgs.h); returns the previous value."""
    from diff_gaussian_rasterization import _C
    old = _C.get_option(name)
    _C.set_option(name, int(value))
    return old


GRAD_NAMES = ["dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dcolors", "dL_dsemantic_feature", "dL_dopacity",
              "dL_dscales", "dL_drotations", "dL_dcov3D"]


def precompute_optionals(scene: dict) -> dict:
    """Add colors_precomp / cov3D_precomp variants (plain torch fp32, same formulas the reference's Python
    fall-backs use: gaussian_renderer/__init__.py:218-238)."""
    from oracle.torch_oracle import _quat_to_rot, _sh_to_rgb
    sc = dict(scene)
    d = scene["means3D"] - scene["campos"][None]
    d = d / d.norm(dim=1, keepdim=True)
    sc["colors_precomp"] = torch.clamp_min(_sh_to_rgb(scene["sh_degree"], scene["shs"], d) + 0.5, 0.0).contiguous()
    R = _quat_to_rot(scene["rotations"])
    L = R * (scene["scale_modifier"] * scene["scales"])[:, None, :]
    S = L @ L.transpose(1, 2)
    sc["cov3D_precomp"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
    return sc


def run_oracle(scene: dict, use_precomp_color=False, use_precomp_cov=False, backward=True):
    from oracle.oracle import Oracle, scene_kwargs
    o = Oracle()
    out = o.forward(**scene_kwargs(scene, use_precomp_color, use_precomp_cov))
    grads = o.backward(scene["dL_dcolor"], scene["dL_dfeature"], scene["dL_ddepth"]) if backward else None
    return o, out, grads


def run_hip(scene: dict, use_precomp_color=False, use_precomp_cov=False, backward=True, device="cuda:0",
            debug=False):
    """Run through the reference-compatible Python surface (-> _C -> C ABI -> HIP kernels)."""
    import diff_gaussian_rasterization as dgr
    dev = torch.device(device)
    t = lambda x: x.to(dev)
    P = scene["means3D"].shape[0]
    settings = dgr.GaussianRasterizationSettings(
        image_height=scene["image_height"], image_width=scene["image_width"], tanfovx=scene["tanfovx"],
        tanfovy=scene["tanfovy"], bg=t(scene["bg"]), scale_modifier=scene["scale_modifier"],
        viewmatrix=t(scene["viewmatrix"]), projmatrix=t(scene["projmatrix"]), sh_degree=scene["sh_degree"],
        campos=t(scene["campos"]), prefiltered=False, debug=debug)
    leaf = lambda x: t(x).clone().requires_grad_(backward)
    L = dict(means3D=leaf(scene["means3D"]), means2D=leaf(torch.zeros(P, 3)), opacities=leaf(scene["opacities"]),
             semantic_feature=leaf(scene["semantic_feature"]))
    if use_precomp_color:
        L["colors_precomp"] = leaf(scene["colors_precomp"])
    else:
        L["shs"] = leaf(scene["shs"])
    if use_precomp_cov:
        L["cov3D_precomp"] = leaf(scene["cov3D_precomp"])
    else:
        L["scales"], L["rotations"] = leaf(scene["scales"]), leaf(scene["rotations"])
    color, feat, radii, depth = dgr.GaussianRasterizer(settings)(**L)
    out = dict(color=color.detach().cpu().numpy(), feature_map=feat.detach().cpu().numpy(),
               depth=depth.detach().cpu().numpy(), radii=radii.cpu().numpy())
    grads = None
    if backward:
        loss = (color * t(scene["dL_dcolor"])).sum() + (depth * t(scene["dL_ddepth"])).sum()
        if scene["C"]:
            loss = loss + (feat * t(scene["dL_dfeature"])).sum()
        loss.backward()
        g = lambda k: (L[k].grad.detach().cpu().numpy() if k in L and L[k].grad is not None else None)
        grads = dict(dL_dmeans3D=g("means3D"), dL_dmeans2D=g("means2D"), dL_dsh=g("shs"),
                     dL_dcolors=g("colors_precomp"), dL_dsemantic_feature=g("semantic_feature"),
                     dL_dopacity=g("opacities"), dL_dscales=g("scales"), dL_drotations=g("rotations"),
                     dL_dcov3D=g("cov3D_precomp"))
    torch.cuda.synchronize()
    return out, grads


def grad_report(name, got, want, rel=1e-3):
    """Error of `got` against `want` relative to the tensor's max magnitude; returns (max_rel, frac_bad)."""
    want = np.asarray(want, np.float64).reshape(-1)
    got = np.asarray(got, np.float64).reshape(-1)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want)
    bad = err > rel * np.abs(want) + rel * 1e-2 * scale
    return float(err.max() / scale), float(bad.mean())


def harsh_scene(kind: str, P: int, C: int, width: int, height: int, seed: int, with_depth_grad: bool = False) -> dict:
    """Scenes OUTSIDE the synthetic family of SURVEY.md 8(d) (uniform depth, log-uniform scales), for the comparison against
    the reference-compiled checker:

      heavy_tail  log-normal scales, independent per axis (median 0.04, sigma_log 1.3, up to 4 world units: splats larger than
                  the image AND needles with axis ratios of 1 : 300, for which the fp32 cov2D / cov3D backward formulas of
                  either implementation are ill-conditioned), half of the Gaussians nearly transparent (opacity 0.004 .. 0.03:
                  just above the 1/255 cut, so that pixels do NOT saturate early and the deep tile lists are actually walked)
      heavy_tail_round  the same size distribution with axis ratios of at most 2 : 1 (one log-normal size per Gaussian):
                  splats larger than the image and tile lists of tens of thousands of entries at 1080p without the needles -
                  the stress on the BLEND kernels alone
      opacity01   a third of the opacities exactly 0, a third exactly 1 (the 0.99 clamp, Q1; alpha = 0 < 1/255)
      zero_scales scales exactly 0 on all three axes (cov2D = the 0.3 low-pass alone), on one axis (flat splats), denormal
                  scales, and a few all-zero quaternions (the kernels do not normalise: R = I)
    """
    import math
    from synth import make_scene
    sc = make_scene(P=P, C=C, width=width, height=height, seed=seed, with_depth_grad=with_depth_grad)
    g = torch.Generator().manual_seed(seed * 7919 + 13)
    if kind in ("heavy_tail", "heavy_tail_round"):
        if kind == "heavy_tail":
            sc["scales"] = torch.exp(math.log(0.04) + 1.3 * torch.randn(P, 3, generator=g)).clamp(1e-5, 4.0).contiguous()
        else:
            size = torch.exp(math.log(0.04) + 1.3 * torch.randn(P, 1, generator=g))
            sc["scales"] = (size * (1.0 + torch.rand(P, 3, generator=g))).clamp(1e-5, 4.0).contiguous()
        low = torch.rand(P, generator=g) < 0.5
        faint = 0.004 + 0.026 * torch.rand(P, 1, generator=g)
        sc["opacities"] = torch.where(low[:, None], faint, sc["opacities"]).contiguous()
    elif kind == "opacity01":
        r = torch.rand(P, generator=g)
        op = sc["opacities"].clone()
        op[r < 1 / 3] = 0.0
        op[r > 2 / 3] = 1.0
        sc["opacities"] = op.contiguous()
    elif kind == "zero_scales":
        r = torch.rand(P, generator=g)
        s = sc["scales"].clone()
        s[r < 0.10] = 0.0
        flat = (r >= 0.10) & (r < 0.20)
        axis = torch.randint(0, 3, (P,), generator=g)
        s[flat, axis[flat]] = 0.0
        s[(r >= 0.20) & (r < 0.25)] = 1e-40           # denormal
        sc["scales"] = s.contiguous()
        q = sc["rotations"].clone()
        q[(r >= 0.25) & (r < 0.27)] = 0.0
        sc["rotations"] = q.contiguous()
    elif kind == "nonfinite_means":
        # 2 % of the Gaussians with a NaN / +inf / -inf coordinate: projected to NaN pixel coordinates, their tile rectangle
        # is empty on either implementation (forward.cu:232-238: the int conversions of NaN give an empty rectangle) - they
        # must vanish without a trace: no hang, no NaN anywhere in the images or in any other Gaussian's gradient
        r = torch.rand(P, generator=g)
        m = sc["means3D"].clone()
        axis = torch.randint(0, 3, (P,), generator=g)
        for lo_, hi_, val in ((0.0, 0.007, float("nan")), (0.007, 0.014, float("inf")), (0.014, 0.02, float("-inf"))):
            sel = (r >= lo_) & (r < hi_)
            m[sel, axis[sel]] = val
        sc["means3D"] = m.contiguous()
    else:
        raise ValueError(kind)
    return sc
