"""fp64 adjudicator for the GPU parity tests (TEST INFRASTRUCTURE; nothing in the product imports it).

`tests/test_gpu_vs_ref.py` compares two fp32 implementations of the same algorithm - the product and the reference's own
kernels compiled for gfx950.  Where they differ by more than a bar, neither is "the truth".  This module asks a third
party: `oracle/torch_oracle.py` evaluated in fp64 FROM THE RAW INPUTS (its own projection, covariances, colours, its own
list order), restricted to the handful of tiles the question is about (it runs on the GPU in fp64; a full 1080p image
would take minutes).  Two questions:

  * `forward_truth(scene, pixels)`: colour / feature / depth / final transmittance / n_contrib of the given pixels as the
    exact arithmetic gives them, plus the same with the two blend thresholds (alpha >= 1/255, T >= 1e-4; forward.cu:349-358)
    moved by a relative `delta` either way - a pixel whose value changes under that move has a discrete decision that fp32
    round-off can flip (a BORDERLINE pixel);
  * `gradient_truth(scene, gaussians, upstream)`: every leaf gradient of the given Gaussians (all tiles of their bounding
    rectangles are evaluated, so their gradients are complete).

The verdict functions turn them into statements the tests assert:

  * a pixel where product and reference differ by more than the bar is acceptable iff the product is within the bar of
    the truth under SOME threshold position in [1 - delta, 1 + delta] (then the difference is a borderline decision taken
    the other way), or the product's error against the truth is within `slack` (= SLACK) x the reference's own error against the
    truth (+ bar) (then the pixel is ill-conditioned for fp32 and the product is no further from the exact value than the
    reference is);
  * the same for gradient elements.
"""
from __future__ import annotations

import numpy as np
import torch

TILE = 16
DELTA = 2e-4          # relative move of the blend thresholds that fp32 evaluation of alpha / T can explain
COND_K = 8.0          # roundings of an fp32 evaluation of the quadratic form (products, sums, the conic's own rounding): |d power| <= COND_K u S


def _tiles_of_pixels(pix: np.ndarray, W: int, H: int) -> np.ndarray:
    gx = (W + TILE - 1) // TILE
    y, x = pix // W, pix % W
    return np.unique((y // TILE) * gx + (x // TILE))


def _run(scene, pc, pv, dev, tiles, want_grads=False, upstream=None, variants=None):
    from oracle import torch_oracle
    return torch_oracle.forward_backward(scene, dtype=torch.float64, use_precomp_color=pc, use_precomp_cov=pv,
                                         want_grads=want_grads, device=dev, tiles=tiles, upstream=upstream, variants=variants)


def forward_truth(scene: dict, pix: np.ndarray, pc=False, pv=False, dev="cuda:0", delta=DELTA, max_tiles=96):
    """fp64 values at the flat pixel indices `pix` (at most `max_tiles` tiles are evaluated; pixels of further tiles are
    reported as not adjudicated).  Returns (done_mask, variants) where variants is a list of dicts - nominal thresholds
    first, then the four (+-, +-) moves of the two thresholds by a relative delta - of numpy arrays indexed like pix[done_mask]:
    color (n,3), depth (n,), feature (n,C), final_T (n,), n_contrib (n,)."""
    W, H = scene["image_width"], scene["image_height"]
    pix = np.asarray(pix, np.int64)
    tiles_all = _tiles_of_pixels(pix, W, H)
    tiles = tiles_all[:max_tiles]
    gx = (W + TILE - 1) // TILE
    t_of = ((pix // W) // TILE) * gx + ((pix % W) // TILE)
    done = np.isin(t_of, tiles)
    sel = pix[done]
    ys, xs = torch.from_numpy(sel // W), torch.from_numpy(sel % W)
    # nominal thresholds first; then each threshold moved either way (a pixel may hold one borderline decision of each kind);
    # then the alpha test moved by what ANY fp32 evaluation of the quadratic form may be off by (COND_K u S, S = the sum of the
    # magnitudes of its terms: needle-shaped splats at an angle, whose terms of 1e3 .. 1e4 cancel to a power of -5, carry an
    # absolute error of 1e-3 in the exponent whatever the order of operations - the product and the reference both do; which
    # of them lands on the exact side at a given pixel is luck), combined with the transmittance threshold moved either way
    moves = ((1.0, 1.0, 0.0), (1.0 + delta, 1.0 + delta, 0.0), (1.0 - delta, 1.0 - delta, 0.0), (1.0 + delta, 1.0 - delta, 0.0),
             (1.0 - delta, 1.0 + delta, 0.0), (1.0, 1.0, COND_K), (1.0, 1.0, -COND_K), (1.0, 1.0 + delta, COND_K), (1.0, 1.0 - delta, COND_K),
             (1.0, 1.0 + delta, -COND_K), (1.0, 1.0 - delta, -COND_K))
    res = _run(scene, pc, pv, dev, tiles.tolist(), variants=[(fa / 255.0, ft * 1e-4, kc) for fa, ft, kc in moves])["out"]["variants"]
    out = [dict(color=r["color"][:, ys, xs].t().cpu().numpy(), depth=r["depth"][0, ys, xs].cpu().numpy(),
                feature=r["feature_map"][:, ys, xs].t().cpu().numpy(), final_T=r["final_T"][ys, xs].cpu().numpy(),
                n_contrib=r["n_contrib"][sel // W, sel % W]) for r in res]
    return done, out


SLACK = 2.0           # "no further from the exact value than the reference is": e_prod <= SLACK e_ref + 1 (round 4: 3.0)


def forward_verdict(truths, prod: dict, ref: dict, bars: dict, slack=SLACK):
    """Per adjudicated pixel: (ok, e_prod, e_ref, borderline) with errors normalised by the bars (<= 1 means inside).
    prod / ref: dicts like a truth variant (without n_contrib).  See the module docstring for the rule."""
    def nerr(val, tr):
        e = np.zeros(len(tr["final_T"]))
        for k, bar in bars.items():
            a, b = np.asarray(val[k], np.float64), np.asarray(tr[k], np.float64)
            if a.size == 0:
                continue
            d = np.abs(a - b)
            if k == "depth_rel":
                continue
            e = np.maximum(e, (d.reshape(len(e), -1).max(axis=1) if d.ndim > 1 else d) / bar)
        return e
    e_p = [nerr(prod, t) for t in truths]
    e_r = [nerr(ref, t) for t in truths]
    ep_best = np.minimum.reduce(e_p)
    borderline = (ep_best <= 1.0) & (e_p[0] > 1.0)
    ok = (ep_best <= 1.0) | (e_p[0] <= slack * e_r[0] + 1.0)
    return ok, e_p[0], e_r[0], borderline


def gradient_truth(scene: dict, gaussians: np.ndarray, upstream, pc=False, pv=False, dev="cuda:0", max_tiles=400):
    """fp64 leaf gradients of the given Gaussians (dict: the op's input names -> (n, ...) arrays), or None when their bounding
    rectangles cover more than `max_tiles` tiles.  `upstream` = the (dL_dcolor, dL_dfeature, dL_ddepth) actually used."""
    W, H = scene["image_width"], scene["image_height"]
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ids = np.asarray(gaussians, np.int64)
    radii = _run(scene, pc, pv, dev, tiles=[])["out"]["radii"].cpu().numpy()       # fp64 projection of every Gaussian, no blending
    # tile rectangles of the selected Gaussians (rasterizer_impl.cu:35-50) with one pixel of margin on the radius: the
    # fp32 radius of either implementation may be one larger than the fp64 one
    m = scene["means3D"][ids].double()
    hom = torch.cat([m, torch.ones(len(ids), 1, dtype=torch.float64)], 1) @ scene["projmatrix"].double()
    ndc = hom[:, :2] / (hom[:, 3:4] + 1e-7)
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    r = torch.from_numpy(radii[ids].astype(np.float64)) + 1.0
    x0 = ((px - r) / TILE).floor().clamp(0, gx - 1).long(); x1 = ((px + r) / TILE).floor().clamp(0, gx - 1).long()
    y0 = ((py - r) / TILE).floor().clamp(0, gy - 1).long(); y1 = ((py + r) / TILE).floor().clamp(0, gy - 1).long()
    tiles = set()
    for a, b, c, d in zip(x0.tolist(), x1.tolist(), y0.tolist(), y1.tolist()):
        for ty in range(c, d + 1):
            for tx in range(a, b + 1):
                tiles.add(ty * gx + tx)
    if len(tiles) > max_tiles:
        return None
    res = _run(scene, pc, pv, dev, sorted(tiles), want_grads=True, upstream=upstream)
    idt = torch.from_numpy(ids).to(dev)
    return {k: v[idt].cpu().numpy() for k, v in res["grads"].items()}


# ---------------------------------------------------------------- the per-Gaussian gradient chain in fp64 -----
def _cov3d_to_cov2d_jacobian(JW: torch.Tensor) -> torch.Tensor:
    """M (n, 6, 3): d(cov2D a, b, c) / d(cov3D six-vector [s00, s01, s02, s11, s12, s22]) for cov2D = T S T^T, T = JW (n, 2, 3);
    an off-diagonal parameter sits in two matrix entries (backward.cu:225-227: "off-diagonals doubled")."""
    T0, T1 = JW[:, 0, :], JW[:, 1, :]
    pairs = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))
    cols_a = [T0[:, i] * T0[:, j] * (1.0 if i == j else 2.0) for i, j in pairs]
    cols_b = [T0[:, i] * T1[:, j] if i == j else T0[:, i] * T1[:, j] + T0[:, j] * T1[:, i] for i, j in pairs]
    cols_c = [T1[:, i] * T1[:, j] * (1.0 if i == j else 2.0) for i, j in pairs]
    return torch.stack([torch.stack(cols_a, 1), torch.stack(cols_b, 1), torch.stack(cols_c, 1)], 2)


def local_chain_truth(scene: dict, gaussians: np.ndarray, dL_dmeans2D, dL_dcolors, dL_dcov3D, pc=False, pv=False, dev="cuda:0"):
    """fp64 evaluation of the PER-GAUSSIAN part of the backward pass (backward.cu:144-404: cov2D, projection, SH colour, cov3D)
    for the given Gaussians, from the blend-level gradients an implementation itself produced: dL_dmeans2D (n, 3; NDC units,
    Q8), dL_dcolors (n, 3) and dL_dcov3D (n, 6) - the last is the cov2D stage's own output (rasterize_points.cu:199 returns it),
    from which the gradient with respect to the 2D covariance is recovered by least squares over the Jacobian (6 equations,
    3 unknowns; exact up to the fp32 rounding of the given values).  No blending, no tiles: this is what adjudicates the leaf
    gradients of splats whose rectangles cover the whole image.  Needs dL_ddepth == 0 (the depth gradient per Gaussian is not
    returned by either implementation).  Returns {means3D, scales, rotations, shs | cov3D_precomp | colors_precomp: (n, ...)}."""
    from oracle import torch_oracle
    ids = torch.as_tensor(np.asarray(gaussians, np.int64))
    n = len(ids)
    f64 = lambda t: t.detach().to("cpu")[ids].to(device=dev, dtype=torch.float64)
    leaf = lambda t: f64(t).requires_grad_(True)
    L = dict(means3D=leaf(scene["means3D"]), means2D=torch.zeros(n, 3, dtype=torch.float64, device=dev),
             opacities=f64(scene["opacities"]), semantic_feature=torch.zeros(n, 1, 0, dtype=torch.float64, device=dev))
    if pc:
        L["colors_precomp"] = leaf(scene["colors_precomp"])
    else:
        L["shs"] = leaf(scene["shs"])
    if pv:
        L["cov3D_precomp"] = leaf(scene["cov3D_precomp"])
    else:
        L["scales"], L["rotations"] = leaf(scene["scales"]), leaf(scene["rotations"])
    st = torch_oracle.rasterize(**L, bg=scene["bg"], viewmatrix=scene["viewmatrix"], projmatrix=scene["projmatrix"], campos=scene["campos"],
                                tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"], image_height=scene["image_height"],
                                image_width=scene["image_width"], sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"],
                                dtype=torch.float64, tiles=[])["state"]
    up = lambda t: torch.as_tensor(t).detach().to(device=dev, dtype=torch.float64)
    M = _cov3d_to_cov2d_jacobian(st["JW"].detach())
    g2 = torch.linalg.lstsq(M, up(dL_dcov3D).reshape(n, 6, 1)).solution.reshape(n, 3)
    loss = (st["ndc"] * up(dL_dmeans2D).reshape(n, -1)[:, :2]).sum() + (st["cov2"] * g2).sum() + (st["rgb"] * up(dL_dcolors).reshape(n, 3)).sum()
    loss.backward()
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)).cpu().numpy() for k, v in L.items() if v.requires_grad}


def gradient_verdict(truth_prod, prod, truth_ref, ref, scale: float, kappa: float = 1.0):
    """One tensor of one Gaussian (ill-conditioning is a property of the Gaussian's whole chain, not of one element): with the
    errors measured in units of the north-star bound, e = max_elements |x - t| / (1e-3 |t| + 1e-5 scale), the product is
    acceptable iff it is inside the bound of the exact value, or no further outside it than the reference is from ITS exact
    value by the rule of the forward verdict (e_p <= 1 + SLACK e_r), or inside what ANY fp32 evaluation of a chain with
    condition number `kappa` can be held to: e_p <= 1 + kappa u / 1e-3, u = 2^-24 (kappa = (largest / smallest scale)^2 for the
    covariance chain: 7e4 at an axis ratio of 1 : 270 allows 5 bounds - the gradients are sums of atomics in a varying order,
    and a needle's error moves by a bound from run to run on either implementation).  Returns (ok, e_p, e_r)."""
    tp, tr = np.asarray(truth_prod, np.float64), np.asarray(truth_ref, np.float64)
    p, r = np.asarray(prod, np.float64), np.asarray(ref, np.float64)
    e_p = float((np.abs(p - tp) / (1e-3 * np.abs(tp) + 1e-5 * scale)).max())
    e_r = float((np.abs(r - tr) / (1e-3 * np.abs(tr) + 1e-5 * scale)).max())
    return e_p <= 1.0 + max(SLACK * e_r, kappa * 2.0 ** -24 / 1e-3), e_p, e_r
