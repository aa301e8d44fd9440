"""GPU parity PINNED TO THE REFERENCE ITSELF: the product's `_C` against the reference's own `_C`
(`oracle/_ref/_refC<n>[s]`, compiled for gfx950 from /root/reference by `oracle/build_ref.py`), both driven
through the same positional calls (`rasterize_points.h:18-72`) on the same seeded inputs.

The reference exists here in two builds of the same sources: the DEFAULT one (the compiler contracts a*b+c into FMAs, as
nvcc does for the original) and the STRICT one (`-ffp-contract=off`, as the product's preprocess is built).  Every case
runs a three-way comparison:

  1. product vs STRICT reference - the north-star bars with nothing added: integer artefacts EXACT (radii, tiles_touched,
     num_rendered), final transmittance <= 1e-5, RGB / feature / depth <= 1e-4 ABSOLUTE, every gradient element inside
     1e-3 |g| + 1e-5 max|g|.  The blend has two hard thresholds (alpha < 1/255 skip, T < 1e-4 stop; `forward.cu:349-358`); a
     pixel where two fp32 evaluations fall on different sides is a FLIP.  Flips are proven from the two implementations' own
     n_contrib / final-T planes (`refutil.flip_pixels`), counted against a budget, and EVERY pixel above a bar - proven flip or
     not - goes to the fp64 adjudicator (`tests/adjudicate.py`): it must be a borderline decision of the exact arithmetic
     (the product's value is the exact one for a threshold moved by <= 2e-4 relative) or the product must be no further from
     the exact value than the reference is.  Those pixels are excluded EXACTLY from the gradient comparison by zeroing their
     upstream gradients in both backward passes (`backward.cu:500-620` is linear in them).
  2. product vs DEFAULT reference and 3. STRICT reference vs DEFAULT reference, same statistics: what the product shows
     against the contracted build beyond the bars of 1. (a radius off by one per 250k Gaussians, a projected mean off by one
     ulp of a pixel coordinate -> final T up to 1e-4, depth up to 1.3e-4 at depth 8, a few more flips, a gradient element at
     2x its bound on the rotated 2M-Gaussian views) the reference's OWN two builds show against each other to the same
     extent.  Round 3 carried these as allowances with an explanation; here they are bounded by measurement 3.
"""
import os

import numpy as np
import pytest
import torch

import adjudicate as adj
import refutil as ru
from util import precompute_optionals, set_option

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(**kw):
    from synth import make_scene
    return make_scene(**kw)


# Threshold flips measured at full c3: 22 of 2,073,600 pixels against the strict build, 34 against the default one (1.6e-5).
# The budget is twice that rate (never below two pixels): a build that flips more often than that has a different exponent
# or a different order of operations.
def flip_budget_for(npix: int) -> int:
    return max(2, npix // 31250)


STRICT_BARS = dict(final_T=1e-5, color=1e-4, feature=1e-4, depth=1e-4)


class _Side:
    """One implementation's forward pass of a scene: outputs, image state (final T, n_contrib on the reference's lists)."""

    def __init__(self, kind, scene, C_mod, pc, pv, strict=False):
        self.kind, self.scene = kind, scene
        self.mod = ru.product_module() if kind == "prod" else ru.load_ref(C_mod, strict=strict)
        self.d = ru.device_inputs(scene, C_mod, DEV, pc, pv)
        self.C_mod = C_mod
        self.f = ru.raw_forward(self.mod, scene, self.d)
        W, H = scene["image_width"], scene["image_height"]
        if kind == "prod":
            # n_contrib is a position in the PRIVATE instance list: comparable with the reference's only when the product keeps
            # the reference's lists (option tile_cull = 0).  Both modes must give bit-identical images.
            old = set_option("tile_cull", 0)
            try:
                self.f0 = ru.raw_forward(self.mod, scene, self.d)
            finally:
                set_option("tile_cull", old)
            for i in (1, 2, 3, 4):
                assert torch.equal(self.f[i], self.f0[i]), "tile culling changed an output"
            self.img = ru.product_image_state(scene, self.f0)
        else:
            self.f0 = self.f
            self.img = ru.ref_image_state(self.f, W, H)

    def backward(self, keep):
        d = self.d
        return ru.raw_backward(self.mod, self.scene, d, self.f, d["dL_dcolor"] * keep, d["dL_dfeature"] * keep, d["dL_ddepth"] * keep)

    def tiles_touched(self):
        P = self.scene["P"]
        if self.kind == "prod":
            return ru.product_read("tiles_touched", self.scene, self.f0, np.uint32, P)
        return ru.ref_geometry_state(self.f, P, self.C_mod, want={"tiles_touched"})["tiles_touched"]


def _forward_stats(A: _Side, B: _Side, same_width: bool):
    """B against A: integer artefacts, flips, per-plane errors at the non-flip pixels, the pixels above the strict bars."""
    sc = A.scene
    W, H = sc["image_width"], sc["image_height"]
    st = {}
    rA, rB = A.f[4].cpu().numpy(), B.f[4].cpu().numpy()
    assert rA.dtype == rB.dtype and rA.shape == rB.shape
    st["radii_mismatch"] = int((rA != rB).sum())
    st["radii_max_diff"] = int(np.abs(rA.astype(np.int64) - rB).max()) if rA.size else 0
    st["num_rendered_diff"] = abs(int(A.f[0]) - int(B.f[0]))
    flips = ru.flip_pixels(A.img, B.img)
    st["flip_pixels"] = int(flips.sum())
    ok = ~flips
    dT = np.abs(A.img["final_T"].astype(np.float64) - B.img["final_T"])
    st["final_T"] = float(dT[ok].max()) if ok.any() else 0.0
    over = flips | (dT > STRICT_BARS["final_T"])
    errs = {}
    for i, k in ((1, "color"), (2, "feature"), (3, "depth")):
        a, b = A.f[i], B.f[i]
        if k == "feature" and not same_width:
            continue
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if a.numel() == 0:
            continue
        e = (a - b).abs().amax(dim=0).reshape(-1).cpu().numpy()
        errs[k] = e
        st[k] = float(e[ok].max()) if ok.any() else 0.0
        over |= e > STRICT_BARS[k]
    st["over_bar_pixels"] = int(over.sum())
    if (ok & over).any():
        sel = ok & over
        st["over_bar_nonflip"] = int(sel.sum())
        st["nonflip_worst"] = {k: float(e[sel].max()) for k, e in errs.items()}
    st["n_contrib_equal_off_flips"] = bool(np.array_equal(A.img["n_contrib"][ok], B.img["n_contrib"][ok]))
    return st, flips, over


def _backward_stats(A: _Side, B: _Side, over, names, self_noise=False):
    sc = A.scene
    W, H = sc["image_width"], sc["image_height"]
    keep = torch.from_numpy((~over).reshape(1, H, W)).to(DEV)
    gA, gB = A.backward(keep), B.backward(keep)
    gA2 = A.backward(keep) if self_noise else None      # the reference's own atomics noise (a second run of the same call)
    st = {}
    for k in sorted(names):
        a, b = gA[k], gB[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if a.numel() == 0:
            continue
        mx, worst = ru.grad_errors(b, a)
        mx_self, worst_self = ru.grad_errors(gA2[k], a) if self_noise else (float("nan"), float("nan"))
        st[k] = (mx, worst, mx_self, worst_self)
    return st, gA, gB


def _adjudicate(A: _Side, B: _Side, over, pc, pv, same_width, max_pixels=600):
    """fp64 verdict on the pixels of `over` (B = the product).  Returns counts; asserts every adjudicated pixel."""
    sc = A.scene
    W = sc["image_width"]
    pix = np.nonzero(over)[0][:max_pixels]
    if len(pix) == 0:
        return dict(adjudicated=0, borderline=0, not_worse_than_reference=0)
    done, truths = adj.forward_truth(sc, pix, pc, pv, DEV)
    sel = pix[done]
    ys, xs = torch.from_numpy(sel // W).to(DEV), torch.from_numpy(sel % W).to(DEV)

    def take(S):
        v = dict(color=S.f[1][:, ys, xs].t().cpu().numpy(), depth=S.f[3][0, ys, xs].cpu().numpy(), final_T=S.img["final_T"][sel])
        if same_width and S.f[2].numel():
            v["feature"] = S.f[2][:, ys, xs].t().cpu().numpy()
        return v
    pv_, rv_ = take(B), take(A)
    bars = {k: v for k, v in STRICT_BARS.items() if k in pv_}
    ok, e_p, e_r, bl = adj.forward_verdict(truths, pv_, rv_, bars)
    bad = np.nonzero(~ok)[0]
    assert len(bad) == 0, (f"{len(bad)} of {len(sel)} pixels above the bars are neither borderline decisions of the exact arithmetic nor as "
                           f"close to it as the reference: first pixel {int(sel[bad[0]])}, product {e_p[bad[0]]:.1f} bars and reference "
                           f"{e_r[bad[0]]:.1f} bars from the fp64 value")
    return dict(adjudicated=int(len(sel)), borderline=int(bl.sum()), not_worse_than_reference=int((ok & ~bl).sum()),
                product_median_bars=float(np.median(e_p)), reference_median_bars=float(np.median(e_r)))


# leaf gradients produced by the per-Gaussian stage (backward.cu:144-404) from the blend-level sums
CHAIN_GRADS = {"dL_dmeans3D": "means3D", "dL_dscales": "scales", "dL_drotations": "rotations", "dL_dsh": "shs"}
LEAF_OF = dict(CHAIN_GRADS, dL_dmeans2D="means2D", dL_dopacity="opacities", dL_dsemantic_feature="semantic_feature",
               dL_dcolors="colors_precomp", dL_dcov3D="cov3D_precomp")


# The conditioning allowance of adjudicate.gradient_verdict applies to the covariance chain only (cov2D -> cov3D -> scale /
# rotation, backward.cu:144-341): these are the tensors it may excuse, and only on the harsh inputs.
KAPPA_GRADS = {"dL_dscales", "dL_drotations", "dL_dcov3D"}
KAPPA_CAP = 1e5           # (largest / smallest scale)^2 beyond 1 : 316 earns nothing more: 6 bounds at most


def _kappa(scene, i, k, pv, harsh):
    """Condition number the verdict may use for tensor k of Gaussian i: 1 (no allowance) for every blend-level tensor, for
    BASELINE-shaped inputs, for precomputed covariances and for a Gaussian with a zero (or denormal) scale axis."""
    if not harsh or pv or k not in KAPPA_GRADS:
        return 1.0
    sc_i = scene["scales"][i].double().abs()
    lo = float(sc_i.min())
    if not lo > 1e-30:
        return 1.0
    return min(float(sc_i.max()) / lo, KAPPA_CAP ** 0.5) ** 2


def _adjudicate_gradients(scene, ref_side, over, gst, g_ref, g_prod, pc, pv, max_gaussians=64, harsh=False):
    """The north-star bar on every gradient element - |prod - ref| <= 1e-3 |ref| + 1e-5 max|ref| - with nothing added; every
    Gaussian that holds an element ABOVE it goes to the fp64 adjudicator (tests/adjudicate.py), as the pixels above a bar do:
    the product must be inside the bar of the exact value or no further outside it than the reference is.  (Needle-shaped
    Gaussians - scale ratios of 1 : 300 - make the cov2D / cov3D backward formulas ill-conditioned in fp32; on the heavy-tailed
    scenes the REFERENCE is up to 30 bounds away from the exact gradient there, profiles/r05_heavy_tail_probe.txt.)
    Exact value: the full fp64 backward over all tiles of the Gaussians' rectangles where those are at most 400 tiles; for
    splats that cover the image the per-Gaussian chain in fp64 from each implementation's own blend-level gradients
    (`adjudicate.local_chain_truth`) - those (dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsemantic_feature, dL_dcov3D) must
    then hold the bar themselves."""
    W, H, P = scene["image_width"], scene["image_height"], scene["P"]
    bad = {}
    for k, (mx, worst, mx_self, worst_self) in gst.items():
        if worst <= 1.0 and mx <= 1e-3:
            continue
        # inputs of the BASELINE family hold the bar as it stands - no adjudication, no allowance
        assert harsh, (f"{k}: worst element {worst:.2f}x its bound (max relative {mx:.2e}) against the strict reference build on an input "
                       f"of the synthetic family; reference run-to-run: {worst_self:.2f}x")
        a, b = g_ref[k].double(), g_prod[k].double()
        scale = float(a.abs().max()) + 1e-30
        ratio = ((b - a).abs() / (1e-3 * a.abs() + 1e-5 * scale)).reshape(P, -1).amax(dim=1)
        idx = torch.nonzero(ratio > 1.0).flatten().cpu().numpy()
        assert len(idx) <= max_gaussians, (f"{k}: {len(idx)} Gaussians hold an element outside 1e-3*|g| + 1e-5*max|g| (worst {worst:.2f}x; "
                                           f"reference run-to-run: {worst_self:.2f}x) - too many to be conditioning")
        for i in idx:
            bad.setdefault(int(i), []).append(k)
    if not bad:
        return dict(gaussians=0)
    ids = np.array(sorted(bad))
    keep = torch.from_numpy((~over).reshape(1, H, W))
    up = tuple((scene[k] * keep).contiguous() for k in ("dL_dcolor", "dL_dfeature", "dL_ddepth"))
    full = adj.gradient_truth(scene, ids, up, pc, pv, DEV)
    mode = "full fp64 backward over the Gaussians' tiles"
    if full is None:
        mode = "per-Gaussian chain in fp64 from each implementation's own blend-level gradients"
        assert float(scene["dL_ddepth"].abs().max()) == 0.0, "local-chain adjudication needs dL_ddepth == 0"
        blend_level = [k for ks in bad.values() for k in ks if k not in CHAIN_GRADS]
        assert not blend_level, f"blend-level gradients above the bar on splats too large for the tile-restricted fp64 backward: {sorted(set(blend_level))}"
        sel = torch.from_numpy(ids).to(DEV)
        t_ref = adj.local_chain_truth(scene, ids, g_ref["dL_dmeans2D"][sel], g_ref["dL_dcolors"][sel], g_ref["dL_dcov3D"][sel], pc, pv, DEV)
        t_prod = adj.local_chain_truth(scene, ids, g_prod["dL_dmeans2D"][sel], g_prod["dL_dcolors"][sel], g_prod["dL_dcov3D"][sel], pc, pv, DEV)
    worst_p, worst_r, n_el = 0.0, 0.0, 0
    for n, i in enumerate(ids):
        for k in bad[int(i)]:
            leaf = LEAF_OF[k]
            tp = (full[leaf][n] if full is not None else t_prod[leaf][n]).reshape(-1)
            tr = (full[leaf][n] if full is not None else t_ref[leaf][n]).reshape(-1)
            r_ = g_ref[k][i].reshape(-1).double().cpu().numpy()
            p_ = g_prod[k][i].reshape(-1).double().cpu().numpy()
            if k == "dL_dmeans2D":        # (the op's third column is always zero)
                tp, tr = np.concatenate([tp[:2], [0.0]])[:len(p_)], np.concatenate([tr[:2], [0.0]])[:len(p_)]
            kappa = _kappa(scene, int(i), k, pv, harsh)
            ok, e_p, e_r = adj.gradient_verdict(tp, p_, tr, r_, float(g_ref[k].abs().max()), kappa)
            assert ok, (f"gaussian {i} {k}: product {e_p:.2f} bounds from the fp64 value, reference {e_r:.2f} ({mode}); "
                        f"scales {scene['scales'][i].tolist()}")
            worst_p, worst_r, n_el = max(worst_p, e_p), max(worst_r, e_r), n_el + 1
    return dict(gaussians=int(len(ids)), tensors=n_el, mode=mode, product_worst_bounds_from_fp64=worst_p, reference_worst_bounds_from_fp64=worst_r)


def _grad_names(pc, pv, same_width):
    # dL_dcolors (gradient w.r.t. the per-Gaussian RGB) is returned in SH mode too (rasterize_points.cu:199)
    # dL_dcov3D (the cov2D stage's output, rasterize_points.cu:199) is returned whether the covariances were given or not
    names = {"dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dcov3D"}
    names |= set() if pc else {"dL_dsh"}
    names |= set() if pv else {"dL_dscales", "dL_drotations"}
    if same_width:
        names |= {"dL_dsemantic_feature"}
    return names


def _compare(scene, C_ref, pc=False, pv=False, check_state=True, self_noise=True, return_grads=False, against_default=True,
             max_adjudicated=64, harsh=False):
    """The three-way comparison of the module docstring; returns a dict of measured numbers (also asserted)."""
    W, H, P, C = scene["image_width"], scene["image_height"], scene["P"], scene["C"]
    npix = W * H
    same = C == C_ref
    names = _grad_names(pc, pv, same)
    budget = flip_budget_for(npix)
    prod = _Side("prod", scene, C, pc, pv)
    strict = _Side("ref", scene, C_ref, pc, pv, strict=True)
    stats = {}

    # ---- 1. product vs the STRICT build: the bars as the north-star states them
    st, flips, over = _forward_stats(strict, prod, same)
    assert st["radii_mismatch"] == 0, f"{st['radii_mismatch']} radii differ from the strict reference build"
    assert st["num_rendered_diff"] == 0
    vis = prod.f[4].cpu().numpy() > 0            # (the reference leaves tiles_touched of culled Gaussians unwritten)
    assert np.array_equal(strict.tiles_touched()[vis], prod.tiles_touched()[vis]), "tiles_touched differs from the strict reference build"
    assert st["flip_pixels"] <= budget, f"{st['flip_pixels']} threshold-flip pixels (budget {budget})"
    assert st["over_bar_pixels"] <= budget, f"{st['over_bar_pixels']} pixels above a bar (budget {budget})"
    assert st["n_contrib_equal_off_flips"]
    st["adjudication"] = _adjudicate(strict, prod, over, pc, pv, same)       # every pixel above a bar, flip or not
    gst, g_ref, g_prod = _backward_stats(strict, prod, over, names, self_noise)
    st["gradient_adjudication"] = _adjudicate_gradients(scene, strict, over, gst, g_ref, g_prod, pc, pv, max_adjudicated, harsh)
    st["grads"] = gst
    stats["vs_strict"] = st
    if not return_grads:
        del g_ref, g_prod

    if check_state:
        geo = ru.ref_geometry_state(strict.f, P, C_ref, want={"means2D", "conic_opacity", "rgb", "depths"})
        rec = ru.product_read("rec", scene, prod.f, np.float32, P * 12).reshape(P, 12)
        for nm, got, want, tol in (("means2D", rec[:, 0:2], geo["means2D"].reshape(P, 2), 2e-3),
                                   ("conic", rec[:, 2:5], geo["conic_opacity"].reshape(P, 4)[:, :3], None),
                                   ("opacity", rec[:, 5], geo["conic_opacity"].reshape(P, 4)[:, 3], 0.0),
                                   ("rgb", rec[:, 6:9], geo["rgb"].reshape(P, 3), 1e-5),
                                   ("depth", rec[:, 9], geo["depths"], 1e-5)):
            if nm == "rgb" and pc:     # colours come from colors_precomp: the reference never fills geom.rgb
                continue
            g_, w_ = got[vis].astype(np.float64), want[vis].astype(np.float64)
            if tol is None:     # conics: relative (they span orders of magnitude)
                e = (np.abs(g_ - w_) / (np.abs(w_).max(axis=1, keepdims=True) + 1e-30)).max() if g_.size else 0.0
                assert e <= 1e-3, (nm, e)
            else:
                e = np.abs(g_ - w_).max() if g_.size else 0.0
                assert e <= tol, (nm, e)
            stats["state_" + nm] = float(e)
            stats["state_" + nm + "_bit_equal"] = bool(np.array_equal(got[vis], want[vis]))
        del geo, rec

    # ---- 2. + 3. against the DEFAULT (FMA-contracted) build: the product, and the strict build of the same sources
    if against_default:
        dflt = _Side("ref", scene, C_ref, pc, pv, strict=False)
        st_p, _f, over_p = _forward_stats(dflt, prod, same)
        st_s, _f, over_s = _forward_stats(dflt, strict, same)
        # a radius is ceil(3 sqrt(lambda_max)) (forward.cu:232): a last-bit difference in lambda can move a value sitting on an
        # integer across it - by one, for at most one Gaussian in 250k; num_rendered follows the rectangles
        gx, gy = (W + 15) // 16, (H + 15) // 16
        # (measured: one per 250k on the BASELINE scenes, two per 200k on the heavy-tailed scene whose radii span four decades - the
        # reference's own two builds against each other; what is asserted of the PRODUCT is that it stands where the strict build does)
        assert st_p["radii_mismatch"] == st_s["radii_mismatch"] <= max(1, P // 50000) and st_p["radii_max_diff"] <= 1
        assert st_p["num_rendered_diff"] == st_s["num_rendered_diff"] <= st_p["radii_mismatch"] * (gx + gy + 1) + P // 100000
        # what the contracted build costs the PRODUCT it costs the reference's own strict build too (x 1.5 + the strict bar)
        assert st_p["flip_pixels"] <= 1.5 * st_s["flip_pixels"] + budget
        assert st_p["over_bar_pixels"] <= 1.5 * st_s["over_bar_pixels"] + budget
        for k in ("final_T", "color", "feature", "depth"):
            if k in st_p:
                assert st_p[k] <= 1.5 * st_s[k] + STRICT_BARS[k], (k, st_p[k], st_s[k])
        both = over_p | over_s
        g_p, _a, _b = _backward_stats(dflt, prod, both, names)
        g_s, _a, _b = _backward_stats(dflt, strict, both, names)
        del _a, _b
        for k in g_p:
            assert g_p[k][0] <= 1e-3, (k, g_p[k])
            assert g_p[k][1] <= max(1.0, 1.5 * g_s[k][1]), (f"{k}: worst element {g_p[k][1]:.2f}x its bound against the default build; the "
                                                             f"reference's strict build shows {g_s[k][1]:.2f}x against it")
        st_p["grads"], st_s["grads"] = g_p, g_s
        stats["vs_default"], stats["strict_reference_vs_default"] = st_p, st_s
        del dflt
    if return_grads:
        return stats, {k: g_ref[k] for k in names}, {k: g_prod[k] for k in names}
    return stats


CASES = [
    # BASELINE config c1: RGB only (the reference cannot be built with 0 channels: run its C = 3 build with zero features)
    dict(id="c1-10k-256x256-C0", seed=1, P=10000, W=256, H=256, C=0, Cref=3),
    dict(id="C3-bg-depthgrad", seed=2, P=6000, W=200, H=120, C=3, bg=(0.3, 0.6, 0.1), depth=True),
    dict(id="C16", seed=3, P=8000, W=256, H=144, C=16),
    dict(id="C32-depthgrad", seed=4, P=8000, W=240, H=136, C=32, depth=True),
    dict(id="C64", seed=5, P=6000, W=192, H=108, C=64),
    dict(id="C128-default-width", seed=6, P=6000, W=192, H=108, C=128),          # config.h:16 default
    dict(id="C256-SAM", seed=7, P=4000, W=160, H=96, C=256),
    dict(id="C16-ragged", seed=8, P=3000, W=97, H=61, C=16),
    dict(id="C16-deg0", seed=9, P=3000, W=128, H=64, C=16, degree=0),
    dict(id="C16-deg1", seed=10, P=3000, W=128, H=64, C=16, degree=1),
    dict(id="C16-deg2", seed=11, P=3000, W=128, H=64, C=16, degree=2),
    dict(id="C16-precomp-color", seed=12, P=4000, W=160, H=96, C=16, pc=True),
    dict(id="C16-precomp-cov", seed=13, P=4000, W=160, H=96, C=16, pv=True),
    dict(id="C16-dense-early-stop", seed=14, P=2500, W=64, H=64, C=16, big=True),
    dict(id="C32-yaw-scale0.7", seed=15, P=6000, W=192, H=108, C=32, yaw=10.0, mod=0.7),
    # c5-shaped: 4K tile grid (240 x 135 tiles), C = 128, depth gradients on
    dict(id="c5-shaped-4K-C128-depthgrad", seed=16, P=60000, W=3840, H=2160, C=128, depth=True, lo=0.003, hi=0.03),
    # reduced c2 / c3 / c4 shapes at 1080p
    dict(id="c2-shaped-1080p-C16", seed=17, P=50000, W=1920, H=1080, C=16, lo=0.003, hi=0.03),
    dict(id="c3-shaped-1080p-C32", seed=18, P=100000, W=1920, H=1080, C=32, lo=0.003, hi=0.03),
    dict(id="c4-shaped-1080p-C256", seed=19, P=50000, W=1920, H=1080, C=256, lo=0.003, hi=0.03),
    # C = 512: the width the reference's authors ran LSeg at (README.md:330, `NUM_SEMANTIC_CHANNELS 512`): four 128-channel
    # forward windows, eight later windows of the blend backward
    dict(id="C512-LSeg", seed=20, P=3000, W=160, H=96, C=512),
    dict(id="C512-LSeg-ragged-depthgrad", seed=21, P=2500, W=113, H=75, C=512, depth=True),
    dict(id="C512-1080p-120k", seed=22, P=120000, W=1920, H=1080, C=512, lo=0.003, hi=0.03),
    # inputs outside the synthetic family (tests/util.py: harsh_scene)
    # (against the strict build only: on ill-conditioned inputs the reference's two builds are noise against each other)
    dict(id="heavy-tail-1080p-C32", seed=23, P=450000, W=1920, H=1080, C=32, harsh="heavy_tail_round", noise=False, dflt=False),
    dict(id="heavy-tail-needles-C16-depthgrad", seed=24, P=3000, W=160, H=96, C=16, harsh="heavy_tail", depth=True, dflt=False, adjud=400),
    dict(id="heavy-tail-needles-320x200-C32", seed=29, P=20000, W=320, H=200, C=32, harsh="heavy_tail", dflt=False, adjud=400),
    dict(id="opacity-0-and-1-C16", seed=25, P=8000, W=256, H=144, C=16, harsh="opacity01"),
    dict(id="opacity-0-and-1-1080p-C32", seed=26, P=100000, W=1920, H=1080, C=32, harsh="opacity01"),
    dict(id="zero-and-denormal-scales-C16", seed=27, P=8000, W=256, H=144, C=16, harsh="zero_scales"),
    dict(id="zero-and-denormal-scales-precomp-cov-C16", seed=28, P=4000, W=160, H=96, C=16, harsh="zero_scales", pv=True),
]


def _build(case):
    big = case.get("big", False)
    if "harsh" in case:
        from util import harsh_scene
        return precompute_optionals(harsh_scene(case["harsh"], P=case["P"], C=case["C"], width=case["W"], height=case["H"],
                                                seed=case["seed"], with_depth_grad=case.get("depth", False)))
    scene = _scene(P=case["P"], C=case["C"], width=case["W"], height=case["H"], seed=case["seed"],
                   sh_degree=case.get("degree", 3), with_depth_grad=case.get("depth", False),
                   scale_lo=case.get("lo", 0.02 if big else 0.005), scale_hi=case.get("hi", 0.4 if big else 0.08),
                   yaw_deg=case.get("yaw", 0.0))
    if "bg" in case:
        scene["bg"] = torch.tensor(case["bg"])
    if "mod" in case:
        scene["scale_modifier"] = case["mod"]
    return precompute_optionals(scene)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["id"])
def test_forward_backward_vs_reference(case, record_property):
    scene = _build(case)
    st = _compare(scene, case.get("Cref", case["C"]), case.get("pc", False), case.get("pv", False), self_noise=case.get("noise", True),
                  against_default=case.get("dflt", True), max_adjudicated=case.get("adjud", 64), harsh="harsh" in case)
    torch.cuda.empty_cache()
    for k, v in st.items():
        record_property(k, str(v))
    print(case["id"], st)


NEEDLE_CASES = [c for c in CASES if c.get("harsh") == "heavy_tail"]


def _needle_errors(scene, case, strict, g_ref, over):
    """One product run on a needle scene: per tensor of the covariance chain and per blend-level tensor, the worst error of the
    PRODUCT in units of the north-star bound against the fp64 gradient (full fp64 backward over the tiles of every Gaussian
    above the bound, as the main test adjudicates them) - and the reference's on the same Gaussians."""
    P, C = scene["P"], scene["C"]
    names = _grad_names(False, False, True)
    prod = _Side("prod", scene, C, False, False)
    keep = torch.from_numpy((~over).reshape(1, scene["image_height"], scene["image_width"])).to(DEV)
    g_prod = prod.backward(keep)
    ids = set()
    for k in names:
        a, b = g_ref[k].double(), g_prod[k].double()
        ratio = ((b - a).abs() / (1e-3 * a.abs() + 1e-5 * (float(a.abs().max()) + 1e-30))).reshape(P, -1).amax(dim=1)
        ids |= set(torch.nonzero(ratio > 1.0).flatten().cpu().numpy().tolist())
    out = {"gaussians_above_bound": len(ids)}
    if not ids:
        return out, g_prod
    ids = np.array(sorted(ids))
    up = tuple((scene[k] * keep.cpu()).contiguous() for k in ("dL_dcolor", "dL_dfeature", "dL_ddepth"))
    full = adj.gradient_truth(scene, ids, up, False, False, DEV)
    assert full is not None, "needle scenes are sized for the tile-restricted fp64 backward"
    for k in sorted(names):
        leaf = LEAF_OF[k]
        if leaf not in full:
            continue
        scale = float(g_ref[k].abs().max())
        e_p = e_r = 0.0
        for n, i in enumerate(ids):
            t = full[leaf][n].reshape(-1)
            p_ = g_prod[k][i].reshape(-1).double().cpu().numpy()
            r_ = g_ref[k][i].reshape(-1).double().cpu().numpy()
            if k == "dL_dmeans2D":
                t = np.concatenate([t[:2], [0.0]])[:len(p_)]
            e_p = max(e_p, float((np.abs(p_ - t) / (1e-3 * np.abs(t) + 1e-5 * scale)).max()))
            e_r = max(e_r, float((np.abs(r_ - t) / (1e-3 * np.abs(t) + 1e-5 * scale)).max()))
        out[k] = (e_p, e_r)
    return out, g_prod


@pytest.mark.parametrize("case", NEEDLE_CASES, ids=lambda c: c["id"])
def test_needles_take_the_exact_contraction_and_the_allowance_is_not_the_splits(case, option, record_property):
    """the conditioning allowance of the gradient verdict was introduced in the round that also moved the
    blend backward's contractions to two-term bf16 operands.  Which of the two needs it?  Five runs per setting (the sums are
    atomics in a varying order) of the needle scenes with option bwd_bf16 = -1 (the default: by the frame), 1 (bf16 forced) and
    0 (exact fp32); per run the worst distance of the product from the fp64 gradient, in bounds, over the Gaussians above the
    bound.  Measured in round 6 (profiles/r06_needles.txt): forced onto these scenes the bf16 shape stands 1 - 14 bounds further
    from the fp64 rotation gradient than the exact shape (everything else within half a bound; the reference: 10 - 29 bounds) -
    the split WAS part of what the allowance covered.  Hence the default: a frame that holds a Gaussian with an axis ratio above
    16 takes the HYBRID shape (the moment block, whose sums the chain amplifies, on exact-fp32 matrix instructions; feature and
    colour blocks on bf16).  Asserted: the default runs that shape here and its median run stands within half a bound of the
    forced exact runs' worst on every tensor; forced bf16 stands within half a bound on every tensor OUTSIDE the covariance chain."""
    from diff_gaussian_rasterization import _C
    scene = _build(case)
    C = scene["C"]
    strict = _Side("ref", scene, C, False, False, strict=True)
    probe = _Side("prod", scene, C, False, False)
    _st, _flips, over = _forward_stats(strict, probe, True)
    keep = torch.from_numpy((~over).reshape(1, scene["image_height"], scene["image_width"])).to(DEV)
    g_ref = strict.backward(keep)
    del probe
    runs = {-1: [], 1: [], 0: []}
    for setting in (-1, 1, 0):
        option("bwd_bf16", setting)
        for _ in range(5):
            res, _g = _needle_errors(scene, case, strict, g_ref, over)
            assert _C.last_backward_contraction() == {-1: 2, 1: 1, 0: 0}[setting], "the default must take the hybrid shape (exact moments) on needles"
            runs[setting].append(res)
            del _g
    keys = sorted({k for rs in runs.values() for r in rs for k in r if k != "gaussians_above_bound"})
    label = {-1: "default (-1)", 1: "bf16 forced ", 0: "exact fp32  "}
    lines = [f"{case['id']}: worst distance from the fp64 gradient in units of 1e-3 |g| + 1e-5 max|g| over the Gaussians above the bound; "
             f"five runs per setting; (product, reference on the same Gaussians)"]
    worst = {}
    for setting in (-1, 1, 0):
        for n, r in enumerate(runs[setting]):
            lines.append(f"  bwd_bf16 {label[setting]} run {n}: above bound {r['gaussians_above_bound']:4d}  " +
                         "  ".join(f"{k[3:]} {r[k][0]:.2f}/{r[k][1]:.2f}" for k in keys if k in r))
        worst[setting] = {k: max((r[k][0] for r in runs[setting] if k in r), default=0.0) for k in keys}
    for setting in (-1, 1, 0):
        lines.append(f"  worst run, {label[setting]}: " + "  ".join(f"{k[3:]} {worst[setting][k]:.2f}" for k in keys))
    text = "\n".join(lines)
    print(text)
    os.makedirs(os.path.join(ru.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ru.ROOT, "gpurun_out", f"r06_needles_{case['id']}.txt"), "w") as f:
        f.write(text + "\n")
    record_property("needle_attribution", text)
    median = {st: {k: float(np.median([r[k][0] for r in runs[st] if k in r] or [0.0])) for k in keys} for st in runs}
    lines.append("  (the default's moment sums ARE the exact shape's on these scenes: its runs differ from the forced exact runs by the order of the sums only)")
    for k in keys:
        # the same fp32 sums in a varying order: on these scenes the chain spreads five runs of ONE setting over a bound or more
        # (drotations 1.23 .. 2.24 in one recorded call), so the default's MEDIAN run is held against the exact shape's WORST
        # run - two medians of five draws from one distribution differ by more than half a bound every few calls
        assert median[-1][k] <= worst[0][k] + 0.5, (f"{k}: the default's median run is {median[-1][k]:.2f} bounds from the fp64 gradient, the "
                                                    f"exact shape's worst {worst[0][k]:.2f} (median {median[0][k]:.2f})")
        if k not in KAPPA_GRADS:
            assert worst[1][k] <= worst[0][k] + 0.5, (f"{k} (not a tensor of the covariance chain): forced bf16 {worst[1][k]:.2f} bounds from the "
                                                      f"fp64 gradient, exact {worst[0][k]:.2f}")


def test_non_finite_positions_vanish_as_in_the_reference():
    """NaN / +inf / -inf coordinates on 2 % of the Gaussians (tests/util.py: harsh_scene "nonfinite_means"): on both
    implementations they project to NaN pixel coordinates and an empty tile rectangle - radii, tiles and num_rendered EXACTLY
    as the strict reference build's, images finite everywhere and within the strict bars, every gradient of every OTHER
    Gaussian within the bound, and the gradients of the non-finite Gaussians themselves zero on both sides."""
    from util import harsh_scene
    scene = precompute_optionals(harsh_scene("nonfinite_means", P=20000, C=16, width=320, height=200, seed=33))
    bad = ~torch.isfinite(scene["means3D"]).all(dim=1)
    assert 300 < int(bad.sum()) < 500
    C = scene["C"]
    prod, ref = _Side("prod", scene, C, False, False), _Side("ref", scene, C, False, False, strict=True)
    assert torch.equal(prod.f[4], ref.f[4]) and int(prod.f[0]) == int(ref.f[0])
    assert int((prod.f[4][bad.to(DEV)] != 0).sum()) == 0
    for i in (1, 2, 3):
        assert bool(torch.isfinite(prod.f[i]).all()) and bool(torch.isfinite(ref.f[i]).all())
    st, flips, over = _forward_stats(ref, prod, True)
    assert st["flip_pixels"] <= flip_budget_for(320 * 200) and st["over_bar_pixels"] <= flip_budget_for(320 * 200), st
    keep = torch.from_numpy((~over).reshape(1, 200, 320)).to(DEV)
    g_ref, g_prod = ref.backward(keep), prod.backward(keep)
    good = (~bad).to(DEV)
    for k in sorted(_grad_names(False, False, True)):
        a, b = g_ref[k], g_prod[k]
        assert a.shape == b.shape
        assert bool(torch.isfinite(b[good]).all()), k
        mx, worst = ru.grad_errors(b[good], a[good])
        assert mx <= 1e-3 and worst <= 1.0, (k, mx, worst)
        # the non-finite Gaussians were culled: whatever the reference leaves in their rows (zeros), the product leaves too
        assert bool((b[~good] == 0).all()) or bool(torch.equal(torch.nan_to_num(b[~good]), torch.nan_to_num(a[~good]))), k


@pytest.mark.parametrize("seed,P,W,H,C", [(1, 5000, 256, 256, 16), (2, 20000, 320, 200, 32), (3, 3000, 97, 61, 128),
                                          (4, 1200, 128, 96, 16), (5, 10000, 256, 256, 3)])     # (4), (5): both sorts in one launch each
def test_instance_lists_match_reference_bit_for_bit(seed, P, W, H, C, option):
    """With the product's tile culling off, its private sorted instance list and tile ranges are the
    reference's (`rasterizer_impl.cu:291-327`: hipCUB radix sort on (tile | depth) keys), bit for bit."""
    option("tile_cull", 0)
    scene = _scene(P=P, C=C, width=W, height=H, seed=seed, scale_lo=0.005, scale_hi=0.08)
    _lists_match(scene, P, W, H, C)


@pytest.mark.parametrize("kind,P,W,H", [("heavy_tail", 4000, 640, 360), ("heavy_tail_round", 6000, 1296, 720)])
def test_instance_lists_with_huge_splats_match_reference_bit_for_bit(kind, P, W, H, option):
    """The same with splats of hundreds to thousands of tiles (up to the whole grid: 920 / 3645 tiles): the emit kernel hands
    every splat above 512 tiles to the whole wave (lane = tile row for the spans, then lane = tile column) and stages the long
    ranges of the other waves through LDS window by window - the lists stay the reference's, bit for bit."""
    from util import harsh_scene
    option("tile_cull", 0)
    scene = harsh_scene(kind, P=P, C=16, width=W, height=H, seed=41)
    _lists_match(scene, P, W, H, 16, min_big=8)


def test_instance_lists_with_more_huge_splats_than_the_queue_holds(option):
    """1583 splats that each cover the whole 920-tile grid: more than the emit kernel's big-splat queue has slots for
    (BIGQ_CAP = 1024, csrc/common.h) - the splats that find it full are emitted by their own wave; every wave of the launch
    publishes, steals and finally claims what is left of its own.  Lists bit-identical to the reference's."""
    option("tile_cull", 0)
    scene = _scene(P=1600, C=16, width=640, height=360, seed=7, scale_lo=1.5, scale_hi=4.0)
    _lists_match(scene, 1600, 640, 360, 16, min_big=1500)


def _lists_match(scene, P, W, H, C, min_big=0):
    ref, prod = ru.load_ref(C), ru.product_module()
    d = ru.device_inputs(scene, C, DEV)
    f_ref, f_prod = ru.raw_forward(ref, scene, d), ru.raw_forward(prod, scene, d)
    n = int(f_ref[0])
    assert int(f_prod[0]) == n
    assert torch.equal(f_ref[4], f_prod[4])
    geo = ru.ref_geometry_state(f_ref, P, C)
    vis = f_ref[4].cpu().numpy() > 0
    tt = ru.product_read("tiles_touched", scene, f_prod, np.uint32, P)
    # (the reference leaves tiles_touched of culled Gaussians unwritten)
    assert np.array_equal(tt[vis], geo["tiles_touched"][vis]) if min_big else np.array_equal(tt, geo["tiles_touched"])
    assert int((tt[vis] > 512).sum()) >= min_big, "the scene holds no splat large enough for the cooperative emission"
    depth_equal = np.array_equal(ru.product_read("rec", scene, f_prod, np.float32, P * 12).reshape(P, 12)[:, 9][vis],
                                 geo["depths"][vis])
    pl_ref = ru.ref_point_list(f_ref)
    pl = ru.product_read("point_list", scene, f_prod, np.uint32, n)
    img = ru.ref_image_state(f_ref, W, H)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    rg = ru.product_read("ranges", scene, f_prod, np.uint32, 2 * tiles)
    # tile ranges of EMPTY tiles are left uninitialised by the reference (forward.cu never writes them,
    # rasterizer_impl.cu:316-327 only touches tile boundaries): compare non-empty tiles
    rr, rp = img["ranges"].reshape(-1, 2), rg.reshape(-1, 2)
    nonempty = rp[:, 1] > rp[:, 0]
    assert np.array_equal(rr[nonempty], rp[nonempty])
    if depth_equal:
        assert np.array_equal(pl, pl_ref)
    else:   # depths differ in the last bit for some splat (FMA contraction): same multiset per tile
        for t in np.nonzero(nonempty)[0]:
            assert np.array_equal(np.sort(pl[rp[t, 0]:rp[t, 1]]), np.sort(pl_ref[rp[t, 0]:rp[t, 1]]))


def test_mark_visible_vs_reference():
    scene = _scene(P=50000, C=16, width=64, height=64, seed=4)
    ref, prod = ru.load_ref(16), ru.product_module()
    m, v, p = (scene[k].to(DEV) for k in ("means3D", "viewmatrix", "projmatrix"))
    a, b = ref.mark_visible(m, v, p), prod.mark_visible(m, v, p)
    assert a.dtype == b.dtype == torch.bool and torch.equal(a, b)


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4", "c5"])
def test_full_size_config_vs_reference(cfg, record_property):
    """BASELINE.json configs c2 (500k, 1080p, C=16), c3 (1M, 1080p, C=32), c4 (2M, 1080p, C=256: 128-channel windows over
    long lists) and c5 (5M, 3840x2160, C=128, depth gradients on: 240 x 135 tiles, ~140M instances) at FULL size against the
    reference's kernels (`rasterizer_impl.cu:198-461`): the three-way comparison of the module docstring, strict bars
    against the `-ffp-contract=off` build, fp64 adjudication of every pixel above a bar."""
    from synth import CONFIGS
    scene = _scene(seed=0, **CONFIGS[cfg])
    big = cfg in ("c4", "c5")
    st = _compare(scene, scene["C"], check_state=True, self_noise=not big)
    for k, v in st.items():
        record_property(k, str(v))
    print(cfg, st)
    torch.cuda.empty_cache()


def test_c4_eight_views_summed_gradients_vs_reference(record_property):
    """BASELINE.json config c4 as it is meant: 2M Gaussians, C = 256, EIGHT views per iteration (view i rotated by
    i x 5 degrees) rendered one after the other; every view against the strict reference build at the strict
    bars (flips adjudicated in fp64), views 1 and 6 also against the default build (where the reference's own two builds are
    furthest apart: 164 flips and a gradient element at 2x its bound between them on the view rotated by 30 degrees); the
    per-Gaussian gradients summed over the eight views (what the data-parallel step all-reduces) against the sum of eight
    reference backward calls."""
    from synth import CONFIGS, make_camera
    scene = _scene(seed=0, **CONFIGS["c4"])
    tot_ref, tot_prod, flips, adjud = {}, {}, 0, 0
    for v in range(8):
        sc = dict(scene)
        sc.update(make_camera(scene["image_width"], scene["image_height"], yaw_deg=5.0 * v))
        st, g_ref, g_prod = _compare(sc, sc["C"], check_state=False, self_noise=False, return_grads=True, against_default=v in (1, 6))
        flips += st["vs_strict"]["flip_pixels"]
        adjud += st["vs_strict"]["adjudication"]["adjudicated"]
        print(f"c4 view {v}: {st['vs_strict']['flip_pixels']} flip pixels, adjudication {st['vs_strict']['adjudication']}")
        for k in g_ref:
            tot_ref[k] = g_ref[k].double() if k not in tot_ref else tot_ref[k].add_(g_ref[k])
            tot_prod[k] = g_prod[k].double() if k not in tot_prod else tot_prod[k].add_(g_prod[k])
        del g_ref, g_prod
        torch.cuda.empty_cache()
    for k in sorted(tot_ref):
        mx, worst = ru.grad_errors(tot_prod[k], tot_ref[k])
        record_property(k, (mx, worst))
        assert mx <= 1e-3 and worst <= 1.0, (k, mx, worst)
    record_property("flip_pixels_8_views", flips)
    record_property("adjudicated_pixels_8_views", adjud)
    print("c4 x 8 views", flips, adjud)


def test_product_library_does_not_depend_on_the_checker():
    import subprocess
    lib = os.path.join(ru.ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so")
    import diff_gaussian_rasterization._C as C_
    for so in (lib, C_.__file__):
        out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
        assert "oracle" not in out and "_ref" not in out, out
