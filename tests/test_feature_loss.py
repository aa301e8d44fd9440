"""f-2: fused feature-map loss (resize -> optional 1x1 decoder -> L1, forward + backward) against the plain-PyTorch
statement of the reference's training-loop lines (oracle/feature_loss_oracle.py: train.py:99-105 verbatim ops).

The L1 gradient is sign(residual)/n: a residual whose SIGN differs between the fp32 kernel and the fp64 oracle can only
be one of magnitude ~1e-6 (round-off of the decoded value); such elements are counted, bounded, and their effect on the
summed gradients is bounded by count/n."""
import numpy as np
import pytest
import torch


def _case(C, H, W, Cout, Hg, Wg, decoder, seed):
    g = torch.Generator().manual_seed(seed)
    fm = torch.randn(C, H, W, generator=g)
    gt = torch.randn(Cout, Hg, Wg, generator=g)
    w = (torch.randn(Cout, C, generator=g) / C ** 0.5) if decoder else None
    b = (torch.randn(Cout, generator=g) * 0.1) if decoder else None
    return fm, gt, w, b


def test_oracle_matches_the_reference_ops_in_fp32():
    """The oracle IS the reference's three lines; in fp32 it must equal them executed in fp32 (sanity of dtype plumbing)."""
    import torch.nn.functional as F
    from oracle.feature_loss_oracle import reference_feature_l1
    fm, gt, w, b = _case(8, 20, 30, 32, 7, 9, True, 1)
    x = F.interpolate(fm.unsqueeze(0), size=(7, 9), mode="bilinear", align_corners=True).squeeze(0)
    x = F.conv2d(x.unsqueeze(0), w.reshape(32, 8, 1, 1), b).squeeze(0)
    want = torch.abs(x - gt).mean()
    got = reference_feature_l1(fm, gt, w, b, dtype=torch.float32)["loss"]
    assert abs(float(got) - float(want)) < 1e-6


CASES = [
    # (C, H, W, Cout, Hg, Wg, decoder)
    (32, 54, 96, 128, 18, 32, True),         # c3-like widths, 3x downsampling
    (64, 40, 72, 256, 23, 41, True),         # SAM-like widths, ragged sizes (tail pixel tile, Ng % 4 != 0)
    (128, 45, 80, 512, 30, 40, True),        # LSeg widths
    (32, 16, 16, 128, 40, 56, True),         # upsampling
    (32, 33, 47, 96, 1, 1, True),            # one output pixel (scale 0), Cout not a multiple of 128
    (5, 30, 50, 5, 12, 17, False),           # no decoder, odd channel count
    (512, 24, 32, 512, 24, 32, False),       # no decoder, identity resize
]


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W,Cout,Hg,Wg,decoder", CASES)
def test_fused_feature_l1_matches_reference_ops(C, H, W, Cout, Hg, Wg, decoder):
    from feature_loss import fused_feature_l1
    from oracle.feature_loss_oracle import reference_feature_l1
    fm, gt, w, b = _case(C, H, W, Cout, Hg, Wg, decoder, 7)
    want = reference_feature_l1(fm, gt, w, b)
    dev = "cuda:0"
    fm_d = fm.to(dev).requires_grad_(True)
    w_d = w.to(dev).reshape(Cout, C, 1, 1).requires_grad_(True) if decoder else None    # nn.Conv2d's weight shape
    b_d = b.to(dev).requires_grad_(True) if decoder else None
    loss = fused_feature_l1(fm_d, gt.to(dev), w_d, b_d)
    (2.5 * loss).backward()                  # an upstream gradient other than 1
    n = Cout * Hg * Wg
    assert abs(float(loss.detach()) - float(want["loss"])) <= 2e-6 * max(1.0, float(want["loss"]))
    # residuals that may change sign in fp32: |r| below the round-off of the decoded value
    r = (want["decoded"] - gt.double()).abs()
    flips = int((r < 2e-5 * (1 + want["decoded"].abs())).sum())      # fp32 contraction over C terms vs fp64
    assert flips <= max(3, n // 20000)
    slack = 2.0 * flips / n
    got = fm_d.grad.cpu().double() / 2.5
    scale = float(want["d_feature_map"].abs().max())
    assert float((got - want["d_feature_map"]).abs().max()) <= 1e-5 * scale + slack * (float(w.abs().max()) if decoder else 1.0)
    if decoder:
        gw = w_d.grad.cpu().double().reshape(Cout, C) / 2.5
        gb = b_d.grad.cpu().double() / 2.5
        assert float((gw - want["d_weight"]).abs().max()) <= 1e-5 * float(want["d_weight"].abs().max()) + slack * 4.0
        assert float((gb - want["d_bias"]).abs().max()) <= 1e-5 * float(want["d_bias"].abs().max()) + slack


@pytest.mark.gpu
def test_fused_feature_l1_rejects_what_it_cannot_do():
    from feature_loss import fused_feature_l1
    dev = "cuda:0"
    with pytest.raises(Exception):           # decoder input width outside {32, 64, 128}
        fused_feature_l1(torch.zeros(48, 8, 8, device=dev), torch.zeros(192, 4, 4, device=dev),
                         torch.zeros(192, 48, device=dev), torch.zeros(192, device=dev))
    with pytest.raises(Exception):           # no decoder but channel counts differ
        fused_feature_l1(torch.zeros(8, 8, 8, device=dev), torch.zeros(16, 4, 4, device=dev))
    with pytest.raises(Exception):           # CPU tensors: no silent fallback
        fused_feature_l1(torch.zeros(8, 8, 8), torch.zeros(8, 4, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W,Cout,Hg,Wg,decoder", CASES)
@pytest.mark.parametrize("half", [False, True])
def test_fused_feature_decode_matches_reference_ops(C, H, W, Cout, Hg, Wg, decoder, half):
    """Forward only (render.py:169-171): F.interpolate(bilinear, align_corners=True) -> cnn_decoder, against the same two
    PyTorch ops in fp64; fp32 output to 1e-5 of the map's scale, fp16 output (render.py:179 stores halves) to one half-ulp
    of the fp64 value on top of that."""
    import torch.nn.functional as F
    from feature_loss import fused_feature_decode
    fm, _gt, w, b = _case(C, H, W, Cout, Hg, Wg, decoder, 11)
    x = F.interpolate(fm.double().unsqueeze(0), size=(Hg, Wg), mode="bilinear", align_corners=True).squeeze(0)
    if decoder:
        x = F.conv2d(x.unsqueeze(0), w.double().reshape(Cout, C, 1, 1), b.double()).squeeze(0)
    dev = "cuda:0"
    got = fused_feature_decode(fm.to(dev), (Hg, Wg), w.to(dev) if decoder else None, b.to(dev) if decoder else None, half=half)
    assert got.shape == (Cout if decoder else C, Hg, Wg) and got.dtype == (torch.float16 if half else torch.float32)
    scale = float(x.abs().max())
    err = (got.double().cpu() - x).abs()
    if half:
        assert float((err - x.abs() * 2.0 ** -11).max()) <= 2e-5 * scale + 2.0 ** -25      # round-to-nearest half + subnormal step
    else:
        assert float(err.max()) <= 1e-5 * scale


# ---- the loss's gradient handed to the rasterizer's backward at the LOSS's resolution (lowres_grad=True) ------------------------

def _render_and_loss(sc, gt, w, b, lowres, extra=None, dev="cuda:0", loss_scale=1.0):
    """One training-style step: render, fused feature loss (+ optional second consumer of the feature map), backward.
    Returns (loss value, leaf gradients)."""
    import diff_gaussian_rasterization as dgr
    from feature_loss import fused_feature_l1
    t = lambda x: x.to(dev)
    P = sc["means3D"].shape[0]
    settings = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]),
                                                 sc["scale_modifier"], t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"],
                                                 t(sc["campos"]), False, False)
    leaf = lambda x: t(x).clone().requires_grad_(True)
    L = dict(means3D=leaf(sc["means3D"]), means2D=leaf(torch.zeros(P, 3)), opacities=leaf(sc["opacities"]), shs=leaf(sc["shs"]),
             semantic_feature=leaf(sc["semantic_feature"]), scales=leaf(sc["scales"]), rotations=leaf(sc["rotations"]))
    color, feat, _radii, _depth = dgr.GaussianRasterizer(settings)(**L)
    w_d = t(w).requires_grad_(True) if w is not None else None
    b_d = t(b).requires_grad_(True) if b is not None else None
    loss = loss_scale * fused_feature_l1(feat, t(gt), w_d, b_d, lowres_grad=lowres)
    total = loss + (color * t(sc["dL_dcolor"])).sum()
    if extra is not None:
        total = total + (feat * t(extra)).sum()            # a second, dense consumer of the feature map
    total.backward()
    torch.cuda.synchronize()
    g = {k: v.grad.detach().cpu().numpy() for k, v in L.items() if v.grad is not None}
    if w_d is not None:
        g["decoder_w"], g["decoder_b"] = w_d.grad.cpu().numpy(), b_d.grad.cpu().numpy()
    return float(loss.detach()), g


LOWRES_CASES = [
    # (C, W, H, Hg, Wg, decoder Cout or 0, second consumer, loss scale)
    (32, 333, 208, 70, 111, 128, False, 1.0),      # ~3x reduction, ragged image, decoder
    (16, 333, 208, 69, 111, 0, False, 1.0),        # 16 channels: the pixel-lane kernel is taken for this call
    (128, 160, 96, 32, 54, 0, False, 2.5),         # three channel windows (the later ones in two staging rounds), scaled loss
    (32, 200, 120, 120, 200, 0, False, 1.0),       # identity resize: every pixel is a sample
    (32, 200, 120, 87, 143, 0, True, 1.0),         # ~1.4x reduction: rows / columns with two samples; dense gradient added
    (40, 96, 64, 1, 1, 0, False, 1.0),             # one output pixel (scale 0): only source pixel (0, 0) is sampled
]


@pytest.mark.gpu
@pytest.mark.parametrize("C,W,H,Hg,Wg,Cout,second,loss_scale", LOWRES_CASES)
def test_lowres_feature_gradient_equals_the_dense_path(C, W, H, Hg, Wg, Cout, second, loss_scale, monkeypatch):
    """fused_feature_l1(lowres_grad=True): the dense (C,H,W) gradient is never written; the rasterizer's backward applies the
    transposed resize per tile with the products and the order of the loss's own transposed-resize kernel.  Every leaf gradient
    equals the dense path's up to the order of the atomic sums (the staged tiles are the same numbers)."""
    from synth import make_scene
    sc = make_scene(P=20000, C=C, width=W, height=H, seed=43)
    g = torch.Generator().manual_seed(5)
    co = Cout if Cout else C
    gt = torch.randn(co, Hg, Wg, generator=g)
    w = (torch.randn(co, C, generator=g) / C ** 0.5) if Cout else None
    b = (torch.randn(co, generator=g) * 0.1) if Cout else None
    extra = (torch.randn(C, H, W, generator=g) / (W * H)) if second else None
    import diff_gaussian_rasterization as dgr
    handed, real = [], dgr._C.set_feature_grad_lowres

    def spy(gx, scale=None):            # the extension is handed the (Hg, Wg, C) gradient exactly once, by the lowres run
        if gx is not None:
            handed.append(tuple(gx.shape))
        return real(gx, scale)
    monkeypatch.setattr(dgr._C, "set_feature_grad_lowres", spy)
    l0, g0 = _render_and_loss(sc, gt, w, b, False, extra, loss_scale=loss_scale)
    assert handed == []
    l1, g1 = _render_and_loss(sc, gt, w, b, True, extra, loss_scale=loss_scale)
    assert handed == [(Hg, Wg, C)] and not dgr._lowres_offers
    assert l0 == l1
    assert set(g0) == set(g1)
    for k, want in g0.items():
        scale = float(np.abs(want).max()) + 1e-30
        assert float(np.abs(g1[k] - want).max()) <= 2e-5 * scale, (k, float(np.abs(g1[k] - want).max()) / scale)
    assert float(np.abs(g0["semantic_feature"]).max()) > 0


@pytest.mark.gpu
def test_lowres_feature_gradient_goes_to_the_call_that_rendered_the_map():
    """Two renders in one graph, the loss on the first one's feature map: the gradient is matched to its rasterizer call by the
    call's serial number (carried by the feature map), whatever order autograd runs the two backward calls in.  A feature map that is not the rasterizer's own output
    is refused; a ground truth larger than the image falls back to the dense path."""
    import diff_gaussian_rasterization as dgr
    from feature_loss import fused_feature_l1
    from synth import make_scene
    dev = "cuda:0"
    t = lambda x: x.to(dev)
    scA = make_scene(P=8000, C=32, width=160, height=96, seed=3)
    scB = make_scene(P=8000, C=32, width=160, height=96, seed=4)
    gt = torch.randn(32, 30, 50, generator=torch.Generator().manual_seed(1)).to(dev)

    def render(sc, feat_leaf):
        P = sc["means3D"].shape[0]
        st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]),
                                               sc["scale_modifier"], t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"],
                                               t(sc["campos"]), False, False)
        return dgr.GaussianRasterizer(st)(means3D=t(sc["means3D"]), means2D=torch.zeros(P, 3, device=dev), opacities=t(sc["opacities"]),
                                          shs=t(sc["shs"]), semantic_feature=feat_leaf, scales=t(sc["scales"]), rotations=t(sc["rotations"]))

    grads = {}
    for lowres in (False, True):
        fa = t(scA["semantic_feature"]).clone().requires_grad_(True)
        fb = t(scB["semantic_feature"]).clone().requires_grad_(True)
        _ca, feat_a, _ra, _da = render(scA, fa)
        cb, _feat_b, _rb, _db = render(scB, fb)          # rendered later: its backward node is the first autograd runs
        (fused_feature_l1(feat_a, gt, lowres_grad=lowres) + cb.sum() * 1e-3).backward()
        torch.cuda.synchronize()
        grads[lowres] = (fa.grad.cpu().numpy(), None if fb.grad is None else fb.grad.cpu().numpy())
    a0, b0 = grads[False]
    a1, b1 = grads[True]
    assert float(np.abs(a0).max()) > 0
    assert float(np.abs(a1 - a0).max()) <= 2e-5 * float(np.abs(a0).max())
    assert (b0 is None and b1 is None) or float(np.abs(b1 - b0).max()) <= 2e-5 * (float(np.abs(b0).max()) + 1e-30)

    fa = t(scA["semantic_feature"]).clone().requires_grad_(True)
    _c, feat_a, _r, _d = render(scA, fa)
    with pytest.raises(ValueError):
        fused_feature_l1(feat_a * 1.0, gt, lowres_grad=True)
    # a map that needs no gradient (evaluation): nothing to hand over, the plain path, the same loss
    l_eval = fused_feature_l1(feat_a.detach(), gt, lowres_grad=True)
    assert float(l_eval) == float(fused_feature_l1(feat_a.detach(), gt))
    big = torch.randn(32, 200, 300, device=dev)          # enlarging: dense path, still the right numbers
    fused_feature_l1(feat_a, big, lowres_grad=True).backward()
    g_low = fa.grad.clone()
    fa.grad = None
    _c, feat_a, _r, _d = render(scA, fa)
    fused_feature_l1(feat_a, big, lowres_grad=False).backward()
    torch.cuda.synchronize()
    assert float((g_low - fa.grad).abs().max()) <= 2e-5 * float(fa.grad.abs().max())


@pytest.mark.gpu
def test_two_lowres_losses_on_one_feature_map_add_up():
    """two `fused_feature_l1(..., lowres_grad=True)` losses on the SAME rendered map (multi-scale, two
    ground truths).  The hand-over beside autograd has one slot per render: the first loss takes it, the second takes the
    dense path by itself and the blend backward adds both - the leaf gradient is the sum, as with two dense losses.  (Until
    round 5 the second offer silently replaced the first.)  A direct second offer for one call raises."""
    import diff_gaussian_rasterization as dgr
    from feature_loss import fused_feature_l1
    from synth import make_scene
    dev = "cuda:0"
    t = lambda x: x.to(dev)
    sc = make_scene(P=8000, C=32, width=160, height=96, seed=5)
    g = torch.Generator().manual_seed(2)
    gt1, gt2 = torch.randn(32, 30, 50, generator=g).to(dev), torch.randn(32, 48, 80, generator=g).to(dev)

    def render(feat_leaf):
        P = sc["means3D"].shape[0]
        st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]),
                                               sc["scale_modifier"], t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"],
                                               t(sc["campos"]), False, False)
        return dgr.GaussianRasterizer(st)(means3D=t(sc["means3D"]), means2D=torch.zeros(P, 3, device=dev), opacities=t(sc["opacities"]),
                                          shs=t(sc["shs"]), semantic_feature=feat_leaf, scales=t(sc["scales"]), rotations=t(sc["rotations"]))

    grads, losses = {}, {}
    for lowres in (False, True):
        f = t(sc["semantic_feature"]).clone().requires_grad_(True)
        _c, feat, _r, _d = render(f)
        loss = fused_feature_l1(feat, gt1, lowres_grad=lowres) + 0.5 * fused_feature_l1(feat, gt2, lowres_grad=lowres)
        loss.backward()
        torch.cuda.synchronize()
        grads[lowres], losses[lowres] = f.grad.cpu().numpy(), float(loss)
        assert not dgr._lowres_offers
    assert losses[True] == losses[False]
    scale = float(np.abs(grads[False]).max())
    assert scale > 0 and float(np.abs(grads[True] - grads[False]).max()) <= 2e-5 * scale
    # each loss alone gives a different gradient: the sum above really contains both
    f = t(sc["semantic_feature"]).clone().requires_grad_(True)
    _c, feat, _r, _d = render(f)
    fused_feature_l1(feat, gt1, lowres_grad=True).backward()
    assert float(np.abs(f.grad.cpu().numpy() - grads[False]).max()) > 1e-2 * scale
    # the raw hand-over refuses a second offer for one call instead of overwriting the first
    gx = torch.zeros(30, 50, 32, device=dev)
    dgr._offer_feature_grad_lowres(10 ** 9, gx, None)
    with pytest.raises(RuntimeError):
        dgr._offer_feature_grad_lowres(10 ** 9, gx, None)
    dgr._lowres_offers.pop(10 ** 9)
