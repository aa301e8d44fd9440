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
