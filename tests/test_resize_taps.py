"""CPU check of the per-axis tap lists of csrc/resize_taps.h (a NumPy restatement of `taps`, `resize_cand`, `resize_build` in
fp32, kept next to the header's code line by line): the blend backward takes the feature-map gradient at the resolution of the
loss and relies on two properties of them -

  * shrinking (in >= out), no source index is read by more than TWO output samples (`resize_build<2>` never overflows), and
  * the lists ARE the transpose of the forward resize, which in turn is PyTorch's bilinear interpolation with align_corners=True
    (`F.interpolate`, the call of the reference's train.py:99-101).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

f32 = np.float32


def scale_of(n_in, n_out):                       # make_resize_geom
    return f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)


def taps(o, scale, n_in):
    src = f32(scale) * f32(o)
    i0 = int(src)
    i1 = i0 + (1 if i0 < n_in - 1 else 0)
    l1 = f32(src - f32(i0))
    return i0, i1, f32(1) - l1, l1


def cand(i, scale, n_out):
    if scale <= 0:
        return (0, 0 if i == 0 else -1)
    lo = max(0, int(np.floor(f32(i - 1) / scale)) - 1)
    hi = min(n_out - 1, int(np.ceil(f32(i + 1) / scale)) + 1)
    return lo, hi


def build(i, scale, n_in, n_out):
    lo, hi = cand(i, scale, n_out)
    out = []
    for o in range(lo, hi + 1):
        a0, a1, l0, l1 = taps(o, scale, n_in)
        w = (l0 if a0 == i else f32(0)) + (l1 if a1 == i else f32(0))
        if w != 0:
            out.append((o, w))
    return out


def forward_matrix(n_in, n_out):
    M = np.zeros((n_out, n_in), np.float32)
    s = scale_of(n_in, n_out)
    for o in range(n_out):
        a0, a1, l0, l1 = taps(o, s, n_in)
        M[o, a0] += l0
        M[o, a1] += l1
    return M


PAIRS = [(a, b) for a in range(1, 41) for b in range(1, a + 1)] + [(1080, 360), (1920, 480), (2160, 720), (3840, 960), (1080, 1079),
                                                                   (1080, 541), (1297, 433), (208, 70), (333, 111)]


def test_shrinking_needs_at_most_two_output_samples_per_source_index_and_the_lists_are_the_transpose():
    for n_in, n_out in PAIRS:
        s = scale_of(n_in, n_out)
        M = forward_matrix(n_in, n_out)
        for i in range(n_in):
            lst = build(i, s, n_in, n_out)
            assert len(lst) <= 2, (n_in, n_out, i, lst)
            assert [o for o, _ in lst] == sorted(o for o, _ in lst)
            col = np.zeros(n_out, np.float32)
            for o, w in lst:
                col[o] = w
            assert np.array_equal(col, M[:, i]), (n_in, n_out, i)


@pytest.mark.parametrize("n_in,n_out", [(1080, 360), (1920, 480), (97, 61), (50, 17), (33, 1), (24, 24)])
def test_forward_taps_are_pytorchs_bilinear_align_corners(n_in, n_out):
    g = torch.Generator().manual_seed(n_in * 7 + n_out)
    # against PyTorch's own fp32 evaluation: the source coordinate scale * o is an fp32 product in both (at 1080 -> 360 it is off the
    # exact coordinate by up to 3e-5 of a pixel, which an fp64 evaluation would show as 1e-4 of the values)
    x = torch.randn(1, 1, 1, n_in, generator=g, dtype=torch.float32)
    want = F.interpolate(x, size=(1, n_out), mode="bilinear", align_corners=True)[0, 0, 0].numpy().astype(np.float64)
    got = forward_matrix(n_in, n_out).astype(np.float64) @ x[0, 0, 0].numpy().astype(np.float64)
    assert np.abs(got - want).max() <= 2e-6 * (1 + np.abs(want).max())
