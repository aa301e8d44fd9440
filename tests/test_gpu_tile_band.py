"""One view split by tile rows (include/f3dgs.h: f3dgs_set_tile_band; SURVEY.md 8(e), "alternative for single huge views").

The reference renders whole views only.  What the band calls must add up to is therefore the product's own whole-view call
(which the rest of the suite pins to the reference): inside a band the images, final T and n_contrib bit-identical; outside it
background / zeros; the bands' list lengths and reference-style counts add up exactly; the maximum of the radii is the whole
view's; and the SUM of the bands' gradients - each band given the upstream gradient of its own rows - is the whole view's
gradient up to the order of the sums."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _lib, _raw_forward, _read, _scene
from util import run_hip

pytestmark = pytest.mark.gpu


@pytest.fixture
def band():
    from diff_gaussian_rasterization import _C
    yield _C.set_tile_band
    _C.set_tile_band(0, 0)


CASES = {"ragged-C16": dict(P=20000, width=333, height=215, C=16, seed=41, scale_lo=0.005, scale_hi=0.08, with_depth_grad=True),
         "small-C0": dict(P=4000, width=128, height=96, C=0, seed=42, scale_lo=0.01, scale_hi=0.1),
         "wide-C200": dict(P=6000, width=200, height=160, C=200, seed=43, scale_lo=0.01, scale_hi=0.08)}


# a grid large enough for the blend kernels' band relabelling (common.h: band_perm; off below ~64 tiles per XCD): 8160 tiles
CASES["1080p-C8"] = dict(P=120000, width=1920, height=1080, C=8, seed=44, scale_lo=0.004, scale_hi=0.04)


@pytest.mark.parametrize("name,world", [(n, w) for n in sorted(CASES) for w in ((8,) if n == "1080p-C8" else (2, 3, 8))])
def test_bands_add_up_to_the_whole_view(name, world, band, option):
    import dp
    scene = _scene(**CASES[name])
    H, W, C = scene["image_height"], scene["image_width"], scene["C"]
    lib = _lib()
    o_full, g_full = run_hip(scene)
    _o, g_again = run_hip(scene)
    res_full = _raw_forward(scene)
    cnt_full = _read(lib, "counters", scene, res_full, np.uint32, 16)
    nc_full = _read(lib, "n_contrib", scene, res_full, np.uint32, W * H).reshape(H, W)
    fT_full = _read(lib, "final_T", scene, res_full, np.float32, W * H).reshape(H, W)
    bg = scene["bg"].numpy().reshape(3, 1, 1)

    g_sum, radii_max, n_own, n_ref = None, np.zeros(scene["P"], np.int32), 0, 0
    for r in range(world):
        r0, r1, y0, y1 = dp.band_rows(H, r, world)
        band(r0, r1)
        # upstream gradients of the band's rows only (what a rank back-propagates after gather_bands)
        sc = dict(scene)
        for k in ("dL_dcolor", "dL_ddepth", "dL_dfeature"):
            m = torch.zeros_like(scene[k])
            m[..., y0:y1, :] = scene[k][..., y0:y1, :]
            sc[k] = m
        o, g = run_hip(sc)
        for k, key in (("color", "color"), ("feature_map", "feature_map"), ("depth", "depth")):
            assert np.array_equal(o[k][..., y0:y1, :], o_full[key][..., y0:y1, :]), (k, r)
        outside = np.ones(H, bool)
        outside[y0:y1] = False
        assert np.array_equal(o["color"][:, outside], np.broadcast_to(bg, (3, H, W))[:, outside])
        assert not o["depth"][:, outside].any() and not o["feature_map"][:, outside].any()
        res = _raw_forward(sc)
        cnt = _read(lib, "counters", sc, res, np.uint32, 16)
        n_own += int(cnt[0]); n_ref += int(cnt[1])
        assert res[0] == int(cnt[1])
        nc = _read(lib, "n_contrib", sc, res, np.uint32, W * H).reshape(H, W)
        fT = _read(lib, "final_T", sc, res, np.float32, W * H).reshape(H, W)
        assert np.array_equal(nc[y0:y1], nc_full[y0:y1]) and np.array_equal(fT[y0:y1], fT_full[y0:y1])
        assert not nc[outside].any() and (fT[outside] == 1.0).all()
        radii_max = np.maximum(radii_max, o["radii"])
        if g_sum is None:
            g_sum = {k: (None if v is None else v.astype(np.float64)) for k, v in g.items()}
        else:
            for k, v in g.items():
                if v is not None:
                    g_sum[k] += v
    band(0, 0)
    assert n_own == int(cnt_full[0]) and n_ref == int(cnt_full[1]), "the bands' tiles partition the grid: the counts add up"
    assert np.array_equal(radii_max, o_full["radii"])
    for k, v in g_full.items():
        if v is None or v.size == 0:
            continue
        bound = 1e-3 * np.abs(v) + 1e-5 * np.abs(v).max()
        noise = float((np.abs(g_again[k] - v) / bound).max())
        worst = float((np.abs(g_sum[k] - v) / bound).max())
        # the same products summed per band first: within a tenth of the gradient tolerance (+ what two whole-view runs differ by)
        assert worst <= 0.1 + 4.0 * noise, (k, worst, noise)
    # and the whole view again after the band was lifted
    o_back, _g = run_hip(scene)
    for k in ("color", "feature_map", "depth", "radii"):
        assert np.array_equal(o_back[k], o_full[k]), k


def test_a_band_outside_the_grid_is_an_empty_view(band):
    scene = _scene(**CASES["small-C0"])
    band(1000, 1001)                      # clipped to the grid: no tile row left
    o, g = run_hip(scene)
    assert not o["radii"].any() and not o["depth"].any()
    assert np.array_equal(o["color"], np.broadcast_to(scene["bg"].numpy().reshape(3, 1, 1), o["color"].shape))
    for k, v in g.items():
        if v is not None:
            assert not v.any(), k
