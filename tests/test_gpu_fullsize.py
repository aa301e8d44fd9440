"""Full-size (BASELINE.json c2/c3 shape) property tests on the GPU: size-independent invariants instead of an
oracle that would need minutes of CPU time."""
import ctypes
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import _lib, _raw_forward, _read
from util import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    from synth import CONFIGS, make_scene
    return make_scene(seed=1, **CONFIGS["c3"])


@pytest.mark.parametrize("onesweep", [0, 1], ids=["three-kernel-passes", "single-pass-lookback"])
def test_c3_lists_are_sorted_and_consistent(c3, onesweep, option):
    option("sort_onesweep", onesweep)
    lib = _lib()
    for cull in ("0", "1"):
        option("tile_cull", int(cull))
        res = _raw_forward(c3)
        cnt = _read(lib, "counters", c3, res, np.uint32, 16)
        n_list, n_ref = int(cnt[0]), int(cnt[1])
        assert res[0] == n_ref
        tt = _read(lib, "tiles_touched", c3, res, np.uint32, c3["P"])
        assert int(tt.astype(np.int64).sum()) == n_list
        if cull == "0":
            assert n_list == n_ref
        else:
            assert n_list < n_ref
        W, H = c3["image_width"], c3["image_height"]
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        rg = _read(lib, "ranges", c3, res, np.uint32, 2 * tiles).reshape(-1, 2).astype(np.int64)
        ne = rg[rg[:, 1] > rg[:, 0]]
        assert int((ne[:, 1] - ne[:, 0]).sum()) == n_list            # ranges partition the list
        order = np.argsort(ne[:, 0])
        assert np.array_equal(ne[order][1:, 0], ne[order][:-1, 1])   # contiguous, no gaps / overlaps
        pl = _read(lib, "point_list", c3, res, np.uint32, n_list).astype(np.int64)
        ts = _read(lib, "tile_sorted", c3, res, np.uint32, n_list).astype(np.int64)
        assert np.all(np.diff(ts) >= 0)                              # sorted by tile
        rec = _read(lib, "rec", c3, res, np.float32, c3["P"] * 12).reshape(-1, 12)
        depth_bits = rec[pl, 9].view(np.uint32).astype(np.int64)
        key = (ts << 32) | depth_bits
        assert np.all(np.diff(key) >= 0)                             # ... then by depth
        same = np.diff(key) == 0
        assert np.all(np.diff(pl)[same] > 0)                         # ties: ascending Gaussian id (stable)
        radii = res[4].cpu().numpy()
        assert np.all(radii[np.unique(pl)] > 0)


def test_c3_culling_and_reruns_are_bit_identical(c3, option):
    option("tile_cull", 0)
    a = _raw_forward(c3)
    option("tile_cull", 1)
    b = _raw_forward(c3)
    c = _raw_forward(c3)
    for i in (1, 2, 3, 4):
        assert torch.equal(a[i], b[i]) and torch.equal(b[i], c[i])
    option("feature_mfma", 0)                     # VALU feature path: same fp32 fma chain order?
    d = _raw_forward(c3)
    assert torch.equal(b[1], d[1]) and torch.equal(b[3], d[3])        # colour / depth never touch the matrix pipe
    assert float((b[2] - d[2]).abs().max()) < 1e-5                    # features: same sums, different association


def _grads(scene, up_scale=1.0, colors=None, dev="cuda:0"):
    import diff_gaussian_rasterization as dgr
    t = lambda x: x.to(dev)
    P = scene["P"]
    st = dgr.GaussianRasterizationSettings(scene["image_height"], scene["image_width"], scene["tanfovx"], scene["tanfovy"],
                                           t(scene["bg"]), 1.0, t(scene["viewmatrix"]), t(scene["projmatrix"]), 3,
                                           t(scene["campos"]), False, False)
    L = dict(means3D=t(scene["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
             opacities=t(scene["opacities"]).requires_grad_(), semantic_feature=t(scene["semantic_feature"]).requires_grad_(),
             scales=t(scene["scales"]).requires_grad_(), rotations=t(scene["rotations"]).requires_grad_())
    if colors is None:
        L["shs"] = t(scene["shs"]).requires_grad_()
    else:
        L["colors_precomp"] = colors.to(dev).requires_grad_()
    color, feat, radii, depth = dgr.GaussianRasterizer(st)(**L)
    torch.autograd.backward([color, feat, depth], [up_scale * t(scene["dL_dcolor"]), up_scale * t(scene["dL_dfeature"]),
                                                   up_scale * t(scene["dL_ddepth"])])
    torch.cuda.synchronize()
    return {k: v.grad for k, v in L.items()}, (color.detach(), feat.detach(), depth.detach())


def test_c3_backward_is_linear_in_the_upstream_gradient(c3):
    g1, _ = _grads(c3, 1.0)
    g3, _ = _grads(c3, 3.0)
    for k in g1:
        a, b = 3.0 * g1[k], g3[k]
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) / scale < 2e-3, k    # atomics reorder sums: fp32 round-off only


def test_c3_weight_sum_identities(c3):
    """sum_g w[px][g] = 1 - T_final[px].  With all colours 1 and bg 0 the image is 1 - T_final, and the sum over
    Gaussians of dL/dcolour (resp. dL/dfeature) equals sum_px (1 - T_final) dL/dpix (a checksum of checksums that
    exercises blend forward and backward at full size without an oracle)."""
    ones = torch.ones(c3["P"], 3)
    g, (color, feat, depth) = _grads(c3, 1.0, colors=ones)
    lib = _lib()
    res = _raw_forward(c3)
    W, H = c3["image_width"], c3["image_height"]
    fT = torch.from_numpy(_read(lib, "final_T", c3, res, np.float32, W * H)).view(H, W).to(color.device)
    cover = 1.0 - fT
    assert float((color[0] - cover).abs().max()) < 2e-4
    up_c = c3["dL_dcolor"].to(color.device).double()
    up_f = c3["dL_dfeature"].to(color.device).double()
    want_c = (cover.double()[None] * up_c).sum(dim=(1, 2))
    got_c = g["colors_precomp"].double().sum(dim=0)
    assert torch.allclose(got_c, want_c, rtol=2e-3, atol=1e-7)
    want_f = (cover.double()[None] * up_f).sum(dim=(1, 2))
    got_f = g["semantic_feature"].double().sum(dim=(0, 1))
    assert torch.allclose(got_f, want_f, rtol=2e-3, atol=1e-7)
