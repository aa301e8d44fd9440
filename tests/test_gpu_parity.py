"""GPU parity tests: HIP path (Python surface -> _C -> C ABI -> kernels) against the CPU oracle.

Tolerances (BASELINE.json north_star): 1e-4 abs on RGB / feature / depth, 1e-3 rel on gradients;
integer artefacts (radii, tile counts, sorted instance list, tile ranges) are bit-exact.  A borderline
alpha (1/255) or transmittance (1e-4) decision may flip between expf implementations; such pixels are
COUNTED and bounded, never hidden.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from util import ROOT, grad_report, precompute_optionals, run_hip, run_oracle

pytestmark = pytest.mark.gpu


def _scene(**kw):
    from synth import make_scene
    return make_scene(**kw)


def _lib():
    import diff_gaussian_rasterization  # noqa: F401 (loads libf3dgs_hip.so next to torch's HIP runtime)
    lib = ctypes.CDLL(os.path.join(ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so"))
    lib.f3dgs_debug_read.restype = ctypes.c_int
    lib.f3dgs_debug_read.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.f3dgs_last_error.restype = ctypes.c_char_p
    return lib


def _raw_forward(scene, dev="cuda:0"):
    """Call _C.rasterize_gaussians directly and keep the opaque buffers for f3dgs_debug_read."""
    from diff_gaussian_rasterization import _C
    t = lambda x: x.to(dev)
    e = torch.Tensor([])
    res = _C.rasterize_gaussians(
        t(scene["bg"]), t(scene["means3D"]), e, t(scene["semantic_feature"]), t(scene["opacities"]),
        t(scene["scales"]), t(scene["rotations"]), scene["scale_modifier"], e, t(scene["viewmatrix"]),
        t(scene["projmatrix"]), scene["tanfovx"], scene["tanfovy"], scene["image_height"], scene["image_width"],
        t(scene["shs"]), scene["sh_degree"], t(scene["campos"]), False, False)
    torch.cuda.synchronize()
    return res


def _read(lib, what, scene, res, dtype, count):
    n, _, _, _, _, geom, binning, img = res
    out = np.zeros(count, dtype)
    rc = lib.f3dgs_debug_read(what.encode(), scene["P"], scene["C"], n, scene["image_width"], scene["image_height"],
                              geom.data_ptr(), binning.data_ptr() if binning.numel() else None, img.data_ptr(),
                              out.ctypes.data_as(ctypes.c_void_p), out.nbytes, None)
    assert rc == 0, lib.f3dgs_last_error()
    return out


@pytest.mark.parametrize("onesweep", [0, 1], ids=["three-kernel-passes", "single-pass-lookback"])
@pytest.mark.parametrize("seed,P,W,H,C", [(1, 5000, 256, 256, 3), (2, 20000, 320, 200, 16), (3, 3000, 97, 61, 32),
                                          # 16384 pairs is the largest problem the one-launch LDS-resident sort takes (binning.hip)
                                          (4, 16384, 160, 96, 0), (5, 16385, 160, 96, 0), (6, 1000, 64, 48, 8)])
def test_binning_is_bit_exact(seed, P, W, H, C, onesweep, option):
    option("tile_cull", 0)   # reference-identical instance lists
    option("sort_onesweep", onesweep)
    scene = _scene(P=P, C=C, width=W, height=H, seed=seed, scale_lo=0.005, scale_hi=0.08)
    o, want, _ = run_oracle(scene, backward=False)
    res = _raw_forward(scene)
    lib = _lib()
    n = res[0]
    assert n == want["num_rendered"]
    radii = res[4].cpu().numpy()
    assert np.array_equal(radii, want["radii"])
    vis = radii > 0
    rec = _read(lib, "rec", scene, res, np.float32, P * 12).reshape(P, 12)
    assert np.array_equal(rec[vis, 0:2], o.read("means2D").reshape(P, 2)[vis])
    co = o.read("conic_opacity").reshape(P, 4)
    assert np.array_equal(rec[vis][:, [2, 3, 4, 5]], co[vis])
    assert np.array_equal(rec[vis][:, [6, 7, 8]], o.read("rgb").reshape(P, 3)[vis])
    assert np.array_equal(rec[vis, 9], o.read("depths")[vis])
    tt = _read(lib, "tiles_touched", scene, res, np.uint32, P)
    assert np.array_equal(tt, o.read("tiles_touched"))
    cl = _read(lib, "clamped", scene, res, np.uint8, P)
    ocl = o.read("clamped").reshape(P, 3)
    assert np.array_equal(cl[vis], (ocl[:, 0] | (ocl[:, 1] << 1) | (ocl[:, 2] << 2))[vis])
    pl = _read(lib, "point_list", scene, res, np.uint32, n)
    assert np.array_equal(pl, o.read("point_list"))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    rg = _read(lib, "ranges", scene, res, np.uint32, 2 * tiles)
    assert np.array_equal(rg, o.read("ranges"))
    nc = _read(lib, "n_contrib", scene, res, np.uint32, W * H)
    onc = o.read("n_contrib")
    mism = int((nc != onc).sum())
    assert mism <= max(2, W * H // 20000), f"{mism} pixels disagree on n_contrib"
    fT = _read(lib, "final_T", scene, res, np.float32, W * H)
    assert np.abs(fT - o.read("final_T"))[nc == onc].max() < 1e-5


CASES = [
    dict(seed=1, P=10000, W=256, H=256, C=0),                       # BASELINE config c1
    dict(seed=2, P=6000, W=200, H=120, C=3, bg=(0.3, 0.6, 0.1), depth=True),
    dict(seed=3, P=8000, W=256, H=144, C=16),
    dict(seed=4, P=8000, W=240, H=136, C=32, depth=True),
    dict(seed=5, P=3000, W=100, H=70, C=5),                          # C not a multiple of 4, ragged image
    dict(seed=6, P=3000, W=128, H=64, C=96),                         # more than one channel window
    dict(seed=7, P=3000, W=128, H=64, C=8, degree=0),
    dict(seed=8, P=3000, W=128, H=64, C=8, degree=1),
    dict(seed=9, P=3000, W=128, H=64, C=8, degree=2),
    dict(seed=10, P=4000, W=160, H=96, C=8, precomp_color=True),
    dict(seed=11, P=4000, W=160, H=96, C=8, precomp_cov=True),
    dict(seed=12, P=2500, W=64, H=64, C=4, big=True),                # dense: early termination everywhere
]


def _strict_compare(scene, pc=False, pv=False):
    """HIP path vs the C++ oracle with the north-star bars: outputs <= 1e-4 abs, gradients <= 1e-3 rel
    (`|err| <= 1e-3 |g| + 1e-5 max|g|` for EVERY element and max err <= 1e-3 max|g|), after the pixels PROVEN to be
    threshold flips (refutil.flip_pixels: n_contrib / final-T evidence from both implementations) have been
    counted, bounded and removed exactly by zeroing their upstream gradients on both sides."""
    import refutil as ru
    from util import set_option
    W, H = scene["image_width"], scene["image_height"]
    npix = W * H
    o, want, _ = run_oracle(scene, pc, pv, backward=False)
    old = set_option("tile_cull", 0)      # n_contrib is a list position: comparable only on the reference's lists
    try:
        d = ru.device_inputs(scene, scene["C"], "cuda:0", pc, pv)
        f0 = ru.raw_forward(ru.product_module(), scene, d)
        img_prod = ru.product_image_state(scene, f0)
    finally:
        set_option("tile_cull", old)
    flips = ru.flip_pixels(dict(n_contrib=o.read("n_contrib"), final_T=o.read("final_T")), img_prod)
    nflip = int(flips.sum())
    assert nflip <= max(2, npix // 10000), f"{nflip} threshold-flip pixels"
    ok = ~flips
    masked = dict(scene)
    keep = torch.from_numpy(ok.reshape(1, H, W))
    for k in ("dL_dcolor", "dL_dfeature", "dL_ddepth"):
        masked[k] = scene[k] * keep
    got, got_g = run_hip(masked, pc, pv)
    want_g = o.backward(masked["dL_dcolor"], masked["dL_dfeature"], masked["dL_ddepth"])
    assert np.array_equal(got["radii"], want["radii"])
    for k in ("color", "feature_map", "depth"):
        if want[k].size == 0:
            assert got[k].shape == want[k].shape
            continue
        err = np.abs(got[k] - want[k]).reshape(want[k].shape[0], -1).max(0)
        assert err[ok].max() <= 1e-4, f"{k}: max abs err {err[ok].max():.3e} outside the {nflip} flip pixels"
    names = {"dL_dmeans3D", "dL_dmeans2D", "dL_dsemantic_feature", "dL_dopacity"}
    names |= {"dL_dcolors"} if pc else {"dL_dsh"}
    names |= {"dL_dcov3D"} if pv else {"dL_dscales", "dL_drotations"}
    for k in sorted(names):
        w = want_g[k]
        if w.size == 0:
            continue
        mx, worst = ru.grad_errors(got_g[k], w)
        assert mx <= 1e-3, f"{k}: max err / max|g| = {mx:.2e}"
        assert worst <= 1.0, f"{k}: worst element {worst:.2f}x outside 1e-3*|g| + 1e-5*max|g|"
    return nflip


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "bg"))
def test_forward_backward_parity(case):
    big = case.get("big", False)
    scene = _scene(P=case["P"], C=case["C"], width=case["W"], height=case["H"], seed=case["seed"],
                   sh_degree=case.get("degree", 3), with_depth_grad=case.get("depth", False),
                   scale_lo=0.02 if big else 0.005, scale_hi=0.4 if big else 0.08)
    if "bg" in case:
        scene["bg"] = torch.tensor(case["bg"])
    scene = precompute_optionals(scene)
    _strict_compare(scene, case.get("precomp_color", False), case.get("precomp_cov", False))


def test_empty_and_degenerate_inputs():
    import diff_gaussian_rasterization as dgr
    from synth import make_camera
    dev = "cuda:0"
    cam = make_camera(64, 48)
    st = dgr.GaussianRasterizationSettings(48, 64, cam["tanfovx"], cam["tanfovy"], torch.tensor([0.2, 0.4, 0.6]).to(dev),
                                           1.0, cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), 3,
                                           cam["campos"].to(dev), False, False)
    r = dgr.GaussianRasterizer(st)
    # P == 0: zero images (rasterize_points.cu:84 skips the kernels; outputs stay zero)
    z = lambda *s: torch.zeros(*s, device=dev)
    color, feat, radii, depth = r(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3),
                                  semantic_feature=z(0, 1, 4), scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, 48, 64) and feat.shape == (4, 48, 64) and depth.shape == (1, 48, 64)
    assert float(color.abs().max()) == 0 and radii.numel() == 0
    # everything behind the camera: background only, zero gradients
    P = 50
    m = torch.randn(P, 3, device=dev)
    m[:, 2] = -m[:, 2].abs() - 1.0
    m.requires_grad_(True)
    f = torch.randn(P, 1, 4, device=dev, requires_grad=True)
    color, feat, radii, depth = r(means3D=m, means2D=z(P, 3), opacities=torch.rand(P, 1, device=dev),
                                  shs=torch.randn(P, 16, 3, device=dev), semantic_feature=f,
                                  scales=torch.rand(P, 3, device=dev) * 0.1, rotations=torch.randn(P, 4, device=dev))
    assert int((radii > 0).sum()) == 0
    assert torch.allclose(color, torch.tensor([0.2, 0.4, 0.6], device=dev)[:, None, None].expand(3, 48, 64))
    (color.sum() + feat.sum()).backward()
    assert float(m.grad.abs().max()) == 0 and float(f.grad.abs().max()) == 0
    # invalid optional combinations raise like the reference (__init__.py:208-212)
    with pytest.raises(Exception):
        r(means3D=m, means2D=z(P, 3), opacities=z(P, 1), semantic_feature=f, scales=z(P, 3), rotations=z(P, 4))
    with pytest.raises(Exception):
        r(means3D=m, means2D=z(P, 3), opacities=z(P, 1), shs=z(P, 16, 3), semantic_feature=f, scales=z(P, 3))
    with pytest.raises(Exception):
        from diff_gaussian_rasterization import _C
        e = torch.Tensor([])
        _C.rasterize_gaussians(st.bg, z(P, 4), e, f, z(P, 1), z(P, 3), z(P, 4), 1.0, e, st.viewmatrix, st.projmatrix,
                               st.tanfovx, st.tanfovy, 48, 64, z(P, 16, 3), 3, st.campos, False, False)


def test_mark_visible_matches_oracle():
    import diff_gaussian_rasterization as dgr
    from oracle import oracle
    scene = _scene(P=5000, C=0, width=64, height=64, seed=4)
    dev = "cuda:0"
    st = dgr.GaussianRasterizationSettings(64, 64, scene["tanfovx"], scene["tanfovy"], scene["bg"].to(dev), 1.0,
                                           scene["viewmatrix"].to(dev), scene["projmatrix"].to(dev), 3,
                                           scene["campos"].to(dev), False, False)
    got = dgr.GaussianRasterizer(st).markVisible(scene["means3D"].to(dev)).cpu().numpy()
    want = oracle.mark_visible(scene["means3D"], scene["viewmatrix"])
    assert got.dtype == np.bool_ and np.array_equal(got, want)


def test_rotated_view_and_scale_modifier():
    scene = _scene(P=6000, C=8, width=192, height=108, seed=21, yaw_deg=10.0, scale_lo=0.005, scale_hi=0.08)
    scene["scale_modifier"] = 0.7
    _strict_compare(scene)


@pytest.mark.parametrize("seed,P,W,H,C,needles", [(31, 30000, 320, 192, 8, False), (32, 4000, 64, 64, 4, False),
                                                  (33, 20000, 640, 360, 8, True)])
def test_tile_culling_changes_lists_not_results(seed, P, W, H, C, needles, option):
    """Default mode drops instances whose 1/255 ellipse misses the tile: shorter private lists, the
    reference's num_rendered, and BIT-identical images (same blends in the same order).  `needles`: splats hundreds of
    pixels long and a fraction of a pixel wide at random angles - the fp32 round-off of the blend's quadratic form is
    largest there, the culling margins must still keep every tile that blends."""
    scene = _scene(P=P, C=C, width=W, height=H, seed=seed, scale_lo=0.005, scale_hi=0.15)
    if needles:
        g = torch.Generator().manual_seed(seed)
        long_axis = torch.randint(0, 3, (P,), generator=g)
        sc = torch.full((P, 3), 0.0015)
        sc[torch.arange(P), long_axis] = torch.exp(torch.rand(P, generator=g) * 2.0 - 1.5)      # 0.22 .. 1.6 world units
        scene["scales"] = sc.contiguous()
    lib = _lib()
    option("tile_cull", 0)
    ref = _raw_forward(scene)
    ref_n = int(_read(lib, "counters", scene, ref, np.uint32, 16)[0])
    option("tile_cull", 1)
    cul = _raw_forward(scene)
    cnt = _read(lib, "counters", scene, cul, np.uint32, 16)
    _, want, _ = run_oracle(scene, backward=False)
    assert ref[0] == cul[0] == want["num_rendered"] == ref_n == int(cnt[1])
    assert 0 < int(cnt[0]) < ref_n
    for i in (1, 2, 3, 4):   # color, feature_map, depth, radii
        assert torch.equal(ref[i], cul[i])
    # the culled list is a subsequence of the reference list, tile by tile
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    pl_ref = _read(lib, "point_list", scene, ref, np.uint32, ref_n)
    rg_ref = _read(lib, "ranges", scene, ref, np.uint32, 2 * tiles).reshape(-1, 2)
    pl_cul = _read(lib, "point_list", scene, cul, np.uint32, int(cnt[0]))
    rg_cul = _read(lib, "ranges", scene, cul, np.uint32, 2 * tiles).reshape(-1, 2)
    for t_ in range(0, tiles, max(1, tiles // 40)):
        a_, b_ = pl_ref[rg_ref[t_, 0]:rg_ref[t_, 1]], pl_cul[rg_cul[t_, 0]:rg_cul[t_, 1]]
        it = iter(a_.tolist())
        assert all(x in it for x in b_.tolist()), f"tile {t_}: culled list is not an ordered subsequence"


@pytest.mark.parametrize("C", [16, 32, 96])
def test_chunk_shapes_of_the_blend_backward_agree(C, option):
    """Option bwd_half (default 1: chunks of 32 instances against two pixel halves; 0: 64-lane chunks) changes only the
    order of the float sums: both shapes give the same gradients (C = 16: VALU feature path, 32: matrix pipe, 96: a
    second channel window that skips the geometric half of the body)."""
    from synth import make_scene
    sc = make_scene(P=30000, C=C, width=320, height=208, seed=23)
    option("bwd_half", 1)
    _, g1 = run_hip(sc)
    option("bwd_half", 0)
    _, g0 = run_hip(sc)
    for k, a in g1.items():
        if a is None:
            continue
        b = g0[k]
        scale = float(np.abs(b).max()) + 1e-30
        assert float(np.abs(a - b).max()) <= 2e-5 * scale, (k, float(np.abs(a - b).max()) / scale)


@pytest.mark.parametrize("C", [0, 3, 16, 32, 48, 96, 200])
def test_pixel_lane_and_instance_lane_backward_agree(C, option):
    """Option bwd_pl (1: pixel-lane pass + all sums on the matrix pipe, render_bwd_pl.hip - the default from 5 channels on;
    0: instance-lane kernel, render_bwd.hip) changes only the order of the float sums: same gradients for no features, ragged channel counts, one
    and several channel windows."""
    from synth import make_scene
    sc = make_scene(P=30000, C=C, width=333, height=208, seed=29)       # ragged image: masked pixels in edge tiles
    option("bwd_pl", 1)
    _, g1 = run_hip(sc)
    option("bwd_pl", 0)
    _, g0 = run_hip(sc)
    for k, a in g1.items():
        if a is None or a.size == 0:
            continue
        b = g0[k]
        scale = float(np.abs(b).max()) + 1e-30
        assert float(np.abs(a - b).max()) <= 2e-5 * scale, (k, float(np.abs(a - b).max()) / scale)


@pytest.mark.parametrize("C", [0, 3, 8, 16, 48, 64, 128])
def test_pixel_lane_backward_quadrant_split_for_up_to_16_channels(C, option):
    """Option bwd_split16 (pixel-lane backward, up to 16 channels: the feature block split over waves 0 and 1 by quadrants - two
    partial sums added in the flush - and the moment block over waves 0, 1 and 3; later channel windows of up to 32 channels - the
    last window at C = 48, 64, 128: two channel blocks x two quadrant pairs on the four waves) against the column-only split and
    against the instance-lane kernel: same gradients up to the order of the float sums."""
    from synth import make_scene
    sc = make_scene(P=30000, C=C, width=333, height=208, seed=53)
    option("bwd_bf16", 0)            # the quadrant split belongs to the fp32 shape of the kernel
    option("bwd_pl", 1)
    _, g1 = run_hip(sc)
    option("bwd_split16", 0)
    _, g2 = run_hip(sc)
    option("bwd_pl", 0)
    _, g0 = run_hip(sc)
    for got in (g1, g2):
        for k, a in got.items():
            if a is None or a.size == 0:
                continue
            b = g0[k]
            scale = float(np.abs(b).max()) + 1e-30
            assert float(np.abs(a - b).max()) <= 2e-5 * scale, (k, float(np.abs(a - b).max()) / scale)


@pytest.mark.parametrize("C", [5, 16, 32, 48, 96, 200, 512])
def test_bf16_and_fp32_contractions_of_the_pixel_lane_backward_agree(C, option):
    """Option bwd_bf16 (default 1): every per-Gaussian sum of the pixel-lane blend backward contracted on
    v_mfma_f32_16x16x32_bf16 with each fp32 operand split into two bf16 terms (x = hi + mid, |rest| <= 2^-18 |x|; products
    hi hi + hi mid + mid hi; the moment block's monomials are exact in bf16) against 0: exact-fp32 matrix instructions.  The
    forward pass is untouched (bit-identical images); every gradient element agrees to 1e-4 |g| + 1e-5 max|g| - a tenth of the
    north-star's relative bar (what the split costs is measured here; the bars against the reference are test_gpu_vs_ref.py's)."""
    from synth import make_scene
    sc = make_scene(P=30000, C=C, width=333, height=208, seed=59, with_depth_grad=True)
    option("bwd_bf16", 1)
    out1, g1 = run_hip(sc)
    option("bwd_bf16", 0)
    out0, g0 = run_hip(sc)
    for k in ("color", "feature_map", "depth", "radii"):
        assert np.array_equal(out1[k], out0[k]), k
    worst = {}
    for k, a in g1.items():
        if a is None or a.size == 0:
            continue
        b = g0[k].astype(np.float64)
        scale = float(np.abs(b).max()) + 1e-30
        worst[k] = float((np.abs(a - b) / (1e-4 * np.abs(b) + 1e-5 * scale)).max())
    print("bf16 vs fp32 contractions, worst element / (1e-4 |g| + 1e-5 max|g|):", {k: round(v, 3) for k, v in worst.items()})
    assert max(worst.values()) <= 1.0, worst


@pytest.mark.parametrize("kind,P,W,H", [("heavy_tail", 20000, 320, 200), ("opacity01", 30000, 333, 208)])
def test_bf16_and_fp32_contractions_agree_on_harsh_inputs(kind, P, W, H, option):
    """The same on inputs whose gradient sums cancel: needles and heavy-tailed sizes (mixed-sign dL/dalpha terms over
    long lists), opacities of exactly 0 and 1.  The error of the split is per PRODUCT TERM (a few 1e-6 of it), so a sum that
    cancels loses more of its value; what it may cost is held against what the ORDER of the atomic sums costs the exact-fp32
    shape on the same input: distance(bf16, fp32) <= 1 + 2 x distance(fp32, fp32 again), in units of 1e-4 |g| + 1e-5 max|g|."""
    from util import harsh_scene
    sc = harsh_scene(kind, P=P, C=32, width=W, height=H, seed=61, with_depth_grad=True)
    option("bwd_bf16", 1)          # forced: left to itself (-1) the library takes the exact contraction on the needles
    out1, g1 = run_hip(sc)
    option("bwd_bf16", 0)
    out0, g0 = run_hip(sc)
    out0b, g0b = run_hip(sc)
    for k in ("color", "feature_map", "depth", "radii"):
        assert np.array_equal(out1[k], out0[k]), k
    split, noise = {}, {}
    for k, a in g1.items():
        if a is None or a.size == 0:
            continue
        b = g0[k].astype(np.float64)
        den = 1e-4 * np.abs(b) + 1e-5 * (float(np.abs(b).max()) + 1e-30)
        split[k] = float((np.abs(a - b) / den).max())
        noise[k] = float((np.abs(g0b[k] - b) / den).max())
    print(kind, "bf16 vs fp32:", {k: round(v, 3) for k, v in split.items()}, "fp32 vs fp32 again:", {k: round(v, 3) for k, v in noise.items()})
    # the covariance chain (cov2D -> cov3D -> scale / rotation) amplifies whatever reaches it by the conditioning of the Gaussian
    # (needles: 1e4 and more) - its outputs are held against the fp64 gradient in tests/test_gpu_vs_ref.py
    # (test_needles_...); here: every tensor the blend level produces, and the position gradient
    for k in split:
        if k in ("dL_dscales", "dL_drotations", "dL_dcov3D"):
            continue
        assert split[k] <= 1.0 + 2.0 * noise[k], (k, split[k], noise[k])


def _needle_scene(ratio, shape="needle", P=20000, seed=71, C=32):
    """The synthetic family with scales (s, s / ratio, s / ratio) ["needle"] or (s, s, s / ratio) ["disc"]."""
    from synth import make_scene
    sc = make_scene(P=P, C=C, width=320, height=200, seed=seed, with_depth_grad=True, scale_lo=0.02, scale_hi=0.2)
    s = sc["scales"][:, :1]
    sc["scales"] = (torch.cat([s, s / ratio, s / ratio], dim=1) if shape == "needle" else torch.cat([s, s, s / ratio], dim=1)).contiguous()
    return sc


def _bound_distance(a, b):
    b = b.astype(np.float64)
    return float((np.abs(a - b) / (1e-3 * np.abs(b) + 1e-5 * (float(np.abs(b).max()) + 1e-30))).max())


@pytest.mark.parametrize("shape,ratio,want_bf16", [("needle", 1, True), ("needle", 8, True), ("needle", 12, True), ("disc", 15, True),
                                                   ("needle", 32, False), ("needle", 256, False), ("disc", 64, False)])
def test_contraction_precision_follows_the_conditioning_of_the_frame(shape, ratio, want_bf16, option):
    """Option bwd_bf16 = -1 (the default): the blend backward contracts on bf16 matrix instructions (two-term operands) while no
    visible Gaussian of the frame is longer than 16 times its width; otherwise the first window takes the HYBRID shape - the
    moment block, whose sums the covariance chain behind the blend (backward.cu:144-341) amplifies by the square of that ratio
    (profiles/r06_ratio_sweep.txt), on exact-fp32 matrix instructions, the feature and colour blocks on bf16 as before.  Where
    the bf16 shape is chosen EVERY gradient element is within half the north-star bound (1e-3 |g| + 1e-5 max|g|) of the exact
    shape's; where the hybrid is, the chain's tensors are the exact shape's up to the order of the sums and every other tensor
    is within half a bound.  `bwd_bf16_max_ratio` moves the switch."""
    from diff_gaussian_rasterization import _C
    sc = _needle_scene(ratio, shape)
    assert _C.get_option("bwd_bf16") == -1 and _C.get_option("bwd_bf16_max_ratio") == 16
    _o, g_auto = run_hip(sc)
    assert _C.last_backward_contraction() == (1 if want_bf16 else 2)     # 2: the hybrid shape (moment block in exact fp32)
    option("bwd_bf16", 0)
    _o, g0 = run_hip(sc)
    assert _C.last_backward_contraction() == 0
    _o, g0b = run_hip(sc)
    option("bwd_bf16", -1)
    worst = {k: _bound_distance(g_auto[k], g0[k]) for k in g_auto if g_auto[k] is not None and g_auto[k].size}
    noise = {k: _bound_distance(g0b[k], g0[k]) for k in worst}
    print(shape, ratio, "auto vs exact:", {k: round(v, 3) for k, v in worst.items()}, "exact vs exact again:", {k: round(v, 3) for k, v in noise.items()})
    for k in worst:
        if not want_bf16 and k in ("dL_dscales", "dL_drotations", "dL_dcov3D"):
            # hybrid against exact: the moment sums are the same fp32 sums in another order - on such input the chain puts that a
            # bound or more apart, as it does two runs of the exact kernel
            assert worst[k] <= 1.0 + 4.0 * noise[k], (k, worst[k], noise[k])
            continue
        assert worst[k] <= 0.5 + 2.0 * noise[k], (k, worst[k], noise[k])
    # the threshold is an option: raised, the needles take the bf16 shape; lowered to 1, nothing does
    if ratio > 1:
        option("bwd_bf16_max_ratio", 100000 if not want_bf16 else 1)
        run_hip(sc)
        assert _C.last_backward_contraction() == (2 if want_bf16 else 1)


def test_later_channel_windows_stay_on_the_bf16_contraction_on_needle_frames(option):
    """Under bwd_bf16 = -1 a frame with needles takes the exact contraction for the FIRST window (the geometric sums feed the
    covariance chain) - and keeps the bf16 contraction for the later windows of a wide feature: they carry feature sums only,
    which nothing amplifies.  C = 200 at an axis ratio of 64: the feature gradient of every channel stays within half the
    gradient bound of the all-exact result (+ what two exact runs differ by), and so does every other blend-level tensor."""
    from diff_gaussian_rasterization import _C
    sc = _needle_scene(64, "needle", P=12000, C=200)
    _o, g_auto = run_hip(sc)
    assert _C.last_backward_contraction() == 2          # (reported for the first window: the hybrid shape)
    option("bwd_bf16", 0)
    _o, g0 = run_hip(sc)
    _o, g0b = run_hip(sc)
    for k in ("dL_dsemantic_feature", "dL_dmeans2D", "dL_dopacity", "dL_dsh"):
        d, n = _bound_distance(g_auto[k], g0[k]), _bound_distance(g0b[k], g0[k])
        print(k, round(d, 3), round(n, 3))
        assert d <= 0.5 + 2.0 * n, (k, d, n)
    # channels 32.. went through the bf16 windows: not bit-equal to the exact run's, while channels 0..31 differ by the order of the sums only
    fa, f0 = g_auto["dL_dsemantic_feature"].reshape(-1, 200), g0["dL_dsemantic_feature"].reshape(-1, 200)
    assert float(np.abs(fa[:, 32:] - f0[:, 32:]).max()) > 0.0


@pytest.mark.parametrize("name", ["fwd_solo", "fwd_wide", "bwd_order", "bwd_m44", "bwd_wide8"])
@pytest.mark.parametrize("C", [16, 32, 200, 512])
def test_scheduling_options_keep_the_results(name, C, option):
    """Options fwd_solo (one workgroup per quadrant wave), fwd_wide (128-channel forward windows), bwd_order (tiles longest
    walk first in the backward), bwd_m44 (colour sums of the pixel-lane backward on 4 x 4 matrix blocks) and bwd_wide8 (later
    channel windows of the bf16 pixel-lane backward: 128 channels on eight waves per tile instead of 64 on four) select between
    complete code paths that do the same arithmetic: the forward images
    are bit-identical with the option off, the gradients equal up to the order of their atomic sums."""
    from synth import make_scene
    if C == 512 and name != "bwd_wide8":
        pytest.skip("the LSeg width is exercised for the channel-window options only")
    sc = make_scene(P=30000 if C < 512 else 8000, C=C, width=333, height=208, seed=31)
    if name == "bwd_m44":
        option("bwd_bf16", 0)        # the 4 x 4 colour blocks belong to the fp32 shape of the pixel-lane kernel
    out1, g1 = run_hip(sc)
    option(name, 0)
    out0, g0 = run_hip(sc)
    for k in ("color", "feature_map", "depth", "radii"):
        assert np.array_equal(out1[k], out0[k]), k
    for k, a in g1.items():
        if a is None or a.size == 0:
            continue
        b = g0[k]
        scale = float(np.abs(b).max()) + 1e-30
        assert float(np.abs(a - b).max()) <= 2e-5 * scale, (k, float(np.abs(a - b).max()) / scale)


@pytest.mark.parametrize("C,W,H", [(8, 656, 400), (16, 1296, 208)])
def test_instance_lane_backward_tile_order_keeps_the_results(C, W, H, option):
    """From 1024 tiles on the instance-lane backward (up to 16 channels) takes its tiles longest walk first inside every XCD's run
    (option bwd_order; the grid is padded to 8 x 4 x ceil(tiles / 8) workgroups, 1025 and 1053 tiles here: both with a ragged last
    round): same forward images bit for bit, gradients equal up to the order of their atomic sums."""
    from synth import make_scene
    sc = make_scene(P=60000, C=C, width=W, height=H, seed=37)
    out1, g1 = run_hip(sc)
    option("bwd_order", 0)
    out0, g0 = run_hip(sc)
    for k in ("color", "feature_map", "depth", "radii"):
        assert np.array_equal(out1[k], out0[k]), k
    for k, a in g1.items():
        if a is None or a.size == 0:
            continue
        b = g0[k]
        scale = float(np.abs(b).max()) + 1e-30
        assert float(np.abs(a - b).max()) <= 2e-5 * scale, (k, float(np.abs(a - b).max()) / scale)


@pytest.mark.gpu
def test_inputs_at_odd_storage_offsets_are_accepted():
    """the per-Gaussian kernels read SH rows / rotations with 16-byte accesses.  A tensor VIEW that starts 4 bytes
    into its storage (the reference accepts it) is copied by the binding and gives bit-identical results; the raw C ABI refuses a
    misaligned pointer with an error instead of faulting (include/f3dgs.h, "Alignment")."""
    import refutil as ru
    scene = _scene(P=3000, C=16, width=128, height=80, seed=31)
    mod = ru.product_module()
    d = ru.device_inputs(scene, 16, "cuda:0")
    f = ru.raw_forward(mod, scene, d)
    g = ru.raw_backward(mod, scene, d, f)
    d2 = dict(d)
    for k in ("shs", "rotations", "semantic_feature", "means3D", "scales", "opacities"):
        flat = torch.empty(d[k].numel() + 1, device=d[k].device, dtype=torch.float32)
        view = flat[1:].view(d[k].shape)
        view.copy_(d[k])
        assert view.data_ptr() % 16 == 4
        d2[k] = view
    f2 = ru.raw_forward(mod, scene, d2)
    for i in (1, 2, 3, 4):
        assert torch.equal(f[i], f2[i])
    g2 = ru.raw_backward(mod, scene, d2, f2)
    for k in g:
        if k in ("dL_dsh", "dL_dscales", "dL_drotations"):           # no atomics behind these: bit-equal given equal blend sums
            continue
        assert g[k].shape == g2[k].shape
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dsemantic_feature"):
        err = (g[k] - g2[k]).abs().max()
        assert float(err) <= 1e-5 * float(g[k].abs().max()), k
