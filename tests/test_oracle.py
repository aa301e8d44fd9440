"""CPU tests of the oracle itself (`-m "not gpu"`): it is pinned by (i) the reference's own Python fall-backs
(frozen by tests/golden/make_reference_fallback_vectors.py), (ii) an independent torch-autograd restatement,
(iii) frozen regression vectors."""
import os

import numpy as np
import pytest
import torch

from util import ROOT, grad_report, precompute_optionals, run_oracle

GOLD = os.path.join(ROOT, "tests", "golden")


def test_sh_to_rgb_matches_reference_eval_sh(oracle_lib):
    g = np.load(os.path.join(GOLD, "reference_fallbacks.npz"))
    for deg in range(4):
        got = oracle_lib.sh_to_rgb(deg, g["sh_means"], g["sh_campos"], g["sh_coeffs"])
        assert np.abs(got - g[f"sh_rgb_deg{deg}"]).max() < 2e-6, deg
        # the clamp quirk (Q10) must bite somewhere in the fixture, otherwise the test is vacuous
        assert (g[f"sh_rgb_deg{deg}"] == 0).any()


def test_cov3d_matches_reference_build_covariance(oracle_lib):
    g = np.load(os.path.join(GOLD, "reference_fallbacks.npz"))
    for mod in (1.0, 0.6):
        got = oracle_lib.cov3d(g["cov_scales"], mod, g["cov_rot"])
        want = g[f"cov6_mod{mod}"]
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), mod


def test_synthetic_cameras_match_the_reference_camera_class():
    """synth.make_camera hands the op the matrices the reference's Camera would build (utils/graphics_utils.py:38-71,
    scene/cameras.py:55-58), frozen from the imported reference code."""
    from synth import make_camera
    g = np.load(os.path.join(GOLD, "reference_fallbacks.npz"))
    for i, (W, H, fovx_deg, yaw_deg) in enumerate(g["cam_params"]):
        cam = make_camera(int(W), int(H), fovx_deg=float(fovx_deg), yaw_deg=float(yaw_deg))
        assert np.allclose(cam["viewmatrix"].numpy(), g[f"cam{i}_view"], rtol=0, atol=1e-7), i
        assert np.allclose(cam["projmatrix"].numpy(), g[f"cam{i}_full"], rtol=1e-6, atol=1e-7), i
        assert np.allclose(cam["campos"].numpy(), g[f"cam{i}_center"], rtol=0, atol=1e-7), i
        assert np.allclose([cam["tanfovx"], cam["tanfovy"]], g[f"cam{i}_tan"], rtol=1e-12), i


def test_oracle_regression_vectors(oracle_lib):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_oracle_regression.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    frozen = np.load(os.path.join(GOLD, "oracle_regression.npz"))
    for name in mk.SCENES:
        res = mk.run(name)
        for k, v in res.items():
            w = frozen[k]
            if np.issubdtype(np.asarray(v).dtype, np.integer):
                assert np.array_equal(v, w), k
            else:
                assert np.allclose(v, w, rtol=1e-6, atol=1e-9), k


@pytest.mark.parametrize("seed,bg,depth,pc,pv", [(5, (0, 0, 0), False, False, False), (6, (0.3, 0.6, 0.1), True, False, False),
                                                 (7, (1, 1, 1), True, True, False), (8, (0.2, 0.2, 0.2), False, False, True)])
def test_analytic_backward_equals_torch_autograd(oracle_lib, seed, bg, depth, pc, pv):
    """The quirks Q1/Q2/Q7 are written once as analytic formulas (C++) and once as detach() tricks under
    autograd (torch, fp64); they must agree."""
    from oracle import torch_oracle
    from synth import make_scene
    sc = make_scene(1500, 5, 96, 64, seed=seed, with_depth_grad=depth, scale_lo=0.02, scale_hi=0.25)
    sc["bg"] = torch.tensor(bg, dtype=torch.float32)
    sc = precompute_optionals(sc)
    o, out, g = run_oracle(sc, pc, pv)
    r = torch_oracle.forward_backward(sc, dtype=torch.float64, use_precomp_color=pc, use_precomp_cov=pv)
    to = r["out"]
    assert out["num_rendered"] == to["num_rendered"]
    assert np.array_equal(out["radii"], to["radii"].numpy())
    assert np.array_equal(o.read("point_list"), to["point_list"])
    nc = o.read("n_contrib").reshape(64, 96)
    same = nc == to["n_contrib"]
    assert same.mean() > 0.999
    for k in ("color", "feature_map", "depth"):
        err = np.abs(out[k] - to[k].detach().numpy())[..., same]
        assert err.max() < 5e-5, (k, err.max())
    pairs = {"dL_dmeans3D": "means3D", "dL_dmeans2D": "means2D", "dL_dopacity": "opacities",
             "dL_dsemantic_feature": "semantic_feature"}
    pairs.update({"dL_dcolors": "colors_precomp"} if pc else {"dL_dsh": "shs"})
    pairs.update({"dL_dcov3D": "cov3D_precomp"} if pv else {"dL_dscales": "scales", "dL_drotations": "rotations"})
    for a, b in pairs.items():
        mx, bad = grad_report(a, g[a], r["grads"][b].numpy().reshape(g[a].shape), rel=1e-3)
        assert mx < 2e-4 and bad < 1e-3, (a, mx, bad)


def test_quirks_are_exercised(oracle_lib):
    """The synthetic recipe must actually hit: near-plane culls, the 1.3x frustum clamp (Q7), early
    termination (Q5), SH clamping (Q10) and the 0.99 alpha clamp (Q1)."""
    from synth import make_scene
    sc = make_scene(1500, 2, 128, 96, seed=9, scale_lo=0.01, scale_hi=0.2)
    sc["opacities"][::50] = 0.999          # opacity * G can exceed the 0.99 clamp only if opacity does
    o, out, _ = run_oracle(sc, backward=False)
    assert (out["radii"] == 0).any() and (out["radii"] > 0).any()
    tx = sc["means3D"][:, 0] / sc["means3D"][:, 2]
    vis = out["radii"] > 0
    assert (np.abs(tx.numpy())[vis] > 1.3 * sc["tanfovx"]).any(), "no visible splat activates the EWA clamp"
    assert o.read("clamped").any()
    fT = o.read("final_T")
    assert (fT < 1e-3).any() and (fT > 0.5).any()
    co = o.read("conic_opacity").reshape(-1, 4)
    assert (co[vis, 3] > 0.99).any()


def test_empty_input(oracle_lib):
    from synth import make_scene
    sc = make_scene(0, 3, 32, 32, seed=1)
    _, out, g = run_oracle(sc)
    assert out["num_rendered"] == 0 and out["color"].shape == (3, 32, 32) and float(np.abs(out["color"]).max()) == 0
    assert g["dL_dmeans3D"].shape == (0, 3)


def test_adjudicator_tile_subset_and_threshold_variants_match_the_full_evaluation():
    """tests/adjudicate.py asks the fp64 torch oracle about a handful of tiles under five positions of the two blend thresholds.
    Pin that machinery on CPU: (1) restricted to a subset of tiles it returns, inside those tiles, exactly what the full-image
    evaluation returns; (2) variant 0 of a
    `variants` call is the nominal evaluation; (3) moving the thresholds by 2e-4 relative changes only a few pixels - by a lot at
    a pixel that holds a borderline decision, by nothing elsewhere; (4) the gradients of a tile-restricted run are those of the
    full image with the upstream gradients masked to the tiles."""
    from oracle import torch_oracle
    from synth import make_scene
    sc = make_scene(P=600, C=5, width=64, height=48, seed=17, scale_lo=0.02, scale_hi=0.2)
    gx = (64 + 15) // 16
    full = torch_oracle.forward_backward(sc, want_grads=False)["out"]
    tiles = [1, 5, 6, 10]
    sub = torch_oracle.forward_backward(sc, want_grads=False, tiles=tiles)["out"]
    for t in tiles:
        ty, tx = divmod(t, gx)
        ys, xs = slice(16 * ty, min(48, 16 * ty + 16)), slice(16 * tx, min(64, 16 * tx + 16))
        for k in ("color", "feature_map", "depth"):
            assert torch.equal(sub[k][:, ys, xs], full[k][:, ys, xs]), (k, t)
        assert torch.equal(sub["final_T"][ys, xs], full["final_T"][ys, xs])
    nominal = (1.0 / 255.0, 1e-4)
    moved = (1.0 / 255.0 * (1 + 2e-4), 1e-4 * (1 - 2e-4))
    var = torch_oracle.forward_backward(sc, want_grads=False, tiles=tiles, variants=[nominal, moved])["out"]["variants"]
    for k in ("color", "feature_map", "depth"):
        assert torch.equal(var[0][k], sub[k]), k
    changed = (var[1]["color"] != var[0]["color"]).any(0)
    assert int(changed.sum()) <= 8            # a threshold moved by 2e-4 decides differently at a few pixels at most
    same = ~changed
    assert torch.equal(var[1]["feature_map"][:, same], var[0]["feature_map"][:, same])
    mask = torch.zeros(48, 64, dtype=torch.bool)
    for t in tiles:
        ty, tx = divmod(t, gx)
        mask[16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16] = True
    up = (sc["dL_dcolor"], sc["dL_dfeature"], sc["dL_ddepth"])
    g_sub = torch_oracle.forward_backward(sc, tiles=tiles, upstream=up)["grads"]
    g_full = torch_oracle.forward_backward(sc, upstream=tuple(u * mask for u in up))["grads"]
    for k, want in g_full.items():
        assert torch.allclose(g_sub[k], want, rtol=1e-12, atol=1e-15 * float(want.abs().max() + 1)), k


def test_local_chain_truth_equals_the_full_fp64_backward():
    """tests/adjudicate.py: the per-Gaussian gradient chain evaluated in fp64 from blend-level gradients (dL_dmeans2D, dL_dcolors,
    dL_dcov3D - the cov2D gradient recovered from the latter by least squares over d cov2D / d cov3D) gives the leaf gradients
    of the full fp64 backward pass; pins the six-vector convention of dL_dcov3D (off-diagonals doubled, backward.cu:225-227)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adjudicate as adj
    from synth import make_scene
    from util import precompute_optionals
    from oracle import torch_oracle
    sc = precompute_optionals(make_scene(P=200, C=3, width=64, height=48, seed=5, scale_lo=0.02, scale_hi=0.3))
    blend = torch_oracle.forward_backward(sc, dtype=torch.float64, use_precomp_color=True, use_precomp_cov=True, device="cpu")["grads"]
    for pv in (False, True):
        want = torch_oracle.forward_backward(sc, dtype=torch.float64, use_precomp_cov=pv, device="cpu")["grads"]
        got = adj.local_chain_truth(sc, np.arange(200), blend["means2D"], blend["colors_precomp"], blend["cov3D_precomp"], pc=False, pv=pv, dev="cpu")
        for k, a in got.items():
            b = want[k].numpy()
            assert float(np.abs(a - b).max()) <= 2e-6 * (float(np.abs(b).max()) + 1e-30), (pv, k)
    ok, e_p, e_r = adj.gradient_verdict(np.array([1.0, 1.0]), np.array([1.0005, 1.01]), np.array([1.0, 1.0]), np.array([1.0, 1.02]), 1.0)
    assert ok and e_p > 9 and e_r > 19
    ok, _, _ = adj.gradient_verdict(np.array([1.0]), np.array([1.01]), np.array([1.0]), np.array([1.001]), 1.0)
    assert not ok


def test_gradient_verdict_rule():
    """tests/adjudicate.py::gradient_verdict, the three ways a (Gaussian, tensor) passes - inside the bound; no further from the
    exact value than the reference is (e_p <= 1 + SLACK e_r); inside what fp32 can resolve of a chain with condition number
    kappa (e_p <= 1 + kappa 2^-24 / 1e-3) - and the case that passes none of them."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adjudicate as adj
    t = np.array([2.0, -1.0])
    one = lambda rel: t * (1.0 + rel)                       # an evaluation `rel` off the exact value, relatively
    ok, e_p, e_r = adj.gradient_verdict(t, one(0.9e-3), t, one(5e-3), 0.0)
    assert ok and e_p < 1.0 < e_r                           # inside the bound whatever the reference does
    ok, e_p, e_r = adj.gradient_verdict(t, one(1.8e-3), t, one(0.45e-3), 0.0)
    assert ok and abs(e_p - 1.8) < 1e-6 and abs(e_r - 0.45) < 1e-6      # 1.8 <= 1 + 2 x 0.45
    ok, _, _ = adj.gradient_verdict(t, one(2.0e-3), t, one(0.45e-3), 0.0)
    assert not ok                                           # 2.0 > 1.9
    kappa = (271.0) ** 2                                    # an axis ratio of 1 : 271 -> 4.4 bounds of allowance
    ok, _, _ = adj.gradient_verdict(t, one(5.0e-3), t, one(0.1e-3), 0.0, kappa)
    assert ok
    ok, _, _ = adj.gradient_verdict(t, one(6.0e-3), t, one(0.1e-3), 0.0, kappa)
    assert not ok
    # the absolute term of the bound: 1e-5 of the tensor's scale
    ok, e_p, _ = adj.gradient_verdict(np.zeros(2), np.array([0.9e-5, 0.0]), np.zeros(2), np.zeros(2), 1.0)
    assert ok and abs(e_p - 0.9) < 1e-9
