"""distCUDA2 (simple_knn._C): CPU tests of the oracle itself, GPU tests of the HIP implementation against the oracle
and against the reference's own simple-knn compiled for gfx950 (oracle/_ref/_ref_simple_knn, oracle/build_ref.py)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from util import ROOT


def _points(P, seed, kind="uniform"):
    g = np.random.default_rng(seed)
    if kind == "uniform":
        return g.uniform(-3, 3, (P, 3)).astype(np.float32)
    if kind == "clustered":   # SfM-like: dense clumps + sparse background, far from the origin
        c = g.normal(0, 4, (max(1, P // 500), 3))
        x = c[g.integers(0, len(c), P)] + g.normal(0, 0.05, (P, 3)) + np.array([50.0, -20.0, 7.0])
        return x.astype(np.float32)
    if kind == "planar":
        x = g.uniform(-1, 1, (P, 3))
        x[:, 2] = 0.25
        return x.astype(np.float32)
    raise ValueError(kind)


def test_oracle_kdtree_equals_bruteforce():
    from oracle import knn_oracle
    for kind in ("uniform", "clustered", "planar"):
        p = _points(700, 3, kind)
        a, b = knn_oracle.mean_dist2_bruteforce(p), knn_oracle.mean_dist2_kdtree(p)
        assert np.allclose(a, b, rtol=2e-6, atol=0), kind


def test_oracle_small_and_degenerate():
    from oracle import knn_oracle
    big = np.finfo(np.float32).max
    one = knn_oracle.mean_dist2_bruteforce(np.zeros((1, 3), np.float32))
    assert one.shape == (1,) and (one[0] == np.inf or one[0] >= big / 3)          # three missing neighbours
    four = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]], np.float32)
    assert np.allclose(knn_oracle.mean_dist2_bruteforce(four)[0], (1 + 4 + 9) / 3)
    dup = np.zeros((5, 3), np.float32)
    assert np.all(knn_oracle.mean_dist2_bruteforce(dup) == 0)


def _product():
    from simple_knn._C import distCUDA2
    return distCUDA2


def _ref():
    p = os.path.join(ROOT, "oracle", "_ref", "_ref_simple_knn" + (__import__("sysconfig").get_config_var("EXT_SUFFIX") or ".so"))
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/_ref_simple_knn not built (python oracle/build_ref.py)")
    spec = importlib.util.spec_from_file_location("_ref_simple_knn", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.distCUDA2


@pytest.mark.gpu
@pytest.mark.parametrize("P,kind", [(1, "uniform"), (2, "uniform"), (3, "uniform"), (4, "uniform"), (63, "uniform"), (257, "planar"),
                                    (5000, "uniform"), (5000, "clustered"), (4097, "planar")])
def test_distcuda2_vs_bruteforce(P, kind):
    from oracle import knn_oracle
    p = _points(P, 11, kind)
    got = _product()(torch.from_numpy(p).cuda()).cpu().numpy()
    want = knn_oracle.mean_dist2_bruteforce(p)
    big = np.finfo(np.float32).max / 4
    fin = want < big
    assert np.array_equal(fin, got < big)            # missing neighbours (P < 4) stay "huge" like the reference's FLT_MAX
    assert np.allclose(got[fin], want[fin], rtol=2e-6, atol=0)


@pytest.mark.gpu
def test_distcuda2_duplicates_and_errors():
    f = _product()
    p = np.repeat(_points(300, 5), 4, axis=0)        # every point four times: all three neighbours at distance 0
    assert float(f(torch.from_numpy(p).cuda()).abs().max()) == 0.0
    assert f(torch.zeros(0, 3, device="cuda")).shape == (0,)
    with pytest.raises(Exception):
        f(torch.zeros(10, 3))                        # CPU tensor: no silent fallback
    with pytest.raises(Exception):
        f(torch.zeros(10, 4, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("P,kind", [(200_000, "uniform"), (200_000, "clustered"), (1_000_000, "clustered")])
def test_distcuda2_vs_reference_module(P, kind):
    """The reference's own kernels (hipified checker) on the same points; also the k-d tree oracle at 200k."""
    p = _points(P, 21, kind)
    t = torch.from_numpy(p).cuda()
    got = _product()(t).cpu().numpy()
    want = _ref()(t).cpu().numpy()
    assert np.allclose(got, want, rtol=2e-6, atol=0)
    if P <= 200_000:
        from oracle import knn_oracle
        assert np.allclose(got, knn_oracle.mean_dist2_kdtree(p), rtol=2e-6, atol=0)
    # the caller's use: initial scales = log(sqrt(clamp_min(dist2, 1e-7)))  (scene/gaussian_model.py:146-147)
    s = torch.log(torch.sqrt(torch.clamp_min(torch.from_numpy(got), 1e-7)))
    assert torch.isfinite(s).all()
