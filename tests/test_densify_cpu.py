"""CPU-side checks for the densification row (f-4): the numpy restatement's invariants (it is the checker of
tests/test_densify.py where oracle/_ref is absent) and the buffer pool's aliasing rules.  The product's plan + gather
run on the GPU only (tests/test_densify.py)."""
import importlib.util
import os
import sys

import numpy as np
import torch

from util import ROOT
from oracle import densify_oracle as orc


def _state(P, C, seed):
    r = np.random.default_rng(seed)
    f = lambda *s: r.standard_normal(s).astype(np.float32)
    st = {"xyz": f(P, 3), "f_dc": f(P, 1, 3), "f_rest": 0.1 * f(P, 15, 3), "opacity": 2 * f(P, 1) - 1, "scaling": 0.7 * f(P, 3) - 3,
          "rotation": f(P, 4), "semantic_feature": f(P, 1, C)}
    for n in list(st):
        st[n + ".exp_avg"], st[n + ".exp_avg_sq"] = 0.01 * f(*st[n].shape), np.abs(0.01 * f(*st[n].shape))
    st["xyz_gradient_accum"] = (r.random((P, 1)) * 1.2e-3).astype(np.float32)
    st["denom"] = r.integers(0, 4, (P, 1)).astype(np.float32)
    st["max_radii2D"] = (r.random(P) * 40).astype(np.float32)
    return st


def test_numpy_restatement_invariants():
    P, C, N = 4000, 8, 2
    st = _state(P, C, 3)
    before = {k: v.copy() for k, v in st.items()}
    drawn = {}

    def normal(std):
        drawn["n"] = std.shape[0]
        return np.zeros_like(std)            # children sit exactly on their parent

    out = orc.densify_and_prune(st, 0.0002, 0.005, 5.0, 20, 0.01, normal, N)
    assert drawn["n"] == N * out["split"]
    n = st["xyz"].shape[0]
    assert n == out["points"] == P + out["cloned"] + (N - 1) * out["split"] - out["pruned"]
    assert out["cloned"] > 0 and out["split"] > 0 and out["pruned"] > 0
    for k, v in st.items():
        assert v.shape[0] == n, k
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert not st[k].any()
    # survivors keep their order: the kept originals come first and are a subsequence of the input
    opac = st["opacity"].reshape(-1)
    assert (1 / (1 + np.exp(-opac)) >= 0.005).all()
    kept = np.flatnonzero((st["f_dc.exp_avg"].reshape(n, -1) != 0).any(axis=1))      # rows with carried moments = originals
    assert (np.diff(kept) == 1).all() and kept[0] == 0                                # ... and they form the head of the result
    src = [int(np.flatnonzero((before["f_dc"] == st["f_dc"][i]).all(axis=(1, 2)))[0]) for i in kept[:200]]
    assert src == sorted(src)
    # new rows start with zero moments; a child with a zero sample keeps the parent's position and has scale / (0.8 N)
    new = np.setdiff1d(np.arange(n), kept)
    for name in orc.PARAMS:
        assert not st[name + ".exp_avg"][new].any() and not st[name + ".exp_avg_sq"][new].any()
    tail = st["scaling"][-1]
    parent = np.flatnonzero(np.isclose(before["xyz"], st["xyz"][-1], atol=1e-6).all(axis=1))
    assert parent.size >= 1
    assert np.allclose(np.exp(tail), np.exp(before["scaling"][parent[0]]) / (0.8 * N), rtol=1e-5)


def test_row_pool_never_hands_out_the_buffer_in_use():
    sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
    spec = importlib.util.spec_from_file_location("densify_pool_only", os.path.join(ROOT, "feature-3dgs_amd", "densify.py"))
    src = open(spec.origin).read().replace("from diff_gaussian_rasterization import _C", "_C = None")   # no GPU extension here
    mod = type(sys)("densify_pool_only")
    exec(compile(src, spec.origin, "exec"), mod.__dict__)
    pool = mod.RowPool(growth=1.5, min_rows=4)
    x = torch.zeros(10, 3)
    a = pool.out("xyz", 12, x)
    assert a.shape == (19, 3) and a.untyped_storage().data_ptr() != x.untyped_storage().data_ptr()
    b = pool.out("xyz", 14, a[:12])                      # the model now lives in `a`: must get the other buffer
    assert b.untyped_storage().data_ptr() != a.untyped_storage().data_ptr()
    c = pool.out("xyz", 13, b[:14])                      # and back: `a` is reused, no allocation
    assert c.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() and pool.reallocations == 2
    d = pool.out("xyz", 40, c[:13])                      # outgrown: the spare buffer is replaced
    assert d.shape[0] >= 40 and pool.reallocations == 3 and d.untyped_storage().data_ptr() != c.untyped_storage().data_ptr()
    assert pool.capacity("xyz") == 19
