"""f-1: the data-parallel training-step wrapper (feature-3dgs_amd/train_step.py).

GPU, world size 1: the reference's own `render()` (bytecode, see test_gpu_dropin.py) + an L1 loss through
`dp_train_step` must leave EXACTLY the gradients and densification statistics of the reference's loop body
(train.py:91-133) written out by hand.  CPU, world size 2 (gloo): gradients are summed, statistics reduced with
SUM / SUM / MAX, replicas end identical."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantic_feature")


class StubModel:
    """The part of the reference's GaussianModel the loop body touches (scene/gaussian_model.py:41-125, 436-438):
    raw parameters + activations + densification statistics."""

    def __init__(self, sc, dev):
        leaf = lambda x: x.to(dev).clone().requires_grad_(True)
        self.active_sh_degree, self.max_sh_degree = sc["sh_degree"], 3
        self._xyz = leaf(sc["means3D"])
        self._features_dc = leaf(sc["shs"][:, :1, :])
        self._features_rest = leaf(sc["shs"][:, 1:, :])
        self._opacity = leaf(torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)))
        self._scaling = leaf(torch.log(sc["scales"]))
        self._rotation = leaf(sc["rotations"] * 1.7)          # un-normalised, like a trained model
        self._semantic_feature = leaf(sc["semantic_feature"])
        P = sc["P"]
        self.max_radii2D = torch.zeros(P, device=dev)
        self.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        self.denom = torch.zeros(P, 1, device=dev)
    get_xyz = property(lambda s: s._xyz)
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_semantic_feature = property(lambda s: s._semantic_feature)


def _scene(view=0, P=5000, C=16, W=160, H=96):
    from synth import make_scene
    return make_scene(P=P, C=C, width=W, height=H, seed=51, yaw_deg=4.0 * view, scale_lo=0.005, scale_hi=0.08)


def _camera(sc, dev):
    import math
    c = types.SimpleNamespace()
    c.FoVx, c.FoVy = 2 * math.atan(sc["tanfovx"]), 2 * math.atan(sc["tanfovy"])
    c.image_height, c.image_width = sc["image_height"], sc["image_width"]
    c.world_view_transform, c.full_proj_transform = sc["viewmatrix"].to(dev), sc["projmatrix"].to(dev)
    c.camera_center = sc["campos"].to(dev)
    g = torch.Generator().manual_seed(7)
    c.original_image = torch.rand(3, c.image_height, c.image_width, generator=g).to(dev)
    c.semantic_feature = torch.randn(sc["C"], c.image_height, c.image_width, generator=g).to(dev)
    return c


def _loss(pkg, cam):   # train.py:98-106 without the ssim term (not on this path)
    return (pkg["render"] - cam.original_image).abs().mean() + (pkg["feature_map"] - cam.semantic_feature).abs().mean()


from test_gpu_dropin import reference_render  # noqa: E402,F401  (fixture: the reference's render() as bytecode)


@pytest.mark.gpu
def test_world_size_one_keeps_the_reference_loop_semantics(reference_render):
    import train_step
    render = reference_render
    dev = torch.device("cuda", 0)
    sc = _scene()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = sc["bg"].to(dev)
    cam = _camera(sc, dev)
    # ---- the reference's loop body, by hand --------------------------------------------------------------
    ref = StubModel(sc, dev)
    pkg = render(cam, ref, pipe, bg)
    _loss(pkg, cam).backward()
    vis, radii, vsp = pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"]
    ref.max_radii2D[vis] = torch.max(ref.max_radii2D[vis], radii[vis])
    ref.xyz_gradient_accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
    ref.denom[vis] += 1
    # ---- the wrapper ---------------------------------------------------------------------------------------
    mine = StubModel(sc, dev)
    res = train_step.dp_train_step(render, _loss, mine, [cam], pipe, bg)
    assert res.views == 1 and set(res.grads) == set(NAMES)
    for n in NAMES:
        a, b = getattr(ref, n).grad, getattr(mine, n).grad
        assert res.grads[n] is b
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12, n     # atomics reorder fp32 sums
    assert torch.equal(ref.max_radii2D, mine.max_radii2D) and torch.equal(ref.denom, mine.denom)
    assert float((ref.xyz_gradient_accum - mine.xyz_gradient_accum).abs().max()) <= 1e-5 * float(ref.xyz_gradient_accum.max())
    assert float(mine.denom.sum()) == float(vis.sum()) > 0


@pytest.mark.gpu
def test_band_step_emulated_rank_by_rank_adds_up_to_the_whole_view(reference_render):
    """`dp_train_step_bands` with the reference's own `render()`: at world size 1 it IS the whole-view step; and the step of a
    three-rank split, emulated rank by rank on this GPU through the op's band (the gather replaced by the whole view's images,
    which is what every rank would hold), leaves gradients that add up to the whole view's."""
    import dp
    import train_step
    from diff_gaussian_rasterization import _C
    render = reference_render
    dev = torch.device("cuda", 0)
    sc = _scene(W=160, H=112)          # seven tile rows: bands of 3 + 2 + 2
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = sc["bg"].to(dev)
    cam = _camera(sc, dev)
    whole = StubModel(sc, dev)
    res = train_step.dp_train_step_bands(render, _loss, whole, cam, pipe, bg)
    ref = StubModel(sc, dev)
    train_step.dp_train_step(render, _loss, ref, [cam], pipe, bg)
    for n in NAMES:
        a, b = getattr(ref, n).grad, getattr(whole, n).grad
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12, n
    assert torch.equal(ref.max_radii2D, whole.max_radii2D) and torch.equal(ref.denom, whole.denom)
    with torch.no_grad():
        full_pkg = render(cam, whole, pipe, bg)
        full = {k: full_pkg[k].detach().clone() for k in ("render", "feature_map", "depth")}
    sums = {n: torch.zeros_like(getattr(whole, n)) for n in NAMES}
    radii = torch.zeros(sc["P"], device=dev)
    world = 3
    try:
        for r in range(world):
            m = StubModel(sc, dev)
            r0, r1, y0, y1 = dp.band_rows(cam.image_height, r, world)
            _C.set_tile_band(r0, r1)
            pkg = render(cam, m, pipe, bg)
            _C.set_tile_band(0, 0)
            stitched = dict(pkg)
            for k in full:        # what dp.gather_bands hands the loss: the other ranks' rows as constants
                stitched[k] = torch.cat([full[k][..., :y0, :], pkg[k][..., y0:y1, :], full[k][..., y1:, :]], dim=-2)
                assert torch.equal(stitched[k], full[k]), (k, r)
            _loss(stitched, cam).backward()
            for n in NAMES:
                sums[n] += getattr(m, n).grad
            radii = torch.maximum(radii, pkg["radii"].float())
    finally:
        _C.set_tile_band(0, 0)
    for n in NAMES:
        a = getattr(ref, n).grad
        assert float((a - sums[n]).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, n
    assert torch.equal(radii, full_pkg["radii"].float())


# ---------------------------------------------------------------- CPU, two ranks ---------------------------
def _fake_render(cam, model, pipe, bg):
    """A differentiable stand-in for the op on CPU: enough structure for gradients, radii and visibility."""
    xyz = model.get_xyz
    vsp = torch.zeros_like(xyz, requires_grad=True) + 0
    vsp.retain_grad()
    w = torch.sigmoid(model._opacity) * torch.exp(model._scaling).sum(1, keepdim=True)
    img = ((xyz + vsp) * w * cam.k).sum(0).reshape(3, 1, 1).expand(3, 2, 2) + model.get_features.sum() * 1e-3
    feat = (model.get_semantic_feature.squeeze(1) * w).sum(0).reshape(-1, 1, 1).expand(-1, 2, 2) * cam.k
    radii = ((torch.arange(xyz.shape[0]) * 7 + cam.k) % 5).int()
    return {"render": img, "viewspace_points": vsp, "visibility_filter": radii > 0, "radii": radii, "feature_map": feat,
            "depth": img[:1]}


def _fake_loss(pkg, cam):
    return pkg["render"].sum() * 0.5 + pkg["feature_map"].sum() * 0.25 + (pkg["render"].mean() * model_rot_term(cam))


def model_rot_term(cam):
    return 1.0


def _worker(rank, world, port, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import train_step
    sc = _scene(P=40, C=3, W=32, H=32)
    model = StubModel(sc, "cpu")
    cam = types.SimpleNamespace(k=float(rank + 2))
    res = train_step.dp_train_step(_fake_render, _fake_loss, model, [cam], None, None, overlap=False)
    np.savez(os.path.join(out_dir, f"t{rank}.npz"), acc=model.xyz_gradient_accum.numpy(), den=model.denom.numpy(),
             rad=model.max_radii2D.numpy(), loss=res.loss.numpy(),
             **{n: getattr(model, n).grad.numpy() for n in NAMES if getattr(model, n).grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_gradients_and_statistics(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(os.path.join(tmp_path, f"t{r}.npz")) for r in range(2)]
    # single-process expectation: both views accumulated into one model
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "feature-3dgs_amd"))
    import train_step
    sc = _scene(P=40, C=3, W=32, H=32)
    model = StubModel(sc, "cpu")
    cams = [types.SimpleNamespace(k=2.0), types.SimpleNamespace(k=3.0)]
    train_step.dp_train_step(_fake_render, _fake_loss, model, cams, None, None, overlap=False)
    for r in range(2):
        for n in NAMES:
            g = getattr(model, n).grad
            if g is not None:
                assert np.allclose(got[r][n], g.numpy(), rtol=1e-5, atol=1e-7), (n, r)
        assert np.allclose(got[r]["acc"], model.xyz_gradient_accum.numpy(), rtol=1e-5)
        assert np.array_equal(got[r]["den"], model.denom.numpy()) and np.array_equal(got[r]["rad"], model.max_radii2D.numpy())
    assert got[0]["den"].max() == 2.0          # a Gaussian visible in both views counts twice


def _sharded_loop_worker(rank, world, port, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dp
    import train_step
    lrs = dict(_xyz=1e-3, _features_dc=2e-3, _features_rest=1e-4, _opacity=5e-2, _scaling=5e-3, _rotation=1e-3, _semantic_feature=1e-3)

    def make_opt(t):
        return torch.optim.Adam([{"params": [t[n]], "lr": lrs[n], "name": n} for n in NAMES], lr=0.0, eps=1e-15)
    sc = _scene(P=41, C=3, W=32, H=32)                   # 41 rows on two ranks: a ragged sharding
    cam = types.SimpleNamespace(k=float(rank + 2))
    # (A) the reference's arrangement made data-parallel: all-reduce of the gradients, the full optimizer on every rank
    ma = StubModel(sc, "cpu")
    oa = make_opt({n: getattr(ma, n) for n in NAMES})
    # (B) local gradients + sharded optimizer
    mb = StubModel(sc, "cpu")
    sh = dp.ShardedOptimizer({n: getattr(mb, n) for n in NAMES}, make_opt)
    for _ in range(3):
        train_step.dp_train_step(_fake_render, _fake_loss, ma, [cam], None, None, overlap=False)
        oa.step()
        res = train_step.dp_train_step(_fake_render, _fake_loss, mb, [cam], None, None, overlap=False, reduce=False)
        sh.step(res.grads)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), acc_a=ma.xyz_gradient_accum.numpy(), acc_b=mb.xyz_gradient_accum.numpy(),
             **{"a" + n: getattr(ma, n).detach().numpy() for n in NAMES}, **{"b" + n: getattr(mb, n).detach().numpy() for n in NAMES})
    dist.barrier()
    dist.destroy_process_group()


def test_training_loop_with_the_sharded_optimizer_equals_the_all_reduce_loop(tmp_path):
    """Three iterations of the loop body on two ranks (gloo), one view each: `dp_train_step` + the full Adam on every rank
    against `dp_train_step(reduce=False)` + `dp.ShardedOptimizer` (reduce-scatter, Adam on this rank's rows, all-gather of the
    parameters).  The parameters agree BIT FOR BIT on both ranks after every parameter has been updated three times, and the
    densification statistics - reduced in both arrangements - are the same."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = (np.load(os.path.join(tmp_path, f"s{r}.npz")) for r in range(2))
    for n in NAMES:
        assert np.array_equal(g0["a" + n], g0["b" + n]), n
        assert np.array_equal(g1["a" + n], g1["b" + n]), n
        assert np.array_equal(g0["b" + n], g1["b" + n]), n
    assert np.array_equal(g0["acc_a"], g0["acc_b"]) and np.array_equal(g0["acc_b"], g1["acc_b"])


# ---------------------------------------------------------------- one view split over two ranks by tile rows ------------
_BAND = [0, 0]


def _set_band(r0, r1):
    _BAND[:] = [r0, r1]


def _banded_render(cam, model, pipe, bg):
    """A differentiable stand-in for the op that honours a tile-row band the way the op does (include/f3dgs.h:
    f3dgs_set_tile_band): inside the band the whole view's pixels, outside it constants; a Gaussian whose footprint misses
    the band has radius 0."""
    H, W = cam.image_height, cam.image_width
    xyz = model.get_xyz
    vsp = torch.zeros_like(xyz, requires_grad=True) + 0
    vsp.retain_grad()
    P = xyz.shape[0]
    ys, xs = torch.arange(H).float().view(1, H, 1), torch.arange(W).float().view(1, 1, W)
    cy = ((xyz[:, 1] + vsp[:, 1]) * 6.0 + H / 2).view(P, 1, 1)
    cx = ((xyz[:, 0] + vsp[:, 0]) * 6.0 + W / 2).view(P, 1, 1)
    wgt = torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 40.0) * torch.sigmoid(model._opacity).view(P, 1, 1)      # (P, H, W)
    rgb = model.get_features[:, 0, :]                                                                              # (P, 3)
    img = (wgt.unsqueeze(1) * rgb.view(P, 3, 1, 1)).sum(0)
    feat = (wgt.unsqueeze(1) * model.get_semantic_feature.squeeze(1).view(P, -1, 1, 1)).sum(0)
    r0, r1 = _BAND
    if (r0, r1) == (0, 0):
        y0, y1 = 0, H
    else:
        y0, y1 = min(H, 16 * r0), min(H, 16 * max(r0, r1))
    rows = torch.zeros(H, dtype=torch.bool)
    rows[y0:y1] = True
    m = rows.view(1, H, 1)
    img = torch.where(m, img, torch.full_like(img, 0.25).detach())
    feat = torch.where(m, feat, torch.zeros_like(feat))
    radii = ((wgt.detach()[:, rows, :].reshape(P, -1).max(dim=1).values > 1e-2).int() * 3) if y1 > y0 else torch.zeros(P, dtype=torch.int32)
    return {"render": img, "viewspace_points": vsp, "visibility_filter": radii > 0, "radii": radii, "feature_map": feat, "depth": img[:1]}


def _window_loss(pkg, cam):      # an L1 term and a 5 x 5 window term that straddles the band border
    img = pkg["render"]
    return (img - cam.gt).abs().mean() + (torch.nn.functional.avg_pool2d(img.unsqueeze(0), 5, stride=1) ** 2).mean() \
        + (pkg["feature_map"] ** 2).mean()


def _band_step_worker(rank, world, port, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import train_step
    sc = _scene(P=30, C=3, W=24, H=40)
    model = StubModel(sc, "cpu")
    cam = types.SimpleNamespace(image_height=40, image_width=24, gt=torch.rand(3, 40, 24, generator=torch.Generator().manual_seed(3)))
    res = train_step.dp_train_step_bands(_banded_render, _window_loss, model, cam, None, None, set_band=_set_band)
    assert _BAND == [0, 0], "the band is lifted after the render"
    np.savez(os.path.join(out_dir, f"b{rank}.npz"), acc=model.xyz_gradient_accum.numpy(), den=model.denom.numpy(),
             rad=model.max_radii2D.numpy(), loss=res.loss.numpy(),
             **{n: getattr(model, n).grad.numpy() for n in NAMES if getattr(model, n).grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_one_view_split_over_two_ranks_by_tile_rows(tmp_path):
    """`dp_train_step_bands` on two ranks (gloo; bands of two and one tile rows of a 40-row image) against the whole view in one
    process: the same loss on every rank, the same gradients (the window term's too), the statistics of ONE view."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_band_step_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(os.path.join(tmp_path, f"b{r}.npz")) for r in range(2)]
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "feature-3dgs_amd"))
    import train_step
    sc = _scene(P=30, C=3, W=24, H=40)
    model = StubModel(sc, "cpu")
    cam = types.SimpleNamespace(image_height=40, image_width=24, gt=torch.rand(3, 40, 24, generator=torch.Generator().manual_seed(3)))
    _set_band(0, 0)
    res = train_step.dp_train_step_bands(_banded_render, _window_loss, model, cam, None, None, set_band=_set_band)      # world size 1: the whole view
    assert float(model.denom.sum()) > 0
    for r in range(2):
        assert np.allclose(got[r]["loss"], res.loss.numpy(), rtol=1e-6)
        for n in NAMES:
            g = getattr(model, n).grad
            if g is not None and n in got[r].files:
                assert np.allclose(got[r][n], g.numpy(), rtol=1e-4, atol=1e-8), (n, r)
        assert np.allclose(got[r]["acc"], model.xyz_gradient_accum.numpy(), rtol=1e-4, atol=1e-9)
        assert np.array_equal(got[r]["den"], model.denom.numpy()) and np.array_equal(got[r]["rad"], model.max_radii2D.numpy())
    assert got[0]["den"].max() == 1.0          # ONE view, however many bands saw the Gaussian
