"""Data-parallel step over the `nccl` backend (= RCCL) with the REAL op: rank r runs on `cuda:(r % device_count)`, so
on any box with two or more GPUs the two ranks are two devices talking over xGMI; on a one-GPU box both ranks land
on cuda:0, RCCL refuses ("Duplicate GPU detected") and the test skips with RCCL's message.
Each rank renders its own view; afterwards `leaf.grad` of every leaf must equal the sum of the two single-process
gradients, with and without the in-backward overlap of the feature all-reduce (dp.FeatureGradOverlap)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KEYS = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(view_id):
    from synth import make_scene
    return make_scene(P=20000, C=32, width=320, height=192, seed=5, yaw_deg=5.0 * view_id, scale_lo=0.005, scale_hi=0.08)


def _run_view(view_id, leaves, dev):
    import diff_gaussian_rasterization as dgr
    sc = _scene(view_id)
    t = lambda x: x.to(dev)
    st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                           t(sc["viewmatrix"]), t(sc["projmatrix"]), 3, t(sc["campos"]), False, False)
    color, feat, radii, depth = dgr.GaussianRasterizer(st)(means2D=torch.zeros(sc["P"], 3, device=dev), **leaves)
    torch.autograd.backward([color, feat], [t(sc["dL_dcolor"]), t(sc["dL_dfeature"])])
    return radii


def _leaves(dev):
    sc = _scene(0)
    return {k: sc[k].to(dev).clone().requires_grad_(True) for k in KEYS}


def _band_step(rank, world, dev):
    """ONE view (view 0) split over the ranks by tile rows: band render, bands gathered, the loss on the whole image, backward
    into the own rows, gradients summed (dp.band_rows / gather_bands; include/f3dgs.h: f3dgs_set_tile_band)."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    import dp
    sc = _scene(0)
    t = lambda x: x.to(dev)
    leaves = _leaves(dev)
    st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                           t(sc["viewmatrix"]), t(sc["projmatrix"]), 3, t(sc["campos"]), False, False)
    r0, r1, _y0, _y1 = dp.band_rows(sc["image_height"], rank, world)
    _C.set_tile_band(r0, r1)
    try:
        color, feat, radii, depth = dgr.GaussianRasterizer(st)(means2D=torch.zeros(sc["P"], 3, device=dev), **leaves)
    finally:
        _C.set_tile_band(0, 0)
    color, feat = dp.gather_bands(color), dp.gather_bands(feat)
    loss = (color * t(sc["dL_dcolor"])).sum() + (feat * t(sc["dL_dfeature"])).sum()
    loss.backward()
    grads = {k: leaves[k].grad for k in KEYS}
    dp.all_reduce_gaussian_grads(grads)
    rad = radii.float()
    import torch.distributed as dist
    dist.all_reduce(rad, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    return {**{f"band_{k}": grads[k].cpu().numpy() for k in KEYS}, "band_color": color.detach().cpu().numpy(),
            "band_feat": feat.detach().cpu().numpy(), "band_radii": rad.cpu().numpy(), "band_loss": float(loss)}


def _worker(rank, world, port, out_dir, backend="nccl"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device("cuda", rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    try:
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        probe = torch.ones(4, device=dev)
        dist.all_reduce(probe)          # RCCL builds that refuse two ranks on one device fail here
        torch.cuda.synchronize()
    except Exception as exc:            # noqa: BLE001
        open(os.path.join(out_dir, f"skip{rank}.txt"), "w").write(repr(exc))
        return
    import dp
    res = {}
    for overlap in (False, True):
        leaves = _leaves(dev)
        # with the overlap: feature gradient reduced from inside the blend stage, SH gradient in three row ranges from inside
        # the per-Gaussian stage (dp.RowsGradOverlap; "shs" is the op's direct input here)
        grads = dp.dp_step(lambda vid: _run_view(vid, leaves, dev), leaves, dp.views_for_rank(8, rank, world), overlap=overlap,
                           rows_leaves={"sh": ("shs",)} if overlap else None, rows_chunks=3)
        torch.cuda.synchronize()
        for k in KEYS:
            assert grads[k] is leaves[k].grad
            res[f"{'ov' if overlap else 'plain'}_{k}"] = leaves[k].grad.cpu().numpy()
    # densification statistics travel the same way
    acc = torch.full((100, 1), float(rank + 1), device=dev)
    den = torch.ones(100, 1, device=dev)
    rad = torch.arange(100, dtype=torch.float32, device=dev) * (rank + 1)
    dp.reduce_densification_stats(acc, den, rad)
    res.update(acc=acc.cpu().numpy(), den=den.cpu().numpy(), rad=rad.cpu().numpy())
    res.update(_band_step(rank, world, dev))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_two_ranks_rccl(tmp_path, backend):
    """backend "nccl": two ranks over RCCL (two devices where there are two; skipped where both ranks land on one).  backend
    "gloo": the SAME two-process step with the real op on whatever devices there are - on a one-GPU box two processes share
    cuda:0 and gloo carries the device tensors through the host: every collective call, side stream, in-backward hook and the
    band split run for real at world size 2 (only the transport is not RCCL's)."""
    world = 2
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)
    skips = [f for f in os.listdir(tmp_path) if f.startswith("skip")]
    if skips:
        if backend == "gloo":
            pytest.fail("the gloo process group failed: " + open(os.path.join(tmp_path, skips[0])).read()[:300])
        pytest.skip("RCCL refused two ranks on one device: " + open(os.path.join(tmp_path, skips[0])).read()[:300])
    dev = torch.device("cuda", 0)
    want = None
    for v in range(world):
        leaves = _leaves(dev)
        _run_view(v, leaves, dev)
        g = {k: leaves[k].grad.cpu().numpy().astype(np.float64) for k in KEYS}
        want = g if want is None else {k: want[k] + g[k] for k in KEYS}
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"r{rank}.npz"))
        for mode in ("plain", "ov"):
            for k in KEYS:
                w = want[k]
                err = np.abs(got[f"{mode}_{k}"] - w).max()
                assert err <= 1e-4 * np.abs(w).max() + 1e-12, (mode, k, rank, err)   # atomics reorder fp32 sums
        assert np.allclose(got["acc"], 3.0) and np.allclose(got["den"], 2.0) and np.allclose(got["rad"], np.arange(100) * 2.0)
    # ---- one view as two tile-row bands against the whole view in one process ---------------------------------------------
    import diff_gaussian_rasterization as dgr
    sc = _scene(0)
    t = lambda x: x.to(dev)
    leaves = _leaves(dev)
    st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                           t(sc["viewmatrix"]), t(sc["projmatrix"]), 3, t(sc["campos"]), False, False)
    color, feat, radii, depth = dgr.GaussianRasterizer(st)(means2D=torch.zeros(sc["P"], 3, device=dev), **leaves)
    loss = (color * t(sc["dL_dcolor"])).sum() + (feat * t(sc["dL_dfeature"])).sum()
    loss.backward()
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"r{rank}.npz"))
        assert np.array_equal(got["band_color"], color.detach().cpu().numpy()) and np.array_equal(got["band_feat"], feat.detach().cpu().numpy())
        assert np.array_equal(got["band_radii"], radii.float().cpu().numpy())
        assert abs(float(got["band_loss"]) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach())) + 1e-9
        for k in KEYS:
            w = leaves[k].grad.cpu().numpy()
            err = np.abs(got[f"band_{k}"] - w).max()
            assert err <= 1e-4 * np.abs(w).max() + 1e-12, ("band", k, rank, err)


def _solo_worker(rank, port, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import dp
    dp._active = lambda group: True          # take the collective code paths although the sum has one term
    res = {}
    for overlap in (False, True):
        leaves = _leaves(dev)
        dp.dp_step(lambda vid: _run_view(vid, leaves, dev), leaves, [0], overlap=overlap,
                   rows_leaves={"sh": ("shs",)} if overlap else None, rows_chunks=3)
        torch.cuda.synchronize()
        for k in KEYS:
            res[f"{'ov' if overlap else 'plain'}_{k}"] = leaves[k].grad.cpu().numpy()
    g = {k: torch.full((1000, 3), 2.0, device=dev) for k in ("a", "b")}
    shards = dp.reduce_scatter_gaussian_grads(g)
    full = {k: torch.zeros(1000, 3, device=dev) for k in g}
    dp.all_gather_params(shards, full)
    res["rs_ok"] = np.array([float(all(torch.equal(full[k], g[k]) for k in g))])
    # dp.ShardedOptimizer over FusedAdam through the same one-rank group (reduce-scatter, the HIP Adam kernel on "its" rows,
    # the hand-back of the updated rows) against FusedAdam on the full tensors: bit for bit
    from fused_adam import FusedAdam
    gen = torch.Generator(device=dev).manual_seed(3)
    pa = {k: torch.randn(1000, w, device=dev, generator=gen).requires_grad_(True) for k, w in (("a", 3), ("b", 48))}
    pb = {k: v.detach().clone().requires_grad_(True) for k, v in pa.items()}
    mk = lambda t: FusedAdam([{"params": [t[k]], "lr": 1e-3} for k in t], lr=0.0, eps=1e-15)
    oa, sh = mk(pa), dp.ShardedOptimizer(pb, mk)
    for _ in range(3):
        gr = {k: torch.randn(pa[k].shape, device=dev, generator=gen) for k in pa}
        for k in pa:
            pa[k].grad = gr[k].clone()
        oa.step()
        sh.step({k: gr[k].clone() for k in pb})
    res["sharded_ok"] = np.array([float(all(torch.equal(pa[k], pb[k]) for k in pa))])
    # the direct exchange (all-to-all + local sum + all-gather) through RCCL on the device: with one rank the "sum" is the tensor
    # itself - ragged length, a row slice, as a handle - and the whole overlapped step with F3DGS_DP_EXCHANGE = direct
    t1, t2 = torch.randn(1003, 7, device=dev, generator=gen), torch.randn(40, 9, device=dev, generator=gen)
    w1, w2 = t1.clone(), t2.clone()
    dp.all_reduce_direct(t1)
    h = dp.all_reduce_direct(t2[8:24], async_op=True)
    h.wait()
    torch.cuda.synchronize()
    ok = torch.equal(t1, w1) and torch.equal(t2, w2)
    dp.EXCHANGE = "direct"
    leaves = _leaves(dev)
    dp.dp_step(lambda vid: _run_view(vid, leaves, dev), leaves, [0], overlap=True, rows_leaves={"sh": ("shs",)}, rows_chunks=3)
    torch.cuda.synchronize()
    dp.EXCHANGE = "allreduce"
    for k in KEYS:
        w = res[f"plain_{k}"].astype(np.float64)
        ok = ok and bool(np.abs(leaves[k].grad.cpu().numpy() - w).max() <= 1e-4 * np.abs(w).max() + 1e-12)
    res["direct_ok"] = np.array([float(ok)])
    np.savez(os.path.join(out_dir, "solo.npz"), **res)
    dist.destroy_process_group()


def test_rccl_code_paths_on_one_rank(tmp_path):
    """RCCL rejects two ranks on one device ("Duplicate GPU detected"), and the test box has one GPU: run the SAME
    code paths - bucketed all-reduce, the in-backward overlap on a side stream with its event hand-over, reduce-scatter
    and all-gather - through a one-rank `nccl` group, where every collective really executes on the device and the
    result must equal the single-process gradients."""
    mp.spawn(_solo_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(tmp_path, "solo.npz"))
    dev = torch.device("cuda", 0)
    leaves = _leaves(dev)
    _run_view(0, leaves, dev)
    for mode in ("plain", "ov"):
        for k in KEYS:
            w = leaves[k].grad.cpu().numpy().astype(np.float64)
            assert np.abs(got[f"{mode}_{k}"] - w).max() <= 1e-4 * np.abs(w).max() + 1e-12, (mode, k)
    assert got["rs_ok"][0] == 1.0
    assert got["sharded_ok"][0] == 1.0
    assert got["direct_ok"][0] == 1.0


def _run_views_fb(leaves, dev):
    """(forward, backward) for dp.dp_step_views over the views of `_scene`."""
    import diff_gaussian_rasterization as dgr
    t = lambda x: x.to(dev)
    P = leaves["means3D"].shape[0]
    means2D = torch.zeros(P, 3, device=dev)

    def forward(view_id):
        sc = _scene(view_id)
        st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                               t(sc["viewmatrix"]), t(sc["projmatrix"]), 3, t(sc["campos"]), False, False)
        color, feat, radii, depth = dgr.GaussianRasterizer(st)(means2D=means2D, **leaves)
        return color, feat, t(sc["dL_dcolor"]), t(sc["dL_dfeature"])

    def backward(h):
        torch.autograd.backward([h[0], h[1]], [h[2], h[3]])
    return forward, backward


def _want_sum(views, dev):
    want = None
    for v in views:
        leaves = _leaves(dev)
        _run_view(v, leaves, dev)
        g = {k: leaves[k].grad.cpu().numpy().astype(np.float64) for k in KEYS}
        want = g if want is None else {k: want[k] + g[k] for k in KEYS}
    return want


@pytest.mark.parametrize("n_streams,accumulate", [(2, None), (1, None), (2, False)], ids=["pipelined+inplace", "one-stream+inplace", "pipelined+per-view-tensors"])
def test_several_views_per_rank_pipelined_and_accumulated(n_streams, accumulate):
    """dp.dp_step_views on one GPU, no process group: four views on two alternating streams with the feature gradient
    accumulated in place by the blend backward (set_feature_grad_accumulator) give the sum of the four single-view gradients;
    so do the same views on one stream, and the pipelined schedule with one gradient tensor per view."""
    import dp
    dev = torch.device("cuda", 0)
    views = [0, 1, 2, 3]
    want = _want_sum(views, dev)
    leaves = _leaves(dev)
    fwd, bwd = _run_views_fb(leaves, dev)
    grads = dp.dp_step_views(fwd, bwd, leaves, views, n_streams=n_streams, accumulate=accumulate)
    torch.cuda.synchronize()
    for k in KEYS:
        assert grads[k] is leaves[k].grad
        err = np.abs(leaves[k].grad.cpu().numpy() - want[k]).max()
        assert err <= 1e-4 * np.abs(want[k]).max() + 1e-12, (k, err)
    # the accumulator is released: a plain step afterwards hands the feature gradient to autograd again
    leaves2 = _leaves(dev)
    _run_view(0, leaves2, dev)
    assert leaves2["semantic_feature"].grad is not None and float(leaves2["semantic_feature"].grad.abs().max()) > 0


def test_in_place_accumulation_refuses_a_transformed_feature_input():
    """the in-place accumulation adds the op's feature gradient straight into `leaf.grad`, past the autograd chain.
    That is only right when the op's `semantic_feature` input IS the leaf: a model that feeds the op a transformed tensor of
    the same size (normalisation, mask) must get an error, not a gradient that skipped the chain."""
    import diff_gaussian_rasterization as dgr
    import dp
    dev = torch.device("cuda", 0)
    leaves = _leaves(dev)
    t = lambda x: x.to(dev)
    P = leaves["means3D"].shape[0]
    means2D = torch.zeros(P, 3, device=dev)

    def forward(view_id):
        sc = _scene(view_id)
        st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                               t(sc["viewmatrix"]), t(sc["projmatrix"]), 3, t(sc["campos"]), False, False)
        fed = dict(leaves)
        fed["semantic_feature"] = leaves["semantic_feature"] * 2.0            # same numel, NOT the leaf
        color, feat, radii, depth = dgr.GaussianRasterizer(st)(means2D=means2D, **fed)
        return color, feat, t(sc["dL_dcolor"]), t(sc["dL_dfeature"])

    def backward(h):
        torch.autograd.backward([h[0], h[1]], [h[2], h[3]])
    with pytest.raises(RuntimeError, match="not that leaf"):
        dp.dp_step_views(forward, backward, leaves, [0, 1], accumulate=True)
    torch.cuda.synchronize()
    # not ASKED to accumulate (the default of dp_train_step): the step notices at the first backward call that the op was fed
    # something else than the leaf and lets autograd carry every view's gradient through the model's chain
    grads = dp.dp_step_views(forward, backward, leaves, [0, 1], accumulate=None)
    torch.cuda.synchronize()
    assert dgr.accumulator_bypassed() == 0            # (reset when the step released the accumulator)
    want2 = _want_sum([0, 1], dev)["semantic_feature"]
    err = np.abs(grads["semantic_feature"].cpu().numpy() - 2.0 * want2).max()
    assert err <= 1e-4 * np.abs(want2).max() * 2.0, err
    # opt out: the gradient takes the autograd chain (d(2 f) = 2) and equals twice the direct one
    grads = dp.dp_step_views(forward, backward, leaves, [0, 1], accumulate=False)
    torch.cuda.synchronize()
    want = _want_sum([0, 1], dev)["semantic_feature"]
    err = np.abs(grads["semantic_feature"].cpu().numpy() - 2.0 * want).max()
    assert err <= 1e-4 * np.abs(want).max() * 2.0


def _solo_views_worker(rank, port, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import dp
    dp._active = lambda group: True          # take the collective code paths although the sum has one term
    leaves = _leaves(dev)
    fwd, bwd = _run_views_fb(leaves, dev)
    dp.dp_step_views(fwd, bwd, leaves, [0, 1, 2], overlap=True)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "solo_views.npz"), **{k: leaves[k].grad.cpu().numpy() for k in KEYS})
    dist.destroy_process_group()


def test_views_step_rccl_code_paths_on_one_rank(tmp_path):
    """The multi-view step through a one-rank `nccl` group: the all-reduce of the in-place accumulated feature gradient starts
    inside the LAST view's backward pass on the side stream (FeatureGradOverlap), the rest follows in the bucketed all-reduce."""
    mp.spawn(_solo_views_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(tmp_path, "solo_views.npz"))
    want = _want_sum([0, 1, 2], torch.device("cuda", 0))
    for k in KEYS:
        assert np.abs(got[k] - want[k]).max() <= 1e-4 * np.abs(want[k]).max() + 1e-12, k
