"""world_size-2 gloo test of the view-sharded data-parallel step (CPU): the rasterizer op is replaced by the
torch-autograd oracle on a tiny scene, the collectives and bucket packing are the product code (dp.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads_for_view(view_id, P=120, C=3):
    from oracle import torch_oracle
    from synth import make_scene
    sc = make_scene(P, C, 48, 32, seed=77, yaw_deg=5.0 * view_id, scale_lo=0.05, scale_hi=0.3)
    r = torch_oracle.forward_backward(sc, dtype=torch.float32)
    g = r["grads"]
    return {k: g[k].float() for k in ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")}, r


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "feature-3dgs_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dp
    vids = dp.views_for_rank(num_views=8, rank=rank, world=world, iteration=0)
    assert vids == [rank]
    grads, r = _grads_for_view(vids[0])
    summed = dp.all_reduce_gaussian_grads(grads, bucket_bytes=4096)   # tiny buckets: several collectives
    acc = torch.full((120, 1), float(rank + 1))
    den = torch.ones(120, 1)
    rad = torch.arange(120, dtype=torch.float32) * (rank + 1)
    dp.reduce_densification_stats(acc, den, rad)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), acc=acc.numpy(), den=den.numpy(), rad=rad.numpy(),
                 **{k: v.numpy() for k, v in summed.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "dp.npz"))
    want = None
    for v in range(world):
        g, _ = _grads_for_view(v)
        want = g if want is None else {k: want[k] + g[k] for k in g}
    for k, w in want.items():
        assert np.allclose(got[k], w.numpy(), rtol=1e-5, atol=1e-8), k
    assert np.allclose(got["acc"], 3.0) and np.allclose(got["den"], 2.0)
    assert np.allclose(got["rad"], np.arange(120) * 2.0)


def test_bucket_layout_roundtrip():
    import dp
    shapes = {"means3D": (10, 3), "shs": (10, 16, 3), "semantic_feature": (10, 1, 5), "opacities": (10, 1)}
    b = dp.GradBuckets(shapes, "cpu", bucket_bytes=600)
    assert len(b.buckets) > 1 and sum(x.numel() for x in b.buckets) == 10 * (3 + 48 + 5 + 1)
    g = {k: torch.randn(*s) for k, s in shapes.items()}
    b.pack(g)
    for k, v in b.unpack().items():
        assert torch.equal(v, g[k])


def test_view_sharding_covers_all_views():
    import dp
    seen = []
    for it in range(4):
        for r in range(2):
            seen += dp.views_for_rank(8, r, 2, it)
    assert sorted(seen) == list(range(8))
