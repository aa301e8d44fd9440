#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native Feature-3DGS rasterizer.

Metric (BASELINE.json): train-step ms + rendered Mpix/s of one rasterizer forward+backward per view,
1M synthetic Gaussians @1920x1080, SH degree 3, feat_dim=32 (config "c3"), inputs resident in HBM.
A "step" is one forward + one backward of the op over one view per GPU (everything inside the op:
buffer sizing, the 4-byte num_rendered read-back, output/gradient allocation; no loss, no optimiser).
With N > 1 GPUs every rank renders its own view of the same Gaussians (view r is rotated r*5 degrees)
and the per-Gaussian gradients ((59+C) floats each) are summed with a RCCL all-reduce inside the step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel: algorithmic bytes per launch
(restated in DESIGN.md) over its mean duration measured with HIP events on the
op's stream inside the timed region.  `cpu_baseline` is the scalar CPU oracle (a port, 1 core) timed
on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("F3DGS_PROFILE", "1")          # event spans around every stage (no syncs inside the op)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(P, Pv, N, N_r, HW, tiles, C, M=16):
    """Per-stage ALGORITHMIC bytes of one forward+backward (k = 6 radix passes of
    the reference's 64-bit key sort for all BASELINE configs)."""
    k = 6
    a = {
        "preprocess": 44 * P + (12 * M + 67) * Pv + 8 * P,
        "binning": 8 * P + (8 * P + 12 * Pv + 12 * N) + (24 * k + 8) * N + (8 * N + 8 * tiles),
        "render_fwd": (44 + 4 * C) * N_r + (4 * (C + 4) + 8) * HW,
        "render_bwd": 44 * N_r + (4 * (C + 4) + 8) * HW + (40 + 4 * C) * N_r,
        "preprocess_bwd": (179 + 24 * M) * Pv,
    }
    a["total"] = sum(a.values())
    return a


def scene_stats(scene, dev):
    """One untimed forward through _C to obtain Pv, N and N_r (= sum over tiles of the deepest list
    position any pixel of the tile blends, from the n_contrib plane)."""
    from diff_gaussian_rasterization import _C
    t = lambda x: x.to(dev)
    e = torch.Tensor([])
    res = _C.rasterize_gaussians(
        t(scene["bg"]), t(scene["means3D"]), e, t(scene["semantic_feature"]), t(scene["opacities"]), t(scene["scales"]),
        t(scene["rotations"]), scene["scale_modifier"], e, t(scene["viewmatrix"]), t(scene["projmatrix"]),
        scene["tanfovx"], scene["tanfovy"], scene["image_height"], scene["image_width"], t(scene["shs"]),
        scene["sh_degree"], t(scene["campos"]), False, False)
    torch.cuda.synchronize()
    n, _, _, _, radii, geom, binning, img = res
    W, H = scene["image_width"], scene["image_height"]
    lib = ctypes.CDLL(os.path.join(ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so"))
    lib.f3dgs_debug_read.restype = ctypes.c_int
    lib.f3dgs_debug_read.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    nc = np.zeros(W * H, np.uint32)
    rc = lib.f3dgs_debug_read(b"n_contrib", scene["P"], scene["C"], n, W, H, geom.data_ptr(),
                              binning.data_ptr() if binning.numel() else None, img.data_ptr(),
                              nc.ctypes.data_as(ctypes.c_void_p), nc.nbytes, None)
    assert rc == 0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), np.uint32)
    pad[:H, :W] = nc.reshape(H, W)
    N_r = int(pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).astype(np.int64).sum())
    return dict(Pv=int((radii > 0).sum()), N=int(n), N_r=N_r, tiles=gx * gy)


def cpu_baseline(cfg_kw):
    """Scalar C++ oracle (1 core) on a bounded sample of the same workload: same image size and feature
    dim, one fifth of the Gaussians."""
    from oracle.oracle import Oracle, scene_kwargs
    from synth import make_scene
    kw = dict(cfg_kw)
    kw["P"] = max(1000, kw["P"] // 5)
    sc = make_scene(seed=0, **kw)
    o = Oracle()
    t0 = time.perf_counter()
    o.forward(**scene_kwargs(sc))
    o.backward(sc["dL_dcolor"], sc["dL_dfeature"], sc["dL_ddepth"])
    dt = time.perf_counter() - t0
    mpix = kw["width"] * kw["height"] / 1e6
    return {"value": mpix / dt, "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": f"1 fwd+bwd of the scalar C++ oracle on {kw['P']} Gaussians @{kw['width']}x{kw['height']}, "
                      f"feat_dim={kw['C']} (the workload with 1/5 of the Gaussians), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--feat-dim", type=int, default=None, help="override the config's feature dim (development)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rasterizer has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import diff_gaussian_rasterization as dgr
    import dp
    from diff_gaussian_rasterization import _C
    from synth import CONFIGS, make_scene

    cfg_kw = dict(CONFIGS[args.config])
    if args.feat_dim is not None:
        cfg_kw["C"] = args.feat_dim
    scene = make_scene(seed=0, yaw_deg=5.0 * rank, **cfg_kw)
    P, C = scene["P"], scene["C"]
    W, H = scene["image_width"], scene["image_height"]
    t = lambda x: x.to(dev)
    settings = dgr.GaussianRasterizationSettings(H, W, scene["tanfovx"], scene["tanfovy"], t(scene["bg"]), 1.0,
                                                 t(scene["viewmatrix"]), t(scene["projmatrix"]), scene["sh_degree"],
                                                 t(scene["campos"]), False, False)
    rasterizer = dgr.GaussianRasterizer(settings)
    leaves = dict(means3D=t(scene["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                  opacities=t(scene["opacities"]).requires_grad_(), shs=t(scene["shs"]).requires_grad_(),
                  semantic_feature=t(scene["semantic_feature"]).requires_grad_(),
                  scales=t(scene["scales"]).requires_grad_(), rotations=t(scene["rotations"]).requires_grad_())
    up = [t(scene["dL_dcolor"]), t(scene["dL_dfeature"]), t(scene["dL_ddepth"])]
    reduce_keys = ["means3D", "shs", "semantic_feature", "opacities", "scales", "rotations"]   # 59 + C floats

    def step():
        for v in leaves.values():
            v.grad = None
        color, feat, _radii, depth = rasterizer(**leaves)
        torch.autograd.backward([color, feat, depth], up)
        if dist is not None:
            dp.all_reduce_gaussian_grads({k: leaves[k].grad for k in reduce_keys})

    stats = None
    if rank == 0:
        # N_r of SURVEY.md 8(d) is defined on the reference's (bounding-rectangle) instance lists
        _C.set_option("tile_cull", 0)
        stats = scene_stats(scene, dev)
        _C.set_option("tile_cull", 1)
    for _ in range(args.warmup):
        step()
    _C.profile_reset()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof = {name: (ms, calls) for name, ms, calls in _C.profile_read()}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        mpix = world * W * H / 1e6 / (elapsed / args.steps)
        stage_ms = {k: v[0] / max(1, args.steps) for k, v in prof.items()}
        alg = algorithmic_bytes(P, stats["Pv"], stats["N"], stats["N_r"], W * H, stats["tiles"], C)
        kernel_stage = {"preprocess": "preprocess", "render_fwd": "render_fwd", "render_bwd": "render_bwd",
                        "preprocess_bwd": "preprocess_bwd"}
        dom = max(kernel_stage, key=lambda k: stage_ms.get(kernel_stage[k], 0.0))
        dom_ms = stage_ms.get(kernel_stage[dom], float("nan"))
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file):
            try:
                traffic = json.load(open(pmc_file)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "rendered Mpix/s of rasterizer fwd+bwd (train-step ms in ms_per_step), 1M Gaussians @1080p, feat_dim=32",
            "value": mpix, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, {W}x{H}, SH degree {scene['sh_degree']}, feat_dim={C}, "
                                   f"one view per GPU (SURVEY.md 8d recipe, seed 0)",
                       "P": P, "Pv": stats["Pv"], "N": stats["N"], "N_r": stats["N_r"],
                       "parallelism": "single GPU" if world == 1 else
                       f"view-sharded dp{world} + RCCL all-reduce of (59+C) floats per Gaussian"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": dom_ms},
            "roofline_whole_step": {"algorithmic_bytes": alg["total"],
                                    "achieved": alg["total"] / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                                    "frac": alg["total"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stage_ms": stage_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg_kw)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
