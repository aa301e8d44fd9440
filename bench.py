#!/usr/bin/env python
"""bench.py - headline benchmark of the MI355X-native Feature-3DGS rasterizer.

Metric (BASELINE.json): train-step ms + rendered Mpix/s of one rasterizer forward+backward per view,
1M synthetic Gaussians @1920x1080, SH degree 3, feat_dim=32 (config "c3"), inputs resident in HBM.
A "step" is one forward + one backward of the op over one view per GPU (everything inside the op:
buffer sizing, the 8-byte instance-count read-back, output/gradient allocation; no loss, no optimiser),
with a FRESH set of upstream gradients every step (a rotating pool generated before the timed region).
With N > 1 GPUs every rank renders its own view of the same Gaussians (view r is rotated r*5 degrees)
and the per-Gaussian gradients ((59+C) floats each) are summed over RCCL inside the step; the feature
gradient's all-reduce starts inside the backward pass (dp.FeatureGradOverlap), the SH gradient's inside its last
stage, row chunk by row chunk (dp.RowsGradOverlap).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1..c5] [--no-cpu-baseline] [--comm-only]

`--gpus N` with N > 1 needs one process per GPU: started under torch.distributed.run it uses the ranks it is
given; started plainly it re-launches itself under torch.distributed.run (same arguments) and relays that
run's JSON line - it never reports a number measured on fewer GPUs than asked for.

Rank 0 prints ONE JSON line.  What is measured where:
  * `value` / `ms_per_step`: wall clock of exactly K steps between barrier + synchronize pairs, max over ranks
    (the contract); `step_ms` adds the per-step distribution (median, p10, p90) from HIP events on the op's stream,
    taken in the auxiliary run described below.
  * `roofline`: the dominant kernel's ALGORITHMIC bytes per launch (restated in DESIGN.md)
    over its mean duration, measured live with HIP events recorded by the library around the two blend kernels
    on the stream they run on, inside the timed region (library option profile = 2: four events per step).  The
    full stage split (`stage_ms`) comes from an auxiliary run of the same K steps with an event at every stage
    boundary (eleven per step, ~4 us of device time each) - it is not part of `value`.  `traffic` is null: HBM
    counters cannot be read from inside this process; the separately profiled figure is quoted under `profiled`.
  * `cpu_baseline`: the scalar C++ oracle (a port, 1 core) on a bounded sample of the same workload;
    `cpu_reference_path_c1`: BASELINE.json config 1 - the PyTorch-CPU autograd restatement on all host cores
    next to the product on the GPU, both at c1 (SURVEY.md 8(d) "CPU reference timing").
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# F3DGS_BENCH_STUB=1 (tests/test_bench_launch.py only): the whole multi-rank control flow of this file - re-launch under
# torch.distributed.run, rank relay, the exchange, every diagnostic leg, the one JSON line - on CPU tensors over gloo with the op
# replaced by a few torch operations.  The line says "data": "stub" and carries no measurement; it exists so that the first real
# multi-GPU lease (driver-run, not debuggable) does not die in plumbing.
STUB = os.environ.get("F3DGS_BENCH_STUB", "0") == "1"
# rehearsal of the multi-rank flow with the real op on ONE GPU (all ranks on cuda:0, gloo as transport): not a measurement
ONE_DEVICE = os.environ.get("F3DGS_BENCH_ONE_DEVICE", "0") == "1"

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_ACHIEVABLE_GBS = 6290.0   # MI355X_MICROARCH.md "Chip-level parameters": what a float4 copy reaches
FP32_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz, un-packed fp32 (157.3 with v_pk_fma)


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
    # strict lower bound A_min: perfect cross-tile cache reuse - every visible Gaussian's record / feature
    # row is read once by the blend forward and its gradient partial written once by the blend backward
    a["total_min"] = a["total"] - (44 + 4 * C) * (N_r - Pv) - (40 + 4 * C) * (N_r - Pv)
    return a


def scene_stats(scene, dev):
    """One untimed forward through _C to obtain Pv, N and N_r (= sum over tiles of the deepest list
    position any pixel of the tile blends, from the n_contrib plane)."""
    import numpy as np
    import torch
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
    # (entry, quadrant wave) evaluations of phase 1 of the pixel-lane blend backward on THESE lists: a quadrant wave evaluates the
    # chunks of sixteen list positions up to its deepest pixel (render_bwd_pl.hip: `pos_hi - 15 < my_max`)
    qmax = pad.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5)).astype(np.int64)
    bwd_evals = int((16 * ((qmax + 15) // 16)).sum())
    return dict(Pv=int((radii > 0).sum()), N=int(n), N_r=N_r, tiles=gx * gy, bwd_evals=bwd_evals)


def blend_stream_cycles():
    """SIMD cycles per (entry, wave) of the blend kernels' inner streams run by themselves at four waves per SIMD
    (tools/ubench/blend_stream.hip: the code of csrc/pl_phase1.h and csrc/fwd_group.h as the kernels compile it, no barriers, no
    staging, no flush).  Measured LIVE when the binary is there (built by __graft_entry__.build(); ~3 s), otherwise the committed
    measurement of this round."""
    exe = os.path.join(ROOT, "tools", "ubench", "blend_stream")
    src = None
    d = None
    # (F3DGS_BENCH_NO_UBENCH=1: the committed measurement - used under rocprofv3, whose trace would otherwise fill with the
    # micro-benchmark's kernels)
    if os.path.exists(exe) and not STUB and os.environ.get("F3DGS_BENCH_NO_UBENCH", "0") != "1":
        try:
            out = subprocess.run([exe, "json"], capture_output=True, timeout=120, text=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            src = "tools/ubench/blend_stream json, run by this process on this GPU"
        except Exception:      # noqa: BLE001
            d = None
    if d is None:
        nm, d = _committed(("r06_blend_stream.json",))
        src = f"profiles/{nm} (committed measurement; the binary was not available here)" if d else None
    return d, src


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
    return {"value": mpix / dt, "unit": "Mpix/s", "cores": 1, "kind": "port", "workload_fraction": kw["P"] / cfg_kw["P"],
            "sample": f"1 fwd+bwd of the scalar C++ oracle on {kw['P']} Gaussians @{kw['width']}x{kw['height']}, "
                      f"feat_dim={kw['C']} (the workload with 1/5 of the Gaussians), {dt:.1f} s"}


def cpu_reference_path_c1(dev):
    """BASELINE.json config 1 (10k Gaussians, 256x256, RGB only): the PyTorch-CPU autograd restatement on the host
    cores and the product on the GPU at the same config.  SURVEY.md 8(d) asks for torch.set_num_threads(all cores);
    on the 256-core GPU host that setting makes the restatement 700x SLOWER than 8 threads (259 s vs 0.34 s per step,
    measured: thread wake-ups on thousands of tiny per-tile tensors), so the thread count is swept over 1 / 8 / 32
    under a time budget and the best is reported next to the all-cores observation."""
    import statistics

    import torch
    from oracle import torch_oracle
    from synth import CONFIGS, make_scene
    sc = make_scene(seed=0, **CONFIGS["c1"])
    cores = os.cpu_count() or 1
    old = torch.get_num_threads()
    sweep = {}
    budget_end = time.perf_counter() + 25.0
    try:
        for n in sorted({1, min(8, cores), min(32, cores)}):
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            torch_oracle.forward_backward(sc, dtype=torch.float32)          # warm-up
            warm = time.perf_counter() - t0
            times = []
            while len(times) < 5 and time.perf_counter() + warm < budget_end:
                t0 = time.perf_counter()
                torch_oracle.forward_backward(sc, dtype=torch.float32)
                times.append(time.perf_counter() - t0)
            sweep[n] = 1e3 * (statistics.median(times) if times else warm)
    finally:
        torch.set_num_threads(old)
    best_n = min(sweep, key=sweep.get)
    cpu_ms = sweep[best_n]
    step, _ = make_step(sc, dev, pool=2)
    # the GPU has idled through the CPU legs above (tens of seconds: clocks down, caches cold) and c1 is launch-bound (~30
    # launches and one 8-byte read-back per step): a short warm-up measured 1.9 ms per step on the driver's box against
    # 0.23 ms for the same loop on a busy GPU - warm up for real, then time
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        step(i)
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / 200
    mpix = sc["image_width"] * sc["image_height"] / 1e6
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"config": "c1: 10000 Gaussians, 256x256, SH degree 3, feat_dim=0",
            "torch_cpu_autograd_ms": cpu_ms, "torch_cpu_autograd_mpix_s": mpix / (cpu_ms * 1e-3), "threads": best_n,
            "ms_by_threads": sweep, "host_cores": cores, "cpu_model": model,
            "gpu_ms": gpu_ms, "gpu_mpix_s": mpix / (gpu_ms * 1e-3),
            "note": "CPU: oracle/torch_oracle.py (restatement of the reference's algorithm; the reference has no CPU "
                    "rasterizer), median of up to 5 runs after 1 warm-up per thread count, best count reported; with "
                    "torch.set_num_threads(256) one step took 259 s on this host class (profiles/r02_notes.md). "
                    "GPU: this library, mean of 200 steps after 300 warm-up steps (launch-bound: ~30 launches and one host read-back per step)"}


def cpu_reference_path_c3_reduced(P=100_000, threads=8):
    """'additionally a reduced c3-shaped run (e.g. 100k Gaussians @1080p, C=32) if it completes in < 10 min,
    otherwise state "did not complete"' - the PyTorch-CPU autograd restatement, one forward+backward (`--cpu-reduced-c3`;
    minutes of host time, so not part of the default run: the default line quotes the committed result)."""
    import torch
    from oracle import torch_oracle
    from synth import CONFIGS, make_scene
    kw = dict(CONFIGS["c3"])
    kw["P"] = P
    sc = make_scene(seed=0, **kw)
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    try:
        torch_oracle.forward_backward(sc, dtype=torch.float32)
        dt = time.perf_counter() - t0
        status = "completed"
    except Exception as exc:      # noqa: BLE001
        dt, status = time.perf_counter() - t0, f"failed: {exc!r}"
    finally:
        torch.set_num_threads(old)
    mpix = kw["width"] * kw["height"] / 1e6
    return {"config": f"c3 reduced: {P} Gaussians, {kw['width']}x{kw['height']}, SH degree 3, feat_dim={kw['C']}", "status": status,
            "seconds": dt, "mpix_s": mpix / dt if status == "completed" else None, "threads": threads,
            "did_not_complete_in_10_min": dt > 600.0}


def _committed(name_candidates):
    for name in name_candidates:
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            try:
                return name, json.load(open(f))
            except Exception:      # noqa: BLE001
                pass
    return None, None


def _stub_leaves(scene, dev):
    import torch
    P = scene["P"]
    t = lambda x: x.to(dev).clone()
    return dict(means3D=t(scene["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                opacities=t(scene["opacities"]).requires_grad_(), shs=t(scene["shs"]).requires_grad_(),
                semantic_feature=t(scene["semantic_feature"]).requires_grad_(),
                scales=t(scene["scales"]).requires_grad_(), rotations=t(scene["rotations"]).requires_grad_())


def _stub_step(scene, dev, dist=None, n_views=1):
    """STUB mode: every leaf gets a gradient from a few torch operations; the exchange is the real one (dp.dp_step / dp_step_views)."""
    import dp
    leaves = _stub_leaves(scene, dev)
    reduce_keys = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")
    sub = {k: leaves[k] for k in reduce_keys}

    def fwd(vid):
        return sum((v * v).sum() * (1.0 + 0.1 * vid) for v in leaves.values())

    def step(i):
        for v in leaves.values():
            v.grad = None
        if n_views > 1:
            dp.dp_step_views(lambda j: fwd(j), lambda h: h.backward(), sub, range(n_views), overlap=False, n_streams=1, accumulate=False)
        elif dist is None:
            fwd(0).backward()
        else:
            dp.dp_step(lambda vid: fwd(vid).backward(), sub, [0], overlap=False)
    return step, leaves


def make_step(scene, dev, pool=4, dist=None, overlap=True):
    """Returns (step(i), leaves): one forward+backward (+ gradient exchange) with the i-th upstream gradient set."""
    import torch
    if STUB:
        return _stub_step(scene, dev, dist=dist)

    import diff_gaussian_rasterization as dgr
    import dp
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
    g = torch.Generator(device="cpu").manual_seed(1234)
    hw = float(W * H)
    ups = []
    for _ in range(pool):   # fresh upstream gradients: same distribution as the recipe, new values every step
        dd = scene["dL_ddepth"] if float(scene["dL_ddepth"].abs().max()) == 0.0 else torch.randn(1, H, W, generator=g) / hw
        ups.append([t(torch.randn(3, H, W, generator=g) / hw), t(torch.randn(C, H, W, generator=g) / hw), t(dd)])
    reduce_keys = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")   # 59 + C floats

    def fwd_bwd(i):
        color, feat, _radii, depth = rasterizer(**leaves)
        torch.autograd.backward([color, feat, depth], ups[i % pool])

    def step(i):
        for v in leaves.values():
            v.grad = None
        if dist is None:
            fwd_bwd(i)
            return
        sub = {k: leaves[k] for k in reduce_keys}
        # overlap: feature gradient reduced from inside the blend stage, SH gradient in four row ranges from inside the
        # per-Gaussian stage ("shs" is the op's direct input here), the four small tensors in one bucket afterwards
        dp.dp_step(lambda _vid: fwd_bwd(i), sub, [0], overlap=overlap, rows_leaves={"sh": ("shs",)} if overlap else None,
                   rows_chunks=4)
    return step, leaves


def make_step_views(scene, yaws, dev, pool=4, dist=None, overlap=True, pipelined=True):
    """Several views of the same Gaussians per rank and step (`--views-per-iter`): returns (step(i), leaves).  One step =
    forward+backward of every view in `yaws` (camera rotated about y by that many degrees) + the gradient exchange, through
    dp.dp_step_views: views alternate between two streams (view v + 1's preprocess / binning under view v's blend kernels),
    the feature gradient is accumulated in place across the views and - data parallel - reduced from inside the last view's
    backward pass.  pipelined=False: the same views strictly one after the other on one stream, one gradient tensor per view
    (what dp.dp_step did with several views until round 3) - the comparison leg."""
    import torch
    if STUB:
        return _stub_step(scene, dev, dist=dist, n_views=len(yaws))

    import diff_gaussian_rasterization as dgr
    import dp
    from synth import make_camera
    P, C = scene["P"], scene["C"]
    W, H = scene["image_width"], scene["image_height"]
    t = lambda x: x.to(dev)
    bg = t(scene["bg"])
    rasterizers = []
    for yaw in yaws:
        cam = make_camera(W, H, yaw_deg=yaw)
        rasterizers.append(dgr.GaussianRasterizer(dgr.GaussianRasterizationSettings(
            H, W, cam["tanfovx"], cam["tanfovy"], bg, 1.0, t(cam["viewmatrix"]), t(cam["projmatrix"]), scene["sh_degree"],
            t(cam["campos"]), False, False)))
    leaves = dict(means3D=t(scene["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                  opacities=t(scene["opacities"]).requires_grad_(), shs=t(scene["shs"]).requires_grad_(),
                  semantic_feature=t(scene["semantic_feature"]).requires_grad_(),
                  scales=t(scene["scales"]).requires_grad_(), rotations=t(scene["rotations"]).requires_grad_())
    g = torch.Generator(device="cpu").manual_seed(1234)
    hw = float(W * H)
    ups = []
    for _ in range(pool):
        dd = scene["dL_ddepth"] if float(scene["dL_ddepth"].abs().max()) == 0.0 else torch.randn(1, H, W, generator=g) / hw
        ups.append([t(torch.randn(3, H, W, generator=g) / hw), t(torch.randn(C, H, W, generator=g) / hw), t(dd)])
    reduce_keys = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")
    sub = {k: leaves[k] for k in reduce_keys}
    state = {"i": 0}

    def forward(j):
        color, feat, _radii, depth = rasterizers[j](**leaves)
        return (color, feat, depth, j)

    def backward(h):
        torch.autograd.backward([h[0], h[1], h[2]], ups[(state["i"] + h[3]) % pool])

    def step(i):
        state["i"] = i
        leaves["means2D"].grad = None
        dp.dp_step_views(forward, backward, sub, range(len(yaws)), overlap=overlap and pipelined, n_streams=2 if pipelined else 1,
                         accumulate=None if pipelined else False)
    return step, leaves


def config_label(P, W, H, C):
    """'1M Gaussians @1080p, feat_dim=32' for the headline config, the same wording for the others."""
    ps = f"{P // 1000000}M" if P % 1000000 == 0 else (f"{P // 1000}k" if P % 1000 == 0 else str(P))
    res = {(1920, 1080): "@1080p", (3840, 2160): "@4K"}.get((W, H), f"@{W}x{H}")
    return f"{ps} Gaussians {res}, feat_dim={C}"


class _BenchModel:
    """Duck-typed like the reference's GaussianModel (scene/gaussian_model.py) for densify.py: raw parameters (log scale,
    logit opacity, SH split into dc / rest), the densification statistics and an optimizer with the reference's groups."""
    percent_dense = 0.01

    def __init__(self, scene, dev):
        import torch
        from fused_adam import FusedAdam
        par = lambda x: torch.nn.Parameter(x.to(dev).contiguous().clone().requires_grad_(True))
        self._xyz = par(scene["means3D"])
        self._features_dc, self._features_rest = par(scene["shs"][:, :1]), par(scene["shs"][:, 1:])
        self._scaling, self._rotation = par(torch.log(scene["scales"])), par(scene["rotations"])
        op = scene["opacities"].clamp(1e-4, 1 - 1e-4)
        self._opacity = par(torch.log(op / (1 - op)))
        self._semantic_feature = par(scene["semantic_feature"])
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        self.denom = torch.zeros(P, 1, device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)
        groups = [{"params": [getattr(self, attr)], "lr": 1e-4, "name": name} for name, attr in (
            ("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
            ("scaling", "_scaling"), ("rotation", "_rotation"), ("semantic_feature", "_semantic_feature"))]
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)

    def leaves(self):
        """What the reference's render() hands to the op (gaussian_renderer/__init__.py:196-236), as fresh leaves."""
        import torch
        with torch.no_grad():
            d = dict(means3D=self._xyz.detach(), opacities=torch.sigmoid(self._opacity), scales=torch.exp(self._scaling),
                     rotations=torch.nn.functional.normalize(self._rotation), shs=torch.cat((self._features_dc, self._features_rest), dim=1),
                     semantic_feature=self._semantic_feature.detach())
        d = {k: v.contiguous().clone().requires_grad_() for k, v in d.items()}
        d["means2D"] = torch.zeros(self._xyz.shape[0], 3, device=self._xyz.device, requires_grad=True)
        return d


def densify_leg(args, dev):
    """BASELINE.json config c5 as it is meant: 'densification on'.  Every --densify-every steps 5 % of the Gaussians are
    re-sampled BETWEEN the timed iterations through densify.densify_and_prune (the reference's own selection rules on
    prepared statistics: a random 5 % are cloned or split, another 5 % - opacity pushed under the threshold - are pruned),
    so the point count P, the instance count N and every state buffer of the op change while it is being timed.
    A step is timed by a HIP event pair around the op's forward + backward; the densification itself is reported beside it."""
    import statistics

    import torch

    import densify
    import diff_gaussian_rasterization as dgr
    from synth import CONFIGS, make_scene
    cfg_kw = dict(CONFIGS[args.config])
    if args.feat_dim is not None:
        cfg_kw["C"] = args.feat_dim
    scene = make_scene(seed=0, **cfg_kw)
    W, H, C = scene["image_width"], scene["image_height"], scene["C"]
    t = lambda x: x.to(dev)
    settings = dgr.GaussianRasterizationSettings(H, W, scene["tanfovx"], scene["tanfovy"], t(scene["bg"]), 1.0, t(scene["viewmatrix"]),
                                                 t(scene["projmatrix"]), scene["sh_degree"], t(scene["campos"]), False, False)
    rasterizer = dgr.GaussianRasterizer(settings)
    model = _BenchModel(scene, dev)
    g = torch.Generator(device="cpu").manual_seed(1234)
    hw = float(W * H)
    ups = [[t(torch.randn(3, H, W, generator=g) / hw), t(torch.randn(C, H, W, generator=g) / hw), t(torch.randn(1, H, W, generator=g) / hw)]
           for _ in range(2)]
    gd = torch.Generator(device=dev).manual_seed(99)
    extent = 1.7          # percent_dense * extent = 0.017: the median of the largest axis of the recipe's scales -> clones and splits

    def resample():
        P = model._xyz.shape[0]
        u = torch.rand(P, generator=gd, device=dev)
        grow, drop = u < 0.05, u > 0.95
        model.xyz_gradient_accum = torch.where(grow, 1.0, 0.0).reshape(P, 1)
        model.denom = torch.ones(P, 1, device=dev)
        with torch.no_grad():
            model._opacity[drop] = -20.0            # sigmoid < min_opacity: pruned
        return densify.densify_and_prune(model, 0.5, 0.005, extent, None)

    n_list, p_list, step_ms, dens_ms, plans = [], [], [], [], []
    leaves = model.leaves()
    total = args.warmup + args.steps
    torch.cuda.synchronize()
    t_wall = None
    for i in range(total):
        if i == args.warmup:
            torch.cuda.synchronize()
            t_wall = time.perf_counter()
        if i > 0 and i % args.densify_every == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan = resample()
            leaves = model.leaves()
            e1.record()
            if i >= args.warmup:
                plans.append(plan)
                dens_ms.append((e0, e1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        color, feat, radii, depth = rasterizer(**leaves)
        torch.autograd.backward([color, feat, depth], ups[i % 2])
        e1.record()
        for v in leaves.values():
            v.grad = None
        if i >= args.warmup:
            step_ms.append((e0, e1))
            p_list.append(int(leaves["means3D"].shape[0]))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_wall
    step_ms = sorted(a.elapsed_time(b) for a, b in step_ms)
    dens = [a.elapsed_time(b) for a, b in dens_ms]
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]
    mean_ms = sum(step_ms) / len(step_ms)
    pool = getattr(model, "_row_pool", None) or getattr(model, "_f3dgs_pool", None)
    out = {
        "metric": f"rendered Mpix/s of rasterizer fwd+bwd with densification on, {config_label(scene['P'], W, H, C)} + depth",
        "value": W * H / 1e6 / (mean_ms * 1e-3), "unit": "Mpix/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": mean_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {scene['P']} Gaussians at the start, {W}x{H}, SH degree {scene['sh_degree']}, feat_dim={C}, depth "
                               f"gradients, 5 % of the Gaussians re-sampled every {args.densify_every} steps between the timed iterations",
                   "P_min": min(p_list), "P_max": max(p_list), "densifications": len(plans),
                   "last_plan": plans[-1] if plans else None},
        "step_ms": {"mean": mean_ms, "median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "max": step_ms[-1], "n": len(step_ms),
                    "source": "HIP event pair around the op's forward + backward of every timed step"},
        "densification_ms": {"mean": (sum(dens) / len(dens)) if dens else None, "max": max(dens) if dens else None,
                             "includes": "densify.densify_and_prune + the activations of the new leaves (outside the step)"},
        "wall_ms_per_step_including_densification": 1e3 * wall / args.steps,
        "row_pool_reallocations": getattr(pool, "reallocations", None),
    }
    print(json.dumps(out), flush=True)


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run and relay its output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm-only", action="store_true", help="time only the gradient exchange of the config (N > 1)")
    ap.add_argument("--no-overlap", action="store_true", help="exchange all gradients after the backward pass")
    ap.add_argument("--feat-dim", type=int, default=None, help="override the config's feature dim (development)")
    ap.add_argument("--cpu-reduced-c3", action="store_true",
                    help="only run the reduced c3-shaped PyTorch-CPU leg of SURVEY.md 8(d) (100k Gaussians @1080p, C=32; minutes) and print it")
    ap.add_argument("--views-per-iter", type=int, default=0,
                    help="V views of the same Gaussians per step over all GPUs (V / N per rank, pipelined over two streams, "
                         "gradients accumulated across the views): total work is fixed as N grows - the line says scaling: strong")
    ap.add_argument("--graph", action="store_true", help="also time the step replayed from a HIP graph (always done for c1)")
    ap.add_argument("--band-split", type=int, default=0, help="also time the view split into this many tile-row bands, band by band on this GPU")
    ap.add_argument("--valu", action="store_true",
                    help="library option feature_mfma = 0: every blend kernel on the vector pipe only (the north-star-literal configuration)")
    ap.add_argument("--densify-every", type=int, default=0,
                    help="config c5 'densification on' (SURVEY.md 8d): every N steps 5 %% of the Gaussians are re-sampled through "
                         "densify.densify_and_prune between the timed iterations (N and the state buffers change)")
    args = ap.parse_args()

    if args.cpu_reduced_c3:
        print(json.dumps(cpu_reference_path_c3_reduced()), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not STUB and not ONE_DEVICE:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to report fewer")
        raise SystemExit(respawn_under_torchrun(args))
    world = max(world, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}; "
                         f"launch with torch.distributed.run --nproc-per-node {args.gpus}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    if STUB:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the rasterizer has no CPU path")
        # F3DGS_BENCH_ONE_DEVICE=1 (a rehearsal, never a measurement: the line says so): every rank on cuda:0, gloo as transport -
        # the whole multi-rank flow of this file with the REAL op on a one-GPU box
        local_dev = 0 if ONE_DEVICE else local_rank
        torch.cuda.set_device(local_dev)
        dev = torch.device("cuda", local_dev)
        sync = torch.cuda.synchronize
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if STUB or ONE_DEVICE:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from synth import CONFIGS, make_scene
    if STUB:
        class _C:      # noqa: N801  (the op's option / profile surface, inert)
            _opts = {"feature_mfma": 1, "bwd_bf16": -1, "bwd_bf16_max_ratio": 16, "bwd_pl": -1, "tile_cull": 1, "profile": 0}
            set_option = staticmethod(lambda k, v: _C._opts.__setitem__(k, v))
            get_option = staticmethod(lambda k: _C._opts.get(k, 0))
            profile_reset = staticmethod(lambda: None)
            profile_read = staticmethod(lambda: [("render_fwd", 1.0, 1), ("render_bwd", 1.0, 1), ("preprocess", 0.5, 1), ("preprocess_bwd", 0.5, 1)])
        CONFIGS = {k: dict(P=1500, width=64, height=48, C=8, with_depth_grad=False) for k in CONFIGS}
    else:
        from diff_gaussian_rasterization import _C

    if args.valu:
        _C.set_option("feature_mfma", 0)
    if args.densify_every > 0:
        if world != 1:
            raise SystemExit("--densify-every is a single-GPU leg")
        densify_leg(args, dev)
        return
    cfg_kw = dict(CONFIGS[args.config])
    if args.feat_dim is not None:
        cfg_kw["C"] = args.feat_dim
    V = args.views_per_iter
    if V and (V % world != 0):
        raise SystemExit(f"--views-per-iter {V} is not a multiple of --gpus {world}")
    per_rank = V // world if V else 1
    scene = make_scene(seed=0, yaw_deg=5.0 * rank * per_rank, **cfg_kw)
    P, C = scene["P"], scene["C"]
    W, H = scene["image_width"], scene["image_height"]
    if V:
        yaws = [5.0 * (rank * per_rank + j) for j in range(per_rank)]
        step, leaves = make_step_views(scene, yaws, dev, dist=dist, overlap=not args.no_overlap)
    else:
        step, leaves = make_step(scene, dev, dist=dist, overlap=not args.no_overlap)

    def timed(n_steps, per_step_events):       # (`step` is looked up at call time)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)] if per_step_events and not STUB else None
        sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            if evs:
                evs[i].record()
            step(i)
        if evs:
            evs[n_steps].record()
        sync()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_steps)] if evs else ([1e3 * el / n_steps] * n_steps if per_step_events else None)
        return el, per

    if args.comm_only:
        import dp
        if dist is None:
            raise SystemExit("--comm-only needs --gpus N > 1")
        for v in leaves.values():
            v.grad = torch.zeros_like(v)
        grads = {k: leaves[k].grad for k in ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")}
        nbytes = sum(g.numel() * 4 for g in grads.values())

        def time_group(group):
            for _ in range(args.warmup):
                dp.all_reduce_gaussian_grads(grads, group=group)
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dp.all_reduce_gaussian_grads(grads, group=group)
            sync()
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        el = time_group(None)
        # the same exchange without a ring, out of an all-to-all and an all-gather (dp.all_reduce_direct; F3DGS_DP_EXCHANGE=direct
        # makes it the training step's exchange)
        dp.EXCHANGE = "direct"
        try:
            t_direct = time_group(None)
            direct = {"ms": 1e3 * t_direct / args.steps, "algbw_GBps": nbytes / (t_direct / args.steps) / 1e9}
        except Exception as exc:      # noqa: BLE001
            direct = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        dp.EXCHANGE = "allreduce"
        # The same exchange on communicators created under other NCCL_ALGO / NCCL_PROTO settings (RCCL reads them when a
        # communicator is initialised): xGMI is a point-to-point mesh, and a ring all-reduce is bound by ONE link (SURVEY.md 5:
        # ~29 ms against ~4 ms for c4) - this table says what the library picked and what the alternatives cost on this node.
        sweep = {}
        if not STUB or os.environ.get("F3DGS_BENCH_STUB_SWEEP", "1") == "1":
            saved = {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO")}
            for algo, proto in ((None, None), ("Ring", None), ("Tree", None), ("Ring", "Simple"), ("Ring", "LL128"), ("Tree", "Simple")):
                if algo is None:
                    continue
                os.environ["NCCL_ALGO"] = algo
                if proto:
                    os.environ["NCCL_PROTO"] = proto
                else:
                    os.environ.pop("NCCL_PROTO", None)
                label = f"NCCL_ALGO={algo}" + (f",NCCL_PROTO={proto}" if proto else "")
                try:
                    g = dist.new_group(backend="gloo" if (STUB or ONE_DEVICE) else "nccl")
                    t = time_group(g)
                    sweep[label] = {"ms": 1e3 * t / args.steps, "algbw_GBps": nbytes / (t / args.steps) / 1e9}
                    dist.destroy_process_group(g)
                except Exception as exc:      # noqa: BLE001  (a setting the library refuses must not cost the line)
                    sweep[label] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if rank == 0:
            print(json.dumps({"metric": "gradient all-reduce only (no rendering)", "value": 1e3 * el / args.steps, "unit": "ms",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": False,
                              "bytes_per_rank": nbytes, "algbw_GBps": nbytes / (el / args.steps) / 1e9,
                              "direct_all_to_all_plus_all_gather": direct,
                              "by_setting": sweep, "data": "stub" if STUB else "synthetic",
                              "config": {"workload": f"{args.config}: (59+{C}) floats x {P} Gaussians"}}), flush=True)
        dist.destroy_process_group()
        return

    stats = None
    if rank == 0 and STUB:
        stats = dict(Pv=P, N=4 * P, N_r=2 * P, tiles=12, bwd_evals=8 * P)
    elif rank == 0:
        # N_r of SURVEY.md 8(d) is defined on the reference's (bounding-rectangle) instance lists
        _C.set_option("tile_cull", 0)
        stats = scene_stats(scene, dev)
        _C.set_option("tile_cull", 1)
        stats["bwd_evals"] = scene_stats(scene, dev)["bwd_evals"]      # of the lists the product walks
    for i in range(args.warmup):
        step(i)
    # events-off reference run (same K steps): shows what the events of the timed region cost
    _C.set_option("profile", 0)
    el_plain, _ = timed(args.steps, per_step_events=False)
    # THE timed region: HIP events around the two blend kernels (the roofline's kernel duration is measured HERE, on
    # the op's stream)
    _C.set_option("profile", 2)
    # (the library creates its HIP events on first use and pools them: an untimed pass of the same K steps creates the 4 K events
    # the timed region records, so that no hipEventCreate - one run in ten read 1.38 instead of 1.26 ms at c3, a 2.4 ms stall in
    # its first profiled pass - falls inside it; counted in steps_run_before_the_timed_region)
    timed(args.steps, per_step_events=False)
    _C.profile_read()
    _C.profile_reset()
    elapsed, _ = timed(args.steps, per_step_events=False)
    blend = {name: (ms, calls) for name, ms, calls in _C.profile_read()}
    # auxiliary run of the same K steps with an event at every stage boundary (11 per step, ~4 us of device time each)
    # and an event pair per step: the stage split and the per-step distribution; not part of `value`
    _C.set_option("profile", 1)
    _C.profile_reset()
    el_stages, per_step = timed(args.steps, per_step_events=True)
    prof = {name: (ms, calls) for name, ms, calls in _C.profile_read()}
    _C.set_option("profile", 0)

    # What ran: the blend backward's per-Gaussian sums go through bf16 matrix instructions (two-term operands, f32 accumulation)
    # when the pixel-lane kernel with its bf16 shape is selected - read from the library's options, not assumed.
    mf = _C.get_option("feature_mfma") != 0
    opt_bf16 = _C.get_option("bwd_bf16")         # 1 always, 0 never, -1 by the frame's conditioning (the library's default)
    if STUB:
        bwd_bf16_active = True
    else:       # what the last backward call of the runs above actually contracted with
        bwd_bf16_active = bool(_C.last_backward_contraction() == 1)
    bwd_hybrid = (not STUB) and _C.last_backward_contraction() == 2
    fp32_exact = None
    if bwd_bf16_active and not V:
        # the same K steps with every contraction of the blend backward on exact-fp32 matrix instructions (option bwd_bf16 = 0):
        # the figure to hold against a reference that computes in fp32 throughout
        _C.set_option("bwd_bf16", 0)
        try:
            for i in range(max(3, args.warmup // 2)):
                step(i)
            _C.set_option("profile", 2)
            _C.profile_reset()
            el_x, _ = timed(args.steps, per_step_events=False)
            bx = {name: (ms, calls) for name, ms, calls in _C.profile_read()}
            fp32_exact = {"ms_per_step": 1e3 * el_x / args.steps,
                          "render_bwd_ms": (bx["render_bwd"][0] / max(1, bx["render_bwd"][1])) if "render_bwd" in bx else None,
                          "render_fwd_ms": (bx["render_fwd"][0] / max(1, bx["render_fwd"][1])) if "render_fwd" in bx else None}
        finally:
            _C.set_option("profile", 0)
            _C.set_option("bwd_bf16", opt_bf16)
        for i in range(2):
            step(i)

    # The same step replayed from a HIP graph (graph_step.CapturedStep: option sync_free, no host read in the forward call).
    # Small scenes are bound by the host's enqueue time (c1: ~30 launches of 5 - 40 us); the reference cannot be captured - it
    # reads num_rendered back in the middle of its forward call (rasterizer_impl.cu:283).  One upstream-gradient set (the
    # graph's inputs are static); the frame's lists are checked against the provision after the timed replays.
    graph_replay = None
    if not STUB and dist is None and not V and (args.graph or args.config == "c1"):
        try:
            from graph_step import CapturedStep
            cs = CapturedStep(lambda: step(0)).capture()
            for _ in range(max(3, args.warmup)):
                cs.replay()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cs.replay()
            sync()
            el_g = time.perf_counter() - t0
            fits = bool(cs.check())
            cnt = cs.counts()
            graph_replay = {"ms_per_step": 1e3 * el_g / args.steps, "value_mpix_s": W * H / 1e6 / (el_g / args.steps),
                            "steps": args.steps, "fits_the_provision": fits,
                            "list_entries": cnt[0] if cnt else None, "entries_provided_for": cnt[3] if cnt else None,
                            "blend_backward_contraction": {1: "bf16 two-term", 2: "hybrid",
                                                           3: "chosen on the device by the frame's long-axis word (bf16 two-term here unless a visible Gaussian has a long axis: then hybrid); both first-window kernels are launched, one runs",
                                                           0: "exact fp32"}.get(_C.last_backward_contraction()),
                            "note": "the whole forward + backward of the op as ONE hipGraphLaunch per step (torch.cuda.graph); eager steps above it in this line"}
            del cs
        except Exception as exc:      # noqa: BLE001 (a leg beside the headline: report, never fail the line)
            graph_replay = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        for i in range(2):
            step(i)

    # ONE view split over W GPUs by tile rows (f3dgs_set_tile_band, dp.band_rows): this GPU plays every rank in turn - the step of
    # band r with the upstream gradients of its own rows.  What an W-GPU split of the view would take per step, exchange excluded,
    # is the slowest band; the per-Gaussian stages are repeated on every rank.
    band_split = None
    if not STUB and dist is None and not V and args.band_split > 1:
        import dp as _dp
        per_band = []
        try:
            for r in range(args.band_split):
                r0, r1, _y0, _y1 = _dp.band_rows(H, r, args.band_split)
                _C.set_tile_band(r0, r1) if r1 > r0 else _C.set_tile_band(1 << 20, 1 << 20)
                for i in range(3):
                    step(i)
                el_b, _ = timed(args.steps, per_step_events=False)
                per_band.append(1e3 * el_b / args.steps)
        finally:
            _C.set_tile_band(0, 0)
        for i in range(2):
            step(i)
        band_split = {"world": args.band_split, "ms_per_band": per_band, "slowest_band_ms": max(per_band),
                      "whole_view_ms": 1e3 * elapsed / args.steps,
                      "note": "each band timed on THIS GPU with the whole view's upstream gradients (rows outside the band meet empty "
                              "tile lists); compute only - the gradient exchange of a W-GPU split is the view-sharded step's"}

    views_breakdown = None
    if V:
        # the same K steps with the views strictly one after the other (one stream, one gradient tensor per view), and one view
        # alone: what the pipelining and the in-place accumulation buy
        _step = step
        step, _lv = make_step_views(scene, yaws, dev, dist=dist, overlap=False, pipelined=False)
        for i in range(3):
            step(i)
        el_seq, _ = timed(args.steps, per_step_events=False)
        del _lv
        step, _lv = make_step_views(scene, yaws[:1], dev, dist=None, overlap=False, pipelined=False)
        for i in range(3):
            step(i)
        el_one, _ = timed(args.steps, per_step_events=False)
        del _lv
        step = _step
        views_breakdown = {"views_per_rank": per_rank, "pipelined_ms_per_step": 1e3 * elapsed / args.steps,
                           "sequential_ms_per_step": 1e3 * el_seq / args.steps, "one_view_ms": 1e3 * el_one / args.steps,
                           "pipelined_over_views_x_one_view": (elapsed / args.steps) / (per_rank * el_one / args.steps),
                           "note": "sequential = the same views one after the other on one stream with a gradient tensor per view; "
                                   "one_view = a single view of this rank without any exchange"}
    dp_breakdown = None
    if dist is not None and not V:
        # the same K steps without the exchange, and the exchange alone on the gradients they left: what the scaling is made of
        import dp
        step_local, leaves_local = make_step(scene, dev, dist=None)
        for i in range(3):
            step_local(i)
        _step = step
        step = step_local
        el_compute, _ = timed(args.steps, per_step_events=False)
        step = _step
        grads = {k: (leaves_local[k].grad if leaves_local[k].grad is not None else torch.zeros_like(leaves_local[k]))
                 for k in ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")}
        for _ in range(3):
            dp.all_reduce_gaussian_grads(grads)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dp.all_reduce_gaussian_grads(grads)
        sync()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        nbytes = sum(g.numel() * 4 for g in grads.values())
        comm_direct_ms = None
        try:          # the same exchange as all-to-all + local sum + all-gather (dp.all_reduce_direct): no ring by construction
            dp.EXCHANGE = "direct"
            for _ in range(2):
                dp.all_reduce_gaussian_grads(grads)
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dp.all_reduce_gaussian_grads(grads)
            sync()
            dist.barrier()
            td = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            comm_direct_ms = 1e3 * float(td.item()) / args.steps
        except Exception:      # noqa: BLE001
            comm_direct_ms = None
        finally:
            dp.EXCHANGE = "allreduce"
        dp_breakdown = {"compute_only_ms": 1e3 * el_compute / args.steps, "comm_only_ms": 1e3 * float(tt.item()) / args.steps,
                        "comm_only_direct_ms": comm_direct_ms,
                        "bytes_per_rank": nbytes, "comm_algbw_GBps": nbytes / (float(tt.item()) / args.steps) / 1e9,
                        "note": "same K steps without the exchange (max over ranks), and the bucketed all-reduce alone; "
                                "ms_per_step below their sum = overlap achieved"}
        # Two whole training steps, optimizer included: (A) this bench's exchange (all-reduce with its in-backward overlap) + the
        # full FusedAdam step on every rank; (B) local gradients -> dp.ShardedOptimizer: reduce-scatter, FusedAdam on this rank's
        # 1/N of the rows, all-gather of the updated parameters (half the gradient bytes; tests/test_dp_gloo.py: bit-identical
        # parameters at world size 2).  What the first multi-GPU lease needs to decide between them.
        try:
            keys6 = ("means3D", "shs", "semantic_feature", "opacities", "scales", "rotations")
            if STUB:
                mk = lambda t: torch.optim.Adam([{"params": [t[k]], "lr": 1e-5} for k in keys6], lr=0.0, eps=1e-15)
            else:
                from fused_adam import FusedAdam
                mk = lambda t: FusedAdam([{"params": [t[k]], "lr": 1e-5} for k in keys6], lr=0.0, eps=1e-15)
            opt_a = mk({k: leaves[k] for k in keys6})

            def step_a(i):
                _step(i)
                opt_a.step()
            step = step_a
            for i in range(3):
                step(i)
            el_a, _ = timed(args.steps, per_step_events=False)
            sh = dp.ShardedOptimizer({k: leaves_local[k] for k in keys6}, mk)

            def step_b(i):
                step_local(i)
                sh.step({k: (leaves_local[k].grad if leaves_local[k].grad is not None else torch.zeros_like(leaves_local[k])) for k in keys6})
            step = step_b
            for i in range(3):
                step(i)
            el_b, _ = timed(args.steps, per_step_events=False)
            step = _step
            dp_breakdown.update({"train_step_allreduce_full_adam_ms": 1e3 * el_a / args.steps,
                                 "train_step_sharded_optimizer_ms": 1e3 * el_b / args.steps,
                                 "train_step_note": "whole steps INCLUDING the optimizer: all-reduce (overlapped with the backward pass) + "
                                                    "FusedAdam on all rows, against reduce-scatter + FusedAdam on 1/N of the rows + all-gather of "
                                                    "the parameters (dp.ShardedOptimizer; not overlapped)"})
        except Exception as exc:     # diagnostic legs must never cost the bench line
            dp_breakdown["train_step_note"] = f"sharded-optimizer legs failed: {type(exc).__name__}: {exc}"
            step = _step
        # Two views per GPU and step (what `--views-per-iter 2N` reports as its own line): the exchange is per STEP, so its share
        # halves - the third column of DESIGN.md 6's table (c3 6.7x, c4 6.3x by the model) from the same lease.
        try:
            yaws2 = [5.0 * (2 * rank + j) for j in range(2)]
            step2, _lv2 = make_step_views(scene, yaws2, dev, dist=dist, overlap=not args.no_overlap)
            step = step2
            for i in range(3):
                step(i)
            el_2v, _ = timed(args.steps, per_step_events=False)
            step = _step
            del _lv2
            dp_breakdown.update({"two_views_per_gpu_ms_per_step": 1e3 * el_2v / args.steps,
                                 "two_views_per_gpu_mpix_s": 2 * world * W * H / 1e6 / (el_2v / args.steps),
                                 "two_views_per_gpu_note": f"{2 * world} views per step, two per GPU, pipelined over two streams, feature gradient "
                                                           "accumulated in place, ONE exchange per step (bench.py --views-per-iter "
                                                           f"{2 * world} gives the full line)"})
        except Exception as exc:     # noqa: BLE001
            dp_breakdown["two_views_per_gpu_note"] = f"leg failed: {type(exc).__name__}: {exc}"
            step = _step
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        mpix = (V if V else world) * W * H / 1e6 / (elapsed / args.steps)
        stage_ms = {k: v[0] / max(1, v[1]) for k, v in prof.items()}
        alg = algorithmic_bytes(P, stats["Pv"], stats["N"], stats["N_r"], W * H, stats["tiles"], C)
        kernel_stage = {"preprocess": "preprocess", "render_fwd": "render_fwd", "render_bwd": "render_bwd",
                        "preprocess_bwd": "preprocess_bwd"}
        dom = max(kernel_stage, key=lambda k: stage_ms.get(kernel_stage[k], 0.0))
        if dom in blend:      # measured inside the timed region
            dom_ms = blend[dom][0] / max(1, blend[dom][1])
        else:                 # a per-Gaussian kernel dominates (no such config so far): from the auxiliary run
            dom_ms = stage_ms.get(kernel_stage[dom], float("nan"))
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
        profiled = None
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
            f = os.path.join(ROOT, "profiles", name)
            if os.path.exists(f):
                try:
                    raw = open(f, "rb").read()
                    d = json.loads(raw)
                    import hashlib
                    profiled = {"source": f"profiles/{name} (rocprofv3 --pmc, separate runs of this config; NOT measured by "
                                          f"this process)", "source_sha256_16": hashlib.sha256(raw).hexdigest()[:16],
                                **({dom: d[dom]} if dom in d else {})}
                    break
                except Exception:
                    pass
        # VALU-issue fraction of both blend kernels from the committed SQ counters (same caveat: a separate run)
        sq_names = {"render_bwd": "render_backward", "render_fwd": "render_forward", "preprocess": "preprocess_kernel",
                    "preprocess_bwd": "preprocess_backward_kernel"}
        for sqf in ("r06_pmc_sq_counters.json", "r05_pmc_sq_counters.json", "r04_pmc_sq_counters.json", "r03_pmc_sq_counters.json", "r02_pmc_sq_counters.json"):
            f = os.path.join(ROOT, "profiles", sqf)
            if profiled is None or not os.path.exists(f):
                continue
            try:
                table = json.load(open(f))
                for stage in dict.fromkeys((dom, "render_fwd", "render_bwd")):
                    rows = [v for k, v in table.items() if k.startswith(sq_names[stage])]
                    if not rows:
                        continue
                    r = max(rows, key=lambda v: v.get("SQ_INSTS_VALU", 0))
                    # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md); SQ_BUSY_CYCLES is summed over the 32
                    # shader engines, so /32 is the kernel's duration in shader cycles; 1024 SIMDs
                    profiled.setdefault("valu_issue", {})[stage] = {
                        "frac": 4.0 * r["SQ_ACTIVE_INST_VALU"] / (r["SQ_BUSY_CYCLES"] / 32.0 * 1024.0),
                        "matrix_pipe_frac": r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (r["SQ_BUSY_CYCLES"] / 32.0 * 1024.0),
                        "valu_instructions_per_launch": r["SQ_INSTS_VALU"], "mfma_instructions_per_launch": r["SQ_INSTS_MFMA"]}
                    # fp32 matrix instructions and vector instructions of a SIMD exclude each other on gfx950 (tools/pipe_probe.hip,
                    # DESIGN.md 3.5): the kernel's issue roofline is the SUM of the two; plain vector instructions = SQ_INSTS_VALU
                    # - SQ_INSTS_MFMA (the counter includes the matrix instructions), 4 cycles each (measured for these kernels'
                    # instruction mixes in round 6: 3.8 in the backward's phase 1, 4.0 in the forward's group step -
                    # profiles/r06_inst_rate.txt, r06_blend_stream.json; `roofline_compute` is the measured version of this figure)
                    cyc = r["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
                    profiled["valu_issue"][stage]["vector_plus_matrix_frac"] = (
                        4.0 * (r["SQ_INSTS_VALU"] - r["SQ_INSTS_MFMA"]) + r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / cyc
                profiled["valu_issue_source"] = (f"profiles/{sqf} (rocprofv3 --pmc, a separate run of this config); frac = 4 x "
                                                 "SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs), matrix_pipe_frac = "
                                                 "SQ_VALU_MFMA_BUSY_CYCLES / the same; vector_plus_matrix_frac = (4 x (SQ_INSTS_VALU - SQ_INSTS_MFMA) + "
                                                 "SQ_VALU_MFMA_BUSY_CYCLES) / the same: the two exclude each other on a SIMD "
                                                 "(profiles/r04_pipe_probe.txt), their sum is what the blend kernels are bound by")
                break
            except Exception:
                pass
        per = sorted(per_step)
        pct = lambda q: per[min(len(per) - 1, int(q * len(per)))]
        if args.valu or not mf:
            blend_label = "vector pipe only (option feature_mfma = 0)"
        elif C == 0:
            blend_label = "RGB only: both blend kernels on the vector pipe (no feature contraction)"
        else:
            blend_label = ("forward: feature contraction on exact-fp32 matrix instructions; backward: " +
                           ("every per-Gaussian sum on bf16 matrix instructions with two-term operands (f32 accumulation; option bwd_bf16 = 1; "
                            "gradients within 1e-4 |g| + 1e-5 max|g| of the exact-fp32 contraction, tests/test_gpu_parity.py)"
                            if bwd_bf16_active else "every per-Gaussian sum on exact-fp32 matrix instructions (option bwd_bf16 = 0)"))
        dtype_label = "f32 (blend-backward sums: bf16 x 2-term products, f32 accumulate)" if bwd_bf16_active else "f32"
        views_here = per_rank if V else 1                      # views one rank renders per step
        alg_step = alg["total"] * views_here                   # algorithmic bytes ONE GPU moves per step
        alg_step_min = alg["total_min"] * views_here
        # compute roofline of the dominant kernel: what the vector pipe (and the LDS traffic of the same stream) allows phase 1 of
        # the pixel-lane blend backward, measured by running that stream alone (tools/ubench/blend_stream.hip)
        roofline_compute = None
        if dom == "render_bwd" and bwd_bf16_active and not V:
            cyc, cyc_src = blend_stream_cycles()
            if cyc:
                n_simd = 4 * (256 if STUB else torch.cuda.get_device_properties(dev).multi_processor_count)
                c1 = cyc["first_window_sched1"][3]
                cw = cyc["later_window_sched1"][3]
                # later channel windows as render_bwd_pl.hip plans them: 128 channels (eight waves) where more than 64 remain and
                # option bwd_wide8 is on, else 64
                later, left = 0, max(0, C - 32)
                wide8 = (not STUB) and _C.get_option("bwd_wide8") != 0
                while left > 0:
                    left -= min(128, left) if (wide8 and left > 64) else min(64, left)
                    later += 1
                evals = stats["bwd_evals"]
                floor_ms = evals * (c1 + later * cw) / (n_simd * 2.4e9) * 1e3
                roofline_compute = {
                    "kernel": "render_bwd", "bound": "vector-instruction issue + LDS of phase 1 (every wave of a SIMD in phase 1 all the time)",
                    "floor_ms": floor_ms, "achieved_ms": dom_ms, "frac": floor_ms / dom_ms,
                    "evaluations_per_launch": evals, "evaluation": "(list entry, quadrant wave of 64 pixel lanes), first channel window",
                    "simd_cycles_per_evaluation": c1, "simd_cycles_per_evaluation_later_window": cw, "later_windows": later,
                    "simds": n_simd, "clock_GHz": 2.4, "cycles_source": cyc_src,
                    "note": "floor = evaluations x SIMD cycles per evaluation at four waves per SIMD / (SIMDs x clock); phase 1 is 2/3 of the "
                            "kernel's vector instructions - staging, the matrix phase and the flush are NOT in the floor (DESIGN.md 3.5)"}
                fw = cyc.get("forward_c32_with_matrix")
                if fw and C == 32:
                    fwd_ms = blend["render_fwd"][0] / max(1, blend["render_fwd"][1]) if "render_fwd" in blend else stage_ms.get("render_fwd")
                    sqv = ((profiled or {}).get("valu_issue") or {}).get("render_fwd", {}).get("valu_instructions_per_launch")
                    if sqv and fwd_ms:
                        ev_f = sqv / 26.0          # 52 vector instructions per entry pair
                        roofline_compute["render_fwd"] = {
                            "floor_ms": ev_f * fw[3] / (n_simd * 2.4e9) * 1e3, "achieved_ms": fwd_ms,
                            "frac": ev_f * fw[3] / (n_simd * 2.4e9) * 1e3 / fwd_ms, "simd_cycles_per_evaluation": fw[3],
                            "simd_cycles_per_evaluation_vector_only": cyc["forward_c32_vector_only"][3],
                            "evaluations_per_launch": ev_f,
                            "evaluations_source": "committed SQ_INSTS_VALU of this config / 26 vector instructions per (entry, wave) "
                                                  "(profiled.valu_issue; a separate run)"}
        out = {
            "metric": f"rendered Mpix/s of rasterizer fwd+bwd (train-step ms in ms_per_step), {config_label(P, W, H, C)}",
            "value": mpix, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if V else "weak", "vs_baseline": None,
            "dtype": dtype_label, "data": "stub" if STUB else ("synthetic; REHEARSAL: all ranks share cuda:0, gloo as transport - not a measurement" if ONE_DEVICE and world > 1 else "synthetic"),
            "blend_kernels": blend_label,
            "options": {k: _C.get_option(k) for k in ("feature_mfma", "bwd_bf16", "bwd_bf16_max_ratio", "bwd_pl", "tile_cull", "sync_free")},
            "blend_backward_contraction": "bf16 two-term" if bwd_bf16_active else
                                          ("hybrid: bf16 feature / colour blocks, exact-fp32 moment block" if bwd_hybrid else "exact fp32"),
            "graph_replay": graph_replay,
            "band_split": band_split,
            "ms_per_step_fp32_exact": fp32_exact["ms_per_step"] if fp32_exact else (ms_per_step if not bwd_bf16_active else None),
            "fp32_exact": ({**fp32_exact,
                            "value_mpix_s": world * W * H / 1e6 / (fp32_exact["ms_per_step"] * 1e-3),
                            "roofline": ({"kernel": "render_bwd", "kernel_ms": fp32_exact["render_bwd_ms"],
                                          "achieved": alg["render_bwd"] / (fp32_exact["render_bwd_ms"] * 1e-3) / 1e9, "unit": "GB/s",
                                          "frac": alg["render_bwd"] / (fp32_exact["render_bwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                         if fp32_exact.get("render_bwd_ms") else None),
                            "note": "the same K steps with option bwd_bf16 = 0: every contraction of the blend backward on exact-fp32 "
                                    "matrix instructions (bit-equal to an fmaf chain); the forward is exact fp32 in both"}
                           if fp32_exact else None),
            "config": {"workload": f"{args.config}: {P} Gaussians, {W}x{H}, SH degree {scene['sh_degree']}, feat_dim={C}, "
                                   + (f"{V} views per step ({per_rank} per GPU, rotated 5 degrees apart, pipelined over two streams, "
                                      f"gradients accumulated across the views)" if V else "one view per GPU")
                                   + " (SURVEY.md 8d recipe, seed 0), fresh upstream gradients per step",
                       "P": P, "Pv": stats["Pv"], "N": stats["N"], "N_r": stats["N_r"],
                       "parallelism": "single GPU" if world == 1 else
                       f"view-sharded dp{world} + RCCL all-reduce of (59+C) floats per Gaussian"
                       + ("" if args.no_overlap else ", feature and SH all-reduces started inside the backward pass")},
            # order of the runs in this process: W warm-up steps, K steps without any event (reported as
            # step_ms.ms_per_step_without_events), K untimed steps with the timed region's events (they create the library's
            # event pool), THEN the K timed steps of `value`
            "steps_run_before_the_timed_region": args.warmup + 2 * args.steps,
            "step_ms": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "n": len(per),
                        "source": "HIP event pair per step on the op's stream, rank 0, in the auxiliary run with all stage events",
                        "ms_per_step_without_events": 1e3 * el_plain / args.steps,
                        "ms_per_step_with_all_stage_events": 1e3 * el_stages / args.steps},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_achievable_6290": achieved / HBM_ACHIEVABLE_GBS,
                         # HBM bytes per launch of this kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE with
                         # the guide's corrections, tools/pmc_summary.py) of this config: counters cannot be read in-process
                         "traffic": (profiled or {}).get(dom) if args.config == "c3" and not V else None,
                         "traffic_source": (profiled or {}).get("source") if args.config == "c3" and not V else None,
                         "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": dom_ms,
                         "kernel_ms_source": "HIP events around the kernel on the op's stream, inside the timed region",
                         "traffic_source_sha256_16": (profiled or {}).get("source_sha256_16") if args.config == "c3" and not V else None,
                         "note": "HBM is the roofline the contract names for this path; the kernel itself is bound by "
                                 "vector-instruction issue (an exponential, a reciprocal, three threshold tests and ~30 dependent operations "
                                 "per (pixel, Gaussian) pair; DESIGN.md 3.5, profiles/): see `profiled.valu_issue.*`"},
            "profiled": profiled,
            "roofline_compute": roofline_compute,
            "roofline_whole_step": {"algorithmic_bytes": alg_step, "algorithmic_bytes_min": alg_step_min, "views_per_gpu_and_step": views_here,
                                    "achieved": alg_step / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s per GPU",
                                    "frac": alg_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "frac_of_achievable_6290": alg_step / (ms_per_step * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS,
                                    "frac_A_min": alg_step_min / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "A = SURVEY.md 8(d) algorithmic bytes of the REFERENCE's algorithm (152 B per instance of sort "
                                            "traffic this design does not move); A_min = the same with perfect cross-tile reuse"},
            "dp_breakdown": dp_breakdown,
            "views_breakdown": views_breakdown,
            "stage_ms": stage_ms,
            "stage_ms_source": "auxiliary run of the same K steps with an event at every stage boundary (not the timed region)",
        }
        if world == 1 and not args.no_cpu_baseline and not STUB:
            out["cpu_baseline"] = cpu_baseline(cfg_kw)
            out["cpu_reference_path_c1"] = cpu_reference_path_c1(dev)
            nm, red = _committed(("r06_cpu_c3_reduced.json", "r05_cpu_c3_reduced.json", "r04_cpu_c3_reduced.json"))
            out["cpu_reference_path_c3_reduced"] = ({"source": f"profiles/{nm} (bench.py --cpu-reduced-c3 on a GPU box's host; minutes of "
                                                               f"host time, not re-run here)", **red} if red else
                                                    {"status": "not measured in this tree: run bench.py --cpu-reduced-c3"})
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
