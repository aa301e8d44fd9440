"""Per-stage timing of one config on the GPU under a list of environment settings (development aid;
bench.py is the contract).

    python tools/quick_bench.py c3 10 "" "bwd_pl=0" "fwd_solo=0 bwd_order=0"
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
os.environ["F3DGS_PROFILE"] = "1"
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
settings = sys.argv[3:] or [""]
kw = dict(CONFIGS[cfg.split("@")[0]])
if "@" in cfg:          # "c2@8": config c2 with 8 feature channels
    kw["C"] = int(cfg.split("@")[1])
sc = make_scene(seed=0, yaw_deg=float(os.environ.get("F3DGS_BENCH_YAW", "0")), **kw)      # (a rotated view: F3DGS_BENCH_YAW=35)
dev = "cuda:0"
t = lambda x: x.to(dev)
st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                       t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"], t(sc["campos"]), False, False)
P = sc["P"]
L = dict(means3D=t(sc["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
         opacities=t(sc["opacities"]).requires_grad_(), shs=t(sc["shs"]).requires_grad_(),
         semantic_feature=t(sc["semantic_feature"]).requires_grad_(), scales=t(sc["scales"]).requires_grad_(),
         rotations=t(sc["rotations"]).requires_grad_())
gc, gf, gd = t(sc["dL_dcolor"]), t(sc["dL_dfeature"]), t(sc["dL_ddepth"])
r = dgr.GaussianRasterizer(st)


def step():
    color, feat, radii, depth = r(**L)
    torch.autograd.backward([color, feat, depth], [gc, gf, gd])
    for v in L.values():
        v.grad = None


for setting in settings:
    added = []
    _C.set_tile_band(0, 0)
    for kv in setting.split():
        k, v = kv.split("=")
        if k == "band":           # "band=9:18": tile rows [9, 18) only (f3dgs_set_tile_band)
            a, b = v.split(":")
            _C.set_tile_band(int(a), int(b))
            continue
        added.append((k, _C.get_option(k)))
        _C.set_option(k, int(v))
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    _C.profile_reset()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    prof = {n: m / max(1, c) for n, m, c in _C.profile_read()}
    print(f"[{setting or 'default'}] {ms:.3f} ms/step | " + " ".join(f"{n}={m:.3f}" for n, m in prof.items()), flush=True)
    for k, old in added:
        _C.set_option(k, old)
