"""Development aid (GPU): one training-style step - render, fused feature loss against a 360 x 480 ground truth (decoder
C -> 4C where the loss supports the width), L1 on the colour, backward - with the loss's gradient handed to the rasterizer's
backward dense (C,H,W) or at the loss's resolution (fused_feature_l1(lowres_grad=True)).

    python tools/lowres_step_bench.py c4 20
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
os.environ["F3DGS_PROFILE"] = "1"
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C
from feature_loss import fused_feature_l1

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = make_scene(seed=0, **CONFIGS[cfg])
dev = "cuda:0"
t = lambda x: x.to(dev)
st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                       t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"], t(sc["campos"]), False, False)
P, C = sc["P"], sc["C"]
L = dict(means3D=t(sc["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
         opacities=t(sc["opacities"]).requires_grad_(), shs=t(sc["shs"]).requires_grad_(),
         semantic_feature=t(sc["semantic_feature"]).requires_grad_(), scales=t(sc["scales"]).requires_grad_(),
         rotations=t(sc["rotations"]).requires_grad_())
r = dgr.GaussianRasterizer(st)
Hg, Wg = 360, 480
decoder = C in (32, 64, 128)
Cout = 4 * C if decoder else C
g = torch.Generator().manual_seed(0)
gt = torch.randn(Cout, Hg, Wg, generator=g).to(dev)
gt_rgb = torch.rand(3, sc["image_height"], sc["image_width"], generator=g).to(dev)
conv = torch.nn.Conv2d(C, Cout, kernel_size=1).to(dev) if decoder else None


def step(lowres):
    color, feat, _radii, _depth = r(**L)
    loss = (color - gt_rgb).abs().mean() + fused_feature_l1(feat, gt, conv.weight if decoder else None, conv.bias if decoder else None,
                                                           lowres_grad=lowres)
    loss.backward()
    for v in L.values():
        v.grad = None
    if decoder:
        conv.zero_grad()


print(f"{cfg}: P={P} C={C} -> {Cout} ({'decoder' if decoder else 'no decoder'}), image {sc['image_width']}x{sc['image_height']}, ground truth {Wg}x{Hg}")
for rep in range(2):
    for lowres in (False, True):
        for _ in range(5):
            step(lowres)
        torch.cuda.synchronize()
        _C.profile_reset()
        t0 = time.perf_counter()
        for _ in range(iters):
            step(lowres)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / iters
        prof = {n: m / max(1, c) for n, m, c in _C.profile_read()}
        print(f"[{'gradient at the loss resolution' if lowres else 'dense (C,H,W) gradient       '}] {ms:.3f} ms/step | render_bwd={prof.get('render_bwd', 0):.3f}", flush=True)
