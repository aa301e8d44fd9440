"""GPU box: stability of the op over many steps on CHANGING views (buffers regrow, huge splats near the camera plane come and
go: the cooperative path of the emit kernel, windowed staging, every channel window).   python tools/soak_views.py [config] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import torch
import diff_gaussian_rasterization as dgr
from synth import make_scene, make_camera, CONFIGS

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sc = make_scene(seed=0, **CONFIGS[cfg])
dev = torch.device("cuda:0")
t = lambda x: x.to(dev)
W, H, P = sc["image_width"], sc["image_height"], sc["P"]
leaves = dict(means3D=t(sc["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
              opacities=t(sc["opacities"]).requires_grad_(), shs=t(sc["shs"]).requires_grad_(),
              semantic_feature=t(sc["semantic_feature"]).requires_grad_(), scales=t(sc["scales"]).requires_grad_(),
              rotations=t(sc["rotations"]).requires_grad_())
ups = [t(sc["dL_dcolor"]), t(sc["dL_dfeature"]), t(sc["dL_ddepth"])]
rasts = []
for yaw in (0.0, 7.0, 15.0, 25.0, 35.0, 50.0, 75.0, -40.0):
    cam = make_camera(W, H, yaw_deg=yaw)
    rasts.append(dgr.GaussianRasterizer(dgr.GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], t(sc["bg"]), 1.0, t(cam["viewmatrix"]),
                                                                         t(cam["projmatrix"]), sc["sh_degree"], t(cam["campos"]), False, False)))
first = {}
m0 = None
t0 = time.perf_counter()
for i in range(steps):
    for v in leaves.values():
        v.grad = None
    j = i % len(rasts)
    color, feat, radii, depth = rasts[j](**leaves)
    torch.autograd.backward([color, feat, depth], ups)
    if i < len(rasts):
        first[j] = (color.detach().clone(), feat.detach().clone(), leaves["means3D"].grad.abs().sum().item())
    elif i >= steps - len(rasts):      # the same views at the end: bit-identical images, gradients up to the order of the atomic sums
        c0, f0, g0 = first[j]
        assert torch.equal(color, c0) and torch.equal(feat, f0), f"view {j}: images changed between step {j} and step {i}"
        g1 = leaves["means3D"].grad.abs().sum().item()
        assert abs(g1 - g0) <= 1e-4 * abs(g0), (j, g0, g1)
    if i == 2 * len(rasts):
        torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
torch.cuda.synchronize()
# a fingerprint of the first pass's images (exact: integer sum of the float bits) - equal between runs under different options
# (F3DGS_SYNC_FREE=1: the provision of a view comes from the previous view's count, so rotating views exercise the retry)
fp = sum(int(first[j][0].view(torch.int32).to(torch.int64).sum()) + int(first[j][1].view(torch.int32).to(torch.int64).sum()) for j in sorted(first))
print("image fingerprint", fp)
ok = all(bool(torch.isfinite(v.grad).all()) for v in leaves.values() if v.grad is not None)
print(f"{cfg}: {steps} steps over {len(rasts)} views, {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step; finite gradients: {ok}; "
      f"allocated MB at step {2 * len(rasts)} / end: {m0 >> 20} / {torch.cuda.memory_allocated() >> 20}; images of every view bit-identical at the end")
assert ok
