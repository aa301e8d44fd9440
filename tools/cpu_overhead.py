import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
sc = make_scene(seed=0, **CONFIGS["c1"])
dev = "cuda:0"; t = lambda x: x.to(dev)
st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                       t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"], t(sc["campos"]), False, False)
P = sc["P"]
L = dict(means3D=t(sc["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
         opacities=t(sc["opacities"]).requires_grad_(), shs=t(sc["shs"]).requires_grad_(),
         semantic_feature=t(sc["semantic_feature"]).requires_grad_(), scales=t(sc["scales"]).requires_grad_(),
         rotations=t(sc["rotations"]).requires_grad_())
gc, gf, gd = t(sc["dL_dcolor"]), t(sc["dL_dfeature"]), t(sc["dL_ddepth"])
r = dgr.GaussianRasterizer(st)
tf = tb = tz = 0.0
N = 300
for it in range(N + 20):
    if it == 20:
        torch.cuda.synchronize(); tf = tb = tz = 0.0; t_all = time.perf_counter()
    a = time.perf_counter()
    color, feat, radii, depth = r(**L)
    b = time.perf_counter()
    torch.autograd.backward([color, feat, depth], [gc, gf, gd])
    c = time.perf_counter()
    for v in L.values(): v.grad = None
    d = time.perf_counter()
    tf += b - a; tb += c - b; tz += d - c
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(f"c1 per step: total {1e3*tot/N:.3f} ms | host time in forward call {1e3*tf/N:.3f} (includes the count read-back wait), backward call {1e3*tb/N:.3f}, grad reset {1e3*tz/N:.3f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(100):
    color, feat, radii, depth = r(**L)
    torch.autograd.backward([color, feat, depth], [gc, gf, gd])
    for v in L.values(): v.grad = None
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
