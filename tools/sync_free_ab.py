"""A/B of option sync_free on a c1-sized eager step (host-bound): alternating blocks in ONE process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "feature-3dgs_amd")]
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
sc = make_scene(seed=0, **CONFIGS[cfg])
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
for _ in range(50): step()
torch.cuda.synchronize()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
res = {0: [], 1: []}
for rep in range(6):
    for sf in (0, 1):
        _C.set_option("sync_free", sf)
        for _ in range(20): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N): step()
        torch.cuda.synchronize()
        res[sf].append(1e3 * (time.perf_counter() - t0) / N)
for sf in (0, 1):
    print(cfg, "sync_free", sf, " ".join(f"{x:.4f}" for x in res[sf]), "median", sorted(res[sf])[len(res[sf]) // 2])
