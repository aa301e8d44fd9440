"""Quick per-stage timing of one config on the GPU (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("F3DGS_PROFILE", "1")
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kw = dict(CONFIGS[cfg]); 
t0 = time.time(); sc = make_scene(seed=0, **kw); print("scene gen %.1fs" % (time.time() - t0), flush=True)
dev = "cuda:0"
t = lambda x: x.to(dev)
st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                       t(sc["viewmatrix"]), t(sc["projmatrix"]), 3, t(sc["campos"]), False, False)
P = sc["P"]
L = dict(means3D=t(sc["means3D"]).requires_grad_(), means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
         opacities=t(sc["opacities"]).requires_grad_(), shs=t(sc["shs"]).requires_grad_(),
         semantic_feature=t(sc["semantic_feature"]).requires_grad_(), scales=t(sc["scales"]).requires_grad_(),
         rotations=t(sc["rotations"]).requires_grad_())
gc, gf, gd = t(sc["dL_dcolor"]), t(sc["dL_dfeature"]), t(sc["dL_ddepth"])
r = dgr.GaussianRasterizer(st)
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    color, feat, radii, depth = r(**L)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    fw = _C.last_stage_times()
    torch.autograd.backward([color, feat, depth], [gc, gf, gd])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    bw = _C.last_stage_times()
    print(f"iter {it}: fwd {1e3*(t1-t0):.3f} ms  bwd {1e3*(t2-t1):.3f} ms", flush=True)
    print("   fwd stages:", " ".join(f"{n}={m:.3f}" for n, m in fw))
    print("   bwd stages:", " ".join(f"{n}={m:.3f}" for n, m in bw), flush=True)
    for v in L.values(): v.grad = None
print("visible", int((radii > 0).sum()), "of", P)
