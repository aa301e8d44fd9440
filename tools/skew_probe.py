"""Does a vertically skewed scene (most splats in the lower part of the image) cost the blend kernels more per unit of work than a
uniform one?  Every XCD owns a contiguous eighth of the tile rows (xcd_remap)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "feature-3dgs_amd")]
os.environ["F3DGS_PROFILE"] = "1"
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C

dev = "cuda:0"
t = lambda x: x.to(dev)
for skew in (0.0, 0.5, 0.8):
    sc = make_scene(seed=0, **CONFIGS["c3"])
    m = sc["means3D"].clone()
    # a fraction `skew` of the Gaussians is moved into the lowest fifth of the view frustum (y grows downwards in the image)
    P = m.shape[0]
    g = torch.Generator().manual_seed(9)
    pick = torch.rand(P, generator=g) < skew
    z = m[:, 2]
    tany = sc["tanfovy"]
    ynew = z * tany * (0.6 + 0.4 * torch.rand(P, generator=g))      # 60 .. 100 % of the half-height below the axis
    m[pick, 1] = ynew[pick]
    sc["means3D"] = m
    st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                           t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"], t(sc["campos"]), False, False)
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
        return color

    for _ in range(5):
        color = step()
    torch.cuda.synchronize()
    # work: sum over tiles of the longest walk (n_contrib), by tile row
    res = _C.rasterize_gaussians(t(sc["bg"]), L["means3D"].detach(), torch.Tensor([]), L["semantic_feature"].detach(), L["opacities"].detach(),
                                 L["scales"].detach(), L["rotations"].detach(), 1.0, torch.Tensor([]), t(sc["viewmatrix"]), t(sc["projmatrix"]),
                                 sc["tanfovx"], sc["tanfovy"], sc["image_height"], sc["image_width"], L["shs"].detach(), sc["sh_degree"], t(sc["campos"]), False, False)
    for setting in ("", "bwd_order=0"):
        for kv in setting.split():
            k, v = kv.split("="); _C.set_option(k, int(v))
        for _ in range(3):
            step()
        torch.cuda.synchronize(); _C.profile_reset()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        prof = {n: m_ / max(1, c) for n, m_, c in _C.profile_read()}
        print(f"skew {skew} [{setting or 'default'}] entries {_C.forward_counts()[0]}  render_fwd {prof['render_fwd']:.3f}  render_bwd {prof['render_bwd']:.3f}  "
              f"fwd us per M entries {1e3 * prof['render_fwd'] / (_C.forward_counts()[0] / 1e6):.1f}  bwd {1e3 * prof['render_bwd'] / (_C.forward_counts()[0] / 1e6):.1f}", flush=True)
        _C.set_option("bwd_order", 1)
