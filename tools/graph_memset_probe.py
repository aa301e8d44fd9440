"""Repro of the HIP-runtime defect DESIGN.md 3.2 describes: forward + backward through _C directly (no autograd) captured into a
graph; with hipMemsetAsync captured as memset nodes the raw outputs go bad from the SECOND replay on (the library now clears its
buffers with a kernel node inside a capture, so this prints "ok" for every replay).  python tools/graph_memset_probe.py [small|small-features|mid]"""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "feature-3dgs_amd"), os.path.join(ROOT, "tests")]
from test_gpu_sync_free import SCENES, _scene
from diff_gaussian_rasterization import _C

name = sys.argv[1] if len(sys.argv) > 1 else "small"
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
_C.set_option("bwd_bf16", 1)
sc = _scene(**SCENES[name])
dev = "cuda:0"
t = lambda x: x.to(dev)
e = torch.Tensor([])
S = {k: t(sc[k]) for k in ("bg", "means3D", "semantic_feature", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "shs", "campos",
                            "dL_dcolor", "dL_ddepth", "dL_dfeature")}
H, W = sc["image_height"], sc["image_width"]
names = ("means2D", "colors", "feature", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")
static = {}

def fn():
    res = _C.rasterize_gaussians(S["bg"], S["means3D"], e, S["semantic_feature"], S["opacities"], S["scales"], S["rotations"],
                                 sc["scale_modifier"], e, S["viewmatrix"], S["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W,
                                 S["shs"], sc["sh_degree"], S["campos"], False, False)
    n, color, feat, depth, radii, geom, binning, img = res
    gf = S["dL_dfeature"] if sc["C"] else torch.zeros(0, H, W, device=dev)
    out = _C.rasterize_gaussians_backward(S["bg"], S["means3D"], radii, e, S["semantic_feature"], S["scales"], S["rotations"],
                                          sc["scale_modifier"], e, S["viewmatrix"], S["projmatrix"], sc["tanfovx"], sc["tanfovy"],
                                          S["dL_dcolor"], gf, S["dL_ddepth"], S["shs"], sc["sh_degree"], S["campos"], geom, n, binning, img, False)
    for k, v in zip(names, out):
        if k not in static:
            static[k] = torch.zeros_like(v)
        if v.numel():
            static[k].copy_(v)
    if "radii" not in static:
        static["radii"] = torch.zeros_like(radii)
    static["radii"].copy_(radii)

def snap():
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in static.items()}

fn(); want = snap()
_C.set_option("sync_free", 1)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn(); fn()
s.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    fn()
_C.set_option("sync_free", 0)
torch.cuda.synchronize()
for i in range(4):
    for v in static.values():
        v.zero_()
    g.replay()
    got = snap()
    bad = []
    for k, v in want.items():
        if v.size == 0:
            continue
        d = np.abs(got[k].astype(np.float64) - v).max() / (np.abs(v).max() + 1e-30)
        if not (d < 1e-4):
            rows = np.where(np.abs(got[k].astype(np.float64) - v).reshape(v.shape[0], -1).max(1) > 1e-4 * np.abs(v).max())[0]
            bad.append((k, float(d), len(rows), rows[:6].tolist()))
    print("replay", i, "BAD" if bad else "ok", bad, flush=True)
for k in ("colors", "feature", "cov3D", "means2D", "opacity"):
    v = want[k].reshape(want[k].shape[0], -1); g_ = got[k].reshape(v.shape)
    if v.size == 0:
        continue
    rows = np.where(np.abs(g_ - v).max(1) > 1e-4 * np.abs(v).max())[0][:4]
    vis = want["radii"] > 0
    print(k, "visible", int(vis.sum()), "rows", rows.tolist())
    for r in rows:
        print("   row", r, "radius", want["radii"][r], "want", v[r][:4], "got", g_[r][:4], "ratio", (g_[r][:4] / (v[r][:4] + 1e-30)))
