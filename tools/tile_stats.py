import os, sys, ctypes, numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import torch
from synth import make_scene, CONFIGS
import diff_gaussian_rasterization as dgr
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = make_scene(seed=0, **CONFIGS[cfg]); dev = "cuda:0"; t = lambda x: x.to(dev)
st = dgr.GaussianRasterizationSettings(sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], t(sc["bg"]), 1.0,
                                       t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["sh_degree"], t(sc["campos"]), False, False)
P = sc["P"]
args = (st.bg, t(sc["means3D"]), torch.Tensor([]).to(dev), t(sc["semantic_feature"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0,
        torch.Tensor([]).to(dev), st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height, st.image_width, t(sc["shs"]), st.sh_degree, st.campos, False, False)
res = dgr._C.rasterize_gaussians(*args)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(ROOT, "feature-3dgs_amd/csrc/libf3dgs_hip.so"))
H, W = sc["image_height"], sc["image_width"]
nc = np.zeros(H * W, np.uint32)
lib.f3dgs_debug_read(b"n_contrib", P, sc["C"], res[0], W, H, ctypes.c_void_p(res[5].data_ptr()), ctypes.c_void_p(res[6].data_ptr()), ctypes.c_void_p(res[7].data_ptr()),
                     nc.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nc.nbytes), None)
nc = nc.reshape(H, W)
gy, gx = (H + 15) // 16, (W + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.uint32); pad[:H, :W] = nc
tl = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).ravel()
ch = (tl + 15) // 16
print(cfg, "tiles", tl.size, "max walk", tl.max(), "p99", np.percentile(tl, 99), "p90", np.percentile(tl, 90), "mean", tl.mean())
print("chunks: total", ch.sum(), "max", ch.max(), "mean", ch.mean(), " total/1024 slots =", ch.sum() / 1024.0, "-> critical path / balanced = %.2f" % (ch.max() / (ch.sum() / 1024.0)))
print("histogram of chunks per tile:", np.bincount(np.minimum(ch, 200) // 10))
