import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import numpy as np, torch
import bench
from synth import make_scene, CONFIGS
sc = make_scene(seed=0, **CONFIGS["c3"])
for cull in ("0", "1"):
    from diff_gaussian_rasterization import _C
    _C.set_option("tile_cull", int(cull))
    t = lambda x: x.to("cuda:0"); e = torch.Tensor([])
    res = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), e, t(sc["semantic_feature"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, e, t(sc["viewmatrix"]), t(sc["projmatrix"]), sc["tanfovx"], sc["tanfovy"], 1080, 1920, t(sc["shs"]), 3, t(sc["campos"]), False, False)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(os.path.join(ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so"))
    lib.f3dgs_debug_read.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    nc = np.zeros(1920 * 1080, np.uint32)
    lib.f3dgs_debug_read(b"n_contrib", sc["P"], 32, res[0], 1920, 1080, res[5].data_ptr(), res[6].data_ptr(), res[7].data_ptr(), nc.ctypes.data_as(ctypes.c_void_p), nc.nbytes, None)
    cnt = np.zeros(16, np.uint32)
    lib.f3dgs_debug_read(b"counters", sc["P"], 32, res[0], 1920, 1080, res[5].data_ptr(), res[6].data_ptr(), res[7].data_ptr(), cnt.ctypes.data_as(ctypes.c_void_p), cnt.nbytes, None)
    bodies = int(((nc.astype(np.int64) + 63) // 64).sum())
    pad = np.zeros((68 * 16, 120 * 16), np.int64); pad[:1080, :1920] = nc.reshape(1080, 1920)
    q = pad.reshape(136, 8, 240, 8).max(axis=(1, 3))
    print("cull", cull, "list entries", int(cnt[0]), "ref N", int(cnt[1]), "sum n_contrib", int(nc.sum()), "bodies", bodies, "quad visits", int(q.sum()), "quad chunks", int(((q + 63) // 64).sum()))
