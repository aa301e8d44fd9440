"""Development aid (needs a GPU): work statistics of the blend kernels at config c3 on a sample of tiles.

How many (instance, pixel-block) evaluations survive the footprint tests at half-tile / quadrant / pixel-row
granularity, and how many (instance, pixel) pairs really blend - i.e. the lane utilisation of the forward
(lanes = pixels) and backward (lanes = instances) bodies."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import numpy as np
import torch
from synth import make_scene, CONFIGS

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = make_scene(seed=0, **CONFIGS[cfg])
W, H, P, C = sc["image_width"], sc["image_height"], sc["P"], CONFIGS[cfg]["C"]
from diff_gaussian_rasterization import _C

t = lambda x: x.to("cuda:0")
e = torch.Tensor([])
res = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), e, t(sc["semantic_feature"]), t(sc["opacities"]),
                             t(sc["scales"]), t(sc["rotations"]), 1.0, e, t(sc["viewmatrix"]), t(sc["projmatrix"]),
                             sc["tanfovx"], sc["tanfovy"], H, W, t(sc["shs"]), sc["sh_degree"], t(sc["campos"]),
                             False, False)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so"))
lib.f3dgs_debug_read.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]


def read(what, n, dtype):
    a = np.zeros(n, dtype)
    lib.f3dgs_debug_read(what.encode(), P, C, res[0], W, H, res[5].data_ptr(), res[6].data_ptr(), res[7].data_ptr(),
                         a.ctypes.data_as(ctypes.c_void_p), a.nbytes, None)
    return a


cnt = read("counters", 16, np.uint32)
N = int(cnt[0])
gx, gy = (W + 15) // 16, (H + 15) // 16
nc = read("n_contrib", W * H, np.uint32).reshape(H, W)
ranges = read("ranges", 2 * gx * gy, np.uint32).reshape(-1, 2)
plist = read("point_list", N, np.uint32)
rec = read("rec", 12 * P, np.float32).reshape(P, 12)
dev = "cuda:0"
rec_t = torch.from_numpy(rec).to(dev)
rng = np.random.default_rng(0)
tiles = rng.choice(gx * gy, size=min(600, gx * gy), replace=False)


def rect_min_q(mx, my, a, b, c, x0, x1, y0, y1):
    """Minimum of the quadratic form over the pixel-centre rectangle (same construction as rect_hit)."""
    xlo, xhi, ylo, yhi = mx - x1, mx - x0, my - y1, my - y0
    inside = (xlo <= 0) & (xhi >= 0) & (ylo <= 0) & (yhi >= 0)
    qf = lambda dx, dy: a * dx * dx + 2 * b * dx * dy + c * dy * dy
    cl = lambda v, lo, hi: torch.minimum(hi, torch.maximum(lo, v))
    best = qf(xlo, cl(-(b * xlo) / c, ylo, yhi))
    best = torch.minimum(best, qf(xhi, cl(-(b * xhi) / c, ylo, yhi)))
    best = torch.minimum(best, qf(cl(-(b * ylo) / a, xlo, xhi), ylo))
    best = torch.minimum(best, qf(cl(-(b * yhi) / a, xlo, xhi), yhi))
    return torch.where(inside, torch.zeros_like(best), best)


tot = dict(list=0, half=0, quad=0, row=0, pairs_fwd=0, quad_b=0, bodies_b=0, pairs_b=0, fwd_slots=0, half_entries=0)
for tile in tiles:
    tx, ty = tile % gx, tile // gx
    r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
    ncs = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
    if r1 <= r0 or ncs.size == 0 or ncs.max() == 0:
        continue
    ids = torch.from_numpy(plist[r0:r1].astype(np.int64)).to(dev)
    r = rec_t[ids]
    mx, my, a, b, c, op = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5]
    thr = 2 * torch.log(255 * op.clamp_min(1e-9))
    pos = torch.arange(r1 - r0, device=dev)
    hh, ww = ncs.shape
    ncs_t = torch.from_numpy(ncs.astype(np.int64)).to(dev)
    ys, xs = torch.meshgrid(torch.arange(hh, device=dev), torch.arange(ww, device=dev), indexing="ij")
    px = (tx * 16 + xs).float().reshape(-1)
    py = (ty * 16 + ys).float().reshape(-1)
    dx = mx[:, None] - px[None, :]
    dy = my[:, None] - py[None, :]
    power = -0.5 * (a[:, None] * dx * dx + c[:, None] * dy * dy) - b[:, None] * dx * dy
    alpha = torch.minimum(torch.tensor(0.99, device=dev), op[:, None] * torch.exp(power))
    blend = (power <= 0) & (alpha >= 1 / 255) & (pos[:, None] < ncs_t.reshape(-1)[None, :])   # pairs the backward visits
    tot["pairs_b"] += int(blend.sum())
    tot["list"] += int((pos < int(ncs.max())).sum())
    # forward: a wave = two horizontally adjacent quadrants (16x8 block), walks until all its pixels are done
    for half in range(2):
        y0, y1 = ty * 16 + 8 * half, ty * 16 + 8 * half + 7
        sub = ncs[8 * half:8 * half + 8, :]
        if sub.size == 0:
            continue
        lim = int(sub.max())          # (the forward really stops a little later: when the last pixel saturates)
        act = pos < lim
        qh = rect_min_q(mx, my, a, b, c, float(tx * 16), float(tx * 16 + 15), float(y0), float(y1)) <= thr
        tot["half_entries"] += int(act.sum())
        tot["half"] += int((act & qh).sum())
        for qx in range(2):
            x0 = tx * 16 + 8 * qx
            qq = rect_min_q(mx, my, a, b, c, float(x0), float(x0 + 7), float(y0), float(y1)) <= thr
            tot["quad"] += int((act & qq).sum())
            sq = ncs[8 * half:8 * half + 8, 8 * qx:8 * qx + 8]
            if sq.size == 0 or sq.max() == 0:
                continue
            actb = pos < int(sq.max())
            nb_inst = int((actb & qq).sum())
            tot["quad_b"] += nb_inst
            # backward bodies: one per (chunk of 64 surviving instances, live pixel); approximate with the mean
            # number of live pixels over the chunk positions
            surv = torch.nonzero(actb & qq).reshape(-1)
            sq_t = torch.from_numpy(sq.astype(np.int64)).to(dev).reshape(-1)
            # the kernel fills chunks walking BACK to FRONT: the partially filled chunk is the front-most one
            rs = torch.flip(surv, dims=[0])
            for k in range(0, nb_inst, 64):
                grp = rs[k:k + 64]
                pmin = int(grp.min())
                nb_ = int((sq_t > pmin).sum())
                tot["bodies_b"] += nb_
                key = "bodies_le16" if len(grp) <= 16 else ("bodies_le32" if len(grp) <= 32 else "bodies_gt32")
                tot[key] = tot.get(key, 0) + nb_
            # the same with chunks of 32 instances against two pixel halves (a body then costs half)
            for k in range(0, nb_inst, 32):
                grp = rs[k:k + 32]
                nb_ = int((sq_t > int(grp.min())).sum())
                tot["half_bodies_cap32"] = tot.get("half_bodies_cap32", 0) + 8 * ((nb_ + 7) // 8)     # trips of 8 pixels
            for k in range(0, nb_inst, 64):
                grp = rs[k:k + 64]
                nb_ = int((sq_t > int(grp.min())).sum())
                tot["bodies_trips_cap64"] = tot.get("bodies_trips_cap64", 0) + 4 * ((nb_ + 3) // 4)   # trips of 4 pixels
            for ry in range(8):
                qr = rect_min_q(mx, my, a, b, c, float(x0), float(x0 + 7), float(y0 + ry), float(y0 + ry)) <= thr
                tot["row"] += int((actb & qr).sum())

print({k: v for k, v in tot.items()})
print("forward : entries walked per half-tile wave %d, surviving 16x8 test %.3f, quadrant evaluations useful %.3f "
      "(quadrant hits / 2 x half hits)" % (tot["half_entries"], tot["half"] / max(1, tot["half_entries"]),
                                           tot["quad"] / max(1, 2 * tot["half"])))
print("backward: instances after the quadrant test %d, bodies %d, lane utilisation %.3f (blending pairs / 64 x bodies); "
      "row-level hits / 8 x quadrant hits = %.3f" % (tot["quad_b"], tot["bodies_b"],
                                                      tot["pairs_b"] / max(1, 64 * tot["bodies_b"]),
                                                      tot["row"] / max(1, 8 * tot["quad_b"])))
