"""GPU box: what do the fp64 truths say about the pixels / gradient elements where product and reference differ by more than
the strict bars?  (development probe behind tests/test_gpu_adjudicate.py)

    python tools/adjudicate_probe.py c3 0        # config, yaw index
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import refutil as ru, adjudicate as adj
from synth import CONFIGS, make_scene, make_camera
from util import set_option

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
yaw = float(sys.argv[2]) * 5.0 if len(sys.argv) > 2 else 0.0
DEV = "cuda:0"
if cfg.startswith("harsh:"):      # harsh:<kind>:<P>:<W>:<H>:<C>:<seed>[:depth]   (tests/util.py: harsh_scene)
    from util import harsh_scene
    f_ = cfg.split(":")
    scene = harsh_scene(f_[1], P=int(f_[2]), width=int(f_[3]), height=int(f_[4]), C=int(f_[5]), seed=int(f_[6]), with_depth_grad=len(f_) > 7)
else:
    scene = make_scene(seed=0, **CONFIGS[cfg])
    scene.update(make_camera(scene["image_width"], scene["image_height"], yaw_deg=yaw))
W, H, P, C = scene["image_width"], scene["image_height"], scene["P"], scene["C"]
mode = [a for a in sys.argv[3:] if not a.startswith("--")]
mode = mode[0] if mode else "default"      # default: product vs contracted reference; strict: product vs -ffp-contract=off reference; refs: strict reference in the product's seat
ref = ru.load_ref(C, strict=(mode == "strict"))
prod = ru.load_ref(C, strict=True) if mode == "refs" else ru.product_module()
d = ru.device_inputs(scene, C, DEV)
f_ref, f_prod = ru.raw_forward(ref, scene, d), ru.raw_forward(prod, scene, d)
if mode == "refs":
    f_prod0 = f_prod
    img_ref, img_prod = ru.ref_image_state(f_ref, W, H), ru.ref_image_state(f_prod, W, H)
else:
    old = set_option("tile_cull", 0); f_prod0 = ru.raw_forward(prod, scene, d); set_option("tile_cull", old)
    img_ref, img_prod = ru.ref_image_state(f_ref, W, H), ru.product_image_state(scene, f_prod0)
print("mode", mode, "| radii differing:", int((f_ref[4] != f_prod[4]).sum()), "| num_rendered", int(f_ref[0]), int(f_prod[0]))
flips = ru.flip_pixels(img_ref, img_prod)
dT = np.abs(img_ref["final_T"] - img_prod["final_T"])
over = flips | (dT > 1e-5)
errs = {}
for i, k in ((1, "color"), (2, "feature"), (3, "depth")):
    e = (f_ref[i] - f_prod[i]).abs().amax(dim=0).reshape(-1).cpu().numpy()
    errs[k] = e
    over |= e > 1e-4
pix = np.nonzero(over)[0]
print(f"{cfg} yaw {yaw}: {int(flips.sum())} proven flips, {int((dT > 1e-5).sum())} final_T over 1e-5, "
      + ", ".join(f"{k} over 1e-4: {int((e > 1e-4).sum())}" for k, e in errs.items()) + f"; adjudicating {len(pix)} pixels")
if len(pix):
    done, truths = adj.forward_truth(scene, pix, dev=DEV)
    sel = pix[done]
    ys, xs = torch.from_numpy(sel // W).to(DEV), torch.from_numpy(sel % W).to(DEV)
    take = lambda f: dict(color=f[1][:, ys, xs].t().cpu().numpy(), feature=f[2][:, ys, xs].t().cpu().numpy(), depth=f[3][0, ys, xs].cpu().numpy())
    pv_, rv_ = take(f_prod), take(f_ref)
    pv_["final_T"], rv_["final_T"] = img_prod["final_T"][sel], img_ref["final_T"][sel]
    bars = dict(color=1e-4, feature=1e-4, depth=1e-4, final_T=1e-5)
    ok, ep, er, bl = adj.forward_verdict(truths, pv_, rv_, bars)
    print(f"adjudicated {len(sel)} pixels: ok {int(ok.sum())}, borderline {int(bl.sum())}; e_prod/bar: median {np.median(ep):.2f} max {ep.max():.2f}; "
          f"e_ref/bar: median {np.median(er):.2f} max {er.max():.2f}")
    # is the product's side of a borderline decision biased?  (exp2 on a pre-scaled conic against the
    # reference's exp): sign of (value - fp64 value) of the final transmittance and of n_contrib on the adjudicated pixels
    dTp = np.asarray(pv_["final_T"], np.float64) - truths[0]["final_T"]
    dTr = np.asarray(rv_["final_T"], np.float64) - truths[0]["final_T"]
    ncp = img_prod["n_contrib"][sel].astype(np.int64) - truths[0]["n_contrib"]
    ncr = img_ref["n_contrib"][sel].astype(np.int64) - truths[0]["n_contrib"]
    big_p, big_r = np.abs(dTp) > 1e-5, np.abs(dTr) > 1e-5
    print(f"decision bias on the {len(sel)} adjudicated pixels: final_T off by more than 1e-5 - product {int(big_p.sum())} ({int((dTp[big_p] > 0).sum())} too "
          f"transparent = an entry dropped, {int((dTp[big_p] < 0).sum())} too opaque = an entry kept), reference {int(big_r.sum())} "
          f"({int((dTr[big_r] > 0).sum())} / {int((dTr[big_r] < 0).sum())}); n_contrib against fp64: product {int((ncp > 0).sum())} longer / "
          f"{int((ncp < 0).sum())} shorter, reference {int((ncr > 0).sum())} / {int((ncr < 0).sum())}")
    for j in np.argsort(-ep)[:12]:
        for k_ in bars:
            dp_ = np.abs(np.asarray(pv_[k_], np.float64)[j] - truths[0][k_][j]).max() / bars[k_]
            dr_ = np.abs(np.asarray(rv_[k_], np.float64)[j] - truths[0][k_][j]).max() / bars[k_]
            if max(dp_, dr_) > 0.3:
                print(f"      {k_}: prod {dp_:.2f} ref {dr_:.2f} bars from the fp64 value")
        print(f"  pixel {sel[j]} flip={bool(flips[sel[j]])} nc ref/prod/truth {img_ref['n_contrib'][sel[j]]}/{img_prod['n_contrib'][sel[j]]}/{truths[0]['n_contrib'][j]} "
              f"e_prod {ep[j]:.2f} e_ref {er[j]:.2f} ok={bool(ok[j])} borderline={bool(bl[j])} T ref/prod/truth {rv_['final_T'][j]:.6g}/{pv_['final_T'][j]:.6g}/{truths[0]['final_T'][j]:.6g}")
# gradients
keep = torch.from_numpy((~over).reshape(1, H, W)).to(DEV)
up = (d["dL_dcolor"] * keep, d["dL_dfeature"] * keep, d["dL_ddepth"] * keep)
g_ref = ru.raw_backward(ref, scene, d, f_ref, *up)
g_prod = ru.raw_backward(prod, scene, d, f_prod, *up)
names = {"dL_dmeans3D": "means3D", "dL_dmeans2D": "means2D", "dL_dopacity": "opacities", "dL_dsh": "shs", "dL_dscales": "scales",
         "dL_drotations": "rotations", "dL_dsemantic_feature": "semantic_feature"}
bad = {}
for k, leaf in names.items():
    a, b = g_ref[k].double(), g_prod[k].double()
    scale = float(a.abs().max()) + 1e-30
    ratio = ((b - a).abs() / (1e-3 * a.abs() + 1e-5 * scale)).reshape(P, -1).amax(dim=1)
    idx = torch.nonzero(ratio > 1.0).flatten().cpu().numpy()
    print(f"{k}: worst {float(ratio.max()):.2f}, {len(idx)} Gaussians over the bound")
    for i in idx[:8]:
        bad.setdefault(int(i), []).append(k)
if bad:
    ids = np.array(sorted(bad))
    tr = adj.gradient_truth(scene, ids, tuple(u.cpu() for u in up), dev=DEV)
    if tr is None:
        print("rectangles too large to adjudicate")
    else:
        for n, i in enumerate(ids):
            for k in bad[int(i)]:
                t = tr[names[k]][n].reshape(-1)
                a = g_ref[k][i].reshape(-1).double().cpu().numpy(); b = g_prod[k][i].reshape(-1).double().cpu().numpy()
                if k == "dL_dmeans2D":
                    t = np.concatenate([t[:2], [0.0]])[:len(a)]
                scale = float(g_ref[k].abs().max())
                bound = 1e-3 * np.abs(t) + 1e-5 * scale
                print(f"  gaussian {i} {k}: e_prod/bound {np.max(np.abs(b - t) / bound):.2f}  e_ref/bound {np.max(np.abs(a - t) / bound):.2f}  "
                      f"prod-ref/bound {np.max(np.abs(b - a) / bound):.2f} | radius {int(f_ref[4][i])} scales {scene['scales'][i].numpy()} opacity {float(scene['opacities'][i]):.4f} "
                      f"truth {t} ref {a} prod {b}")
        # the pieces: dL_dmeans2D (blend backward) and dL_dcov3D (cov2D backward, rasterize_points.cu:199) of the same Gaussians
        for k in ("dL_dmeans2D", "dL_dcov3D", "dL_dopacity"):
            a, b = g_ref[k].double(), g_prod[k].double()
            scale = float(a.abs().max()) + 1e-30
            for i in ids[:8]:
                r_ = ((b[i] - a[i]).abs() / (1e-3 * a[i].abs() + 1e-5 * scale)).max()
                print(f"    gaussian {i} {k}: prod-ref/bound {float(r_):.2f}  ref {a[i].cpu().numpy().reshape(-1)} prod {b[i].cpu().numpy().reshape(-1)}")

# ---- which per-Gaussian state entries explain the continuous (non-flip) deviations?
if len(pix) and "--state" in sys.argv:
    nf = [j for j in np.argsort(-ep) if not flips[sel[j]]][:3]
    from oracle import torch_oracle
    for j in nf:
        p_ = int(sel[j]); y_, x_ = p_ // W, p_ % W
        gx = (W + 15) // 16
        t = (y_ // 16) * gx + x_ // 16
        pl = ru.ref_point_list(f_ref); rg = ru.ref_image_state(f_ref, W, H)["ranges"].reshape(-1, 2)
        ids = pl[rg[t, 0]:rg[t, 0] + int(img_ref["n_contrib"][p_])].astype(np.int64)
        geo = ru.ref_geometry_state(f_ref, P, C, want={"means2D", "conic_opacity"})
        rec = ru.product_read("rec", scene, f_prod, np.float32, P * 12).reshape(P, 12)
        # fp64 state of these Gaussians
        sub = {k: (v[ids] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == P else v) for k, v in scene.items()}
        con_r = geo["conic_opacity"].reshape(P, 4)[ids, :3].astype(np.float64); m_r = geo["means2D"].reshape(P, 2)[ids].astype(np.float64)
        con_p = rec[ids, 2:5].astype(np.float64); m_p = rec[ids, 0:2].astype(np.float64)
        dx_r, dy_r = m_r[:, 0] - x_, m_r[:, 1] - y_
        dx_p, dy_p = m_p[:, 0] - x_, m_p[:, 1] - y_
        pw_r = -0.5 * (con_r[:, 0] * dx_r ** 2 + con_r[:, 2] * dy_r ** 2) - con_r[:, 1] * dx_r * dy_r
        pw_p = -0.5 * (con_p[:, 0] * dx_p ** 2 + con_p[:, 2] * dy_p ** 2) - con_p[:, 1] * dx_p * dy_p
        w = np.argsort(-np.abs(pw_r - pw_p))[:4]
        print(f"pixel {p_} ({x_},{y_}) tile {t}: {len(ids)} contributors; largest |power_ref_state - power_prod_state| (fp64 eval of the fp32 states):")
        for k in w:
            print(f"   gaussian {ids[k]}: power ref-state {pw_r[k]:.6f} prod-state {pw_p[k]:.6f}; conic ref {con_r[k]} prod {con_p[k]}; mean ref {m_r[k]} prod {m_p[k]}; scales {scene['scales'][ids[k]].numpy()}")
