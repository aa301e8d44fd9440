"""Development aid (GPU): print how the product and the reference-compiled checker differ on one scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import refutil as ru
from synth import make_scene
from util import precompute_optionals

P, W, H, C, Cref = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (10000, 256, 256, 0, 3)))
sc = precompute_optionals(make_scene(P=P, C=C, width=W, height=H, seed=1, scale_lo=0.005, scale_hi=0.08))
ref, prod = ru.load_ref(Cref), ru.product_module()
d_ref, d_prod = ru.device_inputs(sc, Cref), ru.device_inputs(sc, C)
f_ref, f_prod = ru.raw_forward(ref, sc, d_ref), ru.raw_forward(prod, sc, d_prod)
print("num_rendered", int(f_ref[0]), int(f_prod[0]))
print("buffer sizes ref", [f_ref[i].numel() for i in (5, 6, 7)], "ptr%128", [f_ref[i].data_ptr() % 128 for i in (5, 6, 7)])
r_ref, r_prod = f_ref[4].cpu().numpy(), f_prod[4].cpu().numpy()
print("radii mismatch", int((r_ref != r_prod).sum()), "vis", int((r_ref > 0).sum()), int((r_prod > 0).sum()))
for i, k in ((1, "color"), (3, "depth")):
    a, b = f_ref[i].cpu().numpy(), f_prod[i].cpu().numpy()
    print(k, "max abs diff", float(np.abs(a - b).max()), "ref range", float(a.min()), float(a.max()))
ir, ip = ru.ref_image_state(f_ref, W, H), ru.product_image_state(sc, f_prod)
print("final_T ref", ir["final_T"][:8], "prod", ip["final_T"][:8])
print("n_contrib ref", ir["n_contrib"][:8], "prod", ip["n_contrib"][:8])
print("n_contrib equal", int((ir["n_contrib"] == ip["n_contrib"]).sum()), "of", W * H)
print("final_T max diff", float(np.abs(ir["final_T"] - ip["final_T"]).max()))
fl = ru.flip_pixels(ir, ip)
print("flips", int(fl.sum()))
idx = np.nonzero(fl)[0][:10]
for j in idx:
    print("  px", j, "nc", ir["n_contrib"][j], ip["n_contrib"][j], "T", ir["final_T"][j], ip["final_T"][j])
