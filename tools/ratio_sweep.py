"""Development aid (needs a GPU): what the two-term bf16 contractions of the blend backward cost the gradients as a function of
the Gaussians' axis ratio.  Scenes of the synthetic family whose scales are (s, s / r, s / r) ["needle"] or
(s, s, s / r) ["disc"]; per ratio r the worst element of |g_bf16 - g_fp32| in units of the north-star bound 1e-3 |g| + 1e-5 max|g|,
next to the same distance between two runs of the exact-fp32 shape (the order of the atomic sums).

    python tools/ratio_sweep.py > profiles/r06_ratio_sweep.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from synth import make_scene
from util import run_hip, set_option

KEYS = ("dL_drotations", "dL_dscales", "dL_dmeans3D", "dL_dopacity", "dL_dsemantic_feature")


def dist(a, b):
    b = b.astype(np.float64)
    return float((np.abs(a - b) / (1e-3 * np.abs(b) + 1e-5 * (np.abs(b).max() + 1e-30))).max())


print("worst element of |g_bf16 - g_fp32| / (1e-3 |g| + 1e-5 max|g|)  [same for fp32 vs a second fp32 run]; 20000 Gaussians, 320x200, C = 32")
for shape in ("needle", "disc"):
    for r in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024):
        sc = make_scene(P=20000, C=32, width=320, height=200, seed=71, with_depth_grad=True, scale_lo=0.02, scale_hi=0.2)
        g = torch.Generator().manual_seed(5)
        s = sc["scales"][:, :1]
        if shape == "needle":
            sc["scales"] = torch.cat([s, s / r, s / r], dim=1).contiguous()
        else:
            sc["scales"] = torch.cat([s, s, s / r], dim=1).contiguous()
        set_option("bwd_bf16", 1)
        _o, g1 = run_hip(sc)
        set_option("bwd_bf16", 0)
        _o, g0 = run_hip(sc)
        _o, g0b = run_hip(sc)
        set_option("bwd_bf16", -1)
        print(f"{shape:6s} ratio {r:5d}: " + "  ".join(f"{k[3:]} {dist(g1[k], g0[k]):7.2f} [{dist(g0b[k], g0[k]):5.2f}]" for k in KEYS), flush=True)
