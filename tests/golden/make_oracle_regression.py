"""Freezes outputs of the CPU oracle for three small scenes (tests/golden/oracle_regression.npz) so that a
later edit of the oracle cannot silently move the target the HIP kernels are checked against.
    python tests/golden/make_oracle_regression.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

SCENES = {
    "a": dict(P=400, C=3, width=64, height=48, seed=101, scale_lo=0.02, scale_hi=0.2, with_depth_grad=True),
    "b": dict(P=900, C=8, width=80, height=40, seed=102, scale_lo=0.005, scale_hi=0.1),
    "c": dict(P=300, C=0, width=33, height=31, seed=103, scale_lo=0.05, scale_hi=0.5, sh_degree=1),
}


def run(name):
    import torch
    from synth import make_scene
    from util import run_oracle
    sc = make_scene(**SCENES[name])
    if name == "a":
        sc["bg"] = torch.tensor([0.1, 0.5, 0.9])
    o, out, g = run_oracle(sc)
    res = {f"{name}_num_rendered": np.int64(out["num_rendered"]), f"{name}_radii": out["radii"]}
    for k in ("color", "feature_map", "depth"):
        res[f"{name}_{k}"] = out[k]
    res[f"{name}_n_contrib"] = o.read("n_contrib")
    res[f"{name}_point_list"] = o.read("point_list")
    for k, v in g.items():
        res[f"{name}_{k}"] = v
    return res


if __name__ == "__main__":
    allres = {}
    for n in SCENES:
        allres.update(run(n))
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), **allres)
    print("wrote oracle_regression.npz with", len(allres), "arrays")
