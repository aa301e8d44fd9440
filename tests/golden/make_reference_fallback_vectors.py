"""Generates tests/golden/reference_fallbacks.npz by IMPORTING the reference's own Python code.

The reference has no tests or golden vectors for the rasterizer path.  The only
reference-owned restatements of any arithmetic on the path are its two Python fall-backs:
  * SH -> RGB:           utils/sh_utils.py:57-112 `eval_sh` + clamp_min(+0.5) (gaussian_renderer/__init__.py:230-234)
  * scale/rot -> cov3D:  utils/general_utils.py:64-110 `build_rotation` / `build_scaling_rotation` /
                         `strip_symmetric` (used by scene/gaussian_model.py:27-31)
This script runs them (here, in the build container - /root/reference does not exist on the GPU box) on
seeded inputs and freezes inputs + outputs.  general_utils hard-wires device="cuda"; torch.zeros is
patched to drop the device argument for the duration of the import/call.

    python tests/golden/make_reference_fallback_vectors.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh  # noqa: E402  (reference code)
    import utils.general_utils as gu  # noqa: E402  (reference code)

    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    g = torch.Generator().manual_seed(1234)
    P = 257
    out = {}
    # ---- SH -> RGB for every degree ----------------------------------------------------------------
    means = torch.randn(P, 3, generator=g) * 3
    campos = torch.tensor([0.3, -0.2, 0.1])
    shs = torch.randn(P, 16, 3, generator=g) * 2.5   # large enough that the max(0, .) clamp bites at every degree
    d = means - campos[None]
    dn = d / d.norm(dim=1, keepdim=True)
    out["sh_means"], out["sh_campos"], out["sh_coeffs"] = means.numpy(), campos.numpy(), shs.numpy()
    for deg in range(4):
        rgb = eval_sh(deg, shs.transpose(1, 2), dn)          # reference expects (P, 3, M)
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()
    # ---- scale / rotation -> 3D covariance (6 unique entries) --------------------------------------------
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.7 - 2.0)
    rot = torch.randn(P, 4, generator=g)
    rot_n = rot / rot.norm(dim=1, keepdim=True)             # the caller passes normalised quaternions
    torch.zeros = cpu_zeros
    try:
        for mod in (1.0, 0.6):
            L = gu.build_scaling_rotation(mod * scales, rot_n)
            cov = L @ L.transpose(1, 2)
            out[f"cov6_mod{mod}"] = gu.strip_symmetric(cov).numpy()
    finally:
        torch.zeros = real_zeros
    out["cov_scales"], out["cov_rot"] = scales.numpy(), rot_n.numpy()
    # ---- camera matrices: utils/graphics_utils.py:38-71 + scene/cameras.py:55-58 ---------------------------
    # (synth.make_camera must hand the op exactly what the reference's Camera class would)
    import math
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2  # noqa: E402  (reference code)
    cams = [(256, 256, 60.0, 0.0), (1920, 1080, 60.0, 0.0), (1920, 1080, 60.0, 10.0), (3840, 2160, 60.0, 35.0),
            (97, 61, 45.0, -20.0)]
    out["cam_params"] = np.array(cams, np.float64)
    for i, (W, H, fovx_deg, yaw_deg) in enumerate(cams):
        fovx = math.radians(fovx_deg)
        fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
        a = math.radians(yaw_deg)
        w2c_rot = np.array([[math.cos(a), 0.0, -math.sin(a)], [0.0, 1.0, 0.0], [math.sin(a), 0.0, math.cos(a)]])
        R = w2c_rot.T                      # the reference stores the camera-to-world rotation (cameras.py:21-22)
        T = np.zeros(3)
        view = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        out[f"cam{i}_view"], out[f"cam{i}_full"] = view.numpy(), full.numpy()
        out[f"cam{i}_center"] = view.inverse()[3, :3].numpy()
        out[f"cam{i}_tan"] = np.array([math.tan(fovx * 0.5), math.tan(fovy * 0.5)])   # gaussian_renderer/__init__.py:189-190
    np.savez_compressed(os.path.join(HERE, "reference_fallbacks.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_fallbacks.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
