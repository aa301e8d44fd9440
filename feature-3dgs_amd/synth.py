"""Deterministic synthetic scenes for parity tests and bench.py.

Recipe = SURVEY.md section 8(d) "Synthetic inputs".  Camera matrices are built the way the
reference builds them (reference: scene/cameras.py:55-58, utils/graphics_utils.py:38-71):
`viewmatrix = W2C.T`, `projmatrix = (W2C.T) @ (P.T)`, i.e. element (r, c) of the mathematical
matrix sits at flat index 4*c + r.

Everything is generated on the CPU with a seeded torch.Generator so that the GPU box, this
container and the oracle see bit-identical inputs.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z_sign=+1 (utils/graphics_utils.py:51-71)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    Pm = torch.zeros(4, 4, dtype=torch.float32)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def make_camera(width: int, height: int, fovx_deg: float = 60.0, yaw_deg: float = 0.0,
                znear: float = 0.01, zfar: float = 100.0) -> Dict[str, object]:
    """Camera at the origin looking down +z, optionally rotated about y (view i of the DP bench)."""
    fovx = math.radians(fovx_deg)
    tanfovx = math.tan(fovx / 2)
    tanfovy = tanfovx * height / width
    fovy = 2 * math.atan(tanfovy)
    a = math.radians(yaw_deg)
    w2c = torch.eye(4, dtype=torch.float32)
    # rotation about y: camera looks along (sin a, 0, cos a)
    w2c[0, 0], w2c[0, 2] = math.cos(a), -math.sin(a)
    w2c[2, 0], w2c[2, 2] = math.sin(a), math.cos(a)
    view = w2c.t().contiguous()                       # world_view_transform
    proj = projection_matrix(znear, zfar, fovx, fovy).t().contiguous()
    full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()  # full_proj_transform
    campos = view.inverse()[3, :3].contiguous()
    return dict(image_width=width, image_height=height, tanfovx=tanfovx, tanfovy=tanfovy,
                viewmatrix=view, projmatrix=full, campos=campos)


def make_scene(P: int, C: int, width: int, height: int, seed: int = 0, sh_degree: int = 3,
               with_depth_grad: bool = False, scale_lo: float = 0.003, scale_hi: float = 0.03,
               yaw_deg: float = 0.0) -> Dict[str, object]:
    """Gaussians + camera + upstream gradients (all CPU fp32 tensors)."""
    g = torch.Generator().manual_seed(seed)
    cam = make_camera(width, height, yaw_deg=yaw_deg)
    tx, ty = cam["tanfovx"], cam["tanfovy"]

    def U(*shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    def N(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float32)

    z = U(P, lo=2.0, hi=10.0)
    u, v = U(P, lo=-1.0, hi=1.0), U(P, lo=-1.0, hi=1.0)
    m = torch.full((P,), 1.1)
    n_wide = int(0.02 * P)           # activates the 1.3x EWA clamp (Q7)
    n_near = int(0.01 * P)           # near-plane cull
    m[:n_wide] = 1.6
    z_near = U(P, lo=-1.0, hi=0.2)
    z[n_wide:n_wide + n_near] = z_near[n_wide:n_wide + n_near]
    means3D = torch.stack([u * m * tx * z, v * m * ty * z, z], dim=1).contiguous()
    scales = torch.exp(U(P, 3, lo=math.log(scale_lo), hi=math.log(scale_hi))).contiguous()
    rot = N(P, 4)
    rotations = (rot / rot.norm(dim=1, keepdim=True)).contiguous()
    opacities = torch.sigmoid(1.5 * N(P, 1)).contiguous()
    M = 16
    shs = (0.3 * N(P, M, 3))
    shs[:, 0, :] += U(P, 3, lo=-1.0, hi=1.0)
    shs = shs.contiguous()
    semantic = N(P, 1, C).contiguous()
    hw = float(width * height)
    dL_dcolor = (N(3, height, width) / hw).contiguous()
    dL_dfeature = (N(C, height, width) / hw).contiguous()
    dL_ddepth = (N(1, height, width) / hw).contiguous() if with_depth_grad else torch.zeros(1, height, width)
    scene = dict(P=P, C=C, M=M, sh_degree=sh_degree, means3D=means3D, scales=scales, rotations=rotations,
                 opacities=opacities, shs=shs, semantic_feature=semantic,
                 bg=torch.zeros(3, dtype=torch.float32), scale_modifier=1.0,
                 dL_dcolor=dL_dcolor, dL_dfeature=dL_dfeature, dL_ddepth=dL_ddepth)
    scene.update(cam)
    return scene


# The five BASELINE.json configurations (P, W, H, C, depth-grad?)
CONFIGS = {
    "c1": dict(P=10_000, width=256, height=256, C=0, with_depth_grad=False),
    "c2": dict(P=500_000, width=1920, height=1080, C=16, with_depth_grad=False),
    "c3": dict(P=1_000_000, width=1920, height=1080, C=32, with_depth_grad=False),
    "c4": dict(P=2_000_000, width=1920, height=1080, C=256, with_depth_grad=False),
    "c5": dict(P=5_000_000, width=3840, height=2160, C=128, with_depth_grad=True),
}
