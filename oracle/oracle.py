"""ctypes front-end of the CPU oracle (oracle/libf3dgs_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Never imported by the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libf3dgs_oracle.so")
_lib = None

_FP = ctypes.POINTER(ctypes.c_float)
_IP = ctypes.POINTER(ctypes.c_int)
_U8P = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "f3dgs_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.f3dgs_oracle_create.restype = ctypes.c_void_p
        lib.f3dgs_oracle_destroy.argtypes = [ctypes.c_void_p]
        lib.f3dgs_oracle_forward.restype = ctypes.c_int
        lib.f3dgs_oracle_forward.argtypes = (
            [ctypes.c_void_p] + [ctypes.c_int] * 4 + [_FP, ctypes.c_int, ctypes.c_int] + [_FP] * 6
            + [ctypes.c_float, _FP, _FP, _FP, _FP, _FP, ctypes.c_float, ctypes.c_float, _FP, _FP, _FP, _IP])
        lib.f3dgs_oracle_backward.restype = ctypes.c_int
        lib.f3dgs_oracle_backward.argtypes = (
            [ctypes.c_void_p] + [_FP] * 5 + [ctypes.c_float] + [_FP] * 5 + [ctypes.c_float, ctypes.c_float]
            + [_FP] * 14)
        lib.f3dgs_oracle_mark_visible.argtypes = [ctypes.c_int, _FP, _FP, _U8P]
        lib.f3dgs_oracle_read.restype = ctypes.c_long
        lib.f3dgs_oracle_read.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long]
        _lib = lib
    return _lib


def _f32(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if a.size else None


def _p(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_FP) if a is not None else ctypes.cast(None, _FP)


_DTYPES = {"clamped": np.uint8, "radii": np.int32, "tiles_touched": np.uint32, "point_list": np.uint32,
           "keys": np.uint64, "ranges": np.uint32, "n_contrib": np.uint32}


class Oracle:
    """One forward/backward pair of the CPU restatement.  Arguments follow the reference's
    `_C.rasterize_gaussians[_backward]` naming (rasterize_points.h:18-67)."""

    def __init__(self):
        self._lib = _load()
        self._h = ctypes.c_void_p(self._lib.f3dgs_oracle_create())
        self._keep = {}

    def __del__(self):
        try:
            self._lib.f3dgs_oracle_destroy(self._h)
        except Exception:
            pass

    def forward(self, *, bg, means3D, opacities, semantic_feature, viewmatrix, projmatrix, campos, tanfovx,
                tanfovy, image_height, image_width, sh_degree=0, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, scale_modifier=1.0) -> Dict[str, np.ndarray]:
        k = dict(bg=_f32(bg), means3D=_f32(means3D), opac=_f32(opacities), feat=_f32(semantic_feature),
                 view=_f32(viewmatrix), proj=_f32(projmatrix), campos=_f32(campos), shs=_f32(shs),
                 colors=_f32(colors_precomp), scales=_f32(scales), rots=_f32(rotations), cov=_f32(cov3D_precomp))
        P = 0 if k["means3D"] is None else k["means3D"].reshape(-1, 3).shape[0]
        M = 0 if k["shs"] is None else k["shs"].reshape(P, -1, 3).shape[1]
        if semantic_feature is None:
            C = 0
        else:
            shp = tuple(semantic_feature.shape)
            C = int(shp[-1])
        H, W = int(image_height), int(image_width)
        out_color = np.zeros((3, H, W), np.float32)
        out_feat = np.zeros((C, H, W), np.float32)
        out_depth = np.zeros((1, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        n = self._lib.f3dgs_oracle_forward(
            self._h, P, int(sh_degree), M, C, _p(k["bg"]), W, H, _p(k["means3D"]), _p(k["shs"]), _p(k["colors"]),
            _p(k["feat"]), _p(k["opac"]), _p(k["scales"]), float(scale_modifier), _p(k["rots"]), _p(k["cov"]),
            _p(k["view"]), _p(k["proj"]), _p(k["campos"]), float(tanfovx), float(tanfovy), _p(out_color),
            _p(out_feat) if C else ctypes.cast(None, _FP), _p(out_depth), radii.ctypes.data_as(_IP))
        self._keep = k
        self._dims = dict(P=P, M=M, C=C, H=H, W=W, tanfovx=float(tanfovx), tanfovy=float(tanfovy),
                          mod=float(scale_modifier))
        return dict(num_rendered=int(n), color=out_color, feature_map=out_feat, depth=out_depth, radii=radii)

    def backward(self, dL_dcolor, dL_dfeature, dL_ddepth) -> Dict[str, np.ndarray]:
        k, d = self._keep, self._dims
        P, M, C = d["P"], d["M"], d["C"]
        gc, gf, gd = _f32(dL_dcolor), _f32(dL_dfeature), _f32(dL_ddepth)
        if gd is None:
            gd = np.zeros((1, d["H"], d["W"]), np.float32)
        o = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
                 dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
                 dL_dsemantic_feature=np.zeros((P, 1, C), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
                 dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
                 dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
                 dL_dz=np.zeros((P, 1), np.float32))

        def q(name):
            a = o[name]
            return _p(a) if a.size else ctypes.cast(None, _FP)

        self._lib.f3dgs_oracle_backward(
            self._h, _p(k["bg"]), _p(k["means3D"]), _p(k["shs"]), _p(k["colors"]), _p(k["scales"]), d["mod"],
            _p(k["rots"]), _p(k["cov"]), _p(k["view"]), _p(k["proj"]), _p(k["campos"]), d["tanfovx"], d["tanfovy"],
            _p(gc), _p(gf), _p(gd), q("dL_dmeans2D"), q("dL_dconic"), q("dL_dopacity"), q("dL_dcolors"),
            q("dL_dsemantic_feature"), q("dL_dmeans3D"), q("dL_dcov3D"), q("dL_dsh"), q("dL_dscales"),
            q("dL_drotations"), q("dL_dz"))
        return o

    def read(self, what: str) -> np.ndarray:
        n = self._lib.f3dgs_oracle_read(self._h, what.encode(), None, 0)
        if n < 0:
            raise KeyError(what)
        a = np.zeros((n,), _DTYPES.get(what, np.float32))
        if n:
            self._lib.f3dgs_oracle_read(self._h, what.encode(), a.ctypes.data_as(ctypes.c_void_p), n)
        return a


def mark_visible(means3D, viewmatrix) -> np.ndarray:
    lib = _load()
    m, v = _f32(means3D), _f32(viewmatrix)
    P = 0 if m is None else m.reshape(-1, 3).shape[0]
    out = np.zeros((P,), np.uint8)
    if P:
        lib.f3dgs_oracle_mark_visible(P, _p(m), _p(v), out.ctypes.data_as(_U8P))
    return out.astype(bool)


def sh_to_rgb(deg: int, means3D, campos, shs) -> np.ndarray:
    """SH -> clamped RGB sub-step of the oracle (pinned against the reference's eval_sh fall-back)."""
    lib = _load()
    m, c, s = _f32(means3D), _f32(campos), _f32(shs)
    P = m.reshape(-1, 3).shape[0]
    M = s.reshape(P, -1, 3).shape[1]
    out = np.zeros((P, 3), np.float32)
    lib.f3dgs_oracle_sh_to_rgb.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, _FP, _FP, _FP, _U8P]
    lib.f3dgs_oracle_sh_to_rgb(P, int(deg), M, _p(m), _p(c), _p(s), _p(out), ctypes.cast(None, _U8P))
    return out


def cov3d(scales, scale_modifier: float, rotations) -> np.ndarray:
    """scale/rotation -> 6 unique covariance entries (pinned against the reference's Python fall-back)."""
    lib = _load()
    s, r = _f32(scales), _f32(rotations)
    P = s.reshape(-1, 3).shape[0]
    out = np.zeros((P, 6), np.float32)
    lib.f3dgs_oracle_cov3d.argtypes = [ctypes.c_int, _FP, ctypes.c_float, _FP, _FP]
    lib.f3dgs_oracle_cov3d(P, _p(s), float(scale_modifier), _p(r), _p(out))
    return out


def scene_kwargs(scene: dict, use_precomp_color=False, use_precomp_cov=False) -> dict:
    """Map a synth.make_scene() dict onto Oracle.forward keyword arguments."""
    kw = dict(bg=scene["bg"], means3D=scene["means3D"], opacities=scene["opacities"],
              semantic_feature=scene["semantic_feature"], viewmatrix=scene["viewmatrix"],
              projmatrix=scene["projmatrix"], campos=scene["campos"], tanfovx=scene["tanfovx"],
              tanfovy=scene["tanfovy"], image_height=scene["image_height"], image_width=scene["image_width"],
              sh_degree=scene["sh_degree"], scale_modifier=scene["scale_modifier"])
    if use_precomp_color:
        kw["colors_precomp"] = scene["colors_precomp"]
    else:
        kw["shs"] = scene["shs"]
    if use_precomp_cov:
        kw["cov3D_precomp"] = scene["cov3D_precomp"]
    else:
        kw["scales"], kw["rotations"] = scene["scales"], scene["rotations"]
    return kw
