"""Helpers for the reference-pinned GPU parity tests (tests/test_gpu_vs_ref.py).

`oracle/_ref/_refC<n>*.so` is the REFERENCE's own rasterizer extension (its `_C` module: `ext.cpp:15-19`),
compiled for gfx950 from the sources under /root/reference by `oracle/build_ref.py`.  Here both modules - the
product's `diff_gaussian_rasterization._C` and the reference's - are driven through ONE function with the
reference's positional signatures (`rasterize_points.h:18-72`), so the same call lands in both.

The reference's opaque state buffers are parsed the way `rasterizer_impl.cu:154-199` carves them (`obtain`,
128-byte alignment of the absolute address), which exposes its radii, depths, projected means, conics,
colours, sorted instance list, tile ranges, final transmittance and n_contrib for element-wise comparison.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import sysconfig

import numpy as np
import pytest
import torch

from util import ROOT

REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def ref_path(C: int, strict: bool = False) -> str:
    return os.path.join(REF_DIR, f"_refC{C}" + ("s" if strict else "") + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


_loaded = {}


def load_ref(C: int, strict: bool = False):
    """The reference's `_C` compiled with NUM_SEMANTIC_CHANNELS = C (config.h:16); skips when not built.
    strict: the flavour built with `-ffp-contract=off` (oracle/build_ref.py: STRICT_CHANNELS) - the product's
    preprocess is built that way too, so integer artefacts must agree EXACTLY with it."""
    key = (C, strict)
    if key in _loaded:
        return _loaded[key]
    p = ref_path(C, strict)
    if not os.path.exists(p):
        pytest.skip(f"{p} not built (run `python oracle/build_ref.py` where /root/reference exists)")
    name = f"_refC{C}" + ("s" if strict else "")
    spec = importlib.util.spec_from_file_location(name, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _loaded[key] = mod
    return mod


def product_module():
    from diff_gaussian_rasterization import _C
    return _C


def device_inputs(scene: dict, C_module: int, dev="cuda:0", precomp_color=False, precomp_cov=False) -> dict:
    """Device tensors for one call.  When the module's baked-in feature width differs from the scene's
    (RGB-only scene against a reference built with C = 3), a zero feature tensor of the module's width is used."""
    t = lambda x: x.to(dev).contiguous()
    e = torch.Tensor([])
    P = scene["P"]
    if scene["C"] == C_module:
        feat, dfeat = t(scene["semantic_feature"]), t(scene["dL_dfeature"])
    else:
        feat = torch.zeros(P, 1, C_module, device=dev)
        dfeat = torch.zeros(C_module, scene["image_height"], scene["image_width"], device=dev)
    return dict(bg=t(scene["bg"]), means3D=t(scene["means3D"]), opacities=t(scene["opacities"]),
                semantic_feature=feat,
                colors_precomp=t(scene["colors_precomp"]) if precomp_color else e,
                shs=e if precomp_color else t(scene["shs"]),
                scales=e if precomp_cov else t(scene["scales"]),
                rotations=e if precomp_cov else t(scene["rotations"]),
                cov3D_precomp=t(scene["cov3D_precomp"]) if precomp_cov else e,
                viewmatrix=t(scene["viewmatrix"]), projmatrix=t(scene["projmatrix"]), campos=t(scene["campos"]),
                dL_dcolor=t(scene["dL_dcolor"]), dL_dfeature=dfeat, dL_ddepth=t(scene["dL_ddepth"]))


def raw_forward(mod, scene: dict, d: dict, debug=False):
    """`_C.rasterize_gaussians` with the reference's positional arguments (rasterize_points.h:18-40)."""
    res = mod.rasterize_gaussians(
        d["bg"], d["means3D"], d["colors_precomp"], d["semantic_feature"], d["opacities"], d["scales"],
        d["rotations"], scene["scale_modifier"], d["cov3D_precomp"], d["viewmatrix"], d["projmatrix"],
        scene["tanfovx"], scene["tanfovy"], scene["image_height"], scene["image_width"], d["shs"],
        scene["sh_degree"], d["campos"], False, debug)
    torch.cuda.synchronize()
    return res


def raw_backward(mod, scene: dict, d: dict, fwd, dL_dcolor=None, dL_dfeature=None, dL_ddepth=None, debug=False):
    """`_C.rasterize_gaussians_backward` (rasterize_points.h:42-67); returns the nine gradients by name."""
    num_rendered, _color, _feat, _depth, radii, geom, binning, img = fwd
    g = mod.rasterize_gaussians_backward(
        d["bg"], d["means3D"], radii, d["colors_precomp"], d["semantic_feature"], d["scales"], d["rotations"],
        scene["scale_modifier"], d["cov3D_precomp"], d["viewmatrix"], d["projmatrix"], scene["tanfovx"],
        scene["tanfovy"], d["dL_dcolor"] if dL_dcolor is None else dL_dcolor,
        d["dL_dfeature"] if dL_dfeature is None else dL_dfeature,
        d["dL_ddepth"] if dL_ddepth is None else dL_ddepth, d["shs"], scene["sh_degree"], d["campos"], geom,
        num_rendered, binning, img, debug)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dsemantic_feature", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh",
             "dL_dscales", "dL_drotations"]
    return dict(zip(names, g))


# ---------------------------------------------------------------- the reference's private state -----
def _carve(buf: torch.Tensor, fields, want=None):
    """Replays `obtain()` (rasterizer_impl.h:25-30) over a reference state buffer; fields = [(name, dtype,
    count)] in carve order; a dtype of None is an opaque byte run.  Returns {name: numpy array}; only the fields
    named in `want` (default: all) are copied to the host (the full-size configs hold GBs of state)."""
    base = buf.data_ptr()
    p = base
    out = {}
    for name, dtype, count in fields:
        p = (p + 127) & ~127
        nbytes = count * (np.dtype(dtype).itemsize if dtype is not None else 1)
        if dtype is not None and name is not None and (want is None or name in want):
            out[name] = buf[p - base:p - base + nbytes].cpu().numpy().view(dtype).copy()
        p += nbytes
    return out


def ref_image_state(fwd, W: int, H: int) -> dict:
    """accum_alpha (= final T), n_contrib per pixel, ranges per tile (rasterizer_impl.cu:174-181; all three are
    carved with N = W*H elements there)."""
    N = W * H
    st = _carve(fwd[7], [("final_T", np.float32, N), ("n_contrib", np.uint32, N), ("ranges", np.uint32, 2 * N)])
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    st["ranges"] = st["ranges"][:2 * tiles]
    return st


def ref_geometry_state(fwd, P: int, C: int, want=None) -> dict:
    """depths, clamped, means2D, cov3D, conic_opacity, rgb, (features), tiles_touched
    (rasterizer_impl.cu:154-172).  `tiles_touched` is NOT overwritten by the scan (it writes point_offsets);
    `internal_radii` is unused whenever the caller passes a radii tensor (rasterizer_impl.cu:236-239)."""
    return _carve(fwd[5], [("depths", np.float32, P), ("clamped", np.uint8, 3 * P), (None, np.int32, P),
                           ("means2D", np.float32, 2 * P), ("cov3D", np.float32, 6 * P),
                           ("conic_opacity", np.float32, 4 * P), ("rgb", np.float32, 3 * P),
                           (None, np.float32, P * C), ("tiles_touched", np.uint32, P)], want)


def ref_point_list(fwd) -> np.ndarray:
    """Sorted instance list (rasterizer_impl.cu:183-199: point_list is the first sub-buffer, num_rendered long)."""
    n = int(fwd[0])
    return _carve(fwd[6], [("point_list", np.uint32, n)])["point_list"] if n else np.zeros(0, np.uint32)


# ---------------------------------------------------------------- the product's private state -------
def _lib():
    import diff_gaussian_rasterization  # noqa: F401
    lib = ctypes.CDLL(os.path.join(ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so"))
    lib.f3dgs_debug_read.restype = ctypes.c_int
    lib.f3dgs_debug_read.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.f3dgs_last_error.restype = ctypes.c_char_p
    return lib


def product_read(what: str, scene: dict, fwd, dtype, count: int) -> np.ndarray:
    lib = _lib()
    n, _, _, _, _, geom, binning, img = fwd
    out = np.zeros(count, dtype)
    rc = lib.f3dgs_debug_read(what.encode(), scene["P"], scene["C"], int(n), scene["image_width"],
                              scene["image_height"], geom.data_ptr(), binning.data_ptr() if binning.numel() else None,
                              img.data_ptr(), out.ctypes.data_as(ctypes.c_void_p), out.nbytes, None)
    assert rc == 0, lib.f3dgs_last_error()
    return out


def product_image_state(scene: dict, fwd) -> dict:
    W, H = scene["image_width"], scene["image_height"]
    return dict(final_T=product_read("final_T", scene, fwd, np.float32, W * H),
                n_contrib=product_read("n_contrib", scene, fwd, np.uint32, W * H))


# ---------------------------------------------------------------- flips ----------------------------
def flip_pixels(ref_img: dict, got_img: dict) -> np.ndarray:
    """Pixels where the two implementations PROVABLY took a different discrete decision.

    The blend has two thresholds (forward.cu:349-358): a splat is skipped when alpha < 1/255 and the walk ends
    when T would drop below 1e-4.  A different `exp` rounding can flip either at a borderline pixel:
      * a termination flip moves the last contributor            -> n_contrib differs;
      * an alpha flip multiplies T by (1 - alpha), alpha >= 1/255  -> final T differs by >= 0.39 % relative
        (all later factors are identical), far above fp32 noise (~1e-6 relative).
    Everything else is a continuous function of the inputs.  Returns a boolean mask (H*W,)."""
    nc = ref_img["n_contrib"] != got_img["n_contrib"]
    tr, tg = ref_img["final_T"].astype(np.float64), got_img["final_T"].astype(np.float64)
    dt = np.abs(tr - tg) > 1e-3 * np.maximum(np.abs(tr), np.abs(tg))
    return nc | dt


def grad_errors(got, want):
    """(normalised max error, worst element-wise excess) for `|got - want| <= 1e-3*|want| + 1e-5*max|want|`;
    the second number is max(|err| / (1e-3*|want| + 1e-5*scale)): <= 1 means every element is inside.
    Device tensors are compared where they live (fp64 on the GPU: the full-size configs hold 10^9 elements)."""
    if isinstance(want, torch.Tensor):
        want = want.detach().reshape(-1).double()
        got = got.detach().reshape(-1).double()
        if want.numel() == 0:
            return 0.0, 0.0
        scale = float(want.abs().max()) + 1e-30
        mx, worst = 0.0, 0.0
        step = 1 << 26                       # bounded temporaries
        for i in range(0, want.numel(), step):
            w, g = want[i:i + step], got[i:i + step]
            err = (g - w).abs()
            mx = max(mx, float(err.max()))
            worst = max(worst, float((err / (1e-3 * w.abs() + 1e-5 * scale)).max()))
        return mx / scale, worst
    want = np.asarray(want, np.float64).reshape(-1)
    got = np.asarray(got, np.float64).reshape(-1)
    if want.size == 0:
        return 0.0, 0.0
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want)
    return float(err.max() / scale), float((err / (1e-3 * np.abs(want) + 1e-5 * scale)).max())


# ---- the reference's GaussianModel class (bytecode), for the densification checker ---------------------------------
def load_reference_gaussian_model():
    """`scene.gaussian_model.GaussianModel` of the reference, executed from `oracle/_ref/ref_gaussian_model.pyc`.
    Its imports resolve to the reference's own utils (bytecode as well), the product's `simple_knn` and a stub
    `plyfile` (not installed here; only load_ply/save_ply would touch it)."""
    import importlib.machinery
    import sys
    import types

    names = ("ref_gaussian_model.pyc", "ref_general_utils.pyc", "ref_graphics_utils.pyc", "ref_system_utils.pyc", "ref_sh_utils.pyc")
    for n in names:
        if not os.path.exists(os.path.join(REF_DIR, n)):
            pytest.skip(f"oracle/_ref/{n} not built (python oracle/build_ref.py where /root/reference exists)")

    def load(name, file):
        loader = importlib.machinery.SourcelessFileLoader(name, os.path.join(REF_DIR, file))
        spec = importlib.util.spec_from_loader(name, loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        return mod

    keys = ("utils", "utils.general_utils", "utils.graphics_utils", "utils.system_utils", "utils.sh_utils", "plyfile")
    saved = {k: sys.modules.get(k) for k in keys}
    try:
        utils_pkg = types.ModuleType("utils")
        sys.modules["utils"] = utils_pkg
        for sub, file in (("general_utils", "ref_general_utils.pyc"), ("graphics_utils", "ref_graphics_utils.pyc"),
                          ("system_utils", "ref_system_utils.pyc"), ("sh_utils", "ref_sh_utils.pyc")):
            m = load("utils." + sub, file)
            sys.modules["utils." + sub] = m
            setattr(utils_pkg, sub, m)
        if "plyfile" not in sys.modules or sys.modules["plyfile"] is None:
            ply = types.ModuleType("plyfile")
            ply.PlyData = ply.PlyElement = type("Unavailable", (), {})
            sys.modules["plyfile"] = ply
        import simple_knn._C  # noqa: F401  (the product's distCUDA2: the reference's import line finds it)
        return load("ref_scene_gaussian_model", "ref_gaussian_model.pyc").GaussianModel
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
