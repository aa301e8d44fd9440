"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/f3dgs.h
declares; the Python package has the reference's surface and refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest
import numpy as np
import torch

from util import ROOT


def _ensure_built():
    so = os.path.join(ROOT, "feature-3dgs_amd", "csrc", "libf3dgs_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return so


def test_c_abi_exports_every_declared_symbol():
    so = _ensure_built()
    header = open(os.path.join(ROOT, "include", "f3dgs.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(f3dgs_[a-z_0-9]+)\s*\(", header)) - {"f3dgs_resize_fn"})
    assert {"f3dgs_forward", "f3dgs_backward", "f3dgs_mark_visible", "f3dgs_last_error"} <= set(names)
    import torch  # noqa: F401  (its bundled HIP runtime must be the one the library binds to)
    lib = ctypes.CDLL(so)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/f3dgs.h but not exported"
    lib.f3dgs_version.restype = ctypes.c_int
    v = lib.f3dgs_version()           # include/f3dgs.h: major * 10000 + minor * 100 + patch
    assert v >= 30000 and v // 10000 == 3 and v % 100 < 100
    lib.f3dgs_backward_scratch_bytes.restype = ctypes.c_size_t
    assert lib.f3dgs_backward_scratch_bytes(1000, 32) >= 1000 * 40


def test_every_option_is_exercised_by_a_test():
    """include/f3dgs.h: "each option selects between complete code paths that are exercised by the test suite" - every name
    the library enumerates must be set by some test (option("name", ...) / set_option("name", ...)), apart from `profile`
    (no code path of its own: it only records events; bench.py drives it) which must at least be read back here."""
    import glob
    so = _ensure_built()
    lib = ctypes.CDLL(so)
    lib.f3dgs_option_name.restype = ctypes.c_char_p
    names, i = [], 0
    while (n := lib.f3dgs_option_name(i)) is not None:
        names.append(n.decode())
        i += 1
    assert "tile_cull" in names and len(names) == len(set(names))
    assert "dev" not in names, "development hooks in a release build"
    v = ctypes.c_int(-7)
    for n in names:
        assert lib.f3dgs_get_option(n.encode(), ctypes.byref(v)) == 0
    assert lib.f3dgs_get_option(b"no_such_option", ctypes.byref(v)) < 0
    text = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "test_*.py")))
    for n in names:
        if n == "profile":
            continue
        assert re.search(r'(option|set_option)\(\s*"%s"' % n, text) or re.search(r'parametrize\("name", \[[^\]]*"%s"' % n, text), \
            f"option {n} is not set by any test"


def test_no_default_kernel_spills():
    """Every kernel of the shipped library: no spilled vector registers, no scratch (tools/kernel_resources.py reads the
    code-object metadata; needs the object files of an in-tree build)."""
    import subprocess, sys
    _ensure_built()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_argument_validation_without_gpu():
    """Errors that must be raised before any device work (so they are testable on CPU)."""
    so = _ensure_built()
    lib = ctypes.CDLL(so)
    lib.f3dgs_last_error.restype = ctypes.c_char_p
    rc = lib.f3dgs_mark_visible(-1, None, None, None, None, None)
    assert rc < 0 and b"P < 0" in lib.f3dgs_last_error()
    assert lib.f3dgs_mark_visible(0, None, None, None, None, None) == 0
    assert lib.f3dgs_mark_visible(5, None, None, None, None, None) < 0
    # the low-resolution feature-map gradient: a thread-local setting, validated when it is made
    lib.f3dgs_set_feature_grad_lowres.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dummy = (ctypes.c_float * 4)()
    assert lib.f3dgs_set_feature_grad_lowres(ctypes.addressof(dummy), 0, 4, None) < 0 and b"low-resolution" in lib.f3dgs_last_error()
    assert lib.f3dgs_set_feature_grad_lowres(ctypes.addressof(dummy), 2, 2, None) == 0
    assert lib.f3dgs_set_feature_grad_lowres(None, 0, 0, None) == 0            # clears it
    # the fused loss: sizes and pointers are checked before anything is launched
    lib.f3dgs_feature_l1.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p] * 10
    assert lib.f3dgs_feature_l1(0, 8, 8, 4, 4, 4, *([None] * 10)) < 0
    assert lib.f3dgs_feature_l1(4, 8, 8, 4, 4, 4, *([None] * 10)) < 0 and b"null" in lib.f3dgs_last_error()
    lib.f3dgs_feature_l1_lowres_grad.restype = ctypes.c_void_p
    lib.f3dgs_feature_l1_lowres_grad.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
    assert lib.f3dgs_feature_l1_lowres_grad(4, 4, 4, 4, 0, None) is None


def test_python_surface_matches_reference():
    _ensure_built()
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    for n in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        assert hasattr(dgr._C, n)
    assert callable(dgr.rasterize_gaussians) and issubclass(dgr.GaussianRasterizer, torch.nn.Module)


def test_no_cpu_fallback():
    _ensure_built()
    import diff_gaussian_rasterization as dgr
    from synth import make_scene
    sc = make_scene(50, 4, 32, 32)
    st = dgr.GaussianRasterizationSettings(32, 32, sc["tanfovx"], sc["tanfovy"], sc["bg"], 1.0, sc["viewmatrix"],
                                           sc["projmatrix"], 3, sc["campos"], False, False)
    r = dgr.GaussianRasterizer(st)
    kw = dict(means3D=sc["means3D"], means2D=torch.zeros(50, 3), opacities=sc["opacities"],
              semantic_feature=sc["semantic_feature"], scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(RuntimeError, match="HIP device"):
        r(shs=sc["shs"], **kw)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(**kw)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=sc["means3D"], means2D=torch.zeros(50, 3), opacities=sc["opacities"], shs=sc["shs"],
          semantic_feature=sc["semantic_feature"], scales=sc["scales"])


def test_band_relabelling_is_a_bijection_that_spreads_the_band():
    """f3dgs_set_tile_band lists a contiguous run of tile ids - one XCD's share under the whole-view workgroup -> tile mapping.
    The blend kernels therefore relabel the tiles (common.h: band_perm); host-side restatement through the C ABI, no GPU: every
    tile exactly once, and every XCD's run of virtual ids (a contiguous eighth) starts with its eighth of the band."""
    import ctypes
    lib = ctypes.CDLL(_ensure_built())
    lib.f3dgs_debug_band_order.restype = ctypes.c_int
    lib.f3dgs_debug_band_order.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    for gx, gy, r0, r1 in [(120, 68, 0, 9), (120, 68, 27, 36), (120, 68, 60, 68), (240, 135, 17, 34), (240, 135, 118, 135),
                           (37, 29, 3, 5), (120, 68, 0, 68), (120, 68, 10, 60), (8, 6, 2, 3), (63, 41, 40, 41), (120, 68, 30, 30)]:
        T = gx * gy
        out = np.zeros(T, np.uint32)
        on = lib.f3dgs_debug_band_order(gx, gy, r0, r1, out.ctypes.data_as(ctypes.c_void_p))
        assert on in (0, 1)
        assert np.array_equal(np.sort(out), np.arange(T, dtype=np.uint32)), (gx, gy, r0, r1)
        nb = gx * (r1 - r0)
        if not on:
            assert np.array_equal(out, np.arange(T, dtype=np.uint32))
            assert nb == 0 or nb == T or T // 8 < nb // 8 + 8
            continue
        in_band = (out >= gx * r0) & (out < gx * r1)
        q, r = divmod(T, 8)
        first = 0
        for x in range(8):
            n = q + (1 if x < r else 0)
            run = in_band[first:first + n]
            want = nb // 8 + (1 if x < nb % 8 else 0)
            assert int(run.sum()) == want and run[:want].all(), (gx, gy, r0, r1, x)
            first += n
