"""In-tree build of the native pieces (gfx950 only).

  csrc/libf3dgs_hip.so                      the C-ABI product library (hipcc, hand-written HIP kernels)
  diff_gaussian_rasterization/_C*.so        pybind11/libtorch binding over the C ABI (g++)
  simple_knn/_C*.so                         same, for distCUDA2

Both are written next to their sources so that they travel with a `gpurun` snapshot.  hipcc
cross-compiles for gfx950 without a GPU.  Run:  python feature-3dgs_amd/build.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
PKG = os.path.join(HERE, "diff_gaussian_rasterization")
HIP_LIB = os.path.join(CSRC, "libf3dgs_hip.so")
EXT = os.path.join(PKG, "_C" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
KNN_EXT = os.path.join(HERE, "simple_knn", "_C" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_hip(force: bool = False) -> str:
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return HIP_LIB


def build_binding(force: bool = False, src_name: str = "binding.cpp", target: str = EXT) -> str:
    src = os.path.join(CSRC, src_name)
    hdr = os.path.join(HERE, "..", "include", "f3dgs.h")
    EXT = target
    if not force and _newer(EXT, [src, hdr, HIP_LIB]):
        return EXT
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    inc = ce.include_paths() + [pybind11.get_include(), sysconfig.get_paths()["include"], "/opt/rocm/include"]
    libdirs = ce.library_paths()
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H"]
           + [f"-I{p}" for p in inc] + [src, "-o", EXT]
           + [f"-L{p}" for p in libdirs] + [f"-L{CSRC}", "-lf3dgs_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
                                            "-ltorch_python"]
           + [f"-Wl,-rpath,{p}" for p in libdirs] + ["-Wl,-rpath,$ORIGIN/../csrc"])
    subprocess.check_call(cmd)
    return EXT


def build_ubench(force: bool = False) -> str:
    """tools/ubench/blend_stream: the blend kernels' inner streams (csrc/pl_phase1.h, csrc/fwd_group.h) run by themselves -
    bench.py's `roofline_compute` measures its floor with it on the GPU it runs on.  A measurement tool, not part of the product."""
    root = os.path.join(HERE, "..")
    src = os.path.join(root, "tools", "ubench", "blend_stream.hip")
    exe = os.path.join(root, "tools", "ubench", "blend_stream")
    deps = [src] + [os.path.join(CSRC, h) for h in ("pl_phase1.h", "fwd_group.h", "render_common.h", "common.h")]
    if not force and _newer(exe, deps):
        return exe
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mcode-object-version=5",
                           "-fno-slp-vectorize", f"-I{CSRC}", src, "-o", exe])
    return exe


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_binding(force)
    build_binding(force, "binding_knn.cpp", KNN_EXT)      # simple_knn._C
    build_ubench(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built", HIP_LIB, "and", EXT)
