"""Per-kernel resource usage of the shipped library, read from the code-object metadata (no GPU needed).

    python tools/kernel_resources.py            # table of every kernel in csrc/*.o
    python tools/kernel_resources.py --check    # exit 1 if any kernel spills or uses scratch

For each object file the gfx950 code object is taken out of the .hip_fatbin section (llvm-objcopy, clang-offload-bundler) and its AMDGPU metadata note is parsed:
.vgpr_count, .agpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane),
.group_segment_fixed_size (static LDS).  tests/test_abi_and_surface.py runs --check.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "feature-3dgs_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj: str):
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "0"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append(dict(name=dem, vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), vspill=int(g("vgpr_spill_count")),
                        sspill=int(g("sgpr_spill_count")), scratch=int(g("private_segment_fixed_size")),
                        lds=int(g("group_segment_fixed_size"))))
    return out


def main():
    check = "--check" in sys.argv
    bad = []
    rows = []
    for obj in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
        for k in kernels_of(obj):
            rows.append((os.path.basename(obj), k))
            if k["vspill"] or k["scratch"]:
                bad.append(k)
    if not check:
        print(f"{'object':18s} {'vgpr':>4s} {'agpr':>4s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}  kernel")
        for o, k in rows:
            short = re.sub(r"^(void )?f3dgs::(\(anonymous namespace\)::)?", "", k["name"])
            short = re.sub(r"\((?!anonymous).*$", "", short)
            print(f"{o:18s} {k['vgpr']:4d} {k['agpr']:4d} {k['vspill']:6d} {k['sspill']:6d} {k['scratch']:7d} {k['lds']:6d}  {short}")
        print(f"{len(rows)} kernels, {len(bad)} with vector spills or scratch; libf3dgs_hip.so: "
              f"{os.path.getsize(os.path.join(CSRC, 'libf3dgs_hip.so'))} bytes")
    if check and bad:
        for k in bad:
            print(f"SPILL: {k['name']}: {k['vspill']} spilled VGPRs, {k['scratch']} B scratch", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
