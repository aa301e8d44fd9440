"""Development aid (needs a GPU): which scenes the forward pass flags as holding a long-axis Gaussian (counters[2])."""
import os, sys, ctypes
ROOT="/root/repo"
for p in (ROOT, ROOT+"/feature-3dgs_amd", ROOT+"/tests"): sys.path.insert(0,p)
import numpy as np, torch
from synth import make_scene, CONFIGS
sys.argv=[""]
import test_gpu_parity as tp
lib = tp._lib()
def ratio(sc):
    res = tp._raw_forward(sc)
    c = tp._read(lib, "counters", sc, res, np.uint32, 16)
    return int(c[2])        # 1: the frame holds a visible Gaussian longer than bwd_bf16_max_ratio (16) times its width
for shape in ("needle","disc"):
    for r in (1,4,8,12,15,16,32):
        print(shape, r, ratio(tp._needle_scene(r, shape)))
for cfg in ("c1","c2","c3"):
    print(cfg, ratio(make_scene(seed=0, **CONFIGS[cfg])))
from util import harsh_scene
print("heavy_tail", ratio(harsh_scene("heavy_tail", P=20000, C=32, width=320, height=200, seed=29)))
print("heavy_tail_round", ratio(harsh_scene("heavy_tail_round", P=20000, C=32, width=320, height=200, seed=29)))
