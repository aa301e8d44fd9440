import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import torch
import bench
from synth import make_scene, CONFIGS
sc = make_scene(seed=0, **CONFIGS["c3"])
dev = torch.device("cuda:0")
step, leaves = bench.make_step(sc, dev)
for i in range(20): step(i)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
ts = []
for blk in range(5):
    t0 = time.perf_counter()
    for i in range(400): step(i)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0) / 400)
print("ms/step per block of 400:", [round(t, 4) for t in ts])
print("allocated MB before/after:", m0 >> 20, torch.cuda.memory_allocated() >> 20, "reserved MB:", r0 >> 20, torch.cuda.memory_reserved() >> 20)
g = {k: float(v.grad.abs().sum()) for k, v in leaves.items() if v.grad is not None}
print("finite grads:", all(x == x and x != float("inf") for x in g.values()))
