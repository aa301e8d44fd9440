"""Soak of a step replayed from a HIP graph (graph_step.CapturedStep): thousands of replays, the gradients of the last replay
against an eager step, flat memory, the provision never exceeded.  python tools/soak_graph.py [config] [replays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "feature-3dgs_amd")]
import torch
import bench
from synth import make_scene, CONFIGS
from graph_step import CapturedStep
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
sc = make_scene(seed=0, **CONFIGS[cfg])
dev = torch.device("cuda:0")
step, leaves = bench.make_step(sc, dev)
for i in range(10):
    step(0)
torch.cuda.synchronize()
want = {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None and v.grad.numel()}
cs = CapturedStep(lambda: step(0)).capture()
m0 = torch.cuda.memory_allocated()
ts = []
for blk in range(4):
    t0 = time.perf_counter()
    for _ in range(n // 4):
        cs.replay()
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0) / (n // 4))
    assert cs.check(), "a replayed frame did not fit its provision"
print(cfg, "ms per replayed step per block:", [round(t, 4) for t in ts], "captures", cs.captures, "counts", cs.counts())
print("allocated MB before / after:", m0 >> 20, torch.cuda.memory_allocated() >> 20)
worst = 0.0
for k, w in want.items():
    g = leaves[k].grad
    worst = max(worst, float((g - w).abs().max() / (w.abs().max() + 1e-30)))
print("last replay against the eager step, worst |diff| / max|g| over the leaves:", worst)
assert worst < 1e-4
