"""Optimizer step of the reference's seven per-Gaussian tensors (59 + C floats per Gaussian, scene/gaussian_model.py:163-178)
on the GPU: FusedAdam as one launch (f3dgs_adam_step_multi), FusedAdam as one launch per tensor (round 2), torch.optim.Adam
with its foreach kernels (what the reference runs).  Development aid; output goes to profiles/rNN_adam.txt.

    python tools/adam_bench.py [P] [C]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import torch
from fused_adam import FusedAdam

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda:0"
specs = [("xyz", (P, 3), 1.6e-4), ("f_dc", (P, 1, 3), 2.5e-3), ("f_rest", (P, 15, 3), 1.25e-4), ("opacity", (P, 1), 0.05),
         ("scaling", (P, 3), 5e-3), ("rotation", (P, 4), 1e-3), ("semantic_feature", (P, 1, C), 1e-3)]


def groups():
    g = torch.Generator().manual_seed(3)
    out = []
    for n, s, lr in specs:
        p = torch.nn.Parameter(torch.randn(*s, generator=g).to(dev))
        p.grad = torch.randn(*s, generator=g).to(dev) * 1e-3
        out.append({"params": [p], "lr": lr, "name": n})
    return out


def bench(name, opt, iters=50):
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / iters
    dev_ms = e0.elapsed_time(e1) / iters
    nbytes = 28 * sum(4 ** 0 * torch.tensor(s).prod().item() for _, s, _ in specs)      # 16 B read + 12 B written per element
    print(f"{name:58s} {dev_ms:7.3f} ms on the stream ({nbytes / dev_ms / 1e6:6.0f} GB/s of the 28 B per element)   {wall:7.3f} ms wall")


print(f"Adam step, P = {P}, C = {C}: {sum(torch.tensor(s).prod().item() for _, s, _ in specs) // P} floats per Gaussian, 7 tensors")
bench("FusedAdam, one launch over the seven tensors", FusedAdam(groups(), lr=0.0, eps=1e-15, multi_tensor=True))
bench("FusedAdam, one launch per tensor (round 2)", FusedAdam(groups(), lr=0.0, eps=1e-15, multi_tensor=False))
bench("torch.optim.Adam(foreach=True) - the reference's optimizer", torch.optim.Adam(groups(), lr=0.0, eps=1e-15, foreach=True))
bench("torch.optim.Adam(fused=True)", torch.optim.Adam(groups(), lr=0.0, eps=1e-15, fused=True))
