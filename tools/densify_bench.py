"""Time one densify_and_prune at P points: the product (densify.py: one plan + one gather launch) and, when
oracle/_ref holds its bytecode, the reference's own method on the same model class (scene/gaussian_model.py:415-431).

    python tools/densify_bench.py [P] [C]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "feature-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import densify  # noqa: E402
from fused_adam import FusedAdam  # noqa: E402
import test_densify as td  # noqa: E402


def timed(fn, make, reps=5):
    out = []
    for _ in range(reps):
        m = make()
        torch.manual_seed(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(m)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return sorted(out)[len(out) // 2]


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    tensors, stats = td._tensors(P, C, seed=1)
    args = (0.0002, 0.005, 5.0, 20)
    pool = densify.RowPool()
    mine = timed(lambda m: densify.densify_and_prune(m, *args, pool=pool), lambda: td._model(None, tensors, stats, FusedAdam, adam_steps=1))
    # the gather launch alone (device time between two events around it)
    real, spans = densify._C, []

    class Proxy:
        @staticmethod
        def densify_gather(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); real.densify_gather(*a); e1.record()
            spans.append((e0, e1))
    densify._C = Proxy
    timed(lambda m: densify.densify_and_prune(m, *args, pool=pool), lambda: td._model(None, tensors, stats, FusedAdam, adam_steps=1), reps=3)
    densify._C = real
    torch.cuda.synchronize()
    gather = sorted(a.elapsed_time(b) for a, b in spans)[len(spans) // 2]
    line = f"P={P} C={C}: product {mine:.2f} ms (gather launch {gather:.3f} ms)"
    try:
        import pytest
        try:
            Ref = td.ru.load_reference_gaussian_model()
        except pytest.skip.Exception:
            Ref = None
        if Ref is not None:
            ref = timed(lambda m: m.densify_and_prune(*args), lambda: td._model(Ref, tensors, stats, torch.optim.Adam, adam_steps=1))
            line += f" | reference method {ref:.2f} ms ({ref / mine:.1f}x)"
    except Exception as e:  # noqa: BLE001
        line += f" | reference not timed: {e}"
    print(line)


if __name__ == "__main__":
    main()
