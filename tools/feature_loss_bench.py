"""Development aid (GPU): time of the fused feature-map loss against the reference's three PyTorch calls + autograd,
at the training shapes (rendered map 1080p, ground truth 360x480; LSeg 128 -> 512, SAM 64 -> 256, c3 32 -> 128)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))
import torch
import torch.nn.functional as F
from feature_loss import fused_feature_l1

dev = "cuda:0"
for C, Cout in ((32, 128), (64, 256), (128, 512)):
    H, W, Hg, Wg = 1080, 1920, 360, 480
    g = torch.Generator().manual_seed(0)
    fm = torch.randn(C, H, W, generator=g).to(dev).requires_grad_(True)
    gt = torch.randn(Cout, Hg, Wg, generator=g).to(dev)
    conv = torch.nn.Conv2d(C, Cout, kernel_size=1).to(dev)

    def ref():
        x = F.interpolate(fm.unsqueeze(0), size=(Hg, Wg), mode="bilinear", align_corners=True).squeeze(0)
        x = conv(x)
        loss = torch.abs(x - gt).mean()
        loss.backward()
        return loss

    def fused():
        loss = fused_feature_l1(fm, gt, conv.weight, conv.bias)
        loss.backward()
        return loss

    from diff_gaussian_rasterization import _C
    w2 = conv.weight.detach().reshape(Cout, C).contiguous()
    b2 = conv.bias.detach().contiguous()

    def fused_lowres():      # the same call with the gradient left at the loss's resolution (fused_feature_l1(lowres_grad=True))
        return _C.feature_l1(fm.detach(), gt, w2, b2, False)[0]

    for name, fn in (("torch ops", ref), ("fused", fused), ("fused, gradient at the loss resolution", fused_lowres)):
        for _ in range(3):
            fn(); fm.grad = None; conv.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            fn(); fm.grad = None; conv.zero_grad()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        flops = 3 * 2.0 * Hg * Wg * C * Cout
        print(f"C={C:3d} -> {Cout:3d}  {name:40s} {ms:7.3f} ms  ({flops / ms / 1e9:6.1f} TFLOP/s on the three contractions)", flush=True)
