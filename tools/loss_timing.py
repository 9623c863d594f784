"""Times gsplat_fused_loss (forward + backward kernels, non-blocking) with HIP events: python tools/loss_timing.py [HxW ...]"""
import os
import sys
import importlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dgs_amd.ops")


def main(shapes):
    for shape in shapes:
        H, W = (int(v) for v in shape.split("x"))
        g = torch.Generator(device="cuda").manual_seed(1)
        pred, gt = (torch.rand(H, W, 3, device="cuda", generator=g) for _ in range(2))
        grad = torch.empty_like(pred)
        for _ in range(20):
            ops.fused_loss(pred, gt, H, W, 0.2, grad, blocking=False)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            ops.fused_loss(pred, gt, H, W, 0.2, grad, blocking=False)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 200 * 1e3
        px = H * W
        print(f"{H}x{W}: {us:.1f} us per fused_loss (both kernels), {px * 132 / us / 1e3:.0f} GB/s of the 132 B/pixel "
              f"algorithmic traffic, checksum {float(grad.double().abs().sum()):.9e}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["840x1297", "1080x1920"])
