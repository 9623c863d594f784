"""Times fused_loss (non-blocking), compute_psnr and adam_step at the headline sizes with HIP events."""
import importlib, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dgs_amd.ops")
H, W, N, S = 1080, 1920, 1_000_000, 59
pred, gt = torch.rand(H, W, 3, device="cuda"), torch.rand(H, W, 3, device="cuda")
grad = torch.empty_like(pred)
p, g, m, v = (torch.rand(N, S, device="cuda") for _ in range(4))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = dict(fused_loss_ms=timeit(lambda: ops.fused_loss(pred, gt, H, W, 0.2, grad, blocking=False)),
           fused_loss_blocking_ms=timeit(lambda: ops.fused_loss(pred, gt, H, W, 0.2, grad)),
           psnr_ms=timeit(lambda: ops.compute_psnr(pred, gt, H, W)),
           adam_ms=timeit(lambda: ops.adam_step(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.001, N, S)))
out["adam_GBps"] = N * S * 4 * 7 / out["adam_ms"] / 1e6
print(json.dumps(out))
