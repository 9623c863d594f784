"""Times gsplat_compact_masked_array / gsplat_scatter_masked_array alone (HIP events): the strides the reference host uses.
usage: python tools/time_compact.py [N]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dgs_amd.ops")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for frac in (1.0, 0.5):
    mask = (torch.rand(N, device="cuda") < frac).to(torch.uint8)
    M = int(mask.sum().item())
    for stride in (1, 2, 3, 4, 45, 7):
        src = torch.randn(N * stride, device="cuda")
        out = ops.compact_masked_array(stride, src, mask, M)
        dst = torch.zeros_like(src)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            out = ops.compact_masked_array(stride, src, mask, M)
        e1.record()
        for _ in range(20):
            ops.scatter_masked_array(stride, out, mask, dst)
        e2.record()
        torch.cuda.synchronize()
        gb = (N + M) * stride * 4 / 1e9
        c, sc = e0.elapsed_time(e1) / 20, e1.elapsed_time(e2) / 20
        print(f"kept {frac:.1f} stride {stride:2d}: compact {c * 1e3:7.1f} us ({gb / c * 1e3 / 1e3:5.2f} TB/s), scatter {sc * 1e3:7.1f} us")
