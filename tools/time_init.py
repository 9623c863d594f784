"""Times gsplat_initialize_gaussians on synthetic clouds against an exact kd-tree on the host cores (scipy cKDTree,
the role nanoflann + OpenMP play in the reference's Gaussians::Initialize)."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial import cKDTree
ops = importlib.import_module("3dgs_amd.ops")
rng = np.random.default_rng(0)
out = {}
for n in (100_000, 1_000_000):
    pts = np.concatenate([rng.normal(0, 1.0, (n * 6 // 10, 3)), rng.normal((5, 0, 0), 0.05, (n * 3 // 10, 3)),
                          rng.random((n - n * 6 // 10 - n * 3 // 10, 3)) * 40 - 20])
    col = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    d_pts, d_col = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
    ops.initialize_gaussians(d_pts, d_col)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g = ops.initialize_gaussians(d_pts, d_col)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    d, _ = cKDTree(pts, leafsize=10).query(pts, k=4, workers=-1)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    err = np.abs(np.exp(g["scale"][:, 0].cpu().numpy()) / d[:, 1:].mean(1) - 1).max()
    out[n] = dict(gpu_ms=round(gpu_ms, 2), kdtree_all_cores_ms=round(cpu_ms, 1), cores=os.cpu_count(), max_rel_err=float(err))
print(json.dumps(out))
