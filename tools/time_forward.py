"""Per-call host time and overall rate of forward-only rasterize_image calls."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
N, W, H, L, _ = scene.WORKLOADS["config3"]
params = scene.make_gaussians(N, W, H, L); cam = scene.make_camera(W, H, 0); c = scene.CONFIG
ctx = raster.RasterContext(N, W, H); dp, dc = raster.device_params(params), raster.device_camera(cam)
if os.environ.get("TIMING"): ctx.set_timing(True)
for _ in range(5): ctx.rasterize_image(dp, dc, c, c["bg"], L)
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for _ in range(40):
    a = time.perf_counter(); ctx.rasterize_image(dp, dc, c, c["bg"], L); ts.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
ts = np.array(ts) * 1e3
print(f"{os.environ.get('GSPLAT_FWD_WAIT','spin')}: {40/tot:.0f} fps; host ms per call: median {np.median(ts):.3f} min {ts.min():.3f} max {ts.max():.3f}")
