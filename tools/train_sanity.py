"""A longer training sanity run than the unit test: 100k target gaussians, 12 views at 640x360, 1500 iterations with
density control, SH growth and periodic Morton re-sorts.  Prints PSNR before / after and the iteration rate."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
ops = importlib.import_module("3dgs_amd.ops"); trainer_mod = importlib.import_module("3dgs_amd.trainer")
N, W, H, V = 100_000, 640, 360, 12
truth = scene.make_gaussians(N, W, H, 0)
truth["opacity"][:] = np.clip(truth["opacity"], 0.5, 3.0)
ctx = raster.RasterContext(N, W, H)
dp = raster.device_params(truth)
views = []
for v in range(V):
    cam = raster.device_camera(scene.make_camera(W, H, v))
    views.append((cam, ctx.rasterize_image(dp, cam, scene.CONFIG, 0.0, 0)["image"].clone()))
idx = np.random.default_rng(2).choice(N, N // 3, replace=False)
pts = torch.from_numpy(truth["xyz"][idx].astype(np.float64)).cuda()
col = torch.from_numpy(np.clip((truth["rgb"][idx] * 0.28209479 + 0.5) * 255, 0, 255).astype(np.uint8)).cuda()
init = ops.initialize_gaussians(pts, col)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = dict(num_iters=iters, add_sh_band_interval=500, max_sh_band=2, adaptive_control_start=100,
           adaptive_control_interval=100, adaptive_control_end=iters - 200, reset_opacity_start=10 ** 9,
           uv_grad_threshold=1e-6, max_gaussians=400_000, use_background=False)
t = trainer_mod.Trainer(init, views, cfg, scene_extent=5.0, seed=3)
p0 = t.evaluate()
torch.cuda.synchronize(); t0 = time.perf_counter()
hist = t.train(iters)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
p1 = t.evaluate()
losses = [h[1] for h in hist]
ok = all(np.isfinite(losses)) and all(torch.isfinite(v).all().item() for v in t.params.values())
print(f"PSNR {p0:.2f} -> {p1:.2f} dB, loss {np.mean(losses[:20]):.4f} -> {np.mean(losses[-20:]):.4f}, "
      f"{t.num_gaussians} gaussians, SH degree {t.l_max}, {iters / dt:.0f} it/s, finite: {ok}")
assert ok and p1 > p0 + 1.0
