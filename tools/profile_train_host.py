#!/usr/bin/env python3
"""Where does a training iteration's HOST time go?  cProfile over N iterations of the Trainer on a synthetic scene
(the GPU work per iteration is well under a millisecond at this size, so the loop is host-bound).
    python tools/profile_train_host.py [iterations]"""
import cProfile, importlib, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
ops = importlib.import_module("3dgs_amd.ops"); trainer_mod = importlib.import_module("3dgs_amd.trainer")
N, W, H, V = 150_000, 1297, 840, 8
truth = scene.make_gaussians(N, W, H, 0)
ctx = raster.RasterContext(N, W, H)
dp = raster.device_params(truth)
views = []
for v in range(V):
    cam = raster.device_camera(scene.make_camera(W, H, v))
    views.append((cam, ctx.rasterize_image(dp, cam, scene.CONFIG, 0.0, 0)["image"].clone()))
init = {k: v.clone() for k, v in dp.items() if k != "sh"}
init["sh"] = torch.zeros(N, 0, 3, device="cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = dict(num_iters=7000, adaptive_control_start=10 ** 9, reset_opacity_start=10 ** 9, add_sh_band_interval=100, use_background=True)
t = trainer_mod.Trainer(init, views, cfg, scene_extent=5.0, seed=1)
t.train(350, loss_every=0)   # reach SH degree 3
torch.cuda.synchronize()
t0 = time.perf_counter(); t.train(iters, loss_every=0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{iters / dt:.0f} it/s, {dt / iters * 1e3:.3f} ms per iteration (SH degree {t.l_max}, {t.num_gaussians} gaussians)")
pr = cProfile.Profile(); pr.enable(); t.train(iters, loss_every=0); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
