import importlib, sys, os, torch, time
sys.path.insert(0, os.getcwd())
raster = importlib.import_module("3dgs_amd.raster"); scene = importlib.import_module("3dgs_amd.scene")
N, W, H, L, _ = scene.WORKLOADS["config3"]
dp = raster.device_params(scene.make_gaussians(N, W, H, L)); cam = raster.device_camera(scene.make_camera(W, H, 0))
ctx = raster.RasterContext(N, W, H)
for mode in (False, True, False, True):
    ctx.set_render_only(mode)
    for _ in range(10): ctx.rasterize_image(dp, cam, scene.CONFIG, scene.CONFIG["bg"], L)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.rasterize_image(dp, cam, scene.CONFIG, scene.CONFIG["bg"], L)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("render_only", mode, round(200 / dt, 1), "fps")
