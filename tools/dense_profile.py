import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raster = importlib.import_module("3dgs_amd.raster"); scene = importlib.import_module("3dgs_amd.scene")
N, W, H, L, _ = scene.WORKLOADS["dense4m"]
dp = raster.device_params(scene.make_gaussians(N, W, H, L)); cam = raster.device_camera(scene.make_camera(W, H, 0))
ctx = raster.RasterContext(N, W, H)
ctx.set_binning_route(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(12):
    ctx.rasterize_image(dp, cam, scene.CONFIG, scene.CONFIG["bg"], L)
torch.cuda.synchronize()
