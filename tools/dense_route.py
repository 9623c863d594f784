import importlib, sys, os, torch
sys.path.insert(0, os.getcwd())
raster = importlib.import_module("3dgs_amd.raster"); scene = importlib.import_module("3dgs_amd.scene")
N, W, H, L, _ = scene.WORKLOADS["dense4m"]
dp = raster.device_params(scene.make_gaussians(N, W, H, L)); cam = raster.device_camera(scene.make_camera(W, H, 0))
ctx = raster.RasterContext(N, W, H)
for route in (2, 1):
    ctx.set_binning_route(route)
    for _ in range(3):
        fwd = ctx.rasterize_image(dp, cam, scene.CONFIG, scene.CONFIG["bg"], L)
    ctx.set_timing(True)
    for _ in range(10):
        fwd = ctx.rasterize_image(dp, cam, scene.CONFIG, scene.CONFIG["bg"], L)
    t = ctx.get_timing(); ctx.set_timing(False)
    r = fwd["ranges"].cpu().numpy() if "ranges" in fwd else None
    print("route", route, {k: round(v[0], 4) for k, v in t.items() if v[0] > 0}, "S", fwd["num_splats"])
    if r is not None:
        import numpy as np
        ln = np.diff(r); print("list lengths: mean", ln.mean(), "p50", np.percentile(ln, 50), "p90", np.percentile(ln, 90), "max", ln.max(), "frac>1024", (ln > 1024).mean(), "frac>2048", (ln > 2048).mean(), "frac>4096", (ln > 4096).mean())
