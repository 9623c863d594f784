#!/usr/bin/env python3
"""Stage times of the forward against the number of gaussians at a fixed image size: what is fixed cost, what scales.
python tools/time_scaling.py [reps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
W, H, L = 1920, 1080, 3
cfg = scene.CONFIG
dc = raster.device_camera(scene.make_camera(W, H, 0))
for N in (1024, 16384, 65536, 262144, 524288, 1000000):
    dp = raster.device_params(scene.make_gaussians(N, W, H, L))
    for lean in (True, False):
        ctx = raster.RasterContext(N, W, H)
        ctx.set_lean_forward(lean)
        for _ in range(5):
            f = ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        ctx.set_timing(True)
        for _ in range(reps):
            ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        st = ctx.get_timing()
        print(f"N {N:8d} lean {int(lean)} S {f['num_splats']:8d}", {k: round(v[0] * 1e3, 1) for k, v in st.items() if v[0] > 0}, "us")
        ctx.close()
