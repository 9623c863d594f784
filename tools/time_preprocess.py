#!/usr/bin/env python3
"""Per-stage times of the forward alone on a named workload (timing only: usable with ablated diagnostic libraries whose
outputs are garbage): python tools/time_preprocess.py [workload] [reps] [full]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
name = sys.argv[1] if len(sys.argv) > 1 else "config3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, W, H, L, _ = scene.WORKLOADS[name]
cfg = scene.CONFIG
dp = raster.device_params(scene.make_workload_gaussians(name)); dc = raster.device_camera(scene.make_camera(W, H, 0))
ctx = raster.RasterContext(N, W, H)
ctx.set_lean_forward(len(sys.argv) <= 3)
for _ in range(5):
    ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
ctx.set_timing(True)
for _ in range(reps):
    ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
st = ctx.get_timing()
print(os.environ.get("GSPLAT_LIB", "product"), "split", os.environ.get("GSPLAT_PRE_SPLIT"), {k: round(v[0], 4) for k, v in st.items() if v[0] > 0})
