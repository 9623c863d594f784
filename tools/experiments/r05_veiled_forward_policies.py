#!/usr/bin/env python3
"""r05: the forward's long-list segments on the veiled1200k workload under different hand-over policies
(RasterContext.set_segment_options): forward stage time per policy, same process, same scene."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
name = sys.argv[1] if len(sys.argv) > 1 else "veiled1200k"
N, W, H, L, _ = scene.WORKLOADS[name]
cfg = scene.CONFIG
params = scene.make_workload_gaussians(name)
dp = raster.device_params(params); dc = raster.device_camera(scene.make_camera(W, H, 0))
for what, opts in (("one workgroup per tile", dict(gate=1e9)), ("default (thin layers < 512 blocks side by side)", dict()),
                   ("every block waits for the one in front", dict(thin_layer_blocks=0)),
                   ("thin < 128", dict(thin_layer_blocks=128)), ("thin < 2048", dict(thin_layer_blocks=2048)),
                   ("all layers side by side", dict(thin_layer_blocks=1 << 20))):
    ctx = raster.RasterContext(N, W, H)
    ctx.set_lean_forward(True)
    ctx.set_render_only(True)
    ctx.set_segment_options(**opts)
    for _ in range(6):
        ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        torch.cuda.synchronize()
    ctx.set_timing(True, stages=["render_forward"])
    for _ in range(40):
        ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    st = ctx.get_timing(); ctx.set_timing(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    torch.cuda.synchronize()
    c = ctx.counters()
    print(f"{what:50s} render_forward {st['render_forward'][0]:.4f} ms; forward {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms; "
          f"segmented {c['segmented_forwards']} of {c['forwards']}")
    ctx.close()
