"""End-to-end training-iteration rate on the benchmark workload (1e6 gaussians, 1920x1080, SH 3, one view):
rasterize -> fused L1+SSIM loss -> backward -> masked in-place Adam, everything resident on the GPU."""
import gc, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
ops = importlib.import_module("3dgs_amd.ops"); opt_mod = importlib.import_module("3dgs_amd.optimizer")
N, W, H, L, _ = scene.WORKLOADS["config3"]
c = scene.CONFIG
dp = raster.device_params(scene.make_gaussians(N, W, H, L)); cam = raster.device_camera(scene.make_camera(W, H, 0))
ctx = raster.RasterContext(N, W, H)
target = ctx.rasterize_image(dp, cam, c, 0.0, L)["image"].clone()
dp["rgb"] += 0.05 * torch.randn_like(dp["rgb"])
opt = opt_mod.AdamOptimizer(dp, L, scene_extent=5.0)
grads = ctx.alloc_gradients(N, L, intermediates=("uv",), factored_sh=os.environ.get("PLAIN_SH") != "1")
grad_image = torch.empty(H, W, 3, device="cuda")


def step(it):
    fwd = ctx.rasterize_image(dp, cam, c, 0.0, L)
    ops.fused_loss(fwd["image"], target, H, W, 0.2, grad_image, blocking=False)
    ctx.backward_pass(dp, cam, grad_image, 0.0, L, grads)
    opt.step(it, fwd, grads, campos=cam["campos"])


for it in range(10):
    step(it)
torch.cuda.synchronize(); gc.collect(); gc.disable()
t0 = time.perf_counter()
K = 50
for it in range(K):
    step(10 + it)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
print(json.dumps({"train_iteration_ms": ms, "train_it_per_s": 1e3 / ms, "workload": "config3, one view, loss + Adam included"}))
