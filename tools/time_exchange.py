"""Time the kernels of the split gradient exchange on ONE GPU (no collective): pack, SH rebuild, common columns.
usage: python tools/time_exchange.py [world]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raster = importlib.import_module("3dgs_amd.raster")
scene = importlib.import_module("3dgs_amd.scene")

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, W, H, L, _ = scene.WORKLOADS["config3"]
dev = torch.device("cuda:0")
params = raster.device_params(scene.make_gaussians(N, W, H, L), dev)
cam = raster.device_camera(scene.make_camera(W, H, 0), dev)
ctx = raster.RasterContext(N, W, H)
grads = ctx.alloc_gradients(N, L)
grads["precompute_rgb"] = torch.empty(N, 3, device=dev)
ctx.rasterize_image(params, cam, scene.CONFIG, scene.CONFIG["bg"], L)
gi = torch.as_tensor(scene.make_grad_image(W, H)).to(dev)
ctx.backward_pass(params, cam, gi, scene.CONFIG["bg"], L, grads)
common = torch.zeros(N, 12, device=dev)
rgb = torch.zeros(N + 1, 3, device=dev)
rgb_all = torch.rand(world, N + 1, 3, device=dev)
packed = torch.empty(N, raster.packed_gradient_width(L), device=dev)


def timeit(name, fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / reps * 1e3:.1f} us")


timeit("pack_split (common + rgb)", lambda: raster.pack_gradients_split(ctx, grads, N, common, rgb))
timeit("pack_split (common only)", lambda: raster.pack_gradients_split(ctx, grads, N, common, None))
timeit(f"unpack SH half, world {world}", lambda: raster.unpack_gradients_split(params["xyz"], None, rgb_all, 3 * (N + 1), L, N, world, packed))
timeit("unpack common half", lambda: raster.unpack_gradients_split(None, common, None, 0, L, N, world, packed))
timeit("unpack both", lambda: raster.unpack_gradients_split(params["xyz"], common, rgb_all, 3 * (N + 1), L, N, world, packed))
