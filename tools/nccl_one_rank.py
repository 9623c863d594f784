"""One-rank rehearsal of the split gradient exchange on the REAL backend (nccl = RCCL): the collectives degenerate to
copies, but tensor shapes, contiguity and the async handles go through the same torch.distributed / RCCL calls as the
multi-GPU run (RCCL refuses two ranks on one device, so this is as close as a single GPU gets).
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_one_rank.py"""
import importlib, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raster = importlib.import_module("3dgs_amd.raster"); scene = importlib.import_module("3dgs_amd.scene")
gdist = importlib.import_module("3dgs_amd.dist")

os.environ["GSPLAT_DIST_BACKEND"] = "nccl"
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
N, W, H, L, _ = scene.WORKLOADS["small"]
dev = torch.device("cuda:0")
dp = raster.device_params(scene.make_gaussians(N, W, H, L), dev); cam = raster.device_camera(scene.make_camera(W, H, 0), dev)
gi = torch.as_tensor(scene.make_grad_image(W, H)).to(dev)
cfg = scene.CONFIG
step = gdist.ViewShardedStep(dp, L, W, H, cfg, cfg["bg"], exchange="split")
assert step.world == 1
# the overlapped multi-rank path, driven by hand
step._set_campos(cam)
fwd = step.ctx.rasterize_image(dp, cam, cfg, cfg["bg"], L)
step.ctx.backward_render(gi, cfg["bg"], step.rgb)
step._rgb_gather = gdist.all_gather_blocks(step.rgb_all, step.rgb, async_op=True)
step.ctx.backward_gaussians(dp, cam, L, step.grads)
packed = step.exchange_gradients(cam).clone()
torch.cuda.synchronize()
# reference: the full rows of the same backward
full = torch.empty_like(packed)
step.ctx.pack_gradients_global(step.grads, L, N, full)
dist.all_reduce(full)
torch.cuda.synchronize()
err = (packed - full).abs().max().item(); scale = full.abs().max().item()
print(f"split exchange over nccl, one rank: max abs diff {err:.3e} (scale {scale:.3e})")
assert err <= 1e-6 * max(scale, 1e-30) + 1e-12, "split exchange differs from the full rows"
# the chunked backward: one all-reduce per range of global indices, started behind that range (async handles on RCCL)
step.chunks = 4
step.ctx.backward_render(gi, cfg["bg"], step.rgb)
step._rgb_gather = gdist.all_gather_blocks(step.rgb_all, step.rgb, async_op=True)
step.backward_gaussians_chunked(cam)
assert len(step._chunk_reduces) == 4
chunked = step.exchange_gradients(cam).clone()
torch.cuda.synchronize()
step.ctx.pack_gradients_global(step.grads, L, N, full)  # the full rows of THIS backward (float atomics: not the first one's bits)
dist.all_reduce(full)
torch.cuda.synchronize()
err = (chunked - full).abs().max().item()
print(f"chunked split exchange over nccl, one rank: max abs diff {err:.3e}")
assert err <= 1e-6 * max(scale, 1e-30) + 1e-12, "chunked exchange differs from the full rows"
dist.barrier()
dist.destroy_process_group()
print("nccl one-rank rehearsal: ok")
