"""One-rank rehearsal of the gradient exchange on the REAL backend (nccl = RCCL): the collectives degenerate to copies,
but tensor shapes, contiguity and the async handles go through the same torch.distributed / RCCL calls as the multi-GPU
run (RCCL refuses two ranks on one device, so this is as close as a single GPU gets).

  python tools/nccl_one_rank.py [workload]           (default workload: small; config3 = the benchmark scene)

Checks every split payload (split, split with 4 ranges, split_direct) against the full rows, then times the whole
view-sharded step with and without the exchange and reports the HOST's share of it -- the microseconds per step the
rank's Python + torch.distributed calls take (ViewShardedStep.host_s) -- as one JSON line: the first real 8-GPU run can
be compared with it (bench.py carries it as exchange_host_cost_one_rank)."""
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
raster = importlib.import_module("3dgs_amd.raster")
scene = importlib.import_module("3dgs_amd.scene")
gdist = importlib.import_module("3dgs_amd.dist")

workload = sys.argv[1] if len(sys.argv) > 1 else "small"
os.environ["GSPLAT_DIST_BACKEND"] = "nccl"
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
N, W, H, L, _ = scene.WORKLOADS[workload]
dev = torch.device("cuda:0")
dp = raster.device_params(scene.make_workload_gaussians(workload), dev)
cam = raster.device_camera(scene.make_camera(W, H, 0), dev)
gi = torch.as_tensor(scene.make_grad_image(W, H)).to(dev)
cfg = scene.CONFIG
ctx = raster.RasterContext(N, W, H)
ctx.set_lean_forward(True)
full = torch.empty(N, raster.packed_gradient_width(L), device=dev)
report = {"workload": workload, "backend": "nccl, one rank (collectives degenerate to copies)"}
ref_grads = ctx.alloc_gradients(N, L)
for name, kw in (("split", dict(exchange="split", chunks=1)), ("split_packed", dict(exchange="split_packed", chunks=1)),
                 ("split_chunks4", dict(exchange="split", chunks=4)),
                 ("split_direct", dict(exchange="split_direct")), ("factored", dict(exchange="factored")),
                 ("full", dict(exchange="full"))):
    step = gdist.ViewShardedStep(dp, L, W, H, cfg, cfg["bg"], ctx=ctx, exchange_at_world_one=True, **kw)
    assert step.world == 1
    step.step(cam, gi)
    packed = step.packed.clone()  # (split: materialised here, on demand, from common + rgb_all)
    torch.cuda.synchronize()
    # reference: the full rows of a plain backward of the same forward through one all-reduce
    ctx.backward_pass(dp, cam, gi, cfg["bg"], L, ref_grads)
    ctx.pack_gradients_global(ref_grads, L, N, full)
    dist.all_reduce(full)
    torch.cuda.synchronize()
    err, scale = (packed - full).abs().max().item(), full.abs().max().item()
    print(f"{name} exchange over nccl, one rank: max abs diff {err:.3e} (scale {scale:.3e})", file=sys.stderr)
    assert err <= 1e-6 * max(scale, 1e-30) + 1e-12, f"{name} exchange differs from the full rows"
    reps = 30 if N >= 100_000 else 100
    for _ in range(5):
        step.step(cam, gi)
    torch.cuda.synchronize()
    step.host_s = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        step.step(cam, gi)
    torch.cuda.synchronize()
    report[name] = {"ms_per_step_with_exchange": round((time.perf_counter() - t0) / reps * 1e3, 4),
                    "host_us_per_step_in_exchange_calls": round(step.host_s / reps * 1e6, 1),
                    "collectives": step.describe_exchange()}
    del step
    torch.cuda.empty_cache()
# the same step with no exchange at all (a group of one skips it): what the figures above are on top of
step = gdist.ViewShardedStep(dp, L, W, H, cfg, cfg["bg"], ctx=ctx, exchange="split")
for _ in range(5):
    step.step(cam, gi)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 30 if N >= 100_000 else 100
for _ in range(reps):
    step.step(cam, gi)
torch.cuda.synchronize()
report["ms_per_step_without_exchange"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
dist.barrier()
dist.destroy_process_group()
print(json.dumps(report))
print("nccl one-rank rehearsal: ok", file=sys.stderr)
