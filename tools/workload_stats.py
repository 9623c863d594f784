#!/usr/bin/env python3
"""Tile-list statistics and per-stage times of a named workload: python tools/workload_stats.py <workload> [reps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
name = sys.argv[1] if len(sys.argv) > 1 else "garden1200k"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, W, H, L, _ = scene.WORKLOADS[name]
cfg = scene.CONFIG
if name == "garden1200k" and len(sys.argv) > 3:  # tuning: scale cluster_fraction cull op_lo op_hi
    a = [float(x) for x in sys.argv[3:8]]
    params = scene.make_garden_like(N, W, H, L, splat_scale=a[0], cluster_fraction=a[1], cull=a[2], opacity_range=(a[3], a[4]))
elif name == "veiled1200k" and len(sys.argv) > 3:  # tuning: faint_fraction faint_scale faint_sigma op_lo op_hi
    a = [float(x) for x in sys.argv[3:8]]
    params = scene.make_veiled(N, W, H, L, faint_fraction=a[0], faint_scale=a[1], faint_sigma=a[2], faint_opacity=(a[3], a[4]))
else:
    params = scene.make_workload_gaussians(name)
dp = raster.device_params(params); dc = raster.device_camera(scene.make_camera(W, H, 0))
dgi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
ctx = raster.RasterContext(N, W, H)
ctx.set_lean_forward(os.environ.get('GSPLAT_STATS_FULL') != '1')  # GSPLAT_STATS_FULL=1: every ForwardPassData array stored
if os.environ.get('GSPLAT_STATS_ROUTE'):  # 1: counting sort + per-tile sorts, 2: rocPRIM radix sorts (A/B of the binning route)
    ctx.set_binning_route(int(os.environ['GSPLAT_STATS_ROUTE']))
grads = ctx.alloc_gradients(N, L)
for _ in range(5):
    fwd = ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
lens = (fwd["ranges"][1:] - fwd["ranges"][:-1]).float()
q = torch.quantile(lens, torch.tensor([0.5, 0.9, 0.99], device=lens.device)).tolist()
print(f"{name}: N {N}, M {fwd['num_culled']}, pairs {fwd['num_pairs']}, S {fwd['num_splats']} ({fwd['num_splats'] / fwd['num_culled']:.2f} per "
      f"gaussian), tiles {lens.numel()}, list length mean {lens.mean().item():.0f} median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} "
      f"max {lens.max().item():.0f}; lists > 1024: {(lens > 1024).sum().item()}, > 2048: {(lens > 2048).sum().item()}, "
      f"> 4096: {(lens > 4096).sum().item()}, > 8192: {(lens > 8192).sum().item()}")
# load balance of the 256 persistent workgroups of preprocess / bin_scatter: instances per slice of global indices
cnt = torch.bincount(fwd["sorted"].long(), minlength=fwd["num_culled"]).float()
sl = (fwd["compact_to_global"].long() * 256 // N)
per = torch.zeros(256, device=cnt.device).index_add_(0, sl, cnt)
kept = torch.zeros(256, device=cnt.device).index_add_(0, sl, torch.ones_like(cnt))
print(f"  per contiguous slice (256): instances max/mean {per.max().item() / per.mean().item():.2f}, kept gaussians max/mean "
      f"{kept.max().item() / kept.mean().item():.2f}; interleaved 64-chunks: instances max/mean "
      f"{(lambda q: q.max().item() / q.mean().item())(torch.zeros(256, device=cnt.device).index_add_(0, (fwd['compact_to_global'].long() // 64) % 256, cnt)):.2f}")
ctx.set_timing(True)
for _ in range(reps):
    ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
st = ctx.get_timing(); ctx.set_timing(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
torch.cuda.synchronize()
print(f"  {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per fwd+bwd; stages (ms): " + ", ".join(f"{k} {v[0]:.3f}" for k, v in st.items() if v[1]))
c = ctx.counters()
n = fwd["n"].view(-1).float()
print(f"  stop indices: mean {n.mean().item():.0f} max {n.max().item():.0f}; longest chain {c['longest_chain']}, chain sum "
      f"{c['chain_sum']} (x 2048 / sum = {c['longest_chain'] * 2048 / max(1, c['chain_sum']):.2f}: the forward splits above 3); "
      f"segmented forwards {c['segmented_forwards']}, backwards {c['segmented_backwards']} of {c['forwards']} forwards")
