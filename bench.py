#!/usr/bin/env python3
"""Benchmark of the rasterizer hot path on MI355X (contract: see the task description / DESIGN.md).

One "step" = one forward + backward pass of one training view per GPU
(gsplat_rasterize_image + gsplat_backward_pass, i.e. rasterize_image + the 7-operator backward chain of the
reference) on BASELINE.json configs[2]: 1e6 synthetic gaussians, 1920x1080, SH degree 3, with inputs resident in
HBM.  With --gpus N > 1 every rank renders its own view and the per-gaussian gradients are summed with one RCCL
all-reduce per step (weak scaling: value = views/s over all ranks).

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    scene = importlib.import_module("3dgs_amd.scene")
    raster = importlib.import_module("3dgs_amd.raster")
    gdist = importlib.import_module("3dgs_amd.dist")
    import torch.distributed as dist

    rank, world, local_rank = gdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device((local_rank % torch.cuda.device_count()) if world > 1 else 0)
    dev = torch.device("cuda", torch.cuda.current_device())

    N, W, H, L, do_bwd = scene.WORKLOADS[args.workload]
    cfg = scene.CONFIG
    t0 = time.time()
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H, view_index=rank)  # every rank its own training view
    gi = scene.make_grad_image(W, H)
    dp, dc = raster.device_params(params, dev), raster.device_camera(cam, dev)
    dgi = torch.as_tensor(gi).to(dev)
    step = gdist.ViewShardedStep(dp, L, W, H, cfg, cfg["bg"], exchange=os.environ.get("GSPLAT_EXCHANGE", "split"))
    gen_s = time.time() - t0

    def one_step():
        if do_bwd:
            return step.step(dc, dgi)
        return step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)

    for _ in range(args.warmup):
        fwd = one_step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()  # a generation-2 collection in the middle of a timed loop costs tens of milliseconds
    # Inside the timed region only the dominant kernel is bracketed by HIP events (the roofline's launch duration):
    # every timed stage costs two event records per step, all eight together 5 % of the step.
    dom = "render_backward" if do_bwd else "render_forward"
    step.ctx.set_timing(True, stages=[dom])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dom_ms = step.ctx.get_timing()[dom][0]
    step.ctx.set_timing(False)

    if rank != 0:
        gc.enable()
        if world > 1:
            dist.barrier()
        return

    # ---- workload statistics and the roofline of the dominant kernel (compositing backward)
    n = fwd["n"]
    P = W * H
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    npad = torch.zeros(nty * 16, ntx * 16, dtype=n.dtype, device=dev)
    npad[:H, :W] = n
    S_eff = int(npad.reshape(nty, 16, ntx, 16).amax(dim=(1, 3)).sum().item())
    M, S = fwd["num_culled"], fwd["num_splats"]
    alg_bytes = (76 * S_eff + 20 * P) if do_bwd else (40 * S_eff + 20 * P)
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")  # filled from the rocprofv3 --pmc passes (profiles/README.md)
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(dom)
        except Exception:
            traffic = None

    # ---- per-stage times: a separate pass with every stage bracketed by events, outside the timed region
    step.ctx.set_timing(True)
    for _ in range(max(5, min(args.steps, 50))):  # rank 0 alone: no exchange here
        step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        if do_bwd:
            step.ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, step.grads)
    stages = step.ctx.get_timing()
    step.ctx.set_timing(False)

    # ---- forward-only rate (render fps), outside the timed region
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reps = max(5, min(args.steps, 30))
    call_ms = []
    for _ in range(reps):
        ta = time.perf_counter()
        step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        call_ms.append((time.perf_counter() - ta) * 1e3)
    torch.cuda.synchronize()
    fps = reps / (time.perf_counter() - t1)
    # the same in a render-only context (serving: nothing kept for a backward)
    step.ctx.set_render_only(True)
    reps_ro = max(5, min(args.steps, 100))
    for _ in range(5):
        step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps_ro):
        step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    torch.cuda.synchronize()
    fps_render_only = reps_ro / (time.perf_counter() - t1)
    step.ctx.set_render_only(False)

    # ---- the whole training iteration (SURVEY 8f rows f1 + f2 around the path): rasterize -> fused L1+SSIM loss ->
    # backward with the uv intermediates -> masked in-place Adam; outside the timed region, last (it moves the parameters)
    train_ms = None
    if do_bwd and os.environ.get("GSPLAT_BENCH_TRAIN_STEP", "1") != "0":
        ops = importlib.import_module("3dgs_amd.ops")
        opt_mod = importlib.import_module("3dgs_amd.optimizer")
        target = step.ctx.rasterize_image(dp, dc, cfg, 0.0, L)["image"].clone()
        opt = opt_mod.AdamOptimizer(dp, L, scene_extent=5.0)
        tgrads = step.ctx.alloc_gradients(N, L, intermediates=True)
        loss_grad = torch.empty(H, W, 3, device=dev)

        def train_step(it):
            f = step.ctx.rasterize_image(dp, dc, cfg, 0.0, L)
            ops.fused_loss(f["image"], target, H, W, 0.2, loss_grad, blocking=False)
            step.ctx.backward_pass(dp, dc, loss_grad, 0.0, L, tgrads)
            opt.step(it, f, tgrads)

        for it in range(10):
            train_step(it)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps_tr = max(5, min(args.steps, 50))
        for it in range(reps_tr):
            train_step(10 + it)
        torch.cuda.synchronize()
        train_ms = (time.perf_counter() - t1) / reps_tr * 1e3
    gc.enable()
    if os.environ.get("GSPLAT_BENCH_DEBUG"):
        print("forward-only host ms per call:", " ".join(f"{t:.2f}" for t in call_ms), file=sys.stderr)

    # ---- CPU baseline: the oracle (a CPU restatement of the reference; the reference has no CPU rasterizer)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        cores = min(os.cpu_count() or 1, 16)  # the GPU box's CPU share for one GPU
        reps_cpu = 2
        t2 = time.perf_counter()
        for _ in range(reps_cpu):
            f = orc.rasterize(params, cam, cfg["near_thresh"], cfg["mh_dist"], cfg["cull_mask_padding"], cfg["bg"], L,
                              threads=cores)
            if do_bwd:
                orc.backward_pass(f, cam, gi, cfg["bg"], L, threads=cores)
        cpu_s = (time.perf_counter() - t2) / reps_cpu
        cpu = {"value": 1.0 / cpu_s, "unit": "it/s", "cores": cores, "kind": "port",
               "sample": f"{reps_cpu} full iterations of the same workload ({N} gaussians, {W}x{H}, SH {L}); "
                         f"compositing on {cores} OpenMP threads, per-gaussian operators and sort single-threaded"}

    ms = elapsed / args.steps * 1e3
    origin = {"config2": "BASELINE configs[1]", "config3": "BASELINE configs[2]"}.get(args.workload,
                                                                                     f"'{args.workload}' (not a BASELINE config)")
    line = {
        "metric": (f"fwd+bwd iterations/s (one view per GPU), {N:.0e} gaussians @{W}x{H}, SH deg {L}".replace("e+0", "e")
                   if do_bwd else f"forward renders/s, {N:.0e} gaussians @{W}x{H}, SH deg {L}".replace("e+0", "e")),
        "value": world * args.steps / elapsed, "unit": "it/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{origin}: synthetic {N} gaussians, {W}x{H}, SH deg {L}, "
                               f"{'forward+backward' if do_bwd else 'forward'}",
                   "views_per_step": world, "parallelism": f"view-sharded dp{world}" if world > 1 else "single GPU",
                   "exchange": step.describe_exchange() if world > 1 else "none",
                   "M": M, "S": S, "S_eff": S_eff, "num_pairs": fwd["num_pairs"], "scene_seed": scene.SEED},
        "render_fps_forward_only": fps, "render_fps_render_only_context": fps_render_only,
        "train_step_ms_with_loss_and_adam": train_ms,
        "stage_ms": {k: round(v[0], 4) for k, v in stages.items()},
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes,
                     "avg_launch_ms": dom_ms},
        # the HBM-bound kernels either side of the compositing, from the per-stage pass (algorithmic bytes per
        # gaussian: SURVEY.md 8d / DESIGN.md section 4); not the dominant kernel, reported for completeness
        "roofline_per_gaussian_kernels": [
            {"kernel": k, "bound": "hbm", "algorithmic_bytes": int(bpg * M), "avg_launch_ms": round(stages[k][0], 4),
             "achieved": (bpg * M / (stages[k][0] * 1e-3) / 1e9) if stages[k][0] > 0 else 0.0, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": (bpg * M / (stages[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if stages[k][0] > 0 else 0.0}
            for k, bpg in (("preprocess", 301), ("preprocess_backward", 560)) if (do_bwd or k == "preprocess")],
        "cpu_baseline": cpu,
        "setup_s": round(gen_s, 1),
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
