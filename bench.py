#!/usr/bin/env python3
"""Benchmark of the rasterizer hot path on MI355X (contract: see the task description / DESIGN.md section 5).

One "step" = one forward + backward pass of one training view per GPU
(gsplat_rasterize_image + gsplat_backward_pass, i.e. rasterize_image + the 7-operator backward chain of the
reference) on BASELINE.json configs[2]: 1e6 synthetic gaussians, 1920x1080, SH degree 3, with inputs resident in
HBM.  With --gpus N > 1 every rank renders its own view and the per-gaussian gradients are summed over the ranks
once per step (weak scaling: value = views/s over all ranks).

`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (one child process per GPU, started
before anything in this process touches the GPU); under `python -m torch.distributed.run` the ranks come from the
environment.  A mismatch between --gpus and the world size is an error, never a silent single-GPU run.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec; 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every 2 cycles
# (`v_fma_f32` (wave64): 2 cyc), 2.4 GHz max clock -> 1228.8 G wave-level VALU instructions per second
HBM_PEAK_GBS = 8000.0
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the non-headline workloads (half-culled config 3, dense4m) reported as extra keys")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args):
    """Parent of a multi-GPU run: never imports torch, never touches a GPU.  Builds the library ONCE (the children load
    it with GSPLAT_NO_BUILD=1: N ranks running `make` in one directory at the same time could dlopen a half-linked
    file), starts one child per rank with the torch.distributed environment, forwards rank 0's JSON line.  All children
    are polled under ONE overall deadline (GSPLAT_BENCH_DEADLINE_S, default 540 s: below the 600 s the driver allows a
    run); the first child that exits non-zero, or the deadline, ends the run: the remaining children are killed (these
    exact processes), so that a rank stuck in a collective its peers never joined cannot hold the GPUs.  The exit code
    follows the HEADLINE: once rank 0 has printed its line with n_gpus == N (it does so only after the timed region of
    all ranks, and whatever the optional payload sweep behind it does -- see HeadlineGuard) the run is a success."""
    import tempfile
    n = args.gpus
    importlib.import_module("3dgs_amd._lib").build()  # make only: no torch, no GPU in this process
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", GSPLAT_NO_BUILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else None, text=True))
    deadline = time.time() + float(os.environ.get("GSPLAT_BENCH_DEADLINE_S", "540"))
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() > deadline:
            failed = f"rank(s) {bad} failed" if bad else "deadline passed"
            grace = time.time() + (3.0 if bad else 0.0)  # ranks that fail for the same reason report it themselves
            for r, p in enumerate(procs):  # these exact children
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=max(0.0, grace - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    text = out0.read()
    out0.close()
    sys.stdout.write(text)
    sys.stdout.flush()
    line = [ln for ln in text.splitlines() if ln.startswith("{")]
    have_headline = False
    try:
        have_headline = bool(line) and json.loads(line[-1]).get("n_gpus") == n
    except ValueError:
        have_headline = False
    if failed or any(c != 0 for c in codes):
        print(f"bench.py: {failed or 'a rank failed'}; rank exit codes {codes}"
              + ("; the headline line had already been printed" if have_headline else ""), file=sys.stderr)
        return 0 if have_headline else 1
    if not have_headline:
        print(f"bench.py: rank 0 did not report n_gpus={n}", file=sys.stderr)
        return 1
    return 0


class SweepAbort(RuntimeError):
    """The optional payload sweep cannot go on (a payload failed on some ranks only, or a peer went silent): the ranks'
    collectives no longer pair up.  The headline is unaffected -- it was measured, on a communicator of its own, before."""


class HeadlineGuard:
    """The ONE JSON line of rank 0, printed exactly once whatever the optional legs behind the timed region do.
    arm(): from now on a timer owns the line -- if finish() has not been called within `deadline_s`, the timer thread
    prints the line as it stands (plus `on_timeout`) and ends the process with exit code 0 via os._exit (the main thread
    may be stuck inside a collective that will never complete; a clean teardown of the process groups could hang too).
    Ranks other than the reporter arm the same timer with line=None: they only leave.  finish(extra) cancels the timer
    and prints the line with `extra` merged in."""

    def __init__(self, line, deadline_s, on_timeout=None, out=None):
        import threading
        self.line, self.deadline_s, self.on_timeout = line, float(deadline_s), dict(on_timeout or {})
        self.out = out if out is not None else sys.stdout
        self._lock, self._printed, self._timer = threading.Lock(), False, None

    def _emit(self, extra):
        with self._lock:
            if self._printed:
                return False
            self._printed = True
        if self.line is not None:
            self.out.write(json.dumps({**self.line, **extra}) + "\n")
            self.out.flush()
        return True

    def arm(self):
        import threading

        def fire():
            if self._emit(self.on_timeout):
                sys.stderr.write(f"bench.py: the optional legs behind the timed region did not finish within "
                                 f"{self.deadline_s:.0f} s; headline kept, leaving\n")
                sys.stderr.flush()
                os._exit(0)
        self._timer = threading.Timer(self.deadline_s, fire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def finish(self, extra=None):
        if self._timer is not None:
            self._timer.cancel()
        return self._emit(extra or {})


def agree_on_payload(comm, mode, ok, timeout_s=120.0):
    """Every rank reports through the group's key-value store (NOT a collective) whether payload `mode` worked for it,
    and reads all reports.  True: worked everywhere; False: failed everywhere (the mode is dropped on every rank).
    A mixed outcome, or a rank that never reports because it is still inside a collective the failing rank left, ends
    the SWEEP (SweepAbort): after an asymmetric failure the ranks' collectives no longer pair up.  The headline has been
    measured by then and is printed regardless (HeadlineGuard)."""
    rank, world = comm.rank, comm.world
    if comm.backend() == "threads":  # in-process ranks: a shared table; a rank that died has broken the barrier
        votes = comm.group.shared
        votes[("exchange", mode, rank)] = ok
        comm.barrier()
        got = [votes[("exchange", mode, r)] for r in range(world)]
        if all(got) or not any(got):
            return all(got)
        raise SweepAbort(f"payload '{mode}' failed on some ranks only: {got}")
    from datetime import timedelta
    from torch.distributed.distributed_c10d import _get_default_store
    store = _get_default_store()
    store.set(f"gsplat/exchange/{mode}/{rank}", "ok" if ok else "fail")
    keys = [f"gsplat/exchange/{mode}/{r}" for r in range(world)]
    try:
        store.wait(keys, timedelta(seconds=timeout_s))
    except Exception as e:  # noqa: BLE001 -- a peer never reported
        raise SweepAbort(f"payload '{mode}': a peer did not report within {timeout_s:.0f} s ({type(e).__name__})")
    votes = [store.get(k).decode() for k in keys]
    if all(v == "ok" for v in votes):
        return True
    if all(v == "fail" for v in votes):
        return False
    raise SweepAbort(f"payload '{mode}' failed on some ranks only ({votes})")


# split_packed = the split payload + the packed[N, 12 + 3 n] rows materialised (what r04's default step produced)
SWEEP_MODES = ("split", "split_packed", "split_chunks4", "split_direct", "factored", "full")


def sweep_payloads(comm, modes, time_mode, torch, dev, vote_timeout_s=30.0):
    """The measured comparison of the exchange payloads: time_mode(mode) -> ms per step on this rank, or raises.  A
    payload that fails on EVERY rank is reported and left out (a decision of all ranks through the store, no collective);
    one that fails on some ranks only raises SweepAbort (agree_on_payload).  Returns {mode: ms, MAX over the ranks}."""
    out = {}
    for mode in modes:
        ms_local, err = None, None
        try:
            ms_local = time_mode(mode)
        except SweepAbort:
            raise
        except Exception as e:  # noqa: BLE001 -- whatever the backend raises
            err = f"{type(e).__name__}: {e}"[:200]
            print(f"bench.py: exchange payload '{mode}' failed on rank {comm.rank}: {err}", file=sys.stderr, flush=True)
        if agree_on_payload(comm, mode, err is None, timeout_s=vote_timeout_s):
            t = torch.tensor([ms_local], dtype=torch.float64, device=dev)
            comm.all_reduce_max(t)
            out[mode] = round(float(t.item()), 4)
        else:
            out[mode] = "failed: " + err
        if dev is not None and getattr(dev, "type", "cpu") == "cuda":
            torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ helpers
def spawning_legs_allowed():
    """The reference_host_path and exchange_host_cost legs start child processes (and, for the former, possibly hipcc)
    from a process whose GPU is initialised.  Under rocprofv3 the children would run under the profiler's preload too --
    and on this pool a profiled process that replaces or forks programs is exactly what must not happen -- so both legs
    default OFF when a profiler preload is visible in the environment; GSPLAT_BENCH_REFERENCE_HOST=1 /
    GSPLAT_BENCH_EXCHANGE_HOST_COST=1 force them on."""
    pre = os.environ.get("LD_PRELOAD", "")
    profiled = "rocprof" in pre.lower() or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ)
    return not profiled


def tile_max_sum(torch, n, W, H):
    """S_eff = sum over tiles of max_px(splats_per_pixel): list entries any pixel of the tile needs."""
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    npad = torch.zeros(nty * 16, ntx * 16, dtype=n.dtype, device=n.device)
    npad[:H, :W] = n
    return int(npad.reshape(nty, 16, ntx, 16).amax(dim=(1, 3)).sum().item())


def stage_pass(ctx, dp, dc, dgi, cfg, L, grads, reps, do_bwd=True):
    ctx.set_timing(True)
    for _ in range(reps):
        ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        if do_bwd:
            ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
    st = ctx.get_timing()
    ctx.set_timing(False)
    return st


def hbm_entry(kernel, nbytes, ms, moved=None):
    ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {"kernel": kernel, "bound": "hbm", "algorithmic_bytes": int(nbytes), "avg_launch_ms": round(ms, 4),
           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    if moved is not None:  # what this instantiation of the kernel actually reads and writes (see preprocess_bytes)
        out["bytes_moved"] = int(moved)
        out["frac_on_bytes_moved"] = (moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0
    return out


def preprocess_bytes(M, N, L, lean):
    """(SURVEY 8d algorithmic bytes, bytes this instantiation moves) of the fused per-gaussian forward.  The survey's
    figure is the reference's operator chain: 288 B per visible gaussian at SH 3 (+ 13 B per gaussian for the cull).  The
    kernel here reads per visible gaussian xyz 12 + band0 12 + sh 12((L+1)^2-1) + opacity 4 + scale 12 + quaternion 16 +
    mask 1 + rank 4 and writes rank 4 + compact_to_global 4 + xyz_c 12 + uv 8 + radius 16 + the 48-byte splat record +
    tile count 4 + hit mask 8 = 104 B; with every ForwardPassData array stored also Sigma 24 + J 24 + conic 12 + colour
    12.  Culled gaussians cost the cull's 12 B read + 5 B written (lean) or 25 B (full: uncompacted xyz_c and uv)."""
    rest = 12 * ((L + 1) ** 2 - 1)
    alg = (12 + 16 + 12 + 4 + 12 + rest + 8 + 12 + 12 + 16 + 4) * M + 13 * N  # 288 B at SH 3
    read = 12 + 12 + rest + 4 + 12 + 16 + 1 + 4
    write = 104 + (0 if lean else 72)
    return alg, (read + write) * M + (12 + (5 if lean else 25)) * N


def profiled_counters(workload):
    """(counters dict | None, note | None) for `workload` from profiles/traffic.json -- the PMC passes of
    tools/refresh_profiles.sh: HBM bytes per launch ((2 FETCH_SIZE + WRITE_SIZE) * 1024) and SQ_INSTS_VALU of the
    compositing kernels -- reported only when they were measured on the sources the LOADED library was built from."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None, None
    try:
        tj = json.load(open(tfile))
        have = importlib.import_module("3dgs_amd._lib").library_source_hash()
        if tj.get("source_sha16") != have:
            return None, (f"profile stale: profiles/traffic.json was measured on sources {tj.get('source_sha16')}, "
                          f"this library is built from {have}; traffic / VALU counters not reported")
        return (tj if workload == "config3" else tj.get("workloads", {}).get(workload)), None
    except Exception:
        return None, None


def compositing_rooflines(S_eff, P, fwd_ms, bwd_ms, counters):
    """HBM roofline entries of the two compositing kernels on SURVEY 8d's algorithmic bytes (40 / 76 B per needed list
    entry + 20 B per pixel), with the PMC traffic and -- against the bound they actually run at -- the VALU issue figures
    when counters of this build are at hand."""
    out = {}
    for key, kern, per, ms in (("render_forward", "render_fwd", 40, fwd_ms), ("render_backward", "render_bwd", 76, bwd_ms)):
        if not ms or ms <= 0:
            continue
        e = hbm_entry(kern, per * S_eff + 20 * P, ms)
        if counters and counters.get(key):
            e["traffic"] = counters[key]
            e["traffic_over_algorithmic"] = round(counters[key] / e["algorithmic_bytes"], 3)
        if counters and counters.get(key + "_valu_insts"):
            ginst = counters[key + "_valu_insts"] / (ms * 1e-3) / 1e9
            e["valu_issue"] = {"valu_instructions_per_launch": counters[key + "_valu_insts"], "achieved": ginst,
                               "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": ginst / VALU_PEAK_GINST}
        out[key] = e
    return out


def extra_workload(torch, scene, raster, name, dev, reps=20):
    """A non-headline workload, per-stage times only (outside every timed region)."""
    N, W, H, L, _ = scene.WORKLOADS[name]
    cfg = scene.CONFIG
    params = scene.make_workload_gaussians(name)
    cam = scene.make_camera(W, H, 0)
    dp, dc = raster.device_params(params, dev), raster.device_camera(cam, dev)
    dgi = torch.as_tensor(scene.make_grad_image(W, H)).to(dev)
    ctx = raster.RasterContext(N, W, H)
    ctx.set_lean_forward(True)
    grads = ctx.alloc_gradients(N, L)
    for _ in range(5):
        fwd = ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
    M, S = fwd["num_culled"], fwd["num_splats"]
    S_eff = tile_max_sum(torch, fwd["n"], W, H)
    st = stage_pass(ctx, dp, dc, dgi, cfg, L, grads, reps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    lens = (fwd["ranges"][1:] - fwd["ranges"][:-1])
    train_it_s = None
    if name == "garden1200k":
        # the whole training iteration on this workload (rasterize -> fused loss -> backward -> masked Adam), as for the
        # headline scene below: the figure next to the from-disk 1.26 M-gaussian run of DESIGN section 9
        ops = importlib.import_module("3dgs_amd.ops")
        opt_mod = importlib.import_module("3dgs_amd.optimizer")
        dpt = {k: v.clone() for k, v in dp.items()}
        target = ctx.rasterize_image(dpt, dc, cfg, 0.0, L)["image"].clone()
        opt = opt_mod.AdamOptimizer(dpt, L, scene_extent=5.0)
        tg = ctx.alloc_gradients(N, L, intermediates=("uv",))
        lg = torch.empty(H, W, 3, device=dev)

        def train_step(it):
            f = ctx.rasterize_image(dpt, dc, cfg, 0.0, L)
            ops.fused_loss(f["image"], target, H, W, 0.2, lg, blocking=False)
            ctx.backward_pass(dpt, dc, lg, 0.0, L, tg)
            opt.step(it, f, tg)

        for it in range(5):
            train_step(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(reps):
            train_step(5 + it)
        torch.cuda.synchronize()
        train_it_s = reps / (time.perf_counter() - t0)
        # the compositing kernels INSIDE that training iteration (loss and optimizer between them): their launch times
        ctx.set_timing(True, stages=["render_forward", "render_backward"])
        for it in range(reps):
            train_step(5 + reps + it)
        train_stage = ctx.get_timing()
        ctx.set_timing(False)
        del dpt, opt, tg, lg, target
    out = {"N": N, "M": M, "S": S, "num_pairs": fwd["num_pairs"], "tile_list_mean": round(float(lens.float().mean().item()), 1),
           "tile_list_max": int(lens.max().item()), "ms_per_step": round(ms, 4), "stage_ms": {k: round(v[0], 4) for k, v in st.items()},
           "preprocess": hbm_entry("preprocess (lean)", preprocess_bytes(M, N, L, True)[0], st["preprocess"][0],
                                   preprocess_bytes(M, N, L, True)[1]),
           "preprocess_backward": hbm_entry("preprocess_backward", 560 * M, st["preprocess_backward"][0])}
    counters, note = profiled_counters(name)
    out["S_eff"] = S_eff
    out["roofline"] = compositing_rooflines(S_eff, W * H, st["render_forward"][0], st["render_backward"][0], counters)
    if note:
        out["roofline"]["note"] = note
    if name == "veiled1200k":
        # the scene of the long-list segments: what the context did, and the forward of a second context
        # whose gate keeps one workgroup per tile (the backward's split has no per-context switch: GSPLAT_NO_SEGMENTS=1)
        c = ctx.counters()
        ctx.close()
        ctx = raster.RasterContext(N, W, H)
        ctx.set_lean_forward(True)
        ctx.set_segment_options(gate=1e9)
        for _ in range(5):
            ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
            ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
        whole = stage_pass(ctx, dp, dc, dgi, cfg, L, grads, reps)
        out["segments"] = {"segmented_forwards": c["segmented_forwards"], "segmented_backwards": c["segmented_backwards"],
                           "forwards": c["forwards"], "longest_chain": c["longest_chain"], "chain_sum": c["chain_sum"],
                           "longest_chain_over_work_per_resident_workgroup": round(c["longest_chain"] * 2048 / max(1, c["chain_sum"]), 2),
                           "render_forward_ms": round(st["render_forward"][0], 4),
                           "render_forward_ms_one_workgroup_per_tile": round(whole["render_forward"][0], 4),
                           "what": "lists beyond 1488 entries are walked in segments of 496 by workgroups of their own, in both "
                                   "compositing kernels (gs_render.h: TileSegments, FwdSegments); the second forward figure is a "
                                   "context whose gate (gsplat_context_set_segment_options) keeps the forward whole"}
    if train_it_s is not None:
        out["train_it_s"] = round(train_it_s, 1)
        out["roofline_in_training_iteration"] = compositing_rooflines(S_eff, W * H, train_stage["render_forward"][0],
                                                                      train_stage["render_backward"][0], counters)
    ctx.close()
    del dp, dgi, grads
    torch.cuda.empty_cache()
    return out


def config2_workload(torch, scene, raster, dev, reps=100):
    """BASELINE configs[1] -- synthetic 1e5 gaussians, 800x800, SH degree 0, forward render only: the "render fps" half of
    BASELINE.json's metric on the config it is quoted on (parity of this size: tests/test_fused_gpu.py).  Renders/s in a
    render-only context (serving: nothing kept for a backward) and in a training context, per-stage times, and the
    compositing forward's roofline entry on SURVEY 8d's 40 B per needed list entry + 20 B per pixel."""
    N, W, H, L, _ = scene.WORKLOADS["config2"]
    cfg = scene.CONFIG
    dp = raster.device_params(scene.make_workload_gaussians("config2"), dev)
    dc = raster.device_camera(scene.make_camera(W, H, 0), dev)
    ctx = raster.RasterContext(N, W, H)
    out = {"N": N, "width": W, "height": H, "sh_degree": L}
    for key, render_only in (("render_fps_render_only_context", True), ("render_fps_training_context", False)):
        ctx.set_render_only(render_only)
        for _ in range(10):
            fwd = ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
        torch.cuda.synchronize()
        out[key] = reps / (time.perf_counter() - t0)
    M, S = fwd["num_culled"], fwd["num_splats"]
    S_eff = tile_max_sum(torch, fwd["n"], W, H)
    st = stage_pass(ctx, dp, dc, None, cfg, L, None, reps, do_bwd=False)
    lens = fwd["ranges"][1:] - fwd["ranges"][:-1]
    out.update({"M": M, "S": S, "S_eff": S_eff, "num_pairs": fwd["num_pairs"],
                "tile_list_mean": round(float(lens.float().mean().item()), 1), "tile_list_max": int(lens.max().item()),
                "ms_per_render": 1e3 / out["render_fps_training_context"],
                "ms_per_step": 1e3 / out["render_fps_training_context"],  # (forward only: a step of this workload is a render)
                "stage_ms": {k: round(v[0], 4) for k, v in st.items()},
                "roofline": compositing_rooflines(S_eff, W * H, st["render_forward"][0], None, None)})
    ctx.close()
    del dp
    torch.cuda.empty_cache()
    return out


def reference_host_path(params, cam, gi, cfg, L, iterations=20):
    """The path the REFERENCE host drives, timed outside the timed region: tests/cpp/reference_host.cpp is a C++ host
    written against the drop-in headers only (include/gsplat_cuda/raster.cuh, cuda_data.cuh, cuda_backward.cuh) that runs,
    per iteration, what TrainerImpl::train does around the rasterizer (cuda/trainer.cu:1294-1360): a fresh
    ForwardPassData, zero_grads, rasterize_image (the shim: full forward, its output blocks handed to pass_data),
    then backward_pass' eight compact_masked_array calls and the seven stand-alone backward operators
    (cuda/trainer.cu:941-1012).  A child process (its own HIP context); parity of the same binary at this size:
    tests/test_reference_host_gpu.py."""
    import tempfile
    scene_io = importlib.import_module("3dgs_amd.scene_io")
    exe = importlib.import_module("3dgs_amd._lib").build_cpp_host("reference_host")
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, "scene.bin")
        scene_io.write_host_scene(inp, params, cam, gi, cfg, L)
        out = {}
        for mode in ("fresh", "keep"):
            r = subprocess.run([exe, inp, "-", str(iterations)] + (["keep"] if mode == "keep" else []),
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(f"reference_host failed ({r.returncode}): {r.stderr[-300:]}")
            out[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    f = out["fresh"]
    return {"what": "tests/cpp/reference_host.cpp: C++ host against include/gsplat_cuda/*.cuh, per iteration a fresh "
                    "ForwardPassData + zero_grads + rasterize_image shim (all ForwardPassData arrays, handed over without "
                    "copies since r05) + 8 compact_masked_array + the 7 stand-alone backward operators "
                    "(cuda/trainer.cu:941-1012, 1294-1360)",
            "ms_per_iteration": f["ms_per_iteration"], "it_s": 1e3 / f["ms_per_iteration"],
            "ms_zero_grads_and_rasterize_image": f["ms_zero_grads_and_rasterize_image"],
            "ms_backward_pass": f["ms_backward_pass"],
            "ms_per_call_synchronised": f["ms_per_call_synchronised"],
            "ms_per_iteration_one_forward_pass_data_kept": out["keep"]["ms_per_iteration"],
            "iterations": iterations, "pool_bytes": f["pool_bytes"]}


def cpu_baseline(scene, args, params, cam, gi, cfg, N, W, H, L, do_bwd):
    """The oracle (a CPU restatement of the reference's operators; the reference has no CPU rasterizer) timed on this
    box's host cores: one iteration at 1 thread, two at all cores; plus the reference's only CPU compute,
    Gaussians::Initialize (src/gaussian.cpp:38-104), as the oracle's kd-tree restatement on 1e5 / 1e6 points."""
    import numpy as np
    from oracle import oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # the GPU box's CPU share for one GPU

    def iteration(threads):
        orc.set_threads(threads)
        t = time.perf_counter()
        f = orc.rasterize(params, cam, cfg["near_thresh"], cfg["mh_dist"], cfg["cull_mask_padding"], cfg["bg"], L,
                          threads=threads)
        if do_bwd:
            orc.backward_pass(f, cam, gi, cfg["bg"], L, threads=threads)
        return time.perf_counter() - t

    s_all = min(iteration(cores) for _ in range(2))
    s_one = iteration(1)
    orc.set_threads(1)
    init = {}
    rng = np.random.default_rng(0x3D65)
    for n in (100_000, 1_000_000):
        pts = rng.normal(size=(n, 3)) * np.array([4.0, 2.0, 4.0])
        col = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        for th in (cores, 1):
            t = time.perf_counter()
            orc.initialize_gaussians(pts, col, threads=th, kdtree=True)
            init[f"{n}_points_{th}_threads_s"] = round(time.perf_counter() - t, 4)
    return {"value": 1.0 / s_all, "unit": "it/s", "cores": cores, "kind": "port",
            "value_1_thread": 1.0 / s_one, "seconds_per_iteration": {"1_thread": round(s_one, 3),
                                                                      f"{cores}_threads": round(s_all, 3)},
            "sample": f"whole iterations of the same workload ({N} gaussians, {W}x{H}, SH {L}): best of 2 on {cores} "
                      f"OpenMP threads (value), 1 on one thread (value_1_thread); per-gaussian operators and compositing "
                      f"are threaded, tile binning (candidate scan + qsort) is serial at both settings",
            "gaussians_initialize": dict(init, what="oracle restatement of Gaussians::Initialize (src/gaussian.cpp:38-104: "
                                         "kd-tree, leaf 10, 3-NN, OpenMP queries; nanoflann/Eigen are not vendored), "
                                         "normal point clouds, tree build included")}


# ------------------------------------------------------------------------------------------------ one rank
def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    threads = os.environ.get("GSPLAT_BENCH_THREAD_RANKS") == "1" and args.gpus > 1 and env_world is None
    if args.gpus > 1 and env_world is None and not threads:
        sys.exit(launch_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}")
    hang = os.environ.get("GSPLAT_BENCH_SELFTEST", "")  # tests/test_bench_cpu.py: a rank that never returns
    if hang.startswith("hang:") and hang[5:] in ("all", os.environ.get("RANK", "0")):
        time.sleep(3600)

    import torch
    gdist = importlib.import_module("3dgs_amd.dist")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if threads:
        # GSPLAT_BENCH_THREAD_RANKS=1: the N ranks are THREADS of this process on one GPU (dist.ThreadGroup): a rehearsal
        # of rank counts a one-GPU box cannot hold as processes (its process guard admits six; config 5 has eight
        # ranks).  Same code per rank, collectives by rendezvous; the times say nothing about a multi-GPU node.
        importlib.import_module("3dgs_amd._lib").load()
        gdist.ThreadGroup(args.gpus).run(lambda comm: run_rank(args, comm, 0))
        return
    rank, world, local_rank = gdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks")
    comm = gdist.TorchComm()
    ndev = torch.cuda.device_count()
    if world > 1 and comm.backend() == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {ndev} "
                         f"(GSPLAT_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
    if world > 1:
        comm = gdist.TorchComm.own_group()  # the headline's own communicator (closed before the payload sweep)
    status = run_rank(args, comm, local_rank % ndev if world > 1 else 0)
    if world > 1:
        sys.stdout.flush()
        sys.stderr.flush()
        if status != "clean":
            os._exit(0)  # the sweep broke the ranks' pairing: no collective teardown; the headline line is out
        HeadlineGuard(None, 20).arm()  # a teardown that hangs must not turn a finished run into a failure
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def run_rank(args, comm, device_index):
    import numpy as np
    import torch
    scene = importlib.import_module("3dgs_amd.scene")
    raster = importlib.import_module("3dgs_amd.raster")
    gdist = importlib.import_module("3dgs_amd.dist")
    rank, world, backend = comm.rank, comm.world, comm.backend()
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)

    N, W, H, L, do_bwd = scene.WORKLOADS[args.workload]
    cfg = scene.CONFIG
    t0 = time.time()
    params = scene.make_workload_gaussians(args.workload)
    cam = scene.make_camera(W, H, view_index=rank)  # every rank its own training view
    gi = scene.make_grad_image(W, H)
    dp, dc = raster.device_params(params, dev), raster.device_camera(cam, dev)
    dgi = torch.as_tensor(gi).to(dev)
    gen_s = time.time() - t0
    gc.collect()
    gc.disable()  # a generation-2 collection in the middle of a timed loop costs tens of milliseconds

    # ---- exchange payload of the HEADLINE (multi-GPU): GSPLAT_EXCHANGE = split (default) | full | factored | split_direct |
    # split_chunks4.  The default is the conservative one -- two standard collectives, an all-gather and an all-reduce -- and
    # nothing but the headline touches a collective before the line is safe: the measured comparison of all payloads (r02-r04
    # ran it BEFORE the warm-up, where a hang in a payload that had never met real RCCL would have lost the number) now
    # runs behind the timed region, on a communicator of its own, under HeadlineGuard.
    want = os.environ.get("GSPLAT_EXCHANGE", "split")
    want = "split" if want == "auto" else want
    ctx = raster.RasterContext(N, W, H)
    # What the timed forward materialises.  Since r04 the headline stores EVERY ForwardPassData array, as the reference's
    # rasterize_image does (2 % slower than the lean forward, which a host of the fused backward would use: that one is
    # measured beside it, ms_per_step_lean_forward).  GSPLAT_BENCH_LEAN=1 swaps the two.
    lean_headline = os.environ.get("GSPLAT_BENCH_LEAN", "0") == "1"
    ctx.set_lean_forward(lean_headline)

    def make_step(mode, on_comm):
        return gdist.ViewShardedStep(dp, L, W, H, cfg, cfg["bg"], exchange=mode.replace("_chunks4", ""), ctx=ctx, comm=on_comm,
                                     chunks=4 if mode.endswith("chunks4") else None)

    step = make_step(want, comm)

    def one_step():
        if do_bwd:
            return step.step(dc, dgi)
        return step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)

    # ---- reporting passes that need no collective run BEFORE the timed region (they also leave the GPU at its
    # steady clocks, as inside a training run): workload statistics and per-stage times with every stage bracketed
    fwd = one_step()
    P = W * H
    S_eff = tile_max_sum(torch, fwd["n"], W, H)
    M, S, num_pairs = fwd["num_culled"], fwd["num_splats"], fwd["num_pairs"]
    stages = stage_pass(step.ctx, dp, dc, dgi, cfg, L, step.grads, max(5, min(args.steps, 50)), do_bwd)
    # the same forward in the OTHER mode (lean <-> every ForwardPassData array materialised)
    step.ctx.set_lean_forward(not lean_headline)
    stages_other = stage_pass(step.ctx, dp, dc, dgi, cfg, L, step.grads, 10, do_bwd)
    step.ctx.set_lean_forward(lean_headline)
    stages_full, stages_lean = (stages_other, stages) if lean_headline else (stages, stages_other)

    # the whole step in the other mode (the timed region below runs the headline's)
    ms_other_mode = None
    if do_bwd and world == 1:
        step.ctx.set_lean_forward(not lean_headline)
        for _ in range(5):
            one_step()
        torch.cuda.synchronize()
        tf = time.perf_counter()
        reps_f = max(5, min(args.steps, 50))
        for _ in range(reps_f):
            one_step()
        torch.cuda.synchronize()
        ms_other_mode = (time.perf_counter() - tf) / reps_f * 1e3
        step.ctx.set_lean_forward(lean_headline)
        one_step()

    # the step with all TWELVE gradient arrays of GaussianGradients written (cuda_data.cuh: six leaf + the six
    # intermediate ones the reference's operator chain passes along; an optimizer needs the six leaf arrays, which is
    # what the timed step asks for -- gsplat_backward_pass takes NULL for the others)
    ms_all_gradients = None
    if do_bwd and world == 1:
        g12 = step.ctx.alloc_gradients(N, L, intermediates=True)
        for _ in range(3):
            step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
            step.ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, g12)
        torch.cuda.synchronize()
        tg_ = time.perf_counter()
        reps_g = max(5, min(args.steps, 50))
        for _ in range(reps_g):
            step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
            step.ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, g12)
        torch.cuda.synchronize()
        ms_all_gradients = (time.perf_counter() - tg_) / reps_g * 1e3
        del g12
        one_step()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    # Inside the timed region only the dominant kernel is bracketed by HIP events (the roofline's launch duration):
    # every timed stage costs two event records per step, all eight together 5 % of the step.
    dom = "render_backward" if do_bwd else "render_forward"
    step.ctx.set_timing(True, stages=[dom])
    if world > 1:
        comm.barrier()
    torch.cuda.synchronize()
    # Per-step marks cost the GPU nothing: every forward ends its host part by waiting for its own count record (the
    # forward's one read-back), which arrives behind the previous step's backward -- so the host's clock after each step
    # advances by one step of GPU time, and the intervals give a median next to the wall-clock mean below.
    marks = [0.0] * args.steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step()
        marks[k] = time.perf_counter()
    torch.cuda.synchronize()
    if world > 1:
        comm.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        comm.all_reduce_max(t)
        elapsed = float(t.item())
    dom_ms = step.ctx.get_timing()[dom][0]
    step.ctx.set_timing(False)

    # ---- the headline is measured.  Multi-GPU: its communicator is closed here; the optional comparison of the other
    # payloads runs further down on a communicator of its own (rank 0 first prepares the line the guard keeps safe).
    sweep_on = world > 1 and do_bwd and os.environ.get("GSPLAT_BENCH_SWEEP", "1") != "0"
    sweep_deadline = float(os.environ.get("GSPLAT_BENCH_SWEEP_DEADLINE_S", "150"))
    selftest = os.environ.get("GSPLAT_BENCH_SELFTEST", "")

    def time_mode(mode, on_comm):
        """ms per step of payload `mode` on this rank (10 steps after 3), or raises what the backend raised."""
        what = selftest.split(":")
        if what[0] == "raise" and what[1] == mode and (len(what) < 3 or int(what[2]) == rank):
            raise RuntimeError(f"selftest: payload '{mode}' raises on rank {rank}")
        if what[0] == "hangsweep":
            time.sleep(3600)
        st = make_step(mode, on_comm)
        for _ in range(3):
            st.step(dc, dgi)
        torch.cuda.synchronize()
        on_comm.barrier()
        ta = time.perf_counter()
        for _ in range(10):
            st.step(dc, dgi)
        torch.cuda.synchronize()
        on_comm.barrier()
        return (time.perf_counter() - ta) / 10 * 1e3

    def run_sweep():
        """Every payload timed on a fresh communicator; {mode: ms (MAX over ranks) | "failed: ..."}.  Raises SweepAbort /
        whatever a broken backend raises: the caller keeps the headline either way."""
        scomm = comm if backend == "threads" else gdist.TorchComm.own_group()
        coll = None
        if backend != "threads":  # the two collectives of the split exchange, each alone on idle links (all ranks)
            try:
                coll = gdist.time_collectives(scomm, N, dev)
            except Exception as e:  # noqa: BLE001 -- reported, the sweep goes on if the ranks still pair up
                coll = {"error": f"{type(e).__name__}: {e}"[:200]}
        res = sweep_payloads(scomm, SWEEP_MODES, lambda m: time_mode(m, scomm), torch, dev)
        res["_collectives"] = coll
        if scomm is not comm:
            scomm.close()
        return res

    if world > 1 and backend != "threads" and hasattr(comm, "close"):
        comm.barrier()
        comm.close()  # the headline's own communicator: gone before anything optional runs
    if rank != 0:
        status = "clean"
        if sweep_on:
            guard = HeadlineGuard(None, sweep_deadline).arm() if backend != "threads" else None
            try:
                run_sweep()
            except BaseException as e:  # noqa: BLE001 -- SweepAbort, RCCL errors, a broken barrier of rank threads
                print(f"bench.py: rank {rank}: payload sweep ended: {type(e).__name__}: {e}"[:300], file=sys.stderr, flush=True)
                status = "aborted"
            if guard is not None:
                guard.finish()
        if backend != "threads":
            gc.enable()
        return status

    # ---- roofline of the dominant kernel (compositing backward): HBM on algorithmic bytes, and the bound the kernel
    # actually runs against, VALU issue (wave-level VALU instructions per launch from the SQ_INSTS_VALU pass kept in
    # profiles/traffic.json, divided by the launch duration measured live above)
    alg_bytes = (76 * S_eff + 20 * P) if do_bwd else (40 * S_eff + 20 * P)
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = valu_insts = valu_busy = None
    counters, profile_note = profiled_counters(args.workload)
    if counters:
        traffic, valu_insts, valu_busy = counters.get(dom), counters.get(dom + "_valu_insts"), counters.get(dom + "_valu_busy")
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes,
                "avg_launch_ms": dom_ms}
    if profile_note:
        roofline["note"] = profile_note
    roofline_valu = None
    if valu_insts and dom_ms > 0:
        ginst = valu_insts / (dom_ms * 1e-3) / 1e9
        # r03 slope experiment (profiles/r03_valu_slope.txt, tools/experiments/r03_valu_slope.sh): k extra independent
        # v_fma_f32 per trip of render_bwd's loop cost +3.51 us per instruction (k = 4, 8, 16: +16.4, +29.8, +56.2 us on
        # 339.8 us), i.e. 3.61 M wave trips / 1024 SIMDs x 2.2 cycles at the kernel's 2.23 GHz: the FULL nominal issue
        # cost of the guide's table (2 cycles per wave64 VALU instruction per SIMD) plus a tenth -- nothing of an added
        # instruction hides, so the loop is issue bound; removing the 22-instruction row reduction and its LDS atomic took
        # -134 us = 3.9 cycles per instruction (DPP operands, selects).  frac_at_measured_issue_cost prices every VALU
        # instruction of the launch at the measured plain cost (a lower bound: DPP / transcendental / 64-bit cost more).
        cyc_plain, clock_ghz, simds = 2.2, 2.23, 1024
        busy_ms = valu_insts * cyc_plain / simds / (clock_ghz * 1e6)
        roofline_valu = {"kernel": dom, "bound": "valu_issue", "achieved": ginst, "peak": VALU_PEAK_GINST,
                         "unit": "G wave-instructions/s", "frac": ginst / VALU_PEAK_GINST,
                         "valu_instructions_per_launch": valu_insts,
                         "measured_issue_cost": {"plain_valu_cycles_per_instruction_per_simd": cyc_plain,
                                                 "row_reduction_block_cycles_per_instruction": 3.9,
                                                 "note_r04": "measured on r03's loop (51 VALU per trip, 22 of them the row "
                                                             "reduction); r04's loop has 39 (14 + the atomic's address): "
                                                             "profiles/r04_ab_carried_dot_and_rows_reduction.txt",
                                                 "wave_trips_per_launch": 3606560,
                                                 "kernel_clock_ghz": clock_ghz,
                                                 "source": "profiles/r03_valu_slope.txt (same-box A/B, 300 steps, two rounds)"},
                         "frac_at_measured_issue_cost": busy_ms / dom_ms,
                         "note": "achieved = SQ_INSTS_VALU per launch (profiles/traffic.json, same sources) / live launch "
                                 "duration; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 v_fma_f32 "
                                 "(MI355X_MICROARCH.md); frac_at_measured_issue_cost = instructions x 2.2 cycles / "
                                 "(1024 SIMDs x 2.23 GHz x launch duration), every instruction at the plain-FMA cost "
                                 "the slope experiment measured"}

    # ---- forward-only rate (render fps), outside the timed region
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reps = max(5, min(args.steps, 30))
    for _ in range(reps):
        step.ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
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
    train_ms = train_ms_partial = train_ms_fused = train_ms_two = None
    if do_bwd and os.environ.get("GSPLAT_BENCH_TRAIN_STEP", "1") != "0":
        ops = importlib.import_module("3dgs_amd.ops")
        opt_mod = importlib.import_module("3dgs_amd.optimizer")
        dp_train = {k: v.clone() for k, v in dp.items()}
        target = step.ctx.rasterize_image(dp_train, dc, cfg, 0.0, L)["image"].clone()
        opt = opt_mod.AdamOptimizer(dp_train, L, scene_extent=5.0)
        # as the Trainer: density statistics need |grad_uv|, the SH gradients are rebuilt by the optimizer instead of
        # stored and read back (gsplat_optimizer_step_sh_factored)
        tgrads = step.ctx.alloc_gradients(N, L, intermediates=("uv",), factored_sh=True)
        loss_grad = torch.empty(H, W, 3, device=dev)

        def train_step(it, fused):
            f = step.ctx.rasterize_image(dp_train, dc, cfg, 0.0, L)
            ops.fused_loss(f["image"], target, H, W, 0.2, loss_grad, blocking=False)
            if fused == 3:  # r06: the SH group's step in a kernel of its own in front (one read of the coefficient rows),
                # then the per-gaussian backward with the five small groups' steps inside
                step.ctx.backward_pass_adam(dp_train, dc, loss_grad, 0.0, L, opt.fused_state(it, mode=2))
            elif fused == 2:  # r06 (opt-in): the per-gaussian backward applies the optimizer step itself
                step.ctx.backward_pass_adam(dp_train, dc, loss_grad, 0.0, L, opt.fused_state(it))
            elif fused == 1:  # r06: four groups + the statistics inside the backward, SH and position groups behind it
                step.ctx.backward_pass_adam(dp_train, dc, loss_grad, 0.0, L, opt.fused_state(it, mode=1), pgrads)
                opt.step_after_partial_backward(it, f, pgrads, dc["campos"])
            else:      # backward, then the two optimizer kernels on the stored gradients
                step.ctx.backward_pass(dp_train, dc, loss_grad, 0.0, L, tgrads)
                opt.step(it, f, tgrads, campos=dc["campos"])

        reps_tr = max(5, min(args.steps, 50))
        train_variants = {}
        it0 = 0
        pgrads = dict(xyz=tgrads["xyz"], precompute_rgb=tgrads["precompute_rgb"])
        for fused in (0, 1, 2, 3):
            for it in range(10):
                train_step(it0 + it, fused)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for it in range(reps_tr):
                train_step(it0 + 10 + it, fused)
            torch.cuda.synchronize()
            train_variants[fused] = (time.perf_counter() - t1) / reps_tr * 1e3
            it0 += 10 + reps_tr
        train_ms, train_ms_partial, train_ms_fused = train_variants[0], train_variants[1], train_variants[2]
        train_ms_two = train_variants[3]
        del dp_train, opt, tgrads, loss_grad, target

    # ---- three views in turn on one context (extra key): the forward queues its tail -- placement, per-tile sorts,
    # compositing -- BEFORE the host has seen this view's counts, from the previous forward's route, cull ratio and longest
    # list; the timed region repeats one view, so it only ever measures that speculation on a hit.  View B stands 16 units
    # further back (everything in view and closer together on screen: the longest list several times view A's, beyond the
    # sort kernels queued on A's figures: a redo), view C 3 units further in (part of the scene behind it: a different
    # cull ratio, so the walk over compacted slots switches on and off).
    alternating = None
    if world == 1 and do_bwd and args.workload == "config3" and os.environ.get("GSPLAT_BENCH_ALTERNATING", "1") != "0":
        try:
            def moved(tz):
                cm = dict(cam)
                vm = np.array(cam["view"], np.float32).copy()
                vm[11] = tz  # t_z (R = I for view 0): camera position (0, 0, -tz)
                cm["view"], cm["campos"] = vm, np.array([0.0, 0.0, -tz], np.float32)
                return raster.device_camera(cm, dev)
            cams = [("view_a", dc), ("view_b_16_back", moved(16.0)), ("view_c_3_forward", moved(-3.0))]
            actx = raster.RasterContext(N, W, H)
            actx.set_lean_forward(True)
            ag = actx.alloc_gradients(N, L)
            stats = {}
            for k in range(9):
                name, d = cams[k % 3]
                f = actx.rasterize_image(dp, d, cfg, cfg["bg"], L)
                actx.backward_pass(dp, d, dgi, cfg["bg"], L, ag)
                lens = f["ranges"][1:] - f["ranges"][:-1]
                stats[name] = {"M": f["num_culled"], "S": f["num_splats"], "longest_list": int(lens.max().item())}
            c0 = actx.counters()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            reps_a = 60
            for k in range(reps_a):
                d = cams[k % 3][1]
                actx.rasterize_image(dp, d, cfg, cfg["bg"], L)
                actx.backward_pass(dp, d, dgi, cfg["bg"], L, ag)
            torch.cuda.synchronize()
            alt_ms = (time.perf_counter() - ta) / reps_a * 1e3
            c1 = actx.counters()
            for name, d in cams:
                for _ in range(3):
                    actx.rasterize_image(dp, d, cfg, cfg["bg"], L)
                    actx.backward_pass(dp, d, dgi, cfg["bg"], L, ag)
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(20):
                    actx.rasterize_image(dp, d, cfg, cfg["bg"], L)
                    actx.backward_pass(dp, d, dgi, cfg["bg"], L, ag)
                torch.cuda.synchronize()
                stats[name]["ms_per_step_alone"] = (time.perf_counter() - ta) / 20 * 1e3
            alternating = {"ms_per_step_alternating": alt_ms,
                           "ms_per_step_mean_of_the_views_alone": sum(v["ms_per_step_alone"] for v in stats.values()) / 3,
                           "views": stats, "steps": reps_a, "tails_redone": c1["tail_redone"] - c0["tail_redone"],
                           "compact_walks": c1["compact_walks"] - c0["compact_walks"],
                           "instance_buffer_growths": c1["instance_growths"] - c0["instance_growths"]}
            actx.close()
            del ag, cams
            torch.cuda.empty_cache()
        except Exception as e:  # never lose the headline line to a side measurement
            alternating = {"error": repr(e)[:300]}

    # ---- the reference host's own sequence on the headline workload (a child process), before the side workloads run
    ref_host = None
    legs_default = "1" if spawning_legs_allowed() else "0"
    if world == 1 and do_bwd and os.environ.get("GSPLAT_BENCH_REFERENCE_HOST", legs_default) != "0":
        try:
            ref_host = reference_host_path(params, cam, gi, cfg, L)
        except Exception as e:  # never lose the headline line to a side measurement
            ref_host = {"error": repr(e)[:300]}

    # ---- non-headline workloads (extra keys): what real training views do to the per-gaussian kernels and the binning
    extra = None
    if world == 1 and do_bwd and not args.no_extra_workloads and args.workload == "config3":
        extra = {}
        try:  # BASELINE configs[1]: the config the metric's "render fps" is quoted on (forward only)
            extra["config2"] = config2_workload(torch, scene, raster, dev)
        except Exception as e:  # never lose the headline line to a side measurement
            extra["config2"] = {"error": repr(e)[:200]}
        for name in ("config3_halfculled", "config3_morton", "dense4m", "bigsplats", "garden1200k", "veiled1200k"):
            try:
                extra[name] = extra_workload(torch, scene, raster, name, dev)
            except Exception as e:  # never lose the headline line to a side measurement
                extra[name] = {"error": repr(e)[:200]}
    gc.enable()

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(scene, args, params, cam, gi, cfg, N, W, H, L, do_bwd)
        except Exception as e:  # never lose the measured line to the baseline leg
            cpu = {"error": repr(e)[:300]}

    ms = elapsed / args.steps * 1e3
    iv = sorted((b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:]))
    step_stats = ({"median": iv[len(iv) // 2], "min": iv[0], "max": iv[-1], "p90": iv[int(0.9 * (len(iv) - 1))],
                   "what": "host-clock intervals between consecutive steps of the timed region (each forward waits for "
                           "its count record, so an interval is one step of GPU time)"} if iv else None)
    legs_default = "1" if spawning_legs_allowed() else "0"
    # the exchange's HOST cost through the real backend, one rank (a child: it needs a process group of its own)
    host_cost = None
    if world == 1 and do_bwd and args.workload == "config3" and os.environ.get("GSPLAT_BENCH_EXCHANGE_HOST_COST", legs_default) != "0":
        try:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                       HSA_ENABLE_IPC_MODE_LEGACY="0", GSPLAT_NO_BUILD="1")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_one_rank.py"), args.workload], env=env,
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-300:])
            host_cost = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:  # never lose the headline line to a side measurement
            host_cost = {"error": repr(e)[:300]}
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
                   "exchange": step.describe_exchange() if world > 1 else "none", "backend": backend,
                   "forward_outputs": ("lean (gsplat_context_set_lean_forward, GSPLAT_BENCH_LEAN=1): Sigma / J / conic / SH "
                                       "colour of ForwardPassData are not materialised, the fused backward recomputes "
                                       "them; the step with all of them stored: ms_per_step_full_forward_outputs"
                                       if lean_headline else
                                       "all: every ForwardPassData array is stored, as by the reference's rasterize_image "
                                       "(cuda/raster.cu:12-136); the step of a host of the fused backward, which needs "
                                       "none of Sigma / J / conic / SH colour: ms_per_step_lean_forward"),
                   "M": M, "S": S, "S_eff": S_eff, "num_pairs": num_pairs, "scene_seed": scene.SEED},
        # (multi-GPU: filled in by the payload sweep BEHIND the timed region -- every payload's ms per step, MAX over the
        # ranks, measured on a communicator of its own; the headline ran config.exchange)
        "exchange_ms_per_step": None,
        # what the collectives run on -- the world size as the backend's group sees it, the RCCL algorithm / protocol
        # overrides in effect -- and, for N > 1, the split exchange's two collectives timed each ALONE behind the timed
        # region (all-reduce of common[N,12], all-gather of g_rgb[N+1,3] per rank; ms, MAX over ranks): to be read against
        # exchange_model.  null where there is nothing to measure (N = 1).
        "collectives": dict(gdist.collective_environment(comm if world > 1 else None), all_reduce_common_ms=None,
                            all_gather_rgb_ms=None),
        # tools/nccl_one_rank.py: every payload through RCCL with ONE rank (collectives = copies): ms per step with the
        # exchange, and the microseconds of it the host spends in the exchange's Python + torch.distributed calls
        "exchange_host_cost_one_rank": host_cost,
        # what the exchange of this workload moves and costs on xGMI by the link arithmetic of 3dgs_amd/dist.py
        # exchange_model (ring / direct bounds per collective), for the payload in use at this world size and for the
        # 8-rank shape; the driver's measured multi-GPU step times can be checked against it
        "exchange_model": ({"this_run": gdist.exchange_model(world, N, L, "split_direct" if step.direct else step.exchange,
                                                             chunks=step.chunks)} if world > 1 else {})
        | {f"{p}_at_8_ranks": gdist.exchange_model(8, N, L, p) for p in ("full", "factored", "split", "split_direct")},
        "render_fps_forward_only": fps, "render_fps_render_only_context": fps_render_only,
        # (the Trainer's choreography: since r06 the small groups' steps inside the per-gaussian backward, the SH and the
        # position group behind it -- the same number as train_step_ms_small_groups_inside_the_backward below)
        "train_step_ms_with_loss_and_adam": train_ms_partial,
        # backward -> stored gradients -> the two optimizer kernels (the Trainer's choreography of r01-r05)
        "train_step_ms_optimizer_kernels_behind_the_backward": train_ms,
        # r06: the same iteration with the WHOLE optimizer step inside the per-gaussian backward (gsplat_backward_gaussians_adam
        # mode 0, Trainer under GSPLAT_FUSED_ADAM=2): bit-identical results, one kernel instead of three -- and no faster
        "train_step_ms_adam_inside_the_backward": train_ms_fused,
        # ... and with band 0 / opacity / scale / rotation + the statistics inside the backward, SH and position behind it
        "train_step_ms_small_groups_inside_the_backward": train_ms_partial,
        # ... and as two kernels: the SH group's step in front (coefficient rows read once: update + the sums the position
        # gradient needs), then the backward with the five small groups' steps inside (gsplat_adam_fused.mode 2)
        "train_step_ms_sh_step_in_front_of_the_backward": train_ms_two,
        "ms_per_step_stats": step_stats,
        "ms_per_step_full_forward_outputs": ms_other_mode if lean_headline else ms,
        "ms_per_step_lean_forward": ms if lean_headline else ms_other_mode,
        "ms_per_step_all_twelve_gradient_arrays": ms_all_gradients,
        "reference_host_path": ref_host,
        "stage_ms": {k: round(v[0], 4) for k, v in stages.items()},
        "preprocess_ms_all_forward_outputs": round(stages_full["preprocess"][0], 4),
        "preprocess_ms_lean_forward": round(stages_lean["preprocess"][0], 4),
        "roofline": roofline,
        "roofline_valu_issue": roofline_valu,
        # the HBM-bound kernels either side of the compositing, from the per-stage pass (algorithmic bytes per
        # gaussian: SURVEY.md 8d / DESIGN.md section 4); not the dominant kernel, reported for completeness
        "roofline_per_gaussian_kernels": (
            [hbm_entry("preprocess (lean)" + (": what the timed step runs" if lean_headline else ""),
                       preprocess_bytes(M, N, L, True)[0], stages_lean["preprocess"][0], preprocess_bytes(M, N, L, True)[1]),
             hbm_entry("preprocess (all ForwardPassData arrays stored)" + ("" if lean_headline else ": what the timed step runs"),
                       preprocess_bytes(M, N, L, False)[0], stages_full["preprocess"][0], preprocess_bytes(M, N, L, False)[1])]
            + ([hbm_entry("preprocess_backward", 560 * M, stages["preprocess_backward"][0])] if do_bwd else [])),
        "alternating_views": alternating,
        "extra_workloads": extra,
        "cpu_baseline": cpu,
        "setup_s": round(gen_s, 1),
    }
    # ---- ONE line, printed exactly once.  Multi-GPU: first the optional payload sweep, under a guard that prints the
    # line as it stands and leaves with exit code 0 if the sweep hangs; a sweep that fails is reported inside the line.
    status = "clean"
    if not sweep_on:
        HeadlineGuard(line, 0).finish()
        return status
    guard = HeadlineGuard(line, sweep_deadline,
                          on_timeout={"exchange_ms_per_step": {"status": f"sweep did not finish within {sweep_deadline:.0f} s"}})
    guard.arm()
    try:
        sweep = run_sweep()
    except BaseException as e:  # noqa: BLE001
        sweep = {"status": f"sweep ended: {type(e).__name__}: {e}"[:300]}
        status = "aborted"
    coll = sweep.pop("_collectives", None) if isinstance(sweep, dict) else None
    extra_keys = {"exchange_ms_per_step": sweep}
    if isinstance(coll, dict):
        extra_keys["collectives"] = {**line["collectives"], **coll}
    guard.finish(extra_keys)
    return status


if __name__ == "__main__":
    main()
