#!/usr/bin/env python3
"""Where render_bwd's wave-cycles go: runs the benchmark workload on a -DGS_STAMP=1 build of the library
(GSPLAT_LIB=tools/ab/libstamp.so) and summarises the per-wave stamps: cycles at barriers / staging / list building /
the trip loop / the flush, the slot occupancy over the launch (tail), and the per-tile duration spread.

    GSPLAT_LIB=tools/ab/libstamp.so python tools/bwd_timeline.py [workload]
"""
import ctypes, importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
lib = importlib.import_module("3dgs_amd._lib").load()
name = sys.argv[1] if len(sys.argv) > 1 else "config3"
which = sys.argv[2] if len(sys.argv) > 2 else "bwd"   # "fwd": render_fwd's stamps (flush / barrier-after-flush columns unused)
N, W, H, L, _ = scene.WORKLOADS[name]
cfg = scene.CONFIG
params = scene.make_workload_gaussians(name)
dp, dc = raster.device_params(params), raster.device_camera(scene.make_camera(W, H, 0))
dgi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
ctx = raster.RasterContext(N, W, H)
grads = ctx.alloc_gradients(N, L)
for _ in range(20):
    fwd = ctx.rasterize_image(dp, dc, cfg, cfg["bg"], L)
    ctx.backward_pass(dp, dc, dgi, cfg["bg"], L, grads)
torch.cuda.synchronize()
ntiles = ((W + 15) // 16) * ((H + 15) // 16)
nblocks = (ntiles + 7) // 8 * 8
WORDS = 16
buf = np.zeros(nblocks * 4 * WORDS, np.uint64)
reader = lib.gsplat_debug_read_stamps_fwd if which == "fwd" else lib.gsplat_debug_read_stamps
reader.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = reader(buf.ctypes.data, buf.size)
assert rc == 0, rc
s = buf.reshape(nblocks * 4, WORDS).astype(np.float64)
s = s[s[:, 2] > 0]  # waves that ran (padding blocks and skipped tiles write nothing)
tile, rt0, rt1, cyc, bar0, stage, lists, loop, flush, trips, batches, xcc, bar1, bar2, bar3, wv = s.T
bar = bar0 + bar1 + bar2 + bar3
print(f"{name} ({which}): {len(s)} waves stamped, {len(np.unique(tile))} tiles")
t0 = rt0.min(); dur_us = (rt1.max() - t0) / 100.0  # s_memrealtime ticks at 100 MHz
print(f"launch span {dur_us:.1f} us (first wave start -> last wave end)")
tot = cyc.sum()
for nm, v in (("barrier waits", bar), ("staging (global loads, LDS stores, acc clear)", stage), ("row-list building", lists),
              ("trip loop", loop), ("flush" if which != "fwd" else "staging: waiting for the id + record loads (stamp build waits here)", flush)):
    print(f"  {nm:48s} {100 * v.sum() / tot:5.1f} % of wave cycles")
for nm, v in (("  barrier at batch top (after flush step 2)", bar0), ("  barrier after staging", bar1), ("  barrier after the loop", bar2), ("  barrier after flush step 1", bar3)):
    print(f"  {nm:48s} {100 * v.sum() / tot:5.1f} %   by wave 0..3: " + " ".join(f"{100 * v[wv == w].sum() / cyc[wv == w].sum():.1f}" for w in range(4)))
for nm, v in (("stage", stage), ("lists", lists), ("loop", loop), ("flush", flush)):
    print(f"  {nm:10s} by wave 0..3: " + " ".join(f"{100 * v[wv == w].sum() / cyc[wv == w].sum():.1f}" for w in range(4)))
print(f"  unaccounted {100 * (1 - (bar + stage + lists + loop + flush).sum() / tot):.1f} %")
print(f"trips {int(trips.sum())}, loop cycles per trip {loop.sum() / trips.sum():.1f}; batches {int(batches.sum())}; "
      f"mean wave life {cyc.mean():.0f} cycles = {(rt1 - rt0).mean() / 100:.1f} us; effective clock "
      f"{cyc.sum() / ((rt1 - rt0).sum() / 100) / 1e3:.2f} GHz")
# slot occupancy over time: waves alive / (256 CUs * 4 SIMDs * 7)
edges = np.linspace(t0, rt1.max(), 41)
alive = [((rt0 < b) & (rt1 > a)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 1/40 of the launch (capacity 7168):", " ".join(str(int(a)) for a in alive))
occ = ((rt1 - rt0).sum()) / ((rt1.max() - t0) * 7168)
print(f"slot occupancy over the launch span: {100 * occ:.1f} %")
# imbalance inside a workgroup: trip counts vs loop cycles of its four waves
tl_sorted = np.argsort(tile, kind="stable")
tt, ll, ww = trips[tl_sorted].reshape(-1, 4), loop[tl_sorted].reshape(-1, 4), (bar2[tl_sorted]).reshape(-1, 4)
print(f"per tile, over its 4 waves: sum(max trips)/sum(mean trips) = {tt.max(1).sum() / tt.mean(1).sum():.3f}; "
      f"sum(max loop cycles)/sum(mean loop cycles) = {ll.max(1).sum() / ll.mean(1).sum():.3f}")
np.savez_compressed(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", f"bwd_stamps_{name}.npz"), stamps=s.astype(np.float32))
per_tile_us = {}
for tl in np.unique(tile):
    m = tile == tl
    per_tile_us[int(tl)] = (rt1[m].max() - rt0[m].min()) / 100.0
v = np.array(list(per_tile_us.values()))
print(f"tile duration us: mean {v.mean():.1f} median {np.median(v):.1f} p90 {np.quantile(v, .9):.1f} max {v.max():.1f}")
# the tail: when did the last 10 % / 1 % of the wave-time finish
order = np.argsort(rt1)
print(f"time at which 90 % of the waves had ended: {(np.quantile(rt1, .9) - t0) / 100:.1f} us, 99 %: {(np.quantile(rt1, .99) - t0) / 100:.1f} us")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCC {x}: {m.sum()} waves, last end {(rt1[m].max() - t0) / 100:.1f} us, wave-time {((rt1 - rt0)[m]).sum() / 100 / 1e3:.1f} ms")
json.dump(dict(workload=name, launch_us=dur_us, occupancy=occ, frac=dict(barrier=bar.sum() / tot, stage=stage.sum() / tot,
          lists=lists.sum() / tot, loop=loop.sum() / tot, flush=flush.sum() / tot), cycles_per_trip=loop.sum() / trips.sum()),
          open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", f"bwd_timeline_{name}.json"), "w"), indent=1)
