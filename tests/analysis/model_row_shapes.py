"""What would other row shapes buy the compositing loops?  For a sample of the benchmark scene's tiles: exact per-pixel
validity (alpha >= 1/255) of every list entry, then for a block shape bw x bh (a "row" of bw*bh lanes owns one block,
a wave holds 64/(bw*bh) rows over its 8x8 quadrant): (instance, block) pairs with at least one valid pixel (what an
exact block test admits), the share of valid lanes in those pairs, and wave trips = sum over (tile, wave) of the longest
of the wave's row lists.  CPU only (oracle = test infrastructure).  Round-2 question: do 2x4-pixel blocks (8-lane rows,
8 lists per wave) cut the invalid lanes by a quarter?
"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
scene = importlib.import_module("3dgs_amd.scene")
from oracle import oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N, W, H, L, _ = scene.WORKLOADS[name]
params = scene.make_gaussians(N, W, H, L)
cam = scene.make_camera(W, H, 0)
c = scene.CONFIG
orc.set_threads(8)
f = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
uv, conic, srt, ranges = f["uv"].astype(np.float64), f["conic"].astype(np.float64), f["sorted"], f["ranges"]
opa = 1.0 / (1.0 + np.exp(-f["opacity"].astype(np.float64)))
ntx = (W + 15) // 16
T = len(ranges) - 1
rng = np.random.default_rng(0)
tiles = rng.choice(T - ntx, n_tiles, replace=False)  # skip the half-empty last tile row
shapes = [(4, 4), (4, 2), (2, 4), (8, 2), (2, 2), (8, 8)]
acc = {s: dict(pairs=0, lanes=0, valid=0, trips=0) for s in shapes}
inst = 0
yy, xx = np.mgrid[0:16, 0:16]
for t in tiles:
    ids = srt[ranges[t]:ranges[t + 1]]
    if len(ids) == 0:
        continue
    inst += len(ids)
    px = (t % ntx) * 16 + xx[None]
    py = (t // ntx) * 16 + yy[None]
    dx = uv[ids, 0][:, None, None] - px
    dy = uv[ids, 1][:, None, None] - py
    a, b, cc = conic[ids, 0][:, None, None], conic[ids, 1][:, None, None], conic[ids, 2][:, None, None]
    power = np.minimum(0.0, -0.5 * (a * dx * dx + 2 * b * dx * dy + cc * dy * dy))
    valid = np.minimum(0.99, opa[ids][:, None, None] * np.exp(power)) >= 1.0 / 255.0   # [n,16,16] (y, x)
    for (bw, bh) in shapes:
        v = valid.reshape(len(ids), 16 // bh, bh, 16 // bw, bw).sum(axis=(2, 4))     # valid pixels per block [n, by, bx]
        hit = v > 0
        A = acc[(bw, bh)]
        A["pairs"] += int(hit.sum()); A["lanes"] += int(hit.sum()) * bw * bh; A["valid"] += int(v.sum())
        # waves = the four 8x8 quadrants; rows of a wave = the blocks inside its quadrant
        per_block = hit.sum(0)                                                      # list length per block [by, bx]
        qb_y, qb_x = 8 // bh if bh <= 8 else 1, 8 // bw if bw <= 8 else 1
        if bw == 8 and bh == 8:
            A["trips"] += int(per_block.sum())                                      # one block per wave: all 64 lanes one pair
        else:
            q = per_block.reshape(2, qb_y, 2, qb_x).transpose(0, 2, 1, 3).reshape(4, -1)
            A["trips"] += int(q.max(1).sum())
print(f"{name}: {n_tiles} tiles, {inst} instances")
base = acc[(4, 4)]
for s in shapes:
    A = acc[s]
    rows = 64 // (s[0] * s[1])
    print(f"block {s[0]}x{s[1]} ({rows:2d} rows/wave): pairs/instance {A['pairs'] / inst:5.2f}  valid lanes {100 * A['valid'] / A['lanes']:5.1f} %  "
          f"invalid lanes vs 4x4 {A['lanes'] - A['valid']:>9d} ({(A['lanes'] - A['valid']) / (base['lanes'] - base['valid']):.2f}x)  "
          f"wave trips {A['trips']:>8d} ({A['trips'] / base['trips']:.2f}x of 4x4)  lane-slots {A['trips'] * 64 / base['trips'] / 64:.2f}x")
