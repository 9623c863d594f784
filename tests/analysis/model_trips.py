"""Numpy model of compositing loop trips on the benchmark scene: today's (quadrant, gaussian) visits vs per-row
(4x4 sub-block) queues where a wave's trip count is the max over its four sub-blocks.  CPU only; uses the oracle, hence it lives under tests/ (test infrastructure)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
scene = importlib.import_module("3dgs_amd.scene")
from oracle import oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
N, W, H, L, _ = scene.WORKLOADS[name]
params = scene.make_gaussians(N, W, H, L)
cam = scene.make_camera(W, H, 0)
c = scene.CONFIG
f = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
uv, conic, sorted_ids, ranges = f["uv"], f["conic"], f["sorted"], f["ranges"]
opa = 1.0 / (1.0 + np.exp(-f["opacity"].reshape(-1))) if "opacity" in f else None
if opa is None:
    mask = f["mask"]
    opa = 1.0 / (1.0 + np.exp(-np.asarray(params["opacity"], np.float32).reshape(-1)[mask]))
a, b, cc = conic[:, 0], conic[:, 1], conic[:, 2]
det = a * cc - b * b
tau2 = 2.0 * np.maximum(0.0, np.log(255.0 * opa)) + 1e-3
hx = np.sqrt(tau2 * cc / det) * 1.0005 + 0.01
hy = np.sqrt(tau2 * a / det) * 1.0005 + 0.01
hx[opa * 255 < 0.999] = -np.inf
hy[opa * 255 < 0.999] = -np.inf
ntx = (W + 15) // 16
T = len(ranges) - 1
tile_of = np.repeat(np.arange(T), np.diff(ranges))
g = sorted_ids
tx0 = (tile_of % ntx) * 16.0
ty0 = (tile_of // ntx) * 16.0
lox, hix, loy, hiy = uv[g, 0] - hx[g], uv[g, 0] + hx[g], uv[g, 1] - hy[g], uv[g, 1] + hy[g]


def hits1d(lo, hi, o, size, nblk):
    return [(~(hi < o + k * size)) & (~(lo > o + k * size + size - 1)) for k in range(nblk)]


# quadrants
qx, qy = hits1d(lox, hix, tx0, 8, 2), hits1d(loy, hiy, ty0, 8, 2)
visits = sum((qx[i] & qy[j]).sum() for i in range(2) for j in range(2))
print("instances S", len(g), "quadrant visits", visits, "per instance", visits / len(g))
# 4x4 sub-blocks, AABB
bx, by = hits1d(lox, hix, tx0, 4, 4), hits1d(loy, hiy, ty0, 4, 4)
trips = 0
pairs = 0
for wy in range(2):
    for wx in range(2):
        cnts = []
        for ry in range(2):
            for rx in range(2):
                hit = bx[wx * 2 + rx] & by[wy * 2 + ry]
                cnts.append(np.bincount(tile_of[hit], minlength=T))
                pairs += hit.sum()
        trips += np.max(np.stack(cnts), axis=0).sum()
print("sub-block pairs", pairs, "row-queue trips (AABB)", trips, "ratio visits/trips", visits / trips,
      "row occupancy", pairs / (4 * trips))


# exact ellipse-vs-rectangle test: min over the rectangle of q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau2
def ellipse_hits_rect(u, v, a, b, c, tau2, x0, x1, y0, y1):
    # closest point in q-metric: check centre inside, else minimise along the 4 edges
    inside = (u >= x0) & (u <= x1) & (v >= y0) & (v <= y1)
    best = np.full(u.shape, np.inf)
    for (xe) in (x0, x1):
        dx = xe - u
        # minimise over y in [y0,y1]: q = a dx^2 + 2 b dx dy + c dy^2 ; dy* = -b dx / c
        dy = np.clip(-b * dx / c, y0 - v, y1 - v)
        best = np.minimum(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    for (ye) in (y0, y1):
        dy = ye - v
        dx = np.clip(-b * dy / a, x0 - u, x1 - u)
        best = np.minimum(best, a * dx * dx + 2 * b * dx * dy + c * dy * dy)
    return inside | (best <= tau2)


ag, bg_, cg, tg = a[g], b[g], cc[g], tau2[g]
ug, vg = uv[g, 0], uv[g, 1]
ok = opa[g] * 255 >= 0.999
trips_e = 0
pairs_e = 0
visits_e = 0
for wy in range(2):
    for wx in range(2):
        cnts = []
        x0, y0 = tx0 + wx * 8, ty0 + wy * 8
        visits_e += (ellipse_hits_rect(ug, vg, ag, bg_, cg, tg, x0, x0 + 7, y0, y0 + 7) & ok).sum()
        for ry in range(2):
            for rx in range(2):
                xs, ys = tx0 + (wx * 2 + rx) * 4, ty0 + (wy * 2 + ry) * 4
                hit = ellipse_hits_rect(ug, vg, ag, bg_, cg, tg, xs, xs + 3, ys, ys + 3) & ok
                cnts.append(np.bincount(tile_of[hit], minlength=T))
                pairs_e += hit.sum()
        trips_e += np.max(np.stack(cnts), axis=0).sum()
print("exact: quadrant visits", visits_e, "sub-block pairs", pairs_e, "row-queue trips", trips_e,
      "ratio visits/trips", visits / trips_e, "row occupancy", pairs_e / (4 * trips_e))


# ---- lane utilisation of the (instance, block) pairs the AABB test admits: pixels with alpha >= 1/255 (no saturation)
def lane_utilisation(sample=400000, seed=1):
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(g), size=min(sample, len(g)), replace=False)
    gi = g[idx]
    ox, oy = tx0[idx], ty0[idx]
    px = np.arange(16, dtype=np.float32)
    X = ox[:, None] + px[None, :]            # [n,16] pixel x
    Y = oy[:, None] + px[None, :]
    dx = uv[gi, 0][:, None] - X               # [n,16]
    dy = uv[gi, 1][:, None] - Y
    A, B, C = a[gi][:, None, None], b[gi][:, None, None], cc[gi][:, None, None]
    power = -0.5 * (A * dx[:, None, :] ** 2 + C * dy[:, :, None] ** 2) - B * dx[:, None, :] * dy[:, :, None]  # [n,y,x]
    alpha = np.minimum(0.99, opa[gi][:, None, None] * np.exp(np.minimum(power, 0)))
    valid = alpha >= 1.0 / 255.0              # [n,16,16]
    blk = valid.reshape(-1, 4, 4, 4, 4).sum(axis=(2, 4))  # [n, by, bx] valid pixels per 4x4 block
    hitx = np.stack([(~(hix[idx] < ox + k * 4)) & (~(lox[idx] > ox + k * 4 + 3)) for k in range(4)], 1)  # [n,bx]
    hity = np.stack([(~(hiy[idx] < oy + k * 4)) & (~(loy[idx] > oy + k * 4 + 3)) for k in range(4)], 1)
    admitted = hity[:, :, None] & hitx[:, None, :]
    assert (blk[~admitted] == 0).all(), "AABB test must be conservative"
    v = blk[admitted]
    print("admitted (instance, block) pairs per instance", admitted.sum() / len(idx), "with no valid pixel", (v == 0).mean(),
          "mean valid lanes of 16", v.mean(), "histogram", np.bincount(v, minlength=17) / len(v))


if os.environ.get("GS_MODEL_LANES"):
    lane_utilisation()


# ---- candidate block tests between the footprint box and the exact one: + the two principal axes of the ellipse
def obb_trips():
    ag, bg2, cg, tg = a[g], b[g], cc[g], tau2[g]
    ug, vg = uv[g, 0], uv[g, 1]
    ok = opa[g] * 255 >= 0.999
    # eigen-decomposition of [[a, b], [b, c]]: q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau2
    mean = 0.5 * (ag + cg)
    diff = 0.5 * (ag - cg)
    rad = np.sqrt(diff * diff + bg2 * bg2)
    l1, l2 = mean + rad, mean - rad            # l1 >= l2 > 0
    th = 0.5 * np.arctan2(2 * bg2, ag - cg)    # direction of the l1 axis
    e1x, e1y = np.cos(th), np.sin(th)
    r1 = np.sqrt(tg / l1)                      # semi-axis along e1 (short), along e2: sqrt(tau2 / l2) (long)
    r2 = np.sqrt(tg / np.maximum(l2, 1e-30))
    trips = 0
    pairs = 0
    for wy in range(2):
        for wx in range(2):
            cnts = []
            for ry in range(2):
                for rx in range(2):
                    bx_, by_ = wx * 2 + rx, wy * 2 + ry
                    xs, ys = tx0 + bx_ * 4, ty0 + by_ * 4
                    hit = bx[bx_] & by[by_]
                    cxb, cyb = xs + 1.5 - ug, ys + 1.5 - vg          # block centre relative to the gaussian
                    p1 = cxb * e1x + cyb * e1y
                    p2 = -cxb * e1y + cyb * e1x
                    h = 1.5 * (np.abs(e1x) + np.abs(e1y))
                    hit = hit & (np.abs(p1) - h <= r1) & (np.abs(p2) - h <= r2) & ok
                    cnts.append(np.bincount(tile_of[hit], minlength=T))
                    pairs += hit.sum()
            trips += np.max(np.stack(cnts), axis=0).sum()
    print("box + principal axes: sub-block pairs", pairs, "row-queue trips", trips)


if os.environ.get("GS_MODEL_OBB"):
    obb_trips()
