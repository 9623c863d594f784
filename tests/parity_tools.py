"""Bookkeeping for the full-size parity tests: WHERE the HIP path and the oracle differ, and a float64 check that every
such place is a borderline decision (a step function evaluated within rounding of its threshold), not an error.

The three step functions on the path (SURVEY 8a parity hazards 1 and 4):
  * tile membership: 4-axis separating-axis test of the OBB (corners from sin/cos/ceil'ed radii) against closed tile
    rectangles -- an instance may enter or leave a list when a corner projects within rounding of a tile edge;
  * alpha > 1/255 (forward) -- a splat contributes or not;
  * T (1 - alpha) < 1e-4 -- the pixel stops (n and everything behind it change).
"""
import numpy as np

ALPHA_MIN, T_MIN = 1.0 / 255.0, 1e-4


def instance_keys(sorted_ids, ranges):
    """(tile << 32) | compacted gaussian id of every list entry."""
    tile_of = np.repeat(np.arange(len(ranges) - 1, dtype=np.int64), np.diff(ranges))
    return (tile_of << 32) | np.asarray(sorted_ids, np.int64)


def sat_slack(uv, radius, gauss, tile, ntx):
    """Signed slack in PIXELS of the reference's membership test (cuda/culling.cu:97-146) for (gaussian, tile) pairs,
    in float64 from the float32 inputs: min over the four axes of the overlap of the two projected intervals
    (>= 0: the tile is hit).  NaN corners (hazard 2) give +inf: such splats pass everywhere."""
    uv, rad = np.asarray(uv, np.float64)[gauss], np.asarray(radius, np.float64)[gauss]
    u, v, rM, rm, s, c = uv[:, 0], uv[:, 1], rad[:, 0], rad[:, 1], rad[:, 2], rad[:, 3]
    v1x, v1y, v2x, v2y = rM * c, rM * s, -rm * s, rm * c
    ox = np.stack([u - v1x - v2x, u + v1x - v2x, u - v1x + v2x, u + v1x + v2x], 1)
    oy = np.stack([v - v1y - v2y, v + v1y - v2y, v - v1y + v2y, v + v1y + v2y], 1)
    tx, ty = (tile % ntx).astype(np.float64) * 16.0, (tile // ntx).astype(np.float64) * 16.0
    cx = np.stack([tx, tx + 16.0, tx, tx + 16.0], 1)
    cy = np.stack([ty, ty, ty + 16.0, ty + 16.0], 1)
    slack = np.full(len(u), np.inf)
    axes = [(np.ones_like(u), np.zeros_like(u)), (np.zeros_like(u), np.ones_like(u)),
            (ox[:, 1] - ox[:, 0], oy[:, 1] - oy[:, 0]), (ox[:, 1] - ox[:, 3], oy[:, 1] - oy[:, 3])]
    with np.errstate(invalid="ignore", divide="ignore"):
        for ax, ay in axes:
            norm = np.sqrt(ax * ax + ay * ay)
            po = ax[:, None] * ox + ay[:, None] * oy
            pt = ax[:, None] * cx + ay[:, None] * cy
            gap = np.minimum(np.nanmax(po, 1) - pt.min(1), pt.max(1) - np.nanmin(po, 1)) / norm
            slack = np.where(np.isnan(gap), slack, np.minimum(slack, gap))
    return slack


def pixel_margins(px, py, ref, ranges, sorted_ids, ntx, upto):
    """For one pixel: the smallest relative distance of any alpha on its list to 1/255 and of any running
    T (1 - alpha) to 1e-4, evaluated in float64 as cuda/render.cu:64-87 does, over the first `upto` list entries."""
    tile = (py // 16) * ntx + px // 16
    ids = sorted_ids[ranges[tile]:ranges[tile] + upto]
    uv, con = ref["uv"][ids].astype(np.float64), ref["conic"][ids].astype(np.float64)
    opa = 1.0 / (1.0 + np.exp(-ref["opacity"][ids].astype(np.float64)))
    dx, dy = uv[:, 0] - px, uv[:, 1] - py
    power = np.minimum(0.0, -0.5 * (con[:, 0] * dx * dx + 2.0 * con[:, 1] * dx * dy + con[:, 2] * dy * dy))
    alpha = np.minimum(0.99, opa * np.exp(power))
    a_margin = np.abs(alpha - ALPHA_MIN).min() / ALPHA_MIN if len(ids) else np.inf
    T, t_margin = 1.0, np.inf
    for a in np.where(alpha > ALPHA_MIN, alpha, 0.0):
        test = T * (1.0 - a)
        t_margin = min(t_margin, abs(test - T_MIN) / T_MIN)
        if test < T_MIN:
            break
        T = test
    return a_margin, t_margin


def forward_parity_report(fwd_np, ref, W, H, pixel_tol=1e-4):
    """fwd_np: the HIP forward's arrays as numpy (image T n sorted ranges radius); ref: the oracle's dict.
    Returns a dict of the real figures; `explain` then checks every difference."""
    ntx = (W + 15) // 16
    err = np.abs(fwd_np["image"].astype(np.float64) - ref["image"].astype(np.float64)).sum(-1)
    kg, kr = instance_keys(fwd_np["sorted"], fwd_np["ranges"]), instance_keys(ref["sorted"], ref["ranges"])
    only_gpu, only_ref = np.setdiff1d(kg, kr), np.setdiff1d(kr, kg)
    rad_diff = np.nonzero((fwd_np["radius"][:, :2] != ref["radius"][:, :2]).any(1) &
                          ~(np.isnan(fwd_np["radius"][:, :2]) & np.isnan(ref["radius"][:, :2])).all(1))[0]
    return dict(ntx=ntx, err=err, max_l1=float(err.max()), p9999_l1=float(np.quantile(err, 0.9999)),
                mean_l1=float(err.mean()), frac_above=float((err > pixel_tol).mean()),
                n_mismatch=int((fwd_np["n"] != ref["n"]).sum()), only_gpu=only_gpu, only_ref=only_ref,
                radius_diff=rad_diff, S_gpu=len(kg), S_ref=len(kr))


def explain(report, fwd_np, ref, W, H, pixel_tol=1e-4, slack_tol_px=1e-3, alpha_rel_tol=2e-3, max_pixels=4000):
    """Asserts that every difference in `report` is a borderline decision.  Tolerances: an OBB/tile-edge slack within
    1e-3 px (a few ulp of a ~1000 px coordinate; sin/cos/atan2 differ between ocml and libm); alpha or T(1-alpha)
    within 2e-3 relative of its threshold (one ulp of a pixel-space mean times the conic's slope at the 1/255 level).
    Returns the worst margins actually seen, for the log."""
    ntx = report["ntx"]
    worst = dict(slack_px=0.0, alpha_rel=0.0)
    # gaussians whose ceil()'ed radius differs: off by exactly one pixel (pre-ceil value at an integer)
    rd = report["radius_diff"]
    if len(rd):
        d = np.abs(fwd_np["radius"][rd, :2].astype(np.float64) - ref["radius"][rd, :2].astype(np.float64))
        assert np.nanmax(d) <= 1.0, f"ceil'ed radii differ by more than one pixel: {np.nanmax(d)}"
    odd = set(int(g) for g in rd)
    tiles_touched = set()
    for keys in (report["only_gpu"], report["only_ref"]):
        if not len(keys):
            continue
        tile, g = (keys >> 32).astype(np.int64), (keys & 0xFFFFFFFF).astype(np.int64)
        tiles_touched.update(int(t) for t in tile)
        plain = np.array([int(x) not in odd for x in g])
        if plain.any():
            s = np.abs(sat_slack(ref["uv"], ref["radius"], g[plain], tile[plain], ntx))
            worst["slack_px"] = max(worst["slack_px"], float(s.max()))
            assert s.max() <= slack_tol_px, f"an instance differs although its OBB clears the tile edge by {s.max():.3e} px"
    bad = np.argwhere((report["err"] > pixel_tol) | (fwd_np["n"] != ref["n"]))
    assert len(bad) <= max_pixels, f"{len(bad)} differing pixels"
    unexplained = []
    for py, px in bad:
        tile = (py // 16) * ntx + px // 16
        if tile in tiles_touched:
            continue  # a borderline instance (checked above) sits on this tile's list
        upto = int(max(fwd_np["n"][py, px], ref["n"][py, px]))
        a_m, t_m = pixel_margins(int(px), int(py), ref, ref["ranges"], ref["sorted"], ntx, upto)
        m = min(a_m, t_m)
        worst["alpha_rel"] = max(worst["alpha_rel"], float(m))
        if m > alpha_rel_tol:
            unexplained.append((int(px), int(py), float(report["err"][py, px]), float(a_m), float(t_m)))
    assert not unexplained, f"pixels differ without a borderline alpha / T on their list: {unexplained[:5]}"
    return worst
