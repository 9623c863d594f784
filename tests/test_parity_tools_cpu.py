"""The full-size parity bookkeeping (tests/parity_tools.py) must be able to FAIL: identical outputs pass with zero
differences, borderline differences are explained, and a pixel / instance / radius that differs without a step
function evaluated at its threshold is reported.  CPU only: the oracle plays both sides."""
import numpy as np
import pytest

import parity_tools


@pytest.fixture(scope="module")
def case(scene, orc):
    N, W, H, L = 4000, 160, 96, 1
    params, cam, c = scene.make_gaussians(N, W, H, L), scene.make_camera(W, H, 1), scene.CONFIG
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=4)
    return ref, W, H


def _as_fwd(ref):
    return {k: np.array(ref[k], copy=True) for k in ("image", "T", "n", "sorted", "ranges", "radius")}


def test_identical_outputs_have_no_differences(case):
    ref, W, H = case
    f = _as_fwd(ref)
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    assert rep["max_l1"] == 0 and rep["n_mismatch"] == 0 and len(rep["only_gpu"]) == 0 and len(rep["only_ref"]) == 0
    assert parity_tools.explain(rep, f, ref, W, H) == dict(slack_px=0.0, alpha_rel=0.0)


def test_sat_slack_agrees_with_the_oracles_lists(case):
    """Every listed instance has a non-negative float64 slack (up to rounding), every unlisted tile of a gaussian's
    neighbourhood a negative one: the numpy restatement of the membership test is the oracle's."""
    ref, W, H = case
    ntx = (W + 15) // 16
    keys = parity_tools.instance_keys(ref["sorted"], ref["ranges"])
    tile, g = keys >> 32, keys & 0xFFFFFFFF
    s = parity_tools.sat_slack(ref["uv"], ref["radius"], g, tile, ntx)
    assert (s > -1e-3).all()
    have = set(int(k) for k in keys)
    rng = np.random.default_rng(0)
    gg = rng.integers(0, len(ref["uv"]), 4000)
    tt = rng.integers(0, len(ref["ranges"]) - 1, 4000)
    absent = np.array([((int(t) << 32) | int(x)) not in have for t, x in zip(tt, gg)])
    s2 = parity_tools.sat_slack(ref["uv"], ref["radius"], gg[absent], tt[absent], ntx)
    # an absent pair either fails the separating-axis test or lies outside the coarse rectangle (slack then unknown)
    assert (s2 < 1e-3).mean() > 0.95


def test_a_wrong_pixel_is_reported(case):
    ref, W, H = case
    f = _as_fwd(ref)
    # the pixel whose list is farthest from any threshold
    ntx = (W + 15) // 16
    best, where = 0.0, None
    for py in range(8, H - 8, 7):
        for px in range(8, W - 8, 11):
            a, t = parity_tools.pixel_margins(px, py, ref, ref["ranges"], ref["sorted"], ntx, int(ref["n"][py, px]))
            if min(a, t) > best and np.isfinite(min(a, t)):
                best, where = min(a, t), (py, px)
    assert best > 0.01
    f["image"][where[0], where[1], 1] += 2e-3
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    assert rep["frac_above"] > 0
    with pytest.raises(AssertionError, match="borderline"):
        parity_tools.explain(rep, f, ref, W, H)


def test_a_missing_instance_is_reported(case):
    ref, W, H = case
    f = _as_fwd(ref)
    ntx = (W + 15) // 16
    keys = parity_tools.instance_keys(ref["sorted"], ref["ranges"])
    s = parity_tools.sat_slack(ref["uv"], ref["radius"], keys & 0xFFFFFFFF, keys >> 32, ntx)
    k = int(np.argmax(np.where(np.isfinite(s), s, -1)))  # the instance deepest inside its tile
    tile = int(keys[k] >> 32)
    f["sorted"] = np.delete(ref["sorted"], k)
    f["ranges"] = ref["ranges"].copy()
    f["ranges"][tile + 1:] -= 1
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    assert len(rep["only_ref"]) == 1 and len(rep["only_gpu"]) == 0
    with pytest.raises(AssertionError, match="clears the tile edge"):
        parity_tools.explain(rep, f, ref, W, H)


def test_a_radius_off_by_more_than_one_is_reported(case):
    ref, W, H = case
    f = _as_fwd(ref)
    f["radius"][5, 0] += 2.0
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    assert list(rep["radius_diff"]) == [5]
    with pytest.raises(AssertionError, match="more than one pixel"):
        parity_tools.explain(rep, f, ref, W, H)
    f["radius"][5, 0] -= 1.0  # off by exactly one: a pre-ceil value at an integer, accepted
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    parity_tools.explain(rep, f, ref, W, H)
