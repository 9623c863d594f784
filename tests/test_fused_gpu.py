"""GPU parity of the fused per-view pass (gsplat_rasterize_image / gsplat_backward_pass) against the
CPU oracle and the committed golden fixtures, plus size-independent properties at the full
benchmark size (1e6 gaussians, 1920x1080, SH degree 3)."""
import os

import numpy as np
import pytest

import parity_tools
from conftest import (ROOT, assert_grad_close, assert_image_close, assert_stop_indices_close, max_pixels_above_tol,
                      max_stop_index_mismatches, perf_check, pkg)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


def _run(torch, scene, name, view_index, backward=True, intermediates=False):
    raster = pkg("raster")
    N, W, H, L, _ = scene.WORKLOADS[name]
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H, view_index)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    out = dict(params=params, cam=cam, ctx=ctx, dp=dp, dc=dc, fwd=fwd, N=N, W=W, H=H, L=L)
    if backward:
        gi = scene.make_grad_image(W, H)
        grads = ctx.alloc_gradients(fwd["num_culled"], L, intermediates)
        for g in grads.values():
            g.fill_(float("nan"))  # every leaf gradient must be overwritten
        ctx.backward_pass(dp, dc, torch.as_tensor(gi).cuda(), c["bg"], L, grads)
        out.update(gi=gi, grads=grads)
    torch.cuda.synchronize()
    return out


def _np(t):
    return t.detach().cpu().numpy()


def _check_forward(fwd, ref, exact_lists=True):
    assert fwd["num_culled"] == int(ref["mask"].sum())
    assert (_np(fwd["mask"]).astype(bool) == ref["mask"]).all()
    assert fwd["num_pairs"] == int(ref["num_pairs"])
    np.testing.assert_allclose(_np(fwd["conic"]), ref["conic"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(_np(fwd["rgb"]), ref["rgb"], rtol=1e-5, atol=2e-6)
    if exact_lists:  # radii are ceil()ed and the SAT uses sin/cos: identical lists unless a corner sits on a tile edge
        assert fwd["num_splats"] == len(ref["sorted"])
        assert (_np(fwd["ranges"]) == ref["ranges"]).all()
        assert (_np(fwd["sorted"]) == ref["sorted"]).all()
    else:
        assert abs(fwd["num_splats"] - len(ref["sorted"])) <= 1e-5 * len(ref["sorted"]) + 2
    assert_image_close(_np(fwd["image"]), ref["image"], "image")
    assert_image_close(_np(fwd["T"]), ref["T"], "transmittance")
    assert_stop_indices_close(_np(fwd["n"]), ref["n"])
    if all(k in ref for k in ("uv", "radius", "opacity", "conic")) and fwd.get("radius") is not None:
        _tight_bookkeeping(fwd, ref)


def _tight_bookkeeping(fwd, ref):
    """The full-size bars on every scene the oracle renders (r03): per-pixel figures from tests/parity_tools.py, every
    pixel above 1e-4 and every stop-index mismatch must sit on a list with a borderline alpha / T (float64
    re-evaluation), every differing instance on a tile edge; the loose fractions of conftest.py are not what passes a
    scene.  Small images cannot be held to '1e-5 of the pixels' (one pixel of 64x48 is 3e-4), hence the '+ 2'."""
    H, W = ref["n"].shape
    f = {k: _np(fwd[k]) for k in ("image", "T", "n", "sorted", "ranges", "radius")}
    if f["sorted"].shape != np.asarray(ref["sorted"]).shape and abs(len(f["sorted"]) - len(ref["sorted"])) > 1e-5 * len(ref["sorted"]) + 2:
        return  # (already failed above)
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    worst = parity_tools.explain(rep, f, ref, W, H)
    P = W * H
    above = int(round(rep["frac_above"] * P))
    line = (f"[parity {W}x{H}, {rep['S_ref']} instances] per-pixel L1 max {rep['max_l1']:.3e} mean {rep['mean_l1']:.3e}; "
            f"{above} pixels > 1e-4; n mismatches {rep['n_mismatch']}; instances in one list only "
            f"{len(rep['only_gpu'])}+{len(rep['only_ref'])}; radii differing {len(rep['radius_diff'])}; worst margins "
            f"{worst['slack_px']:.2e} px, {worst['alpha_rel']:.2e} rel")
    print(line)
    assert rep["mean_l1"] < 1e-6, line
    assert above <= max_pixels_above_tol(P) and rep["n_mismatch"] <= max_stop_index_mismatches(P), line
    assert len(rep["only_gpu"]) + len(rep["only_ref"]) <= 1e-5 * rep["S_ref"] + 2, line


def _full_size_bookkeeping(fwd, ref, W, H, what):
    """Prints the real parity figures of a full-size forward and asserts that every differing instance / pixel is a
    borderline decision (tests/parity_tools.py); the figures land in the pytest log (-s) and in the assertion text."""
    f = {k: _np(fwd[k]) for k in ("image", "T", "n", "sorted", "ranges", "radius")}
    rep = parity_tools.forward_parity_report(f, ref, W, H)
    # explain() ASSERTS (it does not only report): every pixel above 1e-4 and every stop-index mismatch must have an alpha
    # within 2e-3 relative of 1/255 or a T(1 - alpha) that close to 1e-4 on its tile's list, re-evaluated in float64
    worst = parity_tools.explain(rep, f, ref, W, H)
    P = W * H
    above = int(round(rep["frac_above"] * P))
    line = (f"[parity {what}] {above} pixels > 1e-4; per-pixel L1: max {rep['max_l1']:.3e}, 99.99th pct {rep['p9999_l1']:.3e}, mean "
            f"{rep['mean_l1']:.3e}, fraction > 1e-4: {rep['frac_above']:.3e}; n mismatches {rep['n_mismatch']} of {W * H}; "
            f"instances {rep['S_gpu']} vs {rep['S_ref']}: {len(rep['only_gpu'])} only in the HIP lists, "
            f"{len(rep['only_ref'])} only in the oracle's; ceil'ed radii differing: {len(rep['radius_diff'])}; worst "
            f"borderline margins: OBB/tile slack {worst['slack_px']:.2e} px, alpha/T {worst['alpha_rel']:.2e} relative")
    print(line)
    # measured on MI355X: config3 1 pixel of 2 073 600 above 1e-4 (max 1.7e-3), 6 stop-index mismatches, identical lists;
    # config2 1 pixel of 640 000 (max 2.8e-4).  r05: absolute caps (conftest.max_pixels_above_tol: 3 at 1080p, 2 below)
    assert rep["mean_l1"] < 1e-6 and rep["p9999_l1"] < 1e-4, line
    assert above <= max_pixels_above_tol(P) and rep["n_mismatch"] <= max_stop_index_mismatches(P), line
    assert len(rep["only_gpu"]) + len(rep["only_ref"]) <= 1e-5 * rep["S_ref"] + 2, line
    return rep


def _check_backward(grads, ref):
    for k, rk in (("xyz", "xyz"), ("rgb", "band0"), ("sh", "sh"), ("opacity", "opacity"), ("scale", "scale"),
                  ("quaternion", "quaternion")):
        assert_grad_close(_np(grads[k]), ref[rk], "grad_" + k)


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fused_matches_oracle(gpu, scene, orc, name):
    r = _run(gpu, scene, name, view_index=2, intermediates=True)
    c = scene.CONFIG
    ref = orc.rasterize(r["params"], r["cam"], c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], r["L"],
                        threads=8)
    _check_forward(r["fwd"], ref)
    np.testing.assert_allclose(_np(r["fwd"]["sigma"]), ref["sigma"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(_np(r["fwd"]["J"]), ref["J"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(_np(r["fwd"]["uv"]), ref["uv"], rtol=1e-6, atol=1e-4)
    assert (_np(r["fwd"]["compact_to_global"]) == np.nonzero(ref["mask"])[0]).all()
    bref = orc.backward_pass(ref, r["cam"], r["gi"], c["bg"], r["L"], threads=8)
    _check_backward(r["grads"], bref)
    for k in ("conic", "uv", "J", "sigma", "xyz_c"):
        assert_grad_close(_np(r["grads"][k]), bref[k], "intermediate grad_" + k)
    assert_grad_close(_np(r["grads"]["precompute_rgb"]), bref["rgb_pre"], "grad_precompute_rgb")


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fused_matches_golden_fixture(gpu, scene, name):
    """REGRESSION check, not a pin: tests/golden/*.npz are outputs of the oracle itself (make_golden.py), so this
    adds nothing to test_fused_matches_oracle beyond catching an accidental change of the oracle or the generator."""
    gold = dict(np.load(os.path.join(GOLD, name + ".npz")))
    r = _run(gpu, scene, name, view_index=int(gold["view_index"]))
    _check_forward(r["fwd"], gold)
    _check_backward(r["grads"], {k[5:]: v for k, v in gold.items() if k.startswith("grad_")})


@pytest.mark.parametrize("l_max", [0, 1, 2])
def test_lower_sh_degrees(gpu, scene, orc, l_max):
    raster = pkg("raster")
    N, W, H = 3000, 200, 120  # H not a multiple of 16: last tile row is partial
    params = scene.make_gaussians(N, W, H, l_max)
    cam = scene.make_camera(W, H, 3)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, 0.0, l_max)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.0, l_max)
    _check_forward(fwd, ref)
    gi = scene.make_grad_image(W, H)
    grads = ctx.alloc_gradients(fwd["num_culled"], l_max)
    ctx.backward_pass(dp, dc, gpu.as_tensor(gi).cuda(), 0.0, l_max, grads)
    bref = orc.backward_pass(ref, cam, gi, 0.0, l_max)
    _check_backward(grads, bref)


@pytest.mark.parametrize("shape", [(1, 17, 9, 0), (3, 33, 31, 1), (37, 100, 7, 2), (255, 16, 16, 3), (257, 130, 66, 3)])
def test_odd_sizes_match_oracle(gpu, scene, orc, shape):
    """Gaussian counts around the workgroup partition sizes and image sizes that are not multiples of the tile."""
    torch = gpu
    raster = pkg("raster")
    N, W, H, L = shape
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    try:
        fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    except pkg("_lib").GsplatError as e:
        assert e.code == -5  # nothing visible from this pose: the oracle must agree
        ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L)
        assert int(ref["mask"].sum()) == 0
        return
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L)
    _check_forward(fwd, ref)
    gi = scene.make_grad_image(W, H)
    grads = ctx.alloc_gradients(fwd["num_culled"], L)
    ctx.backward_pass(dp, dc, torch.as_tensor(gi).cuda(), c["bg"], L, grads)
    _check_backward(grads, orc.backward_pass(ref, cam, gi, c["bg"], L))


def test_instance_buffers_grow(gpu, scene, orc):
    """Large gaussians: many more than 4 instances per gaussian, so the context has to grow its instance buffers in
    the middle of the forward (after the speculative part has run), on either binning route."""
    raster = pkg("raster")
    N, W, H, L = 300, 256, 144, 1
    params = scene.make_gaussians(N, W, H, L)
    params["scale"] += 2.5          # ~12x larger: rectangles of dozens of tiles (also beyond the 64-tile hit mask)
    params["opacity"][:] = -3.0
    cam = scene.make_camera(W, H)
    c = scene.CONFIG
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.5, L, threads=8)
    assert len(ref["sorted"]) > 6 * N
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    for route in (1, 2):  # fresh contexts: growth on the first call of each route
        ctx = raster.RasterContext(N, W, H)
        ctx.set_binning_route(route)
        before = ctx.workspace_bytes
        _check_forward(ctx.rasterize_image(dp, dc, c, 0.5, L), ref)
        assert ctx.workspace_bytes > before
        _check_forward(ctx.rasterize_image(dp, dc, c, 0.5, L), ref)
    # the same on a NON-BLOCKING side stream (torch's streams are): the fill of the fresh instance buffers must be
    # ordered against the placement and the sorts on that stream, not issued on the NULL stream
    torch = gpu
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for route in (1, 2):
        with torch.cuda.stream(side):
            ctx = raster.RasterContext(N, W, H)
            ctx.set_binning_route(route)
            before = ctx.workspace_bytes
            fwd = ctx.rasterize_image(dp, dc, c, 0.5, L)
            side.synchronize()
            assert ctx.workspace_bytes > before
            _check_forward(fwd, ref)
        torch.cuda.synchronize()


def test_culling_and_empty_view(gpu, scene):
    """Gaussians behind the camera are culled; a view that sees nothing is an error code, not an exit
    (cuda/raster.cu:38-41)."""
    raster, lib_mod = pkg("raster"), pkg("_lib")
    N, W, H, L = 500, 64, 48, 1
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::2, 2] *= -1  # every other gaussian behind the camera
    cam = scene.make_camera(W, H)
    ctx = raster.RasterContext(N, W, H)
    fwd = ctx.rasterize_image(raster.device_params(params), raster.device_camera(cam), scene.CONFIG, 0.5, L)
    assert fwd["num_culled"] == N // 2
    assert _np(fwd["mask"])[::2].sum() == 0
    params["xyz"][:, 2] = -np.abs(params["xyz"][:, 2])
    with pytest.raises(lib_mod.GsplatError) as e:
        ctx.rasterize_image(raster.device_params(params), raster.device_camera(cam), scene.CONFIG, 0.5, L)
    assert e.value.code == -5


def test_saturated_pixels_stop_early(gpu, scene, orc):
    """Dense opaque scene: most pixels hit T < 1e-4; n, T and the image must still match (exercises the wave- and
    block-level early exits and the backward's per-pixel n gating)."""
    raster = pkg("raster")
    N, W, H, L = 20000, 96, 80, 0
    params = scene.make_gaussians(N, W, H, L)
    params["opacity"][:] = 6.0
    cam = scene.make_camera(W, H)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, 0.5, L)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.5, L, threads=8)
    assert (ref["T"] < 1e-4).mean() > 0.5
    _check_forward(fwd, ref)
    gi = scene.make_grad_image(W, H)
    grads = ctx.alloc_gradients(fwd["num_culled"], L)
    ctx.backward_pass(dp, dc, gpu.as_tensor(gi).cuda(), 0.5, L, grads)
    _check_backward(grads, orc.backward_pass(ref, cam, gi, 0.5, L, threads=8))


def test_backward_gate_opaque_gaussians_and_zero_gradient_tiles(gpu, scene, orc):
    """cuda/render_backward.cu:170 through the fused path: gaussians with sigmoid(opacity) == 1.0f (logit 20) and tiles
    whose grad_image is exactly zero.  The reference adds NOTHING for such (gaussian, tile) pairs -- not even the colour
    sums -- so a fully opaque gaussian ends with an exactly zero compositing gradient."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L = 4000, 160, 96, 2
    params = scene.make_gaussians(N, W, H, L)
    opaque = np.arange(N) % 7 == 0
    params["opacity"][opaque] = 20.0
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
    _check_forward(fwd, ref)
    gi = scene.make_grad_image(W, H)
    gi[32:64, 48:112] = 0.0  # eight whole tiles
    gi[:, :16] = 0.0         # and the first tile column
    grads = ctx.alloc_gradients(fwd["num_culled"], L, intermediates=True)
    ctx.backward_pass(dp, dc, torch.as_tensor(gi).cuda(), c["bg"], L, grads)
    bref = orc.backward_pass(ref, cam, gi, c["bg"], L, threads=8)
    _check_backward(grads, bref)
    assert_grad_close(_np(grads["precompute_rgb"]), bref["rgb_pre"], "grad_precompute_rgb")
    sel = opaque[np.nonzero(ref["mask"])[0]]
    assert sel.sum() > 100
    for k, rk in (("precompute_rgb", "rgb_pre"), ("conic", "conic"), ("uv", "uv"), ("opacity", "opacity")):
        assert (bref[rk][sel] == 0).all(), rk + " (oracle)"
        assert (_np(grads[k])[sel] == 0).all(), k + ": fully opaque gaussians get no compositing gradient"


@pytest.mark.parametrize("L", [0, 1, 2, 3])
def test_preprocess_split_is_bit_identical(gpu, scene, L):
    """r06: the per-gaussian forward as two kernels (sh_colour_kernel + preprocess_geom_kernel: one behind the other, mode
    1, and side by side on two streams, mode 2) against the single fused preprocess_kernel (mode 0, the default;
    gsplat_context_set_preprocess_split): the same functions on the same
    inputs, so every forward output -- the ForwardPassData arrays, the lists, the image -- must be bit-identical and the
    gradients, placed through rank[] by the pack kernel, equal to rounding, with every array stored and lean, over all indices and over the compacted slots
    (a view that culls half of the scene, interleaved and in whole slices)."""
    torch, raster = gpu, pkg("raster")
    N, W, H = 30000, 320, 192
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][1::2, 2] *= -1.0
    params["xyz"][5000:9000, 2] = -np.abs(params["xyz"][5000:9000, 2])
    cam = scene.make_camera(W, H, 2)
    c = scene.CONFIG
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    keys = ("mask", "compact_to_global", "xyz_c", "uv", "radius", "sorted", "ranges", "image", "T", "n")
    mids = ("sigma", "J", "conic", "rgb")
    for lean in (False, True):
        got = {}
        for split in (0, 1, 2):
            ctx = raster.RasterContext(N, W, H)
            ctx.set_lean_forward(lean)
            ctx.set_preprocess_split(split)
            outs = []
            for walk in range(2):  # the second forward of a context walks the compacted slots
                f = ctx.rasterize_image(dp, dc, c, c["bg"], L)
                grads = ctx.alloc_gradients(f["num_culled"], L)
                ctx.backward_pass(dp, dc, gi, c["bg"], L, grads)
                packed = torch.empty(N, raster.packed_gradient_width(L), device="cuda")
                ctx.pack_gradients_global(grads, L, N, packed)
                o = {k: _np(f[k]).copy() for k in keys + (() if lean else mids)}
                o["packed"] = _np(packed).copy()
                o["counts"] = (f["num_culled"], f["num_splats"], f["num_pairs"])
                outs.append(o)
            assert ctx.counters()["compact_walks"] == 1
            got[split] = outs
            ctx.close()
        for walk, split in ((0, 1), (1, 1), (0, 2), (1, 2)):
            a, b = got[0][walk], got[split][walk]
            assert a["counts"] == b["counts"]
            assert a["counts"][0] < 0.8 * N
            for k in a:
                if k in ("counts", "packed"):
                    continue
                assert np.array_equal(a[k], b[k], equal_nan=True), f"L={L} lean={lean} walk={walk} split={split}: {k} differs"
            # the gradients go through float atomics (the order of a gaussian's tiles differs from run to run): the same
            # rows at the same places -- rank[] is what places them -- to rounding
            assert np.array_equal(a["packed"][:, -1], b["packed"][:, -1]), "visibility column"
            assert_grad_close(b["packed"], a["packed"], f"L={L} lean={lean} walk={walk} split={split}: packed gradients", rel=1e-4)


def test_lean_forward_and_compacted_walk_change_nothing(gpu, scene, orc):
    """r03: (i) gsplat_context_set_lean_forward drops the stores of Sigma / J / conic / colour (the fused backward
    recomputes them): those four views disappear, everything else -- image, lists, radii, gradients -- is what the full
    forward gives.  (ii) A view that culled more than a fifth of the scene is walked through the slice-local kept lists
    on the NEXT forward of the context (preprocess_kernel<.., kCompact>): every output must be bit-identical to the
    walk over all indices, rank[] (read by the pack kernels) included."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L = 30000, 320, 192, 3
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][1::2, 2] *= -1.0  # every other gaussian behind the camera, interleaved
    params["xyz"][5000:9000, 2] = -np.abs(params["xyz"][5000:9000, 2])  # and whole slices of them
    cam = scene.make_camera(W, H, 2)
    c = scene.CONFIG
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
    bref = orc.backward_pass(ref, cam, scene.make_grad_image(W, H), c["bg"], L, threads=8)
    keys = ("mask", "compact_to_global", "xyz_c", "uv", "radius", "sorted", "ranges", "image", "T", "n")
    mids = ("sigma", "J", "conic", "rgb")
    for lean, route in ((False, 1), (True, 1), (False, 2), (True, 2)):  # route 2: the radix sorts (scan of counts[0..N])
        ctx = raster.RasterContext(N, W, H)
        ctx.set_lean_forward(lean)
        ctx.set_binning_route(route)
        first = ctx.rasterize_image(dp, dc, c, c["bg"], L)          # walks all indices (no previous forward)
        assert first["num_culled"] < 0.8 * N
        if not lean:
            _check_forward(first, ref)
        a = {k: _np(first[k]).copy() for k in keys + (() if lean else mids)}
        if not lean and route == 1:
            full = a
        if not lean:
            assert first["uv_all"] is not None and first["xyz_c_all"] is not None
            for k in keys:
                assert (a[k] == full[k]).all(), f"{k}: binning route {route} differs from route 1"
        else:  # the lean forward recomputes positions instead of reading them back: the same bits
            for k in keys:
                assert (a[k] == full[k]).all(), f"{k}: lean forward differs from the full one"
            assert first["uv_all"] is None and first["xyz_c_all"] is None
        packed_a = torch.empty(N, raster.packed_gradient_width(L), device="cuda")
        grads = ctx.alloc_gradients(first["num_culled"], L)
        ctx.backward_pass(dp, dc, gi, c["bg"], L, grads)
        ctx.pack_gradients_global(grads, L, N, packed_a)
        second = ctx.rasterize_image(dp, dc, c, c["bg"], L)         # compacted walk
        for k in a:
            assert (_np(second[k]) == a[k]).all(), f"lean={lean}: {k} differs between the two walks"
        for k in mids:
            assert (second[k] is None) == lean, k
        assert second["num_culled"] == first["num_culled"] and second["num_splats"] == first["num_splats"]
        assert second["num_pairs"] == first["num_pairs"]
        grads2 = ctx.alloc_gradients(second["num_culled"], L)
        ctx.backward_pass(dp, dc, gi, c["bg"], L, grads2)
        _check_backward(grads2, bref)
        packed_b = torch.empty_like(packed_a)
        ctx.pack_gradients_global(grads2, L, N, packed_b)           # rank[] of the compacted walk
        vis_a, vis_b = _np(packed_a[:, -1]), _np(packed_b[:, -1])
        assert (vis_a == vis_b).all() and (vis_a == ref["mask"]).all()
        assert_grad_close(_np(packed_b), _np(packed_a), "packed rows, compacted walk vs walk over all", rel=1e-4)
        third = ctx.rasterize_image(dp, dc, c, c["bg"], L)
        assert (_np(third["image"]) == a["image"]).all()


def test_backward_in_ranges_of_global_indices(gpu, scene):
    """gsplat_backward_gaussians_range over any partition of [0, N) -- uneven, with an empty range and a range whose
    gaussians are all culled -- writes exactly what gsplat_backward_gaussians writes, and the rows of `common` packed
    range by range are the rows packed at once (the chunked exchange of a view-sharded step)."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L = 6000, 200, 120, 3
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::3, 2] *= -1.0
    params["xyz"][2000:3100, 2] = -np.abs(params["xyz"][2000:3100, 2])  # a whole range behind the camera
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    # bounds on visible AND on culled indices (0, 3, 2000, 2500, 4002 are culled), an empty range, an all-culled range
    bounds = [0, 1, 1, 3, 700, 2000, 2500, 3100, 3163, 4002, 5999, N]
    # Twice on the same context: the first forward walks all indices (rank[] global everywhere), the second -- the view
    # culled more than a fifth -- walks the compacted slots, after which rank[] of a CULLED index still holds the cull's
    # slice-local count (r03 advisor finding: the ranges used to be read from rank[]; they now come from
    # compact_to_global).  A lean context too: what the training loop and ViewShardedStep run.
    for lean in (False, True):
        ctx.set_lean_forward(lean)
        for walk in ("all indices", "compacted slots"):
            fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
            M = fwd["num_culled"]
            assert M * 5 < N * 4  # enough culled for the compacted walk from the second forward on
            mask = fwd["mask"].cpu().numpy()
            assert sum(1 for b in bounds[:-1] if not mask[b]) >= 5
            ctx.backward_render(gi, c["bg"])
            whole = ctx.alloc_gradients(N, L, intermediates=True)
            parts = ctx.alloc_gradients(N, L, intermediates=True)
            for g in list(whole.values()) + list(parts.values()):
                g.fill_(float("nan"))
            ctx.backward_gaussians(dp, dc, L, whole)
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                ctx.backward_gaussians_range(dp, dc, L, parts, lo, hi)
            torch.cuda.synchronize()
            for k in whole:
                assert torch.isfinite(whole[k][:M]).all(), (k, walk)
                assert torch.equal(whole[k][:M], parts[k][:M]), \
                    f"grad_{k} (lean={lean}, forward walked {walk}): the ranges do not add up to the whole backward"
            common_a = torch.full((N, 12), float("nan"), device="cuda")
            common_b = torch.full((N, 12), float("nan"), device="cuda")
            raster.pack_gradients_split(ctx, whole, N, common_a, None)
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                raster.pack_gradients_split_range(ctx, parts, N, lo, hi, common_b, None)
            assert torch.equal(common_a, common_b)
            assert (common_a[2000:3100] == 0).all() and (common_a[:, 11].sum().item() == M)
    with pytest.raises(pkg("_lib").GsplatError):
        ctx.backward_gaussians_range(dp, dc, L, parts, 10, N + 1)


def test_repeatable_forward_and_linear_backward(gpu, scene):
    """Idempotence: same inputs -> bit-identical forward.  Linearity: backward(2*g) == 2*backward(g) up to the
    float-atomic summation order."""
    torch = gpu
    r = _run(torch, scene, "small", 0)
    img1, n1, s1 = _np(r["fwd"]["image"]).copy(), _np(r["fwd"]["n"]).copy(), _np(r["fwd"]["sorted"]).copy()
    c = scene.CONFIG
    fwd2 = r["ctx"].rasterize_image(r["dp"], r["dc"], c, c["bg"], r["L"])
    assert (_np(fwd2["image"]) == img1).all() and (_np(fwd2["n"]) == n1).all() and (_np(fwd2["sorted"]) == s1).all()
    g2 = r["ctx"].alloc_gradients(fwd2["num_culled"], r["L"])
    r["ctx"].backward_pass(r["dp"], r["dc"], torch.as_tensor(2 * r["gi"]).cuda(), c["bg"], r["L"], g2)
    for k in ("xyz", "sh", "opacity", "scale", "quaternion"):
        assert_grad_close(_np(g2[k]), 2 * _np(r["grads"][k]), "linearity " + k, rel=1e-4)


def test_full_size_properties_and_parity(gpu, scene, orc, config3_case):
    """BASELINE config 3 (1e6 gaussians, 1920x1080, SH 3): structural properties of the lists and full parity with
    the oracle (the oracle needs ~10 s on the box's host cores at this size; conftest.config3_case holds its results)."""
    torch = gpu
    r = _run(torch, scene, "config3", 0)
    fwd, W, H = r["fwd"], r["W"], r["H"]
    ranges, srt = _np(fwd["ranges"]), _np(fwd["sorted"])
    assert ranges[0] == 0 and ranges[-1] == fwd["num_splats"] and (np.diff(ranges) >= 0).all()
    z = _np(fwd["xyz_c"])[:, 2]
    tile_of = np.repeat(np.arange(len(ranges) - 1), np.diff(ranges))
    zs = z[srt]
    same = tile_of[1:] == tile_of[:-1]
    assert (zs[1:][same] >= zs[:-1][same]).all(), "lists must be depth-sorted inside every tile"
    img = _np(fwd["image"])
    assert np.isfinite(img).all() and (_np(fwd["T"]) >= 0).all() and (_np(fwd["T"]) <= 1).all()
    for k, g in r["grads"].items():
        assert torch.isfinite(g).all(), k
    c = scene.CONFIG
    ref, bref = config3_case["ref"], config3_case["bref"]
    _check_forward(fwd, ref, exact_lists=False)
    _full_size_bookkeeping(fwd, ref, W, H, "config3")
    _check_backward(r["grads"], bref)
    # the lean forward the benchmark times (no Sigma / J / conic / colour stores, positions recomputed): the same bits
    keep = {k: _np(fwd[k]).copy() for k in ("radius", "sorted", "ranges", "image", "T", "n")}
    r["ctx"].set_lean_forward(True)
    lean = r["ctx"].rasterize_image(r["dp"], r["dc"], c, c["bg"], r["L"])
    for k, v in keep.items():
        assert (_np(lean[k]) == v).all(), f"{k}: the lean forward differs at full size"
    g2 = r["ctx"].alloc_gradients(lean["num_culled"], r["L"])
    r["ctx"].backward_pass(r["dp"], r["dc"], torch.as_tensor(r["gi"]).cuda(), c["bg"], r["L"], g2)
    _check_backward(g2, bref)


def test_pack_gradients_global_layout(gpu, scene):
    """Compacted per-view gradients -> global-order packed rows (the buffer the ranks all-reduce)."""
    torch = gpu
    raster, gdist = pkg("raster"), pkg("dist")
    N, W, H, L = 2000, 128, 96, 2
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::3, 2] *= -1  # a third of the gaussians are culled
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    grads = ctx.alloc_gradients(fwd["num_culled"], L)
    ctx.backward_pass(dp, dc, torch.as_tensor(scene.make_grad_image(W, H)).cuda(), c["bg"], L, grads)
    width = raster.packed_gradient_width(L)
    packed = torch.full((N, width), float("nan"), device="cuda")
    ctx.pack_gradients_global(grads, L, N, packed)
    p = _np(packed)
    mask = _np(fwd["mask"]).astype(bool)
    cols, w2 = gdist.packed_layout(L)
    assert w2 == width
    assert (p[~mask] == 0).all()
    assert (p[mask, cols["visible"][0]] == 1).all()
    for k in ("xyz", "rgb", "sh", "opacity", "scale", "quaternion"):
        a, b = cols[k]
        assert (p[mask, a:b] == _np(grads[k]).reshape(mask.sum(), b - a)).all(), k
    un = gdist.unpack(packed, L)
    assert un["sh"].shape == (N, (L + 1) ** 2 - 1, 3)


def test_dense_scene_takes_the_global_depth_presort(gpu, scene, orc):
    """More than 768 list entries per tile on average and lists beyond the LDS merge's 4096 entries.  The binning route
    follows the previous forward: the first call on a context bins with the counting sort + per-tile kernels (lists >
    4096 entries: in-place global network), the second takes the dense route (global stable depth pre-sort + stable tile
    sort).  Both must match the oracle exactly (lists) / within tolerance (image)."""
    raster = pkg("raster")
    N, W, H, L = 40000, 64, 48, 0
    params = scene.make_gaussians(N, W, H, L)
    params["opacity"][:] = -4.0  # faint: nothing saturates, every list entry matters
    cam = scene.make_camera(W, H)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.5, L, threads=8)
    assert np.diff(ref["ranges"]).max() > 4096 and len(ref["sorted"]) > 768 * (len(ref["ranges"]) - 1)
    for route in (0, 0, 1, 2):  # automatic (counting sort first, then radix), then each forced
        ctx.set_binning_route(route)
        fwd = ctx.rasterize_image(dp, dc, c, 0.5, L)
        _check_forward(fwd, ref)


def test_few_long_tile_lists_in_a_sparse_scene(gpu, scene, orc):
    """Low average list length (per-tile sort kernels) but three hot spots: one tile list above 4096 entries (in-place
    global-memory network), one between 2049 and 4096 and one between 1025 and 2048 (register-sorted runs merged in
    LDS by four resp. two waves); the rest sorts in registers."""
    raster = pkg("raster")
    N, W, H, L = 14500, 256, 144, 0
    params = scene.make_gaussians(N, W, H, L)
    params["opacity"][:] = -4.0
    cam = scene.make_camera(W, H)
    rng = np.random.default_rng(4)
    for lo, hi, (cu, cv) in ((5000, 9600, (40.0, 40.0)), (9600, 11100, (200.0, 100.0)), (11100, 14500, (120.0, 60.0))):
        k = hi - lo
        z = rng.uniform(3.0, 9.0, k)
        u, v = cu + rng.uniform(-2, 2, k), cv + rng.uniform(-2, 2, k)
        params["xyz"][lo:hi, 0] = (u - W / 2) * z / cam["fx"]
        params["xyz"][lo:hi, 1] = (v - H / 2) * z / cam["fy"]
        params["xyz"][lo:hi, 2] = z
        params["scale"][lo:hi] = np.log(0.004)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    fwd = ctx.rasterize_image(raster.device_params(params), raster.device_camera(cam), c, 0.5, L)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.5, L, threads=8)
    lens = np.diff(ref["ranges"])
    assert lens.max() > 4096 and ((lens > 1024) & (lens <= 2048)).any() and ((lens > 2048) & (lens <= 4096)).any()
    assert lens.mean() < 768
    _check_forward(fwd, ref)
    # r06: a context whose previous forward had no list beyond the wave kernel's 1024 entries does not queue the hand-over
    # sort kernel at all; a forward whose longest list then lands just above (1025..2048: the hand-over class alone) must
    # notice from its count record and redo its tail -- the lists exact as ever
    plain = scene.make_gaussians(N, W, H, L)
    plain["opacity"][:] = -4.0
    mild = {k: np.array(v, copy=True) for k, v in plain.items()}
    lo, hi, (cu, cv) = 9600, 11100, (200.0, 100.0)
    mild["xyz"][lo:hi], mild["scale"][lo:hi] = params["xyz"][lo:hi], params["scale"][lo:hi]
    ref2 = orc.rasterize(mild, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.5, L, threads=8)
    lens2 = np.diff(ref2["ranges"])
    assert 1024 < lens2.max() <= 2048
    other = raster.RasterContext(N, W, H)
    short = other.rasterize_image(raster.device_params(plain), raster.device_camera(cam), c, 0.5, L)
    assert np.diff(_np(short["ranges"])).max() * 3 // 2 + 64 <= 1024  # (the tail is queued for 1.5x the last longest + 64)
    before = other.counters()["tail_redone"]
    _check_forward(other.rasterize_image(raster.device_params(mild), raster.device_camera(cam), c, 0.5, L), ref2)
    assert other.counters()["tail_redone"] == before + 1
    _check_forward(other.rasterize_image(raster.device_params(mild), raster.device_camera(cam), c, 0.5, L), ref2)
    assert other.counters()["tail_redone"] == before + 1  # (the hint now covers the list: nothing is redone)


def test_very_long_tile_lists(gpu, scene, orc):
    """Tile lists of 5 000, 9 000 and 17 500 entries in an otherwise sparse scene: eight and sixteen register-sorted
    runs merged in 64 / 128 KB of LDS, and beyond 16 384 entries the in-place global-memory network."""
    raster = pkg("raster")
    N, W, H, L = 36000, 256, 144, 0
    params = scene.make_gaussians(N, W, H, L)
    params["opacity"][:] = -4.0
    cam = scene.make_camera(W, H)
    rng = np.random.default_rng(7)
    for lo, hi, (cu, cv) in ((4000, 9000, (40.0, 40.0)), (9000, 18000, (200.0, 100.0)), (18000, 35500, (120.0, 60.0))):
        k = hi - lo
        z = rng.uniform(3.0, 9.0, k)
        u, v = cu + rng.uniform(-2, 2, k), cv + rng.uniform(-2, 2, k)
        params["xyz"][lo:hi, 0] = (u - W / 2) * z / cam["fx"]
        params["xyz"][lo:hi, 1] = (v - H / 2) * z / cam["fy"]
        params["xyz"][lo:hi, 2] = z
        params["scale"][lo:hi] = np.log(0.004)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    ctx.set_binning_route(1)  # counting sort + per-tile depth sorts, whatever the density
    # a forward with short lists first: the next one queues its sort kernels by THIS forward's longest list, finds its
    # own lists far longer when the counts arrive, and has to redo placement, sorts and compositing
    plain = scene.make_gaussians(N, W, H, L)
    short = ctx.rasterize_image(raster.device_params(plain), raster.device_camera(cam), c, 0.5, L)
    assert int((short["ranges"][1:] - short["ranges"][:-1]).max()) < 1024
    fwd = ctx.rasterize_image(raster.device_params(params), raster.device_camera(cam), c, 0.5, L)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], 0.5, L, threads=8)
    lens = np.diff(ref["ranges"])
    assert lens.max() > 16384 and ((lens > 8192) & (lens <= 16384)).any() and ((lens > 4096) & (lens <= 8192)).any()
    _check_forward(fwd, ref)
    _check_forward(ctx.rasterize_image(raster.device_params(params), raster.device_camera(cam), c, 0.5, L), ref)


def test_long_lists_split_into_segments_for_the_backward(gpu, scene, orc):
    """r05 (gs_render.h: TileSegments): once a forward has seen a list beyond 1488 entries, the next forward stores a
    per-pixel checkpoint {T, colour so far} at every 496th entry of such lists and the backward walks each segment some
    pixel reaches with a workgroup of its own.  Lists of 2 100 .. 7 000 entries whose pixels stop in front of, inside
    and behind the segment boundaries (opacities from faint to opaque); the gradients must be the oracle's, and the
    first (unsplit) backward's.  The forward of such a context runs every segment of those lists as a workgroup of its
    own (gs_render.h: FwdSegments): image, transmittance and stop indices must be the oracle's as before."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L = 24000, 160, 96, 1
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H)
    rng = np.random.default_rng(11)
    for lo, hi, (cu, cv), spread in ((2000, 4200, (24.0, 24.0), 5.0), (4200, 9200, (88.0, 40.0), 7.0),
                                     (9200, 16200, (136.0, 72.0), 4.0), (16200, 18200, (40.0, 72.0), 3.0)):
        k = hi - lo
        z = rng.uniform(3.0, 9.0, k)
        u, v = cu + rng.uniform(-spread, spread, k), cv + rng.uniform(-spread, spread, k)
        params["xyz"][lo:hi, 0] = (u - W / 2) * z / cam["fx"]
        params["xyz"][lo:hi, 1] = (v - H / 2) * z / cam["fy"]
        params["xyz"][lo:hi, 2] = z
        params["scale"][lo:hi] = np.log(rng.uniform(0.004, 0.012, (k, 3)))
        params["opacity"][lo:hi] = rng.choice([-5.0, -4.0, -3.0, -1.0, 3.0], size=k, p=[0.45, 0.3, 0.15, 0.08, 0.02])
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    ctx.set_binning_route(1)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    gi = scene.make_grad_image(W, H)
    gi_d = torch.as_tensor(gi).cuda()
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
    lens = np.diff(ref["ranges"])
    assert lens.max() > 10 * 496 and ((lens > 1488) & (lens < 5 * 496)).any() and ((lens > 496) & (lens <= 1488)).any()
    stops = np.asarray(ref["n"]).reshape(-1)
    assert (stops > 3 * 496).any() and ((stops > 496) & (stops < 992)).any(), "pixels must stop behind several boundaries"
    bref = orc.backward_pass(ref, cam, gi, c["bg"], L, threads=8)
    whole = None
    for it in range(5):
        fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
        _check_forward(fwd, ref)
        grads = ctx.alloc_gradients(fwd["num_culled"], L, intermediates=True)
        ctx.backward_pass(dp, dc, gi_d, c["bg"], L, grads)
        assert ctx.counters()["segmented_backwards"] == it, "the split follows the forward before"
        # The forward's own split follows the figures of the forward TWO back (r06): forward 1 is the first to publish them
        # (forward 0 told it that there are long lists), forward 3 the first that may take them -- the host knows forward
        # 1's kernels complete once it has seen forward 2's record.  Whatever the host / GPU timing: no synchronisation
        # anywhere in this loop, and the count is exact.
        assert ctx.counters()["segmented_forwards"] == max(0, it - 2)
        image = _np(fwd["image"]).copy()
        if it == 3:
            segmented_image = image
            assert ctx.counters()["longest_chain"] == int(stops.max())
        elif it == 4:  # the segments' colours are added in a fixed order: the same bits in every run
            assert (image == segmented_image).all() and (_np(fwd["n"]) == stops_gpu).all()
        stops_gpu = _np(fwd["n"]).copy()
        _check_backward(grads, bref)
        assert_grad_close(_np(grads["precompute_rgb"]), bref["rgb_pre"], "grad_precompute_rgb")
        got = {k: _np(grads[k]).copy() for k in ("precompute_rgb", "conic", "uv", "opacity")}
        if whole is None:
            whole = got
        else:
            for k in whole:  # the same sums in another order of the atomics, plus the checkpoint's rounding
                assert_grad_close(got[k], whole[k], k + " (segments vs whole lists)")
    # However a segment's workgroup obtains the transmittance in front of it -- handed over by the workgroup in front (no
    # layer runs side by side), folded from the products the workgroups in front published (every layer does), or
    # multiplied up by itself because its polls ran out (a budget of one poll) -- the value is the same left fold, so the
    # forward's outputs must be the SAME BITS, and the render-only context's too.
    T_gpu = _np(fwd["T"]).copy()
    for what, opts in (("every workgroup waits for the one in front", dict(thin_layer_blocks=0)),
                       ("all layers side by side", dict(thin_layer_blocks=1 << 20)),
                       ("one poll, then the product from entry 0", dict(poll_budget=1, thin_layer_blocks=1 << 20)),
                       ("one poll, nobody publishes products", dict(poll_budget=1, thin_layer_blocks=0)),
                       ("render only", dict(render_only=True))):
        other = raster.RasterContext(N, W, H)
        other.set_binning_route(1)
        if opts.pop("render_only", False):
            other.set_render_only(True)
        other.set_segment_options(**opts)
        for it in range(4):  # (no synchronisation: the split decision does not depend on what has landed, see above)
            out = other.rasterize_image(dp, dc, c, c["bg"], L)
        assert other.counters()["segmented_forwards"] == 1, what
        assert (_np(out["image"]) == segmented_image).all(), what
        assert (_np(out["n"]) == stops_gpu).all() and (_np(out["T"]) == T_gpu).all(), what


def test_factored_exchange_equals_full_rows(gpu, scene):
    """Simulates a 3-rank view-sharded step on one GPU: per-view factored rows, summed like the all-reduce would,
    then unpacked, must equal the sum of the full packed rows of the three views."""
    torch = gpu
    raster = pkg("raster")
    N, W, H, L, world = 3000, 160, 96, 3, 3
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::7, 2] *= -1
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp = raster.device_params(params)
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    fw, pw = raster.factored_gradient_width(world), raster.packed_gradient_width(L)
    fac_sum = torch.zeros(N + 1, fw, device="cuda")
    full_sum = torch.zeros(N, pw, device="cuda")
    common_sum = torch.zeros(N, 12, device="cuda")           # what the split exchange's all-reduce would hold
    rgb_all = torch.zeros(world, N + 1, 3, device="cuda")    # ... and its all-gather
    for r in range(world):
        cam = scene.make_camera(W, H, view_index=r + 1)
        dc = raster.device_camera(cam)
        fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
        grads = ctx.alloc_gradients(fwd["num_culled"], L)
        grads["precompute_rgb"] = torch.empty(fwd["num_culled"], 3, device="cuda")
        ctx.backward_pass(dp, dc, gi, c["bg"], L, grads)
        fac = torch.zeros(N + 1, fw, device="cuda")
        raster.pack_gradients_factored(ctx, grads, N, r, world, fac)
        fac[N, 12 + 3 * r: 15 + 3 * r] = torch.as_tensor(cam["campos"], device="cuda")
        fac_sum += fac
        common = torch.full((N, 12), float("nan"), device="cuda")
        raster.pack_gradients_split(ctx, grads, N, common, rgb_all[r])
        rgb_all[r, N] = torch.as_tensor(cam["campos"], device="cuda")
        common_sum += common
        full = torch.empty(N, pw, device="cuda")
        ctx.pack_gradients_global(grads, L, N, full)
        full_sum += full
    out = torch.full((N, pw), float("nan"), device="cuda")
    raster.unpack_gradients_factored(dp["xyz"], fac_sum[N, 12:].contiguous(), fac_sum, L, N, world, out)
    a, b = _np(out), _np(full_sum)
    assert np.isfinite(a).all()
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9)
    assert (a[:, -1] == b[:, -1]).all()  # visibility counts
    out2 = torch.full((N, pw), float("nan"), device="cuda")
    raster.unpack_gradients_split(dp["xyz"], common_sum, rgb_all, 3 * (N + 1), L, N, world, out2)
    assert torch.equal(out2, out), "split (all-reduce + all-gather) and factored (one all-reduce) must agree exactly"


def test_split_backward_equals_backward_pass(gpu, scene):
    """gsplat_backward_render + gsplat_backward_gaussians == gsplat_backward_pass, and the g_rgb it leaves in global
    order is exactly the scatter of the per-gaussian kernel's grad_precompute_rgb (what the exchange overlaps)."""
    torch = gpu
    raster = pkg("raster")
    N, W, H, L, _ = scene.WORKLOADS["small"]
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::4, 2] *= -1
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(scene.make_camera(W, H, 2))
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    M = fwd["num_culled"]
    g1 = ctx.alloc_gradients(M, L, intermediates=True)
    ctx.backward_pass(dp, dc, gi, c["bg"], L, g1)
    g2 = ctx.alloc_gradients(M, L, intermediates=True)
    rgb_global = torch.full((N + 1, 3), float("nan"), device="cuda")
    with pytest.raises(pkg("_lib").GsplatError):  # the second half before the first
        ctx.rasterize_image(dp, dc, c, c["bg"], L)
        ctx.backward_gaussians(dp, dc, L, g2)
    ctx.backward_render(gi, c["bg"], rgb_global)
    ctx.backward_gaussians(dp, dc, L, g2)
    torch.cuda.synchronize()
    for k in g1:
        assert_grad_close(_np(g2[k]), _np(g1[k]), "split " + k, rel=1e-5)
    want = torch.zeros(N, 3, device="cuda")
    want[fwd["compact_to_global"].long()] = g2["precompute_rgb"]
    assert torch.equal(rgb_global[:N], want)
    assert torch.isnan(rgb_global[N]).all()  # row N (camera position) is the caller's


def test_config2_forward_only(gpu, scene, orc):
    """BASELINE configs[1]: synthetic 100k gaussians, 800x800, SH degree 0, forward render only."""
    raster = pkg("raster")
    N, W, H, L, _ = scene.WORKLOADS["config2"]
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    fwd = ctx.rasterize_image(raster.device_params(params), raster.device_camera(cam), c, c["bg"], L)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=16)
    _check_forward(fwd, ref, exact_lists=False)
    _full_size_bookkeeping(fwd, ref, W, H, "config2")


def test_interleaved_culling_at_scale(gpu, scene, orc):
    """200k gaussians of which about half are culled, interleaved at random (scene.cull_half): the per-gaussian kernels
    take their non-consecutive-row paths inside almost every wave; forward and backward against the oracle."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L = 200_000, 960, 540, 3
    params = scene.cull_half(scene.make_gaussians(N, W, H, L))
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    assert 0.4 * N < fwd["num_culled"] < 0.6 * N
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=16)
    _check_forward(fwd, ref, exact_lists=False)
    _full_size_bookkeeping(fwd, ref, W, H, "half-culled 200k")
    assert (_np(fwd["compact_to_global"]) == np.nonzero(ref["mask"])[0]).all()
    gi = scene.make_grad_image(W, H)
    grads = ctx.alloc_gradients(fwd["num_culled"], L, intermediates=True)
    for g in grads.values():
        g.fill_(float("nan"))
    ctx.backward_pass(dp, dc, torch.as_tensor(gi).cuda(), c["bg"], L, grads)
    bref = orc.backward_pass(ref, cam, gi, c["bg"], L, threads=16)
    _check_backward(grads, bref)
    for k in ("conic", "uv", "J", "sigma", "xyz_c"):
        assert_grad_close(_np(grads[k]), bref[k], "intermediate grad_" + k)
    # the same view again: this forward walks the compacted slots (the previous one culled more than a fifth), lean too
    keep = {k: _np(fwd[k]).copy() for k in ("compact_to_global", "radius", "sorted", "ranges", "image", "T", "n")}
    ctx.set_lean_forward(True)
    again = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    for k, v in keep.items():
        assert (_np(again[k]) == v).all(), f"{k}: compacted, lean walk differs from the walk over all indices"
    assert again["sigma"] is None and again["num_splats"] == fwd["num_splats"]
    g2 = ctx.alloc_gradients(again["num_culled"], L)
    ctx.backward_pass(dp, dc, torch.as_tensor(gi).cuda(), c["bg"], L, g2)
    _check_backward(g2, bref)


@pytest.mark.parametrize("splat_scale", [6.0, 25.0])
def test_large_splats_match_oracle(gpu, scene, orc, splat_scale):
    """Splats over tens to thousands of tiles (a capture early in training): their tile tests and placements are shared
    by the lanes of a wave (preprocess_kernel / bin_scatter_kernel), rectangles above 64 tiles carry no hit mask.  Lists
    bit-exact against the oracle, image and gradients within tolerance."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L = 3000, 640, 360, 1
    params = scene.make_gaussians(N, W, H, L, splat_scale=splat_scale)
    params["opacity"][:] -= 3.0  # faint: long lists stay relevant, nothing saturates early
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=16)
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    big = (ref["radius"][:, 0] > 64).sum()
    assert fwd["num_pairs"] > 64 * 0.2 * N and big > 0.05 * N, (fwd["num_pairs"], big)
    _check_forward(fwd, ref)
    gi = scene.make_grad_image(W, H)
    grads = ctx.alloc_gradients(fwd["num_culled"], L)
    ctx.backward_pass(dp, dc, torch.as_tensor(gi).cuda(), c["bg"], L, grads)
    _check_backward(grads, orc.backward_pass(ref, cam, gi, c["bg"], L, threads=16))


def test_render_only_context(gpu, scene, orc):
    """gsplat_context_set_render_only: same image, counts and lists; the backward-only outputs come back null and the
    backward is refused until the mode is switched off."""
    torch, raster = gpu, pkg("raster")
    r = _run(torch, scene, "small", 0)
    ctx, dp, dc, c, L = r["ctx"], r["dp"], r["dc"], scene.CONFIG, r["L"]
    ref_img, ref_n = r["fwd"]["image"].clone(), r["fwd"]["n"].clone()
    ref_sorted = r["fwd"]["sorted"].clone()
    ctx.set_render_only(True)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    assert fwd["sigma"] is None and fwd["J"] is None and fwd["conic"] is None and fwd["rgb"] is None
    assert torch.equal(fwd["image"], ref_img) and torch.equal(fwd["n"], ref_n) and torch.equal(fwd["sorted"], ref_sorted)
    W, H = scene.WORKLOADS["small"][1:3]
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    grads = ctx.alloc_gradients(fwd["num_culled"], L)
    with pytest.raises(Exception):
        ctx.backward_pass(dp, dc, gi, c["bg"], L, grads)
    ctx.set_render_only(False)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    assert fwd["sigma"] is not None
    ctx.backward_pass(dp, dc, gi, c["bg"], L, grads)
    torch.cuda.synchronize()


def test_tile_order_changes_nothing(gpu, scene, orc):
    """r04: the compositing backward takes its tiles heaviest first (tile_order_kernel, from the per-tile stop indices
    render_fwd leaves).  The order only changes WHEN a tile is processed: the gradients must meet the oracle on a scene
    whose tiles differ wildly in work -- the benchmark's gaussians plus a dense cluster over a few tiles (longest list
    over 4x the average) -- and equal the plain order's (GSPLAT_NO_TILE_ORDER=1 is read once per process, so the plain
    order runs through the stand-alone operator, which never orders) up to the float atomics' summation order."""
    torch, raster, ops = gpu, pkg("raster"), pkg("ops")
    N, W, H, L = 40000, 480, 272, 3
    params = scene.make_gaussians(N, W, H, L)
    rng = np.random.default_rng(7)
    k = 6000  # a cluster in front of the camera, projected into ~4 tiles around the image centre
    params["xyz"][:k, 0] = rng.normal(0.0, 0.05, k)
    params["xyz"][:k, 1] = rng.normal(0.0, 0.05, k)
    params["xyz"][:k, 2] = rng.uniform(4.0, 9.0, k)
    cam = scene.make_camera(W, H, 0)
    c = scene.CONFIG
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
    bref = orc.backward_pass(ref, cam, scene.make_grad_image(W, H), c["bg"], L, threads=8)
    lens = np.diff(ref["ranges"])
    assert lens.max() > 4 * lens.mean(), "the scene must be skewed"
    ctx = raster.RasterContext(N, W, H)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)      # the first forward of a context knows no figures: plain order
    _check_forward(fwd, ref)
    M = fwd["num_culled"]
    g0 = ctx.alloc_gradients(M, L, intermediates=True)
    ctx.backward_pass(dp, dc, gi, c["bg"], L, g0)
    assert ctx.counters()["ordered_backwards"] == 0
    _check_backward(g0, bref)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)      # skewed by the first forward's figures: the order is made
    g = ctx.alloc_gradients(M, L, intermediates=True)
    ctx.backward_pass(dp, dc, gi, c["bg"], L, g)          # tiles in the order of their stop indices
    assert ctx.counters()["ordered_backwards"] == 1, "the second step of a skewed scene orders its backward"
    _check_backward(g, bref)
    # the same compositing backward in the plain tile order: the stand-alone operator on the forward's arrays
    z = lambda *s_: torch.zeros(*s_, device="cuda")
    p_rgb, p_op, p_uv, p_conic = z(M, 3), z(M), z(M, 2), z(M, 3)
    ops.render_image_backward(fwd["uv"], dp["opacity"][fwd["compact_to_global"].long()].contiguous(), fwd["conic"], fwd["rgb"],
                              c["bg"], fwd["sorted"], fwd["ranges"], fwd["n"], fwd["T"], gi, W, H, p_rgb, p_op, p_uv, p_conic)
    assert_grad_close(_np(g["precompute_rgb"]), _np(p_rgb), "ordered vs plain: grad_rgb", rel=1e-4)
    assert_grad_close(_np(g["opacity"]), _np(p_op), "ordered vs plain: grad_opacity", rel=1e-4)
    assert_grad_close(_np(g["uv"]), _np(p_uv), "ordered vs plain: grad_uv", rel=1e-4)
    assert_grad_close(_np(g["conic"]), _np(p_conic), "ordered vs plain: grad_conic", rel=1e-4)



def test_forward_stops_where_the_pixels_saturate(gpu, scene):
    """A wave of render_fwd leaves its loop when all 64 of its pixels are saturated, a workgroup when its four waves have
    (cuda/render.cu:76-90: "done").  The results do not depend on it -- saturated pixels blend with weight 0 -- so only the
    clock can tell when the exit is lost: r04 lost it for trips whose FIRST splat saturates the wave's last live pixel, and
    the forward of the scenes whose pixels saturate (large splats, dense captures) took twice as long while the benchmark
    scene, where pixels rarely saturate, showed nothing.  The `bigsplats` workload (a capture early in training: splats
    over many tiles, every pixel saturated long before its list ends): 0.062 ms with the exit, 0.131 ms without it
    (profiles/r04_ab_forward_early_exit.txt); the bar sits between the two."""
    torch, raster = gpu, pkg("raster")
    N, W, H, L, _ = scene.WORKLOADS["bigsplats"]
    c = scene.CONFIG
    dp = raster.device_params(scene.make_workload_gaussians("bigsplats"))
    dc = raster.device_camera(scene.make_camera(W, H, 0))
    ctx = raster.RasterContext(N, W, H)
    for _ in range(3):
        ctx.rasterize_image(dp, dc, c, c["bg"], L)
    ctx.set_timing(True, stages=["render_forward"])
    for _ in range(20):
        ctx.rasterize_image(dp, dc, c, c["bg"], L)
    torch.cuda.synchronize()
    ms = ctx.get_timing()["render_forward"][0]
    ctx.close()
    print(f"bigsplats: render_fwd {ms:.4f} ms")
    perf_check(ms < 0.1, f"render_fwd takes {ms:.4f} ms on the saturating workload (0.062 with the exit, 0.131 without)")
