"""GPU parity of every stand-alone operator (through the C ABI) against the CPU oracle.

Inputs come from the deterministic scene generator; the oracle (test infrastructure) is the
checker.  Per-gaussian operators share their arithmetic order with the oracle, so they must
agree to a few ulps; the compositing operators are compared with the north-star tolerances
(1e-4 per-pixel L1, 1e-3 relative on gradients).
"""
import numpy as np
import pytest

from conftest import assert_grad_close, assert_image_close, assert_stop_indices_close, pkg

pytestmark = pytest.mark.gpu

CFG = dict(near_thresh=0.3, mh_dist=3.0, cull_mask_padding=100)


def _dev(torch, a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def case(scene, orc):
    N, W, H, L, _ = scene.WORKLOADS["small"]
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H, view_index=1)  # non-trivial pose
    fwd = orc.rasterize(params, cam, 0.3, 3.0, 100, 0.5, L, threads=8)
    gi = scene.make_grad_image(W, H)
    bwd = orc.backward_pass(fwd, cam, gi, 0.5, L, threads=8)
    return dict(N=N, W=W, H=H, L=L, params=params, cam=cam, fwd=fwd, gi=gi, bwd=bwd)


def test_projection_and_cull(gpu, case):
    torch, ops = gpu, pkg("ops")
    N, W, H, f, cam = case["N"], case["W"], case["H"], case["fwd"], case["cam"]
    xyz = _dev(torch, case["params"]["xyz"])
    view, proj = _dev(torch, cam["view"]), _dev(torch, cam["proj"])
    xyz_c = torch.empty(N, 3, device="cuda")
    uv = torch.empty(N, 2, device="cuda")
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda")
    ops.compute_camera_space_points(xyz, view, N, xyz_c)
    ops.project_to_screen(xyz_c, proj, N, W, H, uv)
    ops.cull_gaussians(uv, xyz_c, N, 0.3, 100, W, H, mask)
    np.testing.assert_allclose(xyz_c.cpu().numpy(), f["xyz_c_all"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(uv.cpu().numpy(), f["uv_all"], rtol=1e-6, atol=1e-4)
    assert (mask.cpu().numpy().astype(bool) == f["mask"]).all()


def test_known_answer_cull(gpu):  # reference tests/cuda_forward_test.cpp:159-230
    torch, ops = gpu, pkg("ops")
    xyz = _dev(torch, np.array([[0, 0, 5], [0, 0, .5], [0, 0, 12], [0, 0, 5], [0, 0, 5], [0, 0, 12], [0, 0, .5]], np.float32))
    uv = _dev(torch, np.array([[960, 540], [960, 540], [960, 540], [-5, 540], [1925, 540], [-11, 540], [960, 1091]], np.float32))
    mask = torch.zeros(7, dtype=torch.uint8, device="cuda")
    ops.cull_gaussians(uv, xyz, 7, 1.0, 10, 1920, 1080, mask)
    assert mask.cpu().tolist() == [1, 0, 1, 1, 1, 0, 0]


def test_sigma_conic_radius(gpu, case):
    torch, ops = gpu, pkg("ops")
    f, cam = case["fwd"], case["cam"]
    M = f["num_culled"]
    q, s = _dev(torch, f["quaternion"]), _dev(torch, f["scale"])
    sigma = torch.empty(M, 6, device="cuda")
    ops.compute_sigma(q, s, M, sigma)
    np.testing.assert_allclose(sigma.cpu().numpy(), f["sigma"], rtol=2e-5, atol=1e-9)
    J, conic, radius = torch.empty(M, 6, device="cuda"), torch.empty(M, 3, device="cuda"), torch.empty(M, 4, device="cuda")
    # feed the oracle's sigma so that only compute_conic is under test
    ops.compute_conic(_dev(torch, f["xyz_c"]), _dev(torch, cam["view"]), _dev(torch, f["sigma"]), cam["fx"], cam["fy"],
                      f["tan_fovx"], f["tan_fovy"], 3.0, M, J, conic, radius)
    np.testing.assert_allclose(J.cpu().numpy(), f["J"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(conic.cpu().numpy(), f["conic"], rtol=1e-5, atol=1e-7)
    r = radius.cpu().numpy()
    assert (r[:, :2] == f["radius"][:, :2]).mean() > 0.999  # ceil() of values that agree to ~1 ulp
    np.testing.assert_allclose(r[:, 2:], f["radius"][:, 2:], atol=2e-6)


def test_known_answer_conic(gpu):  # reference tests/cuda_forward_test.cpp:306-414
    torch, ops = gpu, pkg("ops")
    J, conic, radius = torch.empty(1, 6, device="cuda"), torch.empty(1, 3, device="cuda"), torch.empty(1, 4, device="cuda")
    ops.compute_conic(_dev(torch, np.array([1, 2, 5], np.float32)), _dev(torch, np.eye(4, dtype=np.float32).ravel()),
                      _dev(torch, np.array([1, 0, 0, 1, 0, 1], np.float32)), 1.0, 1.0, 1.0, 1.0, 3.0, 1, J, conic, radius)
    np.testing.assert_allclose(radius.cpu().numpy()[0], [3, 1, np.sqrt(.8), np.sqrt(.2)], atol=1e-5)
    c00, c01, c11 = .04 + .0016 + .3, .04 * .08, .04 + .0064 + .3
    det = c00 * c11 - c01 * c01
    np.testing.assert_allclose(conic.cpu().numpy()[0], [c11 / det, -c01 / det, c00 / det], atol=1e-5)


@pytest.mark.parametrize("l_max", [0, 1, 2, 3])
def test_spherical_harmonics_forward_backward(gpu, orc, scene, l_max):
    torch, ops = gpu, pkg("ops")
    M = 3000
    p = scene.make_gaussians(M, 256, 144, l_max)
    campos = np.array([0.3, -0.2, 0.1], np.float32)
    rgb_ref = orc.precompute_spherical_harmonics(p["xyz"], p["sh"], p["rgb"], campos, l_max)
    xyz, sh, band0 = _dev(torch, p["xyz"]), _dev(torch, p["sh"]), _dev(torch, p["rgb"])
    rgb = torch.empty(M, 3, device="cuda")
    ops.precompute_spherical_harmonics(xyz, sh if l_max else None, band0, campos, l_max, M, rgb)
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb_ref, rtol=1e-5, atol=2e-6)
    g = scene.normal(77, 5, 3 * M).reshape(M, 3).astype(np.float32)
    pre = np.full((M, 3), 0.25, np.float32)
    shg_ref, b0g_ref, xg_ref = orc.precompute_spherical_harmonics_backward(p["xyz"], p["rgb"], p["sh"], campos, g, l_max,
                                                                           xyz_grad=pre)
    n_rest = (l_max + 1) ** 2 - 1
    shg = torch.empty(M, max(n_rest, 1), 3, device="cuda")
    b0g = torch.empty(M, 3, device="cuda")
    xg = _dev(torch, pre)  # "+=" output
    ops.precompute_spherical_harmonics_backward(xyz, band0, sh if l_max else None, campos, _dev(torch, g), l_max, M,
                                                shg if l_max else None, b0g, xg)
    np.testing.assert_allclose(b0g.cpu().numpy(), b0g_ref, rtol=1e-6, atol=1e-8)
    if l_max:
        np.testing.assert_allclose(shg.cpu().numpy()[:, :n_rest], shg_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xg.cpu().numpy(), xg_ref, rtol=1e-4, atol=1e-5)


def test_binning_is_bit_exact(gpu, case):
    """Same uv / depth / radius in -> identical lists out (integer work: bit-exact bar)."""
    torch, ops = gpu, pkg("ops")
    f, W, H = case["fwd"], case["W"], case["H"]
    M = f["num_culled"]
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    uv, xyz_c, radius = _dev(torch, f["uv"]), _dev(torch, f["xyz_c"]), _dev(torch, f["radius"])
    count = ops.get_sorted_gaussian_list(uv, xyz_c, radius, ntx, nty, M, 0, None, None)
    assert count == f["num_pairs"]
    srt = torch.full((count,), -1, dtype=torch.int32, device="cuda")
    ranges = torch.full((ntx * nty + 1,), -1, dtype=torch.int32, device="cuda")
    ops.get_sorted_gaussian_list(uv, xyz_c, radius, ntx, nty, M, count, srt, ranges)
    S = len(f["sorted"])
    assert (ranges.cpu().numpy() == f["ranges"]).all()
    assert (srt.cpu().numpy()[:S] == f["sorted"]).all()


def test_binning_with_depths_outside_the_register_sorts_range(gpu, orc, case):
    """The per-tile register sort holds its keys as positive normal doubles; a list with a key that has no such form
    (negative, zero, denormal-range or infinite depth) is handed to the integer workgroup kernel.  The operator takes
    whatever depths the caller passes, so both paths must give the reference's order."""
    torch, ops = gpu, pkg("ops")
    f, W, H = case["fwd"], case["W"], case["H"]
    M = f["num_culled"]
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    xyz = np.array(f["xyz_c"], np.float32).reshape(-1, 3).copy()
    rng = np.random.default_rng(5)
    odd = np.array([-3.5, 0.0, 1e-41, np.inf, -np.inf, 3e38, 1e-30], np.float32)  # not -0.0: the oracle ties it with +0.0
    pick = rng.random(M) < 0.3
    xyz[pick, 2] = odd[rng.integers(0, len(odd), pick.sum())]
    xyz[~pick, 2] *= np.where(rng.random((~pick).sum()) < 0.5, -1.0, 1.0).astype(np.float32)
    want_sorted, want_ranges, cap = orc.get_sorted_gaussian_list(f["uv"], xyz, f["radius"], ntx, nty)
    uv, xyz_d, radius = _dev(torch, f["uv"]), _dev(torch, xyz), _dev(torch, f["radius"])
    count = ops.get_sorted_gaussian_list(uv, xyz_d, radius, ntx, nty, M, 0, None, None)
    assert count == cap
    srt = torch.full((count,), -1, dtype=torch.int32, device="cuda")
    ranges = torch.full((ntx * nty + 1,), -1, dtype=torch.int32, device="cuda")
    ops.get_sorted_gaussian_list(uv, xyz_d, radius, ntx, nty, M, count, srt, ranges)
    assert (ranges.cpu().numpy() == want_ranges).all()
    assert (srt.cpu().numpy()[:len(want_sorted)] == want_sorted).all()


def test_binning_list_lengths_around_the_sort_network_sizes_with_tied_depths(gpu, orc):
    """One tile per list length around every size class of the per-tile sort (64 E entries in registers for E = 1..16,
    the workgroup kernels beyond 1024) and only five distinct depths, so that most of the order is decided by the
    gaussian id: the DPP / bpermute lane exchanges and the tie-break must give the reference's lists bit for bit."""
    torch, ops = gpu, pkg("ops")
    lengths = [1, 63, 64, 65, 127, 128, 129, 255, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, 4100]
    ntx, nty = len(lengths), 1
    rng = np.random.default_rng(11)
    uv, z = [], []
    for t, n in enumerate(lengths):
        uv.append(np.stack([16.0 * t + 2.0 + 12.0 * rng.random(n), 2.0 + 12.0 * rng.random(n)], 1))
        z.append(rng.choice(np.float32([0.5, 1.25, 2.0, 7.5, 31.0]), n))
    uv = np.concatenate(uv).astype(np.float32)
    M = len(uv)
    order = rng.permutation(M)  # ids unrelated to tiles
    uv = uv[order]
    xyz = np.zeros((M, 3), np.float32)
    xyz[:, 2] = np.concatenate(z)[order]
    radius = np.tile(np.float32([1.0, 1.0, 0.0, 1.0]), (M, 1))  # one pixel: every gaussian stays inside its tile
    want_sorted, want_ranges, cap = orc.get_sorted_gaussian_list(uv, xyz, radius, ntx, nty)
    assert list(np.diff(want_ranges)) == lengths
    d_uv, d_xyz, d_r = _dev(torch, uv), _dev(torch, xyz), _dev(torch, radius)
    count = ops.get_sorted_gaussian_list(d_uv, d_xyz, d_r, ntx, nty, M, 0, None, None)
    assert count == cap
    srt = torch.full((count,), -1, dtype=torch.int32, device="cuda")
    ranges = torch.full((ntx * nty + 1,), -1, dtype=torch.int32, device="cuda")
    ops.get_sorted_gaussian_list(d_uv, d_xyz, d_r, ntx, nty, M, count, srt, ranges)
    assert (ranges.cpu().numpy() == want_ranges).all()
    got = srt.cpu().numpy()[:len(want_sorted)]
    for t in range(ntx):
        a, b = want_ranges[t], want_ranges[t + 1]
        assert (got[a:b] == want_sorted[a:b]).all(), f"tile {t} ({lengths[t]} entries)"


def test_known_answer_binning(gpu):  # reference tests/cuda_forward_test.cpp:422-538
    torch, ops = gpu, pkg("ops")
    uv = _dev(torch, np.array([24, 24, 32, 24, 40, 40], np.float32))
    xyz = _dev(torch, np.array([0, 0, 10, 0, 0, 20, 0, 0, 5], np.float32))
    radius = _dev(torch, np.array([4, 4, 0, 1, 4, 4, 0, 1, 6, 6, 0, 1], np.float32))
    assert ops.get_sorted_gaussian_list(uv, xyz, radius, 4, 4, 3, 0, None, None) == 48
    srt = torch.zeros(48, dtype=torch.int32, device="cuda")
    rng = torch.zeros(17, dtype=torch.int32, device="cuda")
    ops.get_sorted_gaussian_list(uv, xyz, radius, 4, 4, 3, 48, srt, rng)
    assert srt.cpu().tolist()[:4] == [0, 1, 1, 2]
    r = rng.cpu().tolist()
    assert [r[5], r[6], r[7], r[10], r[11], r[16]] == [0, 2, 3, 3, 4, 4]


def _render_inputs(torch, f):
    return (_dev(torch, f["uv"]), _dev(torch, f["opacity"]), _dev(torch, f["conic"]), _dev(torch, f["rgb"]),
            _dev(torch, f["sorted"]), _dev(torch, f["ranges"]))


def test_render_image(gpu, case):
    torch, ops = gpu, pkg("ops")
    f, W, H = case["fwd"], case["W"], case["H"]
    uv, op, conic, rgb, srt, rng = _render_inputs(torch, f)
    n = torch.zeros(H, W, dtype=torch.int32, device="cuda")
    T = torch.zeros(H, W, device="cuda")
    img = torch.zeros(H, W, 3, device="cuda")
    ops.render_image(uv, op, conic, rgb, 0.5, srt, rng, W, H, n, T, img)
    assert_image_close(img.cpu().numpy(), f["image"], "image")
    assert_image_close(T.cpu().numpy(), f["T"], "final transmittance")
    assert_stop_indices_close(n.cpu().numpy(), f["n"])


def test_known_answer_render(gpu):  # reference tests/cuda_forward_test.cpp:631-767
    torch, ops = gpu, pkg("ops")
    from test_oracle_known_answers import _expected_color
    uv = [7.5, 7.5, 3.5, 3.5, 11.5, 11.5]
    opacity, rgb = [0.5, 0.6, 0.4], [1.0, 0.8, 0.4, 0.4, 0.8, 1.0, 0.8, 1.0, 0.4]
    conic = [1.0, 0.0, 1.0, 2.0, 0.5, 2.0, 1.5, -0.5, 1.5]
    f32 = lambda a: _dev(torch, np.array(a, np.float32))
    n = torch.zeros(16, 16, dtype=torch.int32, device="cuda")
    T, img = torch.zeros(16, 16, device="cuda"), torch.zeros(16, 16, 3, device="cuda")
    ops.render_image(f32(uv), f32(opacity), f32(conic), f32(rgb), 1.0, _dev(torch, np.array([0, 1, 2], np.int32)),
                     _dev(torch, np.array([0, 3], np.int32)), 16, 16, n, T, img)
    img = img.cpu().numpy()
    np.testing.assert_allclose(img[7, 7], _expected_color(7.0, 7.0, uv, opacity, conic, rgb, 1.0), atol=1e-3)
    np.testing.assert_allclose(img[0, 0], [1, 1, 1], atol=1e-3)
    assert (n.cpu().numpy() == 3).all()


def test_render_image_backward(gpu, case):
    torch, ops = gpu, pkg("ops")
    f, b, W, H = case["fwd"], case["bwd"], case["W"], case["H"]
    M = f["num_culled"]
    uv, op, conic, rgb, srt, rng = _render_inputs(torch, f)
    g_rgb, g_op = torch.zeros(M, 3, device="cuda"), torch.zeros(M, device="cuda")
    g_uv, g_conic = torch.zeros(M, 2, device="cuda"), torch.zeros(M, 3, device="cuda")
    ops.render_image_backward(uv, op, conic, rgb, 0.5, srt, rng, _dev(torch, f["n"]), _dev(torch, f["T"]),
                              _dev(torch, case["gi"]), W, H, g_rgb, g_op, g_uv, g_conic)
    assert_grad_close(g_rgb.cpu().numpy(), b["rgb_pre"], "grad_rgb")
    assert_grad_close(g_op.cpu().numpy(), b["opacity"], "grad_opacity")
    assert_grad_close(g_uv.cpu().numpy(), b["uv"], "grad_uv")
    assert_grad_close(g_conic.cpu().numpy(), b["conic"], "grad_conic")
    # "+=": a second call doubles the result
    ops.render_image_backward(uv, op, conic, rgb, 0.5, srt, rng, _dev(torch, f["n"]), _dev(torch, f["T"]),
                              _dev(torch, case["gi"]), W, H, g_rgb, g_op, g_uv, g_conic)
    assert_grad_close(g_op.cpu().numpy(), 2 * b["opacity"], "grad_opacity accumulated")


@pytest.mark.parametrize("bg", [0.0, 0.5])
def test_render_image_backward_gate(gpu, orc, bg):
    """cuda/render_backward.cu:170: ALL nine atomics of a (gaussian, tile) pair are skipped unless some thread's
    d/d logit is non-zero.  Three ways to get there: sigmoid(opacity) == 1.0f (logit 20), a tile whose grad_image is
    exactly zero, and -- over a zero background -- a colour equal to the colour behind it (d/d alpha == 0 although the
    colour sums alpha*T*grad are not)."""
    torch, ops = gpu, pkg("ops")
    W, H = 48, 16
    uv = np.array([[8, 8], [6, 9], [24.5, 8], [40, 7.5]], np.float32)
    opacity = np.array([20.0, 0.5, 1.0, 0.3], np.float32)
    conic = np.array([[0.05, 0, 0.05], [0.08, 0.01, 0.06], [0.1, 0, 0.1], [0.07, -0.01, 0.09]], np.float32)
    rgb = np.array([[0.9, 0.2, 0.1], [0.3, 0.8, 0.5], [0.6, 0.6, 0.2], [0.0, 0.0, 0.0]], np.float32)
    srt, rng = np.array([1, 0, 2, 3], np.int32), np.array([0, 2, 3, 4], np.int32)
    assert np.float32(1) / (np.float32(1) + np.exp(np.float32(-20))) == np.float32(1)
    n, T, _ = orc.render_image(uv, opacity, conic, rgb, bg, srt, rng, W, H)
    gi = np.random.default_rng(5).uniform(-1, 1, (H, W, 3)).astype(np.float32)
    gi[:, 16:32] = 0.0  # tile 1
    want = orc.render_image_backward(uv, opacity, conic, rgb, bg, srt, rng, n, T, gi, W, H)
    got = [torch.zeros(4, 3, device="cuda"), torch.zeros(4, device="cuda"), torch.zeros(4, 2, device="cuda"),
           torch.zeros(4, 3, device="cuda")]
    ops.render_image_backward(_dev(torch, uv), _dev(torch, opacity), _dev(torch, conic), _dev(torch, rgb), bg,
                              _dev(torch, srt), _dev(torch, rng), _dev(torch, n), _dev(torch, T), _dev(torch, gi), W, H, *got)
    got = [g.cpu().numpy() for g in got]
    for g, w, what in zip(got, want, ("rgb", "opacity", "uv", "conic")):
        np.testing.assert_allclose(g, w, rtol=1e-3, atol=1e-6 * np.abs(w).max(), err_msg=what)
        assert (g[0] == 0).all() and (w[0] == 0).all(), what + ": the fully opaque gaussian gets no gradient"
        assert (g[2] == 0).all() and (w[2] == 0).all(), what + ": zero pixel gradients add nothing"
        if bg == 0.0:  # d/d alpha == 0 on every pixel: the reference drops the colour sums too
            assert (g[3] == 0).all() and (w[3] == 0).all(), what + ": black over black"
    assert np.abs(want[0][1]).max() > 0
    if bg != 0.0:
        assert np.abs(want[0][3]).max() > 0 and np.abs(got[0][3]).max() > 0


def test_per_gaussian_backward_chain(gpu, case):
    """conic -> (J, Sigma) -> (xyz_c, quaternion, scale), uv -> xyz_c -> xyz, with the reference's += chaining."""
    torch, ops = gpu, pkg("ops")
    f, b, cam, W, H = case["fwd"], case["bwd"], case["cam"], case["W"], case["H"]
    M = f["num_culled"]
    view, proj = _dev(torch, cam["view"]), _dev(torch, cam["proj"])
    gJ, gS = torch.zeros(M, 6, device="cuda"), torch.zeros(M, 6, device="cuda")
    ops.compute_conic_backward(_dev(torch, f["J"]), _dev(torch, f["sigma"]), view, _dev(torch, f["conic"]),
                               _dev(torch, b["conic"]), M, gJ, gS)
    assert_grad_close(gJ.cpu().numpy(), b["J"], "grad_J", rel=1e-4)
    assert_grad_close(gS.cpu().numpy(), b["sigma"], "grad_sigma", rel=1e-4)
    tfx = float(np.tan(np.float32(2) * np.arctan(np.float32(W) / (np.float32(2) * np.float32(cam["fx"]))) * np.float32(.5)))
    tfy = float(np.tan(np.float32(2) * np.arctan(np.float32(H) / (np.float32(2) * np.float32(cam["fy"]))) * np.float32(.5)))
    g_xyz_c = torch.zeros(M, 3, device="cuda")
    ops.compute_projection_jacobian_backward(_dev(torch, f["xyz_c"]), cam["fx"], cam["fy"], tfx, tfy,
                                             _dev(torch, b["J"]), M, g_xyz_c)
    gq, gs = torch.empty(M, 4, device="cuda"), torch.empty(M, 3, device="cuda")
    ops.compute_sigma_backward(_dev(torch, f["quaternion"]), _dev(torch, f["scale"]), _dev(torch, b["sigma"]), M, gq, gs)
    assert_grad_close(gq.cpu().numpy(), b["quaternion"], "grad_quaternion", rel=1e-4)
    assert_grad_close(gs.cpu().numpy(), b["scale"], "grad_scale", rel=1e-4)
    ops.project_to_screen_backward(_dev(torch, f["xyz_c"]), proj, _dev(torch, b["uv"]), M, W, H, g_xyz_c)
    assert_grad_close(g_xyz_c.cpu().numpy(), b["xyz_c"], "grad_xyz_c", rel=1e-4)
    sh_part = b["xyz"] - 0  # oracle's final grad_xyz = SH term + R^T grad_xyz_c
    g_xyz = torch.zeros(M, 3, device="cuda")
    ops.compute_camera_space_points_backward(_dev(torch, f["xyz"]), view, _dev(torch, b["xyz_c"]), M, g_xyz)
    R = np.asarray(cam["view"]).reshape(4, 4)[:3, :3]
    np.testing.assert_allclose(g_xyz.cpu().numpy(), b["xyz_c"] @ R, rtol=1e-4, atol=1e-7)
    assert np.isfinite(sh_part).all()


def test_projection_jacobian_backward_alone(gpu, orc, case):
    """H1 (cuda/gaussian_backward.cu:6-95) on its own, against the oracle: the `+=` into a pre-filled xyz_c gradient
    and the plain value (the chain test above only sees it summed with Q1)."""
    torch, ops = gpu, pkg("ops")
    f, b, cam, W, H = case["fwd"], case["bwd"], case["cam"], case["W"], case["H"]
    M = f["num_culled"]
    tfx = float(np.tan(np.float32(2) * np.arctan(np.float32(W) / (np.float32(2) * np.float32(cam["fx"]))) * np.float32(.5)))
    tfy = float(np.tan(np.float32(2) * np.arctan(np.float32(H) / (np.float32(2) * np.float32(cam["fy"]))) * np.float32(.5)))
    want = orc.compute_projection_jacobian_backward(f["xyz_c"], cam["fx"], cam["fy"], tfx, tfy, b["J"])
    assert np.abs(want).max() > 0
    got = torch.zeros(M, 3, device="cuda")
    ops.compute_projection_jacobian_backward(_dev(torch, f["xyz_c"]), cam["fx"], cam["fy"], tfx, tfy, _dev(torch, b["J"]), M, got)
    assert_grad_close(got.cpu().numpy(), want, "H1 alone", rel=1e-5)
    seed = np.random.default_rng(0).normal(size=(M, 3)).astype(np.float32) * np.abs(want).mean()
    got2 = _dev(torch, seed.copy())
    ops.compute_projection_jacobian_backward(_dev(torch, f["xyz_c"]), cam["fx"], cam["fy"], tfx, tfy, _dev(torch, b["J"]), M, got2)
    assert_grad_close(got2.cpu().numpy(), seed + want, "H1 accumulates", rel=1e-5)


def test_compact_and_scatter(gpu, orc):  # reference tests/cuda_data_test.cpp:38-125
    torch, ops = gpu, pkg("ops")
    rng = np.random.default_rng(3)
    for N, stride in [(5, 1), (4, 3), (1000, 45), (2, 3)]:
        src = rng.normal(size=N * stride).astype(np.float32)
        for mask in [rng.random(N) < 0.5, np.ones(N, bool), np.zeros(N, bool)]:
            out = ops.compact_masked_array(stride, _dev(torch, src), _dev(torch, mask.astype(np.uint8)))
            ref = orc.compact_masked_array(src, mask, stride)
            assert out.shape[0] == ref.shape[0]
            if ref.size:
                assert (out.cpu().numpy() == ref).all()
            dst = torch.zeros(N * stride, device="cuda")
            ops.scatter_masked_array(stride, out if ref.size else None, _dev(torch, mask.astype(np.uint8)), dst)
            assert (dst.cpu().numpy() == orc.scatter_masked_array(ref, mask, stride, np.zeros(N * stride))).all()
    empty = ops.compact_masked_array(3, torch.empty(0, device="cuda"), torch.empty(0, dtype=torch.uint8, device="cuda"))
    assert empty.numel() == 0


def test_error_codes_instead_of_exit(gpu):
    """Null / host pointers are reported as status codes (the reference exits: cuda/checks.cuh:17-38)."""
    torch, ops, lib_mod = gpu, pkg("ops"), pkg("_lib")
    good = torch.zeros(16, device="cuda")
    with pytest.raises(lib_mod.GsplatError) as e:
        ops.compute_camera_space_points(None, good, 1, good)
    assert e.value.code == -1
    import ctypes
    host = (ctypes.c_float * 16)()
    lib = lib_mod.load()
    rc = lib.gsplat_compute_camera_space_points(ctypes.cast(host, ctypes.c_void_p), ctypes.c_void_p(good.data_ptr()), 1,
                                                ctypes.c_void_p(good.data_ptr()), None)
    assert rc == -2 and b"device" in lib.gsplat_last_error()
    with pytest.raises(lib_mod.GsplatError) as e:
        ops.precompute_spherical_harmonics(good, good, good, [0, 0, 0], 7, 1, good)
    assert e.value.code == -3


def test_device_block_pool(gpu):
    """gsplat_pool_alloc / _free / _release (r04: what the drop-in headers' per-iteration vectors draw from): a freed block
    is handed to the next request of its size class, a foreign or doubly freed pointer is refused, release returns the idle
    blocks to the runtime."""
    import ctypes
    torch, lib = gpu, pkg("_lib").load()
    lib.gsplat_pool_release()
    base_idle = lib.gsplat_pool_bytes(1)
    p1, p2, p3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.gsplat_pool_alloc(ctypes.byref(p1), 1_000_000) == 0 and p1.value
    assert lib.gsplat_pool_alloc(ctypes.byref(p2), 1_000_000) == 0 and p2.value and p2.value != p1.value
    live = lib.gsplat_pool_bytes(0)
    assert live >= 2_000_000 and live <= 2 * 1_125_000 + base_idle  # size classes: at most 12.5 % slack
    # the blocks are ordinary device memory
    t = torch.empty(0, device="cuda")
    assert lib.gsplat_compute_camera_space_points(p1, p1, 0, p2, None) == 0  # (N = 0: pointer checks only)
    assert lib.gsplat_pool_free(p1) == 0
    assert lib.gsplat_pool_bytes(1) > base_idle
    assert lib.gsplat_pool_alloc(ctypes.byref(p3), 1_010_000) == 0  # same class (16 x 64 KiB)
    assert p3.value == p1.value, "a request of the same size class takes the cached block"
    assert lib.gsplat_pool_free(p3) == 0 and lib.gsplat_pool_free(p2) == 0
    assert lib.gsplat_pool_free(p2) == -3 and b"not allocated" in lib.gsplat_last_error()  # freed twice
    assert lib.gsplat_pool_free(ctypes.c_void_p(t.data_ptr() or 4096)) == -3              # never ours
    zero = ctypes.c_void_p(1)
    assert lib.gsplat_pool_alloc(ctypes.byref(zero), 0) == 0 and not zero.value
    assert lib.gsplat_pool_release() == 0 and lib.gsplat_pool_bytes(1) == 0


def test_device_block_pool_orders_reuse_across_streams(gpu):
    """r06: a block returned while stream A still has work queued on it and taken by stream B is
    ordered behind A (gsplat_pool_free_on / gsplat_pool_alloc_on).  Stream A queues a long chain of fills of the block
    with 1.0 and returns it; stream B takes the block of that class -- the same one -- and fills it with 2.0 at once.
    Without the ordering B's single fill finishes long before A's chain does and the block ends as 1.0.  A block taken
    by the stream that returned it needs no ordering and gets none."""
    import ctypes
    torch, lib, raster = gpu, pkg("_lib").load(), pkg("raster")
    lib.gsplat_pool_release()
    n = 64 << 20  # floats: 256 MB, a fill takes ~0.1 ms
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    a, b = ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(sb.cuda_stream)
    p, q, r = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.gsplat_pool_alloc_on(ctypes.byref(p), 4 * n, a) == 0 and p.value
    before = lib.gsplat_pool_cross_stream_reuses()
    for _ in range(40):
        assert lib.gsplat_fill_f32(p, n, 1.0, a) == 0
    assert lib.gsplat_pool_free_on(p, a) == 0
    assert lib.gsplat_pool_alloc_on(ctypes.byref(q), 4 * n, b) == 0
    assert q.value == p.value, "the cached block of the class is reused"
    assert lib.gsplat_pool_cross_stream_reuses() == before + 1
    assert lib.gsplat_fill_f32(q, n, 2.0, b) == 0
    torch.cuda.synchronize()
    got = raster._view(q.value, (n,), "<f4", None)
    assert float(got.min()) == 2.0 and float(got.max()) == 2.0, "stream B's fill ran before stream A's queued fills"
    del got
    # the same stream again: stream order, no event
    assert lib.gsplat_pool_free_on(q, b) == 0
    assert lib.gsplat_pool_alloc_on(ctypes.byref(r), 4 * n, b) == 0 and r.value == p.value
    assert lib.gsplat_pool_cross_stream_reuses() == before + 1
    assert lib.gsplat_pool_free_on(r, b) == 0
    torch.cuda.synchronize()
    # trim: nothing above the budget stays cached on this device; a budget above what is cached frees nothing
    idle = lib.gsplat_pool_bytes(1)
    assert idle >= 4 * n
    assert lib.gsplat_pool_trim(idle) == 0 and lib.gsplat_pool_bytes(1) == idle
    assert lib.gsplat_pool_trim(0) == 0 and lib.gsplat_pool_bytes(1) == 0


def test_compact_with_trusted_count_matches_counted(gpu, orc):
    """compact_masked_array with num_culled given (the reference's call sites: no read-back, asynchronous) against the
    counting form, for the strides with a compile-time instantiation and two without."""
    torch, ops = gpu, pkg("ops")
    rng = np.random.default_rng(9)
    N = 70001  # not a multiple of anything: ragged last slice, last block, last wave
    mask = rng.random(N) < 0.37
    d_mask = _dev(torch, mask.astype(np.uint8))
    for stride in (1, 2, 3, 4, 6, 9, 24, 45, 5, 7):
        src = rng.normal(size=N * stride).astype(np.float32)
        ref = orc.compact_masked_array(src, mask, stride)
        a = ops.compact_masked_array(stride, _dev(torch, src), d_mask)
        b = ops.compact_masked_array(stride, _dev(torch, src), d_mask, int(mask.sum()))
        assert (a.cpu().numpy() == ref).all() and (b.cpu().numpy() == ref).all(), stride
        dst = torch.full((N * stride,), -1.0, device="cuda")
        ops.scatter_masked_array(stride, b, d_mask, dst)
        want = orc.scatter_masked_array(ref, mask, stride, np.full(N * stride, -1.0, np.float32))
        assert (dst.cpu().numpy() == want).all(), stride


def test_compaction_helpers_of_the_drop_in_headers(gpu, orc):
    """r05 entry points behind include/gsplat_cuda/cuda_data.cuh: the bounded compaction (a count that is too small drops
    rows instead of writing past dst), the compaction with known slots (gsplat_compact_rows_ranked: what a forward's
    context already knows about its mask), the row list of a mask, and the asynchronous fill the thrust::fill_n overload
    routes to -- each against the oracle's compaction / numpy."""
    import ctypes
    torch, lib = gpu, pkg("_lib").load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rng = np.random.default_rng(11)
    N = 70001
    mask = rng.random(N) < 0.37
    M = int(mask.sum())
    d_mask = _dev(torch, mask.astype(np.uint8))
    slots = _dev(torch, (np.cumsum(mask) - mask).astype(np.int32))  # exclusive scan of the mask
    slots[~torch.as_tensor(mask).cuda()] = 2 ** 30  # entries of culled rows are unspecified: must never be used
    rows = torch.full((M + 5,), -7, dtype=torch.int32, device="cuda")
    assert lib.gsplat_mask_selected_rows(P(d_mask), N, P(rows), M, None) == 0
    assert (rows[:M].cpu().numpy() == np.nonzero(mask)[0]).all() and (rows[M:] == -7).all()
    for stride in (1, 2, 3, 4, 6, 9, 24, 45, 5):
        src = rng.normal(size=N * stride).astype(np.float32)
        ref = orc.compact_masked_array(src, mask, stride)
        d_src = _dev(torch, src)
        out = torch.full((M * stride + 64,), -3.0, device="cuda")
        assert lib.gsplat_compact_rows_ranked(P(d_src), P(d_mask), P(slots), N, stride, P(out), M, None) == 0
        assert (out[:M * stride].cpu().numpy() == ref).all() and (out[M * stride:] == -3.0).all(), stride
        # room for fewer rows than are selected: the rest is dropped, nothing behind dst is touched
        short = M // 2
        for fn, extra in ((lib.gsplat_compact_rows_ranked, (P(slots),)), (lib.gsplat_compact_masked_array_bounded, ())):
            out = torch.full((M * stride + 64,), -3.0, device="cuda")
            tail = (short, None) if extra else (short, None, None)
            assert fn(P(d_src), P(d_mask), *extra, N, stride, P(out), *tail) == 0
            assert (out[:short * stride].cpu().numpy() == ref[:short * stride]).all(), stride
            assert (out[short * stride:] == -3.0).all(), stride
    buf = torch.full((1000,), 5.0, device="cuda")
    assert lib.gsplat_fill_f32(ctypes.c_void_p(buf.data_ptr() + 4 * 10), 900, 0.0, None) == 0
    assert lib.gsplat_fill_f32(ctypes.c_void_p(buf.data_ptr() + 4 * 910), 37, -2.5, None) == 0
    b = buf.cpu().numpy()
    assert (b[:10] == 5).all() and (b[10:910] == 0).all() and (b[910:947] == -2.5).all() and (b[947:] == 5).all()


def test_forward_outputs_can_be_handed_over(gpu, scene, orc):
    """gsplat_context_detach_forward_outputs (r05: how the rasterize_image shim fills ForwardPassData without copies): the
    thirteen output blocks stay valid in the caller's hands while the context runs its next forward into fresh pool
    blocks -- and finds the returned ones there afterwards; gsplat_context_last_compaction describes the handed-over mask
    until that next forward."""
    import ctypes
    torch, raster, lib = gpu, pkg("raster"), pkg("_lib").load()
    N, W, H, L, _ = scene.WORKLOADS["small"]
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::4, 2] *= -1
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp = raster.device_params(params)
    cams = [raster.device_camera(scene.make_camera(W, H, v)) for v in (0, 3)]
    refs = [orc.rasterize(params, scene.make_camera(W, H, v), c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=8)
            for v in (0, 3)]
    f0 = ctx.rasterize_image(dp, cams[0], c, c["bg"], L)
    ALL = ("mask", "uv_all", "xyz_c_all", "sigma", "conic", "J", "rgb", "radius", "sorted", "ranges", "image", "T", "n")
    held = {k: f0[k] for k in ALL}  # views of the context's thirteen output blocks
    ptrs = {k: v.data_ptr() for k, v in held.items()}
    m, s, r = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    n_, m_ = ctypes.c_int(), ctypes.c_int()
    assert lib.gsplat_context_last_compaction(ctx._h, ctypes.byref(m), ctypes.byref(s), ctypes.byref(r), ctypes.byref(n_), ctypes.byref(m_)) == 0
    assert m.value == ptrs["mask"] and n_.value == N and m_.value == f0["num_culled"]
    assert lib.gsplat_context_detach_forward_outputs(ctx._h) == 0
    f1 = ctx.rasterize_image(dp, cams[1], c, c["bg"], L)  # runs into OTHER blocks
    assert all(f1[k].data_ptr() != ptrs[k] for k in ptrs)
    torch.cuda.synchronize()
    np.testing.assert_allclose(held["image"].cpu().numpy(), refs[0]["image"], atol=2e-5)
    assert (held["sorted"].cpu().numpy() == refs[0]["sorted"]).all() and (held["mask"].cpu().numpy().astype(bool) == refs[0]["mask"]).all()
    np.testing.assert_allclose(f1["image"].cpu().numpy(), refs[1]["image"], atol=2e-5)
    assert (f1["sorted"].cpu().numpy() == refs[1]["sorted"]).all()
    lib.gsplat_context_last_compaction(ctx._h, ctypes.byref(m), None, None, None, None)
    assert m.value == f1["mask"].data_ptr()  # the association moved on with the forward
    for p in ptrs.values():  # the caller returns its blocks ...
        assert lib.gsplat_pool_free(ctypes.c_void_p(p)) == 0
    assert lib.gsplat_context_detach_forward_outputs(ctx._h) == 0
    f2 = ctx.rasterize_image(dp, cams[0], c, c["bg"], L)  # ... and the next forward finds them in the pool
    assert {f2[k].data_ptr() for k in ptrs} & set(ptrs.values())
    torch.cuda.synchronize()
    np.testing.assert_allclose(f2["image"].cpu().numpy(), refs[0]["image"], atol=2e-5)
    for k in ALL:
        assert lib.gsplat_pool_free(ctypes.c_void_p(f1[k].data_ptr())) == 0  # (the second forward's blocks, handed over above)
    # once the caller has returned the mask block of a forward, the context forgets that it described it: the pointer may
    # name somebody else's data from now on
    f3 = ctx.rasterize_image(dp, cams[1], c, c["bg"], L)
    lib.gsplat_context_last_compaction(ctx._h, ctypes.byref(m), None, None, None, None)
    assert m.value == f3["mask"].data_ptr()
    assert lib.gsplat_context_detach_forward_outputs(ctx._h) == 0
    assert lib.gsplat_pool_free(ctypes.c_void_p(f3["mask"].data_ptr())) == 0
    lib.gsplat_context_last_compaction(ctx._h, ctypes.byref(m), None, None, None, None)
    assert m.value is None
    for k in ALL:
        if k != "mask":
            assert lib.gsplat_pool_free(ctypes.c_void_p(f3[k].data_ptr())) == 0
    # (f2's arrays were the context's own when f3 overwrote them: the same blocks, returned just now)
    assert lib.gsplat_pool_free(ctypes.c_void_p(f2["image"].data_ptr())) != 0  # already returned: refused, not freed twice
