"""Pins the CPU oracle (oracle/) against the reference's own known-answer vectors.

Every case restates the inputs and expected values of a test in
/root/reference/tests/cuda_forward_test.cpp, tests/cuda_backward_test.cpp or
tests/cuda_data_test.cpp (cited per test).  The backward operators are pinned the way the
reference pins them -- central finite differences of the forward -- but in float64, so the
check is sharp (1e-6) instead of the reference's 1e-1..1e-3.
"""
import numpy as np
import pytest

F32, F64 = np.float32, np.float64


# ------------------------------------------------------------------------------- forward pins
def test_compute_sigma(orc):  # cuda_forward_test.cpp:37-90
    q = [1, 0, 0, 0, np.sqrt(0.5), 0, 0, np.sqrt(0.5)]
    s = np.log([2, 3, 4, 1, 2, 3])
    sigma = orc.compute_sigma(q, s)
    np.testing.assert_allclose(sigma, [[4, 0, 0, 9, 0, 16], [4, 0, 0, 1, 0, 9]], atol=1e-4)


def test_project_to_screen(orc):  # cuda_forward_test.cpp:93-156
    proj = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0]
    xyz = np.array([[1, 1, 2], [2, -3, 5], [0, 0, 1], [-4, 2, 10]], F32)
    uv = orc.project_to_screen(xyz, proj, 1920, 1080)
    exp = np.stack([(xyz[:, 0] / xyz[:, 2] * 0.5 + 0.5) * 1920, (xyz[:, 1] / xyz[:, 2] * 0.5 + 0.5) * 1080], 1)
    np.testing.assert_allclose(uv, exp, atol=1e-3)
    np.testing.assert_allclose(uv[0], [0.75 * 1920, 0.75 * 1080], atol=1e-3)


def test_cull_gaussians(orc):  # cuda_forward_test.cpp:159-230
    xyz = np.array([[0, 0, 5], [0, 0, .5], [0, 0, 12], [0, 0, 5], [0, 0, 5], [0, 0, 12], [0, 0, .5]], F32)
    uv = np.array([[960, 540], [960, 540], [960, 540], [-5, 540], [1925, 540], [-11, 540], [960, 1091]], F32)
    mask = orc.cull_gaussians(uv, xyz, 1.0, 10, 1920, 1080)
    assert mask.tolist() == [True, False, True, True, True, False, False]
    # inclusive bounds (cuda/culling.cu:72-75)
    edge = orc.cull_gaussians([[-10, 1090], [1930, -10]], [[0, 0, 1.0], [0, 0, 1.0]], 1.0, 10, 1920, 1080)
    assert edge.tolist() == [True, True]


def test_compute_camera_space_points(orc):  # cuda_forward_test.cpp:233-302
    view = [1, 0, 0, 10, 0, 1, 0, 20, 0, 0, 1, 30, 0, 0, 0, 1]
    xyz = np.array([[1, 2, 3], [-5, 4, -1], [0, 0, 0]], F32)
    np.testing.assert_allclose(orc.compute_camera_space_points(xyz, view), xyz + [10, 20, 30], atol=1e-5)


def test_compute_conic(orc):  # cuda_forward_test.cpp:306-414
    J, conic, radius = orc.compute_conic([1, 2, 5], np.eye(4).ravel(), [1, 0, 0, 1, 0, 1], 1, 1, 1, 1, 3.0)
    x, y, z = 1.0, 2.0, 5.0
    j00, j02, j11, j12 = 1 / z, -x / z ** 2, 1 / z, -y / z ** 2
    np.testing.assert_allclose(J[0], [j00, 0, j02, 0, j11, j12], atol=1e-6)
    c00, c01, c11 = j00 ** 2 + j02 ** 2 + 0.3, j02 * j12, j11 ** 2 + j12 ** 2 + 0.3
    det = c00 * c11 - c01 ** 2
    np.testing.assert_allclose(conic[0], [c11 / det, -c01 / det, c00 / det], atol=1e-5)
    np.testing.assert_allclose(radius[0], [3.0, 1.0, np.sqrt(0.8), np.sqrt(0.2)], atol=1e-5)


def test_get_sorted_gaussian_list(orc):  # cuda_forward_test.cpp:422-538
    uv = [24, 24, 32, 24, 40, 40]
    xyz = [0, 0, 10, 0, 0, 20, 0, 0, 5]
    radius = [4, 4, 0, 1, 4, 4, 0, 1, 6, 6, 0, 1]
    assert orc.count_tile_pairs(uv, radius, 4, 4) == 3 * 4 * 4
    sorted_ids, ranges, cap = orc.get_sorted_gaussian_list(uv, xyz, radius, 4, 4)
    assert cap == 48
    assert sorted_ids.tolist() == [0, 1, 1, 2]
    assert [ranges[5], ranges[6], ranges[7], ranges[10], ranges[11]] == [0, 2, 3, 3, 4]
    assert ranges[9] >= ranges[8]
    assert ranges[16] == 4


def test_sorted_list_orders_by_depth_then_id(orc):
    uv = [8, 8] * 4
    radius = [2, 2, 0, 1] * 4
    xyz = [0, 0, 3.0, 0, 0, 1.0, 0, 0, 3.0, 0, 0, 2.0]
    s, r, _ = orc.get_sorted_gaussian_list(uv, xyz, radius, 2, 2)
    assert s.tolist() == [1, 3, 0, 2]
    assert r.tolist() == [0, 4, 4, 4, 4]


def test_nan_minor_radius_passes_every_coarse_tile(orc):  # SURVEY 8a hazard 2, cuda/gaussian.cu:162-169
    uv, xyz = [40, 40], [0, 0, 1]
    s, r, cap = orc.get_sorted_gaussian_list(uv, xyz, [4, np.nan, 0.6, 0.8], 6, 6)
    assert cap == 25 and len(s) == 25


def test_precompute_spherical_harmonics_l1(orc):  # cuda_forward_test.cpp:541-627
    xyz = [0, 0, 1, 1, 0, 0]
    band0 = [0.5, -0.2, 0.8, 0.1, 0.5, 0.9]
    sh = [0.1, 0.1, 0.1, 0.2, 0.2, 0.2, 0.3, 0.3, 0.3, 0.2, 0.6, 0.0, 0.3, 0.7, 0.1, 0.4, 0.8, 0.2]
    rgb = orc.precompute_spherical_harmonics(xyz, sh, band0, [0, 0, 0], 1)
    exp = [0.5 * .28209 + .5 + .2 * .4886, -.2 * .28209 + .5 + .2 * .4886, .8 * .28209 + .5 + .2 * .4886,
           .1 * .28209 + .5 + .4 * .4886, .5 * .28209 + .5 + .8 * .4886, .9 * .28209 + .5 + .2 * .4886]
    np.testing.assert_allclose(rgb.ravel(), exp, atol=1e-4)


def test_sh_basis_l2_matches_reference_formula_list(orc):  # cuda_backward_test.cpp:610-624
    rng = np.random.default_rng(0)
    C0, C1, C2, C3, C4 = 0.28209479177387814, 0.4886025119029199, 1.0925484305920792, 0.31539156525252005, \
        0.5462742152960399
    for _ in range(5):
        p = rng.normal(size=3)
        x, y, z = p / np.linalg.norm(p)
        Y = [C0, C1 * y, C1 * z, C1 * x, C2 * x * y, C2 * y * z, C3 * (3 * z * z - 1), C2 * x * z, C4 * (x * x - y * y)]
        for k in range(9):  # isolate basis k with one-hot coefficients
            band0 = np.zeros(3)
            sh = np.zeros((8, 3))
            if k == 0:
                band0[0] = 1
            else:
                sh[k - 1, 0] = 1
            rgb = orc.precompute_spherical_harmonics(p, sh, band0, [0, 0, 0], 2, dtype=F64)
            assert abs(rgb[0, 0] - 0.5 - Y[k]) < 2e-8, (k, rgb[0, 0] - 0.5, Y[k])


def test_sh_basis_l3_is_orthonormal(orc):
    """l=3 is unpinned by the reference (sphericart is not vendored); check it is the real orthonormal basis."""
    mu, wmu = np.polynomial.legendre.leggauss(16)  # exact for the degree-6 products in cos(theta)
    nphi = 32
    phi = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi
    ct, ph = np.meshgrid(mu, phi, indexing="ij")
    wt = np.repeat(wmu, nphi) * (2 * np.pi / nphi)
    st = np.sqrt(1 - ct ** 2)
    pts = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    Y = np.zeros((len(pts), 16))
    for k in range(16):
        band0 = np.zeros((len(pts), 3))
        sh = np.zeros((len(pts), 15, 3))
        if k == 0:
            band0[:, 0] = 1
        else:
            sh[:, k - 1, 0] = 1
        Y[:, k] = orc.precompute_spherical_harmonics(pts, sh, band0, [0, 0, 0], 3, dtype=F64)[:, 0] - 0.5
    gram = (Y * wt[:, None]).T @ Y
    np.testing.assert_allclose(gram, np.eye(16), atol=1e-7)


def _scipy_real_sh(pts, l_max=3):
    """Real orthonormal spherical harmonics without the Condon-Shortley phase, index l*l + l + m -- sphericart's
    published convention (the reference calls sphericart::cuda::SphericalHarmonics, cuda/spherical_harmonics.cu:72,89;
    the submodule is not vendored) -- built from scipy's complex Y_l^m, which carry the phase:
    m > 0: sqrt(2) (-1)^m Re Y_l^m,  m < 0: sqrt(2) (-1)^m Im Y_l^|m|,  m = 0: Y_l^0."""
    from scipy.special import sph_harm_y
    pts = np.asarray(pts, np.float64)
    r = np.linalg.norm(pts, axis=-1)
    theta, phi = np.arccos(pts[..., 2] / r), np.arctan2(pts[..., 1], pts[..., 0])
    Y = np.zeros(pts.shape[:-1] + ((l_max + 1) ** 2,))
    for l in range(l_max + 1):
        for m in range(-l, l + 1):
            c = sph_harm_y(l, abs(m), theta, phi)
            v = c.real if m == 0 else np.sqrt(2.0) * (-1) ** m * (c.real if m > 0 else c.imag)
            Y[..., l * l + l + m] = v
    return Y


def test_sh_basis_all_16_functions_match_scipy(orc):
    """The pin for the l = 3 block (and once more for l <= 2): every basis function of the oracle against
    scipy.special.sph_harm_y with the Condon-Shortley phase removed, index l^2 + l + m.  The residual is the 1e-9 the
    reference adds to the direction's norm (cuda/spherical_harmonics.cu:8-26)."""
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(200, 3)) * rng.uniform(0.2, 30.0, size=(200, 1))
    want = _scipy_real_sh(pts)
    got = np.zeros_like(want)
    for k in range(16):  # isolate basis k with one-hot coefficients
        band0 = np.zeros((len(pts), 3))
        sh = np.zeros((len(pts), 15, 3))
        if k == 0:
            band0[:, 0] = 1
        else:
            sh[:, k - 1, 0] = 1
        got[:, k] = orc.precompute_spherical_harmonics(pts, sh, band0, [0, 0, 0], 3, dtype=F64)[:, 0] - 0.5
    assert np.abs(got - want).max() < 2e-8, np.abs(got - want).max(axis=0)
    # and through a camera position, in float32 (what the parity checker runs)
    campos = np.array([0.3, -0.2, 0.1])
    coef = rng.normal(size=(len(pts), 16, 3))
    rgb32 = orc.precompute_spherical_harmonics(pts, coef[:, 1:], coef[:, 0], campos, 3)
    want32 = 0.5 + np.einsum("nk,nkc->nc", _scipy_real_sh(pts - campos), coef)
    np.testing.assert_allclose(rgb32, want32, rtol=0, atol=2e-5)


@pytest.mark.parametrize("l_max", [1, 2, 3])
def test_sh_backward_matches_scipy(orc, l_max):
    """Per-coefficient sh_grad (= rgb_grad x Y_k: unpinned by the reference's tests for every l) against the scipy basis,
    and the position gradient against central differences of the scipy colour (cuda/spherical_harmonics_backward.cu:
    168-209 reads sphericart's derivative buffer; the reference's own test checks it by finite differences too)."""
    rng = np.random.default_rng(5 + l_max)
    M, n = 40, (l_max + 1) ** 2
    pts = rng.normal(size=(M, 3)) * rng.uniform(0.5, 10.0, size=(M, 1))
    campos = np.array([0.25, 0.1, -0.4])
    coef = rng.normal(size=(M, n, 3))
    g = rng.normal(size=(M, 3))
    sh_g, b0_g, x_g = orc.precompute_spherical_harmonics_backward(pts, coef[:, 0], coef[:, 1:], campos, g, l_max, dtype=F64)
    Y = _scipy_real_sh(pts - campos, l_max)
    np.testing.assert_allclose(b0_g, g * Y[:, :1], atol=1e-8)
    np.testing.assert_allclose(sh_g.reshape(M, n - 1, 3), Y[:, 1:, None] * g[:, None, :], atol=2e-8)

    def loss(p):
        return (g * (0.5 + np.einsum("nk,nkc->nc", _scipy_real_sh(p - campos, l_max), coef))).sum(-1)

    fd = np.zeros((M, 3))
    for a in range(3):
        e = np.zeros(3)
        e[a] = 1e-5
        fd[:, a] = (loss(pts + e) - loss(pts - e)) / 2e-5
    np.testing.assert_allclose(x_g, fd, rtol=1e-5, atol=1e-7)


def _expected_color(px, py, uv, opacity, conic, rgb, bg):  # cuda_forward_test.cpp:705-744
    r = g = b = bg
    acc = 0.0
    for i in range(len(opacity)):
        du, dv = px - uv[2 * i], py - uv[2 * i + 1]
        a, bb, c = conic[3 * i:3 * i + 3]
        mh = a * du * du + 2 * bb * du * dv + c * dv * dv
        alpha = 0.0
        if mh > 0:
            alpha = 1 / (1 + np.exp(-opacity[i])) * np.exp(-0.5 * mh)
        alpha *= (1 - acc)
        r += (rgb[3 * i] - r) * alpha
        g += (rgb[3 * i + 1] - g) * alpha
        b += (rgb[3 * i + 2] - b) * alpha
        acc += alpha
    return [r, g, b]


def test_render_image_multiple_gaussians(orc):  # cuda_forward_test.cpp:631-767
    uv = [7.5, 7.5, 3.5, 3.5, 11.5, 11.5]
    opacity = [0.5, 0.6, 0.4]
    rgb = [1.0, 0.8, 0.4, 0.4, 0.8, 1.0, 0.8, 1.0, 0.4]
    conic = [1.0, 0.0, 1.0, 2.0, 0.5, 2.0, 1.5, -0.5, 1.5]
    n, T, image = orc.render_image(uv, opacity, conic, rgb, 1.0, [0, 1, 2], [0, 3], 16, 16)
    np.testing.assert_allclose(image[7, 7], _expected_color(7.0, 7.0, uv, opacity, conic, rgb, 1.0), atol=1e-3)
    np.testing.assert_allclose(image[0, 0], _expected_color(0.0, 0.0, uv, opacity, conic, rgb, 1.0), atol=1e-3)
    np.testing.assert_allclose(image[0, 0], [1, 1, 1], atol=1e-3)
    assert (n == 3).all()


def test_render_early_termination_counts(orc):
    """n = index of the splat that drives T below 1e-4, plus one; that splat is still blended (render.cu:70-87)."""
    k = 6
    uv = [8.0, 8.0] * k
    conic = [0.01, 0.0, 0.01] * k
    rgb = [1.0, 0.5, 0.25] * k
    n, T, image = orc.render_image(uv, [20.0] * k, conic, rgb, 0.0, list(range(k)), [0, k], 16, 16)
    # alpha is capped at 0.99: T = 0.01^m; 0.01^2 = 1e-4 is not < 1e-4 in float32? check against a direct loop
    t, cnt = np.float32(1), 0
    for _ in range(k):
        cnt += 1
        t = np.float32(t * (np.float32(1) - np.float32(0.99)))
        if t < np.float32(1e-4):
            break
    assert n[8, 8] == cnt and cnt < k
    np.testing.assert_allclose(T[8, 8], t, rtol=1e-5)
    np.testing.assert_allclose(image[8, 8], np.array([1.0, 0.5, 0.25]) * (1 - t), rtol=1e-4)


def test_compact_and_scatter(orc):  # cuda_data_test.cpp:38-125
    assert orc.compact_masked_array([1, 2, 3, 4, 5], [1, 0, 1, 0, 1], 1).tolist() == [1, 3, 5]
    src = np.array([1.0, 1.1, 1.2, 2.0, 2.1, 2.2, 3.0, 3.1, 3.2, 4.0, 4.1, 4.2], F32)
    np.testing.assert_array_equal(orc.compact_masked_array(src, [1, 0, 1, 0], 3), src[[0, 1, 2, 6, 7, 8]])
    assert orc.compact_masked_array([], [], 3).size == 0
    np.testing.assert_array_equal(orc.compact_masked_array(src[:6], [1, 1], 3), src[:6])
    assert orc.compact_masked_array(src[:6], [0, 0], 3).size == 0
    out = orc.scatter_masked_array([1, 3, 5], [1, 0, 1, 0, 1], 1, np.zeros(5))
    assert out.tolist() == [1, 0, 3, 0, 5]
    out = orc.scatter_masked_array(src[[0, 1, 2, 6, 7, 8]], [1, 0, 1, 0], 3, np.zeros(12))
    np.testing.assert_array_equal(out, np.where(np.repeat([1, 0, 1, 0], 3), src, 0).astype(F32))


# ------------------------------------------------------------------- backward pins (finite differences)
def _fd(f, x, h=1e-6):
    x = np.array(x, F64)
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        g[i] = (f(xp) - f(xm)) / (2 * h)
    return g


def test_project_to_screen_backward(orc):  # contract of cuda_backward_test.cpp:41-113
    rng = np.random.default_rng(1)
    xyz = rng.normal(size=(5, 3)) + [0, 0, 4]
    proj = rng.normal(size=16)
    guv = rng.normal(size=(5, 2))
    W, H = 640, 480
    # forward adds 1e-6 to w; the backward differentiates x/w (projection_backward.cu:57): compare with that model
    def loss(x):
        x = x.reshape(-1, 3)
        xc = x @ proj[0:3] + proj[3]
        yc = x @ proj[4:7] + proj[7]
        wc = x @ proj[12:15] + proj[15]
        return ((xc / wc * 0.5 + 0.5) * W * guv[:, 0] * (W * 0.5) / (W * 0.5) + (yc / wc * 0.5 + 0.5) * H * guv[:, 1]).sum()
    # operator semantic: d_ndc = grad_uv * {W,H}/2, so d loss/d xyz for loss = sum(uv * grad_uv)
    an = orc.project_to_screen_backward(xyz, proj, guv, W, H, dtype=F64)
    np.testing.assert_allclose(an, _fd(loss, xyz), rtol=1e-5, atol=1e-6)
    an32 = orc.project_to_screen_backward(xyz, proj, guv, W, H, dtype=F32)
    np.testing.assert_allclose(an32, an, rtol=2e-3, atol=1e-3)
    # "+=" semantics
    acc = orc.project_to_screen_backward(xyz, proj, guv, W, H, xyz_c_grad=np.ones((5, 3)), dtype=F64)
    np.testing.assert_allclose(acc, an + 1, rtol=1e-12)


def test_camera_space_points_backward(orc):  # cuda_backward_test.cpp:116-170
    rng = np.random.default_rng(2)
    view = np.eye(4)
    view[:3, :4] = rng.normal(size=(3, 4))
    xyz, g = rng.normal(size=(4, 3)), rng.normal(size=(4, 3))
    an = orc.compute_camera_space_points_backward(xyz, view.ravel(), g, dtype=F64)
    fd = _fd(lambda x: (orc.compute_camera_space_points(x.reshape(-1, 3), view.ravel(), F64) * g).sum(), xyz)
    np.testing.assert_allclose(an, fd, rtol=1e-6, atol=1e-8)


def test_projection_jacobian_backward(orc):  # cuda_backward_test.cpp:173-252
    rng = np.random.default_rng(3)
    fx, fy, tx, ty = 500.0, 480.0, 0.7, 0.5
    xyz = np.array([[0.3, -0.2, 3.0], [4.0, 0.1, 2.0], [0.1, -3.0, 2.5], [-5.0, 4.0, 3.0]])  # inside, x/y/both clamped
    gJ = rng.normal(size=(4, 6))
    an = orc.compute_projection_jacobian_backward(xyz, fx, fy, tx, ty, gJ, dtype=F64)
    fd = _fd(lambda x: (orc.projection_jacobian(x.reshape(-1, 3), fx, fy, tx, ty, F64) * gJ).sum(), xyz)
    # the backward uses 1/(z+1e-6): relative 1e-6 model difference
    np.testing.assert_allclose(an, fd, rtol=1e-4, atol=1e-4)


def test_conic_backward(orc):  # cuda_backward_test.cpp:255-409 (loss = c00 g0 + 2 c01 g1 + c11 g2)
    J = np.array([[0.6, 0.0, -0.1, 0.0, 0.6, -0.2]])
    sigma = np.array([[0.5, 0.1, 0.05, 0.5, 0.1, 0.5]])
    view = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 2.0, 0, 0, 0, 1.0])
    g = np.array([[0.5, -0.2, 0.8]])
    w = np.array([1.0, 2.0, 1.0])

    def loss_J(j):
        c, _ = orc.conic_from_J(sigma, view, j.reshape(1, 6), 3.0, F64)
        return (c * g * w).sum()

    def loss_S(s):
        c, _ = orc.conic_from_J(s.reshape(1, 6), view, J, 3.0, F64)
        return (c * g * w).sum()

    conic, _ = orc.conic_from_J(sigma, view, J, 3.0, F64)
    gJ, gS = orc.compute_conic_backward(J, sigma, view, conic, g, dtype=F64)
    np.testing.assert_allclose(gJ, _fd(loss_J, J), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(gS, _fd(loss_S, sigma), rtol=1e-6, atol=1e-8)  # off-diagonals = sum of both entries
    gJ32, gS32 = orc.compute_conic_backward(J, sigma, view, conic, g, dtype=F32)
    np.testing.assert_allclose(gJ32, gJ, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gS32, gS, rtol=1e-4, atol=1e-5)


def test_sigma_backward(orc):  # cuda_backward_test.cpp:411-540
    rng = np.random.default_rng(4)
    q, s, g = rng.normal(size=(3, 4)), rng.normal(size=(3, 3)) * 0.3, rng.normal(size=(3, 6))
    # sigma_grad stores the SUM of both symmetric entries for off-diagonals (hazard 6): loss = sum(sigma6 * g)
    dq, ds = orc.compute_sigma_backward(q, s, g, dtype=F64)
    np.testing.assert_allclose(dq, _fd(lambda x: (orc.compute_sigma(x.reshape(-1, 4), s, F64) * g).sum(), q), rtol=2e-5,
                               atol=1e-6)
    np.testing.assert_allclose(ds, _fd(lambda x: (orc.compute_sigma(q, x.reshape(-1, 3), F64) * g).sum(), s), rtol=1e-6,
                               atol=1e-8)


@pytest.mark.parametrize("l_max", [0, 1, 2, 3])
def test_spherical_harmonics_backward(orc, l_max):  # cuda_backward_test.cpp:543-675 (only d/dxyz there)
    rng = np.random.default_rng(5 + l_max)
    n = (l_max + 1) ** 2
    xyz = rng.normal(size=(4, 3)) * 2
    campos = [0.3, -0.2, 0.1]
    band0, sh, g = rng.normal(size=(4, 3)), rng.normal(size=(4, n - 1, 3)), rng.normal(size=(4, 3))
    shg, b0g, xg = orc.precompute_spherical_harmonics_backward(xyz, band0, sh, campos, g, l_max, dtype=F64)
    f = lambda x, b, s: (orc.precompute_spherical_harmonics(x.reshape(-1, 3), s, b, campos, l_max, F64) * g).sum()
    np.testing.assert_allclose(xg, _fd(lambda x: f(x, band0, sh), xyz), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b0g, _fd(lambda b: f(xyz, b.reshape(-1, 3), sh), band0), rtol=1e-6, atol=1e-8)
    if n > 1:
        np.testing.assert_allclose(shg, _fd(lambda s: f(xyz, band0, s.reshape(4, n - 1, 3)), sh), rtol=1e-6, atol=1e-8)


def test_render_backward_reference_case(orc):  # cuda_backward_test.cpp:678-896
    W = H = 16
    uv = np.array([4.5, 4.5, 8.5, 8.5, 12.5, 12.5])
    opacity = np.array([10.0, 10.0, 10.0])
    conic = np.array([2.0, 0.1, 2.0] * 3)
    rgb = np.array([0.5, 0.2, 0.2, 0.2, 0.2, 0.5, 0.2, 0.5, 0.2])
    bg = 0.5
    gi = np.full((H, W, 3), 1e-3)
    srt, rng_ = [0, 1, 2], [0, 3]

    def loss(u=uv, o=opacity, c=conic, r=rgb):
        return (orc.render_image(u, o, c, r, bg, srt, rng_, W, H, F64)[2] * gi).sum()

    n, T, _ = orc.render_image(uv, opacity, conic, rgb, bg, srt, rng_, W, H, F64)
    g_rgb, g_op, g_uv, g_con = orc.render_image_backward(uv, opacity, conic, rgb, bg, srt, rng_, n, T, gi, W, H, F64)
    h = 1e-6
    # grad_uv carries the extra 0.5*W / 0.5*H (cuda_backward_test.cpp:838)
    np.testing.assert_allclose(g_uv.ravel(), _fd(lambda x: loss(u=x), uv, h) * np.tile([0.5 * W, 0.5 * H], 3), atol=1e-6)
    np.testing.assert_allclose(g_con.ravel(), _fd(lambda x: loss(c=x), conic, h), atol=1e-7)
    np.testing.assert_allclose(g_rgb.ravel(), _fd(lambda x: loss(r=x), rgb, h), atol=1e-8)
    # the opacity gradient ignores the 0.99 clamp (render_backward.cu:154): with logit 10 the clamp is active on the
    # centre pixels, so the reference contract is only its own tolerance of 1e-3
    np.testing.assert_allclose(g_op, _fd(lambda x: loss(o=x), opacity, h), atol=1e-3)
    f32 = orc.render_image_backward(uv, opacity, conic, rgb, bg, srt, rng_, n, T.astype(F32), gi, W, H, F32)
    for a, b in zip(f32, (g_rgb, g_op, g_uv, g_con)):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-6)


def test_render_backward_unclamped_matches_fd(orc):
    """Same contract on a scene without active clamps: all four gradients match finite differences."""
    rng = np.random.default_rng(7)
    W, H, k = 32, 16, 6
    uv = rng.uniform([2, 2], [30, 14], size=(k, 2))
    opacity = rng.uniform(-1, 1.5, size=k)
    conic = np.stack([rng.uniform(0.05, 0.3, k), rng.uniform(-0.03, 0.03, k), rng.uniform(0.05, 0.3, k)], 1)
    rgb = rng.uniform(0, 1, size=(k, 3))
    bg = 0.3
    gi = rng.normal(size=(H, W, 3))
    z = rng.uniform(1, 5, k)
    radius = np.tile([40.0, 40.0, 0.0, 1.0], (k, 1))
    srt, rng_, _ = orc.get_sorted_gaussian_list(uv, np.stack([z * 0, z * 0, z], 1), radius, 2, 1, F64)

    def loss(u=uv, o=opacity, c=conic, r=rgb):
        return (orc.render_image(u, o, c, r, bg, srt, rng_, W, H, F64)[2] * gi).sum()

    n, T, _ = orc.render_image(uv, opacity, conic, rgb, bg, srt, rng_, W, H, F64)
    g_rgb, g_op, g_uv, g_con = orc.render_image_backward(uv, opacity, conic, rgb, bg, srt, rng_, n, T, gi, W, H, F64)
    # the 1/255 floor makes the loss piecewise smooth; use a step small enough to stay inside one piece
    h = 1e-7
    np.testing.assert_allclose(g_rgb, _fd(lambda x: loss(r=x.reshape(k, 3)), rgb, h), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g_op, _fd(lambda x: loss(o=x), opacity, h), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g_con, _fd(lambda x: loss(c=x.reshape(k, 3)), conic, h), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g_uv, _fd(lambda x: loss(u=x.reshape(k, 2)), uv, h) * [0.5 * W, 0.5 * H], rtol=1e-4,
                               atol=1e-4)
