"""Oracle pinning for the "next" rows f1 (fused_loss / compute_psnr) and f2 (adam_step): the reference's own
known-answer tests restated on the CPU oracle (tests/cuda_forward_test.cpp:783-915, tests/optimizer_test.cpp:104-138),
plus an f64 finite-difference check of dL/dimage that does not depend on the reference's expectations."""
import numpy as np

GAUSS = np.array([0.00102838, 0.00759876, 0.03600077, 0.10936069, 0.21300553, 0.26601171,
                  0.21300553, 0.10936069, 0.03600077, 0.00759876, 0.00102838], dtype=np.float32)


def test_fused_loss_uniform_rgb_known_answer(orc):  # cuda_forward_test.cpp:783-915
    rows = cols = 16
    w = np.float32(0.2)
    vp, vg = np.float32([0.5, 0.4, 0.1]), np.float32([0.6, 0.4, 0.9])
    pred = np.broadcast_to(vp, (rows, cols, 3)).copy()
    gt = np.broadcast_to(vg, (rows, cols, 3)).copy()
    loss, grad = orc.fused_loss(pred, gt, w)
    C1 = np.float32(0.01) ** 2
    s2d = np.float32(GAUSS.sum()) ** 2
    exp_loss, exp_grad = 0.0, np.zeros(3, np.float32)
    for c in range(3):
        num, den = 2 * vp[c] * vg[c] + C1, vp[c] ** 2 + vg[c] ** 2 + C1
        exp_loss += (1 - w) * abs(vp[c] - vg[c]) + w * (1 - num / den)
        l1_dir = 1.0 if vp[c] > vg[c] else -1.0
        dssim = ((2 * vg[c]) * den - num * (2 * vp[c])) / (den * den)
        exp_grad[c] = ((1 - w) * l1_dir + w * (-dssim * s2d)) / (rows * cols * 3)
    assert abs(loss - exp_loss / 3) < 1e-4
    np.testing.assert_allclose(grad[5:11, 5:11], np.broadcast_to(exp_grad, (6, 6, 3)), atol=1e-6, rtol=0)


def test_fused_loss_gradient_matches_finite_differences_f64(orc):
    """L is piecewise smooth (|.| and clamped borders); away from pred == gt the analytic gradient must equal the
    central difference.  Border pixels are included: the forward clamps, the adjoint zero-pads (cuda/loss.cu:262-420),
    so only pixels at least 10 px from every edge are exact and those are the ones compared."""
    rng = np.random.default_rng(5)
    H, W = 27, 29
    pred, gt = rng.random((H, W, 3)), rng.random((H, W, 3))
    loss, grad = orc.fused_loss(pred, gt, 0.2, dtype=np.float64)
    eps = 1e-6
    for (y, x, c) in [(13, 14, 0), (10, 18, 2), (16, 10, 1)]:
        p, m = pred.copy(), pred.copy()
        p[y, x, c] += eps
        m[y, x, c] -= eps
        fd = (orc.fused_loss(p, gt, 0.2, dtype=np.float64)[0] - orc.fused_loss(m, gt, 0.2, dtype=np.float64)[0]) / (2 * eps)
        assert abs(fd - grad[y, x, c]) < 1e-9 + 1e-5 * abs(fd), (y, x, c, fd, grad[y, x, c])


def test_fused_loss_identical_images(orc):
    img = np.random.default_rng(1).random((20, 33, 3)).astype(np.float32)
    loss, grad = orc.fused_loss(img, img, 0.2)
    assert abs(loss) < 1e-6
    assert orc.compute_psnr(img, img) == 100.0  # cuda/loss.cu:520-522


def test_psnr_known_value(orc):
    a = np.full((8, 8, 3), 0.5, np.float32)
    b = np.full((8, 8, 3), 0.6, np.float32)
    assert abs(orc.compute_psnr(a, b) - 20.0) < 1e-3


def test_adam_step_known_answer(orc):  # optimizer_test.cpp:104-138
    rng = np.random.default_rng(0)
    N, lr, b1, b2, eps = 1024, np.float32(1e-3), np.float32(0.9), np.float32(0.999), np.float32(1e-8)
    p, g = rng.random(N, dtype=np.float32), rng.random(N, dtype=np.float32) - 0.5
    m, v = rng.random(N, dtype=np.float32) * 0.1, rng.random(N, dtype=np.float32) * 0.01
    po, mo, vo = orc.adam_step(p, g, m, v, lr, b1, b2, eps, 1 - b1, 1 - b2)
    me = b1 * m + (1 - b1) * g
    ve = b2 * v + (1 - b2) * g * g
    pe = p + (-lr * (me / (1 - b1)) / (np.sqrt(ve / (1 - b2)) + eps))
    np.testing.assert_allclose(po, pe, atol=1e-6)
    np.testing.assert_allclose(mo, me, atol=1e-6)
    np.testing.assert_allclose(vo, ve, atol=1e-6)


def test_adam_nan_gradient_is_zero(orc):  # cuda/optimizer.cu:12-14
    p, m, v = np.float32([1.0, 2.0]), np.float32([0.1, 0.2]), np.float32([0.01, 0.02])
    g = np.float32([np.nan, 0.5])
    po, mo, vo = orc.adam_step(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.001)
    assert np.isfinite(po).all() and abs(mo[0] - 0.09) < 1e-7 and abs(vo[0] - 0.00999) < 1e-7
