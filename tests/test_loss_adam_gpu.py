"""GPU parity of the "next" rows f1/f2 through the C ABI against the CPU oracle."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("shape", [(16, 16), (37, 53), (128, 200), (1, 1), (5, 300), (33, 65), (17, 31), (48, 96)])
def test_fused_loss_matches_oracle(gpu, orc, shape):
    import torch
    ops = pkg("ops")
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    pred, gt = rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)
    loss_o, grad_o = orc.fused_loss(pred, gt, 0.2)
    d_grad = torch.full((H, W, 3), float("nan"), device="cuda")
    loss = ops.fused_loss(_dev(pred), _dev(gt), H, W, 0.2, d_grad)
    assert abs(loss - loss_o) <= 1e-5 * max(1.0, abs(loss_o))
    g = d_grad.cpu().numpy()
    scale = np.abs(grad_o).max()
    assert np.abs(g - grad_o).max() <= 1e-3 * scale  # north_star gradient bar: 1e-3 relative
    # non-blocking variant leaves the same gradient
    d2 = torch.empty_like(d_grad)
    assert ops.fused_loss(_dev(pred), _dev(gt), H, W, 0.2, d2, blocking=False) is None
    torch.cuda.synchronize()
    assert torch.equal(d2, d_grad)


def test_fused_loss_reference_known_answer(gpu):  # cuda_forward_test.cpp:783-915 through the C ABI
    import torch
    ops = pkg("ops")
    vp, vg = np.float32([0.5, 0.4, 0.1]), np.float32([0.6, 0.4, 0.9])
    pred = np.broadcast_to(vp, (16, 16, 3)).copy()
    gt = np.broadcast_to(vg, (16, 16, 3)).copy()
    d_grad = torch.empty(16, 16, 3, device="cuda")
    loss = ops.fused_loss(_dev(pred), _dev(gt), 16, 16, 0.2, d_grad)
    assert abs(loss - 0.2931189) < 1e-4
    np.testing.assert_allclose(d_grad.cpu().numpy()[8, 8], [-0.00113403, -0.00104167, -0.00159930], atol=1e-6)


def test_fused_loss_full_hd(gpu, orc):
    import torch
    ops = pkg("ops")
    H, W = 1080, 1920
    rng = np.random.default_rng(3)
    pred, gt = rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)
    loss_o, grad_o = orc.fused_loss(pred, gt, 0.2, threads=8)
    d_grad = torch.empty(H, W, 3, device="cuda")
    loss = ops.fused_loss(_dev(pred), _dev(gt), H, W, 0.2, d_grad)
    assert abs(loss - loss_o) <= 1e-5
    assert np.abs(d_grad.cpu().numpy() - grad_o).max() <= 1e-3 * np.abs(grad_o).max()
    psnr = ops.compute_psnr(_dev(pred), _dev(gt), H, W)
    assert abs(psnr - orc.compute_psnr(pred, gt)) < 1e-3


def test_psnr_identical_is_100(gpu):
    ops = pkg("ops")
    a = _dev(np.random.default_rng(0).random((64, 64, 3), dtype=np.float32))
    assert ops.compute_psnr(a, a, 64, 64) == 100.0


def test_adam_step_matches_oracle(gpu, orc):
    ops = pkg("ops")
    rng = np.random.default_rng(0)
    N, S = 1000, 59
    p, g = rng.random((N, S), dtype=np.float32), rng.random((N, S), dtype=np.float32) - 0.5
    g[3, 7] = np.nan
    m, v = rng.random((N, S), dtype=np.float32) * 0.1, rng.random((N, S), dtype=np.float32) * 0.01
    args = (np.float32(1e-3), np.float32(0.9), np.float32(0.999), np.float32(1e-8), np.float32(0.1), np.float32(0.001))
    po, mo, vo = orc.adam_step(p, g, m, v, *args)
    dp, dm, dv = _dev(p), _dev(m), _dev(v)
    ops.adam_step(dp, _dev(g), dm, dv, *[float(a) for a in args], N, S)
    np.testing.assert_allclose(dp.cpu().numpy(), po.reshape(N, S), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(dm.cpu().numpy(), mo.reshape(N, S), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(dv.cpu().numpy(), vo.reshape(N, S), rtol=1e-6, atol=1e-9)


def test_loss_argument_errors(gpu):
    import torch
    ops = pkg("ops")
    a = torch.zeros(4, 4, 3, device="cuda")
    with pytest.raises(Exception):
        ops.fused_loss(a, a, 0, 4, 0.2, a)
    with pytest.raises(Exception):
        ops.fused_loss(a, torch.zeros(4, 4, 3), 4, 4, 0.2, a)
