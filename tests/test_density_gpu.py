"""GPU parity of the f4 operators (gsplat_compute_morton_codes, gsplat_clone_gaussians, gsplat_split_gaussians)
against the oracle: Morton codes and copies bit-exact, split positions within float rounding of the same draw."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
ATTRS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


def _cloud(n, nsh, seed=0):
    rng = np.random.default_rng(seed)
    return dict(xyz=rng.normal(size=(n, 3)).astype(np.float32) * 3, rgb=rng.normal(size=(n, 3)).astype(np.float32),
                opacity=rng.normal(size=n).astype(np.float32), scale=(rng.normal(size=(n, 3)) - 2).astype(np.float32),
                quaternion=rng.normal(size=(n, 4)).astype(np.float32), sh=rng.normal(size=(n, nsh * 3)).astype(np.float32))


def test_morton_codes_bit_exact(gpu, orc):
    torch, ops = gpu, pkg("ops")
    rng = np.random.default_rng(1)
    n = 100003
    xyz = (rng.random((n, 3)) * [20, 10, 20] + [-10, -5, 0]).astype(np.float32)
    xyz[:5] = [[-10, -5, 0], [10, 5, 20], [0, 0, 10], [5, 2, 5], [-5, -2, 15]]  # tests/cuda_forward_test.cpp:936-942
    xyz[5] = [-11, 6, 25]     # outside the box: negative -> 0, above -> past the 21-bit range, masked by the spread
    want = orc.compute_morton_codes(xyz, (10, 5, 20), (-10, -5, 0))
    codes = torch.zeros(n, dtype=torch.int64, device="cuda")
    ops.compute_morton_codes(n, torch.from_numpy(xyz).cuda(), 10.0, 5.0, 20.0, -10.0, -5.0, 0.0, codes)
    assert (codes.cpu().numpy().view(np.uint64) == want).all()


@pytest.mark.parametrize("nsh", [0, 3, 15])
def test_clone_and_split_match_oracle(gpu, orc, nsh):
    torch, ops = gpu, pkg("ops")
    n = 5000
    g = _cloud(n, nsh, seed=nsh)
    mask = (np.random.default_rng(9).random(n) < 0.3).astype(np.uint8)
    wid = (np.cumsum(mask) - mask).astype(np.int32)
    m = int(mask.sum())
    src = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    d_mask, d_wid = torch.from_numpy(mask).cuda(), torch.from_numpy(wid).cuda()
    for split in (False, True):
        rows = m * (2 if split else 1)
        dst = {k: torch.full((rows,) + tuple(v.shape[1:]), float("nan"), device="cuda") for k, v in src.items()}
        if split:
            ops.split_gaussians(n, 1.6, nsh, d_mask, d_wid, src, dst, seed=42)
        else:
            ops.clone_gaussians(n, nsh, d_mask, d_wid, src, dst)
        want = orc.clone_split(g, mask, nsh, split=split, scale_factor=1.6, seed=42)
        for k in ATTRS:
            got = dst[k].cpu().numpy().reshape(want[k].shape)
            if split and k in ("xyz", "scale"):
                np.testing.assert_allclose(got, want[k], rtol=2e-5, atol=2e-6, err_msg=k)
            else:
                assert (got == want[k]).all(), k


def test_reference_known_answers_and_errors(gpu):  # tests/adaptive_density_test.cpp:187-297
    torch, ops, lib_mod = gpu, pkg("ops"), pkg("_lib")
    g = dict(xyz=np.float32([[1, 2, 3], [4, 5, 6]]), rgb=np.zeros((2, 3), np.float32), opacity=np.float32([0.8, 0.7]),
             scale=np.log(np.float32([[2, 2, 2], [.1, .1, .1]])), quaternion=np.float32([[1, 0, 0, 0]] * 2),
             sh=np.zeros((2, 0), np.float32))
    src = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    mask, wid = torch.tensor([1, 0], dtype=torch.uint8).cuda(), torch.tensor([0, 1], dtype=torch.int32).cuda()
    dst = {k: torch.zeros((4,) + tuple(v.shape[1:]), device="cuda") for k, v in src.items()}
    ops.clone_gaussians(2, 0, mask, wid, src, dst)
    assert dst["xyz"][0].tolist() == [1.0, 2.0, 3.0] and dst["opacity"][0].item() == np.float32(0.8)
    ops.split_gaussians(2, 1.6, 0, mask, wid, src, dst, seed=7)
    np.testing.assert_allclose(dst["scale"][:2].cpu().numpy(), np.log(np.float32(2.0) / np.float32(1.6)), atol=1e-6)
    assert dst["opacity"][:2].tolist() == [np.float32(0.8)] * 2
    with pytest.raises(lib_mod.GsplatError):
        ops.clone_gaussians(2, 0, torch.tensor([1, 0], dtype=torch.uint8), wid, src, dst)  # host mask
