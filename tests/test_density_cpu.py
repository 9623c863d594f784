"""Oracle pins for the f4 operators: compute_morton_codes (tests/cuda_forward_test.cpp:918-1020) and clone / split
(tests/adaptive_density_test.cpp:187-297)."""
import numpy as np

MAXC = (1 << 21) - 1


def _ref_spread(n):  # the expectation the reference's own test builds (cuda_forward_test.cpp:964-972)
    n &= MAXC
    n = (n | (n << 32)) & 0x1F000000FFFF
    n = (n | (n << 16)) & 0x1F0000FF0000FF
    n = (n | (n << 8)) & 0x100F807C0F807C0F
    n = (n | (n << 4)) & 0x1084210842108421
    n = (n | (n << 2)) & 0x1249249249249249
    return n


def test_morton_codes_reference_case(orc):
    lo, hi = np.float32([-10, -5, 0]), np.float32([10, 5, 20])
    pts = np.float32([[-10, -5, 0], [10, 5, 20], [0, 0, 10], [5, 2, 5], [-5, -2, 15]])
    want = []
    for p in pts:
        norm = np.clip((p - lo) / (hi - lo), np.float32(0), np.float32(1)).astype(np.float32)
        q = [int(np.float32(v) * np.float32(MAXC)) for v in norm]
        want.append((_ref_spread(q[2]) << 2) | (_ref_spread(q[1]) << 1) | _ref_spread(q[0]))
    got = orc.compute_morton_codes(pts, hi, lo)
    assert [int(c) for c in got] == want
    assert got[0] == 0


def test_clone_and_split_reference_cases(orc):
    g = dict(xyz=np.float32([[1, 2, 3], [4, 5, 6]]), rgb=np.float32([[.1, .2, .3], [.4, .5, .6]]),
             opacity=np.float32([0.8, 0.7]), scale=np.log(np.float32([[2, 2, 2], [.1, .1, .1]])),
             quaternion=np.float32([[1, 0, 0, 0], [1, 0, 0, 0]]), sh=np.zeros((2, 0), np.float32))
    c = orc.clone_split(g, [1, 0], 0)
    assert c["xyz"].tolist() == [[1, 2, 3]] and c["opacity"].tolist() == [np.float32(0.8)]
    s = orc.clone_split(g, [1, 0], 0, split=True, scale_factor=1.6, seed=3)
    np.testing.assert_allclose(s["scale"], np.log(np.float32(2.0) / np.float32(1.6)), atol=1e-6)
    assert s["opacity"].tolist() == [np.float32(0.8)] * 2 and s["xyz"].shape == (2, 3)
    assert not np.allclose(s["xyz"][0], s["xyz"][1])  # two different draws
    assert (orc.clone_split(g, [1, 0], 0, split=True, scale_factor=1.6, seed=3)["xyz"] == s["xyz"]).all()  # reproducible


def test_split_positions_follow_the_gaussian(orc):
    """Many splits of one anisotropic, rotated gaussian: sample mean -> xyz, covariance -> R diag(exp(s))^2 R^T."""
    n = 20000
    q = np.float32([0.9, 0.1, -0.3, 0.2])
    g = dict(xyz=np.tile(np.float32([1, -2, 0.5]), (n, 1)), rgb=np.zeros((n, 3), np.float32), opacity=np.zeros(n, np.float32),
             scale=np.tile(np.log(np.float32([0.5, 0.1, 0.02])), (n, 1)), quaternion=np.tile(q, (n, 1)),
             sh=np.arange(n * 9, dtype=np.float32).reshape(n, 9))
    s = orc.clone_split(g, np.ones(n, np.uint8), 3, split=True, scale_factor=1.6, seed=11)
    assert s["xyz"].shape == (2 * n, 3) and (s["sh"][0::2] == g["sh"]).all() and (s["sh"][1::2] == g["sh"]).all()
    w, x, y, z = q / np.linalg.norm(q)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    cov = R @ np.diag([0.25, 0.01, 0.0004]) @ R.T
    d = s["xyz"].astype(np.float64) - [1, -2, 0.5]
    assert np.abs(d.mean(0)).max() < 0.01
    np.testing.assert_allclose(d.T @ d / len(d), cov, atol=0.004)
