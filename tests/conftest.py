import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def perf_check(ok, message):
    """A clock-dependent expectation inside the parity suite.  Boxes differ by a few percent and one throttled box ran
    the host path 3.5x slow: a slow box must not turn `pytest -x` red before the parity rows are
    reached.  So by default a missed bar is a warning in the test report; with GSPLAT_PERF_ASSERT=1
    (what tools/experiments/*_final.sh set on a box of their own) it is an assertion."""
    if ok:
        return
    if os.environ.get("GSPLAT_PERF_ASSERT") == "1":
        raise AssertionError(message)
    import warnings
    warnings.warn("performance bar missed (GSPLAT_PERF_ASSERT=1 makes this fail): " + message)


def pkg(name=""):
    """The package directory starts with a digit, so it is imported through importlib."""
    return importlib.import_module("3dgs_amd" + ("." + name if name else ""))


@pytest.fixture(scope="session")
def scene():
    return pkg("scene")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle  # test infrastructure: the CPU parity oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    """torch + the HIP library; fails loudly (no CPU fallback) when either is missing."""
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    lib = pkg("_lib").load()
    assert lib.gsplat_abi_version() == pkg("_lib").ABI_VERSION
    return torch


@pytest.fixture(scope="session")
def config3_case(scene, orc):
    """BASELINE configs[2] (1e6 gaussians, 1920x1080, SH 3, view 0) with the oracle's forward and backward, computed once
    per session (about 25 s on the box's 16 host threads) and shared by every full-size parity test."""
    N, W, H, L, _ = scene.WORKLOADS["config3"]
    c = scene.CONFIG
    params = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H, 0)
    gi = scene.make_grad_image(W, H)
    ref = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=16)
    bref = orc.backward_pass(ref, cam, gi, c["bg"], L, threads=16)
    return dict(N=N, W=W, H=H, L=L, params=params, cam=cam, gi=gi, ref=ref, bref=bref)


# ---------------------------------------------------------------- comparison helpers (tolerances live here)
PIXEL_L1_TOL = 1e-4      # north_star: rendered pixels within 1e-4 per-pixel L1
GRAD_REL_TOL = 1e-3      # north_star: gradients within 1e-3 relative
# The alpha > 1/255 and T < 1e-4 tests are step functions of float expressions: two correct implementations that round
# an exponent differently may flip one on a (pixel, gaussian) pair whose value sits on the threshold (the HIP loop
# evaluates 2^(q + log2 sigma) where the reference evaluates sigma * __expf(p)).  Measured against the oracle: 1 pixel of
# 2 073 600 at 1920x1080 (1.7e-3), 1 of 640 000 at 800x800 (2.8e-4), 0-1 pixel on the small scenes (max 1.4e-4).
# r05: the bar is an ABSOLUTE count at what is measured plus one or two -- at most 3 pixels above 1e-4 at 1920x1080 and
# larger, at most 2 on anything smaller (r04 allowed 1e-5 * P + 2 = 22 pixels at 1080p) -- and where the oracle's lists
# are at hand (tests/test_fused_gpu.py::_tight_bookkeeping / _full_size_bookkeeping) every such pixel must additionally
# be explained in float64 by a borderline alpha / T on its tile's list (parity_tools.explain asserts it).
MEAN_L1_TOL = 1e-6       # mean per-pixel L1 (measured: 1.0e-7)
FLIP_MAX = 2e-2          # what one flipped decision can be worth: alpha * |colour - behind| summed over three channels


def max_pixels_above_tol(P):
    """How many pixels of a P-pixel image may differ from the oracle by more than PIXEL_L1_TOL (see above)."""
    return 3 if P >= 1920 * 1080 else 2


def max_stop_index_mismatches(P):
    """Pixels that may stop one splat earlier / later: 1e-5 of the image, at least 2 (measured: 6 of 2 073 600 at
    1920x1080, 0-1 elsewhere; r04 allowed 1e-4 * P + 2 = 209)."""
    return max(2, int(np.ceil(1e-5 * P)))


def assert_image_close(got, ref, what="image"):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    per_pixel_l1 = err.reshape(-1, got.shape[-1]).sum(1) if got.ndim == 3 else err.reshape(-1)
    P = per_pixel_l1.size
    assert per_pixel_l1.mean() < MEAN_L1_TOL, f"{what}: mean per-pixel L1 {per_pixel_l1.mean():.3e}"
    above = int((per_pixel_l1 > PIXEL_L1_TOL).sum())
    assert above <= max_pixels_above_tol(P), f"{what}: {above} of {P} pixels differ by more than {PIXEL_L1_TOL}"
    assert per_pixel_l1.max() < FLIP_MAX, f"{what}: max per-pixel L1 {per_pixel_l1.max():.3e}"


def assert_stop_indices_close(got, ref, what="splats_per_pixel"):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    bad = int((got != ref).sum())
    assert bad <= max_stop_index_mismatches(got.size), f"{what}: {bad} of {got.size} stop indices differ"


def assert_grad_close(got, ref, what="grad", rel=GRAD_REL_TOL):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if ref.size == 0:
        return
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    scale = np.abs(ref).mean() + 1e-30
    denom = np.sqrt((ref ** 2).sum()) + 1e-30
    l2 = np.sqrt(((got - ref) ** 2).sum()) / denom
    assert l2 < rel, f"{what}: relative L2 error {l2:.3e}"
    tol = rel * np.abs(ref) + rel * scale
    bad = (np.abs(got - ref) > tol).mean()
    assert bad <= 5e-4, f"{what}: {bad:.2e} of elements outside {rel} relative (+{rel}*mean|ref|)"
