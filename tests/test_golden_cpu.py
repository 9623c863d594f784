"""CPU: the oracle must keep reproducing the committed golden fixtures (guards against oracle or generator drift;
the same fixtures are what the HIP path is compared with on the GPU box)."""
import os

import numpy as np
import pytest

from conftest import ROOT, assert_grad_close, assert_image_close

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_oracle_reproduces_golden(scene, orc, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = dict(np.load(os.path.join(GOLD, name + ".npz")))
    params, out = mg.run(name, int(gold["view_index"]))
    assert out["checksum"] == str(gold["checksum"]), "scene generator changed: regenerate the fixtures"
    if name == "tiny":
        for k, v in params.items():
            assert (gold["in_" + k] == v).all()
    assert (out["sorted"] == gold["sorted"]).all() and (out["ranges"] == gold["ranges"]).all()
    assert (out["n"] == gold["n"]).all()
    assert_image_close(out["image"], gold["image"])
    for k in ("xyz", "band0", "sh", "opacity", "scale", "quaternion"):
        assert_grad_close(out["grad_" + k], gold["grad_" + k], k, rel=1e-6)


def test_threaded_oracle_is_deterministic(scene, orc):
    N, W, H, L, _ = scene.WORKLOADS["small"]
    p, cam = scene.make_gaussians(N, W, H, L), scene.make_camera(W, H)
    a = orc.rasterize(p, cam, 0.3, 3.0, 100, 0.5, L, threads=1)
    b = orc.rasterize(p, cam, 0.3, 3.0, 100, 0.5, L, threads=4)
    assert (a["image"] == b["image"]).all() and (a["n"] == b["n"]).all()
    gi = scene.make_grad_image(W, H)
    ga = orc.backward_pass(a, cam, gi, 0.5, L, threads=1)
    gb = orc.backward_pass(b, cam, gi, 0.5, L, threads=4)
    for k in ga:
        assert (ga[k] == gb[k]).all(), k
