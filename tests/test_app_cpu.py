"""Host logic of the disk -> training program (3dgs_amd/app.py, the reference's src/main.cpp + the camera set-up and
test/train split of cuda/trainer.cu) and of the dataset generator, without a GPU."""
import importlib.util
import math
import os

import numpy as np

from conftest import ROOT, pkg


def _generator():
    spec = importlib.util.spec_from_file_location("make_colmap_dataset", os.path.join(ROOT, "tools", "make_colmap_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_model_reads_back_through_the_colmap_readers(tmp_path):
    ds, app, gen = pkg("dataset"), pkg("app"), _generator()
    ds.build()
    rng = np.random.default_rng(1)
    poses, names = [], []
    for v in range(9):
        C = np.array([4 * math.cos(v), -1.5, 4 * math.sin(v)])
        poses.append(gen.look_at(C, np.zeros(3)))
        names.append(f"frame_{v:05d}.png")
    pts = rng.normal(size=(500, 3))
    cols = rng.integers(0, 256, (500, 3)).astype(np.uint8)
    sparse = tmp_path / "garden" / "sparse" / "0"
    gen.write_model(str(sparse), 5187, 3361, 3838.0, poses, names, pts, cols)
    cams = ds.ReadCamerasBinary(sparse / "cameras.bin", 4)
    assert cams[1]["model"] == "PINHOLE" and (cams[1]["width"], cams[1]["height"]) == (1297, 840)  # src/colmap.cpp:91-92
    assert cams[1]["params"][0] == 3838.0 / 4
    imgs = ds.ReadImagesBinary(sparse / "images.bin", str(tmp_path / "garden") + "/", 4)
    assert len(imgs) == 9 and imgs[3]["name"] == str(tmp_path / "garden") + "/images_4/frame_00002.png"
    ids, xyz, rgb = ds.ReadPoints3DArrays(sparse / "points3D.bin")
    assert np.array_equal(ids, np.arange(1, 501)) and np.array_equal(xyz, pts) and np.array_equal(rgb, cols)
    full = ds.ReadPoints3DBinary(sparse / "points3D.bin")
    assert np.array_equal(full[7]["xyz"], pts[6]) and np.array_equal(full[7]["rgb"], cols[6])
    for i, (R, t) in enumerate(poses):
        assert abs(np.linalg.det(R) - 1) < 1e-12
        assert np.allclose(ds.qvec_to_rotmat(imgs[i + 1]["qvec"]), R, atol=1e-12) and np.allclose(imgs[i + 1]["tvec"], t)
        cam = app.camera_from_colmap(cams[1], imgs[i + 1])
        # the camera looks at the origin: it projects to the image centre, in front of the camera
        p = cam["view"].reshape(4, 4) @ np.array([0, 0, 0, 1.0])
        assert p[2] > 3.9 and abs(p[0]) < 1e-5 and abs(p[1]) < 1e-5
        assert np.allclose(cam["campos"], -R.T @ t, atol=1e-5)
        # cuda/trainer.cu:1310-1318
        P = cam["proj"].reshape(4, 4)
        assert np.isclose(P[0, 0], 2 * cam["fx"] / 1297, rtol=1e-6) and np.isclose(P[1, 1], 2 * cam["fy"] / 840, rtol=1e-6)
        assert P[3, 2] == 1 and np.isclose(P[2, 2], 100 / 99.99) and np.isclose(P[2, 3], -1 / 99.99)


def test_test_train_split_follows_the_reference():
    """cuda/trainer.cu:203-231: sorted by name, every split-th image is a test image AND stays a training image."""
    app = pkg("app")
    images = {i: dict(id=i, name=f"img_{(37 * i) % 20:03d}.png") for i in range(20)}
    train, test = app.test_train_split(images, 8)
    assert [im["name"] for im in train] == sorted(im["name"] for im in images.values())
    assert [im["name"] for im in test] == ["img_000.png", "img_008.png", "img_016.png"]
    train, test = app.test_train_split(images, 0)
    assert len(train) == 20 and test == []


def test_main_usage_error(capsys):
    assert pkg("app").main(["train.py", "only-one-argument"]) == 1  # src/main.cpp:12-15
    assert "Usage:" in capsys.readouterr().err


def test_written_config_parses_with_every_key(tmp_path):
    """tools/write_config.py -> gsplat_parse_config: all 42 keys of ConfigParameters, overrides applied."""
    import subprocess, sys
    ds = pkg("dataset")
    ds.build()
    out = tmp_path / "garden.yaml"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "write_config.py"), str(out), "num_iters=1234",
                        "dataset_path=bicycle"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    c = ds.parseConfig(out)
    assert len(c) == 42 and c["num_iters"] == 1234 and c["dataset_path"] == "bicycle"
    assert c["max_gaussians"] == 4250000 and c["use_background"] is True and abs(c["uv_grad_threshold"] - 2e-4) < 1e-12
