"""Row f3 (dataset plumbing) on the CPU: libgsplat_host.so against the reference's own fixtures and known answers.

tests/golden/colmap/*.bin are the reference's test_data files (data fixtures); the expectations restate
tests/colmap_test.cpp:7-78, tests/utils_test.cpp:8-140 and the formulas of src/colmap.cpp."""
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT, pkg

DATA = os.path.join(ROOT, "tests", "golden", "colmap")


@pytest.fixture(scope="module")
def ds():
    mod = pkg("dataset")
    mod.build()
    return mod


def test_host_library_exports_every_declared_symbol(ds):
    text = open(os.path.join(ROOT, "include", "gsplat_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(gsplat_[a-z0-9_]+)\s*\(", text)))
    lib = ds.load()
    assert set(names) == set(ds.HOST_SIGNATURES)
    for n in names:
        assert hasattr(lib, n)


def test_read_cameras_binary(ds):  # tests/colmap_test.cpp:7-29
    cams = ds.ReadCamerasBinary(os.path.join(DATA, "cameras.bin"), 1)
    assert list(cams) == [1]
    c = cams[1]
    assert (c["id"], c["model"], c["width"], c["height"]) == (1, "SIMPLE_PINHOLE", 100, 80)
    assert c["params"] == [150.5, 50.2, 40.8]
    half = ds.ReadCamerasBinary(os.path.join(DATA, "cameras.bin"), 2)[1]  # src/colmap.cpp:89-94
    assert (half["width"], half["height"]) == (50, 40) and half["params"] == [75.25, 25.1, 20.4]


def test_read_images_binary(ds):  # tests/colmap_test.cpp:32-54
    imgs = ds.ReadImagesBinary(os.path.join(DATA, "images.bin"), "root/dir/", 1)
    assert list(imgs) == [1]
    im = imgs[1]
    assert im["name"] == "root/dir/images/test.jpg" and im["camera_id"] == 1
    np.testing.assert_allclose(im["qvec"], [0.8, 0.1, 0.2, 0.3], atol=1e-9)
    np.testing.assert_allclose(im["tvec"], [5.1, 6.2, 7.3], atol=1e-9)
    assert im["xys"].shape == (2, 2) and tuple(im["xys"][0]) == (10.1, 11.2)
    assert list(im["point3D_ids"]) == [1, -1]
    assert ds.ReadImagesBinary(os.path.join(DATA, "images.bin"), "r/", 4)[1]["name"] == "r/images_4/test.jpg"


def test_read_points3d_binary(ds):  # tests/colmap_test.cpp:57-78
    pts = ds.ReadPoints3DBinary(os.path.join(DATA, "points3D.bin"))
    assert list(pts) == [1]
    p = pts[1]
    assert tuple(p["xyz"]) == (1.1, 2.2, 3.3) and list(p["rgb"]) == [10, 20, 30]
    assert abs(p["error"] - 0.01) < 1e-9
    assert list(p["image_ids"]) == [1] and list(p["point2D_idxs"]) == [0]


def test_reader_errors(ds, tmp_path):
    with pytest.raises(ds.HostError) as e:
        ds.ReadCamerasBinary(tmp_path / "missing.bin")
    assert e.value.code == -7
    cut = tmp_path / "cut.bin"
    cut.write_bytes(open(os.path.join(DATA, "images.bin"), "rb").read()[:60])
    with pytest.raises(ds.HostError) as e:
        ds.ReadImagesBinary(cut)
    assert e.value.code == -8
    fisheye = tmp_path / "fisheye.bin"  # only (SIMPLE_)PINHOLE is accepted, src/colmap.cpp:70-73
    fisheye.write_bytes(struct.pack("<QiiQQ", 1, 7, 5, 10, 10) + struct.pack("<8d", *range(8)))
    with pytest.raises(ds.HostError) as e:
        ds.ReadCamerasBinary(fisheye)
    assert e.value.code == -9


def test_multi_record_files_round_trip(ds, tmp_path):
    """Files written here in COLMAP's binary layout with several records and ragged tracks."""
    rng = np.random.default_rng(0)
    cams = tmp_path / "cameras.bin"
    cams.write_bytes(struct.pack("<Q", 2) + struct.pack("<iiQQ3d", 3, 0, 640, 480, 500.0, 320.0, 240.0) +
                     struct.pack("<iiQQ4d", 9, 1, 1280, 720, 900.0, 901.0, 640.0, 360.0))
    c = ds.ReadCamerasBinary(cams, 1)
    assert c[9]["model"] == "PINHOLE" and c[9]["params"] == [900.0, 901.0, 640.0, 360.0] and c[3]["width"] == 640
    blob, want = struct.pack("<Q", 3), {}
    for pid, tl in ((5, 0), (70000000000, 3), (8, 1)):
        xyz, rgb, err = rng.normal(size=3), rng.integers(0, 255, 3), float(rng.random())
        tr = [(int(rng.integers(1, 50)), int(rng.integers(0, 900))) for _ in range(tl)]
        blob += struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, err, tl) + b"".join(struct.pack("<ii", *t) for t in tr)
        want[pid] = (xyz, rgb, err, tr)
    f = tmp_path / "points3D.bin"
    f.write_bytes(blob)
    pts = ds.ReadPoints3DBinary(f)
    assert set(pts) == set(want)
    for pid, (xyz, rgb, err, tr) in want.items():
        assert (pts[pid]["xyz"] == xyz).all() and (pts[pid]["rgb"] == rgb).all() and pts[pid]["error"] == err
        assert list(zip(pts[pid]["image_ids"], pts[pid]["point2D_idxs"])) == tr


def test_camera_geometry(ds):  # Image::QvecToRotMat / CamPos / computeMaxDiagonal, src/colmap.cpp:30-39,198-236
    rng = np.random.default_rng(1)
    imgs = {}
    for i in range(7):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        imgs[i] = dict(qvec=q, tvec=rng.normal(size=3) * 3)
    centres = []
    for im in imgs.values():
        w, x, y, z = im["qvec"]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(ds.qvec_to_rotmat(im["qvec"]), R, atol=1e-14)
        centres.append(-R.T @ im["tvec"])
        np.testing.assert_allclose(ds.camera_position(im["qvec"], im["tvec"]), centres[-1], atol=1e-13)
    centres = np.array(centres)
    want = np.linalg.norm(centres - centres.mean(0), axis=1).max()
    assert abs(ds.computeMaxDiagonal(imgs) - want) < 1e-12
    assert ds.computeMaxDiagonal({}) == 0.0


CONFIG_TEXT = """dataset_path: "/data/nerf_synthetic/lego"
output_dir: "/output/lego"
downsample_factor: 2
print_interval: 100
num_iters: 30000
ssim_frac: 0.2   # a comment
test_eval_interval: 1000
test_split_ratio: 8
initial_opacity: 0.1
initial_scale_num_neighbors: 3
initial_scale_factor: 1.0
max_initial_scale: 1.0
near_thresh: 0.01
mh_dist: 1000.0
cull_mask_padding: 1
base_lr: 1e-3
xyz_lr_multiplier_init: 1.0
xyz_lr_multiplier_final: 1.0
quat_lr_multiplier: 1.0
scale_lr_multiplier: 1.0
opacity_lr_multiplier: 25
rgb_lr_multiplier: 1.0
sh_lr_multiplier: 1.0
use_background: true
use_background_end: 15000
reset_opacity_interval: 3000
reset_opacity_value: 0.01
reset_opacity_start: 4000
reset_opacity_end: 15000
use_sh_precompute: true
max_sh_band: 2
add_sh_band_interval: 1000
use_split: true
use_clone: false
use_delete: true
adaptive_control_start: 500
adaptive_control_end: 20000
adaptive_control_interval: 100
max_gaussians: 1000000
delete_opacity_threshold: 0.005
uv_grad_threshold: 0.0002
split_scale_factor: 1.5
"""


def test_parse_config(ds, tmp_path):  # tests/utils_test.cpp:8-117
    f = tmp_path / "valid_config.yaml"
    f.write_text(CONFIG_TEXT)
    c = ds.parseConfig(f)
    assert c["dataset_path"] == "/data/nerf_synthetic/lego" and c["downsample_factor"] == 2
    assert abs(c["ssim_frac"] - 0.2) < 1e-12 and c["use_background"] is True and c["use_clone"] is False
    assert c["max_sh_band"] == 2 and c["split_scale_factor"] == 1.5 and c["base_lr"] == 1e-3
    assert c["opacity_lr_multiplier"] == 25.0
    with pytest.raises(ds.HostError) as e:
        ds.parseConfig(tmp_path / "non_existent_file.yaml")
    assert e.value.code == -7
    g = tmp_path / "missing_key.yaml"
    g.write_text('output_dir: "/output/lego"\n')
    with pytest.raises(ds.HostError) as e:
        ds.parseConfig(g)
    assert e.value.code == -8 and "dataset_path" in str(e.value)


def test_save_ply(ds, tmp_path):  # tests/utils_test.cpp:119-152 + a full read-back
    rng = np.random.default_rng(2)
    n, k = 5, 9
    xyz, rgb, sh = rng.normal(size=(n, 3)), rng.normal(size=(n, 3)), rng.normal(size=(n, k))
    op, sc, q = rng.normal(size=n), rng.normal(size=(n, 3)), rng.normal(size=(n, 4))
    f = tmp_path / "test_output.ply"
    ds.save_ply(f, xyz, rgb, op, sc, q, sh)
    raw = f.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == f"element vertex {n}"
    props = [l.split()[-1] for l in lines[3:]]
    assert props == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(k)] +
                     ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    rows = np.frombuffer(body, "<f4").reshape(n, len(props))
    f32 = lambda a: np.asarray(a, np.float32)
    assert (rows[:, 0:3] == f32(xyz)).all() and (rows[:, 3:6] == 0).all() and (rows[:, 6:9] == f32(rgb)).all()
    assert (rows[:, 9:9 + k] == f32(sh)).all() and (rows[:, 9 + k] == f32(op)).all()
    assert (rows[:, 10 + k:13 + k] == f32(sc)).all()
    assert (rows[:, 13 + k:17 + k] == f32(q)[:, [1, 2, 3, 0]]).all()  # file order x y z w, src/utils.cpp:165-168
    ds.save_ply(tmp_path / "no_sh.ply", xyz, rgb, op, sc, q)  # without SH: 17 floats per vertex
    assert len((tmp_path / "no_sh.ply").read_bytes().split(b"end_header\n", 1)[1]) == n * 17 * 4


# ---------------------------------------------------------------------------- Gaussians::Initialize (oracle pins)
def _clouds():
    rng = np.random.default_rng(3)
    uniform = rng.random((700, 3)) * [4.0, 2.0, 1.0]
    clustered = np.concatenate([rng.normal(c, s, (200, 3)) for c, s in (((0, 0, 0), 0.05), ((3, 1, 0), 0.5), ((-2, 4, 1), 0.01))])
    dup = np.concatenate([uniform[:50], uniform[:50], uniform[50:120]])
    return dict(uniform=uniform, clustered=clustered, duplicates=dup)


def test_oracle_knn_matches_kdtree(orc):
    """The oracle's brute-force statistic against an independent exact kd-tree (scipy), the role nanoflann plays in
    the reference (src/gaussian.cpp:57-79)."""
    from scipy.spatial import cKDTree
    for name, pts in _clouds().items():
        d, _ = cKDTree(pts).query(pts, k=4)
        want = d[:, 1:].mean(1).astype(np.float32)
        np.testing.assert_allclose(orc.knn_mean_distance(pts, 3), want, rtol=1e-6, atol=1e-12, err_msg=name)


def test_oracle_knn_known_answers(orc):
    line = np.array([[0, 0, 0], [1, 0, 0], [3, 0, 0], [7, 0, 0], [15, 0, 0]], float)
    np.testing.assert_allclose(orc.knn_mean_distance(line, 3), [(1 + 3 + 7) / 3, (1 + 2 + 6) / 3, (2 + 3 + 4) / 3,
                                                                (4 + 6 + 7) / 3, (8 + 12 + 14) / 3], rtol=1e-7)
    assert orc.knn_mean_distance(line[:1], 3)[0] == np.float32(0.01)        # no neighbour: src/gaussian.cpp:91
    np.testing.assert_allclose(orc.knn_mean_distance(line[:3], 3), [2.0, 1.5, 2.5])  # fewer than 3 neighbours


def test_oracle_initialize_attributes(orc):  # src/gaussian.cpp:93-101
    pts = np.array([[0, 0, 0], [0, 3, 4], [1, 0, 0], [0, 0, 2]], float)
    col = np.array([[255, 0, 128], [10, 20, 30], [0, 0, 0], [255, 255, 255]], np.uint8)
    g = orc.initialize_gaussians(pts, col)
    C0 = np.float32(0.28209479177387814)
    np.testing.assert_allclose(g["rgb"], (col.astype(np.float32) / np.float32(255) - np.float32(0.5)) / C0, rtol=1e-6)
    np.testing.assert_allclose(g["opacity"], np.log(0.2) - np.log(0.8), rtol=1e-6)
    assert (g["quaternion"] == [1, 0, 0, 0]).all() and (g["xyz"] == pts.astype(np.float32)).all()
    np.testing.assert_allclose(g["scale"][0], np.log((1 + 2 + 5) / 3), rtol=1e-6)
    assert (g["scale"][:, 0] == g["scale"][:, 1]).all() and (g["scale"][:, 1] == g["scale"][:, 2]).all()


def test_kdtree_knn_equals_brute_force(orc):
    """The CPU baseline of Gaussians::Initialize (kd-tree, leaf size 10, as the reference's nanoflann index) must
    return the brute-force answer on clustered, planar, duplicated and tiny clouds."""
    rng = np.random.default_rng(5)
    clouds = [rng.normal(size=(3000, 3)), np.concatenate([rng.normal(size=(500, 3)) * 0.01, rng.uniform(-5, 5, (700, 3))]),
              np.c_[rng.uniform(0, 1, (800, 2)), np.zeros(800)], np.repeat(rng.normal(size=(40, 3)), 9, axis=0),
              rng.normal(size=(1, 3)), rng.normal(size=(2, 3)), rng.normal(size=(11, 3)), np.zeros((30, 3))]
    for pts in clouds:
        for k in (1, 3, 8):
            a = orc.knn_mean_distance(pts, k)
            b = orc.knn_mean_distance(pts, k, threads=3, kdtree=True)
            assert np.array_equal(a, b), (len(pts), k, np.abs(a - b).max())


def test_oracle_threads_do_not_change_results(orc, scene):
    """orc.set_threads parallelises the per-gaussian operators only (independent per gaussian): bit-identical."""
    N, W, H, L = 3000, 128, 96, 3
    params, cam, c = scene.make_gaussians(N, W, H, L), scene.make_camera(W, H, 2), scene.CONFIG
    gi = scene.make_grad_image(W, H)

    def run(threads):
        orc.set_threads(threads)
        try:
            f = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L, threads=threads)
            return f, orc.backward_pass(f, cam, gi, c["bg"], L, threads=threads)
        finally:
            orc.set_threads(1)

    (f1, b1), (f4, b4) = run(1), run(4)
    for k in ("image", "sorted", "ranges", "conic", "rgb", "radius", "sigma", "J"):
        assert np.array_equal(f1[k], f4[k]), k
    for k in ("xyz", "sh", "band0", "scale", "quaternion", "xyz_c", "J", "sigma"):
        assert np.array_equal(b1[k], b4[k]), k
