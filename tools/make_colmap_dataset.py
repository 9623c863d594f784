#!/usr/bin/env python3
"""Writes a synthetic COLMAP dataset in the layout the reference trains from (src/main.cpp:46-50):

    <root>/<name>/sparse/0/{cameras,images,points3D}.bin      COLMAP binary model, one shared PINHOLE camera
    <root>/<name>/images_<d>/frame_00000.png ...              the views at 1/d resolution (d = downsample factor)

Default = the shape of BASELINE config 4 (Mip-NeRF 360 "garden" with config/base.yaml: 185 views, 5187x3361 stored
camera, downsample 4 -> 1297x840 images, 138k SfM points).  There is no network for the real capture, so the views
are rendered -- with this repo's own rasterizer -- from a procedural ground-truth scene of small textured gaussians:
a ground disc, a table with an object in the centre, a ring of bushes; cameras orbit on a ring looking inwards.
The SfM point cloud is a noisy subsample of the ground-truth centres.  Deterministic for a given --seed.

    python tools/make_colmap_dataset.py <root> [--name garden] [--views 185] [--gt 3000000] [--points 138000]
"""
import argparse
import importlib
import math
import os
import struct
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C0 = 0.28209479177387814


def rotmat_to_qvec(R):
    """(w, x, y, z) with w >= 0 of a rotation matrix (the inverse of Image::QvecToRotMat, src/colmap.cpp)."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    q /= np.linalg.norm(q)
    return -q if q[0] < 0 else q


def look_at(C, target):
    """world -> camera rotation with COLMAP's axes (x right, y down, z forward) and t = -R C."""
    f = target - C
    f /= np.linalg.norm(f)
    up = np.array([0.0, -1.0, 0.0])  # the world's "up" is -y, as in COLMAP reconstructions
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f])
    return R, -R @ C


def texture(p, rng_phase):
    """procedural colour in [0,1]^3: a few smooth octaves plus hard-edged stripe and checker patterns -- coherent
    pixel-scale structure (edges), which is what produces the positional gradients that drive densification."""
    out = np.zeros((len(p), 3))
    for o, (freq, amp) in enumerate(((0.9, 0.22), (2.7, 0.16), (7.1, 0.10))):
        for c in range(3):
            k = rng_phase[o, c, :3] * freq
            out[:, c] += amp * np.sin(p @ k + rng_phase[o, c, 3] * 6.283)
    for o, (freq, amp) in enumerate(((23.0, 0.16), (61.0, 0.12)), start=3):
        for c in range(3):
            k = rng_phase[o, c, :3] * freq
            out[:, c] += amp * np.sign(np.sin(p @ k + rng_phase[o, c, 3] * 6.283))
    # random-coloured 3-D cells of 0.09 units (a hash of the cell index): high-contrast patches with sharp borders
    ci = np.floor(p * 11.0).astype(np.int64)
    h = (ci[:, 0] * 73856093) ^ (ci[:, 1] * 19349663) ^ (ci[:, 2] * 83492791)
    for c in range(3):
        out[:, c] += 0.30 * ((((h >> (8 * c)) & 255) / 255.0) - 0.5)
    return np.clip(0.5 + out, 0.02, 0.98)


def ground_truth(n, rng):
    """dict of gaussian parameters (device layout of cuda_data.cuh:11-16, SH degree 0) + which part each belongs to."""
    n_ground, n_table = int(n * 0.40), int(n * 0.27)
    n_bush = n - n_ground - n_table
    parts = []
    # ground disc (y = 0.9 is "down": COLMAP's y axis points down), radius 6
    r = 6.0 * np.sqrt(rng.uniform(0, 1, n_ground))
    a = rng.uniform(0, 2 * math.pi, n_ground)
    parts.append((np.c_[r * np.cos(a), 0.9 + 0.01 * rng.normal(size=n_ground), r * np.sin(a)], 0.016, 0.35))
    # table top + a vase-like object of revolution in the centre
    nt = n_table // 2
    r = 0.9 * np.sqrt(rng.uniform(0, 1, nt))
    a = rng.uniform(0, 2 * math.pi, nt)
    table = np.c_[r * np.cos(a), 0.2 + 0.004 * rng.normal(size=nt), r * np.sin(a)]
    nv = n_table - nt
    h = rng.uniform(0, 1, nv)
    rad = 0.16 + 0.12 * np.sin(h * 5.0) ** 2 + 0.05 * h
    a = rng.uniform(0, 2 * math.pi, nv)
    vase = np.c_[rad * np.cos(a), 0.2 - 0.75 * h, rad * np.sin(a)] + 0.003 * rng.normal(size=(nv, 3))
    parts.append((np.r_[table, vase], 0.007, 0.30))
    # a ring of bushes: volumetric blobs between radius 2.2 and 5
    nb = 60
    centres = np.c_[rng.uniform(2.2, 5.0, nb), rng.uniform(-0.2, 0.6, nb), rng.uniform(0, 2 * math.pi, nb)]
    which = rng.integers(0, nb, n_bush)
    cr, cy, ca = centres[which].T
    blob = rng.normal(size=(n_bush, 3)) * np.array([0.28, 0.30, 0.28])
    parts.append((np.c_[cr * np.cos(ca), cy, cr * np.sin(ca)] + blob, 0.018, 0.40))
    xyz = np.concatenate([p for p, _, _ in parts])
    sigma = np.concatenate([np.full(len(p), s) for p, s, _ in parts])
    jitter = np.concatenate([np.full(len(p), j) for p, _, j in parts])
    phase = rng.uniform(-1, 1, (6, 3, 4))
    # per-gaussian colour jitter on top of the smooth texture: pixel-scale detail, what drives densification
    rgb01 = np.clip(texture(xyz, phase) + 0.03 * rng.normal(size=(len(xyz), 3)), 0.02, 0.98)
    scale = np.log(sigma[:, None] * np.exp(jitter[:, None] * rng.normal(size=(len(xyz), 3))))
    quat = rng.normal(size=(len(xyz), 4))
    opacity = rng.uniform(1.5, 4.0, len(xyz))
    f32 = np.float32
    return dict(xyz=xyz.astype(f32), rgb=((rgb01 - 0.5) / C0).astype(f32), sh=np.zeros((len(xyz), 0, 3), f32),
                opacity=opacity.astype(f32), scale=scale.astype(f32), quaternion=quat.astype(f32)), rgb01


def write_model(sparse, full_w, full_h, focal, poses, names, pts, cols):
    os.makedirs(sparse, exist_ok=True)
    with open(os.path.join(sparse, "cameras.bin"), "wb") as f:  # COLMAP camera model 1 = PINHOLE (fx, fy, cx, cy)
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<iiQQ", 1, 1, full_w, full_h))
        f.write(struct.pack("<4d", focal, focal, full_w / 2.0, full_h / 2.0))
    with open(os.path.join(sparse, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(poses)))
        for i, ((R, t), name) in enumerate(zip(poses, names)):
            f.write(struct.pack("<i4d3di", i + 1, *rotmat_to_qvec(R), *t, 1))
            f.write(name.encode() + b"\0")
            f.write(struct.pack("<Q", 0))  # no 2D observations: the trainer never reads them
    with open(os.path.join(sparse, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(pts)))
        rec = np.zeros(len(pts), dtype=np.dtype([("id", "<u8"), ("xyz", "<f8", 3), ("rgb", "u1", 3), ("err", "<f8"),
                                                  ("track", "<u8")]))
        rec["id"] = np.arange(1, len(pts) + 1)
        rec["xyz"], rec["rgb"], rec["err"] = pts, cols, 0.5
        f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--name", default="garden")
    ap.add_argument("--views", type=int, default=185)
    ap.add_argument("--full-width", type=int, default=5187)   # Mip-NeRF 360 garden
    ap.add_argument("--full-height", type=int, default=3361)
    ap.add_argument("--focal", type=float, default=3838.0)
    ap.add_argument("--downsample", type=int, default=4)
    ap.add_argument("--gt", type=int, default=3_000_000, help="ground-truth gaussians")
    ap.add_argument("--points", type=int, default=138_000, help="SfM points written to points3D.bin")
    ap.add_argument("--seed", type=int, default=0x3D65)
    args = ap.parse_args()

    import torch
    from PIL import Image
    raster = importlib.import_module("3dgs_amd.raster")
    app = importlib.import_module("3dgs_amd.app")
    assert torch.cuda.is_available(), "the views are rendered on the GPU"
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    gt, rgb01 = ground_truth(args.gt, rng)
    d = args.downsample
    W, H = int(round(args.full_width / float(d))), int(round(args.full_height / float(d)))  # src/colmap.cpp:91-92
    poses, names = [], []
    for v in range(args.views):
        ang = 2 * math.pi * v / args.views
        radius = 4.0 + 0.5 * math.sin(3 * ang)
        height = -1.1 - 0.9 * (0.5 + 0.5 * math.sin(2 * ang + 0.7))  # above the table (y points down)
        C = np.array([radius * math.cos(ang), height, radius * math.sin(ang)])
        poses.append(look_at(C, np.array([0.0, 0.05, 0.0]) + 0.15 * rng.normal(size=3)))
        names.append(f"frame_{v:05d}.png")
    base = os.path.join(args.root, args.name)
    pick = rng.choice(len(gt["xyz"]), min(args.points, len(gt["xyz"])), replace=False)
    pts = gt["xyz"][pick].astype(np.float64) + 0.004 * rng.normal(size=(len(pick), 3))
    cols = np.clip(rgb01[pick] * 255.0 + rng.normal(size=(len(pick), 3)) * 6.0, 0, 255).astype(np.uint8)
    write_model(os.path.join(base, "sparse", "0"), args.full_width, args.full_height, args.focal, poses, names, pts, cols)

    img_dir = os.path.join(base, f"images_{d}" if d > 1 else "images")
    os.makedirs(img_dir, exist_ok=True)
    dp = raster.device_params(gt)
    ctx = raster.RasterContext(len(gt["xyz"]), W, H)
    ctx.set_render_only(True)
    cfg = dict(near_thresh=0.3, mh_dist=3.0, cull_mask_padding=100)
    cam_rec = dict(width=W, height=H, params=[args.focal / d, args.focal / d, W / 2.0, H / 2.0])
    for v, ((R, t), name) in enumerate(zip(poses, names)):
        cam = app.camera_from_colmap(cam_rec, dict(id=v + 1, qvec=rotmat_to_qvec(R), tvec=t, name=name))
        img = ctx.rasterize_image(dp, raster.device_camera(cam), cfg, 0.0, 0)["image"]
        a = (img.clamp(0.0, 1.0) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
        Image.fromarray(a, "RGB").save(os.path.join(img_dir, name), compress_level=1)
    print(f"wrote {args.views} views of {W}x{H} from {len(gt['xyz'])} ground-truth gaussians and {len(pts)} SfM points "
          f"under {base} in {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
