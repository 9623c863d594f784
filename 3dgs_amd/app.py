"""Disk -> training -> PLY: the host program of the reference (src/main.cpp:10-98) on top of the C ABI.

    python train.py <path_to_config_file.yaml> <path_to_root_directory>

parseConfig -> ReadCamerasBinary / ReadImagesBinary / ReadPoints3DBinary under <root>/<dataset_path>/sparse/0 ->
Gaussians::Initialize (gsplat_initialize_gaussians, on the GPU) -> Trainer(config, gaussians, images, cameras) ->
test_train_split -> train (evaluation every 3000 iterations, cuda/trainer.cu:1388) -> save_to_ply("gaussians.ply").

MI355X-first differences from the reference's host side, none of which changes what is computed:
  * the reference streams one ground-truth image per iteration through a loader thread, OpenCV and two pinned
    buffers (cuda/trainer.cu:145-199); here every training image is decoded once (PIL) and kept in HBM as float32
    (185 views of 1297x840 are 2.4 GB of 288 GB), so an iteration never waits for the host;
  * under torch.distributed (WORLD_SIZE > 1, one process per GPU) the loop is view-sharded (3dgs_amd/trainer.py).
There is no CPU fallback: without a GPU and the HIP library this exits with an error.
"""
import math
import os
import sys
import time

import numpy as np


def camera_from_colmap(cam, img):
    """Camera dict for the rasterizer from a COLMAP camera + image, exactly as the reference's loop builds its
    matrices (cuda/trainer.cu:1299-1331): focal = params[0], params[1]; proj from znear 0.01 / zfar 100 and the
    field of view; view = [R|t] row-major; campos = -R^T t (Image::CamPos)."""
    from . import dataset
    f32 = np.float32
    W, H = int(cam["width"]), int(cam["height"])
    fx, fy = float(cam["params"][0]), float(cam["params"][1])
    znear, zfar = f32(0.01), f32(100.0)
    fov_x = f32(2.0 * math.atan(W / (2.0 * fx)))
    fov_y = f32(2.0 * math.atan(H / (2.0 * fy)))
    top = np.tan(fov_y / f32(2.0)).astype(f32) * znear
    right = np.tan(fov_x / f32(2.0)).astype(f32) * znear
    bottom, left = -top, -right
    proj = np.zeros(16, f32)
    proj[0] = f32(2.0) * znear / (right - left)
    proj[5] = f32(2.0) * znear / (top - bottom)
    proj[2] = (right + left) / (right - left)
    proj[6] = (top + bottom) / (top - bottom)
    proj[14] = 1.0
    proj[10] = zfar / (zfar - znear)
    proj[11] = -(zfar * znear) / (zfar - znear)
    R = dataset.qvec_to_rotmat(img["qvec"])
    t = np.asarray(img["tvec"], np.float64)
    view = np.zeros(16, f32)
    view[[0, 1, 2, 4, 5, 6, 8, 9, 10]] = R.astype(f32).reshape(-1)
    view[[3, 7, 11]] = t.astype(f32)
    view[15] = 1.0
    campos = dataset.camera_position(img["qvec"], img["tvec"]).astype(f32)
    return dict(width=W, height=H, fx=float(f32(fx)), fy=float(f32(fy)), view=view, proj=proj, campos=campos,
                image_id=int(img["id"]), name=img["name"])


def test_train_split(images, split):
    """TrainerImpl::test_train_split (cuda/trainer.cu:203-231): images sorted by name; every split-th one is ALSO a
    test image -- the reference keeps every image in the training list."""
    ordered = sorted(images.values(), key=lambda im: im["name"])
    if split <= 0:
        return ordered, []
    return ordered, [im for i, im in enumerate(ordered) if i % split == 0]


def decode_image(path, width, height, device):
    """8-bit RGB file -> float32 [H,W,3] in [0,1] on the device (the reference: cv::imread + BGR2RGB + convertTo 1/255)."""
    import torch
    from PIL import Image
    with Image.open(path) as im:
        a = np.array(im.convert("RGB"))  # a writable copy: torch.from_numpy refuses to share a read-only buffer silently
    if a.shape[0] != height or a.shape[1] != width:
        raise RuntimeError(f"{path}: {a.shape[1]}x{a.shape[0]} pixels, the camera says {width}x{height}")
    return torch.from_numpy(a).to(device).to(torch.float32).mul_(1.0 / 255.0)


def load_scene(config, root_dir, device="cuda", log=print):
    """Steps 2-3 of main.cpp: the three COLMAP files, the initial gaussians, the (camera, image) views."""
    import torch
    from . import dataset, ops, raster
    base = os.path.join(root_dir, config["dataset_path"])
    sparse = os.path.join(base, "sparse", "0")
    ds = int(config["downsample_factor"])
    cameras = dataset.ReadCamerasBinary(os.path.join(sparse, "cameras.bin"), ds)
    log(f"Successfully read {len(cameras)} cameras.")
    images = dataset.ReadImagesBinary(os.path.join(sparse, "images.bin"), base + "/", ds)
    log(f"Successfully read {len(images)} images.")
    _, xyz, rgb = dataset.ReadPoints3DArrays(os.path.join(sparse, "points3D.bin"))
    log(f"Successfully read {len(xyz)} 3D points.")
    t0 = time.perf_counter()
    params = ops.initialize_gaussians(torch.from_numpy(xyz).to(device), torch.from_numpy(rgb).to(device))
    torch.cuda.synchronize()
    log(f"Successfully initialized {len(xyz)} Gaussians. ({(time.perf_counter() - t0) * 1e3:.1f} ms on the GPU)")
    train_images, test_images = test_train_split(images, int(config["test_split_ratio"]))
    t0 = time.perf_counter()
    views, by_id = [], {}
    for im in train_images:
        cam = camera_from_colmap(cameras[im["camera_id"]], im)
        dc = raster.device_camera(cam, device)
        gt = decode_image(im["name"], cam["width"], cam["height"], device)
        by_id[im["id"]] = (dc, gt)
        views.append((dc, gt))
    test_views = [by_id[im["id"]] for im in test_images]
    log(f"Decoded {len(views)} training images ({len(test_views)} also used for evaluation) into HBM in "
        f"{time.perf_counter() - t0:.1f} s: {sum(v[1].numel() for v in views) * 4 / 1e9:.2f} GB")
    extent = 1.1 * dataset.computeMaxDiagonal(images)  # cuda/trainer.cu:1275
    return params, views, test_views, extent


def save_render(image, path, pool=None):
    """rendered_image_<iter>.png (cuda/trainer.cu:1364-1385).  The 8-bit conversion runs on the GPU and only 3 bytes per
    pixel cross PCIe; with `pool` (a one-thread executor) the PNG is encoded beside the training loop instead of in it."""
    import torch
    from PIL import Image
    a = (image.detach().clamp(0.0, 1.0) * 255.0).to(torch.uint8).to("cpu").numpy()
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)

    def encode():
        Image.fromarray(a, "RGB").save(path, compress_level=1)

    return pool.submit(encode) if pool is not None else encode()



def main(argv=None, log=print):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) != 3:
        print(f"Usage: {argv[0]} <path_to_config_file.yaml> <path_to_root_directory>", file=sys.stderr)
        return 1
    import torch
    from . import _lib, dataset, dist as gdist
    from .trainer import Trainer
    if not torch.cuda.is_available():
        print("error: no GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        return 1
    _lib.load()
    dataset.build()
    rank, world, local_rank = gdist.init_from_env()
    if rank != 0:
        log = lambda *a, **k: None
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if world > 1 else 0)
    config_path, root_dir = argv[1], argv[2]
    log(f"Attempting to read {config_path}")
    try:
        config = dataset.parseConfig(config_path)
    except dataset.HostError as e:
        print(f"Failed to load config file: {e}", file=sys.stderr)
        return 1
    log("Successfully loaded config file")
    try:
        params, views, test_views, extent = load_scene(config, root_dir, log=log)
    except (dataset.HostError, OSError, RuntimeError) as e:
        print(f"Error: {e}", file=sys.stderr)
        return 1
    seed = int(os.environ.get("GSPLAT_SEED", "0"))  # the reference seeds from std::random_device
    trainer = Trainer(params, views, config, scene_extent=extent, seed=seed)
    iters, every = int(config["num_iters"]), max(1, int(config["print_interval"]))
    stats = dict(peak_gaussians=trainer.num_gaussians, evals=[])

    def on_eval(it, psnr):
        stats["evals"].append((it, psnr))
        log(f"\n[ITER {it}] Eval PSNR: {psnr:.3f} on {len(test_views)} test images", flush=True)

    from concurrent.futures import ThreadPoolExecutor
    dump_pool, dumps = ThreadPoolExecutor(max_workers=1), []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    while done < iters:  # in chunks of print_interval: log line + the reference's rendered_image_<iter>.png
        n = min(every, iters - done)
        hist = trainer.train(n, loss_every=every, eval_every=3000, eval_views=test_views, on_eval=on_eval)
        done += n
        stats["peak_gaussians"] = max(stats["peak_gaussians"], trainer.num_gaussians)
        loss = hist[-1][1] if hist else float("nan")
        elapsed = time.perf_counter() - t0
        log(f"iter {done}/{iters}  loss {loss:.4f}  gaussians {trainer.num_gaussians}  SH {trainer.l_max}  "
            f"{done / elapsed:.1f} it/s", flush=True)
        if rank == 0 and os.environ.get("GSPLAT_NO_RENDER_DUMPS") != "1":
            cam, _ = views[0]
            ctx = trainer._context_for(trainer.num_gaussians)
            img = ctx.rasterize_image(dict(trainer.params), cam, trainer.cfg, 0.0, trainer.l_max)["image"]
            dumps.append(save_render(img, os.path.join(config["output_dir"], f"rendered_image_{done}.png"), dump_pool))
    for d in dumps:
        d.result()
    dump_pool.shutdown()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if os.environ.get("GSPLAT_DEBUG_STAGES") and rank == 0 and world == 1:
        # where the GPU time of an iteration goes at the end of training: 50 more iterations with every stage of the
        # rasterizer bracketed by events (diagnostic; they move the parameters a little further)
        ctx = trainer._context_for(trainer.num_gaussians)
        ctx.set_timing(True)
        ta = time.perf_counter()
        trainer.train(50, loss_every=0)
        torch.cuda.synchronize()
        per_it = (time.perf_counter() - ta) / 50 * 1e3
        st = ctx.get_timing()
        ctx.set_timing(False)
        f = ctx.rasterize_image(dict(trainer.params), views[0][0], trainer.cfg, 0.0, trainer.l_max)
        lens = (f["ranges"][1:] - f["ranges"][:-1]).float()
        log(f"[stages] {per_it:.3f} ms per iteration at {trainer.num_gaussians} gaussians; rasterizer stages (ms): "
            + ", ".join(f"{k} {v[0]:.3f}" for k, v in st.items() if v[1])
            + f"; view 0: M {f['num_culled']}, candidate pairs {f['num_pairs']}, instances {f['num_splats']}, tile lists "
              f"mean {lens.mean().item():.0f} / max {lens.max().item():.0f}", flush=True)
        # r05: the compositing kernels' HBM roofline on THIS capture (40 / 76 bytes per list entry some pixel of
        # its tile needs + 20 bytes per pixel), S_eff averaged over 16 of the training views
        s_eff, inst = [], []
        for cam, _ in views[:: max(1, len(views) // 16)][:16]:
            fv = ctx.rasterize_image(dict(trainer.params), cam, trainer.cfg, 0.0, trainer.l_max)
            H, W = fv["n"].shape
            nty, ntx = (H + 15) // 16, (W + 15) // 16
            npad = torch.zeros(nty * 16, ntx * 16, dtype=fv["n"].dtype, device=fv["n"].device)
            npad[:H, :W] = fv["n"]
            s_eff.append(int(npad.reshape(nty, 16, ntx, 16).amax(dim=(1, 3)).sum().item()))
            inst.append(int(fv["num_splats"]))
        se, P = sum(s_eff) / len(s_eff), float(H * W)
        fw, bw = st["render_forward"][0], st["render_backward"][0]
        log(f"[roofline] {len(s_eff)} views: instances {sum(inst) / len(inst):.0f}, S_eff {se:.0f} (entries some pixel of "
            f"their tile needs); render_fwd {fw:.3f} ms = {(40 * se + 20 * P) / (fw * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s on "
            f"{(40 * se + 20 * P) / 1e6:.0f} MB, {fw * 1e9 / se:.0f} ps per needed entry; render_bwd {bw:.3f} ms = "
            f"{(76 * se + 20 * P) / (bw * 1e-3) / 8e12 * 100:.1f} % on {(76 * se + 20 * P) / 1e6:.0f} MB, "
            f"{bw * 1e9 / se:.0f} ps per needed entry", flush=True)
    train_psnr = trainer.evaluate(views[:: max(1, len(views) // 16)])
    test_psnr = trainer.evaluate(test_views) if test_views else float("nan")
    if rank == 0:
        trainer.save_to_ply("gaussians.ply")
    log(f"\ntraining done: {iters} iterations x {world} view(s) in {wall:.1f} s = {iters / wall:.1f} it/s; "
        f"gaussians {trainer.num_gaussians} (peak {stats['peak_gaussians']}), SH degree {trainer.l_max}; "
        f"PSNR train {train_psnr:.2f} dB, test {test_psnr:.2f} dB; saved gaussians.ply", flush=True)
    summary = dict(iterations=iters, world=world, wall_s=wall, it_per_s=iters / wall, gaussians=trainer.num_gaussians,
                   peak_gaussians=stats["peak_gaussians"], psnr_train=train_psnr, psnr_test=test_psnr,
                   evals=stats["evals"], views=len(views), test_views=len(test_views))
    if rank == 0 and os.environ.get("GSPLAT_SUMMARY_JSON"):
        import json
        with open(os.environ["GSPLAT_SUMMARY_JSON"], "w") as f:
            json.dump(summary, f, indent=1)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0
