"""Deterministic synthetic scenes for parity tests and the benchmark.

Counter-based RNG: splitmix64 of (seed, stream, index) -> 24-bit uniform floats, so every
array element is a pure function of its index and the same scene can be regenerated at any
size, on any host, without shipping data.  Seed 0x3D65.

Camera model and matrices follow the reference trainer
(/root/reference/cuda/trainer.cu:1299-1331): PINHOLE, proj built from znear=0.01 /
zfar=100, view = [R|t] row-major.
"""
import math

import numpy as np

SEED = 0x3D65
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(state):
    z = state.copy()
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def _stream_base(seed, stream):
    s = np.array([np.uint64(seed) ^ (np.uint64(stream + 1) * np.uint64(0xD1B54A32D192ED03))], dtype=np.uint64)
    return _splitmix64(s)[0]


def uniform24(seed, stream, count, offset=0):
    """count floats in [0,1) with 24 random bits each (exact in float32)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = _splitmix64(_stream_base(seed, stream) + idx * _GOLDEN)
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def normal(seed, stream, count):
    u1 = uniform24(seed, stream, count)
    u2 = uniform24(seed, stream + 1000, count)
    return np.sqrt(-2.0 * np.log(u1 + 0.5 / (1 << 24))) * np.cos(2.0 * math.pi * u2)


def make_camera(width, height, view_index=0):
    """60-degree horizontal FOV pinhole; view 0 is the identity pose, view k>0 a small deterministic
    rotation + translation (used to give each rank of a view-sharded run its own training view)."""
    fx = fy = width / (2.0 * math.tan(math.radians(30.0)))
    znear, zfar = 0.01, 100.0
    fov_x = 2.0 * math.atan(width / (2.0 * fx))
    fov_y = 2.0 * math.atan(height / (2.0 * fy))
    top = math.tan(fov_y / 2.0) * znear
    right = math.tan(fov_x / 2.0) * znear
    proj = np.zeros(16, np.float32)
    proj[0] = 2.0 * znear / (2.0 * right)
    proj[5] = 2.0 * znear / (2.0 * top)
    proj[14] = 1.0
    proj[10] = zfar / (zfar - znear)
    proj[11] = -(zfar * znear) / (zfar - znear)
    if view_index == 0:
        R = np.eye(3)
        t = np.zeros(3)
    else:
        u = uniform24(SEED, 900 + view_index, 6)
        ay, ax = (u[0] - 0.5) * 0.12, (u[1] - 0.5) * 0.08
        Ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
        R = Rx @ Ry
        t = (u[2:5] - 0.5) * np.array([0.6, 0.4, 0.6])
    view = np.zeros(16, np.float32)
    view[[0, 1, 2, 4, 5, 6, 8, 9, 10]] = R.astype(np.float32).reshape(-1)
    view[[3, 7, 11]] = t.astype(np.float32)
    view[15] = 1.0
    campos = (-R.T @ t).astype(np.float32)  # Image::CamPos(), src/colmap.cpp
    return dict(width=int(width), height=int(height), fx=float(np.float32(fx)), fy=float(np.float32(fy)),
                view=view, proj=proj, campos=campos)


def make_gaussians(num, width, height, l_max, seed=SEED, splat_scale=1.0, cluster=None, opacity_range=(-2.0, 3.0)):
    """Gaussian parameters in the reference's device layout (cuda_data.cuh:11-16).  splat_scale multiplies the
    median projected size (1.5 px by default: the benchmark scene; real captures early in training are far larger).
    cluster = (fraction, centre_u, centre_v, sigma) in units of the image width / height: that fraction of the gaussians
    (chosen per index) is drawn around one image point instead of uniformly -- an object in the middle of a capture,
    whose tiles hold ten times the average list."""
    fx = width / (2.0 * math.tan(math.radians(30.0)))
    u = (uniform24(seed, 1, num) * 1.10 - 0.05) * width
    v = (uniform24(seed, 2, num) * 1.10 - 0.05) * height
    if cluster is not None:
        frac, cu, cv, sig = cluster
        inside = uniform24(seed, 4, num) < frac
        u = np.where(inside, (cu + sig * normal(seed, 5, num)) * width, u)
        v = np.where(inside, cv * height + sig * normal(seed, 6, num) * width, v)
    z = 2.0 + 10.0 * uniform24(seed, 3, num)
    xyz = np.stack([(u - width / 2.0) * z / fx, (v - height / 2.0) * z / fx, z], 1).astype(np.float32)
    s0 = splat_scale * 1.5 * 7.0 / fx
    scale = np.log(s0 * np.exp(0.35 * normal(seed, 10, 3 * num))).reshape(num, 3).astype(np.float32)
    quaternion = normal(seed, 20, 4 * num).reshape(num, 4).astype(np.float32)
    opacity = (opacity_range[0] + (opacity_range[1] - opacity_range[0]) * uniform24(seed, 30, num)).astype(np.float32)
    rgb = (-1.5 + 3.0 * uniform24(seed, 40, 3 * num)).reshape(num, 3).astype(np.float32)
    n_rest = (l_max + 1) ** 2 - 1
    sh = (0.1 * normal(seed, 50, 3 * n_rest * num)).reshape(num, n_rest, 3).astype(np.float32) if n_rest else \
        np.zeros((num, 0, 3), np.float32)
    return dict(xyz=xyz, rgb=rgb, sh=sh, opacity=opacity, scale=scale, quaternion=quaternion)


def cull_half(params, seed=SEED, fraction=0.5):
    """About half of the gaussians moved behind the camera (z -> -z), chosen per index by the counter-based generator:
    culled and visible rows interleave at random, as in a real training view, instead of the benchmark scene's M = N."""
    out = {k: v.copy() for k, v in params.items()}
    behind = uniform24(seed, 70, len(out["xyz"])) < fraction
    out["xyz"][behind, 2] *= -1.0
    return out


def morton_order(params):
    """The same gaussians re-ordered by the 63-bit Morton code of their positions -- the order the training loop keeps
    them in (TrainerImpl::sort_gaussians after every density step, cuda/trainer.cu:853-922): neighbours in memory are
    neighbours in space, so their tile instances land next to each other."""
    xyz = params["xyz"].astype(np.float64)
    lo, hi = xyz.min(0), xyz.max(0)
    q = np.clip((xyz - lo) / np.maximum(hi - lo, 1e-30) * 2097151.0, 0, 2097151).astype(np.uint64)

    def spread(v):
        v &= np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    code = (spread(q[:, 2].copy()) << np.uint64(2)) | (spread(q[:, 1].copy()) << np.uint64(1)) | spread(q[:, 0].copy())
    order = np.argsort(code, kind="stable")
    return {k: np.ascontiguousarray(v[order]) for k, v in params.items()}


def make_grad_image(width, height, seed=SEED):
    g = (uniform24(seed, 60, 3 * width * height) * 2.0 - 1.0) / (3.0 * width * height)
    return g.reshape(height, width, 3).astype(np.float32)


CONFIG = dict(near_thresh=0.3, mh_dist=3.0, cull_mask_padding=100, bg=0.5)  # config/base.yaml:9-11

WORKLOADS = {
    # name: (N, W, H, l_max, backward)
    "tiny": (200, 64, 48, 3, True),
    "small": (5000, 256, 144, 3, True),
    "config2": (100_000, 800, 800, 0, False),
    "config3": (1_000_000, 1920, 1080, 3, True),
    # not a BASELINE config: 4x the density of config3 (~1700 list entries per tile), exercises the dense-scene
    # binning route and the long-list behaviour of the compositing kernels
    "dense4m": (4_000_000, 1920, 1080, 3, True),
    # not a BASELINE config: config3 with about half of the gaussians culled, interleaved (cull_half): the
    # per-gaussian kernels' non-consecutive-row paths, which M = N never takes
    "config3_halfculled": (1_000_000, 1920, 1080, 3, True),
    # not a BASELINE config: config3's gaussians in Morton order (morton_order), the memory order of a training run
    "config3_morton": (1_000_000, 1920, 1080, 3, True),
    # not a BASELINE config: the regime of a capture early in training (the generated garden dataset of DESIGN section 9
    # after 7 000 iterations: 145 k visible gaussians, 54 candidate tiles and 12.6 instances per gaussian, tile lists of
    # 420 entries on average): few, large splats at the Mip-NeRF 360 1/4 resolution
    "bigsplats": (150_000, 1297, 840, 3, True),
    # not a BASELINE config: the density of a real capture late in training (the generated garden dataset with 1.2 M SfM
    # points, DESIGN section 9: 1.26 M gaussians at 1297x840, tile lists of ~1100 entries on average and ~10 000 in the
    # tiles of the object in the middle), in Morton order as a training run keeps them
    "garden1200k": (1_260_000, 1297, 840, 3, True),
    # not a BASELINE config: a TRAINED view's shape (make_veiled): lists of ~10 000 entries whose pixels never saturate --
    # the scene on which one tile's list used to be the duration of both compositing launches (r05: lists in segments)
    "veiled1200k": (1_260_000, 1297, 840, 3, True),
}


def make_garden_like(N, W, H, L, seed=SEED, splat_scale=1.8, cluster_fraction=0.15, cull=0.40, opacity_range=(-5.0, 0.5)):
    """The shape of a view of the generated garden capture late in training (DESIGN section 9, 1.2 M SfM points): a third
    of the gaussians outside the view, larger and more transparent splats than the benchmark scene (about six tile
    instances per visible gaussian, pixels that saturate late), an object in the middle whose tiles hold ~10 000
    entries against ~1100 on average, Morton order in memory."""
    p = make_gaussians(N, W, H, L, seed, splat_scale=splat_scale, cluster=(cluster_fraction, 0.5, 0.55, 0.05),
                       opacity_range=opacity_range)
    return morton_order(cull_half(p, seed, cull))


def make_veiled(N, W, H, L, seed=SEED, faint_fraction=0.35, faint_scale=0.4, faint_sigma=0.05, faint_opacity=(-5.5, -3.5)):
    """The shape of a TRAINED view of the generated garden capture ("on a trained capture"): the
    garden-like scene without its opaque object, plus a veil of small, faint splats over the middle of the image -- tile
    lists of ~10 000 entries there whose pixels never saturate (each splat reaches a few pixels with alpha of a few
    hundredths), so the longest list IS the longest chain of the compositing kernels.  That is what training produces and
    what neither the uniform benchmark scene nor the garden-like one (whose long lists saturate early) has."""
    n_faint = int(N * faint_fraction)
    base = cull_half(make_gaussians(N - n_faint, W, H, L, seed, splat_scale=1.8, opacity_range=(-5.0, 0.5)), seed, 0.40)
    faint = make_gaussians(n_faint, W, H, L, seed + 17, splat_scale=faint_scale, cluster=(1.0, 0.5, 0.55, faint_sigma),
                           opacity_range=faint_opacity)
    return morton_order({k: np.concatenate([base[k], faint[k]], 0) for k in base})


def make_workload_gaussians(name, seed=SEED):
    """The gaussians of a named workload, with its variant applied (half culled, Morton order, large splats)."""
    N, W, H, L, _ = WORKLOADS[name]
    if name == "garden1200k":
        return make_garden_like(N, W, H, L, seed)
    if name == "veiled1200k":
        return make_veiled(N, W, H, L, seed)
    params = make_gaussians(N, W, H, L, seed, splat_scale=5.0 if name == "bigsplats" else 1.0)
    if name == "config3_halfculled":
        params = cull_half(params, seed)
    if name == "config3_morton":
        params = morton_order(params)
    return params

