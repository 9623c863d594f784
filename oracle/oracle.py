"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front end of the CPU parity oracle.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``3dgs_amd/``) never does.

The functions mirror the reference operator surface one to one
(``/root/reference/include/gsplat_cuda/cuda_forward.cuh:26-131``,
``cuda_backward.cuh:21-123``) plus the two sequencing functions
``rasterize`` (``cuda/raster.cu:12-136``) and ``backward_pass``
(``cuda/trainer.cu:926-1015``).  ``dtype`` selects the float32 parity instantiation or the
float64 one used for finite-difference checks of the backward formulas.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liboracle.so with gcc (plain C, no reference sources)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("gsplat_oracle.c", "gsplat_oracle_impl.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_count_tile_pairs_f32.restype = ctypes.c_long
        _LIB.orc_count_tile_pairs_f64.restype = ctypes.c_long
        _LIB.orc_sorted_gaussian_list_f32.restype = ctypes.c_long
        _LIB.orc_sorted_gaussian_list_f64.restype = ctypes.c_long
        _LIB.orc_effective_instances_f32.restype = ctypes.c_long
        _LIB.orc_effective_instances_f64.restype = ctypes.c_long
        for n in ("orc_fused_loss", "orc_psnr"):
            getattr(_LIB, n + "_f32").restype = ctypes.c_float
            getattr(_LIB, n + "_f64").restype = ctypes.c_double
    return _LIB


def _suf(dtype):
    return "_f32" if np.dtype(dtype) == np.float32 else "_f64"


def _fn(name, dtype):
    return getattr(lib(), name + _suf(dtype))


def _a(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)


def _p(x):
    return x.ctypes.data_as(ctypes.c_void_p)


def _r(v, dtype):
    return ctypes.c_float(float(v)) if np.dtype(dtype) == np.float32 else ctypes.c_double(float(v))


# ----------------------------------------------------------------------------- forward ops
def compute_camera_space_points(xyz_w, view, dtype=np.float32):
    xyz_w, view = _a(xyz_w, dtype).reshape(-1, 3), _a(view, dtype).reshape(16)
    out = np.empty_like(xyz_w)
    _fn("orc_camera_space_points", dtype)(_p(xyz_w), _p(view), len(xyz_w), _p(out))
    return out


def project_to_screen(xyz, proj, width, height, dtype=np.float32):
    xyz, proj = _a(xyz, dtype).reshape(-1, 3), _a(proj, dtype).reshape(16)
    uv = np.empty((len(xyz), 2), dtype)
    _fn("orc_project_to_screen", dtype)(_p(xyz), _p(proj), len(xyz), int(width), int(height), _p(uv))
    return uv


def cull_gaussians(uv, xyz, near_thresh, padding, width, height, dtype=np.float32):
    uv, xyz = _a(uv, dtype).reshape(-1, 2), _a(xyz, dtype).reshape(-1, 3)
    mask = np.empty(len(uv), np.uint8)
    _fn("orc_cull_gaussians", dtype)(_p(uv), _p(xyz), len(uv), _r(near_thresh, dtype), int(padding), int(width),
                                     int(height), _p(mask))
    return mask.astype(bool)


def compute_sigma(quaternion, scale, dtype=np.float32):
    q, s = _a(quaternion, dtype).reshape(-1, 4), _a(scale, dtype).reshape(-1, 3)
    sigma = np.empty((len(q), 6), dtype)
    _fn("orc_compute_sigma", dtype)(_p(q), _p(s), len(q), _p(sigma))
    return sigma


def compute_conic(xyz, view, sigma, focal_x, focal_y, tan_fovx, tan_fovy, mh_dist, dtype=np.float32):
    xyz, view, sigma = _a(xyz, dtype).reshape(-1, 3), _a(view, dtype).reshape(16), _a(sigma, dtype).reshape(-1, 6)
    n = len(xyz)
    J, conic, radius = np.empty((n, 6), dtype), np.empty((n, 3), dtype), np.empty((n, 4), dtype)
    _fn("orc_compute_conic", dtype)(_p(xyz), _p(view), _p(sigma), _r(focal_x, dtype), _r(focal_y, dtype),
                                    _r(tan_fovx, dtype), _r(tan_fovy, dtype), _r(mh_dist, dtype), n, _p(J), _p(conic),
                                    _p(radius))
    return J, conic, radius


def conic_from_J(sigma, view, J, mh_dist, dtype=np.float32):
    """The second kernel of compute_conic alone (used by the f64 finite-difference tests)."""
    sigma, view, J = _a(sigma, dtype).reshape(-1, 6), _a(view, dtype).reshape(16), _a(J, dtype).reshape(-1, 6)
    n = len(sigma)
    conic, radius = np.empty((n, 3), dtype), np.empty((n, 4), dtype)
    _fn("orc_conic_from_J", dtype)(_p(sigma), _p(view), _p(J), n, _r(mh_dist, dtype), _p(conic), _p(radius))
    return conic, radius


def projection_jacobian(xyz, focal_x, focal_y, tan_fovx, tan_fovy, dtype=np.float32):
    xyz = _a(xyz, dtype).reshape(-1, 3)
    J = np.empty((len(xyz), 6), dtype)
    _fn("orc_projection_jacobian", dtype)(_p(xyz), _r(focal_x, dtype), _r(focal_y, dtype), _r(tan_fovx, dtype),
                                          _r(tan_fovy, dtype), len(xyz), _p(J))
    return J


def count_tile_pairs(uv, radius, n_tiles_x, n_tiles_y, dtype=np.float32):
    """Call 1 of get_sorted_gaussian_list (sorted == nullptr): candidate-pair count."""
    uv, radius = _a(uv, dtype).reshape(-1, 2), _a(radius, dtype).reshape(-1, 4)
    return int(_fn("orc_count_tile_pairs", dtype)(_p(uv), _p(radius), int(n_tiles_x), int(n_tiles_y), len(uv)))


def get_sorted_gaussian_list(uv, xyz, radius, n_tiles_x, n_tiles_y, dtype=np.float32):
    """Call 2: returns (sorted ids [S], ranges [T+1], candidate count)."""
    uv, xyz, radius = _a(uv, dtype).reshape(-1, 2), _a(xyz, dtype).reshape(-1, 3), _a(radius, dtype).reshape(-1, 4)
    cap = count_tile_pairs(uv, radius, n_tiles_x, n_tiles_y, dtype)
    sorted_ids = np.zeros(max(cap, 1), np.int32)
    ranges = np.zeros(n_tiles_x * n_tiles_y + 1, np.int32)
    S = int(_fn("orc_sorted_gaussian_list", dtype)(_p(uv), _p(xyz), _p(radius), int(n_tiles_x), int(n_tiles_y),
                                                   len(uv), _p(sorted_ids), _p(ranges)))
    return sorted_ids[:S].copy(), ranges, cap


def precompute_spherical_harmonics(xyz, sh, band0, campos, l_max, dtype=np.float32):
    xyz, band0 = _a(xyz, dtype).reshape(-1, 3), _a(band0, dtype).reshape(-1, 3)
    n = len(xyz)
    sh = _a(sh if sh is not None else np.zeros(1), dtype).reshape(-1)
    campos = _a(campos, dtype).reshape(3)
    rgb = np.empty((n, 3), dtype)
    _fn("orc_sh_forward", dtype)(_p(xyz), _p(sh), _p(band0), _p(campos), int(l_max), n, _p(rgb))
    return rgb


def render_image(uv, opacity, conic, rgb, bg, sorted_ids, ranges, width, height, dtype=np.float32, threads=1):
    uv, opacity = _a(uv, dtype).reshape(-1, 2), _a(opacity, dtype).reshape(-1)
    conic, rgb = _a(conic, dtype).reshape(-1, 3), _a(rgb, dtype).reshape(-1, 3)
    sorted_ids, ranges = _a(sorted_ids, np.int32), _a(ranges, np.int32)
    n = np.zeros((height, width), np.int32)
    T = np.zeros((height, width), dtype)
    image = np.zeros((height, width, 3), dtype)
    _fn("orc_render_image", dtype)(_p(uv), _p(opacity), _p(conic), _p(rgb), _r(bg, dtype), _p(sorted_ids), _p(ranges),
                                   int(width), int(height), _p(n), _p(T), _p(image), int(threads))
    return n, T, image


def effective_instances(n, dtype=np.float32):
    n = _a(n, np.int32)
    return int(_fn("orc_effective_instances", dtype)(_p(n), n.shape[1], n.shape[0]))


# ---------------------------------------------------------------------------- backward ops
def render_image_backward(uv, opacity, conic, rgb, bg, sorted_ids, ranges, n, T, grad_image, width, height,
                          dtype=np.float32, threads=1):
    """Returns fresh (zero-initialised, then accumulated) grad_rgb, grad_opacity, grad_uv, grad_conic."""
    uv, opacity = _a(uv, dtype).reshape(-1, 2), _a(opacity, dtype).reshape(-1)
    conic, rgb = _a(conic, dtype).reshape(-1, 3), _a(rgb, dtype).reshape(-1, 3)
    sorted_ids, ranges = _a(sorted_ids, np.int32), _a(ranges, np.int32)
    n, T, grad_image = _a(n, np.int32), _a(T, dtype), _a(grad_image, dtype)
    m = len(uv)
    g_rgb, g_op = np.zeros((m, 3), dtype), np.zeros(m, dtype)
    g_uv, g_conic = np.zeros((m, 2), dtype), np.zeros((m, 3), dtype)
    _fn("orc_render_image_backward", dtype)(_p(uv), _p(opacity), _p(conic), _p(rgb), _r(bg, dtype), _p(sorted_ids),
                                            _p(ranges), _p(n), _p(T), _p(grad_image), int(width), int(height),
                                            _p(g_rgb), _p(g_op), _p(g_uv), _p(g_conic), int(threads))
    return g_rgb, g_op, g_uv, g_conic


def project_to_screen_backward(xyz_c, proj, uv_grad, width, height, xyz_c_grad=None, dtype=np.float32):
    xyz_c, proj, uv_grad = _a(xyz_c, dtype).reshape(-1, 3), _a(proj, dtype).reshape(16), _a(uv_grad, dtype).reshape(-1, 2)
    out = np.zeros_like(xyz_c) if xyz_c_grad is None else _a(xyz_c_grad, dtype).reshape(-1, 3).copy()
    _fn("orc_project_to_screen_backward", dtype)(_p(xyz_c), _p(proj), _p(uv_grad), len(xyz_c), int(width), int(height),
                                                 _p(out))
    return out


def compute_camera_space_points_backward(xyz_w, view, xyz_c_grad, xyz_w_grad=None, dtype=np.float32):
    xyz_w, view = _a(xyz_w, dtype).reshape(-1, 3), _a(view, dtype).reshape(16)
    xyz_c_grad = _a(xyz_c_grad, dtype).reshape(-1, 3)
    out = np.zeros_like(xyz_w) if xyz_w_grad is None else _a(xyz_w_grad, dtype).reshape(-1, 3).copy()
    _fn("orc_camera_space_points_backward", dtype)(_p(xyz_w), _p(view), _p(xyz_c_grad), len(xyz_w), _p(out))
    return out


def compute_projection_jacobian_backward(xyz_c, focal_x, focal_y, tan_fovx, tan_fovy, J_grad, xyz_c_grad=None,
                                         dtype=np.float32):
    xyz_c, J_grad = _a(xyz_c, dtype).reshape(-1, 3), _a(J_grad, dtype).reshape(-1, 6)
    out = np.zeros_like(xyz_c) if xyz_c_grad is None else _a(xyz_c_grad, dtype).reshape(-1, 3).copy()
    _fn("orc_projection_jacobian_backward", dtype)(_p(xyz_c), _r(focal_x, dtype), _r(focal_y, dtype),
                                                   _r(tan_fovx, dtype), _r(tan_fovy, dtype), _p(J_grad), len(xyz_c),
                                                   _p(out))
    return out


def compute_conic_backward(J, sigma, view, conic, conic_grad, J_grad=None, sigma_grad=None, dtype=np.float32):
    J, sigma, view = _a(J, dtype).reshape(-1, 6), _a(sigma, dtype).reshape(-1, 6), _a(view, dtype).reshape(16)
    conic, conic_grad = _a(conic, dtype).reshape(-1, 3), _a(conic_grad, dtype).reshape(-1, 3)
    jg = np.zeros_like(J) if J_grad is None else _a(J_grad, dtype).reshape(-1, 6).copy()
    sg = np.zeros_like(sigma) if sigma_grad is None else _a(sigma_grad, dtype).reshape(-1, 6).copy()
    _fn("orc_conic_backward", dtype)(_p(J), _p(sigma), _p(view), _p(conic), _p(conic_grad), len(J), _p(jg), _p(sg))
    return jg, sg


def compute_sigma_backward(quaternion, scale, sigma_grad, dtype=np.float32):
    q, s = _a(quaternion, dtype).reshape(-1, 4), _a(scale, dtype).reshape(-1, 3)
    sigma_grad = _a(sigma_grad, dtype).reshape(-1, 6)
    dq, ds = np.empty_like(q), np.empty_like(s)
    _fn("orc_sigma_backward", dtype)(_p(q), _p(s), _p(sigma_grad), len(q), _p(dq), _p(ds))
    return dq, ds


def precompute_spherical_harmonics_backward(xyz, band0, sh, campos, rgb_grad, l_max, xyz_grad=None, dtype=np.float32):
    xyz, band0 = _a(xyz, dtype).reshape(-1, 3), _a(band0, dtype).reshape(-1, 3)
    n, nc = len(xyz), (l_max + 1) ** 2
    sh = _a(sh if sh is not None else np.zeros(1), dtype).reshape(-1)
    campos, rgb_grad = _a(campos, dtype).reshape(3), _a(rgb_grad, dtype).reshape(-1, 3)
    sh_grad = np.zeros((n, max(nc - 1, 0), 3), dtype)
    band0_grad = np.empty((n, 3), dtype)
    xg = np.zeros_like(xyz) if xyz_grad is None else _a(xyz_grad, dtype).reshape(-1, 3).copy()
    shg_buf = sh_grad if sh_grad.size else np.zeros(1, dtype)
    _fn("orc_sh_backward", dtype)(_p(xyz), _p(band0), _p(sh), _p(campos), _p(rgb_grad), int(l_max), n, _p(shg_buf),
                                  _p(band0_grad), _p(xg))
    return sh_grad, band0_grad, xg


def compact_masked_array(src, mask, stride, dtype=np.float32):
    src = _a(src, dtype).reshape(-1)
    mask = np.ascontiguousarray(mask, np.uint8)
    dst = np.empty(int(mask.sum()) * stride, dtype)
    buf = dst if dst.size else np.zeros(1, dtype)
    _fn("orc_compact_masked", dtype)(_p(src if src.size else np.zeros(1, dtype)), _p(mask if mask.size else np.zeros(1, np.uint8)),
                                     len(mask), int(stride), _p(buf))
    return dst


def scatter_masked_array(src, mask, stride, dst, dtype=np.float32):
    src = _a(src, dtype).reshape(-1)
    mask = np.ascontiguousarray(mask, np.uint8)
    dst = _a(dst, dtype).reshape(-1).copy()
    if len(mask):
        _fn("orc_scatter_masked", dtype)(_p(src if src.size else np.zeros(1, dtype)), _p(mask), len(mask), int(stride),
                                         _p(dst if dst.size else np.zeros(1, dtype)))
    return dst


# ------------------------------------------------------------------------- "next" rows f1 / f2
def fused_loss(pred, gt, ssim_weight, dtype=np.float32, threads=1):
    """fused_loss (cuda/loss.cu:430-471): returns (loss, image_grad[H,W,3])."""
    pred, gt = _a(pred, dtype), _a(gt, dtype)
    H, W = pred.shape[0], pred.shape[1]
    grad = np.empty_like(pred)
    loss = _fn("orc_fused_loss", dtype)(_p(pred), _p(gt), H, W, _r(ssim_weight, dtype), _p(grad), int(threads))
    return float(loss), grad


def compute_psnr(pred, gt, dtype=np.float32):
    pred, gt = _a(pred, dtype), _a(gt, dtype)
    return float(_fn("orc_psnr", dtype)(_p(pred), _p(gt), pred.shape[0], pred.shape[1]))


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, b1, b2, eps, bias1, bias2, dtype=np.float32):
    """adam_step (cuda/optimizer.cu:31-44): returns updated (params, exp_avg, exp_avg_sq)."""
    p, g = _a(params, dtype).copy(), _a(grads, dtype)
    m, v = _a(exp_avg, dtype).copy(), _a(exp_avg_sq, dtype).copy()
    _fn("orc_adam", dtype)(_p(p), _p(g), _p(m), _p(v), _r(lr, dtype), _r(b1, dtype), _r(b2, dtype), _r(eps, dtype),
                           _r(bias1, dtype), _r(bias2, dtype), ctypes.c_long(p.size))
    return p, m, v


def set_threads(n):
    """OpenMP threads of the per-gaussian operators (independent per gaussian: same arithmetic at any count).  The
    compositing functions take their own `threads` argument; tile binning is serial at every setting."""
    lib().orc_set_threads(int(n))


def knn_mean_distance(points, k=3, threads=1, kdtree=False):
    """mean distance to the k nearest other points in double (src/gaussian.cpp:66-91): brute force, or (kdtree=True)
    through a kd-tree with leaf size 10 as the reference's nanoflann index does."""
    pts = _a(points, np.float64).reshape(-1, 3)
    out = np.empty(len(pts), np.float32)
    fn = lib().orc_knn_mean_distance_kdtree if kdtree else lib().orc_knn_mean_distance
    fn(_p(pts), ctypes.c_long(len(pts)), int(k), _p(out), int(threads))
    return out


def initialize_gaussians(points, colors, threads=1, kdtree=False):
    """Gaussians::Initialize (src/gaussian.cpp:38-104): dict xyz rgb opacity scale quaternion(w,x,y,z)."""
    pts, col = _a(points, np.float64).reshape(-1, 3), _a(colors, np.uint8).reshape(-1, 3)
    n = len(pts)
    md = knn_mean_distance(pts, 3, threads, kdtree)
    z = lambda *s: np.empty(s, np.float32)
    out = dict(xyz=z(n, 3), rgb=z(n, 3), opacity=z(n), scale=z(n, 3), quaternion=z(n, 4))
    lib().orc_init_attributes(_p(pts), _p(col), _p(md), ctypes.c_long(n), _p(out["xyz"]), _p(out["rgb"]),
                              _p(out["opacity"]), _p(out["scale"]), _p(out["quaternion"]))
    return out


def compute_morton_codes(xyz, bbox_max, bbox_min):
    """compute_morton_codes (cuda/culling.cu:41-63): uint64 codes, bit-exact restatement."""
    p = _a(xyz, np.float32).reshape(-1, 3)
    codes = np.empty(len(p), np.uint64)
    f = ctypes.c_float
    lib().orc_morton_codes(ctypes.c_long(len(p)), _p(p), f(bbox_max[0]), f(bbox_max[1]), f(bbox_max[2]), f(bbox_min[0]),
                           f(bbox_min[1]), f(bbox_min[2]), _p(codes))
    return codes


def clone_split(g, mask, num_sh_coef, split=False, scale_factor=1.0, seed=0):
    """clone_gaussians / split_gaussians (cuda/adaptive_density.cu:12-164) on a dict xyz rgb opacity scale quaternion
    sh; returns the dict of new gaussians (count(mask) rows, twice that for a split)."""
    mask = _a(mask, np.uint8)
    n, copies = len(mask), 2 if split else 1
    wid = (np.cumsum(mask) - mask).astype(np.int32)
    m = int(mask.sum()) * copies
    src = {k: _a(g[k], np.float32) for k in ("xyz", "rgb", "opacity", "scale", "quaternion")}
    sh = _a(g["sh"], np.float32).reshape(n, -1) if num_sh_coef else np.zeros((n, 0), np.float32)
    out = dict(xyz=np.zeros((m, 3), np.float32), rgb=np.zeros((m, 3), np.float32), opacity=np.zeros(m, np.float32),
               scale=np.zeros((m, 3), np.float32), quaternion=np.zeros((m, 4), np.float32),
               sh=np.zeros((m, num_sh_coef * 3), np.float32))
    lib().orc_clone_split(ctypes.c_long(n), int(split), ctypes.c_float(scale_factor), int(num_sh_coef), _p(mask), _p(wid),
                          _p(src["xyz"]), _p(src["rgb"]), _p(src["opacity"]), _p(src["scale"]), _p(src["quaternion"]),
                          _p(sh), _p(out["xyz"]), _p(out["rgb"]), _p(out["opacity"]), _p(out["scale"]),
                          _p(out["quaternion"]), _p(out["sh"]), ctypes.c_ulonglong(seed))
    return out


# ------------------------------------------------------------------------------ sequencing
def rasterize(params, camera, near_thresh, mh_dist, padding, bg, l_max, dtype=np.float32, threads=1):
    """Restates rasterize_image (cuda/raster.cu:12-136).

    params: dict xyz[N,3] rgb[N,3] (=SH band 0) sh[N,(l_max+1)^2-1,3] opacity[N] scale[N,3] quaternion[N,4]
    camera: dict width height fx fy view[16] proj[16] campos[3]
    Returns the ForwardPassData equivalent (compacted order for per-gaussian buffers).
    """
    W, H = int(camera["width"]), int(camera["height"])
    nc = (l_max + 1) ** 2
    xyz = _a(params["xyz"], dtype).reshape(-1, 3)
    xyz_c = compute_camera_space_points(xyz, camera["view"], dtype)
    uv = project_to_screen(xyz_c, camera["proj"], W, H, dtype)
    mask = cull_gaussians(uv, xyz_c, near_thresh, padding, W, H, dtype)
    M = int(mask.sum())
    sel = lambda a, s: _a(a, dtype).reshape(len(mask), s)[mask]
    out = dict(mask=mask, num_culled=M, uv_all=uv, xyz_c_all=xyz_c)
    out["xyz"] = sel(xyz, 3)
    out["uv"], out["xyz_c"] = uv[mask], xyz_c[mask]
    out["band0"], out["opacity"] = sel(params["rgb"], 3), _a(params["opacity"], dtype).reshape(-1)[mask]
    out["quaternion"], out["scale"] = sel(params["quaternion"], 4), sel(params["scale"], 3)
    out["sh"] = _a(params["sh"], dtype).reshape(len(mask), -1)[:, :(nc - 1) * 3][mask] if nc > 1 else None
    out["rgb"] = precompute_spherical_harmonics(out["xyz"], out["sh"], out["band0"], camera["campos"], l_max, dtype)
    out["sigma"] = compute_sigma(out["quaternion"], out["scale"], dtype)
    fx, fy = float(camera["fx"]), float(camera["fy"])
    rt = np.dtype(dtype).type
    tan_fovx = rt(W) / (rt(2.0) * rt(fx))  # cuda/raster.cu:92-93
    tan_fovy = rt(H) / (rt(2.0) * rt(fy))
    out["tan_fovx"], out["tan_fovy"] = float(tan_fovx), float(tan_fovy)
    out["J"], out["conic"], out["radius"] = compute_conic(out["xyz_c"], camera["view"], out["sigma"], fx, fy, tan_fovx,
                                                          tan_fovy, mh_dist, dtype)
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    out["sorted"], out["ranges"], out["num_pairs"] = get_sorted_gaussian_list(out["uv"], out["xyz_c"], out["radius"],
                                                                              ntx, nty, dtype)
    out["n"], out["T"], out["image"] = render_image(out["uv"], out["opacity"], out["conic"], out["rgb"], bg,
                                                    out["sorted"], out["ranges"], W, H, dtype, threads)
    return out


def backward_pass(fwd, camera, grad_image, bg, l_max, dtype=np.float32, threads=1):
    """Restates TrainerImpl::backward_pass after fused_loss (cuda/trainer.cu:941-1012).

    Returns gradients in compacted (post-cull) order, with the same '=' / '+=' chaining.
    """
    W, H = int(camera["width"]), int(camera["height"])
    g = {}
    g["rgb_pre"], g["opacity"], g["uv"], g["conic"] = render_image_backward(
        fwd["uv"], fwd["opacity"], fwd["conic"], fwd["rgb"], bg, fwd["sorted"], fwd["ranges"], fwd["n"], fwd["T"],
        grad_image, W, H, dtype, threads)
    g["sh"], g["band0"], g["xyz"] = precompute_spherical_harmonics_backward(
        fwd["xyz"], fwd["band0"], fwd["sh"], camera["campos"], g["rgb_pre"], l_max, None, dtype)
    g["J"], g["sigma"] = compute_conic_backward(fwd["J"], fwd["sigma"], camera["view"], fwd["conic"], g["conic"],
                                                None, None, dtype)
    rt = np.dtype(dtype).type
    fx, fy = rt(camera["fx"]), rt(camera["fy"])
    # cuda/trainer.cu:992-995: tan(atan(.)) in float
    tan_fovx = np.tan(rt(2.0) * np.arctan(rt(W) / (rt(2.0) * fx)) * rt(0.5))
    tan_fovy = np.tan(rt(2.0) * np.arctan(rt(H) / (rt(2.0) * fy)) * rt(0.5))
    g["xyz_c"] = compute_projection_jacobian_backward(fwd["xyz_c"], fx, fy, tan_fovx, tan_fovy, g["J"], None, dtype)
    g["quaternion"], g["scale"] = compute_sigma_backward(fwd["quaternion"], fwd["scale"], g["sigma"], dtype)
    g["xyz_c"] = project_to_screen_backward(fwd["xyz_c"], camera["proj"], g["uv"], W, H, g["xyz_c"], dtype)
    g["xyz"] = compute_camera_space_points_backward(fwd["xyz"], camera["view"], g["xyz_c"], g["xyz"], dtype)
    return g
