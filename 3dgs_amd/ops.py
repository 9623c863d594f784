"""Host-side mirror of the reference operator surface on torch device tensors.

Same names, argument order and output semantics ("=" vs "+=") as
/root/reference/include/gsplat_cuda/cuda_forward.cuh:26-131 and cuda_backward.cuh:21-123.
Each function only forwards raw device pointers to the C ABI (include/gsplat_hip.h) on
torch's current HIP stream; torch is used for device memory, nothing else.  Errors come back
as GsplatError (the C++ shim headers in include/gsplat_cuda/ reproduce the reference's
print-and-exit instead).
"""
import ctypes

import torch

from . import _lib
from ._lib import check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch tensor or None")
    if t.numel() == 0:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32")
    return t


def compute_camera_space_points(xyz_w, view, N, xyz_c):
    check(_lib.load().gsplat_compute_camera_space_points(_p(_f32(xyz_w, "xyz_w")), _p(view), N, _p(xyz_c), _stream()))


def project_to_screen(xyz, proj, N, width, height, uv):
    check(_lib.load().gsplat_project_to_screen(_p(_f32(xyz, "xyz")), _p(proj), N, width, height, _p(uv), _stream()))


def cull_gaussians(uv, xyz, N, near_thresh, padding, width, height, mask):
    """mask: torch.bool or uint8 tensor [N]; true = keep."""
    check(_lib.load().gsplat_cull_gaussians(_p(uv), _p(xyz), N, near_thresh, padding, width, height, _p(mask), _stream()))


def compute_sigma(quaternion, scale, N, sigma):
    check(_lib.load().gsplat_compute_sigma(_p(quaternion), _p(scale), N, _p(sigma), _stream()))


def compute_conic(xyz, view, sigma, focal_x, focal_y, tan_fovx, tan_fovy, mh_dist, N, J, conic, radius):
    check(_lib.load().gsplat_compute_conic(_p(xyz), _p(view), _p(sigma), focal_x, focal_y, tan_fovx, tan_fovy, mh_dist,
                                           N, _p(J), _p(conic), _p(radius), _stream()))


def get_sorted_gaussian_list(uv, xyz, radius, n_tiles_x, n_tiles_y, N, count, sorted_gaussians, ranges):
    """Two-call protocol.  ``count`` is the in/out size_t of the reference, passed as an int; returns the count
    (call 1: number of candidate pairs; call 2: unchanged)."""
    c = ctypes.c_size_t(int(count))
    check(_lib.load().gsplat_get_sorted_gaussian_list(_p(uv), _p(xyz), _p(radius), n_tiles_x, n_tiles_y, N,
                                                      ctypes.byref(c), _p(sorted_gaussians), _p(ranges), _stream()))
    return int(c.value)


def precompute_spherical_harmonics(xyz, sh_coefficients, sh_coeffs_band_0, campos, l_max, N, rgb):
    check(_lib.load().gsplat_precompute_spherical_harmonics(_p(xyz), _p(sh_coefficients), _p(sh_coeffs_band_0),
                                                            float(campos[0]), float(campos[1]), float(campos[2]),
                                                            l_max, N, _p(rgb), _stream()))


def render_image(uv, opacity, conic, rgb, background_opacity, sorted_splats, splat_range_by_tile, image_width,
                 image_height, splats_per_pixel, weight_per_pixel, image):
    check(_lib.load().gsplat_render_image(_p(uv), _p(opacity), _p(conic), _p(rgb), background_opacity,
                                          _p(sorted_splats), _p(splat_range_by_tile), image_width, image_height,
                                          _p(splats_per_pixel), _p(weight_per_pixel), _p(image), _stream()))


def project_to_screen_backward(xyz_c, proj, uv_grad_out, N, width, height, xyz_c_grad_in):
    check(_lib.load().gsplat_project_to_screen_backward(_p(xyz_c), _p(proj), _p(uv_grad_out), N, width, height,
                                                        _p(xyz_c_grad_in), _stream()))


def compute_camera_space_points_backward(xyz_w, view, xyz_c_grad_out, N, xyz_w_grad_in):
    check(_lib.load().gsplat_compute_camera_space_points_backward(_p(xyz_w), _p(view), _p(xyz_c_grad_out), N,
                                                                  _p(xyz_w_grad_in), _stream()))


def compute_projection_jacobian_backward(xyz_c, focal_x, focal_y, tan_fovx, tan_fovy, J_grad_out, N, xyz_c_grad_in):
    check(_lib.load().gsplat_compute_projection_jacobian_backward(_p(xyz_c), focal_x, focal_y, tan_fovx, tan_fovy,
                                                                  _p(J_grad_out), N, _p(xyz_c_grad_in), _stream()))


def compute_conic_backward(J, sigma, view, conic, conic_grad_out, N, J_grad_in, sigma_grad_in):
    check(_lib.load().gsplat_compute_conic_backward(_p(J), _p(sigma), _p(view), _p(conic), _p(conic_grad_out), N,
                                                    _p(J_grad_in), _p(sigma_grad_in), _stream()))


def compute_sigma_backward(quaternion, scale, sigma_grad_out, N, quaternion_grad_in, scale_grad_in):
    check(_lib.load().gsplat_compute_sigma_backward(_p(quaternion), _p(scale), _p(sigma_grad_out), N,
                                                    _p(quaternion_grad_in), _p(scale_grad_in), _stream()))


def precompute_spherical_harmonics_backward(xyz_c, rgb_vals, sh_coeffs, campos, rgb_grad_out, l_max, N, sh_grad_in,
                                            sh_grad_band_0_in, xyz_c_grad_in):
    check(_lib.load().gsplat_precompute_spherical_harmonics_backward(
        _p(xyz_c), _p(rgb_vals), _p(sh_coeffs), float(campos[0]), float(campos[1]), float(campos[2]),
        _p(rgb_grad_out), l_max, N, _p(sh_grad_in), _p(sh_grad_band_0_in), _p(xyz_c_grad_in), _stream()))


def render_image_backward(uvs, opacity, conic, rgb, background_opacity, sorted_splats, splat_range_by_tile,
                          num_splats_per_pixel, final_weight_per_pixel, grad_image, image_width, image_height,
                          grad_rgb, grad_opacity, grad_uv, grad_conic):
    check(_lib.load().gsplat_render_image_backward(
        _p(uvs), _p(opacity), _p(conic), _p(rgb), background_opacity, _p(sorted_splats), _p(splat_range_by_tile),
        _p(num_splats_per_pixel), _p(final_weight_per_pixel), _p(grad_image), image_width, image_height, _p(grad_rgb),
        _p(grad_opacity), _p(grad_uv), _p(grad_conic), _stream()))


def fused_loss(predicted_data, gt_data, rows, cols, ssim_weight, image_grad, blocking=True):
    """fused_loss(...) -> loss (cuda_forward.cuh:146-147).  blocking=False skips the read-back and returns None."""
    out = ctypes.c_float(0.0)
    check(_lib.load().gsplat_fused_loss(_p(predicted_data), _p(gt_data), rows, cols, ssim_weight, _p(image_grad),
                                        ctypes.byref(out) if blocking else None, _stream()))
    return float(out.value) if blocking else None


def compute_psnr(predicted_data, gt_data, rows, cols):
    out = ctypes.c_float(0.0)
    check(_lib.load().gsplat_compute_psnr(_p(predicted_data), _p(gt_data), rows, cols, ctypes.byref(out), _stream()))
    return float(out.value)


def adam_step(params, param_grads, exp_avg, exp_avg_sq, lr, b1, b2, eps, bias1, bias2, N, S):
    check(_lib.load().gsplat_adam_step(_p(params), _p(param_grads), _p(exp_avg), _p(exp_avg_sq), lr, b1, b2, eps,
                                       bias1, bias2, N, S, _stream()))


def initialize_gaussians(points_xyz, points_rgb):
    """Gaussians::Initialize (src/gaussian.cpp:38-104) on the GPU.  points_xyz: float64 [N,3] device tensor,
    points_rgb: uint8 [N,3] device tensor.  Returns the parameter dict (xyz rgb opacity scale quaternion)."""
    import torch
    N = int(points_xyz.shape[0])
    assert points_xyz.dtype == torch.float64 and points_rgb.dtype == torch.uint8
    z = lambda *s: torch.empty(*s, dtype=torch.float32, device=points_xyz.device)
    out = dict(xyz=z(N, 3), rgb=z(N, 3), opacity=z(N), scale=z(N, 3), quaternion=z(N, 4))
    check(_lib.load().gsplat_initialize_gaussians(_p(points_xyz.contiguous()), _p(points_rgb.contiguous()), N,
                                                  _p(out["xyz"]), _p(out["rgb"]), _p(out["opacity"]), _p(out["scale"]),
                                                  _p(out["quaternion"]), _stream()))
    return out


def knn_mean_distance(points_xyz, k=3):
    import torch
    N = int(points_xyz.shape[0])
    out = torch.empty(N, dtype=torch.float32, device=points_xyz.device)
    check(_lib.load().gsplat_knn_mean_distance(_p(points_xyz.contiguous()), N, int(k), _p(out), _stream()))
    return out


def compute_morton_codes(N, d_xyz, x_max, y_max, z_max, x_min, y_min, z_min, codes):
    """compute_morton_codes(...) (cuda_forward.cuh:186-188); codes: int64/uint64 device tensor [N]."""
    check(_lib.load().gsplat_compute_morton_codes(N, _p(d_xyz), x_max, y_max, z_max, x_min, y_min, z_min, _p(codes),
                                                  _stream()))


_ATTRS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


def clone_gaussians(N, num_sh_coef, mask, write_ids, src, dst):
    """clone_gaussians(...) (adaptive_density.cuh:27-31); src / dst: dicts xyz rgb opacity scale quaternion sh."""
    check(_lib.load().gsplat_clone_gaussians(N, num_sh_coef, _p(mask), _p(write_ids), *[_p(src.get(k)) for k in _ATTRS],
                                             *[_p(dst.get(k)) for k in _ATTRS], _stream()))


def split_gaussians(N, scale_factor, num_sh_coef, mask, write_ids, src, dst, seed=0):
    """split_gaussians(...) (adaptive_density.cuh:53-57) with an explicit seed."""
    check(_lib.load().gsplat_split_gaussians(N, scale_factor, num_sh_coef, _p(mask), _p(write_ids),
                                             *[_p(src.get(k)) for k in _ATTRS], *[_p(dst.get(k)) for k in _ATTRS],
                                             int(seed), _stream()))


def density_masks(opacity, scale, uv_grad_accum, grad_accum_dur, op_threshold, max_scale, uv_grad_threshold,
                  clone_scale_threshold):
    """Masks of TrainerImpl::adaptive_density_step (cuda/trainer.cu:416-575).
    Returns (prune, clone, split, keep) uint8 tensors and the (pruned, cloned, split) counts."""
    import torch
    N = int(opacity.shape[0])
    m = [torch.empty(N, dtype=torch.uint8, device=opacity.device) for _ in range(4)]
    counts = torch.zeros(3, dtype=torch.int32, device=opacity.device)
    check(_lib.load().gsplat_density_masks(N, _p(opacity), _p(scale), _p(uv_grad_accum), _p(grad_accum_dur),
                                           op_threshold, max_scale, uv_grad_threshold, clone_scale_threshold,
                                           _p(m[0]), _p(m[1]), _p(m[2]), _p(m[3]), _p(counts), _stream()))
    return m[0], m[1], m[2], m[3], [int(c) for c in counts.tolist()]


def expand_sh(sh, l_max_old):
    """add_sh_band's re-layout: [N,(l+1)^2-1,3] -> [N,(l+2)^2-1,3] with the new band zero."""
    import torch
    N = int(sh.shape[0])
    out = torch.empty(N, (l_max_old + 2) ** 2 - 1, 3, dtype=torch.float32, device=sh.device)
    check(_lib.load().gsplat_expand_sh(N, l_max_old, _p(sh) if l_max_old > 0 else None, _p(out), _stream()))
    return out


def gather_rows(src, order):
    """out[i] = src[order[i]] along dim 0 (order: int32 device tensor)."""
    import torch
    out = torch.empty_like(src)
    N = int(src.shape[0])
    stride = src.numel() // N if N else 0
    check(_lib.load().gsplat_gather_rows(N, stride, _p(order), _p(src), _p(out), _stream()))
    return out


def compact_masked_array(stride, d_source, d_mask, num_culled=None):
    """compact_masked_array<STRIDE>(d_source, d_mask, num_culled) -> new tensor [num_culled*stride].  With num_culled
    given the count is trusted, as the reference's template does (cuda_data.cuh:106-127): no read-back, the call stays
    asynchronous; without it the library reads the count back (blocks the host)."""
    N = int(d_mask.numel())
    if num_culled is not None:
        out = torch.empty(max(int(num_culled) * stride, 1), dtype=torch.float32, device=d_mask.device if N else "cuda")
        # the room of `out` is stated: a count that is too small drops rows, it does not write past the tensor (r05)
        check(_lib.load().gsplat_compact_masked_array_bounded(_p(d_source), _p(d_mask), N, stride, _p(out), int(num_culled),
                                                              None, _stream()))
        return out[: int(num_culled) * stride]
    out = torch.empty(max(N * stride, 1), dtype=torch.float32, device=d_mask.device if N else "cuda")
    n = ctypes.c_int(0)
    check(_lib.load().gsplat_compact_masked_array(_p(d_source), _p(d_mask), N, stride, _p(out), ctypes.byref(n),
                                                  _stream()))
    return out[: n.value * stride]


def scatter_masked_array(stride, d_compacted, d_mask, d_destination):
    check(_lib.load().gsplat_scatter_masked_array(_p(d_compacted), _p(d_mask), int(d_mask.numel()), stride,
                                                  _p(d_destination), _stream()))
