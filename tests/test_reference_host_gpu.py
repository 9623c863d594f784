"""The path the REFERENCE host drives, at BASELINE configs[2] (1e6 gaussians, 1920x1080, SH 3).

TrainerImpl::train calls rasterize_image (cuda/raster.cu:12-136) and then TrainerImpl::backward_pass
(cuda/trainer.cu:926-1015): eight compact_masked_array calls, render_image_backward on the raw arrays, and the six
per-gaussian adjoints with zero_grads()-style pre-zeroed `+=` buffers (cuda/trainer.cu:247-261).  The fused entry points
the benchmark times are an additive API; these tests hold the drop-in surface itself to the parity bars at full size:

  * the per-operator C ABI in the trainer's order (Python/ctypes host),
  * get_sorted_gaussian_list's two-call protocol at the full 22 M candidate pairs (bit-exact lists),
  * a C++ host written against include/gsplat_cuda/*.cuh (tests/cpp/reference_host.cpp): rasterize_image shim +
    the same chain through the shim headers, compared with the oracle and timed.
"""
import json
import subprocess

import numpy as np
import pytest

from conftest import assert_grad_close, assert_image_close, pkg

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def test_operator_chain_in_trainer_order_at_config3(gpu, scene, config3_case):
    """Full (non-lean) gsplat_rasterize_image, then exactly cuda/trainer.cu:941-1012 through the stand-alone operators."""
    torch, ops, raster = gpu, pkg("ops"), pkg("raster")
    cs = config3_case
    N, W, H, L, cam, bref = cs["N"], cs["W"], cs["H"], cs["L"], cs["cam"], cs["bref"]
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(cs["params"]), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)  # every ForwardPassData array materialised
    M = fwd["num_culled"]
    assert M == cs["ref"]["num_culled"]
    for k in ("sigma", "conic", "J", "rgb", "uv_all", "xyz_c_all"):
        assert fwd[k] is not None, k
    mask = fwd["mask"]
    n_rest = (L + 1) ** 2 - 1
    # trainer.cu:941-964: the host compacts what the operators read
    sel = lambda src, stride: ops.compact_masked_array(stride, src.reshape(-1), mask, M)
    uv_s, op_s, xyz_c_s = sel(fwd["uv_all"], 2), sel(dp["opacity"], 1), sel(fwd["xyz_c_all"], 3)
    quat_s, scale_s, xyz_s, rgb_s = sel(dp["quaternion"], 4), sel(dp["scale"], 3), sel(dp["xyz"], 3), sel(dp["rgb"], 3)
    sh_s = sel(dp["sh"], 3 * n_rest)
    assert torch.equal(uv_s.reshape(M, 2), fwd["uv"]) and torch.equal(xyz_c_s.reshape(M, 3), fwd["xyz_c"])
    # zero_grads(): trainer.cu:247-261
    z = lambda *s: torch.zeros(*s, device="cuda")
    g = dict(xyz=z(M, 3), rgb=z(M, 3), sh=z(M, n_rest, 3), opacity=z(M), scale=z(M, 3), quaternion=z(M, 4), conic=z(M, 3),
             uv=z(M, 2), J=z(M, 6), sigma=z(M, 6), xyz_c=z(M, 3), precompute_rgb=z(M, 3))
    gi = torch.as_tensor(cs["gi"]).cuda()
    view, proj = dc["view"], dc["proj"]
    ops.render_image_backward(uv_s, op_s, fwd["conic"], fwd["rgb"], c["bg"], fwd["sorted"], fwd["ranges"], fwd["n"],
                              fwd["T"], gi, W, H, g["precompute_rgb"], g["opacity"], g["uv"], g["conic"])
    ops.precompute_spherical_harmonics_backward(xyz_s, rgb_s, sh_s, cam["campos"], g["precompute_rgb"], L, M, g["sh"],
                                                g["rgb"], g["xyz"])
    ops.compute_conic_backward(fwd["J"], fwd["sigma"], view, fwd["conic"], g["conic"], M, g["J"], g["sigma"])
    f32 = np.float32
    tfx = float(np.tan(f32(2) * np.arctan(f32(W) / (f32(2) * f32(cam["fx"]))) * f32(.5)))  # trainer.cu:992-995
    tfy = float(np.tan(f32(2) * np.arctan(f32(H) / (f32(2) * f32(cam["fy"]))) * f32(.5)))
    ops.compute_projection_jacobian_backward(xyz_c_s, cam["fx"], cam["fy"], tfx, tfy, g["J"], M, g["xyz_c"])
    ops.compute_sigma_backward(quat_s, scale_s, g["sigma"], M, g["quaternion"], g["scale"])
    ops.project_to_screen_backward(xyz_c_s, proj, g["uv"], M, W, H, g["xyz_c"])
    ops.compute_camera_space_points_backward(xyz_s, view, g["xyz_c"], M, g["xyz"])
    torch.cuda.synchronize()
    for k, rk in (("xyz", "xyz"), ("rgb", "band0"), ("sh", "sh"), ("opacity", "opacity"), ("scale", "scale"),
                  ("quaternion", "quaternion"), ("conic", "conic"), ("uv", "uv"), ("J", "J"), ("sigma", "sigma"),
                  ("xyz_c", "xyz_c"), ("precompute_rgb", "rgb_pre")):
        got = _np(g[k])
        assert_grad_close(got.reshape(np.asarray(bref[rk]).shape), bref[rk], "operator chain at config 3: grad_" + k)
    # the fused backward of the same forward: the two routes agree far inside the parity bar
    fused = ctx.alloc_gradients(M, L)
    ctx.backward_pass(dp, dc, gi, c["bg"], L, fused)
    for k in ("xyz", "rgb", "sh", "opacity", "scale", "quaternion"):
        assert_grad_close(_np(g[k]), _np(fused[k]), "operator chain vs fused backward: grad_" + k, rel=2e-4)


def test_get_sorted_gaussian_list_two_calls_at_config3(gpu, config3_case):
    """cuda/culling.cu:386-475 at the full size: call 1 reports the ~22 M coarse candidate pairs, call 2 fills the lists;
    same uv / depth / radius in -> the oracle's lists bit for bit."""
    torch, ops = gpu, pkg("ops")
    ref, W, H = config3_case["ref"], config3_case["W"], config3_case["H"]
    M = ref["num_culled"]
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    uv, xyz_c, radius = d(ref["uv"]), d(ref["xyz_c"]), d(ref["radius"])
    count = ops.get_sorted_gaussian_list(uv, xyz_c, radius, ntx, nty, M, 0, None, None)
    assert count == ref["num_pairs"] and count > 20_000_000
    srt = torch.full((count,), -1, dtype=torch.int32, device="cuda")
    ranges = torch.full((ntx * nty + 1,), -1, dtype=torch.int32, device="cuda")
    ops.get_sorted_gaussian_list(uv, xyz_c, radius, ntx, nty, M, count, srt, ranges)
    S = len(ref["sorted"])
    assert (_np(ranges) == ref["ranges"]).all()
    assert (_np(srt[:S]) == ref["sorted"]).all()
    # a second pair of calls on the same buffers (the trainer does this every iteration)
    assert ops.get_sorted_gaussian_list(uv, xyz_c, radius, ntx, nty, M, 0, None, None) == count
    srt.fill_(-1)
    ops.get_sorted_gaussian_list(uv, xyz_c, radius, ntx, nty, M, count, srt, ranges)
    assert (_np(srt[:S]) == ref["sorted"]).all()


def _build_reference_host():
    return pkg("_lib").build_cpp_host("reference_host")


def test_cpp_reference_host_at_config3(gpu, scene, config3_case, tmp_path):
    """tests/cpp/reference_host.cpp is a host written against the reference's headers only (raster.cuh, cuda_data.cuh,
    cuda_backward.cuh): CudaDataManager, rasterize_image(...), zero_grads + the backward_pass chain with the
    compact_masked_array templates.  Its image and all twelve gradient arrays must meet the oracle at full size."""
    scene_io = pkg("scene_io")
    cs = config3_case
    exe = _build_reference_host()
    inp, outp = str(tmp_path / "scene.bin"), str(tmp_path / "result.bin")
    scene_io.write_host_scene(inp, cs["params"], cs["cam"], cs["gi"], scene.CONFIG, cs["L"])
    out = subprocess.run([exe, inp, outp, "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    stats = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    print("[reference host, C++ shims]", stats)
    res = scene_io.read_host_result(outp)
    ref, bref = cs["ref"], cs["bref"]
    assert res["num_culled"] == ref["num_culled"] and stats["num_culled"] == ref["num_culled"]
    assert_image_close(res["image"], ref["image"], "C++ reference host at config 3: image")
    for k, rk in (("xyz", "xyz"), ("rgb", "band0"), ("sh", "sh"), ("opacity", "opacity"), ("scale", "scale"),
                  ("quaternion", "quaternion"), ("conic", "conic"), ("uv", "uv"), ("J", "J"), ("sigma", "sigma"),
                  ("xyz_c", "xyz_c"), ("precompute_rgb", "rgb_pre")):
        want = np.asarray(bref[rk])
        assert_grad_close(res["grad_" + k].reshape(want.shape), want, "C++ reference host at config 3: grad_" + k)
    assert stats["iterations"] == 3 and stats["ms_per_iteration"] > 0
