"""Binary hand-over between the Python harness and tests/cpp/reference_host.cpp (the C++ host written against the
reference's headers): one training view's inputs in, the image and the twelve GaussianGradients arrays out.

scene file  : int32[8]  = magic 'GSH1', N, W, H, l_max, cull_mask_padding, 0, 0
              float32[8] = fx, fy, campos[3], near_thresh, mh_dist, bg
              float32 arrays: view[16] proj[16] xyz[N,3] rgb[N,3] sh[N,(l_max+1)^2-1,3] opacity[N] scale[N,3]
              quaternion[N,4] grad_image[H,W,3]                (GaussianParameters layout, cuda_data.cuh:11-16)
result file : int32[8]  = magic 'GHR1', N, M, W, H, l_max, sorted-list capacity, 0
              float32 arrays: image[H,W,3], then in compacted order [M, .]: grad_xyz 3, grad_rgb 3, grad_sh 3((l_max+1)^2-1),
              grad_opacity 1, grad_scale 3, grad_quaternion 4, grad_conic 3, grad_uv 2, grad_J 6, grad_sigma 6,
              grad_xyz_c 3, grad_precompute_rgb 3              (GaussianGradients, cuda_data.cuh:28-36)
"""
import numpy as np

SCENE_MAGIC = 0x31485347   # 'GSH1'
RESULT_MAGIC = 0x31524847  # 'GHR1'
GRADIENTS = (("xyz", 3), ("rgb", 3), ("sh", None), ("opacity", 1), ("scale", 3), ("quaternion", 4), ("conic", 3), ("uv", 2),
             ("J", 6), ("sigma", 6), ("xyz_c", 3), ("precompute_rgb", 3))


def write_host_scene(path, params, cam, grad_image, config, l_max):
    N = int(np.asarray(params["xyz"]).reshape(-1, 3).shape[0])
    W, H = int(cam["width"]), int(cam["height"])
    rest = ((l_max + 1) ** 2 - 1) * 3
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))
    with open(path, "wb") as f:
        np.array([SCENE_MAGIC, N, W, H, l_max, int(config["cull_mask_padding"]), 0, 0], np.int32).tofile(f)
        np.array([cam["fx"], cam["fy"], *[float(c) for c in cam["campos"]], config["near_thresh"], config["mh_dist"],
                  config["bg"]], np.float32).tofile(f)
        f32(cam["view"]).tofile(f)
        f32(cam["proj"]).tofile(f)
        f32(params["xyz"]).tofile(f)
        f32(params["rgb"]).tofile(f)
        sh = np.asarray(params["sh"], np.float32).reshape(N, -1)[:, :rest] if rest else np.zeros((N, 0), np.float32)
        f32(sh).tofile(f)
        f32(params["opacity"]).tofile(f)
        f32(params["scale"]).tofile(f)
        f32(params["quaternion"]).tofile(f)
        g = f32(grad_image)
        assert g.size == W * H * 3
        g.tofile(f)


def read_host_result(path):
    with open(path, "rb") as f:
        head = np.fromfile(f, np.int32, 8)
        if int(head[0]) != RESULT_MAGIC:
            raise ValueError(f"{path} is not a reference_host result file")
        N, M, W, H, L, cap = (int(v) for v in head[1:7])
        out = dict(num_gaussians=N, num_culled=M, width=W, height=H, l_max=L, sorted_capacity=cap)
        out["image"] = np.fromfile(f, np.float32, W * H * 3).reshape(H, W, 3)
        rest = ((L + 1) ** 2 - 1) * 3
        for name, stride in GRADIENTS:
            s = rest if stride is None else stride
            a = np.fromfile(f, np.float32, M * s)
            if a.size != M * s:
                raise ValueError(f"{path}: truncated at grad_{name}")
            out["grad_" + name] = a.reshape(M, s) if s != 1 else a
    return out
