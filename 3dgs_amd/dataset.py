"""ctypes binding of libgsplat_host.so (include/gsplat_host.h): COLMAP binary readers, camera geometry, the config
subset and the PLY writer -- SURVEY.md section 8f row f3.  Mirrors the reference's loader interface
(include/dataloader/colmap.hpp, include/gsplat/utils.hpp): same function names, dicts keyed by id."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
HOST_LIB_PATH = os.path.join(_HERE, "libgsplat_host.so")
_lib = None


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgsplat_host error {code}: {msg}")
        self.code = code


class ColmapCamera(ctypes.Structure):  # gsplat_colmap_camera
    _fields_ = [("id", ctypes.c_int), ("model_id", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("num_params", ctypes.c_int), ("params", ctypes.c_double * 12)]


class ColmapImage(ctypes.Structure):  # gsplat_colmap_image
    _fields_ = [("id", ctypes.c_int), ("camera_id", ctypes.c_int), ("qvec", ctypes.c_double * 4),
                ("tvec", ctypes.c_double * 3), ("name", ctypes.c_char * 1024), ("first_point2d", ctypes.c_uint64),
                ("num_points2d", ctypes.c_uint64)]


class ColmapPoint3D(ctypes.Structure):  # gsplat_colmap_point3d
    _fields_ = [("id", ctypes.c_uint64), ("xyz", ctypes.c_double * 3), ("rgb", ctypes.c_ubyte * 3),
                ("error", ctypes.c_double), ("first_track", ctypes.c_uint64), ("track_length", ctypes.c_uint64)]


_CONFIG_FIELDS = (
    [("dataset_path", ctypes.c_char * 1024), ("output_dir", ctypes.c_char * 1024)] +
    [(k, ctypes.c_int) for k in ("downsample_factor", "print_interval", "num_iters")] + [("ssim_frac", ctypes.c_double)] +
    [(k, ctypes.c_int) for k in ("test_eval_interval", "test_split_ratio")] + [("initial_opacity", ctypes.c_double)] +
    [("initial_scale_num_neighbors", ctypes.c_int)] +
    [(k, ctypes.c_double) for k in ("initial_scale_factor", "max_initial_scale", "near_thresh", "mh_dist")] +
    [("cull_mask_padding", ctypes.c_int)] +
    [(k, ctypes.c_double) for k in ("base_lr", "xyz_lr_multiplier_init", "xyz_lr_multiplier_final", "quat_lr_multiplier",
                                    "scale_lr_multiplier", "opacity_lr_multiplier", "rgb_lr_multiplier",
                                    "sh_lr_multiplier")] +
    [(k, ctypes.c_int) for k in ("use_background", "use_background_end", "reset_opacity_interval")] +
    [("reset_opacity_value", ctypes.c_double)] +
    [(k, ctypes.c_int) for k in ("reset_opacity_start", "reset_opacity_end", "use_sh_precompute", "max_sh_band",
                                 "add_sh_band_interval", "use_split", "use_clone", "use_delete",
                                 "adaptive_control_start", "adaptive_control_end", "adaptive_control_interval",
                                 "max_gaussians")] +
    [(k, ctypes.c_double) for k in ("delete_opacity_threshold", "uv_grad_threshold", "split_scale_factor")])
_BOOL_KEYS = {"use_background", "use_sh_precompute", "use_split", "use_clone", "use_delete"}


class Config(ctypes.Structure):  # gsplat_config
    _fields_ = _CONFIG_FIELDS


_P, _I, _S = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_SP = ctypes.POINTER(ctypes.c_size_t)
HOST_SIGNATURES = {
    "gsplat_host_last_error": (ctypes.c_char_p, []),
    "gsplat_colmap_model_name": (ctypes.c_char_p, [_I]),
    "gsplat_colmap_read_cameras": (_I, [ctypes.c_char_p, _I, _P, _S, _SP]),
    "gsplat_colmap_read_images": (_I, [ctypes.c_char_p, ctypes.c_char_p, _I, _P, _S, _SP, _P, _P, _S, _SP]),
    "gsplat_colmap_read_points3d": (_I, [ctypes.c_char_p, _P, _S, _SP, _P, _P, _S, _SP]),
    "gsplat_qvec_to_rotmat": (None, [_P, _P]),
    "gsplat_camera_position": (None, [_P, _P, _P]),
    "gsplat_scene_extent": (_I, [_P, _P, _S, _P]),
    "gsplat_parse_config": (_I, [ctypes.c_char_p, _P]),
    "gsplat_save_ply": (_I, [ctypes.c_char_p, _S, _I, _P, _P, _P, _P, _P, _P]),
}


def build():
    src = os.path.join(_CSRC, "host", "gs_dataset.cpp")
    hdr = os.path.join(_HERE, "..", "include", "gsplat_host.h")
    if not os.path.exists(HOST_LIB_PATH) or os.path.getmtime(HOST_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _CSRC, "-s", "../libgsplat_host.so"])
    return HOST_LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build() (no Python fallback)")
        lib = ctypes.CDLL(HOST_LIB_PATH)
        for name, (res, args) in HOST_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(code):
    if code != 0:
        raise HostError(code, load().gsplat_host_last_error().decode())


def _enc(path):
    return os.fspath(path).encode()


def ReadCamerasBinary(path, downsample_factor=1):
    """dict id -> {id, model, width, height, params}  (src/colmap.cpp:41-100)"""
    lib, n = load(), ctypes.c_size_t(0)
    _check(lib.gsplat_colmap_read_cameras(_enc(path), downsample_factor, None, 0, ctypes.byref(n)))
    arr = (ColmapCamera * max(n.value, 1))()
    _check(lib.gsplat_colmap_read_cameras(_enc(path), downsample_factor, arr, n.value, ctypes.byref(n)))
    return {c.id: dict(id=c.id, model=lib.gsplat_colmap_model_name(c.model_id).decode(), width=c.width, height=c.height,
                       params=[c.params[k] for k in range(c.num_params)]) for c in arr[:n.value]}


def ReadImagesBinary(path, img_root_dir="", downsample_factor=1):
    """dict id -> {id, qvec, tvec, camera_id, name, xys[P,2], point3D_ids[P]}  (src/colmap.cpp:102-150)"""
    lib, n, p = load(), ctypes.c_size_t(0), ctypes.c_size_t(0)
    root = img_root_dir.encode()
    _check(lib.gsplat_colmap_read_images(_enc(path), root, downsample_factor, None, 0, ctypes.byref(n), None, None, 0,
                                         ctypes.byref(p)))
    arr = (ColmapImage * max(n.value, 1))()
    xys = np.empty((max(p.value, 1), 2), np.float64)
    ids = np.empty(max(p.value, 1), np.int64)
    _check(lib.gsplat_colmap_read_images(_enc(path), root, downsample_factor, arr, n.value, ctypes.byref(n),
                                         xys.ctypes.data, ids.ctypes.data, p.value, ctypes.byref(p)))
    out = {}
    for im in arr[:n.value]:
        a, b = im.first_point2d, im.first_point2d + im.num_points2d
        out[im.id] = dict(id=im.id, qvec=np.array(im.qvec[:]), tvec=np.array(im.tvec[:]), camera_id=im.camera_id,
                          name=im.name.decode(), xys=xys[a:b].copy(), point3D_ids=ids[a:b].copy())
    return out


def ReadPoints3DBinary(path):
    """dict id -> {id, xyz, rgb, error, image_ids, point2D_idxs}  (src/colmap.cpp:152-196)"""
    lib, n, t = load(), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _check(lib.gsplat_colmap_read_points3d(_enc(path), None, 0, ctypes.byref(n), None, None, 0, ctypes.byref(t)))
    arr = (ColmapPoint3D * max(n.value, 1))()
    img = np.empty(max(t.value, 1), np.int32)
    idx = np.empty(max(t.value, 1), np.int32)
    _check(lib.gsplat_colmap_read_points3d(_enc(path), arr, n.value, ctypes.byref(n), img.ctypes.data, idx.ctypes.data,
                                           t.value, ctypes.byref(t)))
    out = {}
    for pt in arr[:n.value]:
        a, b = pt.first_track, pt.first_track + pt.track_length
        out[pt.id] = dict(id=pt.id, xyz=np.array(pt.xyz[:]), rgb=np.array(pt.rgb[:], np.uint8), error=pt.error,
                          image_ids=img[a:b].copy(), point2D_idxs=idx[a:b].copy())
    return out


def ReadPoints3DArrays(path):
    """The same file as flat arrays -- (ids uint64[N], xyz float64[N,3], rgb uint8[N,3]) in file order -- without one
    Python dict per point: what Gaussians::Initialize consumes (src/gaussian.cpp:54-58 copies exactly these fields)."""
    lib, n, t = load(), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _check(lib.gsplat_colmap_read_points3d(_enc(path), None, 0, ctypes.byref(n), None, None, 0, ctypes.byref(t)))
    arr = (ColmapPoint3D * max(n.value, 1))()
    img = np.empty(max(t.value, 1), np.int32)
    idx = np.empty(max(t.value, 1), np.int32)
    _check(lib.gsplat_colmap_read_points3d(_enc(path), arr, n.value, ctypes.byref(n), img.ctypes.data, idx.ctypes.data,
                                           t.value, ctypes.byref(t)))
    P = ColmapPoint3D
    dt = np.dtype(dict(names=["id", "xyz", "rgb"], formats=["<u8", ("<f8", 3), ("u1", 3)],
                       offsets=[P.id.offset, P.xyz.offset, P.rgb.offset], itemsize=ctypes.sizeof(P)))
    rec = np.frombuffer(arr, dtype=dt, count=n.value)
    return rec["id"].copy(), np.ascontiguousarray(rec["xyz"], np.float64), np.ascontiguousarray(rec["rgb"], np.uint8)


def qvec_to_rotmat(qvec):
    q, R = np.ascontiguousarray(qvec, np.float64), np.empty(9, np.float64)
    load().gsplat_qvec_to_rotmat(q.ctypes.data, R.ctypes.data)
    return R.reshape(3, 3)


def camera_position(qvec, tvec):
    q, t, o = np.ascontiguousarray(qvec, np.float64), np.ascontiguousarray(tvec, np.float64), np.empty(3, np.float64)
    load().gsplat_camera_position(q.ctypes.data, t.ctypes.data, o.ctypes.data)
    return o


def computeMaxDiagonal(images):
    """images: the dict ReadImagesBinary returns (src/colmap.cpp:198-236)."""
    q = np.ascontiguousarray([im["qvec"] for im in images.values()], np.float64).reshape(-1, 4)
    t = np.ascontiguousarray([im["tvec"] for im in images.values()], np.float64).reshape(-1, 3)
    out = ctypes.c_double(0.0)
    _check(load().gsplat_scene_extent(q.ctypes.data, t.ctypes.data, len(q), ctypes.byref(out)))
    return out.value


def parseConfig(path):
    """ConfigParameters as a dict (src/utils.cpp:17-87); raises HostError on a missing file or key."""
    c = Config()
    _check(load().gsplat_parse_config(_enc(path), ctypes.byref(c)))
    out = {}
    for name, typ in _CONFIG_FIELDS:
        v = getattr(c, name)
        out[name] = v.decode() if isinstance(v, bytes) else (bool(v) if name in _BOOL_KEYS else v)
    return out


def save_ply(path, xyz, rgb, opacity, scale, quaternion, sh=None):
    """quaternion [N,4] as (w,x,y,z); sh [N,k] flattened per gaussian or None (src/utils.cpp:89-175)."""
    f = lambda a, w: np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, w)) if w else None
    xyz, rgb, scale, quaternion = f(xyz, 3), f(rgb, 3), f(scale, 3), f(quaternion, 4)
    opacity = np.ascontiguousarray(np.asarray(opacity, np.float32).reshape(-1))
    n = len(xyz)
    k = 0 if sh is None else int(np.asarray(sh).size // max(n, 1))
    shf = None if k == 0 else np.ascontiguousarray(np.asarray(sh, np.float32).reshape(n, k))
    _check(load().gsplat_save_ply(_enc(path), n, k, xyz.ctypes.data, rgb.ctypes.data,
                                  None if shf is None else shf.ctypes.data, opacity.ctypes.data, scale.ctypes.data,
                                  quaternion.ctypes.data))
